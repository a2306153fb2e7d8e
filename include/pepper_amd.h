/* pepper_amd C ABI -- MI355X (gfx950) drop-in for PEPPER's RNN inference hot path.
 *
 * Plain C, opaque handles, int return codes (0 = PA_OK), caller-owned buffers, one HIP stream
 * per handle, no global state besides a thread-local error string.  A handle is
 * thread-compatible, not thread-safe.  Device pointers are ordinary HIP device allocations
 * (e.g. torch.Tensor.data_ptr() of a CUDA/ROCm tensor); no torch types cross this boundary.
 *
 * Each entry point names the reference interface it replaces (paths relative to the
 * kishwarshafin/pepper repository root).
 */
#ifndef PEPPER_AMD_H
#define PEPPER_AMD_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define PA_OK 0
#define PA_ERR_INVALID 1   /* bad argument / missing or mis-shaped tensor */
#define PA_ERR_HIP 2       /* HIP runtime error (message in pa_last_error) */
#define PA_ERR_NO_DEVICE 3 /* no gfx950 device visible */
#define PA_ERR_UNSUPPORTED 4 /* well-formed input this entry point does not take (the message names the one that does) */

/* Thread-local message describing the last non-zero return on this thread. */
const char* pa_last_error(void);
/* "pepper_amd <version> gfx950"; lets a binding check it loaded the right library. */
const char* pa_version(void);
/* Number of visible HIP devices (0 if none / runtime unavailable). */
int pa_device_count(void);

/* ------------------------------------------------------------------------------------------
 * Variant model  (bi-LSTM x2 + 5 x (Linear + SELU) + Linear + Softmax)
 * replaces: pepper_variant/modules/python/models/simple_model.py:6-82  (class TransducerGRU)
 *           and its construction from a checkpoint,
 *           pepper_variant/modules/python/models/ModelHander.py:18-44
 * ------------------------------------------------------------------------------------------ */
typedef struct pa_variant_model pa_variant_model;

typedef struct {
    int32_t image_features;   /* ImageSizeOptions.IMAGE_HEIGHT = 26 (Options.py:6)             */
    int32_t window;           /* CANDIDATE_WINDOW_SIZE + 1 = 33 (Options.py:8)                 */
    int32_t gru_layers;       /* checkpoint['gru_layers'] -> nn.LSTM(num_layers)               */
    int32_t num_classes_type; /* TOTAL_TYPE_LABELS = 3 (Options.py:11)                         */
    int32_t device;           /* HIP device ordinal                                            */
    int32_t max_chunk;        /* windows per device pass (0 = default 16384)                   */
} pa_variant_config;

/* Build a model from a state_dict given as parallel arrays: names[i] is the reference
 * state_dict key ("encoder.weight_ih_l0", ..., "output_layer_type.bias"; a leading "module."
 * is stripped as ModelHander.py:35-39 does), data[i] a HOST pointer to numel[i] float32 values
 * in PyTorch's row-major layout.  Missing keys or wrong sizes fail with PA_ERR_INVALID
 * (load_state_dict(strict=True) semantics).  hip_stream: a hipStream_t to run on, or NULL to
 * create a private stream. */
int pa_variant_create(const pa_variant_config* cfg, const char* const* names,
                      const float* const* data, const int64_t* numel, int32_t n_tensors,
                      void* hip_stream, pa_variant_model** out);
void pa_variant_destroy(pa_variant_model* m);

/* Rows (windows) since creation whose MLP tile was re-run in plain f32 because an activation did not fit the f16-split
 * operand format (|x| >= 65504 or NaN; simple_model.py:60-78 is f32 throughout): diagnostics, results are the f32 ones either way. */
int pa_variant_overflow_rows(pa_variant_model* m, int64_t* rows);
/* Calls of at most 1024 windows run their recurrent layers with a tile's hidden units split over eight (above 512 windows:
 * four) workgroups that meet every step; they must be resident on the GPU together, which other work on the device can prevent.  A call whose
 * workgroups did not meet is run again with the ordinary schedule (same results): this counts those calls since creation
 * (diagnostics; PA_UNIT_SPLIT=0 in the environment at creation switches the split off).  Such calls are synchronous: they
 * return when their results are in the output buffers. */
int pa_variant_split_fallbacks(pa_variant_model* m, int64_t* calls);

/* forward(x, train_mode=False): images int8 [n, window, image_features] (the dtype the images
 * HDF5 stores: pepper_variant/modules/python/DataStore.py:68) -> probs float32 [n, classes].
 * logits (pre-softmax, = forward(x, train_mode=True)) may be NULL.  All pointers are DEVICE
 * pointers; the call is asynchronous on the handle's stream (calls of at most 1024 windows: see
 * pa_variant_split_fallbacks).
 * replaces: simple_model.py:48-82 as called by predict_distributed_gpu.py:58-65. */
int pa_variant_forward_device(pa_variant_model* m, const int8_t* images, int64_t n, float* probs,
                              float* logits);
/* Same with float32 images (the reference feeds FloatTensor; values need not be integral). */
int pa_variant_forward_device_f32(pa_variant_model* m, const float* images, int64_t n,
                                  float* probs, float* logits);
/* Same with HOST pointers, any n: the call is cut into device passes of max_chunk windows and the H2D copy of
 * pass i+1 / the D2H copy of pass i-1 run on their own streams beside the kernels of pass i (asynchronous only from
 * page-locked buffers, see pa_host_register); returns when the results are in `probs`.  A call that fails half way
 * returns only after what it had queued has drained: nothing reads or writes the caller's buffers after the return
 * (the same holds for pa_polish_predict_host / _parts).
 * replaces predict_distributed_gpu.py:60-67 (.cuda() ... .cpu() per batch). */
int pa_variant_forward_host(pa_variant_model* m, const int8_t* images, int64_t n, float* probs,
                            float* logits);

/* ------------------------------------------------------------------------------------------
 * Polish model  (bi-GRU x2 + Linear, sliding windows with hidden-state carry)
 * replaces: pepper/modules/python/models/simple_model.py:5-42 (class TransducerGRU)
 *           pepper/modules/python/models/predict_distributed_cpu.py:43-90 (window loop)
 * ------------------------------------------------------------------------------------------ */
typedef struct pa_polish_model pa_polish_model;

typedef struct {
    int32_t image_features; /* ImageSizeOptions.IMAGE_HEIGHT = 10 (pepper Options.py:2)        */
    int32_t hidden_size;    /* checkpoint['hidden_size'] (TrainOptions.HIDDEN_SIZE = 128)      */
    int32_t gru_layers;     /* checkpoint['gru_layers']                                        */
    int32_t num_classes;    /* TOTAL_LABELS = 5                                                */
    int32_t seq_length;     /* SEQ_LENGTH = 1000                                               */
    int32_t window;         /* TrainOptions.TRAIN_WINDOW = 100                                 */
    int32_t jump;           /* TrainOptions.WINDOW_JUMP = 50                                   */
    int32_t overlap;        /* SEQ_OVERLAP = 50                                                */
    int32_t device;
    int32_t max_chunk;      /* chunks per device pass (0 = default 16384)                       */
} pa_polish_config;

int pa_polish_create(const pa_polish_config* cfg, const char* const* names,
                     const float* const* data, const int64_t* numel, int32_t n_tensors,
                     void* hip_stream, pa_polish_model** out);
void pa_polish_destroy(pa_polish_model* m);

/* One module forward: x float32 [n, T, image_features], hidden float32 [n, 2*layers, H] ->
 * logits float32 [n, T, classes], hidden_out [n, 2*layers, H].  DEVICE pointers.
 * replaces: pepper simple_model.py:27-42 (forward(x, hidden)). */
int pa_polish_forward_device(pa_polish_model* m, const float* x, const float* hidden, int64_t n,
                             int32_t T, float* logits, float* hidden_out);

/* Whole-chunk prediction: images uint8 [n, seq_length, image_features] -> labels uint8
 * [n, seq_length], phred uint8 [n, seq_length]; acc (float32 [n, seq_length, classes], the
 * overlap-added softmax) may be NULL.  DEVICE pointers, asynchronous.
 * replaces: predict_distributed_cpu.py:43-93 (zero hidden, 19 windows, softmax accumulate,
 * max, phred) producing what DataStorePredict.py:70-76 stores as bases / phred_score. */
int pa_polish_predict_device(pa_polish_model* m, const uint8_t* images, int64_t n, uint8_t* labels,
                             uint8_t* phred, float* acc);
int pa_polish_predict_host(pa_polish_model* m, const uint8_t* images, int64_t n, uint8_t* labels,
                           uint8_t* phred, float* acc);
/* The same over n_parts HOST blocks taken as one sequence of chunks (images[p] uint8 [counts[p], seq_length, features] ->
 * labels[p], phred[p] uint8 [counts[p], seq_length]): one series of device passes of up to max_chunk chunks, whatever the
 * blocks' sizes.  A polish pass costs about the same for 2 048 chunks as for 16 384 (one workgroup walks the time steps of
 * 128 chunks), so callers that hold chunks in several buffers -- one per reader process of the HDF5 loop standing in for the
 * reference's DataLoader(num_workers) (predict_distributed_gpu.py:40-47) -- hand them over together. */
int pa_polish_predict_host_parts(pa_polish_model* m, int32_t n_parts, const uint8_t* const* images, const int64_t* counts,
                                 uint8_t* const* labels, uint8_t* const* phred);

/* ------------------------------------------------------------------------------------------
 * Per-kernel timing with HIP events recorded on the handle's own stream (what bench.py's
 * roofline block is computed from).  `model` is a pa_variant_model* or pa_polish_model*.
 * ------------------------------------------------------------------------------------------ */
int pa_profile_enable(void* model, int32_t on);          /* also clears collected samples      */
int pa_profile_count(void* model);                       /* distinct kernel labels so far      */
int pa_profile_get(void* model, int32_t idx, char* label, int32_t label_cap, double* total_ms,
                   int64_t* launches, double* flops);    /* synchronises the stream            */
/* Block until everything queued on the handle's stream has finished. */
int pa_synchronize(void* model);

/* Page-lock / release caller memory (hipHostRegister / hipHostUnregister).  The *_host entry points copy asynchronously --
 * H2D of the next device pass and D2H of the previous one beside the kernels -- only from page-locked buffers; the
 * reference gets the same effect from DataLoader(pin_memory=True).  Used for the shared-memory slots of the reader /
 * writer lanes (pepper_amd/hostpipe.py), which a torch allocation cannot provide. */
int pa_host_register(void* ptr, int64_t bytes);
int pa_host_unregister(void* ptr);

#ifdef __cplusplus
}
#endif
#endif /* PEPPER_AMD_H */
