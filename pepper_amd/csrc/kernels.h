// Internal launcher declarations shared between the kernel translation units and api.hip.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <string>

namespace pa {

// records the thread-local message returned by pa_last_error(); returns `code`
int set_error(int code, const std::string& msg);

enum AKind { A_F32 = 0, A_I8 = 1, A_U8 = 2, A_F32_SCALAR = 3 };

// gemm.hip: C[M,N] = act(A[M,K] W[N,K]^T + bias); act 0 = identity, 1 = SELU.
// a_rpb > 0 remaps logical row m to A + (m / a_rpb) * a_bstride + (m % a_rpb) * lda.
// frag_T > 0 selects the recurrent-seed form: A is [frag_nb, frag_T, K] sequences, M = padded
// batch * frag_T logical rows ordered (32-batch block, step, batch in block), and C is written in
// MFMA fragment order [M/32][N/32][4][64 lanes][4] (see rnn.hip), bias folded in.
hipError_t launch_gemm_nt(int a_type, const void* A, int lda, const float* W, int ldw,
                          const float* bias, float* C, int ldc, int M, int N, int K, int act,
                          int a_rpb, int64_t a_bstride, int frag_T, int frag_nb, hipStream_t stream);

// gemm_h2.hip: the same contraction with both operands in the h2 split format (hi/lo f16 pairs, 4
// bytes per element, see the file header) on the f16 matrix pipe; K % 32 == 0.  lda/ldw/a_bstride
// are in elements as for launch_gemm_nt; a_bytes / w_bytes bound the buffer descriptors (< 4 GiB).
hipError_t launch_gemm_h2(const void* A, int lda, size_t a_bytes, const void* W, int ldw, size_t w_bytes,
                          const float* bias, float* C, int ldc, int M, int N, int K, int act, int a_rpb,
                          int64_t a_bstride, int frag_T, int frag_nb, hipStream_t stream,
                          float* splitk_ws = nullptr, int splits = 1);   // splitk_ws: [splits][M][N] floats
hipError_t launch_f32_to_h2(const float* src, void* dst, int64_t rows, int K, int64_t ld, hipStream_t stream);
void split_h2_host(const float* src, uint32_t* dst, int64_t rows, int K, int64_t ld_src, int64_t ld_dst);

// rnn.hip: Xp [Bpad*T, ldx], Y [Bpad*T, ldy]; Bpad = B rounded up to 64 rows (workspace buffers).
// Packed recurrent weights: [dir][G*H/32][H/8][64 lanes][4] (see pack_rec_weights in api.hip).
hipError_t launch_lstm_rec(int H, const float* Xp, int ldx, const float* Wp, float* Y, int ldy,
                           int B, int T, hipStream_t stream);
// Fused first layer: int8 X [B, T, F] (F <= 32) and Wcat = [W_hh | W_ih zero-padded to 32] packed
// with K = H + 32; bias = b_ih + b_hh [2*4H].
hipError_t launch_lstm_rec_fused(int H, const int8_t* X, int F, const float* bias, const float* Wcat,
                                 float* Y, int ldy, int B, int T, hipStream_t stream);
hipError_t launch_gru_rec(int H, const float* Xp, int ldx, const float* Wp, const float* bhn,
                          const float* h0, int ldh0, float* hn, int ldhn, float* Y, int ldy,
                          int B, int T, hipStream_t stream);

// Fused first GRU layer: uint8 X rows (F <= 16) at X + batch * x_bstride + step * F; Wcat packed
// with K = H + 16 per gate as [W_h* | W_i*]; bias = b_ih + (b_hr, b_hz, 0) [2*3H]; bhn [2*H].
hipError_t launch_gru_rec_fused(int H, const uint8_t* X, int F, int64_t x_bstride, const float* bias,
                                const float* Wcat, const float* bhn, const float* h0, int ldh0, float* hn, int ldhn,
                                float* Y, int ldy, int B, int T, hipStream_t stream);

// rnn_h2.hip: the LSTM step loop on the f16 matrix pipe.  Weights packed by pack_rec_weights_h2
// (wih = {nullptr, nullptr}, KX = 0 for W_hh alone; KX = 32 for the fused first layer).  X != nullptr
// selects the fused form (int8 rows [B, T, F], bias [2*4H]); otherwise Xp seeds the accumulators as in
// launch_lstm_rec.  Y receives the layer output in the h2 split format (ldy in 4-byte elements).
void pack_rec_weights_h2(const float* const whh[2], const float* const wih[2], int G, int H, int F, int KX,
                         uint32_t* out, const float* const* bias = nullptr);   // bias: column H + F (fused layers, F < KX)
size_t rec_weights_h2_words(int G, int H, int KX);
int gru_fused_input_kx(int H, int F);   // 16 / 128: padded width of the uint8-input step loop; 0: projection as a GEMM
// prescaled: weights / bias / Xp were multiplied per gate row by the exp2 constants (see rnn_h2.hip).
hipError_t launch_lstm_rec_h2(int H, const float* Xp, int ldx, const int8_t* X, int F, const float* bias,
                              const void* Wp, void* Y, int ldy, int B, int T, hipStream_t stream,
                              bool prescaled = false, bool small = false);   // small: 32-row workgroups (small calls)

// The step loop of a call of at most 512 windows with a tile's hidden units split over eight workgroups that exchange h_t
// through `exch` every step (rnn_h2.hip lstm_rec_h2_split_kernel); prescaled weights only.  *failed is set when a group
// of workgroups did not meet (not resident together): the caller runs the layer again another way.  sabotage: tests only.
size_t lstm_split_exchange_bytes(int B);
size_t lstm_split_counter_bytes(int B);
// workgroups of a split launch for B windows, and how many of them the current device holds at once (the launch needs all)
int lstm_split_grid(int B);
int lstm_split_resident_workgroups(int ntw);
hipError_t launch_lstm_rec_h2_split(int H, const float* Xp, int ldx, const void* Wp, void* Y, int ldy, int B, int T, void* exch,
                                    void* counters, int* failed, hipStream_t stream, int sabotage = 0);

// LSTM layer fed by an h2 layer output Xh [B*T, 2H]: projection contracted inside the step loop (weights packed
// with pack_rec_weights_h2(..., F = 2H, KX = 2H); bias = b_ih + b_hh).
hipError_t launch_lstm_dec_h2(int H, const void* Xh, int ldxh, const float* bias, const void* Wp, void* Y, int ldy, int B,
                              int T, hipStream_t stream, bool prescaled);
// GRU (H = 128) counterpart; arguments as launch_gru_rec / launch_gru_rec_fused, Y in h2 format.
hipError_t launch_gru_rec_h2(int H, const float* Xp, int ldx, const uint8_t* X, int F, int64_t x_bstride,
                             const float* bias, const void* Wp, const float* bhn, const float* h0, int ldh0, float* hn,
                             int ldhn, void* Y, int ldy, int B, int T, hipStream_t stream);

// GRU layer whose input is an h2 layer output [B*T, 2H]: projection fused into the step loop (weights packed
// with pack_rec_weights_h2(..., F = 2H, KX = 2H); bias = b_ih + (b_hr, b_hz, 0)).
// Small calls (H = 128): 16-row workgroups with the recurrent weights in registers (rnn_h2.hip gru_small_h2_kernel); Xp from
// the projection GEMM (bias folded), Wp from pack_gru_small_weights_h2, Y in h2 format.
void pack_gru_small_weights_h2(const float* const whh[2], int H, uint32_t* out);
size_t gru_small_weights_h2_words(int H);
hipError_t launch_gru_small_h2(int H, const float* Xp, int ldx, const void* Wp, const float* bhn, const float* h0, int ldh0,
                               float* hn, int ldhn, void* Y, int ldy, int B, int T, hipStream_t stream);
hipError_t launch_gru_dec_h2(int H, const void* Xh, int ldxh, const float* bias, const void* Wp, const float* bhn,
                             const float* h0, int ldh0, float* hn, int ldhn, void* Y, int ldy, int B, int T,
                             hipStream_t stream);

// The polish model's last decoder layer with dense1 (2H -> C <= 5 classes) contracted inside the step loop: no layer
// output at all, per-direction partial logits P[dir][batch tile of 128][T][5][128 rows] f32 instead
// (dense_partials_floats(B, T) floats).  Wd: pack_dense_head_h2 fragments.  head.hip's launch_polish_combine consumes P.
void pack_dense_head_h2(const float* W, int C, int H, uint32_t* out);
size_t dense_head_h2_words(int H);
size_t dense_partials_floats(int B, int T);
hipError_t launch_gru_dec_h2_dense(int H, const void* Xh, int ldxh, const float* bias, const void* Wp, const float* bhn,
                                   const float* h0, int ldh0, float* hn, int ldhn, const void* Wd, float* P, int B, int T,
                                   hipStream_t stream);
// head.hip: acc[(b * S + off + t) * C + c] += softmax_c(P[0][..] + P[1][..] + bias) for b < B, t < T
hipError_t launch_polish_combine(const float* P, const float* bias, float* acc, int B, int T, int C, int S, int off,
                                 hipStream_t stream);

// mlp_h2.hip: linear_2..5 (512 -> 512, SELU) + output layer + softmax fused, 64 rows per workgroup.
// scale[NL] receives each layer's power-of-two packing factor; bias for launch_mlp_tail_h2: [NL][512] x scale | [NL] 1 / scale | [NL][512] raw
void pack_mlp_weights_h2(const float* const* W, int NL, uint32_t* out, float* scale);
size_t mlp_weights_h2_words(int NL);
// W32: DEVICE array of the NL f32 weight matrices [512][512] (the exact re-run of a 64-row tile in which an activation left the
// f16 range: >= 65504 or NaN); overflow_rows (device counter, may be null) counts the rows that took it.
hipError_t launch_mlp_tail_h2(const float* X, int ldx, const void* Wp, const float* bias, int NL, const float* Wout,
                              const float* bout, int C, float* probs, float* logits, int n, hipStream_t stream,
                              const float* const* W32, int* overflow_rows);

// head.hip
hipError_t launch_dense_small(int mode, const float* X, int ldx, const float* W, const float* bias,
                              float* out0, float* out1, int rows, int K, int C, int T, int S, int off,
                              hipStream_t stream);
// polish head on an h2 layer output (K = 256, C <= 5): acc[(row / T) * S + off + row % T][c] += softmax(...)
hipError_t launch_polish_dense_acc_h2(const void* X, int ldx, const float* W, const float* bias, float* acc, int rows,
                                      int K, int C, int T, int S, int off, hipStream_t stream);
hipError_t launch_h2_to_f32(void* buf, int64_t rows, int K, int64_t ld, hipStream_t stream);
hipError_t launch_polish_finalize(const float* acc, uint8_t* labels, uint8_t* phred, int64_t B, int S,
                                  int C, int overlap, hipStream_t stream);

// inflate.hip: one wavefront per BGZF block (tables of block b: comp_off/comp_len = its raw DEFLATE bytes in `comp`,
// out_off/out_len = where its ISIZE bytes go in `out`); status[b] != 0 names the block's error (inflate_status_text).
// comp_bytes = the size of `comp`: a block whose 4-byte CRC-32 trailer (behind its DEFLATE bytes) lies inside it has its
// inflated bytes checked against it by the same wavefront; one whose span ends at comp_bytes is not checked
void launch_bgzf_inflate(hipStream_t stream, const uint8_t* comp, const int64_t* comp_off, const int32_t* comp_len,
                         const int64_t* out_off, const int32_t* out_len, uint8_t* out, int32_t* status, int n_blocks,
                         int64_t comp_bytes, unsigned long long* debug_counts = nullptr);
const char* inflate_status_text(int32_t s);
// the BAM records of an inflated span (inflate.hip): entries = record starts (ascending; a lane follows the records from each
// up to the next), slots [n_entries][cap] and out [<= n_entries * cap] of 40-byte pa_record_header, counts [n_entries],
// base [n_entries + 1] (base[n_entries] = the number of records), flags [2] (zeroed by the caller)
void launch_record_walk(hipStream_t stream, const uint8_t* data, int64_t data_bytes, const int64_t* entries, int n_entries, int cap,
                        void* slots, int32_t* counts, int32_t* base, int32_t* flags, void* out, int64_t out_cap, int32_t* tail);
// (out: room for out_cap headers, device or mapped page-locked memory; tail, if not NULL: [0] the number of records, [1..2] the flags)
}  // namespace pa
