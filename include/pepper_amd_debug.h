/* pepper_amd diagnostic entry points -- NOT part of the drop-in surface (include/pepper_amd.h and its siblings are).
 * libpepper_amd.so exports these four symbols for the tuning tools and one test; nothing in the product path calls them and
 * a caller of the reference's interfaces never needs them.  They replace nothing in the reference.
 */
#ifndef PEPPER_AMD_DEBUG_H
#define PEPPER_AMD_DEBUG_H

#ifdef __cplusplus
extern "C" {
#endif

/* Per-phase cycle stamps of the last LSTM / GRU step-loop launches (tools/phase_timing.py, tools/phase_timing_gru.py):
 * host_out receives 2 * 8 * 80 * 2 (LSTM) / 128 (GRU) 64-bit values; 1: stamps were never enabled, 2: the copy failed. */
int pa_debug_dump_timing(unsigned long long* host_out);
int pa_debug_dump_gru_timing(unsigned long long* host_out);
/* The split-f16 GEMM on host operands (tests/test_gpu_gemm_h2.py, tools/bench_gemm_h2*.py): A [a_rows, K] and W [N, K] in f32 ->
 * h2 form -> gemm_h2 -> C [M, N]; `iters` launches between two HIP events, their mean in *ms_out.  _experiment selects a
 * scheduling variant of the kernel under test (0: the product's). */
void pa_debug_gemm_h2_experiment(int e);
int pa_debug_gemm_h2(const float* A, const float* W, const float* bias, float* C, int a_rows, int M, int N, int K, int act,
                     int frag_T, int frag_nb, int iters, float* ms_out);

#ifdef __cplusplus
}
#endif
#endif
