// Debug aid (not product): can one wave of a SIMD issue VALU work while its partner wave streams MFMAs?
//   hipcc --offload-arch=gfx950 -O3 -o tools/microbench/overlap tools/microbench/overlap.hip && tools/microbench/overlap
// One workgroup per CU, 8 waves (two per SIMD): waves 0-3 run `nm` x 6 MFMAs (32x32x16 f16, 32 cycles each), waves 4-7
// `nv` x 8 VALU operations; s_memtime (shader clock) around each role, alone and together.  GAP: s_nop cycles the
// MFMA waves put behind every MFMA (so that their next MFMA does not sit in the VALU issue stage waiting for the pipe).
// INTRA: VALU operations interleaved behind every MFMA inside the MFMA waves themselves.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int KIND, int GAP, int INTRA, bool BOTH = false>
__global__ __launch_bounds__(512, 1) void k(int nm, int nv, float* out, unsigned long long* t) {
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    __shared__ float sink[512];
    unsigned long long t0 = 0, t1 = 0;
    __syncthreads();
    float v[8];
    for (int i = 0; i < 8; ++i) v[i] = 0.001f * (lane + i);
    auto valu = [&](int q) {
        if (KIND == 0) v[q] = __builtin_fmaf(v[q], 0.999f, 0.001f);
        if (KIND == 1) v[q] = __builtin_amdgcn_exp2f(v[q]) * 0.25f;
        if (KIND == 2) v[q] = __builtin_amdgcn_rcpf(v[q] + 1.5f);
    };
    if (wave < 4 || BOTH) {
        h8 a, b;
        for (int i = 0; i < 8; ++i) { a[i] = (_Float16)(lane + i); b[i] = (_Float16)(lane - i); }
        f32x16 acc[6] = {};
        t0 = __builtin_amdgcn_s_memtime();
        for (int i = 0; i < nm; ++i) {
#pragma unroll
            for (int q = 0; q < 6; ++q) {
                acc[q] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc[q], 0, 0, 0);
                if (GAP >= 16) asm volatile("s_nop 15");
                if (GAP % 16 == 8) asm volatile("s_nop 7");
                if (GAP % 16 == 12) asm volatile("s_nop 11");
                if (GAP >= 32) asm volatile("s_nop 15");
#pragma unroll
                for (int e = 0; e < INTRA; ++e) valu((q * INTRA + e) & 7);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        float s = 0;
        for (int q = 0; q < 6; ++q) s += acc[q][0] + acc[q][7];
        for (int q = 0; q < 8; ++q) s += v[q];
        t1 = __builtin_amdgcn_s_memtime();
        sink[threadIdx.x] = s;
    } else {
        t0 = __builtin_amdgcn_s_memtime();
        for (int i = 0; i < nv; ++i) {
#pragma unroll
            for (int q = 0; q < 8; ++q) valu(q);
        }
        float s = 0;
        for (int q = 0; q < 8; ++q) s += v[q];
        t1 = __builtin_amdgcn_s_memtime();
        sink[threadIdx.x] = s;
    }
    __syncthreads();
    out[blockIdx.x * 512 + threadIdx.x] = sink[threadIdx.x ^ 1];
    if (blockIdx.x == 3 && lane == 0) t[wave] = t1 - t0;
}

template <int KIND, int INTRA>
void run_both(const char* name, int nm) {
    float* out; unsigned long long* t;
    (void)hipMalloc(&out, 256 * 512 * 4); (void)hipMalloc(&t, 64);
    unsigned long long h[8];
    for (int r = 0; r < 2; ++r) { hipLaunchKernelGGL((k<KIND, 0, INTRA, true>), dim3(256), dim3(512), 0, 0, nm, 0, out, t); (void)hipDeviceSynchronize(); }
    (void)hipMemcpy(h, t, 64, hipMemcpyDeviceToHost);
    printf("%-5s all 8 waves: MFMA + %d VALU behind each   waves 0-3 %8llu  waves 4-7 %8llu  (%.1f cycles per MFMA of the SIMD)\n", name, INTRA, h[0], h[4],
           (double)(h[4] > h[0] ? h[4] : h[0]) / (12.0 * nm));
    (void)hipFree(out); (void)hipFree(t);
}

template <int KIND, int GAP, int INTRA>
void run(const char* name, int nm, int nv) {
    float* out; unsigned long long* t;
    (void)hipMalloc(&out, 256 * 512 * 4); (void)hipMalloc(&t, 64);
    unsigned long long h[8];
    auto go = [&](int a, int b, const char* what) {
        for (int r = 0; r < 2; ++r) { hipLaunchKernelGGL((k<KIND, GAP, INTRA>), dim3(256), dim3(512), 0, 0, a, b, out, t); (void)hipDeviceSynchronize(); }
        (void)hipMemcpy(h, t, 64, hipMemcpyDeviceToHost);
        printf("%-5s gap %2d intra %d %-9s  mfma waves %8llu (%.1f cyc/mfma)   valu waves %8llu (%.1f cyc/op)\n", name, GAP, INTRA, what, h[0],
               a ? (double)h[0] / (6.0 * a) : 0.0, h[4], b ? (double)h[4] / (8.0 * b) : 0.0);
    };
    go(nm, 0, "mfma only"); go(0, nv, "valu only"); go(nm, nv, "together");
    (void)hipFree(out); (void)hipFree(t);
}
int main() {
    run<0, 0, 0>("fma", 20000, 60000);
    run<0, 8, 0>("fma", 20000, 60000);
    run<0, 16, 0>("fma", 20000, 60000);
    run<0, 24, 0>("fma", 20000, 60000);
    run<0, 28, 0>("fma", 20000, 60000);
    run<0, 32, 0>("fma", 20000, 60000);
    run<0, 0, 2>("fma", 20000, 60000);
    run<0, 0, 4>("fma", 20000, 60000);
    run<0, 0, 6>("fma", 20000, 60000);
    run<0, 0, 8>("fma", 20000, 60000);
    run<1, 24, 0>("exp2", 20000, 15000);
    run<1, 0, 2>("exp2", 20000, 15000);
    // both waves of every SIMD run the mixed stream: is the VALU work of one hidden under the MFMAs of the pair?
    run_both<0, 0>("fma", 20000);
    run_both<0, 2>("fma", 20000);
    run_both<0, 4>("fma", 20000);
    run_both<0, 6>("fma", 20000);
    run_both<0, 8>("fma", 20000);
    run_both<0, 12>("fma", 20000);
    run_both<1, 1>("exp2", 20000);
    run_both<1, 2>("exp2", 20000);
    run_both<1, 3>("exp2", 20000);
    run_both<2, 2>("rcp", 20000);
    return 0;
}
