// Micro-benchmark (not product code): v_mfma_f32_32x32x16_f16 on gfx950 -- denormal handling of the
// f16 inputs, peak issue rate, and how many LDS reads / VALU ops / global loads ride along per MFMA.
// hipcc --offload-arch=gfx950 -O3 tools/ubench_f16.hip -o tools/ubench_f16.so && tools/ubench_f16.so
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cmath>
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
#define MF(a, b, c) __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0)

__global__ void denorm_kernel(float* out, uint16_t abits, uint16_t bbits) {
    union { uint16_t u; _Float16 h; } ua{abits}, ub{bbits};
    h8 a, b;
    for (int i = 0; i < 8; ++i) { a[i] = ua.h; b[i] = ub.h; }
    f32x16 c;
    for (int r = 0; r < 16; ++r) c[r] = 0.f;
    c = MF(a, b, c);
    if (threadIdx.x == 0) out[0] = c[0];
}

// One k16 step of a wave tile with NT C tiles = 3 * NT MFMAs (hi*hi, hi*lo, lo*hi per tile; consecutive
// MFMAs go to different accumulators).  Riders per step: NDS ds_read_b128, NVALU v_fma, NVM global
// dwordx4 loads, spread evenly between the MFMAs with sched_group_barrier.
template <int I, int N, int NDS, int NVALU, int NVM>
__device__ __forceinline__ void spread() {
    if constexpr (I < N) {
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
        constexpr int d = (I + 1) * NDS / N - I * NDS / N, v = (I + 1) * NVALU / N - I * NVALU / N,
                      g = (I + 1) * NVM / N - I * NVM / N;
        if constexpr (d > 0) __builtin_amdgcn_sched_group_barrier(0x100, d, 0);
        if constexpr (g > 0) __builtin_amdgcn_sched_group_barrier(0x020, g, 0);
        if constexpr (v > 0) __builtin_amdgcn_sched_group_barrier(0x002, v, 0);
        spread<I + 1, N, NDS, NVALU, NVM>();
    }
}

template <int NT, int NDS, int NVALU, int NVM>
__global__ __launch_bounds__(256) void k(float* out, const f32x4* __restrict__ gsrc, int iters, float seed) {
    __shared__ __attribute__((aligned(16))) _Float16 lds[4 * 128 * 72];   // 73.7 KB
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    for (int i = threadIdx.x; i < 4 * 128 * 72; i += 256) lds[i] = (_Float16)(seed * (i & 7));
    __syncthreads();
    f32x16 acc[NT];
    for (int j = 0; j < NT; ++j) for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;
    constexpr int NF = NDS > 8 ? NDS : 8;
    h8 f[NF];
    for (int i = 0; i < NF; ++i) for (int e = 0; e < 8; ++e) f[i][e] = (_Float16)(seed * e + i);
    const _Float16* base = lds + (w * 32 + (lane & 31)) * 72 + (lane >> 5) * 8;
    float v[8];
    for (int i = 0; i < 8; ++i) v[i] = seed + i;
    f32x4 g = {0, 0, 0, 0};
    const f32x4* gp = gsrc + (blockIdx.x * 256 + threadIdx.x);
    for (int it = 0; it < iters; ++it) {
        h8 nf[NDS > 0 ? NDS : 1];
#pragma unroll
        for (int d = 0; d < NDS; ++d) nf[d] = *(const h8*)(base + (d & 3) * 128 * 72 + (d >> 2) * 16 + (it & 1) * 8);
        f32x4 ng[NVM > 0 ? NVM : 1];
#pragma unroll
        for (int d = 0; d < NVM; ++d) ng[d] = gp[((it * NVM + d) * 65536) & 0xfffff];
#pragma unroll
        for (int d = 0; d < NVALU; ++d) v[d & 7] = __builtin_fmaf(v[d & 7], 1.0001f, v[(d + 3) & 7]);
#pragma unroll
        for (int term = 0; term < 3; ++term)
#pragma unroll
            for (int t = 0; t < NT; ++t)
                acc[t] = MF(f[(t + term) % NF], f[(t * 3 + term + 1) % NF], acc[t]);
#pragma unroll
        for (int d = 0; d < NDS; ++d) f[d % NF] = nf[d];
#pragma unroll
        for (int d = 0; d < NVM; ++d) g += ng[d];
        spread<0, 3 * NT, NDS, NVALU, NVM>();
    }
    float s = g.x + g.y + g.z + g.w;
    for (int i = 0; i < 8; ++i) s += v[i];
    for (int j = 0; j < NT; ++j) for (int r = 0; r < 16; ++r) s += acc[j][r];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}

template <int NT, int NDS, int NVALU, int NVM>
void run(const char* name, int grid, float* d, const f32x4* g) {
    const int iters = 2000;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL((k<NT, NDS, NVALU, NVM>), dim3(grid), dim3(256), 0, 0, d, g, 50, 1.0f);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL((k<NT, NDS, NVALU, NVM>), dim3(grid), dim3(256), 0, 0, d, g, iters, 1.0f);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double mfmas = (double)grid * 4 * iters * NT * 3;
    const double flops = mfmas * 32768.0;
    // cycles per MFMA per SIMD assuming 2.4 GHz and waves spread evenly over 1024 SIMDs
    const double waves_per_simd = grid * 4 / 1024.0;
    const double cyc = ms * 1e-3 * 2.4e9 / ((double)iters * NT * 3 * (waves_per_simd < 1 ? 1 : waves_per_simd));
    printf("%-44s grid %4d  %8.3f ms  %7.1f TF(f16)  %6.1f TF(f32-equiv)  %5.1f cyc/mfma\n", name, grid, ms,
           flops / ms / 1e9, flops / 3 / ms / 1e9, cyc);
}

int main() {
    float* d; hipMalloc(&d, 4096 * 256 * 4);
    f32x4* g; hipMalloc(&g, (size_t)(1 << 21) * 16); hipMemset(g, 0, (size_t)(1 << 21) * 16);
    // denormals: 0x0010 = 2^-20 (subnormal), 0x3c00 = 1.0, 0x0400 = 2^-14 (smallest normal)
    float h[1];
    struct { uint16_t a, b; const char* what; double want; } cases[] = {
        {0x0010, 0x3c00, "subnormal 2^-20 x 1.0 (x16)", 16 * std::ldexp(1.0, -20)},
        {0x3c00, 0x0010, "1.0 x subnormal 2^-20 (x16)", 16 * std::ldexp(1.0, -20)},
        {0x0001, 0x3c00, "min subnormal 2^-24 x 1.0 (x16)", 16 * std::ldexp(1.0, -24)},
        {0x0400, 0x0400, "2^-14 x 2^-14 (x16)", 16 * std::ldexp(1.0, -28)},
        {0x0010, 0x0010, "2^-20 x 2^-20 (x16)", 16 * std::ldexp(1.0, -40)},
    };
    for (auto& c : cases) {
        hipLaunchKernelGGL(denorm_kernel, dim3(1), dim3(64), 0, 0, d, c.a, c.b);
        hipMemcpy(h, d, 4, hipMemcpyDeviceToHost);
        printf("denorm: %-34s got %.6e want %.6e %s\n", c.what, h[0], c.want, h[0] == (float)c.want ? "PRESERVED" : "DIFFERENT");
    }
    for (int grid : {256, 512}) {
        run<8, 0, 0, 0>("regs only, 8 tiles (24 mfma/step)", grid, d, g);
        run<16, 0, 0, 0>("regs only, 16 tiles (48 mfma/step)", grid, d, g);
        run<8, 12, 0, 0>("8 tiles: 12 ds_read_b128 /step", grid, d, g);
        run<8, 16, 0, 0>("8 tiles: 16 ds_read_b128 /step", grid, d, g);
        run<16, 16, 0, 0>("16 tiles: 16 ds_read_b128 /step", grid, d, g);
        run<8, 0, 24, 0>("8 tiles: 24 valu /step", grid, d, g);
        run<8, 0, 48, 0>("8 tiles: 48 valu /step", grid, d, g);
        run<8, 0, 96, 0>("8 tiles: 96 valu /step", grid, d, g);
        run<8, 0, 192, 0>("8 tiles: 192 valu /step", grid, d, g);
        run<8, 0, 0, 4>("8 tiles: 4 global x4 /step", grid, d, g);
        run<8, 0, 0, 8>("8 tiles: 8 global x4 /step", grid, d, g);
        run<8, 12, 24, 4>("8 tiles: 12 ds + 24 valu + 4 vmem", grid, d, g);
        run<16, 16, 32, 8>("16 tiles: 16 ds + 32 valu + 8 vmem", grid, d, g);
    }
    return 0;
}
