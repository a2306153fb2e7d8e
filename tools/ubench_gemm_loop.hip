// Micro-benchmark (not product code): which ingredient of the LDS-staged GEMM loop costs MFMA issue.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
#define MF(a, b, c) __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c, 0, 0, 0)
constexpr int LDT = 36;

// FLAGS bit0: ds_write staging (8 x b128 / k-tile), bit1: global loads (8 x dwordx4 / k-tile),
// bit2: loads predicated per row (exec branches), bit3: double-buffered LDS (one barrier), else 2 barriers
template <int FLAGS>
__global__ __launch_bounds__(256) void k(const float* __restrict__ A, float* out, int iters, int M) {
    __shared__ __attribute__((aligned(16))) float lds[2 * 2 * 128 * LDT];
    const int tid = threadIdx.x, lane = tid & 63, li = lane & 31, hf = lane >> 5, w = tid >> 6;
    for (int i = tid; i < 2 * 2 * 128 * LDT; i += 256) lds[i] = 0.001f * (i & 7);
    __syncthreads();
    f32x16 acc[4];
    for (int j = 0; j < 4; ++j) for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;
    const int kq = tid & 7, r0 = tid >> 3;
    f32x4 ra[4], rb[4];
    for (int i = 0; i < 4; ++i) { ra[i] = f32x4{1, 2, 3, 4}; rb[i] = f32x4{4, 3, 2, 1}; }
    const float* rowp[4];
    for (int i = 0; i < 4; ++i) { int m = blockIdx.x * 128 + r0 + 32 * i; rowp[i] = (!(FLAGS & 4) || m < M) ? A + (size_t)(m % 4096) * 512 : nullptr; }
    for (int it = 0; it < iters; ++it) {
        const int buf = (FLAGS & 8) ? (it & 1) : 0;
        if (FLAGS & 1) {
            for (int i = 0; i < 4; ++i) {
                *(f32x4*)&lds[((buf ^ 1) * 256 + r0 + 32 * i) * LDT + kq * 4] = ra[i];
                *(f32x4*)&lds[((buf ^ 1) * 256 + 128 + r0 + 32 * i) * LDT + kq * 4] = rb[i];
            }
        }
        if (FLAGS & 16) {   // direct global -> LDS DMA, 8 x 1 KiB per wave per k-tile, lane-linear LDS image
            const int kofs = (it & 15) * 32 + kq * 4;
            for (int i = 0; i < 4; ++i) {
                float* dstA = &lds[((buf ^ 1) * 256) * 32 + (i * 256 + (tid & ~63)) * 4];
                float* dstB = &lds[((buf ^ 1) * 256 + 128) * 32 + (i * 256 + (tid & ~63)) * 4];
                __builtin_amdgcn_global_load_lds(rowp[i] + kofs, (__attribute__((address_space(3))) void*)dstA, 16, 0, 0);
                __builtin_amdgcn_global_load_lds(rowp[i] + 2048 * 512 + kofs, (__attribute__((address_space(3))) void*)dstB, 16, 0, 0);
            }
        }
        if (FLAGS & 2) {
            const int kofs = (it & 15) * 32 + kq * 4;
            for (int i = 0; i < 4; ++i) {
                if (!(FLAGS & 4) || rowp[i] != nullptr) { ra[i] = *(const f32x4*)(rowp[i] + kofs); rb[i] = *(const f32x4*)(rowp[i] + 2048 * 512 + kofs); }
            }
        }
        const float* Ab = lds + (buf * 256 + (w >> 1) * 64 + li) * LDT + hf * 4;
        const float* Bb = lds + (buf * 256 + 128 + (w & 1) * 64 + li) * LDT + hf * 4;
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) {
            f32x4 a[2], b[2];
            a[0] = *(const f32x4*)(Ab + kk * 8); a[1] = *(const f32x4*)(Ab + 32 * LDT + kk * 8);
            b[0] = *(const f32x4*)(Bb + kk * 8); b[1] = *(const f32x4*)(Bb + 32 * LDT + kk * 8);
#pragma unroll
            for (int s = 0; s < 4; ++s)
#pragma unroll
                for (int m = 0; m < 2; ++m)
#pragma unroll
                    for (int n = 0; n < 2; ++n) acc[m * 2 + n] = MF(a[m][s], b[n][s], acc[m * 2 + n]);
        }
        __syncthreads();
        if (!(FLAGS & 8)) __syncthreads();
    }
    float s = 0;
    for (int j = 0; j < 4; ++j) for (int r = 0; r < 16; ++r) s += acc[j][r];
    for (int i = 0; i < 4; ++i) s += ra[i].x + rb[i].y;
    out[blockIdx.x * 256 + tid] = s;
}

template <int FLAGS>
void run(const char* name, int grid, const float* A, float* d) {
    const int iters = 4000;
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    hipLaunchKernelGGL((k<FLAGS>), dim3(grid), dim3(256), 0, 0, A, d, 100, 1 << 30);
    (void)hipDeviceSynchronize();
    (void)hipEventRecord(e0);
    hipLaunchKernelGGL((k<FLAGS>), dim3(grid), dim3(256), 0, 0, A, d, iters, 1 << 30);
    (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    double flops = (double)grid * 4 * iters * 64.0 * 4096.0;
    printf("%-44s grid %4d  %8.3f ms  %7.1f TF\n", name, grid, ms, flops / ms / 1e9);
}

int main() {
    float *d, *A; (void)hipMalloc(&d, 4096 * 256 * 4); (void)hipMalloc(&A, (size_t)4096 * 512 * 4 * 2); (void)hipMemset(A, 0, (size_t)4096 * 512 * 4 * 2);
    for (int grid : {256, 512}) {
        run<8>("reads + 1 barrier (dbuf)", grid, A, d);
        run<0>("reads + 2 barriers", grid, A, d);
        run<8 | 1>("+ ds_write staging", grid, A, d);
        run<8 | 2>("+ global loads (no staging)", grid, A, d);
        run<8 | 1 | 2>("+ ds_write + global loads", grid, A, d);
        run<8 | 1 | 2 | 4>("+ ds_write + predicated global loads", grid, A, d);
        run<8 | 16>("+ global_load_lds DMA staging", grid, A, d);
    }
    return 0;
}
