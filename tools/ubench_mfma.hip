// Micro-benchmark (not product code): ceiling of v_mfma_f32_32x32x2_f32 streams shaped like the
// pepper_amd kernels.  hipcc --offload-arch=gfx950 -O3 tools/ubench_mfma.hip -o /tmp/ubench && /tmp/ubench
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
#define MF(a, b, c) __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c, 0, 0, 0)

// MODE 0: registers only. 1: + 4 ds_read_b128 per 16 MFMA. 2: + barrier every 64 MFMA.
// 3: like 1 but reads issued one group ahead (explicit double buffer + sched_barrier).
template <int MODE, int NACC>
__global__ __launch_bounds__(256) void k(float* out, int iters, float seed) {
    __shared__ __attribute__((aligned(16))) float lds[128 * 36 * 2];
    const int lane = threadIdx.x & 63, li = lane & 31, hf = lane >> 5, w = threadIdx.x >> 6;
    for (int i = threadIdx.x; i < 128 * 36 * 2; i += 256) lds[i] = seed * (i & 7);
    __syncthreads();
    f32x16 acc[NACC];
    for (int j = 0; j < NACC; ++j) for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;
    f32x4 a[2] = {{seed, 1, 2, 3}, {seed, 2, 3, 4}}, b[2] = {{1, seed, 1, 1}, {2, 1, seed, 1}};
    f32x4 a2[2], b2[2];
    const float* Ab = lds + ((w >> 1) * 64 + li) * 36 + hf * 4;
    const float* Bb = lds + 128 * 36 + ((w & 1) * 64 + li) * 36 + hf * 4;
    if (MODE == 3) { a[0] = *(const f32x4*)(Ab); a[1] = *(const f32x4*)(Ab + 32 * 36); b[0] = *(const f32x4*)(Bb); b[1] = *(const f32x4*)(Bb + 32 * 36); }
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) {
            if (MODE == 1 || MODE == 2) {
                a[0] = *(const f32x4*)(Ab + kk * 8); a[1] = *(const f32x4*)(Ab + 32 * 36 + kk * 8);
                b[0] = *(const f32x4*)(Bb + kk * 8); b[1] = *(const f32x4*)(Bb + 32 * 36 + kk * 8);
            }
            if (MODE == 3) {
                const int nk = (kk + 1) & 3;
                a2[0] = *(const f32x4*)(Ab + nk * 8); a2[1] = *(const f32x4*)(Ab + 32 * 36 + nk * 8);
                b2[0] = *(const f32x4*)(Bb + nk * 8); b2[1] = *(const f32x4*)(Bb + 32 * 36 + nk * 8);
                __builtin_amdgcn_sched_barrier(0);
            }
#pragma unroll
            for (int s = 0; s < 4; ++s)
#pragma unroll
                for (int m = 0; m < 2; ++m)
#pragma unroll
                    for (int n = 0; n < 2; ++n) acc[(m * 2 + n) % NACC] = MF(a[m][s], b[n][s], acc[(m * 2 + n) % NACC]);
            if (MODE == 3) { __builtin_amdgcn_sched_barrier(0); a[0] = a2[0]; a[1] = a2[1]; b[0] = b2[0]; b[1] = b2[1]; }
        }
        if (MODE == 2) __syncthreads();
    }
    float s = 0;
    for (int j = 0; j < NACC; ++j) for (int r = 0; r < 16; ++r) s += acc[j][r];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}

template <int MODE, int NACC>
void run(const char* name, int grid, float* d) {
    const int iters = 4000;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL((k<MODE, NACC>), dim3(grid), dim3(256), 0, 0, d, 100, 1.0f);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL((k<MODE, NACC>), dim3(grid), dim3(256), 0, 0, d, iters, 1.0f);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    double flops = (double)grid * 4 * iters * 64.0 * 4096.0;
    printf("%-34s grid %4d  %8.3f ms  %7.1f TF\n", name, grid, ms, flops / ms / 1e9);
}

int main() {
    float* d; hipMalloc(&d, 4096 * 256 * 4);
    for (int grid : {256, 512, 1024}) {
        run<0, 4>("regs only, 4 acc", grid, d);
        run<0, 2>("regs only, 2 acc", grid, d);
        run<1, 4>("ds_read_b128 per 16 mfma", grid, d);
        run<3, 4>("ds_read prefetched + pinned", grid, d);
        run<2, 4>("ds_read + barrier per 64 mfma", grid, d);
    }
    return 0;
}
