#include <hip/hip_runtime.h>
__global__ void k(int* out) {
    int v = threadIdx.x * 3 + 1;
    int r = __builtin_amdgcn_update_dpp(-7, v, 0x138, 0xf, 0xf, false);
    out[threadIdx.x] = r;
}
int main() {
    int* d; hipMalloc(&d, 256); k<<<1, 64>>>(d); int h[64]; hipMemcpy(h, d, 256, hipMemcpyDeviceToHost);
    int bad = 0; for (int i = 0; i < 64; ++i) { int want = i == 0 ? -7 : (i - 1) * 3 + 1; if (h[i] != want) { ++bad; } }
    printf("wave_shr bad=%d h0=%d h1=%d h32=%d h63=%d\n", bad, h[0], h[1], h[32], h[63]);
    return 0;
}
