#include <hip/hip_runtime.h>
__global__ void k(int* out) {
    int v = threadIdx.x;
    int r = __builtin_amdgcn_update_dpp(0, v, 0x134, 0xf, 0xf, false);
    int s = __builtin_amdgcn_update_dpp(0, v, 0x130, 0xf, 0xf, false);
    out[threadIdx.x] = r; out[64 + threadIdx.x] = s;
}
int main() { int* d; hipMalloc(&d, 128 * 4); k<<<1, 64>>>(d); int h[128]; hipMemcpy(h, d, 512, hipMemcpyDeviceToHost);
  for (int i = 0; i < 128; ++i) printf("%d ", h[i]); printf("\n"); return 0; }
