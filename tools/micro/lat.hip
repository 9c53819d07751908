// Dependent-chain latencies of the instructions on the chase's critical path (one wavefront alone on its CU).
#include <hip/hip_runtime.h>
#include <cstdio>
#define N 4096
template <int K>
__global__ void lat(double* out, long long* cyc, double a, double b) {
    double x = a + threadIdx.x * 1e-9;
    float xf = (float)x;
    long long t0 = __builtin_amdgcn_s_memtime();
    if (K == 0) { for (int i = 0; i < N; ++i) x = __builtin_fma(x, a, b); }
    if (K == 1) { for (int i = 0; i < N; ++i) x = __builtin_amdgcn_rcp(x) + b; }
    if (K == 2) { for (int i = 0; i < N; ++i) x = __builtin_amdgcn_rsq(x) + b; }
    if (K == 3) { for (int i = 0; i < N; ++i) { xf = __builtin_amdgcn_rcpf(xf) + 1.0f; } x = xf; }
    if (K == 4) { for (int i = 0; i < N; ++i) { x = (double)__builtin_amdgcn_rcpf((float)x) + b; } }
    if (K == 5) { for (int i = 0; i < N; ++i) { int lo = __builtin_amdgcn_readlane(__double2loint(x), 3), hi = __builtin_amdgcn_readlane(__double2hiint(x), 3); x = __hiloint2double(hi, lo) + b; } }
    if (K == 6) { for (int i = 0; i < N; ++i) x = x * a; }
    if (K == 7) { for (int i = 0; i < N; ++i) x = x + b; }
    if (K == 8) { __shared__ double sh[64]; sh[threadIdx.x] = x; for (int i = 0; i < N; ++i) { sh[threadIdx.x] = x; asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); x = sh[(threadIdx.x + 1) & 63] + b; } }
    if (K == 9) { double y = b; for (int i = 0; i < N; ++i) { x = __builtin_fma(x, a, b); y = __builtin_fma(y, a, b); } x += y; }
    if (K == 10) { __shared__ double sh[256]; for (int i = 0; i < N; ++i) { x = __builtin_fma(x, a, b); sh[threadIdx.x] = x; sh[threadIdx.x + 64] = x; sh[threadIdx.x + 128] = x; } x += sh[(threadIdx.x + 1) & 63]; }
    if (K == 11) { for (int i = 0; i < N; ++i) { x = __builtin_fma(x, a, b); if (threadIdx.x == 5) { double2 v; v.x = x; v.y = b; ((double2*)out)[64 + 2 * (i & 255)] = v; ((double2*)out)[65 + 2 * (i & 255)] = v; } } }
    if (K == 12) { for (int i = 0; i < N; ++i) { x = __builtin_fma(x, a, b); if (threadIdx.x == 5) { out[64 + (i & 255)] = x; } } }
    if (K == 13) { for (int i = 0; i < N; ++i) { x = __builtin_fma(x, a, b); __builtin_amdgcn_s_barrier(); } }
    if (K == 14) { for (int i = 0; i < N; ++i) { x = __builtin_fma(x, a, b); x += (double)(__builtin_amdgcn_s_memtime() & 1); } }
    long long t1 = __builtin_amdgcn_s_memtime();
    out[threadIdx.x] = x;
    if (threadIdx.x == 0) cyc[0] = t1 - t0;
}
int main() {
    double* out; long long* cyc; hipMalloc(&out, 1 << 16); hipMalloc(&cyc, 8);
    const char* names[] = {"v_fma_f64 dependent", "v_rcp_f64 + add", "v_rsq_f64 + add", "v_rcp_f32 + add (f32 chain)", "cvt + v_rcp_f32 + cvt + add", "2 v_readlane + add", "v_mul_f64 dependent", "v_add_f64 dependent", "ds_write + ds_read + add", "two independent fma chains (per pair)", "fma + 3 ds_write_b64", "fma + 2 global_store_dwordx4 (one lane)", "fma + 1 global_store_dwordx2 (one lane)", "fma + s_barrier (lone wave)", "fma + s_memtime"};
#define RUN(K) { hipLaunchKernelGGL(lat<K>, dim3(1), dim3(64), 0, 0, out, cyc, 0.999, 0.5); hipLaunchKernelGGL(lat<K>, dim3(1), dim3(64), 0, 0, out, cyc, 0.999, 0.5); long long h; hipMemcpy(&h, cyc, 8, hipMemcpyDeviceToHost); printf("%-40s %6.1f cycles per iteration\n", names[K], (double)h / N); }
    RUN(0) RUN(1) RUN(2) RUN(3) RUN(4) RUN(5) RUN(6) RUN(7) RUN(8) RUN(9) RUN(10) RUN(11) RUN(12) RUN(13) RUN(14)
    return 0;
}
