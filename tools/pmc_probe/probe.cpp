// Minimal HIP program used to check whether `rocprofv3 --pmc` works on the box at all (it crashes inside the profiler
// for every program on this pool's image).
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void axpy(double* y, const double* x, double a, int n) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) y[i] += a * x[i];
}
int main() {
    const int n = 1 << 24;
    double *x, *y;
    if (hipMalloc(&x, n * 8) != hipSuccess || hipMalloc(&y, n * 8) != hipSuccess) return 2;
    (void)hipMemset(x, 0, n * 8);
    (void)hipMemset(y, 0, n * 8);
    for (int r = 0; r < 3; ++r) hipLaunchKernelGGL(axpy, dim3(n / 256), dim3(256), 0, 0, y, x, 2.0, n);
    (void)hipDeviceSynchronize();
    printf("done\n");
    return 0;
}
