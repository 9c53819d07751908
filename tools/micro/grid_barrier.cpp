// Microbenchmark: cost of a software grid barrier (agent-scope atomic counter + release/acquire fences) against the
// cost of a kernel boundary on the same stream, for the grid shape of the Hessenberg chain launches (257 x 256).
// Every spin loop is bounded (a timeout sets an abort flag and the kernel drains).
// build: hipcc --offload-arch=gfx950 -O3 tools/micro/grid_barrier.cpp -o tools/micro/grid_barrier
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

__global__ void __launch_bounds__(256) k_barrier(int* counter, int* abortf, double* data, int iters, int nblocks) {
    const int tid = threadIdx.x, b = blockIdx.x;
    int target = 0;
    double acc = 0.0;
    for (int it = 0; it < iters; ++it) {
        // a little dependent traffic through memory, as the chain has: every block writes one value, reads a neighbour's
        if (tid == 0) data[(it & 1) * nblocks + b] = acc + 1.0;
        __syncthreads();
        target += nblocks;
        if (tid == 0) {
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
            __hip_atomic_fetch_add(counter, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            int spins = 0;
            while (__hip_atomic_load(counter, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) {
                if (++spins > (1 << 22) || __hip_atomic_load(abortf, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) {
                    __hip_atomic_store(abortf, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    break;
                }
                __builtin_amdgcn_s_sleep(1);
            }
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        }
        __syncthreads();
        if (__hip_atomic_load(abortf, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) return;
        if (tid == 0) acc = __builtin_nontemporal_load(&data[(it & 1) * nblocks + (b + 1) % nblocks]);
    }
    if (tid == 0) data[2 * nblocks + b] = acc;
}

__global__ void __launch_bounds__(256) k_step(double* data, int it, int nblocks) {
    const int tid = threadIdx.x, b = blockIdx.x;
    if (tid == 0) {
        const double acc = (it > 0) ? data[((it - 1) & 1) * nblocks + (b + 1) % nblocks] : 0.0;
        data[(it & 1) * nblocks + b] = acc + 1.0;
    }
}

int main(int argc, char** argv) {
    const int nblocks = argc > 1 ? atoi(argv[1]) : 257, iters = argc > 2 ? atoi(argv[2]) : 2000;
    int *counter, *abortf;
    double* data;
    hipMalloc(&counter, 4);
    hipMalloc(&abortf, 4);
    hipMalloc(&data, sizeof(double) * 3 * nblocks);
    hipStream_t s;
    hipStreamCreate(&s);
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    for (int rep = 0; rep < 3; ++rep) {
        hipMemsetAsync(counter, 0, 4, s);
        hipMemsetAsync(abortf, 0, 4, s);
        hipMemsetAsync(data, 0, sizeof(double) * 3 * nblocks, s);
        hipEventRecord(e0, s);
        hipLaunchKernelGGL(k_barrier, dim3(nblocks), dim3(256), 0, s, counter, abortf, data, iters, nblocks);
        hipEventRecord(e1, s);
        hipStreamSynchronize(s);
        float ms = 0;
        hipEventElapsedTime(&ms, e0, e1);
        int ab = 0;
        double last = 0;
        hipMemcpy(&ab, abortf, 4, hipMemcpyDeviceToHost);
        hipMemcpy(&last, data + 2 * nblocks, 8, hipMemcpyDeviceToHost);
        printf("grid barrier: %d blocks, %d iterations: %.3f ms = %.3f us per barrier (abort %d, chain value %.0f of %d)\n", nblocks, iters,
               ms, 1e3 * ms / iters, ab, last, iters);
    }
    for (int rep = 0; rep < 3; ++rep) {
        hipMemsetAsync(data, 0, sizeof(double) * 3 * nblocks, s);
        hipEventRecord(e0, s);
        for (int it = 0; it < iters; ++it) hipLaunchKernelGGL(k_step, dim3(nblocks), dim3(256), 0, s, data, it, nblocks);
        hipEventRecord(e1, s);
        hipStreamSynchronize(s);
        float ms = 0;
        hipEventElapsedTime(&ms, e0, e1);
        printf("kernel boundary: %d blocks, %d launches: %.3f ms = %.3f us per launch\n", nblocks, iters, ms, 1e3 * ms / iters);
    }
    return 0;
}
