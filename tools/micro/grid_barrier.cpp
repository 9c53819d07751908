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

// Dataflow variant (no fences, no read-modify-write atomics): block b writes its value and then a tag (= iteration) with
// agent-scope stores; thread t of every block polls tag[t] with agent-scope loads until all nblocks tags carry the
// iteration, then reads a neighbour's value (agent-scope load).  This is what a chain launch that waits for its
// predecessor's partial norms would do.  tags: [2][nblocks] by iteration parity, vals likewise.
__global__ void __launch_bounds__(256) k_flow(unsigned long long* tags, double* vals, int* abortf, double* out, int iters, int nblocks) {
    const int tid = threadIdx.x, b = blockIdx.x;
    double acc = 0.0;
    for (int it = 1; it <= iters; ++it) {
        unsigned long long* tg = tags + (size_t)(it & 1) * 512;
        double* vl = vals + (size_t)(it & 1) * 512;
        if (tid == 0) {
            __hip_atomic_store(vl + b, acc + 1.0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __hip_atomic_store(tg + b, (unsigned long long)it, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        int spins = 0;
        for (;;) {
            bool ok = true;
            for (int q = tid; q < nblocks; q += 256)
                ok = ok && (__hip_atomic_load(tg + q, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == (unsigned long long)it);
            if (__syncthreads_and(ok ? 1 : 0)) break;
            if (++spins > (1 << 20)) {
                if (tid == 0) __hip_atomic_store(abortf, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                return;
            }
        }
        if (tid == 0) acc = __hip_atomic_load(vl + (b + 1) % nblocks, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    if (tid == 0) out[b] = acc;
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
    if (nblocks <= 512) {
        unsigned long long* tags;
        double *vals, *out;
        hipMalloc(&tags, 8 * 1024);
        hipMalloc(&vals, 8 * 1024);
        hipMalloc(&out, 8 * 512);
        for (int rep = 0; rep < 3; ++rep) {
            hipMemsetAsync(tags, 0, 8 * 1024, s);
            hipMemsetAsync(vals, 0, 8 * 1024, s);
            hipMemsetAsync(abortf, 0, 4, s);
            hipEventRecord(e0, s);
            hipLaunchKernelGGL(k_flow, dim3(nblocks), dim3(256), 0, s, tags, vals, abortf, out, iters, nblocks);
            hipEventRecord(e1, s);
            hipStreamSynchronize(s);
            float ms = 0;
            hipEventElapsedTime(&ms, e0, e1);
            int ab = 0;
            double last = 0;
            hipMemcpy(&ab, abortf, 4, hipMemcpyDeviceToHost);
            hipMemcpy(&last, out, 8, hipMemcpyDeviceToHost);
            printf("dataflow tags: %d blocks, %d iterations: %.3f ms = %.3f us per step (abort %d, chain value %.0f of %d)\n", nblocks, iters, ms,
                   1e3 * ms / iters, ab, last, iters);
        }
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
