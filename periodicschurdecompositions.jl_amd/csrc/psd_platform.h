// Platform layer of the MI355X periodic Schur engine.
//
// The product is built by hipcc for gfx950.  The same sources can additionally be built with g++
// under -DPSD_HOSTSIM into a *test-only* serial simulation of the device code (tests/hostsim/):
// every kernel body is written in "uniform control + PSD_PAR_FOR" style, so the simulation runs
// the identical algorithm text block by block, lane by lane.  The simulation is never shipped,
// never loaded by the package, and exists so that the device state machines can be exercised by
// the CPU-only test tier and by host sanitizers (there is no GPU ASan on the target pool).
#pragma once
#include <cmath>
#include <cstddef>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>

#ifdef PSD_HOSTSIM
// ------------------------------------------------------------------------------------------
#define PSD_HD inline
#define PSD_D inline
struct psd_dim3 {
    int x, y, z;
    psd_dim3(int x_ = 1, int y_ = 1, int z_ = 1) : x(x_), y(y_), z(z_) {}
};
struct psd_simctx {
    psd_dim3 block, grid;
    int nthreads;
    char* lds;
};
extern psd_simctx psd_sim;  // the block currently being simulated (serial)
#define PSD_KERNEL static void
#define PSD_BLOCK_X (psd_sim.block.x)
#define PSD_BLOCK_Y (psd_sim.block.y)
#define PSD_BLOCK_Z (psd_sim.block.z)
#define PSD_GRID_X (psd_sim.grid.x)
#define PSD_NTHREADS (psd_sim.nthreads)
#define PSD_LDS_DECL char* psd_lds = psd_sim.lds
#define PSD_SYNC() ((void)0)
// data-parallel loop over [0,count): iterations must be independent of each other
#define PSD_PAR_FOR(t, count) for (int t = 0; t < (int)(count); ++t)
#define PSD_ONE if (true)
typedef int psd_stream_t;
#define PSD_LAUNCH(kern, grid_, nthreads_, ldsbytes_, stream_, ...)                  \
    do {                                                                              \
        psd_dim3 _g = (grid_);                                                        \
        size_t _lb = (ldsbytes_);                                                      \
        char* _lds = (char*)malloc(_lb ? _lb : 16);                                   \
        psd_sim.grid = _g;                                                            \
        psd_sim.nthreads = (nthreads_);                                               \
        psd_sim.lds = _lds;                                                           \
        for (int _z = 0; _z < _g.z; ++_z)                                             \
            for (int _y = 0; _y < _g.y; ++_y)                                         \
                for (int _x = 0; _x < _g.x; ++_x) {                                   \
                    psd_sim.block = psd_dim3(_x, _y, _z);                             \
                    kern(__VA_ARGS__);                                                \
                }                                                                     \
        free(_lds);                                                                   \
    } while (0)
static inline int psd_rt_malloc(void** p, size_t bytes) {
    *p = malloc(bytes ? bytes : 16);
    return *p ? 0 : 1;
}
static inline void psd_rt_free(void* p) { free(p); }
static inline int psd_rt_memset(void* p, int v, size_t bytes, psd_stream_t) {
    memset(p, v, bytes);
    return 0;
}
static inline int psd_rt_h2d(void* d, const void* h, size_t bytes, psd_stream_t) {
    memcpy(d, h, bytes);
    return 0;
}
static inline int psd_rt_d2h(void* h, const void* d, size_t bytes, psd_stream_t) {
    memcpy(h, d, bytes);
    return 0;
}
static inline int psd_rt_d2d(void* d, const void* s, size_t bytes, psd_stream_t) {
    memmove(d, s, bytes);
    return 0;
}
static inline int psd_rt_sync(psd_stream_t) { return 0; }
static inline int psd_rt_last_error() { return 0; }
#else
// ------------------------------------------------------------------------------------------
#include <hip/hip_runtime.h>
#define PSD_HD __host__ __device__ __forceinline__
#define PSD_D __device__ __forceinline__
typedef dim3 psd_dim3;
#define PSD_KERNEL __global__ void
#define PSD_BLOCK_X ((int)blockIdx.x)
#define PSD_BLOCK_Y ((int)blockIdx.y)
#define PSD_BLOCK_Z ((int)blockIdx.z)
#define PSD_GRID_X ((int)gridDim.x)
#define PSD_NTHREADS ((int)blockDim.x)
#define PSD_LDS_DECL extern __shared__ __attribute__((aligned(16))) char psd_lds[]
#define PSD_SYNC() __syncthreads()
#define PSD_PAR_FOR(t, count) for (int t = (int)threadIdx.x; t < (int)(count); t += (int)blockDim.x)
#define PSD_ONE if (threadIdx.x == 0)
typedef hipStream_t psd_stream_t;
#define PSD_LAUNCH(kern, grid_, nthreads_, ldsbytes_, stream_, ...) \
    hipLaunchKernelGGL(kern, (grid_), dim3(nthreads_), (ldsbytes_), (stream_), __VA_ARGS__)
static inline int psd_rt_malloc(void** p, size_t bytes) { return (int)hipMalloc(p, bytes ? bytes : 16); }
static inline void psd_rt_free(void* p) { (void)hipFree(p); }
static inline int psd_rt_memset(void* p, int v, size_t bytes, psd_stream_t s) {
    return (int)hipMemsetAsync(p, v, bytes, s);
}
static inline int psd_rt_h2d(void* d, const void* h, size_t bytes, psd_stream_t s) {
    return (int)hipMemcpyAsync(d, h, bytes, hipMemcpyHostToDevice, s);
}
static inline int psd_rt_d2h(void* h, const void* d, size_t bytes, psd_stream_t s) {
    return (int)hipMemcpyAsync(h, d, bytes, hipMemcpyDeviceToHost, s);
}
static inline int psd_rt_d2d(void* d, const void* s_, size_t bytes, psd_stream_t s) {
    return (int)hipMemcpyAsync(d, s_, bytes, hipMemcpyDeviceToDevice, s);
}
static inline int psd_rt_sync(psd_stream_t s) { return (int)hipStreamSynchronize(s); }
static inline int psd_rt_last_error() { return (int)hipGetLastError(); }
#endif

// Column-major n x n device matrix view with 1-based access (the reference's indexing, so that
// kernels can cite reference lines directly).
template <typename T>
struct psd_mat {
    T* a;
    int ld;
    PSD_HD T& operator()(int r, int c) const { return a[(size_t)(c - 1) * ld + (r - 1)]; }
};
