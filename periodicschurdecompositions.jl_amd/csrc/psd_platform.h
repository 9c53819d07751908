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
#define PSD_D_NOINLINE inline
#define PSD_KEEP(x) ((void)0)
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
#define PSD_KERNEL_B(nt) static void
#define PSD_BLOCK_X (psd_sim.block.x)
#define PSD_BLOCK_Y (psd_sim.block.y)
#define PSD_BLOCK_Z (psd_sim.block.z)
#define PSD_GRID_X (psd_sim.grid.x)
#define PSD_GRID_Y (psd_sim.grid.y)
#define PSD_NTHREADS (psd_sim.nthreads)
#define PSD_LDS_DECL char* psd_lds = psd_sim.lds
#define PSD_SYNC() ((void)0)
#define PSD_WAVE_SYNC() ((void)0)
// two-wave workgroups of the chase kernels (block = 64 x 2): the simulation runs both roles one after the other
#define PSD_WAVE_ROLE 0
#define PSD_GLOBAL(T, ptr) (ptr)
#define PSD_PAIR_BARRIER() ((void)0)
#define PSD_PAIR_BARRIER_KEEP(n) ((void)0)
#define PSD_PAIR_BARRIER_BARE() ((void)0)
// data-parallel loop over [0,count): iterations must be independent of each other
#define PSD_PAR_FOR(t, count) for (int t = 0; t < (int)(count); ++t)
#define PSD_ONE if (true)
// Single-pass data-parallel region (count <= PSD_NTHREADS) with per-lane variables that live across
// regions and can be broadcast from a given lane (v_readlane on the GPU, array slot here).
#define PSD_MAXLANES 64
#define PSD_PAR_ONCE(t, count) for (int t = 0; t < (int)(count); ++t)
// every lane of a 64-lane block, no bound check
#define PSD_PAR_ALL64(t) for (int t = 0; t < 64; ++t)
// the wavefronts of a workgroup as independent workers (serial here; they touch disjoint data between two PSD_SYNCs)
#define PSD_WAVES_FOR(g, G) for (int g = 0; g < (int)(G); ++g)
#define PSD_LANEVAR(type, name) type name[PSD_MAXLANES]
#define PSD_LANEVAR_REF(type, name) type* name
#define PSD_LV(name) name[t]
#define PSD_BCAST(name, lane) (name[lane])
#define PSD_BCASTZ(name, lane) (name[lane])
// 1/x and (sqrt(s), 1/sqrt(s)) for arguments known to be normal and far from the range limits
static inline double psd_rcp_fast(double x) { return 1.0 / x; }
static inline void psd_sqrt_pair_fast(double s, double& g, double& rg) {
    g = sqrt(s);
    rg = 1.0 / g;
}
static inline void psd_rsqrt2_fast(double a, double b, double& ra, double& rb) {
    ra = 1.0 / sqrt(a);
    rb = 1.0 / sqrt(b);
}
static inline long long psd_clock() { return 0; }
static inline long long psd_wallclock() { return 0; }
// cross-workgroup words (agent-scope atomics on the GPU; the simulation runs the workgroups one after the other)
static inline int psd_atomic_add(int* q, int v) { const int o = *q; *q = o + v; return o; }
static inline long long psd_atomic_add_ll(long long* q, long long v) { const long long o = *q; *q = o + v; return o; }
static inline int psd_atomic_cas(int* q, int expect, int v) { const int o = *q; if (o == expect) *q = v; return o; }
static inline int psd_atomic_load(const int* q) { return *q; }
static inline void psd_atomic_store(int* q, int v) { *q = v; }
static inline int psd_atomic_max(int* q, int v) { const int o = *q; if (v > o) *q = v; return o; }
static inline void psd_release_fence() {}
static inline void psd_acquire_fence() {}
typedef int psd_stream_t;
#define PSD_LAUNCH(kern, grid_, nthreads_, ldsbytes_, stream_, ...)                  \
    do {                                                                              \
        psd_dim3 _g = (grid_);                                                        \
        size_t _lb = (ldsbytes_);                                                      \
        char* _lds = (char*)malloc(_lb ? _lb : 16);                                   \
        memset(_lds, 0xFF, _lb ? _lb : 16); /* LDS is not zero on the GPU: poison with NaNs */ \
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
// (a block of nx x ny threads is nx simulated lanes: the second wavefront of a chase workgroup has no separate existence here)
#define PSD_LAUNCH2(kern, grid_, nx_, ny_, ldsbytes_, stream_, ...) PSD_LAUNCH(kern, grid_, nx_, ldsbytes_, stream_, __VA_ARGS__)
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
#define PSD_D_NOINLINE __device__ __attribute__((noinline))
// pins the value at this point of the instruction stream (keeps the optimiser from sinking its computation into a
// later branch)
#define PSD_KEEP(x) asm volatile("" : "+v"(x))
typedef dim3 psd_dim3;
#define PSD_KERNEL __global__ void
#define PSD_KERNEL_B(nt) __global__ void __launch_bounds__(nt)
#define PSD_BLOCK_X ((int)blockIdx.x)
#define PSD_BLOCK_Y ((int)blockIdx.y)
#define PSD_BLOCK_Z ((int)blockIdx.z)
#define PSD_GRID_X ((int)gridDim.x)
#define PSD_GRID_Y ((int)gridDim.y)
#define PSD_NTHREADS ((int)blockDim.x)
#define PSD_LDS_DECL extern __shared__ __attribute__((aligned(16))) char psd_lds[]
// Workgroups with blockDim.x <= 64 are single wavefronts as far as the code written in terms of PSD_TID / PSD_SYNC is
// concerned (blockDim.y = 2 adds a helper wavefront that such code never sees: the chase kernels, PSD_WAVE_ROLE): their
// hand-offs through LDS and memory need the counters drained, not a hardware barrier.
__device__ __forceinline__ void psd_sync() {
    if (blockDim.x > 64) __syncthreads();
    else asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
}
#define PSD_SYNC() psd_sync()
#define PSD_WAVE_ROLE ((int)threadIdx.y)
// a pointer known to point to device memory (one read from a structure is generic: stores through it would be FLAT
// instructions, which count against the LDS counter as well)
#define PSD_GLOBAL(T, ptr) ((__attribute__((address_space(1))) T*)(ptr))
// the hardware barrier of the two wavefronts of a chase workgroup (everything either stored to LDS before it is visible
// to the other behind it)
#define PSD_PAIR_BARRIER()                                      \
    do {                                                        \
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");      \
        __builtin_amdgcn_s_barrier();                           \
        asm volatile("" ::: "memory");                          \
    } while (0)
// the rendezvous alone (the caller argues which of its LDS operations are complete)
#define PSD_PAIR_BARRIER_BARE()                                 \
    do {                                                        \
        asm volatile("" ::: "memory");                          \
        __builtin_amdgcn_s_barrier();                           \
        asm volatile("" ::: "memory");                          \
    } while (0)
// the same, but the n youngest LDS operations of this wavefront — loads it issued AFTER its last store, for its own next
// step — may still be in flight behind the barrier (DS operations of a wave complete in order: the stores are out)
#define PSD_PAIR_BARRIER_KEEP(n)                                \
    do {                                                        \
        asm volatile("s_waitcnt lgkmcnt(" #n ")" ::: "memory"); \
        __builtin_amdgcn_s_barrier();                           \
        asm volatile("" ::: "memory");                          \
    } while (0)
// LDS hand-off between lanes of ONE wavefront (only valid in single-wave workgroups): DS operations
// of a wave execute in order, so only the compiler has to be fenced; unlike __syncthreads() this does
// not drain outstanding global stores (vmcnt), which would put their acknowledge latency on the chain.
// (measured: dropping the s_waitcnt and keeping only the compiler fence is bit-identical on the whole GPU test tier
// and 1.4 % faster; the conservative form stays)
#define PSD_WAVE_SYNC() asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory")
// PSD_TID / PSD_TSTRIDE: the thread index the data-parallel macros run over.  Default: the workgroup.  A header that
// is included a second time with PSD_TID = lane of the wavefront, PSD_TSTRIDE = 64 yields wave-scoped copies of its
// functions, which the wavefronts of a multi-wave workgroup call independently (psd_rgz.h, namespace psd_wv).
#define PSD_TID_BLOCK ((int)threadIdx.x)
#define PSD_TSTRIDE_BLOCK ((int)blockDim.x)
#define PSD_TID_WAVE ((int)threadIdx.x & 63)
#define PSD_TSTRIDE_WAVE 64
#define PSD_TID PSD_TID_BLOCK
#define PSD_TSTRIDE PSD_TSTRIDE_BLOCK
#define PSD_PAR_FOR(t, count) for (int t = PSD_TID; t < (int)(count); t += PSD_TSTRIDE)
#define PSD_ONE if (PSD_TID == 0)
#define PSD_MAXLANES 64
#define PSD_PAR_ONCE(t, count) if (const int t = PSD_TID; t < (int)(count))
#define PSD_PAR_ALL64(t) if (const int t = PSD_TID; true)
// the wavefronts of a workgroup as independent workers: g = wave index, runs for g < G
#define PSD_WAVES_FOR(g, G) if (const int g = (int)threadIdx.x >> 6; g < (int)(G))
#define PSD_LANEVAR(type, name) type name = type()
#define PSD_LANEVAR_REF(type, name) type name
#define PSD_LV(name) name
#define PSD_BCAST(name, lane) psd_readlane_f64(name, lane)
#define PSD_BCASTZ(name, lane) zmk(psd_readlane_f64((name).re, lane), psd_readlane_f64((name).im, lane))
__device__ __forceinline__ double psd_readlane_f64(double v, int lane) {
    const int lo = __builtin_amdgcn_readlane(__double2loint(v), lane);
    const int hi = __builtin_amdgcn_readlane(__double2hiint(v), lane);
    return __hiloint2double(hi, lo);
}
// 1/x and (sqrt(s), 1/sqrt(s)) for arguments known to be normal and far from the range limits:
// hardware seed (v_rcp_f64 / v_rsq_f64) + two Newton steps, without the scale/fixup wrapper the
// IEEE division and sqrt expansions carry (that wrapper is most of their latency).
__device__ __forceinline__ double psd_rcp_fast(double x) {
    double r = __builtin_amdgcn_rcp(x);
    double e = __builtin_fma(-x, r, 1.0);
    r = __builtin_fma(r, e, r);
    e = __builtin_fma(-x, r, 1.0);
    r = __builtin_fma(r, e, r);
    return r;
}
__device__ __forceinline__ void psd_sqrt_pair_fast(double s, double& g, double& rg) {
    const double y = __builtin_amdgcn_rsq(s);
    double gg = s * y, h = 0.5 * y;
    double r = __builtin_fma(-h, gg, 0.5);
    gg = __builtin_fma(gg, r, gg);
    h = __builtin_fma(h, r, h);
    r = __builtin_fma(-h, gg, 0.5);
    gg = __builtin_fma(gg, r, gg);
    h = __builtin_fma(h, r, h);
    // final correction of the root: g += (s - g*g) * h
    const double d = __builtin_fma(-gg, gg, s);
    gg = __builtin_fma(d, h, gg);
    g = gg;
    rg = h + h;
}
// 1/sqrt(a) and 1/sqrt(b) together (same preconditions): seeds + two Newton steps each, the two chains written
// interleaved (a lone wavefront issues a dependent f64 operation every 6.4 cycles: two independent chains fill the gaps)
__device__ __forceinline__ void psd_rsqrt2_fast(double a, double b, double& ra, double& rb) {
    double ya = __builtin_amdgcn_rsq(a), yb = __builtin_amdgcn_rsq(b);
    double ta = a * ya, tb = b * yb, ha = 0.5 * ya, hb = 0.5 * yb;
    double ea = __builtin_fma(-ta, ya, 1.0), eb = __builtin_fma(-tb, yb, 1.0);
    ya = __builtin_fma(ha, ea, ya);
    yb = __builtin_fma(hb, eb, yb);
    ta = a * ya; tb = b * yb; ha = 0.5 * ya; hb = 0.5 * yb;
    ea = __builtin_fma(-ta, ya, 1.0);
    eb = __builtin_fma(-tb, yb, 1.0);
    ra = __builtin_fma(ha, ea, ya);
    rb = __builtin_fma(hb, eb, yb);
}
// Cross-workgroup words of one launch (slot roles, done flags, global counters): agent-scope atomics.  Data handed
// from one workgroup to another INSIDE a launch goes behind a release fence on the producer and an acquire fence on
// the consumer (MI355X: per-XCD L2s are not coherent, a CU's L1 is never refreshed by other CUs' stores).
__device__ __forceinline__ int psd_atomic_add(int* q, int v) {
    return __hip_atomic_fetch_add(q, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ long long psd_atomic_add_ll(long long* q, long long v) {
    return __hip_atomic_fetch_add(q, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ int psd_atomic_cas(int* q, int expect, int v) {
    __hip_atomic_compare_exchange_strong(q, &expect, v, __ATOMIC_RELAXED, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    return expect;
}
__device__ __forceinline__ int psd_atomic_load(const int* q) {
    return __hip_atomic_load(q, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ void psd_atomic_store(int* q, int v) {
    __hip_atomic_store(q, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ int psd_atomic_max(int* q, int v) {
    return __hip_atomic_fetch_max(q, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ void psd_release_fence() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}
__device__ __forceinline__ void psd_acquire_fence() {
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}
__device__ __forceinline__ long long psd_clock() { return (long long)__builtin_amdgcn_s_memtime(); }
__device__ __forceinline__ long long psd_wallclock() { return (long long)__builtin_amdgcn_s_memrealtime(); }
typedef hipStream_t psd_stream_t;
#define PSD_LAUNCH(kern, grid_, nthreads_, ldsbytes_, stream_, ...) \
    hipLaunchKernelGGL(kern, (grid_), dim3(nthreads_), (ldsbytes_), (stream_), __VA_ARGS__)
#define PSD_LAUNCH2(kern, grid_, nx_, ny_, ldsbytes_, stream_, ...) \
    hipLaunchKernelGGL(kern, (grid_), dim3(nx_, ny_), (ldsbytes_), (stream_), __VA_ARGS__)
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

// State that one workgroup publishes for ANOTHER workgroup to pick up in a later launch (cursor states of a multishift
// train): a sequence lock on an epoch word.  The writer raises PSD_EPOCH_NEVER, writes, then stores the tick; a reader
// takes the block only if the epoch is older than its own launch and unchanged after the copy, so it can never act on a
// block that is being written in the launch it runs in (the workgroups of a launch are not ordered against each other).
#ifndef PSD_EPOCH_NEVER
#define PSD_EPOCH_NEVER 0x7fffffff
#endif
PSD_D void psd_pub_begin(int* ep) {
    psd_atomic_store(ep, PSD_EPOCH_NEVER);
    psd_release_fence();
}
PSD_D void psd_pub_end(int* ep, int tick) {
    psd_release_fence();
    psd_atomic_store(ep, tick);
}
template <class S>
PSD_D bool psd_pub_read(const int* ep, int tick, const S* src, S& dst) {
    const int e1 = psd_atomic_load(ep);
    if (e1 >= tick) return false;
    psd_acquire_fence();
    dst = *src;
    psd_acquire_fence();
    return psd_atomic_load(ep) == e1;
}

// A window emitted more transforms for one owner than its list holds (PSD_*TR_CAP): the lists would be applied
// truncated.  The window kernels never do that by construction (<= 2 records per chase position and owner, windows of
// <= 28 positions), but a change of a width or a capacity must fail loudly, not return wrong factors with info = 0:
// the state machine stops with this code, the host maps it to PSD_INFO_RUNTIME + 77.
#define PSD_LIST_OVERFLOW (-7777)
PSD_D bool psd_list_overflow(const int* lcnt, int p, int cap) {  // (every lane, after the barrier that published lcnt)
    bool over = false;
    for (int m = 0; m < p; ++m)
        if (lcnt[m] > cap) over = true;
    return over;
}
