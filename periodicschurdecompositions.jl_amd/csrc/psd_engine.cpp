// Host side of libpsd_mi355x: launch sequencing and the C ABI (include/psd_mi355x.h).
// Built by hipcc (-x hip, gfx950).  tests/hostsim builds the same file with g++ -DPSD_HOSTSIM as a
// serial simulation for the CPU-only test tier; the package never loads that build.
#include <atomic>
#include "psd_hess.h"
#include "psd_hess2.h"
#include "psd_formq2.h"
#include "psd_real_qr.h"
#include "psd_apply2.h"
#include "psd_zhess.h"
#include "psd_zhess2.h"
#include "psd_zqz.h"
#include "psd_zord.h"
#include "psd_rord.h"
#include "psd_rgz.h"
#include "psd_zgz.h"
#include "psd_zgord.h"
#include "psd_grord.h"
#include "psd_rhessx.h"
#include "psd_check.h"

#include "../../include/psd_mi355x.h"

#include <chrono>
#include <new>
#include <vector>

#ifdef PSD_HOSTSIM
psd_simctx psd_sim;
#define PSD_LIB_FLAVOUR "hostsim"
#else
#ifdef PSD_DIAG
#define PSD_LIB_FLAVOUR "hip-gfx950, diagnostic build"
#else
#define PSD_LIB_FLAVOUR "hip-gfx950"
#endif
#endif

#define PSD_CHECK(expr)                                  \
    do {                                                 \
        int _e = (int)(expr);                            \
        if (_e != 0) return PSD_INFO_RUNTIME + (_e & 0xffff); \
    } while (0)

// Environment switches (INTEGRATION.md lists them).
//  * psd_env(): selectors between code paths that ALL return a correct decomposition (the tests use them to reach every
//    path of the product library).
//  * psd_env_diag(): diagnostics, timing experiments and tuning knobs, some of which void the results (the Hessenberg chain
//    without its panel updates, the panel kernel run on finished matrices, a hand-over that never validates).  They exist
//    only in a build with -DPSD_DIAG: libpsd_mi355x_diag.so, built beside the product library for tools/ and the
//    fault-injection test, and the test-only serial simulation.  In the product library a stray environment variable
//    cannot change what a call returns.
#define PSD_SL_MAXG_API 8  // (= PSD_SL_MAXG of psd_slice3.h, which the serial simulation does not see)
static inline const char* psd_env(const char* k) { return getenv(k); }
#ifdef PSD_DIAG
static inline const char* psd_env_diag(const char* k) { return getenv(k); }
#else
static inline const char* psd_env_diag(const char*) { return nullptr; }
#endif

namespace {

struct Timer {
#ifdef PSD_HOSTSIM
    std::chrono::steady_clock::time_point t0;
    void start(psd_stream_t) { t0 = std::chrono::steady_clock::now(); }
    double stop(psd_stream_t) {
        return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
    }
#else
    hipEvent_t e0 = nullptr, e1 = nullptr;
    Timer() {
        (void)hipEventCreate(&e0);
        (void)hipEventCreate(&e1);
    }
    ~Timer() {
        if (e0) (void)hipEventDestroy(e0);
        if (e1) (void)hipEventDestroy(e1);
    }
    void start(psd_stream_t s) { (void)hipEventRecord(e0, s); }
    double stop(psd_stream_t s) {
        (void)hipEventRecord(e1, s);
        (void)hipEventSynchronize(e1);
        float ms = 0.f;
        (void)hipEventElapsedTime(&ms, e0, e1);
        return (double)ms;
    }
#endif
};

size_t step_lds_bytes(int p, int W, int esize, bool even);
// even: the window area is laid out for an even column pitch (real standard engine: psd_win_area)
int choose_window(int p, int esize = 8, bool even = false) {
    // largest W <= 32 with p blocks of W x (W+1) elements (+ scratch) inside the 160 KiB LDS of one CU;
    // PSD_WINDOW (test hook) lowers it
    int cap = 32;
    if (const char* e = psd_env("PSD_WINDOW")) {
        const int w = atoi(e);
        if (w >= 6 && w < cap) cap = w;
    }
    for (int W = cap; W >= 6; --W)
        if (step_lds_bytes(p, W, esize, even) <= (size_t)160 * 1024) return W;
    return 0;
}
// (real standard engine, even = true: the reflector table of the scan chase, psd_chase3.h, sits behind the scratch)
size_t step_lds_scratch_end(int p, int W, int esize, bool even) {
    const size_t area = even ? (size_t)psd_win_area(W) : (size_t)W * (W + 1);
    size_t b = (size_t)p * area * esize + PSD_STEP_NT * 8 + (2 * PSD_STEP_NT + (size_t)p) * 4;
    return (b + 15) & ~(size_t)15;
}
size_t step_lds_bytes(int p, int W, int esize = 8, bool even = false) {
    size_t b = step_lds_scratch_end(p, W, esize, even);
    if (even && p <= PSD_C3_MAXP) b += (size_t)p * PSD_C3_TAB * sizeof(double);
    // (complex standard engine — the only caller with 16-byte elements and the plain pitch that launches psd_zq_step —:
    //  the rotation table of its scan chase, psd_zchase3.h)
    if (esize == 16 && !even && p <= PSD_ZC3_MAXP) b += (size_t)p * PSD_ZC3_TAB * sizeof(double);
    return (b + 15) & ~(size_t)15;
}
// ordschur! alignments (ordschur.jl:20-33, rordschur.jl:15-27, utils.jl:6-85): the swap kernels work on the right-
// oriented form with the quasi-triangular factor first.  slotA[j] / slotZ[j] (0-based) = user slot of the internal
// factor / transformation j: 'R' with schurindex p is a circular shift by one (_circshift), 'L' is the reversal
// (_rev_alias) — with schurindex 1 followed by that shift.  false: schurindex not in (1, p).
bool ord_slots(char orient, int schurindex, int p, std::vector<int>& slotA, std::vector<int>& slotZ) {
    if (schurindex != 1 && schurindex != p) return false;
    slotA.assign(p, 0);
    slotZ.assign(p, 0);
    const bool left = orient == 'L';
    const bool shift = p > 1 && ((!left && schurindex == p) || (left && schurindex == 1));
    for (int j = 0; j < p; ++j) {
        const int q = shift ? (j + p - 1) % p : j;  // position before the shift
        slotA[j] = left ? (p - 1 - q) : q;
        slotZ[j] = (!left || q == 0) ? q : (p - q);
    }
    return true;
}

size_t apply_lds_bytes() { return PSD_TR_LDS_BYTES + (size_t)32 * (PSD_APPLY_NT + 1) * 8; }

}  // namespace

// Contexts alive in this process.  The pipe form of the Hessenberg reduction parks a launch of 129 workgroups on the chip
// while it waits for its predecessor; one context never holds more than two such launches (258 of the chip's 1024 workgroup
// slots), but several contexts reducing at once could fill the slots with waiting workgroups whose predecessors then cannot
// become resident (every wait is bounded, so that would end in a runtime error, not in a hang).  With more than one context
// alive the reduction therefore takes the back-to-back form.  Other PROCESSES on the same GPU are not seen: PSD_H2_PIPE=0.
static std::atomic<int> g_live_contexts{0};

struct psd_ctx {
    int device = 0;
    psd_stream_t stream = 0;
    int profile = 0;
#ifndef PSD_HOSTSIM
    // state polling without draining the stream: two pinned slots and events (iterate_dev)
    void* pin[2] = {nullptr, nullptr};
    hipEvent_t pev[2] = {nullptr, nullptr};
#endif
    // workspace (grown on demand)
    int cap_n = 0, cap_p = 0;
    bool cap_mats = false;
    double *dH = nullptr, *dZ = nullptr;  // staging for the host entry points
    double *tau = nullptr, *vbuf = nullptr;
    double *hdiag = nullptr, *hsub = nullptr, *hsup = nullptr, *Pd = nullptr, *Pe = nullptr, *Pf = nullptr;
    double *hnorms = nullptr, *wr = nullptr, *wi = nullptr;
    psd_rstate* st = nullptr;
    psd_apply_desc* desc = nullptr;
    psd_tr* tr = nullptr;
    int* cnt = nullptr;
    int* log = nullptr;
    int logcap = 0;
    size_t step_lds_set = 0, zstep_lds_set = 0, rostep_lds_set = 0;
    psd_hess_args* hargs = nullptr;  // device argument block of the graph-replayed Hessenberg reduction
    // multishift trains: bulges per train (0/1 = off), per-cursor state / descriptor / lists
    int ztrain_m = 48;  // complex single-shift engine (shifts from a block of order <= PSD_ZHQR_MAX, reused by longer trains) (psd_set_train sets both; psd_set_train_z / PSD_TRAIN_Z this one)
    int train_m = 32;  // default: trains of up to 32 bulges (psd_set_train / PSD_TRAIN; 0 or 1 = the reference's iteration)
    int tcap_p = 0;
    psd_rstate* tcst = nullptr;
    double* tshift = nullptr;
    psd_apply_desc* tdesc = nullptr;
    int* tcnt = nullptr;
    psd_tr* ttr = nullptr;
    // complex engine
    int ztcap_p = 0;
    psd_zstate* ztcst = nullptr;
    psd_zapply_desc* ztdesc = nullptr;
    int* ztcnt = nullptr;
    psd_ztr* zttr = nullptr;
    psd_z* ztshift = nullptr;
    int ztreserve(int p) {
        if (ztcst && p <= ztcap_p) return 0;
        ztrelease();
        PSD_CHECK(psd_rt_malloc((void**)&ztcst, sizeof(psd_zstate) * PSD_TRAIN_MAX + sizeof(int) * (PSD_TRAIN_MAX + 8)));
        // (two sets, by tick parity: the Schur-vector updates of a tick run on the second stream while the next tick writes its lists)
        PSD_CHECK(psd_rt_malloc((void**)&ztdesc, 2 * sizeof(psd_zapply_desc) * PSD_TRAIN_MAX));
        PSD_CHECK(psd_rt_malloc((void**)&ztcnt, 2 * sizeof(int) * PSD_TRAIN_MAX * (size_t)(p + 8)));
        PSD_CHECK(psd_rt_malloc((void**)&zttr, 2 * sizeof(psd_ztr) * PSD_TRAIN_MAX * (size_t)p * PSD_ZTR_CAP));
        PSD_CHECK(psd_rt_malloc((void**)&ztshift, sizeof(psd_z) * (PSD_TRAIN_MAX + 2)));
        ztcap_p = p;
        return 0;
    }
    void ztrelease() {
        if (ztcst) psd_rt_free(ztcst);
        if (ztdesc) psd_rt_free(ztdesc);
        if (ztcnt) psd_rt_free(ztcnt);
        if (zttr) psd_rt_free(zttr);
        if (ztshift) psd_rt_free(ztshift);
        ztshift = nullptr;
        ztcst = nullptr; ztdesc = nullptr; ztcnt = nullptr; zttr = nullptr;
        ztcap_p = 0;
    }
    // (sized for the PSD_SLOTS slots of the multi-block scheduler; the single-range train mode uses the first
    //  PSD_TRAIN_MAX entries)
    psd_rglobal* tgl = nullptr;
    int* tslotw = nullptr;  // role[PSD_SLOTS] | epoch[PSD_SLOTS] | cdone[PSD_SLOTS]
    int mblock = 1;         // PSD_MB=0: one active range at a time, as the reference
    int train_mb_m = 64;    // bulges per train under the multi-block scheduler (PSD_TRAIN_MB)
    int cgap = 1;           // ticks between the cursors of a train under the multi-block scheduler (PSD_CGAP=2: two windows)
    int treserve(int p) {
        if (tcst && p <= tcap_p) return 0;
        trelease();
        PSD_CHECK(psd_rt_malloc((void**)&tcst, sizeof(psd_rstate) * PSD_SLOTS));
        PSD_CHECK(psd_rt_malloc((void**)&tshift, sizeof(double) * PSD_TSHIFT_STRIDE * PSD_SLOTS));
        // (descriptors, counts and lists twice: ticks alternate between the two sets, so that the far part of a tick's
        //  bulk update can still read its lists while the next tick's chases write theirs)
        PSD_CHECK(psd_rt_malloc((void**)&tdesc, 2 * sizeof(psd_apply_desc) * PSD_SLOTS));
        PSD_CHECK(psd_rt_malloc((void**)&tcnt, 2 * sizeof(int) * PSD_SLOTS * (size_t)(p + 8)));
        PSD_CHECK(psd_rt_malloc((void**)&ttr, 2 * sizeof(psd_tr) * PSD_SLOTS * (size_t)p * PSD_TR_CAP));
        PSD_CHECK(psd_rt_malloc((void**)&tgl, sizeof(psd_rglobal)));
        PSD_CHECK(psd_rt_malloc((void**)&tslotw, sizeof(int) * (8 * PSD_SLOTS + PSD_PLAN_INTS)));
        tcap_p = p;
        return 0;
    }
    void trelease() {
        if (tcst) psd_rt_free(tcst);
        if (tshift) psd_rt_free(tshift);
        if (tdesc) psd_rt_free(tdesc);
        if (tcnt) psd_rt_free(tcnt);
        if (ttr) psd_rt_free(ttr);
        if (tgl) psd_rt_free(tgl);
        if (tslotw) psd_rt_free(tslotw);
        tcst = nullptr; tshift = nullptr; tdesc = nullptr; tcnt = nullptr; ttr = nullptr;
        tgl = nullptr; tslotw = nullptr;
        tcap_p = 0;
    }
    // blocked Q formation (psd_formq2.h): the T factors of all blocks of all factors
    double* fqT = nullptr;
    size_t fqT_doubles = 0;
    int formq_blocked = 1;
#ifndef PSD_HOSTSIM
    hipGraphExec_t hess_exec = nullptr;
    int hess_graph_n = 0, hess_graph_p = 0;
    // look-ahead reduction (psd_hess2.h)
    double* h2ring = nullptr;
    int h2ring_n = 0;
    double* zh2ring = nullptr;  // (the complex form's ring: psd_zhess2.h)
    int zh2ring_n = 0;
    std::vector<hipEvent_t> h2ev;
    int hess_async = -1;  // links per panel-update launch of the two-stream form (0: off, < 0: by size)
    int hess_lookahead = 1;  // PSD_HESS_LOOKAHEAD=0: the two-launch form of psd_hess.h
    int hess_xcd = 1;        // chain strips that share a 128-byte line on one XCD (PSD_H2_XCD)
#endif
    // period sharding (psd_set_shard): this context holds the Schur vectors Z_j of a contiguous slice of the period
    int shard_rank = 0, shard_world = 1;
    void slice(int p, int& lo, int& hi) const {  // [lo, hi), 0-based internal factor index
        const int base = p / shard_world, rem = p % shard_world;
        lo = shard_rank * base + (shard_rank < rem ? shard_rank : rem);
        hi = lo + base + (shard_rank < rem ? 1 : 0);
    }
    int train_stop = 1;    // a long train stops admitting bulges once one of them leaves the bottom converged (PSD_TRAIN_STOP=0: never)
    int train_wdiv = 0;    // a long train has at most (range width) / train_wdiv bulges (PSD_TRAIN_WDIV); 0: 6 for periods 12 .. 32 (round 4,
                           // scan chase: 3 .. 11 % faster there — 1024 x 16: 377 -> 334 ms — with the residuals where they were;
                           // 5 % slower at p = 64); 16 for periods below 12 at n >= 512 (fewer bulges per train: the residual of
                           // large orders with short periods sits close to the gate); 8 otherwise.  4 is 2.5 % (n = 1024, p = 64) to
                           // 16 % (n = 512, p = 16) faster on the iteration and was tried as the default at the end of round 3: the
                           // residual grows by 8-9 % on the bench inputs, and 1 of 485 random cases (n = 371, p = 8) left the residual
                           // gate by 3.6 % (tests/gpu_fuzz_real.py; worst case 0.80 of the gate with 8) - not kept
    int train_long = 256;  // bulges per train of the multi-block scheduler when slots can be recycled (PSD_TRAIN_LONG; 0: one bulge per slot)
    int band_helper = 1;  // PSD_BAND_HELPER=0: a leader computes the product band of its decisions itself
    int zstream = 3;  // rows-deferral form: the Schur-vector updates on stream3 (beside the far H updates) or behind them on stream2 (PSD_ZSTREAM)
    int zcdefer = 0;  // ComplexF64 engine: far rows of the column roles on the second stream (PSD_ZCDEFER; see ziterate_dev)
    int redge = 0, cedge = 0;  // near / far boundaries of the deferred roles (0: psd_rdefer_edge / psd_cdefer_edge; PSD_RDEFER_EDGE, PSD_CDEFER_EDGE)
    int rdefer = 1;   // far columns of the rows roles on stream2 as well, in front of the far column roles (psd_rdefer_edge); needs cdefer.
                      // 1: for n < 1536 (measured: iteration 358 -> 346 ms at n = 1024, p = 64; 1641 -> 1661 ms at n = 2048: the tick
                      // is bound by the bytes of the bulk updates there, whatever runs beside what), 2: always, 0: never (PSD_RDEFER)
    int cdefer = 1;   // far rows of the column roles on stream2 beside the next tick's chases (psd_cdefer_edge): 0 off, 1 where the
                      // Schur vectors go there too (`overlap`), 2 always (PSD_CDEFER)
    int overlap = 3;  // Schur-vector updates on stream2 beside the next tick's chases: 0 off, 2 on, 3 = on for n >= 1024 (PSD_OVERLAP)
    int far_grid = 0;  // grid of the far bulk-update launches (0: apply_wl_grid)
    bool counted = false;  // (this context is in g_live_contexts)
    bool pipe_ok = false;  // created while no other context of the process was alive: its Hessenberg reductions may take the pipe form
#ifndef PSD_HOSTSIM
    hipStream_t stream2 = nullptr;  // the far parts of the bulk updates (beside the next tick's chases)
    hipStream_t stream3 = nullptr;  // the panel updates of the Hessenberg reduction (beside its chain)
    hipStream_t stream4 = nullptr;  // every other chain launch of the Hessenberg reduction in pipe mode (hessenberg2_pipe)
    int hess_pipe_depth = 3;        // chain launches in flight in the pipe form (real reduction): 3 with stream2 — idle during the reduction — as the
                                    // third chain stream, or 2 (PSD_H2_DEPTH).  Measured at n = 1024, p = 64: 409 -> 399 ms
    int hess_pipe = 1;              // PSD_H2_PIPE=0: chain launches back to back on one stream; 2: pipe form also beside other contexts
    hipEvent_t evE[2] = {nullptr, nullptr}, evF[2] = {nullptr, nullptr}, evG[2] = {nullptr, nullptr};
    hipEvent_t evC[2] = {nullptr, nullptr};  // a tick's chase launch is done (rows-role deferral: psd_rdefer_edge)
    // (the Schur-vector updates of a tick, when stream2 carries both far parts of the H updates, run on stream3: the panel
    //  stream of the Hessenberg reduction, idle during the iteration and confined to the same compute units as stream2.  A
    //  fifth stream of its own was measured first: the runtime multiplexes streams onto four hardware queues, the two chain
    //  streams of the pipe form then shared one, and the reduction took 1.6 s instead of 0.41 s.)
#endif
    int apply_worklist = 1;   // PSD_APPLY_WL=0: the grid-per-cursor bulk-apply kernels
    int apply_wl_grid = 2048; // workgroups of the work-list bulk apply (PSD_APPLY_WL_GRID)
    int ncu = 0;              // compute units of the device (0: unknown)
    int slices = 1;           // factor-sliced sweep windows of the real engine (psd_set_slices, psd_slice3.h): workgroups per window
    unsigned char* slmem = nullptr;  // command blocks and inboxes of the sliced windows, then the error word of their waits
    int chase3 = 1;           // scan chase of the real periodic QR sweep (psd_chase3.h; PSD_C3=0: the two-wave / one-wave chases)
    int chase2 = 1;           // two-wave chase of the real periodic QR sweep (psd_c2_run; PSD_C2=0: one wavefront per bulge)
    int apply_wl2 = 1;        // register-line form of the work-list bulk apply (psd_apply2.h) where it is the faster one; PSD_APPLY_WL2=0: never, 2: always
    int apply_wl2_grid = 1024;  // its grid of four-wave workgroups (PSD_APPLY_WL2_GRID)
    size_t wl2_lds_set[3] = {0, 0, 0};
    psd_rostate* rost = nullptr;
    psd_tq* rotq = nullptr;
    unsigned char* rosel = nullptr;
    double* roxscr = nullptr;
    // pipelined real ordschur! (psd_rord_plan / psd_rord_step_mb): scheduler state, slots, and PSD_RO_SLOTS sets of
    // transforms, counts and descriptors
    psd_romb* romb = nullptr;
    psd_roslot* roslots = nullptr;
    psd_tq* rotq_mb = nullptr;
    int* rocnt_mb = nullptr;
    psd_apply_desc* rodesc_mb = nullptr;
    int ord_pipe = 1;  // PSD_ORD_PIPE=0: one selected block at a time (psd_rord_step)
    size_t rostep_mb_lds_set = 0;
    int rocap_n = 0, rocap_p = 0;
    // complex path
    int zcap_n = 0, zcap_p = 0, zlogcap = 0;
    bool zcap_mats = false;
    psd_z *zH = nullptr, *zZ = nullptr, *ztau = nullptr, *zvbuf = nullptr, *zalpha = nullptr;
    double* zbeta = nullptr;
    int *zascale = nullptr, *zcnt = nullptr, *zlog = nullptr;
    psd_zstate* zst = nullptr;
    psd_zapply_desc* zdesc = nullptr;
    psd_ztr *ztr = nullptr, *zdG = nullptr;
    psd_ostate* ost = nullptr;
    // pipelined complex ordschur! drivers (psd_ord1_plan): slot states, slot bookkeeping, scheduler state
    psd_ostate* ombst = nullptr;
    psd_oslot* oslots = nullptr;
    psd_omb* omb = nullptr;
    size_t ostep_mb_lds_set = 0, zgostep_mb_lds_set = 0;
    unsigned char* osel = nullptr;
    size_t ostep_lds_set = 0;
    // real generalized path
    int gcap_n = 0, gcap_p = 0, glogcap = 0;
    psd_gstate* gst = nullptr;
    psd_gapply_desc* gdesc = nullptr;
    psd_gtr *gtr = nullptr, *gdG = nullptr;
    unsigned char* gS = nullptr;
    psd_z* galpha = nullptr;
    double *gbeta = nullptr, *gxscr = nullptr;
    int *gascale = nullptr, *gcnt = nullptr, *glog = nullptr;
    size_t gstep_lds_set = 0, ghess_lds_set = 0;
    // multishift trains of the real signed engine (psd_set_train sets all engines; psd_set_train_g / PSD_TRAIN_G this one)
    int gtrain_m = 48, gtcap_p = 0;  // (real signed engine: 48 bulges W positions apart; the complex signed one caps at 16)
    psd_gstate* gtcst = nullptr;
    double* gtshift = nullptr;
    psd_gapply_desc* gtdesc = nullptr;
    int* gtcnt = nullptr;
    psd_gtr* gttr = nullptr;
    void gtrelease() {
        void* ptrs[] = {gtcst, gtshift, gtdesc, gtcnt, gttr};
        for (void* q : ptrs)
            if (q) psd_rt_free(q);
        gtcst = nullptr; gtshift = nullptr; gtdesc = nullptr; gtcnt = nullptr; gttr = nullptr;
        gtcap_p = 0;
    }
    int gtreserve(int p) {
        if (gtcst && p <= gtcap_p) return 0;
        gtrelease();
        PSD_CHECK(psd_rt_malloc((void**)&gtcst, sizeof(psd_gstate) * PSD_TRAIN_MAX + sizeof(int) * (PSD_TRAIN_MAX + 8)));
        PSD_CHECK(psd_rt_malloc((void**)&gtshift, sizeof(double) * (4 * PSD_TRAIN_MAX + 8)));
        PSD_CHECK(psd_rt_malloc((void**)&gtdesc, sizeof(psd_gapply_desc) * PSD_TRAIN_MAX));
        PSD_CHECK(psd_rt_malloc((void**)&gtcnt, sizeof(int) * PSD_TRAIN_MAX * (size_t)(p + 8)));
        PSD_CHECK(psd_rt_malloc((void**)&gttr, sizeof(psd_gtr) * PSD_TRAIN_MAX * (size_t)p * PSD_GTR_CAP));
        gtcap_p = p;
        return 0;
    }
    // complex generalized path (shares zalpha/zbeta/zascale/zlog/zvbuf with the complex path)
    int zgcap_n = 0, zgcap_p = 0;
    psd_zgstate* zgst = nullptr;
    psd_gapply_desc* zgdesc = nullptr;
    psd_ztr *zgtr = nullptr, *zgdG = nullptr;
    unsigned char* zgS = nullptr;
    int* zgcnt = nullptr;
    size_t zgstep_lds_set = 0, zgostep_lds_set = 0, zghess_lds_set = 0;
    // multishift trains of the complex signed engine (width: gtrain_m, as the real signed engine)
    int zgtcap_p = 0;
    psd_zgstate* zgtcst = nullptr;
    psd_z* zgtshift = nullptr;
    psd_gapply_desc* zgtdesc = nullptr;
    int* zgtcnt = nullptr;
    psd_ztr* zgttr = nullptr;
    void zgtrelease() {
        void* ptrs[] = {zgtcst, zgtshift, zgtdesc, zgtcnt, zgttr};
        for (void* q : ptrs)
            if (q) psd_rt_free(q);
        zgtcst = nullptr; zgtshift = nullptr; zgtdesc = nullptr; zgtcnt = nullptr; zgttr = nullptr;
        zgtcap_p = 0;
    }
    int zgtreserve(int p) {
        if (zgtcst && p <= zgtcap_p) return 0;
        zgtrelease();
        PSD_CHECK(psd_rt_malloc((void**)&zgtcst, sizeof(psd_zgstate) * PSD_TRAIN_MAX + sizeof(int) * (PSD_TRAIN_MAX + 8)));
        PSD_CHECK(psd_rt_malloc((void**)&zgtshift, sizeof(psd_z) * (PSD_TRAIN_MAX + 2)));
        PSD_CHECK(psd_rt_malloc((void**)&zgtdesc, sizeof(psd_gapply_desc) * PSD_TRAIN_MAX));
        PSD_CHECK(psd_rt_malloc((void**)&zgtcnt, sizeof(int) * PSD_TRAIN_MAX * (size_t)(p + 8)));
        PSD_CHECK(psd_rt_malloc((void**)&zgttr, sizeof(psd_ztr) * PSD_TRAIN_MAX * (size_t)p * PSD_GTR_CAP));
        zgtcap_p = p;
        return 0;
    }

    void zgrelease() {
        void* ptrs[] = {zgst, zgdesc, zgtr, zgdG, zgS, zgcnt};
        for (void* q : ptrs)
            if (q) psd_rt_free(q);
        zgst = nullptr; zgdesc = nullptr; zgtr = zgdG = nullptr; zgS = nullptr; zgcnt = nullptr;
        zgcap_n = zgcap_p = 0;
    }
    int zgreserve(int n, int p) {
        if (n <= zgcap_n && p <= zgcap_p) return 0;
        zgrelease();
#define PSD_ALLOC(ptr, type, count) PSD_CHECK(psd_rt_malloc((void**)&ptr, sizeof(type) * (size_t)(count)))
        PSD_ALLOC(zgst, psd_zgstate, 1);
        PSD_ALLOC(zgdesc, psd_gapply_desc, 1);
        PSD_ALLOC(zgtr, psd_ztr, (size_t)p * PSD_GTR_CAP);
        PSD_ALLOC(zgdG, psd_ztr, n + 8);
        PSD_ALLOC(zgS, unsigned char, p + 16);
        PSD_ALLOC(zgcnt, int, p + 8);
#undef PSD_ALLOC
        zgcap_n = n;
        zgcap_p = p;
        return 0;
    }

    void grelease() {
        void* ptrs[] = {gst, gdesc, gtr, gdG, gS, galpha, gbeta, gxscr, gascale, gcnt, glog};
        for (void* q : ptrs)
            if (q) psd_rt_free(q);
        gst = nullptr; gdesc = nullptr; gtr = gdG = nullptr; gS = nullptr; galpha = nullptr;
        gbeta = gxscr = nullptr; gascale = gcnt = glog = nullptr;
        gcap_n = gcap_p = glogcap = 0;
    }
    int greserve(int n, int p, int maxlog) {
        if (n <= gcap_n && p <= gcap_p && maxlog <= glogcap) return 0;
        grelease();
#define PSD_ALLOC(ptr, type, count) PSD_CHECK(psd_rt_malloc((void**)&ptr, sizeof(type) * (size_t)(count)))
        PSD_ALLOC(gst, psd_gstate, 1);
        PSD_ALLOC(gdesc, psd_gapply_desc, 1);
        PSD_ALLOC(gtr, psd_gtr, (size_t)p * PSD_GTR_CAP);
        PSD_ALLOC(gdG, psd_gtr, n + 8);
        PSD_ALLOC(gS, unsigned char, p + 16);
        PSD_ALLOC(galpha, psd_z, n + 8);
        PSD_ALLOC(gbeta, double, n + 8);
        PSD_ALLOC(gxscr, double, (size_t)16 * p + 16);
        PSD_ALLOC(gascale, int, n + 8);
        PSD_ALLOC(gcnt, int, p + 8);
        PSD_ALLOC(glog, int, 3 * (size_t)maxlog + 8);
#undef PSD_ALLOC
        gcap_n = n;
        gcap_p = p;
        glogcap = maxlog;
        return 0;
    }

    void release() {
        if (hargs) psd_rt_free(hargs);
        hargs = nullptr;
#ifndef PSD_HOSTSIM
        if (hess_exec) (void)hipGraphExecDestroy(hess_exec);
        hess_exec = nullptr;
        hess_graph_n = hess_graph_p = 0;
        if (h2ring) psd_rt_free(h2ring);
        h2ring = nullptr;
        h2ring_n = 0;
        if (zh2ring) psd_rt_free(zh2ring);
        zh2ring = nullptr;
        zh2ring_n = 0;
#endif
        if (fqT) psd_rt_free(fqT);
        fqT = nullptr;
        fqT_doubles = 0;
        void* ptrs[] = {dH, dZ, tau, vbuf, hdiag, hsub, hsup, Pd, Pe, Pf, hnorms, wr, wi, st, desc, tr, cnt, log};
        for (void* q : ptrs)
            if (q) psd_rt_free(q);
        dH = dZ = tau = vbuf = hdiag = hsub = hsup = Pd = Pe = Pf = hnorms = wr = wi = nullptr;
        st = nullptr; desc = nullptr; tr = nullptr; cnt = nullptr; log = nullptr;
        cap_n = cap_p = 0;
        cap_mats = false;
        logcap = 0;
    }

    void rorelease() {
        void* ptrs[] = {rost, rotq, rosel, roxscr, romb, roslots, rotq_mb, rocnt_mb, rodesc_mb};
        romb = nullptr; roslots = nullptr; rotq_mb = nullptr; rocnt_mb = nullptr; rodesc_mb = nullptr;
        for (void* q : ptrs)
            if (q) psd_rt_free(q);
        rost = nullptr; rotq = nullptr; rosel = nullptr; roxscr = nullptr;
        rocap_n = rocap_p = 0;
    }
    int roreserve(int n, int p) {
        if (n <= rocap_n && p <= rocap_p) return 0;
        rorelease();
        PSD_CHECK(psd_rt_malloc((void**)&rost, sizeof(psd_rostate)));
        PSD_CHECK(psd_rt_malloc((void**)&rotq, sizeof(psd_tq) * (size_t)p * PSD_RORD_CAP));
        PSD_CHECK(psd_rt_malloc((void**)&rosel, (size_t)n + 16));
        PSD_CHECK(psd_rt_malloc((void**)&roxscr, sizeof(double) * (size_t)n * p * 8 + 64));
        PSD_CHECK(psd_rt_malloc((void**)&romb, sizeof(psd_romb)));
        PSD_CHECK(psd_rt_malloc((void**)&roslots, sizeof(psd_roslot) * PSD_RO_SLOTS));
        PSD_CHECK(psd_rt_malloc((void**)&rotq_mb, sizeof(psd_tq) * (size_t)PSD_RO_SLOTS * p * PSD_RORD_CAP));
        PSD_CHECK(psd_rt_malloc((void**)&rocnt_mb, sizeof(int) * (size_t)PSD_RO_SLOTS * p));
        PSD_CHECK(psd_rt_malloc((void**)&rodesc_mb, sizeof(psd_apply_desc) * PSD_RO_SLOTS));
        rocap_n = n;
        rocap_p = p;
        return 0;
    }

    void zrelease() {
        void* ptrs[] = {zH, zZ, ztau, zvbuf, zalpha, zbeta, zascale, zcnt, zlog, zst, zdesc, ztr, zdG, ost, osel, ombst, oslots, omb};
        ost = nullptr;
        ombst = nullptr; oslots = nullptr; omb = nullptr;
        osel = nullptr;
        for (void* q : ptrs)
            if (q) psd_rt_free(q);
        zH = zZ = ztau = zvbuf = zalpha = nullptr;
        zbeta = nullptr;
        zascale = zcnt = zlog = nullptr;
        zst = nullptr; zdesc = nullptr; ztr = zdG = nullptr;
        zcap_n = zcap_p = zlogcap = 0;
        zcap_mats = false;
    }

    int zreserve(int n, int p, bool mats, int maxlog) {
        if (n <= zcap_n && p <= zcap_p && (!mats || zcap_mats) && maxlog <= zlogcap) return 0;
        const bool keep = mats || zcap_mats;
        zrelease();
        const size_t nn = (size_t)n * n;
#define PSD_ALLOC(ptr, type, count) PSD_CHECK(psd_rt_malloc((void**)&ptr, sizeof(type) * (size_t)(count)))
        if (keep) {
            PSD_ALLOC(zH, psd_z, nn * p);
            PSD_ALLOC(zZ, psd_z, nn * p);
        }
        PSD_ALLOC(ztau, psd_z, (size_t)n * p);
        PSD_ALLOC(zvbuf, psd_z, 2 * (n + 8));  // (two reflectors, as vbuf)
        PSD_ALLOC(zalpha, psd_z, n + 8);
        PSD_ALLOC(zbeta, double, n + 8);
        PSD_ALLOC(zascale, int, n + 8);
        PSD_ALLOC(zcnt, int, p + 8);
        PSD_ALLOC(zlog, int, 3 * (size_t)maxlog + 8);
        PSD_ALLOC(zst, psd_zstate, 1);
        PSD_ALLOC(zdesc, psd_zapply_desc, 1);
        PSD_ALLOC(ztr, psd_ztr, (size_t)p * PSD_ZTR_CAP);
        PSD_ALLOC(zdG, psd_ztr, n + 8);
        PSD_ALLOC(ost, psd_ostate, 1);
        PSD_ALLOC(ombst, psd_ostate, PSD_O_SLOTS);
        PSD_ALLOC(oslots, psd_oslot, PSD_O_SLOTS);
        PSD_ALLOC(omb, psd_omb, 1);
        PSD_ALLOC(osel, unsigned char, n + 16);
#undef PSD_ALLOC
        zcap_n = n;
        zcap_p = p;
        zcap_mats = keep;
        zlogcap = maxlog;
        return 0;
    }

    int reserve(int n, int p, bool mats, int maxlog) {
        if (n <= cap_n && p <= cap_p && (!mats || cap_mats) && maxlog <= logcap) return 0;
        const bool keep_mats = mats || cap_mats;
        release();
        const size_t nn = (size_t)n * n;
#define PSD_ALLOC(ptr, type, count) PSD_CHECK(psd_rt_malloc((void**)&ptr, sizeof(type) * (size_t)(count)))
        if (keep_mats) {
            PSD_ALLOC(dH, double, nn * p);
            PSD_ALLOC(dZ, double, nn * p);
        }
        PSD_ALLOC(tau, double, (size_t)n * p);
        PSD_ALLOC(vbuf, double, 2 * (n + 8));  // (two reflectors: the QR sweeps of the signed reduction form the next one behind the panel update)
        PSD_ALLOC(hdiag, double, n + 8);
        PSD_ALLOC(hsub, double, n + 8);
        PSD_ALLOC(hsup, double, n + 8);
        PSD_ALLOC(Pd, double, n + 8);
        PSD_ALLOC(Pe, double, n + 8);
        PSD_ALLOC(Pf, double, n + 8);
        PSD_ALLOC(hnorms, double, p + 8);
        PSD_ALLOC(wr, double, n + 8);
        PSD_ALLOC(wi, double, n + 8);
        PSD_ALLOC(st, psd_rstate, 1);
        PSD_ALLOC(desc, psd_apply_desc, 1);
        PSD_ALLOC(tr, psd_tr, (size_t)p * PSD_TR_CAP);
        PSD_ALLOC(cnt, int, p + 8);
        PSD_ALLOC(log, int, 3 * (size_t)maxlog + 8);
#undef PSD_ALLOC
        cap_n = n;
        cap_p = p;
        cap_mats = keep_mats;
        logcap = maxlog;
        return 0;
    }
};

#ifndef PSD_HOSTSIM
// State polling of the window drivers without draining the stream: the state of the batch just enqueued is copied to a
// pinned slot behind it, and the host looks at the PREVIOUS batch's state while this one runs (a drained poll costs
// ~70 us).  The batch in flight when DONE is seen consists of launches that return at once.  The first batch is polled
// drained (small problems end there).  HIP-event samples of the chase kernel are read back one batch late as well.
struct psd_poller {
    typedef std::vector<std::pair<hipEvent_t, hipEvent_t>> evlist;
    psd_ctx* c;
    double& sample_ms;
    int& samples;
    int slot = 0;
    bool have_prev = false;
    evlist pend_prev;
    psd_poller(psd_ctx* c_, double& ms, int& ns) : c(c_), sample_ms(ms), samples(ns) {}
    ~psd_poller() {  // (error paths leave without finish())
        for (auto& pr : pend_prev) {
            (void)hipEventDestroy(pr.first);
            (void)hipEventDestroy(pr.second);
        }
    }
    void harvest(evlist& v) {
        for (auto& pr : v) {
            float ms = 0.f;
            if (hipEventElapsedTime(&ms, pr.first, pr.second) == hipSuccess) {
                sample_ms += ms;
                ++samples;
            }
            (void)hipEventDestroy(pr.first);
            (void)hipEventDestroy(pr.second);
        }
        v.clear();
    }
    int poll(void* hst, const void* dev, size_t bytes, evlist& pend) {
        if (bytes > 4096) return PSD_INFO_RUNTIME + 6;
        int rc = psd_rt_d2h(c->pin[slot], dev, bytes, c->stream);
        if (rc) return rc;
        if ((rc = (int)hipEventRecord(c->pev[slot], c->stream)) != 0) return rc;
        if (!have_prev) {
            if ((rc = (int)hipEventSynchronize(c->pev[slot])) != 0) return rc;
            memcpy(hst, c->pin[slot], bytes);
            harvest(pend);
            have_prev = true;
        } else {
            if ((rc = (int)hipEventSynchronize(c->pev[slot ^ 1])) != 0) return rc;
            memcpy(hst, c->pin[slot ^ 1], bytes);
            harvest(pend_prev);
        }
        pend_prev.swap(pend);
        slot ^= 1;
        return 0;
    }
    int finish(evlist& pend) {  // the batch behind the one that reported DONE
        const int rc = psd_rt_sync(c->stream);
        harvest(pend_prev);
        harvest(pend);
        return rc;
    }
};
#endif

namespace {

void fill_bytes(psd_stats* s, int n, int p, int wantT, int wantZ, const std::vector<int>& log) {
    if (!s) return;
    double b = 0.0;
    const double E = 8.0;
    for (size_t q = 0; q + 2 < log.size(); q += 3) {
        if (log[q] != 0) continue;
        const double w = log[q + 2] - log[q + 1] + 1;
        if (!wantT) b += 2 * E * p * w * w + (wantZ ? 2 * E * p * w * n : 0.0);
        else b += 2 * E * p * w * (wantZ ? (2.0 * n + 1) : (n + 1.0));
    }
    s->bytes_sweeps = b;
}

// Which form the multi-stream Hessenberg reductions of this context take (see psd_create): the pipe form needs the
// fourth stream; PSD_H2_PIPE=2 forces it, =0 forbids it; a period-sharded context always takes it (every rank must
// run the same form: the replicated chains have to agree bit for bit, and a rank is one process with one GPU).
static inline bool psd_pipe_form(const psd_ctx* c) {
#ifndef PSD_HOSTSIM
    if (!c->stream4 || c->hess_pipe == 0) return false;
    return c->hess_pipe == 2 || c->shard_world > 1 || c->pipe_ok;
#else
    (void)c;
    return false;
#endif
}

#ifndef PSD_HOSTSIM
// look-ahead form (psd_hess2.h): one launch per chain link, the panel updates ride one launch behind the chain
template <int NK, int CR>
int hessenberg2_launches(psd_ctx* c, int n, int p, const psd_hess2_args& ha) {
    const int nC = ((ha.xcd && CR < 16) ? (((n + CR - 1) / CR + 128 / CR - 1) / (128 / CR)) * (128 / CR) : (n + CR - 1) / CR) + 1, nT = (n + PSD_H2_ROWS - 1) / PSD_H2_ROWS, nB = (n + 3) / 4;
    const size_t lds = ((size_t)n + 8 + 2 * PSD_H2_NT + 64) * sizeof(double);
    int gridx = nC + nT + nB;
    auto link = [&](int i, int j) {
        hipLaunchKernelGGL((psd_hess2_link<NK, CR>), dim3(gridx), dim3(PSD_H2_NT), lds, c->stream, ha, n, i, j, nC, nT);
    };
    link(0, 1);  // the position before link (1, p): stages column 1 of A_p
    for (int i = 1; i <= n - 1; ++i)
        for (int j = p; j >= 1; --j) link(i, j);
    link(n, p);      // drain: the panel updates of the last two links
    link(n, p - 1);
    return 0;
}
// Two-stream form: the chain launches carry no bulk part; the panel updates of K consecutive links (K distinct
// matrices) are ONE launch on the CU-masked second stream, up to p - K links behind the chain.  What the chain needs of a
// matrix — its update by the previous link on it, p links earlier — is awaited by event before the launch that reads it.
template <int NK, int CR>
int hessenberg2_async(psd_ctx* c, int n, int p, const psd_hess2_args& ha, int K) {
    // Every way out (a failed runtime call in the middle of the launch sequence included) first waits for the side streams:
    // their launches read and write the caller's factors and this context's ring.
    struct SideStreams {
        psd_ctx* c;
        ~SideStreams() {
            if (c->stream3) (void)hipStreamSynchronize(c->stream3);
            if (c->stream4) (void)hipStreamSynchronize(c->stream4);
        }
    } side_streams{c};

    const int nC = ((ha.xcd && CR < 16) ? (((n + CR - 1) / CR + 128 / CR - 1) / (128 / CR)) * (128 / CR) : (n + CR - 1) / CR) + 1, nT = (n + PSD_H2_ROWS - 1) / PSD_H2_ROWS, nB = (n + 3) / 4;
    const size_t lds = ((size_t)n + 8 + 2 * PSD_H2_NT + 64) * sizeof(double);
    const int Q = (n - 1) * p;
    const int nbatch = Q / K + 1;  // link indices 0 .. Q (Q: the drain position)
    if ((int)c->h2ev.size() < 16) {
        c->h2ev.resize(16, nullptr);
        for (auto& e : c->h2ev) PSD_CHECK(hipEventCreateWithFlags(&e, hipEventDisableTiming));
    }
    // (an event is re-recorded 8 batches later; the chain awaits batch b while at most (p - K) / K < 8 later ones exist)
    hipEvent_t* evA = c->h2ev.data();      // [8]: chain reached the end of a batch
    hipEvent_t* evB = c->h2ev.data() + 8;  // [8]: a batch of panel updates is done
    // order against whatever ran on the main stream before
    PSD_CHECK(hipEventRecord(evA[0], c->stream));
    PSD_CHECK(hipStreamWaitEvent(c->stream3, evA[0], 0));
    hipLaunchKernelGGL((psd_hess2_link<NK, CR>), dim3(nC), dim3(PSD_H2_NT), lds, c->stream, ha, n, 0, 1, nC, 0);  // staging
    const int lag = 0;  // (launching a batch later than it could be — so that the chain finds its matrix in the Infinity Cache — was measured in rounds 2 and 3: no gain)
    int nextb = 0;  // next batch to launch
    const bool nobulk = psd_env_diag("PSD_H2_NOBULK") != nullptr;  // (timing experiment: the chain alone; results are wrong)
    auto batch = [&](int b) -> int {
        PSD_CHECK(hipEventRecord(evA[b & 7], c->stream));
        PSD_CHECK(hipStreamWaitEvent(c->stream3, evA[b & 7], 0));
        if (!nobulk) hipLaunchKernelGGL((psd_hess2_bulk<NK>), dim3(nT + nB, K), dim3(PSD_H2_NT), lds, c->stream3, ha, n, b * K, nT);
        PSD_CHECK(hipEventRecord(evB[b & 7], c->stream3));
        return 0;
    };
    int idx = 0;
    for (int i = 1; i <= n - 1; ++i)
        for (int j = p; j >= 1; --j, ++idx) {
            // chain#idx reads the matrix of link idx + 1, last updated by B(idx + 1 - p): batch (idx + 1 - p) / K
            const int need = idx + 1 - p;
            if (need >= 0 && need % K == 0) PSD_CHECK(hipStreamWaitEvent(c->stream, evB[(need / K) & 7], 0));
            hipLaunchKernelGGL((psd_hess2_link<NK, CR>), dim3(nC), dim3(PSD_H2_NT), lds, c->stream, ha, n, i, j, nC, 0);
            // links nextb K .. nextb K + K - 1 have their reflectors once the chain has passed the last of them
            if (idx >= nextb * K + K - 1 + lag) {
                PSD_CHECK(batch(nextb));
                ++nextb;
            }
        }
    for (; nextb < nbatch; ++nextb) PSD_CHECK(batch(nextb));  // the rest, up to the drain position Q
    PSD_CHECK(hipStreamWaitEvent(c->stream, evB[(nbatch - 1) & 7], 0));
    return 0;
}
// Pipe form of the two-stream reduction: the chain launches alternate between two streams, so that the launch of link
// q + 1 is resident and has requested its strip of the next matrix while the launch of link q still runs; it then polls the
// staged column, which travels as self-validating records (psd_h2_tag).  At most two chain launches are in flight (a
// stream runs its own launches in order), a waiting launch holds one workgroup slot of four per CU, and every wait is
// bounded.  A batch of panel updates waits for BOTH chain streams (a launch can end before its predecessor's block 0 has
// stored v and tau), and what the chain needs of a batch is awaited on both streams.
template <int NK, int CR>
int hessenberg2_pipe(psd_ctx* c, int n, int p, const psd_hess2_args& ha, int K) {
    // Every way out (a failed runtime call in the middle of the launch sequence included) first waits for the side streams:
    // their launches read and write the caller's factors and this context's ring.
    struct SideStreams {
        psd_ctx* c;
        ~SideStreams() {
            if (c->stream3) (void)hipStreamSynchronize(c->stream3);
            if (c->stream4) (void)hipStreamSynchronize(c->stream4);
            if (c->stream2) (void)hipStreamSynchronize(c->stream2);
        }
    } side_streams{c};

    const int nC = ((ha.xcd && CR < 16) ? (((n + CR - 1) / CR + 128 / CR - 1) / (128 / CR)) * (128 / CR) : (n + CR - 1) / CR) + 1, nT = (n + PSD_H2_ROWS - 1) / PSD_H2_ROWS, nB = (n + 3) / 4;
    const size_t lds = ((size_t)n + 8 + 2 * PSD_H2_NT + 64) * sizeof(double);
    const int Q = (n - 1) * p;
    const int nbatch = Q / K + 1;
    if ((int)c->h2ev.size() < 34) {
        const size_t old = c->h2ev.size();
        c->h2ev.resize(34, nullptr);
        for (size_t q = old; q < c->h2ev.size(); ++q) PSD_CHECK(hipEventCreateWithFlags(&c->h2ev[q], hipEventDisableTiming));
    }
    hipEvent_t* evA = c->h2ev.data();       // [8]: chain stream 0 reached the end of a batch
    hipEvent_t* evB = c->h2ev.data() + 8;   // [8]: a batch of panel updates is done
    hipEvent_t* evC = c->h2ev.data() + 16;  // [8]: chain stream 1 reached the end of a batch
    hipEvent_t evJ = c->h2ev[24], evK = c->h2ev[25];
    hipEvent_t* evD = c->h2ev.data() + 26;  // [8]: chain stream 2 reached the end of a batch (three-deep form)
    // NS chain streams: 2, or 3 with the iteration's second stream (idle here) as the third (c->hess_pipe_depth)
    // (three only while three launches fit the chip beside each other with room to spare — n <= 1024: 3 x 129 workgroups of
    //  the 16-column kernel, four to a CU.  The 32-column kernel of larger orders runs two workgroups per CU: a third launch
    //  would take slots the second still needs, and the first could wait for ever — measured: n = 2048 ran into the bounded
    //  wait, n = 1536 took 1.67 s instead of 1.41 s)
    const int NS = (c->hess_pipe_depth >= 3 && c->stream2 && n <= 1024) ? 3 : 2;
    hipStream_t S[3] = {c->stream, c->stream4, c->stream2};
    PSD_CHECK(hipEventRecord(evJ, c->stream));  // (whatever ran on the main stream before: the memsets, the caller's work)
    PSD_CHECK(hipStreamWaitEvent(c->stream3, evJ, 0));
    PSD_CHECK(hipStreamWaitEvent(c->stream4, evJ, 0));
    if (NS == 3) PSD_CHECK(hipStreamWaitEvent(c->stream2, evJ, 0));
    hipLaunchKernelGGL((psd_hess2_link<NK, CR>), dim3(nC), dim3(PSD_H2_NT), lds, S[0], ha, n, 0, 1, nC, 0);  // staging
    int nextb = 0;
    const bool nobulk = psd_env_diag("PSD_H2_NOBULK") != nullptr;  // (timing experiment: the chain alone; results are wrong)
    auto batch = [&](int b) -> int {
        PSD_CHECK(hipEventRecord(evA[b & 7], S[0]));
        PSD_CHECK(hipEventRecord(evC[b & 7], S[1]));
        PSD_CHECK(hipStreamWaitEvent(c->stream3, evA[b & 7], 0));
        PSD_CHECK(hipStreamWaitEvent(c->stream3, evC[b & 7], 0));
        if (NS == 3) {
            PSD_CHECK(hipEventRecord(evD[b & 7], S[2]));
            PSD_CHECK(hipStreamWaitEvent(c->stream3, evD[b & 7], 0));
        }
        if (!nobulk) hipLaunchKernelGGL((psd_hess2_bulk<NK>), dim3(nT + nB, K), dim3(PSD_H2_NT), lds, c->stream3, ha, n, b * K, nT);
        PSD_CHECK(hipEventRecord(evB[b & 7], c->stream3));
        return 0;
    };
    int idx = 0;
    for (int i = 1; i <= n - 1; ++i)
        for (int j = p; j >= 1; --j, ++idx) {
            hipStream_t s = S[(idx + 1) % NS];
            // chain#idx reads the matrix of link idx + 1, last updated by B(idx + 1 - p): batch (idx + 1 - p) / K; the launch
            // behind it, on the other stream, reads a matrix of the same batch
            const int need = idx + 1 - p;
            if (need >= 0 && need % K == 0) PSD_CHECK(hipStreamWaitEvent(s, evB[(need / K) & 7], 0));
            if (need >= 1 && (need - 1) % K == 0) PSD_CHECK(hipStreamWaitEvent(s, evB[((need - 1) / K) & 7], 0));
            if (NS == 3 && need >= 2 && (need - 2) % K == 0) PSD_CHECK(hipStreamWaitEvent(s, evB[((need - 2) / K) & 7], 0));
            hipLaunchKernelGGL((psd_hess2_link<NK, CR>), dim3(nC), dim3(PSD_H2_NT), lds, s, ha, n, i, j, nC, 0);
            if (idx >= nextb * K + K - 1) {
                PSD_CHECK(batch(nextb));
                ++nextb;
            }
        }
    for (; nextb < nbatch; ++nextb) PSD_CHECK(batch(nextb));
    PSD_CHECK(hipEventRecord(evK, S[1]));
    PSD_CHECK(hipStreamWaitEvent(c->stream, evK, 0));
    if (NS == 3) {
        PSD_CHECK(hipEventRecord(evJ, S[2]));
        PSD_CHECK(hipStreamWaitEvent(c->stream, evJ, 0));
    }
    PSD_CHECK(hipStreamWaitEvent(c->stream, evB[(nbatch - 1) & 7], 0));
    if (psd_env_diag("PSD_H2_BULKBENCH")) {
        // diagnostics: the panel kernel ALONE on the finished matrices (the transformations it applies are the ring's
        // leftovers: results void), the first 100 batches back to back on the panel stream
        PSD_CHECK(hipDeviceSynchronize());
        hipEvent_t e0, e1;
        PSD_CHECK(hipEventCreate(&e0));
        PSD_CHECK(hipEventCreate(&e1));
        const int nb = nbatch < 100 ? nbatch : 100;
        for (int which = 0; which < 2; ++which) {
            hipStream_t st = which ? c->stream : c->stream3;
            PSD_CHECK(hipEventRecord(e0, st));
            for (int b = 0; b < nb; ++b)
                hipLaunchKernelGGL((psd_hess2_bulk<NK>), dim3(nT + nB, K), dim3(PSD_H2_NT), lds, st, ha, n, b * K, nT);
            PSD_CHECK(hipEventRecord(e1, st));
            PSD_CHECK(hipStreamSynchronize(st));
            float ms = 0;
            PSD_CHECK(hipEventElapsedTime(&ms, e0, e1));
            double by = 0;
            for (int idx2 = 0; idx2 < nb * K; ++idx2) by += 16.0 * n * (n - (idx2 / p + 1));
            fprintf(stderr, "psd hess2 bulk alone (%s stream): %d batches of %d links in %.3f ms = %.1f us per batch, %.0f GB/s algorithmic (read + write)\n",
                    which ? "unmasked" : "panel", nb, K, ms, 1e3 * ms / nb, by / ms / 1e6);
        }
        (void)hipEventDestroy(e0);
        (void)hipEventDestroy(e1);
    }
    return 0;
}
int hessenberg2_dev(psd_ctx* c, int n, int p, double* dH, double* dtau) {
    if (!c->h2ring || c->h2ring_n < n) {
        if (c->h2ring) psd_rt_free(c->h2ring);
        c->h2ring = nullptr;
        PSD_CHECK(psd_rt_malloc((void**)&c->h2ring, PSD_H2_RING * psd_h2_slot_doubles(n) * sizeof(double) + 64));
        c->h2ring_n = n;
    }
    PSD_CHECK(psd_rt_memset(c->h2ring, 0, PSD_H2_RING * psd_h2_slot_doubles(c->h2ring_n) * sizeof(double) + 64, c->stream));
    psd_hess2_args ha;
    ha.H = dH;
    ha.tau = dtau;
    ha.ring = c->h2ring;
    ha.p = p;
    ha.ringmask = 3;
    ha.xcd = c->hess_xcd;
    ha.trace = nullptr;
    ha.trace_hi = 0x7fffffff;
    ha.pipe = 0;
    ha.fault = -1;
    if (const char* e = psd_env_diag("PSD_H2_FAULT")) ha.fault = atoi(e);  // (test hook: the hand-over to this link never validates)
    ha.err = (int*)(c->h2ring + PSD_H2_RING * psd_h2_slot_doubles(c->h2ring_n));
    if (const char* e = psd_env_diag("PSD_H2_TRACE")) { if (atoi(e) > 1) ha.trace_hi = atoi(e); }
    long long* h2trace = nullptr;
    if (psd_env_diag("PSD_H2_TRACE") && psd_rt_malloc((void**)&h2trace, (1024 * 8 + 1024 * 4) * sizeof(long long)) == 0) {
        PSD_CHECK(psd_rt_memset(h2trace, 0, (1024 * 8 + 1024 * 4) * sizeof(long long), c->stream));
        ha.trace = h2trace;
    }
    struct H2TraceDump {
        psd_ctx* c;
        long long* t;
        ~H2TraceDump() {
            if (!t) return;
            std::vector<long long> h(1024 * 8 + 1024 * 4);
            (void)psd_rt_sync(c->stream);
            (void)hipMemcpy(h.data(), t, sizeof(long long) * h.size(), hipMemcpyDeviceToHost);
            double acc[8] = {0};
            int cnt = 0;
            double per = 0;
            for (int q = 1; q < 1023; ++q) {
                const long long* a = &h[(size_t)q * 8];
                const long long* pr = &h[(size_t)(q - 1) * 8];
                if (a[0] == 0 || a[6] == 0 || pr[0] == 0 || a[0] < pr[0]) continue;
                for (int k = 1; k <= 6; ++k) acc[k] += (double)(a[k] - a[k - 1]);
                per += (double)(a[0] - pr[0]);
                ++cnt;
            }
            if (cnt) fprintf(stderr, "psd hess2 trace (last %d links, block 2, us): link period %.2f | ring loads %.2f | norm reduce %.2f | larfg %.2f | v to LDS + publish %.2f | gemv %.2f | finish %.2f\n", cnt, per / cnt / 100.0, acc[1] / cnt / 100.0, acc[2] / cnt / 100.0, acc[3] / cnt / 100.0, acc[4] / cnt / 100.0, acc[5] / cnt / 100.0, acc[6] / cnt / 100.0);
            {
                const long long* bk = &h[1024 * 8];
                long long t0 = 0;
                for (int q = 0; q < 1024; ++q)
                    if (bk[4 * q] && (!t0 || bk[4 * q] < t0)) t0 = bk[4 * q];
                if (t0) {
                    fprintf(stderr, "psd hess2 trace, one link, per block (start, end in us after the first start):");
                    for (int q = 0; q < 1024; ++q)
                        if (bk[4 * q] && (q < 12 || q % 16 == 0)) fprintf(stderr, " b%d %.2f-%.2f", q, (bk[4 * q] - t0) / 100.0, (bk[4 * q + 1] - t0) / 100.0);
                    double lastend = 0, maxstart = 0;
                    for (int q = 0; q < 1024; ++q)
                        if (bk[4 * q]) {
                            if ((bk[4 * q + 1] - t0) / 100.0 > lastend) lastend = (bk[4 * q + 1] - t0) / 100.0;
                            if ((bk[4 * q] - t0) / 100.0 > maxstart) maxstart = (bk[4 * q] - t0) / 100.0;
                        }
                    fprintf(stderr, " | last start %.2f last end %.2f\n", maxstart, lastend);
                }
            }
            psd_rt_free(t);
        }
    } h2dump{c, h2trace};
    // two-stream form: K links per panel-update launch; the chain may run p - K links ahead of the updates and the ring
    // keeps every link the pending updates still read (p + K + 2 <= PSD_H2_RING)
    int K = c->hess_async;
    if (K < 0) K = (p >= 32 && n >= 512) ? ((p >= 48 && c->hess_pipe && c->stream4) ? 24 : 16) : 0;  // (measured at p = 64: 16 / 24 / 32 links per batch 548 / 529 / 540 ms)
    if (K > 0 && p >= 9 * K) K = (p + 7) / 8;
    if (K > 0 && p >= 2 * K && p + K + 2 <= PSD_H2_RING && c->stream3) {
        ha.ringmask = PSD_H2_RING - 1;
        if (c->stream4 && psd_pipe_form(c)) {  // (PSD_H2_PIPE=2: also beside other contexts)
            ha.pipe = 2;  // (every poll round reads the whole column; 1: a watch round on one record per strip first — one more round trip per link, 477 against 442 ms)
            if (n > 512 && n <= 1024) ha.xcd = 0;  // (8-row strips: two per line, the mapping no longer pays: 482 -> 474 ms)
            int rc;
            if (n <= 256) rc = hessenberg2_pipe<4, 8>(c, n, p, ha, K);
            else if (n <= 512) rc = hessenberg2_pipe<8, 8>(c, n, p, ha, K);
            // (strips of 8 rows here, 4 in the one-stream forms: with the strip requested before the wait the GEMV no longer
            //  needs every CU's memory pipeline, and half as many workgroups poll; measured 4 / 8 / 16 / 32 rows: 519 / 483 / 542 / 570 ms)
            else if (n <= 1024) rc = hessenberg2_pipe<16, 8>(c, n, p, ha, K);
            else rc = hessenberg2_pipe<32, 8>(c, n, p, ha, K);
            if (rc != 0) return rc;
            // (a launch that gave up waiting left void results: say so.  One word, read when the reduction is done)
            int herr = 0;
            PSD_CHECK(psd_rt_d2h(&herr, ha.err, sizeof(int), c->stream));
            PSD_CHECK(psd_rt_sync(c->stream));
            return herr ? PSD_INFO_RUNTIME + 0xfffb : 0;
        }
        if (n <= 256) return hessenberg2_async<4, 8>(c, n, p, ha, K);
        if (n <= 512) return hessenberg2_async<8, 8>(c, n, p, ha, K);
        if (n <= 1024) return hessenberg2_async<16, 4>(c, n, p, ha, K);
        return hessenberg2_async<32, 8>(c, n, p, ha, K);
    }
    if (n <= 256) return hessenberg2_launches<4, 8>(c, n, p, ha);
    if (n <= 512) return hessenberg2_launches<8, 8>(c, n, p, ha);
    if (n <= 1024) return hessenberg2_launches<16, 4>(c, n, p, ha);
    return hessenberg2_launches<32, 8>(c, n, p, ha);
}
#endif

// PSD.jl:213-259 on device: dH [p][n][n] internal order, overwritten LAPACK-style; tau [p][n]
int hessenberg_dev(psd_ctx* c, int n, int p, double* dH, double* dtau) {
    PSD_CHECK(psd_rt_memset(dtau, 0, sizeof(double) * (size_t)n * p, c->stream));
    const size_t lds_refl = PSD_HESS_NT * 8;
    const size_t lds_apply = (PSD_HESS_NT + (size_t)n + 8) * 8;
    if (n < 2) return 0;
#ifndef PSD_HOSTSIM
    if (p >= 3 && n <= 2048 && c->hess_lookahead) return hessenberg2_dev(c, n, p, dH, dtau);
#endif
    if (!c->hargs) PSD_CHECK(psd_rt_malloc((void**)&c->hargs, sizeof(psd_hess_args)));
    psd_hess_args ha;
    ha.H = dH;
    ha.tau = dtau;
    ha.vbuf = c->vbuf;
    ha.i = 1;
    ha.p = p;
    PSD_CHECK(psd_rt_h2d(c->hargs, &ha, sizeof(ha), c->stream));
    PSD_CHECK(psd_rt_sync(c->stream));  // (ha lives on the stack)
    const int nLmax = (n - 2 + 1 + 3) / 4;  // lc0 = 2
    const int nR = (n + PSD_HESS_RS - 1) / PSD_HESS_RS;
    auto column = [&]() {  // the launches of one column i = hargs->i  (PSD.jl:229-247)
        for (int j = p; j >= 1; --j) {
            PSD_LAUNCH(psd_hess_refl_g, psd_dim3(1), PSD_HESS_NT, lds_refl, c->stream, (const psd_hess_args*)c->hargs, n, j);
            if (p > 1) {
                PSD_LAUNCH(psd_hess_apply_g, psd_dim3(nLmax + nR), PSD_HESS_NT, lds_apply, c->stream,
                           (const psd_hess_args*)c->hargs, n, j, nLmax, 0);
            } else {  // p == 1: same matrix, left then right (PSD.jl:245-246)
                PSD_LAUNCH(psd_hess_apply_g, psd_dim3(nLmax + nR), PSD_HESS_NT, lds_apply, c->stream,
                           (const psd_hess_args*)c->hargs, n, j, nLmax, 1);
                PSD_LAUNCH(psd_hess_apply_g, psd_dim3(nLmax + nR), PSD_HESS_NT, lds_apply, c->stream,
                           (const psd_hess_args*)c->hargs, n, j, nLmax, 2);
            }
        }
        PSD_LAUNCH(psd_hess_next, psd_dim3(1), 64, 0, c->stream, c->hargs);
    };
#ifdef PSD_HOSTSIM
    for (int i = 1; i <= n - 1; ++i) column();
#else
    if (c->hess_exec == nullptr || c->hess_graph_n != n || c->hess_graph_p != p) {
        if (c->hess_exec) {
            (void)hipGraphExecDestroy(c->hess_exec);
            c->hess_exec = nullptr;
        }
        hipGraph_t graph = nullptr;
        PSD_CHECK(hipStreamBeginCapture(c->stream, hipStreamCaptureModeThreadLocal));
        column();
        PSD_CHECK(hipStreamEndCapture(c->stream, &graph));
        PSD_CHECK(hipGraphInstantiate(&c->hess_exec, graph, nullptr, nullptr, 0));
        (void)hipGraphDestroy(graph);
        c->hess_graph_n = n;
        c->hess_graph_p = p;
    }
    for (int i = 1; i <= n - 1; ++i) PSD_CHECK(hipGraphLaunch(c->hess_exec, c->stream));
#endif
    return 0;
}

int formq_dev(psd_ctx* c, int n, int p, const double* dH, const double* dtau, double* dQ) {
    // a period-sharded context forms only the Q_j of its slice (psd_set_shard)
    int jlo = 0, jhi = p;
    c->slice(p, jlo, jhi);
    if (jhi <= jlo) return 0;
    PSD_LAUNCH(psd_set_identity, psd_dim3(n, jhi - jlo), 64, 0, c->stream, dQ + (size_t)jlo * n * n, n);
    if (c->formq_blocked && n >= 2 * PSD_FQ_KB) {
        // compact-WY blocks of 32 reflectors on the matrix cores (psd_formq2.h)
        const int nfac = jhi - jlo;
        const int nblk = (n - 1 + PSD_FQ_KB - 1) / PSD_FQ_KB;
        const size_t need = (size_t)nblk * nfac * PSD_FQ_KB * PSD_FQ_KB;
        if (c->fqT_doubles < need) {
            if (c->fqT) psd_rt_free(c->fqT);
            c->fqT = nullptr;
            c->fqT_doubles = 0;
            PSD_CHECK(psd_rt_malloc((void**)&c->fqT, need * sizeof(double)));
            c->fqT_doubles = need;
        }
        psd_fq_args g;
        g.Hp = dH;
        g.tau = dtau;
        g.Q = dQ;
        g.T = c->fqT;
        g.n = n;
        g.j0 = jlo;
        g.nblk = nblk;
        PSD_LAUNCH(psd_fq_tfactor, psd_dim3(nblk, nfac), PSD_FQ_NT, (64 * 33 + 2 * 32 * 33 + 32) * sizeof(double), c->stream, g);
        for (int b = nblk - 1; b >= 0; --b) {
#ifdef PSD_HOSTSIM
            psd_fq_apply_sim(g, b, nfac);
#else
            const int tiles = (n - b * PSD_FQ_KB + PSD_FQ_TN - 1) / PSD_FQ_TN;
            hipLaunchKernelGGL(psd_fq_apply, dim3(tiles, nfac), dim3(PSD_FQ_NT), 0, c->stream, g, b);
#endif
        }
        return 0;
    }
    const size_t lds = PSD_HESS_NT * 8;
    for (int i = n - 1; i >= 1; --i) {
        const int tiles = (n - i + 1 + 3) / 4;
        PSD_LAUNCH(psd_formq_step, psd_dim3(tiles, jhi - jlo), PSD_HESS_NT, lds, c->stream, dH, dtau, dQ, n, i, jlo);
    }
    return 0;
}

// One launch of the work-list bulk apply (pass 0: rows + Z roles, pass 1: column roles; modes: psd_rq_apply_wl).  W: the
// window width of the call (its spans are at most W).  grid_old: workgroups of the single-wave form.
int launch_apply_wl(psd_ctx* c, psd_stream_t stream, const psd_rparams& Pq, int n, int p, int pass, int NSL, int zlo1, int zhi1,
                    int mode, int W, int grid_old) {
#ifndef PSD_HOSTSIM
    // Which form: measured per role on synthetic windows (tools/apply_bench.py, n = 1024, p = 64, W = 17, 16-60 windows):
    // column roles 3.2-4.5 TB/s algorithmic with the register-line form against 2.2-3.4 with the LDS-streaming one; the
    // rows role (120-byte column segments) 1.4-1.7 against 1.9-2.0; at W = 32 the register-line form needs 230 registers
    // and loses everywhere.  So: register lines for launches without a rows role at W <= 17, LDS streaming otherwise.
    const bool rowsrole_in_launch = pass == 0 && mode != 4;
    const bool use2 = c->apply_wl2 == 2 || (c->apply_wl2 == 1 && W <= 17 && !rowsrole_in_launch);
    if (use2) {
        const int g = c->apply_wl2_grid;
        void (*kern)(psd_rparams, int, int, int, int, int, int, int, int) = nullptr;
        size_t lb = 0;
        int slot = 0;
        if (W <= 17) {
            lb = psd_wl2_lds_bytes<17>();
            kern = psd_rq_apply_wl2<17, 3>;  // (three waves per SIMD is what its LDS admits; a 128-register build spills and measured slower)
            slot = 0;
        } else {
            lb = psd_wl2_lds_bytes<32>();
            kern = psd_rq_apply_wl2<32, 2>;
            slot = 2;
        }
        if (lb > c->wl2_lds_set[slot]) {
            PSD_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lb));
            c->wl2_lds_set[slot] = lb;
        }
        hipLaunchKernelGGL(kern, dim3(g), dim3(PSD_WL2_NT), lb, stream, Pq, n, p, p + 8, pass, NSL, zlo1, zhi1, mode);
        return 0;
    }
#endif
    const int ld = psd_wl_pitch(W);
    PSD_LAUNCH(psd_rq_apply_wl, psd_dim3(grid_old), PSD_WL_NT, psd_wl_lds_bytes(ld), stream, Pq, n, p, p + 8, pass, NSL, zlo1, zhi1, mode, ld);
    return 0;
}

// PSD.jl:322-1096 on device.  dH: H_1 Hessenberg, H_j triangular; dZ: Q_j (or identity) or null.
// nprob > 1 (batch, psd_d_pschur_hess_batch): dH / dZ hold nprob problems back to back, bws the per-problem band arrays
// hdiag | hsub | hsup | Pd | Pe | Pf | wr | wi (nprob (n + 8) doubles each) and hnorms (nprob (p + 8)); pinfo_out: info per
// problem
int iterate_dev(psd_ctx* c, int n, int p, double* dH, double* dZ, int wantT, int wantZ, int maxitfac,
                psd_rstate* st_out, psd_stats* stats, int maxlog, int nprob = 1, double* bws = nullptr,
                int* pinfo_out = nullptr) {
    const int W = choose_window(p, 8, true);
    if (W == 0) return PSD_INFO_NOTIMPL;
#ifndef PSD_HOSTSIM
    // Every way out (a failed runtime call or the runaway cap in the middle of the launch sequence included) first waits
    // for the side streams: their launches read this call's lists and write the caller's factors and Schur vectors.
    struct BulkStreams {
        psd_ctx* c;
        ~BulkStreams() {
            if (c->stream2) (void)hipStreamSynchronize(c->stream2);
            if (c->stream3) (void)hipStreamSynchronize(c->stream3);
        }
    } bulk_streams{c};
#endif
    psd_rparams P;
    P.H = dH;
    P.Z = wantZ ? dZ : nullptr;
    P.st = c->st;
    P.desc = c->desc;
    P.tr = c->tr;
    P.cnt = c->cnt;
    P.hdiag = c->hdiag;
    P.hsub = c->hsub;
    P.hsup = c->hsup;
    P.Pd = c->Pd;
    P.Pe = c->Pe;
    P.Pf = c->Pf;
    P.hnorms = c->hnorms;
    P.wr = c->wr;
    P.wi = c->wi;
    P.log = c->log;
    const size_t lds_step = step_lds_bytes(p, W, 8, true);
#ifndef PSD_HOSTSIM
    if (lds_step > c->step_lds_set) {
        PSD_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(psd_rq_step),
                                      hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_step));
        c->step_lds_set = lds_step;
    }
#endif
    // multishift trains (default: up to 32 bulges; psd_set_train / PSD_TRAIN): M cursors, each with its own state,
    // descriptor and lists; cursor 0 is the ordinary state machine
    const int M = (c->train_m >= 2) ? ((c->train_m > PSD_TRAIN_MAX) ? PSD_TRAIN_MAX : c->train_m) : 1;
    P.cst = nullptr;
    P.tshift = nullptr;
    P.lead = c->st;
    P.tick = 0;
    P.gl = nullptr;
    P.cep = nullptr;
    P.role = P.epoch = P.cdone = nullptr;
    P.nprob = 1;
    P.ticklog = nullptr;
    P.ticklog_n = 0;
    P.bandinfo = nullptr;
    P.ccancel = nullptr;
    P.plan = nullptr;
    // two-wave chase (psd_c2_run): the command block sits in the reduction scratch behind the window image
    // scan chase (psd_chase3.h, the default): PSD_C3_WAVES wavefronts per chase workgroup, the command block where the
    // two-wave chase has it, the reflector table behind the scratch
    const bool scan3 = c->chase3 && p >= PSD_C3_MINP && p <= PSD_C3_MAXP;
    P.c2off = (scan3 || (c->chase2 && p >= PSD_C2_MINP)) ? (int)((size_t)p * psd_win_area(W) * sizeof(double)) : 0;
    P.c3off = scan3 ? (int)step_lds_scratch_end(p, W, 8, true) : 0;
    P.cdefer = 0;
    P.slG = 1;
    P.slmem = nullptr;
    P.slerr = nullptr;
    const int c2waves = scan3 ? PSD_C3_WAVES : (P.c2off ? 2 : 1);
    (void)c2waves;
    // Every way out of this function (the runaway cap, a failed runtime call) first waits for the second stream — its
    // Schur-vector launches read the caller's dZ and the lists — and frees the tick log.
    struct ExitGuard {
        psd_ctx* c;
        int** ticklog;
        ~ExitGuard() {
#ifndef PSD_HOSTSIM
            if (c->stream2) (void)hipStreamSynchronize(c->stream2);
#endif
            if (*ticklog) {
                psd_rt_free(*ticklog);
                *ticklog = nullptr;
            }
        }
    } exit_guard{c, &P.ticklog};
    const char* ticklog_path = psd_env_diag("PSD_TICKLOG");  // diagnostics: per tick the longest workgroup of the chase launch
    const int ticklog_cap = 1 << 16;
    if (ticklog_path && nprob == 1) {
        if (psd_rt_malloc((void**)&P.ticklog, sizeof(int) * ticklog_cap) == 0) {
            PSD_CHECK(psd_rt_memset(P.ticklog, 0, sizeof(int) * ticklog_cap, c->stream));
            P.ticklog_n = ticklog_cap;
        } else {
            P.ticklog = nullptr;
        }
    }
    if (bws) {
        if (nprob < 1 || nprob > PSD_SLOTS / 2) return -3;
        const size_t sb = (size_t)nprob * (n + 8);
        P.hdiag = bws; P.hsub = bws + sb; P.hsup = bws + 2 * sb; P.Pd = bws + 3 * sb; P.Pe = bws + 4 * sb;
        P.Pf = bws + 5 * sb; P.wr = bws + 6 * sb; P.wi = bws + 7 * sb;
        P.hnorms = bws + 8 * sb;
        P.nprob = nprob;
    }
    // multi-block scheduler (default with trains on; PSD_MB=0: one active range at a time): PSD_SLOTS workgroup slots,
    // leaders of independent active ranges and the cursors of their trains
    const bool mb = ((M > 1) && c->mblock) || bws != nullptr;  // (a batch always runs on the slot scheduler)
    const int NSL = mb ? PSD_SLOTS : M;
    const bool ovl2 = mb && (c->overlap == 2 || (c->overlap == 3 && n >= 1024));
    const bool zdef = ovl2 && wantZ;  // Schur-vector updates on the second stream, beside the next chases
    const bool cdef = mb && c->cdefer && (c->cdefer == 2 || ovl2) && nprob == 1;  // far rows of the column roles likewise (psd_cdefer_edge)
    bool rdef = cdef && (c->rdefer == 2 || (c->rdefer == 1 && n < 1536));  // and the far columns of the rows roles in front of them (psd_rdefer_edge)
#ifndef PSD_HOSTSIM
    if (zdef && !c->stream3) rdef = false;
#endif
    bool far_pending = false;  // (the far column roles of the previous tick have been launched / are still to run)
    P.cdefer = cdef ? 1 : 0;
    P.redge = c->redge;
    P.cedge = c->cedge;
    // factor-sliced sweep windows (psd_set_slices; psd_slice3.h): G workgroups per slot, each with the window blocks of
    // its contiguous slice of the period; needs the scan chase, the slot scheduler and at least two factors per slice
    int slG = 1;
#ifndef PSD_HOSTSIM
    if (c->slices > 1 && mb && scan3 && nprob == 1 && p / c->slices >= 2 && c->slices <= PSD_SL_MAXG) {
        slG = c->slices;
        // (every workgroup of the launch takes the LDS of a whole CU and the slices of a window wait for one another:
        //  all PSD_SLOTS x G of them have to be resident together — G <= 4 on the 256 CUs of an MI355X; more is clamped)
        while (slG > 1 && c->ncu > 0 && PSD_SLOTS * slG > c->ncu) slG /= 2;
        const size_t slbytes = (size_t)PSD_SLOTS * (PSD_SL_CMD_BYTES + PSD_SL_MAXG * PSD_SL_BOX_BYTES) + 64;
        if (!c->slmem) PSD_CHECK(psd_rt_malloc((void**)&c->slmem, slbytes));
        PSD_CHECK(psd_rt_memset(c->slmem, 0, slbytes, c->stream));
        P.slG = slG;
        P.slmem = c->slmem;
        P.slerr = (int*)(c->slmem + slbytes - 64);
    }
#endif
    if (M > 1 || mb) {
        PSD_CHECK(c->treserve(p));
        PSD_CHECK(psd_rt_memset(c->tgl, 0, sizeof(psd_rglobal), c->stream));
        PSD_CHECK(psd_rt_memset(c->tcst, 0, sizeof(psd_rstate) * PSD_SLOTS, c->stream));
        PSD_CHECK(psd_rt_memset(c->tdesc, 0, 2 * sizeof(psd_apply_desc) * PSD_SLOTS, c->stream));
        P.cst = c->tcst;
        P.cep = c->tslotw;  // (single-range train mode; the multi-block scheduler uses the same words as role/epoch/cdone)
        PSD_CHECK(psd_rt_memset(c->tslotw, 0xff, sizeof(int) * 6 * PSD_SLOTS, c->stream));  // (band info: tick -1)
        PSD_CHECK(psd_rt_memset(c->tslotw, 0, sizeof(int) * 3 * PSD_SLOTS, c->stream));
        PSD_CHECK(psd_rt_memset(c->tslotw + 6 * PSD_SLOTS, 0, sizeof(int) * 2 * PSD_SLOTS, c->stream));  // (stopped-train words)
        PSD_CHECK(psd_rt_memset(c->tslotw + 8 * PSD_SLOTS, 0xff, sizeof(int) * PSD_PLAN_INTS, c->stream));  // (slot plan: tick -1)
        P.tshift = c->tshift;
        P.desc = c->tdesc;  // slot 0 of the cursor arrays: the fused bulk-update kernel indexes them by cursor
        P.cnt = c->tcnt;
        P.tr = c->ttr;
        if (mb) {
            P.st = c->tcst;
            P.lead = c->tcst;
            P.gl = c->tgl;
            P.role = c->tslotw;
            P.epoch = c->tslotw + PSD_SLOTS;
            P.cdone = c->tslotw + 2 * PSD_SLOTS;
            if (c->band_helper) P.bandinfo = c->tslotw + 3 * PSD_SLOTS;
            if (c->train_stop) P.ccancel = c->tslotw + 6 * PSD_SLOTS;
            P.plan = c->tslotw + 8 * PSD_SLOTS;
        }
#ifndef PSD_HOSTSIM
        PSD_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(psd_rq_step_train),
                                      hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_step));
        PSD_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(psd_rq_step_mb),
                                      hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_step));
#endif
    }
    int train_oc = 104;
    if (const char* e = psd_env_diag("PSD_TRAIN_OC")) train_oc = atoi(e);  // (tuning hook)
    int Mw = M;
    if (mb && c->train_mb_m > Mw && c->train_m >= 32) Mw = (c->train_mb_m > PSD_TRAIN_MAX) ? PSD_TRAIN_MAX : c->train_mb_m;
    PSD_LAUNCH(psd_rq_init, psd_dim3(p, nprob), 256, 256 * 8, c->stream, P, n, p, wantT, wantZ, W, maxitfac, maxlog, Mw, train_oc,
               mb ? 1 : 0, mb ? c->cgap : 2, c->train_long, (c->train_wdiv > 0) ? c->train_wdiv : ((p >= 12 && p <= 32) ? 6 : ((p < 12 && n >= 512) ? 16 : 8)));
    const size_t lds_apply = apply_lds_bytes();
    const int tiles = (n + PSD_APPLY_NT - 1) / PSD_APPLY_NT;
    const int batch = 32;
    int zlo0 = 0, zhi0 = p;
    c->slice(p, zlo0, zhi0);
    const int zlo1 = zlo0 + 1, zhi1 = zhi0;  // owners (1-based, inclusive) whose Z_m this context updates
    psd_rstate hst;
    memset(&hst, 0, sizeof(hst));
    psd_rglobal hgl;
    memset(&hgl, 0, sizeof(hgl));
    long long launched = 0;
    // every window advances the chase by >= 1 position; generous cap against a runaway loop
    const long long cap = (long long)maxitfac * n * ((long long)n / 8 + 4) + 4LL * n + 1024;  // (trains run windows down to 8 positions)
    double sample_ms = 0.0;
    int samples = 0;
#ifndef PSD_HOSTSIM
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
    std::vector<std::pair<hipEvent_t, hipEvent_t>> pend;
    psd_poller poller(c, sample_ms, samples);
#endif
    for (;;) {
        for (int b = 0; b < batch; ++b) {
#ifndef PSD_HOSTSIM
            if ((zdef || cdef) && launched >= 2) PSD_CHECK(hipStreamWaitEvent(c->stream, c->evF[(int)(launched & 1)], 0));
#endif
            P.tick = (int)launched;
            // ticks alternate between two sets of descriptors / counts / lists (multi-block mode)
            const int par = mb ? (int)(launched & 1) : 0;
            psd_rparams Pq = P, Pprev = P;
            if (mb) {
                Pq.desc = P.desc + (size_t)par * PSD_SLOTS;
                Pq.cnt = P.cnt + (size_t)par * PSD_SLOTS * (p + 8);
                Pq.tr = P.tr + (size_t)par * PSD_SLOTS * p * PSD_TR_CAP;
                Pprev.desc = P.desc + (size_t)(par ^ 1) * PSD_SLOTS;
                Pprev.cnt = P.cnt + (size_t)(par ^ 1) * PSD_SLOTS * (p + 8);
                Pprev.tr = P.tr + (size_t)(par ^ 1) * PSD_SLOTS * p * PSD_TR_CAP;
            }
            // (multi-block: in front of the chase launch the product bands of the wide decisions pending, spread over the chip)
            // and the slot plan of the tick (last row of the grid; problems narrower than the band threshold launch that row alone)
            if (mb) {
                const bool bands = Pq.bandinfo && n >= PSD_DECIDE_YIELD;
                PSD_LAUNCH(psd_rq_band, psd_dim3(bands ? (n + 63) / 64 : 1, bands ? PSD_SLOTS + 1 : 1), 64,
                           sizeof(int) * PSD_PLAN_LDS_INTS, c->stream, Pq, n, p);
            }
#ifndef PSD_HOSTSIM
            const bool sample = c->profile && ((launched & 15) == 0);  // HIP events around the chase launch alone
            if (sample) {
                (void)hipEventCreate(&ev0);
                (void)hipEventCreate(&ev1);
                (void)hipEventRecord(ev0, c->stream);
            }
#endif
            if (M == 1 && !mb)
                PSD_LAUNCH2(psd_rq_step, psd_dim3(1), PSD_STEP_NT, c2waves, lds_step, c->stream, P);
            else if (mb)  // every slot of the scheduler in one launch
                PSD_LAUNCH2(psd_rq_step_mb, psd_dim3(PSD_SLOTS, slG), PSD_STEP_NT, c2waves, lds_step, c->stream, Pq, p, p + 8);
            else  // every cursor of the tick in one launch, one workgroup each
                PSD_LAUNCH2(psd_rq_step_train, psd_dim3(M), PSD_STEP_NT, c2waves, lds_step, c->stream, P, p, p + 8);
#ifndef PSD_HOSTSIM
            if (sample) {
                (void)hipEventRecord(ev1, c->stream);
                pend.emplace_back(ev0, ev1);
            }
#endif
            if (rdef) {
                // Both far parts of the H updates on stream2 (psd_rdefer_edge, psd_cdefer_edge), in the order rows, columns:
                // the far columns of this tick's rows roles start as soon as the chases are done (evC) and run beside the
                // near parts on this stream, the far rows of the column roles follow when those near parts are done (evE);
                // the NEXT tick's near parts wait for both (evG).  The Schur vectors go to a stream of their own (stream3)
                // so that they do not hold up the far H updates the next tick waits for.  The serial simulation runs both
                // far parts at the latest point the streams allow, behind the next tick's chases.
                const int wl_grid = c->apply_wl_grid;
#ifndef PSD_HOSTSIM
                PSD_CHECK(hipEventRecord(c->evC[par], c->stream));
#endif
                if (far_pending) {
#ifndef PSD_HOSTSIM
                    PSD_CHECK(hipStreamWaitEvent(c->stream, c->evG[par ^ 1], 0));
#else
                    PSD_CHECK(launch_apply_wl(c, c->stream, Pprev, n, p, 0, NSL, zlo1, zhi1, 8, W, wl_grid));
                    PSD_CHECK(launch_apply_wl(c, c->stream, Pprev, n, p, 1, NSL, zlo1, zhi1, 6, W, wl_grid));
#endif
                    far_pending = false;
                }
                PSD_CHECK(launch_apply_wl(c, c->stream, Pq, n, p, 0, NSL, zlo1, zhi1, 7, W, wl_grid));
                if (wantZ && !zdef) PSD_CHECK(launch_apply_wl(c, c->stream, Pq, n, p, 0, NSL, zlo1, zhi1, 4, W, wl_grid));
                PSD_CHECK(launch_apply_wl(c, c->stream, Pq, n, p, 1, NSL, zlo1, zhi1, 5, W, wl_grid));
#ifndef PSD_HOSTSIM
                PSD_CHECK(hipEventRecord(c->evE[par], c->stream));
                const int fgrid = c->far_grid > 0 ? c->far_grid : wl_grid;
                PSD_CHECK(hipStreamWaitEvent(c->stream2, c->evC[par], 0));
                PSD_CHECK(launch_apply_wl(c, c->stream2, Pq, n, p, 0, NSL, zlo1, zhi1, 8, W, fgrid));
                PSD_CHECK(hipStreamWaitEvent(c->stream2, c->evE[par], 0));
                PSD_CHECK(launch_apply_wl(c, c->stream2, Pq, n, p, 1, NSL, zlo1, zhi1, 6, W, fgrid));
                PSD_CHECK(hipEventRecord(c->evG[par], c->stream2));
                if (zdef) {
                    hipStream_t zs = (c->zstream == 2) ? c->stream2 : c->stream3;  // (PSD_ZSTREAM=2: behind the far H updates on stream2)
                    if (zs != c->stream2) PSD_CHECK(hipStreamWaitEvent(zs, c->evC[par], 0));
                    PSD_CHECK(launch_apply_wl(c, zs, Pq, n, p, 0, NSL, zlo1, zhi1, 4, W, fgrid));
                    PSD_CHECK(hipEventRecord(c->evF[par], zs));
                } else {
                    PSD_CHECK(hipEventRecord(c->evF[par], c->stream2));
                }
#else
                if (zdef) PSD_CHECK(launch_apply_wl(c, c->stream, Pq, n, p, 0, NSL, zlo1, zhi1, 4, W, wl_grid));
#endif
                far_pending = true;
            } else if (zdef || cdef) {
                // Schur-vector updates one stream over (psd_rq_apply_wl modes 3 / 4): the Z launch of this tick starts on
                // stream2 when the tick's H updates are done (evE) — beside the NEXT tick's chases, which leave most of the
                // chip idle — and its lists must not be rewritten before it is done (evF, awaited in front of the chase
                // that reuses this parity's lists, two ticks on).  The serial simulation runs the Z launch last.
                // The far rows of the column roles (modes 5 / 6, psd_cdefer_edge) go the same way, in front of the Z launch:
                // the next tick's rows roles cross them, so that launch waits for them (evG); nothing else reads them
                // earlier.  The serial simulation runs them at the latest point the streams allow — behind the next
                // tick's chases — which shows that no chase, band or decision depends on them.
                const int wl_grid = c->apply_wl_grid;
                const int hmode = cdef ? 5 : 3;
                if (cdef && far_pending) {
#ifndef PSD_HOSTSIM
                    PSD_CHECK(hipStreamWaitEvent(c->stream, c->evG[par ^ 1], 0));
#else
                    PSD_CHECK(launch_apply_wl(c, c->stream, Pprev, n, p, 1, NSL, zlo1, zhi1, 6, W, wl_grid));
#endif
                    far_pending = false;
                }
                if (!zdef && wantZ) {  // (the Schur vectors stay on this stream)
                    PSD_CHECK(launch_apply_wl(c, c->stream, Pq, n, p, 0, NSL, zlo1, zhi1, 0, W, wl_grid));
                } else {
                    PSD_CHECK(launch_apply_wl(c, c->stream, Pq, n, p, 0, NSL, zlo1, zhi1, 3, W, wl_grid));
                }
                PSD_CHECK(launch_apply_wl(c, c->stream, Pq, n, p, 1, NSL, zlo1, zhi1, hmode, W, wl_grid));
#ifndef PSD_HOSTSIM
                PSD_CHECK(hipEventRecord(c->evE[par], c->stream));
                PSD_CHECK(hipStreamWaitEvent(c->stream2, c->evE[par], 0));
                const int fgrid = c->far_grid > 0 ? c->far_grid : wl_grid;
                if (cdef) {
                    PSD_CHECK(launch_apply_wl(c, c->stream2, Pq, n, p, 1, NSL, zlo1, zhi1, 6, W, fgrid));
                    PSD_CHECK(hipEventRecord(c->evG[par], c->stream2));
                    far_pending = true;
                }
                if (zdef) PSD_CHECK(launch_apply_wl(c, c->stream2, Pq, n, p, 0, NSL, zlo1, zhi1, 4, W, fgrid));
                PSD_CHECK(hipEventRecord(c->evF[par], c->stream2));
#else
                if (cdef) far_pending = true;
                if (zdef) PSD_CHECK(launch_apply_wl(c, c->stream, Pq, n, p, 0, NSL, zlo1, zhi1, 4, W, wl_grid));
#endif
            } else if (c->apply_worklist || c->shard_world > 1 || mb) {
                // work-list form: one grid of single-wave workgroups loops over the items of the tick
                const int wl_grid = c->apply_wl_grid;
                if (c->apply_wl2 == 1 && W <= 17 && wantZ) {
                    // (rows role and Schur vectors as separate launches: each on the form that is faster for it)
                    PSD_CHECK(launch_apply_wl(c, c->stream, Pq, n, p, 0, NSL, zlo1, zhi1, 3, W, wl_grid));
                    PSD_CHECK(launch_apply_wl(c, c->stream, Pq, n, p, 0, NSL, zlo1, zhi1, 4, W, wl_grid));
                } else {
                    PSD_CHECK(launch_apply_wl(c, c->stream, Pq, n, p, 0, NSL, zlo1, zhi1, 0, W, wl_grid));
                }
                PSD_CHECK(launch_apply_wl(c, c->stream, Pq, n, p, 1, NSL, zlo1, zhi1, 0, W, wl_grid));
            } else if (M == 1) {
                PSD_LAUNCH(psd_rq_apply, psd_dim3(tiles, p, 3), PSD_APPLY_NT, lds_apply, c->stream, P, n, p);
            } else {
                PSD_LAUNCH(psd_rq_apply_train, psd_dim3(tiles, p, 2 * M), PSD_APPLY_NT, lds_apply, c->stream, P, n, p, p + 8, 0);
                PSD_LAUNCH(psd_rq_apply_train, psd_dim3(tiles, p, M), PSD_APPLY_NT, lds_apply, c->stream, P, n, p, p + 8, 1);
            }
            ++launched;
        }
        if (mb) {
#ifdef PSD_HOSTSIM
            PSD_CHECK(psd_rt_d2h(&hgl, c->tgl, sizeof(hgl), c->stream));
            PSD_CHECK(psd_rt_sync(c->stream));
#else
            PSD_CHECK(poller.poll(&hgl, c->tgl, sizeof(hgl), pend));
#endif
            if (hgl.done) {
                hst.phase = PSD_PH_DONE;
                hst.info = hgl.info;
                hst.niter = hgl.niter;
                hst.maxits = hgl.maxits;
                hst.nsweeps = hgl.nsweeps;
                hst.nrqpass = hgl.nrqpass;
                hst.ndefl1 = hgl.ndefl1;
                hst.ndefl2 = hgl.ndefl2;
                hst.nwindows = hgl.nwindows;
                hst.nlog = hgl.nlog;
                hst.ntrains = hgl.ntrains;
                hst.ntrainsweeps = hgl.ntrainsweeps;
                for (int q = 0; q < 6; ++q) hst.cyc[q] = hgl.cyc[q];
                if (pinfo_out)  // (a call-wide abort — a list overflow — leaves the problems still active unfinished: say so)
                    for (int q = 0; q < nprob; ++q)
                        pinfo_out[q] = (hgl.abort && hgl.pinfo[q] == 0 && hgl.pactive[q] > 0) ? hgl.info : hgl.pinfo[q];
                break;
            }
            if (launched > cap) {
#ifdef PSD_HOSTSIM
                fprintf(stderr, "mb runaway: launched %lld nactive %d nspawn %d nslotmax %d nsweeps %d\n", launched, hgl.nactive,
                        hgl.nspawn, hgl.nslotmax, hgl.nsweeps);
                for (int q = 0; q < PSD_SLOTS; ++q) {
                    const psd_rstate& x = c->tcst[q];
                    if (c->tslotw[q] != PSD_ROLE_FREE)
                        fprintf(stderr, " slot %d role %d epoch %d phase %d cursor %d parent %d lo %d l %d i %d its %d train_n %d key %d kcur %d tick0 %d\n",
                                q, c->tslotw[q], c->tslotw[PSD_SLOTS + q], x.phase, x.cursor, x.parent, x.lo, x.l, x.i, x.its,
                                x.train_n, x.train_key, x.kcur, x.train_tick0);
                    if (c->tslotw[q] == PSD_ROLE_LEADER && x.phase == PSD_PH_TWAIT)
                        for (int b = 1; b < x.train_n; ++b)
                            fprintf(stderr, "   cursor %d slot %d cdone %d (slot state: cursor %d parent %d key %d phase %d kcur %d i %d l %d tick0 %d)\n", b, x.cslots[b],
                                    c->tslotw[2 * PSD_SLOTS + x.cslots[b]], c->tcst[x.cslots[b]].cursor, c->tcst[x.cslots[b]].parent,
                                    c->tcst[x.cslots[b]].train_key, c->tcst[x.cslots[b]].phase, c->tcst[x.cslots[b]].kcur,
                                    c->tcst[x.cslots[b]].i, c->tcst[x.cslots[b]].l, c->tcst[x.cslots[b]].train_tick0);
                }
#endif
                *st_out = hst;
                return PSD_INFO_RUNTIME + 0xfffe;
            }
            continue;
        }
#ifdef PSD_HOSTSIM
        PSD_CHECK(psd_rt_d2h(&hst, c->st, sizeof(hst), c->stream));
        PSD_CHECK(psd_rt_sync(c->stream));
#else
        PSD_CHECK(poller.poll(&hst, c->st, sizeof(hst), pend));
#endif
        if (hst.phase == PSD_PH_DONE) break;
        if (launched > cap) {
            *st_out = hst;
            return PSD_INFO_RUNTIME + 0xfffe;
        }
    }
#ifndef PSD_HOSTSIM
    if (zdef || cdef) PSD_CHECK(hipStreamSynchronize(c->stream2));
    if (rdef && zdef) PSD_CHECK(hipStreamSynchronize(c->stream3));
#else
    if (cdef && far_pending) {  // (the last tick's far column roles: its parity is the one the last launch used)
        psd_rparams Pl = P;
        const int parl = (int)((launched - 1) & 1);
        Pl.desc = P.desc + (size_t)parl * PSD_SLOTS;
        Pl.cnt = P.cnt + (size_t)parl * PSD_SLOTS * (p + 8);
        Pl.tr = P.tr + (size_t)parl * PSD_SLOTS * p * PSD_TR_CAP;
        if (rdef) PSD_CHECK(launch_apply_wl(c, c->stream, Pl, n, p, 0, NSL, zlo1, zhi1, 8, W, c->apply_wl_grid));
        PSD_CHECK(launch_apply_wl(c, c->stream, Pl, n, p, 1, NSL, zlo1, zhi1, 6, W, c->apply_wl_grid));
        far_pending = false;
    }
#endif
#ifndef PSD_HOSTSIM
    PSD_CHECK(poller.finish(pend));
#endif
    PSD_CHECK(psd_rt_last_error());
#ifndef PSD_HOSTSIM
    if (slG > 1) {  // (a sliced window whose hand-over never arrived: the results are void, say so)
        int serr = 0;
        PSD_CHECK(psd_rt_d2h(&serr, P.slerr, sizeof(int), c->stream));
        PSD_CHECK(psd_rt_sync(c->stream));
        if (serr) {
            *st_out = hst;
            return PSD_INFO_RUNTIME + 0xfffa;
        }
    }
#endif
    if (P.ticklog) {
        std::vector<int> tl((size_t)ticklog_cap);
        PSD_CHECK(psd_rt_d2h(tl.data(), P.ticklog, sizeof(int) * ticklog_cap, c->stream));
        PSD_CHECK(psd_rt_sync(c->stream));
        fprintf(stderr, "psd ticklog: phase cycles (decide, spawn, train shifts, small QR, claims, cursor states, deflate, RQ window):");
        for (int q = 0; q < 8; ++q) fprintf(stderr, " %lld/%d", hgl.dbg[q], hgl.dbgn[q]);
        fprintf(stderr, "\n");
        if (hgl.c3dbg[7] > 0) {
            fprintf(stderr, "psd ticklog: scan chase, cycles per position (loads + scan 1, reflectors + B, scan 2, 2-reflectors + records, barrier, apply 0, apply 1) over %lld positions:", hgl.c3dbg[7]);
            for (int q = 0; q < 7; ++q) fprintf(stderr, " %.0f", (double)hgl.c3dbg[q] / (double)hgl.c3dbg[7]);
            fprintf(stderr, "\n");
        }
        if (FILE* fh = fopen(ticklog_path, "w")) {
            for (long long t = 0; t < launched && t < ticklog_cap; ++t) fprintf(fh, "%lld %d %d\n", t, tl[(size_t)t] >> 12, tl[(size_t)t] & 0xfff);
            fclose(fh);
        }
    }
    *st_out = hst;
    if (stats) {
        stats->nlaunch_step = (int32_t)launched;
        stats->window = W;
        stats->step_kernel_ms_avg = samples ? sample_ms / samples : 0.0;
        stats->step_kernel_samples = samples;
    }
    return 0;
}

void stats_from_state(psd_stats* s, const psd_rstate& st) {
    if (!s) return;
    s->niter = st.niter;
    s->maxits = st.maxits;
    s->nsweeps = st.nsweeps;
    s->nrqpass = st.nrqpass;
    s->ndefl1 = st.ndefl1;
    s->ndefl2 = st.ndefl2;
    s->nwindows = st.nwindows;
    s->nlog = st.nlog;
    for (int q = 0; q < 6; ++q) s->step_cycles[q] = st.cyc[q];
    s->reserved = st.ntrainsweeps;
}

// shared tail: run the iteration, fetch eigenvalues / log
int run_iteration(psd_ctx* c, int n, int p, double* dH, double* dZ, int wantT, int wantZ, int maxitfac, double* wr,
                  double* wi, psd_stats* stats, int32_t* sweeplog, int64_t maxlog_user, int* info) {
    const int maxlog = 2 * maxitfac * n + n + 16;
    if (n == 1) {  // PSD.jl:333-352
        PSD_LAUNCH(psd_scalar_product, psd_dim3(1), 64, 0, c->stream, (const double*)dH, p, c->wr, c->wi);
        if (wantZ && dZ) PSD_LAUNCH(psd_set_identity, psd_dim3(1, p), 64, 0, c->stream, dZ, 1);
        PSD_CHECK(psd_rt_d2h(wr, c->wr, sizeof(double), c->stream));
        PSD_CHECK(psd_rt_d2h(wi, c->wi, sizeof(double), c->stream));
        PSD_CHECK(psd_rt_sync(c->stream));
        *info = 0;
        return 0;
    }
    psd_rstate st;
    int rc = iterate_dev(c, n, p, dH, dZ, wantT, wantZ, maxitfac, &st, stats, maxlog);
    if (rc != 0) {
        *info = rc;
        return rc;
    }
    stats_from_state(stats, st);
    PSD_CHECK(psd_rt_d2h(wr, c->wr, sizeof(double) * n, c->stream));
    PSD_CHECK(psd_rt_d2h(wi, c->wi, sizeof(double) * n, c->stream));
    const int nl = st.nlog < maxlog ? st.nlog : maxlog;
    std::vector<int> hlog((size_t)3 * nl + 3, 0);
    if (nl > 0) PSD_CHECK(psd_rt_d2h(hlog.data(), c->log, sizeof(int) * 3 * (size_t)nl, c->stream));
    PSD_CHECK(psd_rt_sync(c->stream));
    hlog.resize((size_t)3 * nl);
    fill_bytes(stats, n, p, wantT, wantZ, hlog);
    if (sweeplog) {
        const int64_t m = nl < maxlog_user ? nl : maxlog_user;
        for (int64_t q = 0; q < 3 * m; ++q) sweeplog[q] = hlog[q];
    }
    *info = (st.info == PSD_LIST_OVERFLOW) ? (PSD_INFO_RUNTIME + 77) : ((st.info != 0) ? (PSD_INFO_NOCONV + st.info) : 0);
    return *info;
}

int check_dims(int n, int p) {
    if (n < 1) return -2;
    if (p < 1) return -3;
    return 0;
}

}  // namespace

extern "C" {

const char* psd_version(void) { return "psd_mi355x 0.1 (" PSD_LIB_FLAVOUR ")"; }

int psd_create(psd_ctx** ctx, int device) {
    if (!ctx) return -1;
    *ctx = nullptr;
#ifndef PSD_HOSTSIM
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) return PSD_INFO_RUNTIME + 1;
    if (device < 0 || device >= ndev) return -2;
    if (hipSetDevice(device) != hipSuccess) return PSD_INFO_RUNTIME + 2;
#endif
    psd_ctx* c = new (std::nothrow) psd_ctx();
    if (!c) return PSD_INFO_RUNTIME + 3;
    c->device = device;
#ifndef PSD_HOSTSIM
    if (hipStreamCreate(&c->stream) != hipSuccess) {
        delete c;
        return PSD_INFO_RUNTIME + 4;
    }
    for (int q = 0; q < 2; ++q) {
        if (hipHostMalloc(&c->pin[q], 4096, hipHostMallocDefault) != hipSuccess ||
            hipEventCreateWithFlags(&c->pev[q], hipEventDisableTiming) != hipSuccess ||
            hipEventCreateWithFlags(&c->evE[q], hipEventDisableTiming) != hipSuccess ||
            hipEventCreateWithFlags(&c->evF[q], hipEventDisableTiming) != hipSuccess ||
            hipEventCreateWithFlags(&c->evG[q], hipEventDisableTiming) != hipSuccess ||
            hipEventCreateWithFlags(&c->evC[q], hipEventDisableTiming) != hipSuccess) {
            psd_destroy(c);
            return PSD_INFO_RUNTIME + 5;
        }
    }
    if (const char* e = psd_env("PSD_OVERLAP")) c->overlap = atoi(e);
    {
        // The far bulk updates run beside the next tick's chases.  A chase workgroup takes the whole LDS of a compute
        // unit (p*W*(W+1) doubles), so it can only start on a CU that holds no bulk-update workgroup: the second stream
        // is confined to the CUs the chases do not need (CU mask), PSD_OVERLAP_CUS of them stay free for the slots.
        int ncu = 0;
        (void)hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, device);
        c->ncu = ncu;
        int keep = PSD_SLOTS;
        if (const char* e = psd_env_diag("PSD_OVERLAP_CUS")) keep = atoi(e);
        hipError_t rc = hipErrorInvalidValue;
        if (c->overlap && ncu > keep + 32 && keep > 0 && ncu <= 1024) {
            uint32_t mask[32];
            memset(mask, 0, sizeof(mask));
            for (int q = keep; q < ncu; ++q) mask[q >> 5] |= (1u << (q & 31));
            rc = hipExtStreamCreateWithCUMask(&c->stream2, (uint32_t)((ncu + 31) / 32), mask);
            if (rc == hipSuccess) c->far_grid = (ncu - keep) * 8;
        }
        if (rc != hipSuccess) rc = hipStreamCreate(&c->stream2);
        if (rc != hipSuccess) {
            psd_destroy(c);
            return PSD_INFO_RUNTIME + 4;
        }
        // the same for the panel updates of the Hessenberg reduction: its chain launches are HBM-latency chains that
        // slow down under the panel traffic; half of the chip for the panels measured best (PSD_HESS_CUS)
        int keeph = 64;  // (128 until the chain launches overlapped: hessenberg2_pipe)
        if (const char* e = psd_env_diag("PSD_HESS_CUS")) keeph = atoi(e);
        rc = hipErrorInvalidValue;
        if (ncu > keeph + 32 && keeph > 0 && ncu <= 1024) {
            uint32_t mask[32];
            memset(mask, 0, sizeof(mask));
            for (int q = keeph; q < ncu; ++q) mask[q >> 5] |= (1u << (q & 31));
            rc = hipExtStreamCreateWithCUMask(&c->stream3, (uint32_t)((ncu + 31) / 32), mask);
        }
        if (rc != hipSuccess) rc = hipStreamCreate(&c->stream3);
        if (rc != hipSuccess) {
            psd_destroy(c);
            return PSD_INFO_RUNTIME + 4;
        }
        if (hipStreamCreate(&c->stream4) != hipSuccess) c->stream4 = nullptr;
        if (const char* e = psd_env("PSD_H2_PIPE")) c->hess_pipe = atoi(e);
        if (const char* e = psd_env("PSD_H2_DEPTH")) c->hess_pipe_depth = atoi(e);
    }
#endif
#ifndef PSD_HOSTSIM
    if (const char* e = psd_env("PSD_HESS_LOOKAHEAD")) c->hess_lookahead = atoi(e);
    if (const char* e = psd_env("PSD_HESS_ASYNC")) c->hess_async = atoi(e);
#endif
#ifdef PSD_HOSTSIM
    if (const char* e = psd_env("PSD_OVERLAP")) c->overlap = atoi(e);
#endif
    if (const char* e = psd_env("PSD_CDEFER")) c->cdefer = atoi(e);
    if (const char* e = psd_env("PSD_RDEFER")) c->rdefer = atoi(e);
    if (const char* e = psd_env("PSD_ZCDEFER")) c->zcdefer = atoi(e);
    if (const char* e = psd_env("PSD_ZSTREAM")) c->zstream = atoi(e);
    if (const char* e = psd_env("PSD_RDEFER_EDGE")) c->redge = atoi(e);
    if (const char* e = psd_env("PSD_CDEFER_EDGE")) c->cedge = atoi(e);
    if (const char* e = psd_env("PSD_ORD_PIPE")) c->ord_pipe = atoi(e);
    if (const char* e = psd_env("PSD_FORMQ_BLOCKED")) c->formq_blocked = atoi(e);
    if (const char* e = psd_env("PSD_BAND_HELPER")) c->band_helper = atoi(e);
    if (const char* e = psd_env_diag("PSD_TRAIN_LONG")) c->train_long = atoi(e);
    if (const char* e = psd_env_diag("PSD_TRAIN_STOP")) c->train_stop = atoi(e);
    if (const char* e = psd_env_diag("PSD_TRAIN_WDIV")) c->train_wdiv = atoi(e) > 0 ? atoi(e) : 0;
    if (const char* e = psd_env("PSD_MB")) c->mblock = atoi(e);
    if (const char* e = psd_env_diag("PSD_TRAIN_MB")) c->train_mb_m = atoi(e);
    if (const char* e = psd_env_diag("PSD_APPLY_WL")) c->apply_worklist = atoi(e);
    if (const char* e = psd_env_diag("PSD_APPLY_WL_GRID")) c->apply_wl_grid = atoi(e) > 0 ? atoi(e) : 2048;
    if (const char* e = psd_env_diag("PSD_APPLY_WL2")) c->apply_wl2 = atoi(e);
    if (const char* e = psd_env("PSD_C2")) c->chase2 = atoi(e);
    if (const char* e = psd_env("PSD_C3")) c->chase3 = atoi(e);
#ifndef PSD_HOSTSIM
#endif
    if (const char* e = psd_env_diag("PSD_APPLY_WL2_GRID")) c->apply_wl2_grid = atoi(e) > 0 ? atoi(e) : 1024;
#ifdef PSD_HOSTSIM
    c->apply_wl_grid = 3;  // (serial simulation: a few workgroups exercise the item loop)
#endif
    if (const char* e = psd_env("PSD_TRAIN")) c->train_m = c->ztrain_m = c->gtrain_m = atoi(e);
    if (const char* e = psd_env("PSD_TRAIN_Z")) c->ztrain_m = atoi(e);
    if (const char* e = psd_env("PSD_TRAIN_G")) c->gtrain_m = atoi(e);
    c->counted = true;
    // The form of the multi-stream Hessenberg reductions is a property of the context, fixed here: the pipe form for a
    // context that is created while no other context of the process is alive, the back-to-back form for every other one
    // (the two forms round differently — plain sum of squares against dlassq —, so a context must not change form from
    // call to call depending on what else happens to exist at that moment).  psd_get_hess_pipe reports it.
    c->pipe_ok = g_live_contexts.fetch_add(1) == 0;
    *ctx = c;
    return 0;
}

int psd_set_train(psd_ctx* c, int bulges) {
    if (!c) return -1;
    c->train_m = (bulges < 0) ? 0 : ((bulges > 32) ? 32 : bulges);
    c->ztrain_m = c->gtrain_m = c->train_m;
    if (c->train_m >= 32) c->ztrain_m = c->gtrain_m = 48;  // (the maximum = every engine's own default)
    return 0;
}

int psd_get_train(psd_ctx* c) { return c ? c->train_m : -1; }
int psd_set_train_z(psd_ctx* c, int bulges) {
    if (!c) return -1;
    c->ztrain_m = (bulges < 0) ? 0 : ((bulges > PSD_TRAIN_MAX) ? PSD_TRAIN_MAX : bulges);
    return 0;
}
int psd_get_train_z(psd_ctx* c) { return c ? c->ztrain_m : -1; }
int psd_set_train_g(psd_ctx* c, int bulges) {
    if (!c) return -1;
    c->gtrain_m = (bulges == -2) ? -2 : ((bulges < 0) ? 0 : ((bulges > PSD_TRAIN_MAX) ? PSD_TRAIN_MAX : bulges));
    return 0;
}
int psd_get_train_g(psd_ctx* c) { return c ? c->gtrain_m : -1; }
int psd_get_hess_pipe(psd_ctx* c) { return c ? (psd_pipe_form(c) ? 1 : 0) : -1; }
int psd_set_slices(psd_ctx* c, int slices) {
    if (!c) return -1;
    if (slices < 1 || slices > PSD_SL_MAXG_API) return -2;
    c->slices = slices;
    return 0;
}
int psd_get_slices(psd_ctx* c) { return c ? c->slices : -1; }

int psd_destroy(psd_ctx* c) {
    if (!c) return 0;
    if (c->slmem) psd_rt_free(c->slmem);
    c->slmem = nullptr;
    if (c->counted) g_live_contexts.fetch_sub(1);
    c->counted = false;
    c->grelease();
    c->gtrelease();
    c->zgrelease();
    c->zgtrelease();
    c->release();
    c->zrelease();
    c->rorelease();
    c->trelease();
    c->ztrelease();
#ifndef PSD_HOSTSIM
    for (int q = 0; q < 2; ++q) {
        if (c->pev[q]) (void)hipEventDestroy(c->pev[q]);
        if (c->evE[q]) (void)hipEventDestroy(c->evE[q]);
        if (c->evF[q]) (void)hipEventDestroy(c->evF[q]);
        if (c->evG[q]) (void)hipEventDestroy(c->evG[q]);
        if (c->evC[q]) (void)hipEventDestroy(c->evC[q]);
        if (c->pin[q]) (void)hipHostFree(c->pin[q]);
    }
    for (auto& e : c->h2ev)
        if (e) (void)hipEventDestroy(e);
    if (c->stream2) (void)hipStreamDestroy(c->stream2);
    if (c->stream3) (void)hipStreamDestroy(c->stream3);

    if (c->stream4) (void)hipStreamDestroy(c->stream4);
    if (c->stream) (void)hipStreamDestroy(c->stream);
#endif
    delete c;
    return 0;
}

int psd_set_shard(psd_ctx* c, int rank, int world) {
    if (!c) return -1;
    if (world < 1) return -3;
    if (rank < 0 || rank >= world) return -2;
    c->shard_rank = rank;
    c->shard_world = world;
    return 0;
}

int psd_shard_owned(psd_ctx* c, int p, char orient, uint8_t* owned) {
    if (!c) return -1;
    if (p < 1) return -2;
    if (orient != 'R' && orient != 'L') return -3;
    if (!owned) return -4;
    int lo = 0, hi = p;
    c->slice(p, lo, hi);
    for (int s = 0; s < p; ++s) owned[s] = 0;
    for (int j = lo; j < hi; ++j) {
        // internal factor j (0-based) -> user slot of Z_j: 'R' identity; 'L': Z_1 stays, Z_2..Z_p reverse (PSD.jl:1078-1092)
        const int slot = (orient == 'R' || j == 0) ? j : (p - j);
        owned[slot] = 1;
    }
    return 0;
}

int psd_set_profile(psd_ctx* c, int profile) {
    if (!c) return -1;
    c->profile = profile;
    return 0;
}

int psd_d_phessenberg(psd_ctx* c, int n, int p, double* const* A, double* tau, psd_stats* stats, int* info) {
    int dummy;
    if (!info) info = &dummy;
    if (stats) memset(stats, 0, sizeof(*stats));
    if (!c) return *info = -1;
    if ((*info = check_dims(n, p)) != 0) return *info;
    if (!A || !tau) return *info = -4;
    const int maxlog = 16;
    if ((*info = c->reserve(n, p, true, maxlog)) != 0) return *info;
    const size_t nn = (size_t)n * n;
    Timer tc, tk;
    tc.start(c->stream);
    for (int j = 0; j < p; ++j) PSD_CHECK(psd_rt_h2d(c->dH + j * nn, A[j], nn * 8, c->stream));
    double ms_copy = tc.stop(c->stream);
    tk.start(c->stream);
    if ((*info = hessenberg_dev(c, n, p, c->dH, c->tau)) != 0) return *info;
    double ms = tk.stop(c->stream);
    tc.start(c->stream);
    for (int j = 0; j < p; ++j) PSD_CHECK(psd_rt_d2h(A[j], c->dH + j * nn, nn * 8, c->stream));
    PSD_CHECK(psd_rt_d2h(tau, c->tau, sizeof(double) * (size_t)n * p, c->stream));
    PSD_CHECK(psd_rt_sync(c->stream));
    ms_copy += tc.stop(c->stream);
    PSD_CHECK(psd_rt_last_error());
    if (stats) {
        stats->ms_hess = ms;
        stats->ms_total = ms;
        stats->ms_copy = ms_copy;
        stats->bytes_hess = 2.0 * 8.0 * p * (5.0 / 6.0) * (double)n * n * n;
    }
    return *info = 0;
}

int psd_d_pschur_dev(psd_ctx* c, int n, int p, double* dA, char orient, int wantT, int wantZ, int maxitfac, double* dZ,
                     double* wr, double* wi, int* schurindex, psd_stats* stats, int32_t* sweeplog, int64_t maxlog,
                     int* info) {
    int dummy;
    if (!info) info = &dummy;
    if (stats) memset(stats, 0, sizeof(*stats));
    if (!c) return *info = -1;
    if ((*info = check_dims(n, p)) != 0) return *info;
    if (!dA) return *info = -4;
    if (orient != 'R' && orient != 'L') return *info = -5;  // PSD.jl:175-177
    if (maxitfac < 1) return *info = -8;
    if (wantZ && !dZ) return *info = -9;
    if (!wr || !wi) return *info = -10;
    const bool left = orient == 'L';
    const int mlog = 2 * maxitfac * n + n + 16;
    if ((*info = c->reserve(n, p, false, mlog)) != 0) return *info;
    Timer tall, tph;
    tall.start(c->stream);
    // PSD.jl:127-131: 'L' works on the reversed sequence
    if (left && p > 1) PSD_LAUNCH(psd_reverse_blocks, psd_dim3(n, p / 2), 64, 0, c->stream, dA, n, n, 0, p);
    tph.start(c->stream);
    if ((*info = hessenberg_dev(c, n, p, dA, c->tau)) != 0) return *info;
    const double ms_hess = tph.stop(c->stream);
    tph.start(c->stream);
    if (wantZ) {
        if ((*info = formq_dev(c, n, p, dA, c->tau, dZ)) != 0) return *info;
    }
    PSD_LAUNCH(psd_triu, psd_dim3(n, p), 64, 0, c->stream, dA, n);
    const double ms_formq = tph.stop(c->stream);
    tph.start(c->stream);
    psd_stats local;
    memset(&local, 0, sizeof(local));
    psd_stats* s = stats ? stats : &local;
    int rc = run_iteration(c, n, p, dA, dZ, wantT, wantZ, maxitfac, wr, wi, s, sweeplog, maxlog, info);
    const double ms_iter = tph.stop(c->stream);
    // PSD.jl:1078-1092: undo the reversal; Z_1 stays, Z_2..Z_p reverse
    if (left && p > 1) {
        PSD_LAUNCH(psd_reverse_blocks, psd_dim3(n, p / 2), 64, 0, c->stream, dA, n, n, 0, p);
        if (wantZ && p > 2) PSD_LAUNCH(psd_reverse_blocks, psd_dim3(n, (p - 1) / 2), 64, 0, c->stream, dZ, n, n, 1, p - 1);
    }
    s->ms_hess = ms_hess;
    s->ms_formq = ms_formq;
    s->ms_iter = ms_iter;
    s->ms_total = tall.stop(c->stream);
    s->bytes_hess = 2.0 * 8.0 * p * (5.0 / 6.0) * (double)n * n * n;
    s->bytes_formq = wantZ ? 2.0 * 8.0 * p * (double)n * n * n / 3.0 : 0.0;
    if (schurindex) *schurindex = left ? p : 1;
    return rc;
}

int psd_d_pschur(psd_ctx* c, int n, int p, double* const* A, const uint8_t* S, char orient, int wantT, int wantZ,
                 int maxitfac, double* const* Z, double* wr, double* wi, int* schurindex, psd_stats* stats,
                 int32_t* sweeplog, int64_t maxlog, int* info) {
    int dummy;
    if (!info) info = &dummy;
    if (stats) memset(stats, 0, sizeof(*stats));
    if (!c) return *info = -1;
    if ((*info = check_dims(n, p)) != 0) return *info;
    if (!A) return *info = -4;
    if (S)
        for (int j = 0; j < p; ++j)
            if (!S[j]) return *info = PSD_INFO_NOTIMPL;  // signed case: src/rgeneralized.jl (not in this build)
    if (orient != 'R' && orient != 'L') return *info = -6;
    if (wantZ && !Z) return *info = -10;
    const int mlog = 2 * (maxitfac > 0 ? maxitfac : 1) * n + n + 16;
    if ((*info = c->reserve(n, p, true, mlog)) != 0) return *info;
    const size_t nn = (size_t)n * n;
    Timer tc;
    tc.start(c->stream);
    for (int j = 0; j < p; ++j) PSD_CHECK(psd_rt_h2d(c->dH + j * nn, A[j], nn * 8, c->stream));
    double ms_copy = tc.stop(c->stream);
    psd_stats local;
    psd_stats* s = stats ? stats : &local;
    int rc = psd_d_pschur_dev(c, n, p, c->dH, orient, wantT, wantZ, maxitfac, wantZ ? c->dZ : nullptr, wr, wi,
                              schurindex, s, sweeplog, maxlog, info);
    if (rc < 0 || rc >= PSD_INFO_NOTIMPL) return rc;
    tc.start(c->stream);
    for (int j = 0; j < p; ++j) PSD_CHECK(psd_rt_d2h(A[j], c->dH + j * nn, nn * 8, c->stream));
    if (wantZ)
        for (int j = 0; j < p; ++j) PSD_CHECK(psd_rt_d2h(Z[j], c->dZ + j * nn, nn * 8, c->stream));
    PSD_CHECK(psd_rt_sync(c->stream));
    ms_copy += tc.stop(c->stream);
    s->ms_copy = ms_copy;
    return rc;
}

int psd_d_pschur_hess(psd_ctx* c, int n, int p, double* const* H, double* const* Q, int wantT, int wantZ, int maxitfac,
                      double* wr, double* wi, psd_stats* stats, int32_t* sweeplog, int64_t maxlog, int* info) {
    int dummy;
    if (!info) info = &dummy;
    if (stats) memset(stats, 0, sizeof(*stats));
    if (!c) return *info = -1;
    if ((*info = check_dims(n, p)) != 0) return *info;
    if (!H) return *info = -4;
    if (wantZ && !Q) return *info = -5;
    if (maxitfac < 1) return *info = -8;
    const int mlog = 2 * maxitfac * n + n + 16;
    if ((*info = c->reserve(n, p, true, mlog)) != 0) return *info;
    const size_t nn = (size_t)n * n;
    for (int j = 0; j < p; ++j) PSD_CHECK(psd_rt_h2d(c->dH + j * nn, H[j], nn * 8, c->stream));
    if (wantZ)
        for (int j = 0; j < p; ++j) PSD_CHECK(psd_rt_h2d(c->dZ + j * nn, Q[j], nn * 8, c->stream));
    psd_stats local;
    memset(&local, 0, sizeof(local));
    psd_stats* s = stats ? stats : &local;
    Timer t;
    t.start(c->stream);
    int rc = run_iteration(c, n, p, c->dH, wantZ ? c->dZ : nullptr, wantT, wantZ, maxitfac, wr, wi, s, sweeplog, maxlog,
                           info);
    s->ms_iter = s->ms_total = t.stop(c->stream);
    if (rc < 0 || rc >= PSD_INFO_NOTIMPL) return rc;
    for (int j = 0; j < p; ++j) PSD_CHECK(psd_rt_d2h(H[j], c->dH + j * nn, nn * 8, c->stream));
    if (wantZ)
        for (int j = 0; j < p; ++j) PSD_CHECK(psd_rt_d2h(Q[j], c->dZ + j * nn, nn * 8, c->stream));
    PSD_CHECK(psd_rt_sync(c->stream));
    return rc;
}

}  // extern "C"

// =================================================================================================
// complex path
namespace {

#ifndef PSD_HOSTSIM
// look-ahead form (psd_zhess2.h), the complex counterparts of hessenberg2_launches / hessenberg2_async
template <int NK, int CR>
int zhessenberg2_launches(psd_ctx* c, int n, int p, const psd_zhess2_args& ha) {
    constexpr int LG = 64 / CR;  // chain blocks per group of 8 lines (16-byte elements: 8 rows per 128-byte line)
    const int nC = ((ha.xcd && CR < 8) ? (((n + CR - 1) / CR + LG - 1) / LG) * LG : (n + CR - 1) / CR) + 1, nT = (n + PSD_ZH2_ROWS - 1) / PSD_ZH2_ROWS, nB = (n + 3) / 4;
    const size_t lds = ((size_t)n + 8 + 2 * PSD_ZH2_NT + 64) * sizeof(psd_z);
    const int gridx = nC + nT + nB;
    auto link = [&](int i, int j) {
        hipLaunchKernelGGL((psd_zhess2_link<NK, CR>), dim3(gridx), dim3(PSD_ZH2_NT), lds, c->stream, ha, n, i, j, nC, nT);
    };
    link(0, 1);
    for (int i = 1; i <= n - 1; ++i)
        for (int j = p; j >= 1; --j) link(i, j);
    link(n, p);
    link(n, p - 1);
    return 0;
}
template <int NK, int CR>
int zhessenberg2_async(psd_ctx* c, int n, int p, const psd_zhess2_args& ha, int K) {
    // Every way out (a failed runtime call in the middle of the launch sequence included) first waits for the side streams:
    // their launches read and write the caller's factors and this context's ring.
    struct SideStreams {
        psd_ctx* c;
        ~SideStreams() {
            if (c->stream3) (void)hipStreamSynchronize(c->stream3);
            if (c->stream4) (void)hipStreamSynchronize(c->stream4);
        }
    } side_streams{c};

    constexpr int LG = 64 / CR;
    const int nC = ((ha.xcd && CR < 8) ? (((n + CR - 1) / CR + LG - 1) / LG) * LG : (n + CR - 1) / CR) + 1, nT = (n + PSD_ZH2_ROWS - 1) / PSD_ZH2_ROWS, nB = (n + 3) / 4;
    const size_t lds = ((size_t)n + 8 + 2 * PSD_ZH2_NT + 64) * sizeof(psd_z);
    const int Q = (n - 1) * p;
    const int nbatch = Q / K + 1;
    if ((int)c->h2ev.size() < 16) {
        c->h2ev.resize(16, nullptr);
        for (auto& e : c->h2ev) PSD_CHECK(hipEventCreateWithFlags(&e, hipEventDisableTiming));
    }
    hipEvent_t* evA = c->h2ev.data();
    hipEvent_t* evB = c->h2ev.data() + 8;
    PSD_CHECK(hipEventRecord(evA[0], c->stream));
    PSD_CHECK(hipStreamWaitEvent(c->stream3, evA[0], 0));
    hipLaunchKernelGGL((psd_zhess2_link<NK, CR>), dim3(nC), dim3(PSD_ZH2_NT), lds, c->stream, ha, n, 0, 1, nC, 0);
    int nextb = 0;
    auto batch = [&](int b) -> int {
        PSD_CHECK(hipEventRecord(evA[b & 7], c->stream));
        PSD_CHECK(hipStreamWaitEvent(c->stream3, evA[b & 7], 0));
        hipLaunchKernelGGL((psd_zhess2_bulk<NK>), dim3(nT + nB, K), dim3(PSD_ZH2_NT), lds, c->stream3, ha, n, b * K, nT);
        PSD_CHECK(hipEventRecord(evB[b & 7], c->stream3));
        return 0;
    };
    int idx = 0;
    for (int i = 1; i <= n - 1; ++i)
        for (int j = p; j >= 1; --j, ++idx) {
            const int need = idx + 1 - p;
            if (need >= 0 && need % K == 0) PSD_CHECK(hipStreamWaitEvent(c->stream, evB[(need / K) & 7], 0));
            hipLaunchKernelGGL((psd_zhess2_link<NK, CR>), dim3(nC), dim3(PSD_ZH2_NT), lds, c->stream, ha, n, i, j, nC, 0);
            if (idx >= nextb * K + K - 1) {
                PSD_CHECK(batch(nextb));
                ++nextb;
            }
        }
    for (; nextb < nbatch; ++nextb) PSD_CHECK(batch(nextb));
    PSD_CHECK(hipStreamWaitEvent(c->stream, evB[(nbatch - 1) & 7], 0));
    return 0;
}
// pipe form for ComplexF64 (see hessenberg2_pipe)
template <int NK, int CR>
int zhessenberg2_pipe(psd_ctx* c, int n, int p, const psd_zhess2_args& ha, int K) {
    // Every way out (a failed runtime call in the middle of the launch sequence included) first waits for the side streams:
    // their launches read and write the caller's factors and this context's ring.
    struct SideStreams {
        psd_ctx* c;
        ~SideStreams() {
            if (c->stream3) (void)hipStreamSynchronize(c->stream3);
            if (c->stream4) (void)hipStreamSynchronize(c->stream4);
        }
    } side_streams{c};

    constexpr int LG = 64 / CR;
    const int nC = ((ha.xcd && CR < 8) ? (((n + CR - 1) / CR + LG - 1) / LG) * LG : (n + CR - 1) / CR) + 1, nT = (n + PSD_ZH2_ROWS - 1) / PSD_ZH2_ROWS, nB = (n + 3) / 4;
    const size_t lds = ((size_t)n + 8 + 2 * PSD_ZH2_NT + 64) * sizeof(psd_z);
    const int Q = (n - 1) * p;
    const int nbatch = Q / K + 1;
    if ((int)c->h2ev.size() < 26) {
        const size_t old = c->h2ev.size();
        c->h2ev.resize(26, nullptr);
        for (size_t q = old; q < c->h2ev.size(); ++q) PSD_CHECK(hipEventCreateWithFlags(&c->h2ev[q], hipEventDisableTiming));
    }
    hipEvent_t* evA = c->h2ev.data();
    hipEvent_t* evB = c->h2ev.data() + 8;
    hipEvent_t* evC = c->h2ev.data() + 16;
    hipEvent_t evJ = c->h2ev[24], evK = c->h2ev[25];
    hipStream_t S[2] = {c->stream, c->stream4};
    PSD_CHECK(hipEventRecord(evJ, c->stream));
    PSD_CHECK(hipStreamWaitEvent(c->stream3, evJ, 0));
    PSD_CHECK(hipStreamWaitEvent(c->stream4, evJ, 0));
    hipLaunchKernelGGL((psd_zhess2_link<NK, CR>), dim3(nC), dim3(PSD_ZH2_NT), lds, S[0], ha, n, 0, 1, nC, 0);
    int nextb = 0;
    auto batch = [&](int b) -> int {
        PSD_CHECK(hipEventRecord(evA[b & 7], S[0]));
        PSD_CHECK(hipEventRecord(evC[b & 7], S[1]));
        PSD_CHECK(hipStreamWaitEvent(c->stream3, evA[b & 7], 0));
        PSD_CHECK(hipStreamWaitEvent(c->stream3, evC[b & 7], 0));
        hipLaunchKernelGGL((psd_zhess2_bulk<NK>), dim3(nT + nB, K), dim3(PSD_ZH2_NT), lds, c->stream3, ha, n, b * K, nT);
        PSD_CHECK(hipEventRecord(evB[b & 7], c->stream3));
        return 0;
    };
    int idx = 0;
    for (int i = 1; i <= n - 1; ++i)
        for (int j = p; j >= 1; --j, ++idx) {
            hipStream_t s = S[(idx + 1) & 1];
            const int need = idx + 1 - p;
            if (need >= 0 && need % K == 0) PSD_CHECK(hipStreamWaitEvent(s, evB[(need / K) & 7], 0));
            if (need >= 1 && (need - 1) % K == 0) PSD_CHECK(hipStreamWaitEvent(s, evB[((need - 1) / K) & 7], 0));
            hipLaunchKernelGGL((psd_zhess2_link<NK, CR>), dim3(nC), dim3(PSD_ZH2_NT), lds, s, ha, n, i, j, nC, 0);
            if (idx >= nextb * K + K - 1) {
                PSD_CHECK(batch(nextb));
                ++nextb;
            }
        }
    for (; nextb < nbatch; ++nextb) PSD_CHECK(batch(nextb));
    PSD_CHECK(hipEventRecord(evK, S[1]));
    PSD_CHECK(hipStreamWaitEvent(c->stream, evK, 0));
    PSD_CHECK(hipStreamWaitEvent(c->stream, evB[(nbatch - 1) & 7], 0));
    return 0;
}
int zhessenberg2_dev(psd_ctx* c, int n, int p, psd_z* dH, psd_z* dtau) {
    const size_t ringbytes = PSD_H2_RING * psd_zh2_slot_doubles(n) * sizeof(double) + 64;  // (+ the pipe form's error word)
    if (!c->zh2ring || c->zh2ring_n < n) {
        if (c->zh2ring) psd_rt_free(c->zh2ring);
        c->zh2ring = nullptr;
        PSD_CHECK(psd_rt_malloc((void**)&c->zh2ring, ringbytes));
        c->zh2ring_n = n;
    }
    PSD_CHECK(psd_rt_memset(c->zh2ring, 0, ringbytes, c->stream));
    psd_zhess2_args ha;
    ha.H = dH;
    ha.tau = dtau;
    ha.ring = c->zh2ring;
    ha.p = p;
    ha.ringmask = 3;
    ha.xcd = c->hess_xcd;
    ha.pipe = 0;
    ha.err = (int*)(c->zh2ring + PSD_H2_RING * psd_zh2_slot_doubles(n));
    const size_t lds = ((size_t)n + 8 + 2 * PSD_ZH2_NT + 64) * sizeof(psd_z);
    if (lds > 64 * 1024) return PSD_INFO_NOTIMPL;
    int K = c->hess_async;
    if (K < 0) K = (p >= 32 && n >= 512) ? ((p >= 48 && c->hess_pipe && c->stream4 && n <= 1024) ? 24 : 16) : 0;  // (as hessenberg2_dev: 16 / 24 / 32 links per batch 633 / 604 / 621 ms at p = 64)
    if (K > 0 && p >= 9 * K) K = (p + 7) / 8;
    if (K > 0 && p >= 2 * K && p + K + 2 <= PSD_H2_RING && c->stream3) {
        ha.ringmask = PSD_H2_RING - 1;
        if (c->stream4 && n <= 1024 && psd_pipe_form(c)) {
            ha.pipe = 1;
            int rc;
            if (n <= 256) rc = zhessenberg2_pipe<4, 8>(c, n, p, ha, K);
            else if (n <= 512) rc = zhessenberg2_pipe<8, 8>(c, n, p, ha, K);
            else rc = zhessenberg2_pipe<16, 8>(c, n, p, ha, K);
            if (rc != 0) return rc;
            int herr = 0;
            PSD_CHECK(psd_rt_d2h(&herr, ha.err, sizeof(int), c->stream));
            PSD_CHECK(psd_rt_sync(c->stream));
            return herr ? PSD_INFO_RUNTIME + 0xfffb : 0;
        }
        if (n <= 256) return zhessenberg2_async<4, 8>(c, n, p, ha, K);
        if (n <= 512) return zhessenberg2_async<8, 8>(c, n, p, ha, K);
        if (n <= 1024) return zhessenberg2_async<16, 4>(c, n, p, ha, K);
        return zhessenberg2_async<32, 4>(c, n, p, ha, K);
    }
    if (n <= 256) return zhessenberg2_launches<4, 8>(c, n, p, ha);
    if (n <= 512) return zhessenberg2_launches<8, 8>(c, n, p, ha);
    if (n <= 1024) return zhessenberg2_launches<16, 4>(c, n, p, ha);
    return zhessenberg2_launches<32, 4>(c, n, p, ha);
}
#endif

int zhessenberg_dev(psd_ctx* c, int n, int p, psd_z* dH, psd_z* dtau) {
    const size_t nn = (size_t)n * n;
    PSD_CHECK(psd_rt_memset(dtau, 0, sizeof(psd_z) * (size_t)n * p, c->stream));
    if (n < 2) return 0;
#ifndef PSD_HOSTSIM
    if (p >= 3 && n <= 2048 && c->hess_lookahead) return zhessenberg2_dev(c, n, p, dH, dtau);
#endif
    const size_t lds_refl = PSD_HESS_NT * 8;
    const size_t lds_apply = (PSD_HESS_NT + (size_t)n + 8) * sizeof(psd_z);
    for (int i = 1; i <= n - 1; ++i) {
        for (int j = p; j >= 1; --j) {
            const int r0 = (j == 1) ? (i + 1) : i;
            if (n - r0 + 1 < 1) continue;
            psd_z* Aj = dH + (size_t)(j - 1) * nn;
            psd_z* Ajm1 = dH + (size_t)((j == 1 ? p : j - 1) - 1) * nn;
            PSD_LAUNCH(psd_zhess_refl, psd_dim3(1), PSD_HESS_NT, lds_refl, c->stream, Aj, n, r0, i, c->zvbuf,
                       dtau + (size_t)(j - 1) * n + (i - 1));
            const int lc0 = i + 1;
            const int nL = (n - lc0 + 1 + 3) / 4;
            const int nR = (n + PSD_HESS_RS - 1) / PSD_HESS_RS;
            if (Aj != Ajm1) {
                PSD_LAUNCH(psd_zhess_apply, psd_dim3(nL + nR), PSD_HESS_NT, lds_apply, c->stream, Aj, Ajm1, n, r0, lc0,
                           (const psd_z*)c->zvbuf, nL);
            } else {
                PSD_LAUNCH(psd_zhess_apply, psd_dim3(nL), PSD_HESS_NT, lds_apply, c->stream, Aj, (psd_z*)nullptr, n, r0,
                           lc0, (const psd_z*)c->zvbuf, nL);
                PSD_LAUNCH(psd_zhess_apply, psd_dim3(nR), PSD_HESS_NT, lds_apply, c->stream, (psd_z*)nullptr, Ajm1, n, r0,
                           lc0, (const psd_z*)c->zvbuf, 0);
            }
        }
    }
    return 0;
}

int zformq_dev(psd_ctx* c, int n, int p, const psd_z* dH, const psd_z* dtau, psd_z* dQ) {
    // a period-sharded context forms only the Q_j of its slice (psd_set_shard)
    int jlo = 0, jhi = p;
    c->slice(p, jlo, jhi);
    if (jhi <= jlo) return 0;
    PSD_LAUNCH(psd_zset_identity, psd_dim3(n, jhi - jlo), 64, 0, c->stream, dQ + (size_t)jlo * n * n, n);
    const size_t lds = PSD_HESS_NT * sizeof(psd_z);
#ifndef PSD_HOSTSIM
    if (n >= 64 && n <= 1024 && c->formq_blocked) {
        // B reflectors per pass over the Q_j (psd_zformq_blk); PSD_FORMQ_BLOCKED=0: one launch per reflector
        const int B = 16;
        for (int i = n - 1; i >= 1; i -= B) {
            const int ilow = (i - B + 1 >= 1) ? (i - B + 1) : 1;
            const int tiles = (n - ilow + 1 + 3) / 4;
            if (n <= 512) hipLaunchKernelGGL((psd_zformq_blk<8>), dim3(tiles, jhi - jlo), dim3(256), 0, c->stream, dH, dtau, dQ, n, i, B, jlo);
            else hipLaunchKernelGGL((psd_zformq_blk<16>), dim3(tiles, jhi - jlo), dim3(256), 0, c->stream, dH, dtau, dQ, n, i, B, jlo);
        }
        return 0;
    }
#endif
    for (int i = n - 1; i >= 1; --i) {
        const int tiles = (n - i + 1 + 3) / 4;
        PSD_LAUNCH(psd_zformq_step, psd_dim3(tiles, jhi - jlo), PSD_HESS_NT, lds, c->stream, dH, dtau, dQ, n, i, jlo);
    }
    return 0;
}

// generalized.jl:166-931 on device (all signatures true)
int ziterate_dev(psd_ctx* c, int n, int p, psd_z* dH, psd_z* dZ, int wantT, int wantZ, int maxitfac, psd_zstate* st_out,
                 psd_stats* stats, int maxlog) {
    const int W = choose_window(p, 16);
    if (W == 0) return PSD_INFO_NOTIMPL;
#ifndef PSD_HOSTSIM
    struct BulkStream {  // (as iterate_dev: no way out leaves a launch of the second stream behind)
        psd_ctx* c;
        ~BulkStream() {
            if (c->stream2) (void)hipStreamSynchronize(c->stream2);
        }
    } bulk_stream{c};
#endif
    psd_zparams P;
    P.H = dH;
    {
        int zl = 0, zh = p;
        c->slice(p, zl, zh);
        P.zlo = zl + 1;
        P.zhi = zh;
    }
    P.Z = wantZ ? dZ : nullptr;
    P.st = c->zst;
    P.desc = c->zdesc;
    P.tr = c->ztr;
    P.cnt = c->zcnt;
    P.dG = c->zdG;
    P.alpha = c->zalpha;
    P.beta = c->zbeta;
    P.ascale = c->zascale;
    P.log = c->zlog;
    const size_t lds_step = step_lds_bytes(p, W, 16);
    // scan chase (psd_zchase3.h, the default): PSD_ZC3_WAVES wavefronts per chase workgroup, the command block in the
    // reduction scratch behind the window image (idle while a window is chased), the rotation table behind the scratch
    const bool zscan3 = c->chase3 && p >= PSD_ZC3_MINP && p <= PSD_ZC3_MAXP;
    P.zcdefer = 0;
    P.zcoff = zscan3 ? (int)((size_t)p * W * (W + 1) * sizeof(psd_z)) : 0;
    P.zc3off = zscan3 ? (int)step_lds_scratch_end(p, W, 16, false) : 0;
    const int zwaves = zscan3 ? PSD_ZC3_WAVES : 1;
    (void)zwaves;  // (the serial simulation launches one simulated wavefront)
    P.zslG = 1;
    P.zslmem = nullptr;
    P.zslerr = nullptr;
#ifndef PSD_HOSTSIM
    if (lds_step > c->zstep_lds_set) {
        PSD_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(psd_zq_step),
                                      hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_step));
        c->zstep_lds_set = lds_step;
    }
#endif
    // multishift trains (see iterate_dev): cursor 0 = slot 0 of the cursor arrays
    const int M = (c->ztrain_m >= 2) ? ((c->ztrain_m > PSD_TRAIN_MAX) ? PSD_TRAIN_MAX : c->ztrain_m) : 1;
    P.cst = nullptr;
    P.cep = nullptr;
    P.tshift = nullptr;
    P.tick = 0;
    if (M > 1) {
        PSD_CHECK(c->ztreserve(p));
        PSD_CHECK(psd_rt_memset(c->ztcst, 0, sizeof(psd_zstate) * PSD_TRAIN_MAX + sizeof(int) * (PSD_TRAIN_MAX + 8), c->stream));
        PSD_CHECK(psd_rt_memset(c->ztdesc, 0, 2 * sizeof(psd_zapply_desc) * PSD_TRAIN_MAX, c->stream));
        P.cst = c->ztcst;
        P.cep = (int*)(c->ztcst + PSD_TRAIN_MAX);
        P.tshift = c->ztshift;
        P.desc = c->ztdesc;
        P.cnt = c->ztcnt;
        P.tr = c->zttr;
#ifndef PSD_HOSTSIM
        PSD_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(psd_zq_step_train),
                                      hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_step));
#endif
    }
    // factor-sliced sweep windows (psd_set_slices; psd_zslice3.h): G workgroups per cursor slot; needs the scan chase,
    // the train kernel and at least two factors per slice
    int zslG = 1;
#ifndef PSD_HOSTSIM
    if (c->slices > 1 && M > 1 && zscan3 && p / c->slices >= 2 && c->slices <= PSD_SL_MAXG) {
        zslG = c->slices;
        while (zslG > 1 && c->ncu > 0 && M * zslG > c->ncu) zslG /= 2;  // (as iterate_dev: the M x G workgroups must be resident together)
        const size_t slbytes = (size_t)PSD_SLOTS * (PSD_SL_CMD_BYTES + PSD_SL_MAXG * PSD_SL_BOX_BYTES) + 64;
        if (!c->slmem) PSD_CHECK(psd_rt_malloc((void**)&c->slmem, slbytes));
        PSD_CHECK(psd_rt_memset(c->slmem, 0, slbytes, c->stream));
        P.zslG = zslG;
        P.zslmem = c->slmem;
        P.zslerr = (int*)(c->slmem + slbytes - 64);
    }
#endif
    int train_oc = 200;  // (100 until the cursors moved to W positions apart; 200-400 measured alike)
    if (const char* e = psd_env_diag("PSD_TRAIN_OC")) train_oc = atoi(e);  // (tuning hook)
    PSD_LAUNCH(psd_zq_init, psd_dim3(1), 256, 0, c->stream, P, n, p, wantT, wantZ, W, maxitfac, maxlog, M, train_oc);
#ifndef PSD_HOSTSIM
    const bool zdef = M > 1 && wantZ && c->stream2 && c->evE[0] && (c->overlap == 2 || (c->overlap == 3 && n >= 1024));
#else
    const bool zdef = M > 1 && wantZ && c->overlap == 2;
#endif
    // far rows of the column roles on the second stream as well (psd_zparams::zcdefer): OFF by default — measured at
    // configs[2] (n = 1024, p = 64): iteration 2.02 s without, 2.10-2.12 s with it (the tick is bound by the bytes of its
    // bulk updates; the extra launch and the cross-stream events cost more than the overlap gives).  PSD_ZCDEFER=1 turns
    // it on (bit-identical results: tests/test_hostsim_complex.py::test_zcolumn_roles_deferred).
    const bool zcdef = zdef && c->zcdefer != 0 && c->apply_worklist != 0;
    bool far_pending = false;
    P.zcdefer = zcdef ? 1 : 0;
    // (the tile holds the rows a window's lists span: at most W + 1, not the 32 the kernel could serve — at p = 64 (W = 12) that is
    //  13.5 KiB instead of 33 KiB per single-wave workgroup, twelve of them per CU instead of four)
    const size_t lds_apply = PSD_ZTR_LDS_BYTES + (size_t)((W + 2 < 32) ? (W + 2) : 32) * (PSD_ZAPPLY_NT + 1) * sizeof(psd_z);
    const size_t lds_wl = lds_apply + sizeof(int) * 2 * (PSD_TRAIN_MAX + 2);  // (+ the work list's item table)
#ifndef PSD_HOSTSIM
    int wl_grid = 2048;  // (single-wave workgroups; 1024 .. 2560 measured alike, 3072 and more slower)
    if (const char* e = psd_env_diag("PSD_ZWL_GRID")) wl_grid = atoi(e);
#else
    const int wl_grid = 6;
#endif
    const int tiles = (n + PSD_ZAPPLY_NT - 1) / PSD_ZAPPLY_NT;
    // one role of the tick's bulk updates (pass: 1 columns, 2 rows, 3 Schur vectors): work list, or the grid-per-cursor form (PSD_APPLY_WL=0)
    auto zapply = [&](psd_stream_t st_, const psd_zparams& Pz, int pass) -> int {
        if (c->apply_worklist)
            PSD_LAUNCH(psd_zq_apply_wl, psd_dim3(wl_grid), PSD_ZAPPLY_NT, lds_wl, st_, Pz, n, p, p + 8, pass, M, (int)lds_apply);
        else
            PSD_LAUNCH(psd_zq_apply_train, psd_dim3(tiles, p, M), PSD_ZAPPLY_NT, lds_apply, st_, Pz, n, p, p + 8, pass);
        return 0;
    };
    const int dtiles = (n + 255) / 256;
    const int batch = 32;
    psd_zstate hst;
    memset(&hst, 0, sizeof(hst));
    long long launched = 0;
    const int nbmin = (W - 3 > 0) ? ((W - 3 < 8) ? (W - 3) : 8) : 1;  // (trains run windows down to 8 positions)
    const long long cap = (long long)maxitfac * n * ((long long)n / nbmin + 4) + 4LL * n + 1024;
    double sample_ms = 0.0;
    int samples = 0;
#ifndef PSD_HOSTSIM
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
    std::vector<std::pair<hipEvent_t, hipEvent_t>> pend;
    psd_poller poller(c, sample_ms, samples);
#endif
    for (;;) {
        for (int b = 0; b < batch; ++b) {
#ifndef PSD_HOSTSIM
            const bool sample = c->profile && ((launched & 15) == 0);
            if (sample) {
                (void)hipEventCreate(&ev0);
                (void)hipEventCreate(&ev1);
                (void)hipEventRecord(ev0, c->stream);
            }
#endif
            P.tick = (int)launched;
            const int par = zdef ? (int)(launched & 1) : 0;
            psd_zparams Pq = P, Pprev = P;
            if (zdef) {
#ifndef PSD_HOSTSIM
                if (launched >= 2) PSD_CHECK(hipStreamWaitEvent(c->stream, c->evF[par], 0));  // (the Z launch that read this parity's lists)
#endif
                Pq.desc = P.desc + (size_t)par * PSD_TRAIN_MAX;
                Pq.cnt = P.cnt + (size_t)par * PSD_TRAIN_MAX * (p + 8);
                Pq.tr = P.tr + (size_t)par * PSD_TRAIN_MAX * p * PSD_ZTR_CAP;
                Pprev.desc = P.desc + (size_t)(par ^ 1) * PSD_TRAIN_MAX;
                Pprev.cnt = P.cnt + (size_t)(par ^ 1) * PSD_TRAIN_MAX * (p + 8);
                Pprev.tr = P.tr + (size_t)(par ^ 1) * PSD_TRAIN_MAX * p * PSD_ZTR_CAP;
            }
            (void)Pprev;
            if (M == 1)
                PSD_LAUNCH2(psd_zq_step, psd_dim3(1), PSD_STEP_NT, zwaves, lds_step, c->stream, P);
            else
                PSD_LAUNCH2(psd_zq_step_train, psd_dim3(M, zslG), PSD_STEP_NT, zwaves, lds_step, c->stream, Pq, p, p + 8);
#ifndef PSD_HOSTSIM
            if (sample) {
                (void)hipEventRecord(ev1, c->stream);
                pend.emplace_back(ev0, ev1);
            }
#endif
            if (M == 1) {
                PSD_LAUNCH(psd_zq_apply, psd_dim3(tiles, p, 3), PSD_ZAPPLY_NT, lds_apply, c->stream, P, n, p);
            } else if (zdef) {
                // Schur-vector updates one stream over (as iterate_dev: nothing reads Z_m before the iteration ends and only
                // owner m's lists touch it): they start when the tick's H updates are done and run beside the next tick's
                // chases; the lists of a tick are double-buffered by parity, the chase that reuses a parity awaits evF.
                // With zcdef the far rows of the sweep windows' column roles (psd_zapply_desc::rcut) go the same way, in
                // front of the Z launch: the next tick's rows roles cross them, so those wait for them (evG); the last
                // window of a sweep keeps its column role whole, so that the check in the next launch finds nothing under
                // way.  The serial simulation runs them at the latest point the streams
                // allow, behind the next tick's chases.
                if (zcdef && far_pending) {
#ifndef PSD_HOSTSIM
                    PSD_CHECK(hipStreamWaitEvent(c->stream, c->evG[par ^ 1], 0));
#else
                    PSD_CHECK(zapply(c->stream, Pprev, 5));
#endif
                    far_pending = false;
                }
                PSD_CHECK(zapply(c->stream, Pq, 2));
                PSD_CHECK(zapply(c->stream, Pq, zcdef ? 4 : 1));
#ifndef PSD_HOSTSIM
                PSD_CHECK(hipEventRecord(c->evE[par], c->stream));
                PSD_CHECK(hipStreamWaitEvent(c->stream2, c->evE[par], 0));
                if (zcdef) {
                    PSD_CHECK(zapply(c->stream2, Pq, 5));
                    PSD_CHECK(hipEventRecord(c->evG[par], c->stream2));
                    far_pending = true;
                }
                PSD_CHECK(zapply(c->stream2, Pq, 3));
                PSD_CHECK(hipEventRecord(c->evF[par], c->stream2));
#else
                if (zcdef) far_pending = true;
                PSD_CHECK(zapply(c->stream, Pq, 3));
#endif
            } else if (c->apply_worklist) {
                PSD_CHECK(zapply(c->stream, P, 2));
                if (wantZ) PSD_CHECK(zapply(c->stream, P, 3));
                PSD_CHECK(zapply(c->stream, P, 1));
            } else {
                PSD_LAUNCH(psd_zq_apply_train, psd_dim3(tiles, p, 2 * M), PSD_ZAPPLY_NT, lds_apply, c->stream, P, n, p, p + 8, 0);
                PSD_LAUNCH(psd_zq_apply_train, psd_dim3(tiles, p, M), PSD_ZAPPLY_NT, lds_apply, c->stream, P, n, p, p + 8, 1);
            }
            PSD_LAUNCH(psd_zq_defer, psd_dim3(dtiles), 256, 0, c->stream, Pq, n);
            ++launched;
        }
#ifdef PSD_HOSTSIM
        PSD_CHECK(psd_rt_d2h(&hst, c->zst, sizeof(hst), c->stream));
        PSD_CHECK(psd_rt_sync(c->stream));
#else
        PSD_CHECK(poller.poll(&hst, c->zst, sizeof(hst), pend));
#endif
        if (hst.phase == PSD_ZPH_DONE) break;
        if (launched > cap) {
#ifndef PSD_HOSTSIM
            if (zdef) (void)hipStreamSynchronize(c->stream2);
#endif
            *st_out = hst;
            return PSD_INFO_RUNTIME + 0xfffe;
        }
    }
#ifdef PSD_HOSTSIM
    if (zcdef && far_pending) {  // (the last tick's far column roles: its parity is the one the last launch used)
        psd_zparams Pl = P;
        const int parl = (int)((launched - 1) & 1);
        Pl.desc = P.desc + (size_t)parl * PSD_TRAIN_MAX;
        Pl.cnt = P.cnt + (size_t)parl * PSD_TRAIN_MAX * (p + 8);
        Pl.tr = P.tr + (size_t)parl * PSD_TRAIN_MAX * p * PSD_ZTR_CAP;
        PSD_CHECK(zapply(c->stream, Pl, 5));
        far_pending = false;
    }
#endif
#ifndef PSD_HOSTSIM
    if (zdef) PSD_CHECK(hipStreamSynchronize(c->stream2));  // (the phase pass below scales columns of Z)
    PSD_CHECK(poller.finish(pend));
    if (zslG > 1) {  // (a sliced window whose hand-over never arrived: the results are void, say so)
        int serr = 0;
        PSD_CHECK(psd_rt_d2h(&serr, P.zslerr, sizeof(int), c->stream));
        PSD_CHECK(psd_rt_sync(c->stream));
        if (serr) {
            *st_out = hst;
            return PSD_INFO_RUNTIME + 0xfffa;
        }
    }
#endif
    if (hst.info == 0 && wantT) {  // generalized.jl:860-908
        for (int l = p; l >= 2; --l)
            PSD_LAUNCH(psd_zq_phase, psd_dim3(n), 64, 64, c->stream, P, n, l, wantZ);
    }
    PSD_CHECK(psd_rt_last_error());
    *st_out = hst;
    if (stats) {
        stats->nlaunch_step = (int32_t)launched;
        stats->window = W;
        stats->step_kernel_ms_avg = samples ? sample_ms / samples : 0.0;
        stats->step_kernel_samples = samples;
    }
    return 0;
}

int zrun_iteration(psd_ctx* c, int n, int p, psd_z* dH, psd_z* dZ, int wantT, int wantZ, int maxitfac, double* alpha,
                   double* beta, int32_t* ascale, psd_stats* stats, int32_t* sweeplog, int64_t maxlog_user, int* info) {
    const int maxlog = 2 * maxitfac * n + n + 16;
    psd_zstate st;
    int rc = ziterate_dev(c, n, p, dH, dZ, wantT, wantZ, maxitfac, &st, stats, maxlog);
    if (rc != 0) {
        *info = rc;
        return rc;
    }
    if (stats) {
        stats->niter = st.jiter;
        stats->nsweeps = st.nsweeps;
        stats->nrqpass = st.nzshift;
        stats->ndefl1 = st.nsplit;
        stats->ndefl2 = st.ncase2;
        stats->nwindows = st.nwindows;
        stats->nlog = st.nlog;
        for (int q = 0; q < 6; ++q) stats->step_cycles[q] = st.cyc[q];
    }
    PSD_CHECK(psd_rt_d2h(alpha, c->zalpha, sizeof(psd_z) * n, c->stream));
    PSD_CHECK(psd_rt_d2h(beta, c->zbeta, sizeof(double) * n, c->stream));
    std::vector<int> hsc(n, 0);
    PSD_CHECK(psd_rt_d2h(hsc.data(), c->zascale, sizeof(int) * n, c->stream));
    const int nl = st.nlog < maxlog ? st.nlog : maxlog;
    std::vector<int> hlog((size_t)3 * nl + 3, 0);
    if (nl > 0) PSD_CHECK(psd_rt_d2h(hlog.data(), c->zlog, sizeof(int) * 3 * (size_t)nl, c->stream));
    PSD_CHECK(psd_rt_sync(c->stream));
    for (int q = 0; q < n; ++q) ascale[q] = hsc[q];
    if (stats) {  // algorithmic bytes of the sweeps, E = 16
        double b = 0.0;
        for (int q = 0; q < nl; ++q)
            if (hlog[3 * q] == 0 || hlog[3 * q] == 4) {
                const double w = hlog[3 * q + 2] - hlog[3 * q + 1] + 1;
                if (!wantT) b += 2 * 16.0 * p * w * w + (wantZ ? 2 * 16.0 * p * w * n : 0.0);
                else b += 2 * 16.0 * p * w * (wantZ ? (2.0 * n + 1) : (n + 1.0));
            }
        stats->bytes_sweeps = b;
    }
    if (sweeplog) {
        const int64_t m = nl < maxlog_user ? nl : maxlog_user;
        for (int64_t q = 0; q < 3 * m; ++q) sweeplog[q] = hlog[q];
    }
    *info = (st.info == PSD_LIST_OVERFLOW) ? (PSD_INFO_RUNTIME + 77) : ((st.info != 0) ? (PSD_INFO_NOCONV + st.info) : 0);
    return *info;
}

}  // namespace

namespace {

int zgiterate_dev(psd_ctx* c, int n, int p, psd_z* dH, psd_z* dZ, const uint8_t* S, int wantT, int wantZ, int maxitfac,
                  psd_zgstate* st_out, psd_stats* stats, int maxlog, int hessmode) {
    const int W = choose_window(p, 16);
    if (W == 0) return PSD_INFO_NOTIMPL;
    std::vector<unsigned char> hS(p, 1);
    for (int l = 0; l < p; ++l) hS[l] = (!S || S[l]) ? 1 : 0;
    PSD_CHECK(psd_rt_h2d(c->zgS, hS.data(), (size_t)p, c->stream));
    PSD_CHECK(psd_rt_sync(c->stream));
    psd_zgparams P;
    P.H = dH;
    {
        int zl = 0, zh = p;
        c->slice(p, zl, zh);
        P.zlo = zl + 1;
        P.zhi = zh;
    }
    P.Z = wantZ ? dZ : nullptr;
    P.S = c->zgS;
    P.st = c->zgst;
    P.desc = c->zgdesc;
    P.tr = c->zgtr;
    P.cnt = c->zgcnt;
    P.dG = c->zgdG;
    P.alpha = c->zalpha;
    P.beta = c->zbeta;
    P.ascale = c->zascale;
    P.log = c->zlog;
    const size_t lds_step = step_lds_bytes(p, W, 16);
#ifndef PSD_HOSTSIM
    if (lds_step > c->zgstep_lds_set) {
        PSD_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(psd_zgq_step),
                                      hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_step));
        c->zgstep_lds_set = lds_step;
    }
#endif
    // stage 2 of the signed Hessenberg reduction: pipeline over the factors (see giterate_dev)
    const size_t lds_hess = psd_zghess_lds_bytes(p, W);
    const int hess_links = psd_ghess_links(p), hess_waves = psd_ghess_waves(p);
    const char* hserial = psd_env("PSD_HESS_SERIAL");
    const bool hess_pipe = hessmode && lds_hess <= (size_t)160 * 1024 && !(hserial && hserial[0] == '1');
    // (scan form of the stage-2 kernel, the default; PSD_HESS_SCAN=0: the pipeline of beats over the factors of round 2)
    int hess_scan = 1;
    if (const char* e = psd_env("PSD_HESS_SCAN")) hess_scan = atoi(e);
    (void)hess_scan;
#ifndef PSD_HOSTSIM
    if (hess_pipe && lds_hess > c->zghess_lds_set) {
        PSD_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(psd_zgq_hess_step),
                                      hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_hess));
        c->zghess_lds_set = lds_hess;
    }
#endif
    // multishift trains (as giterate_dev)
    const int tw = hessmode ? 0 : c->gtrain_m;
    const int Mcap = (PSD_ZHQR_MAX < PSD_TRAIN_MAX) ? PSD_ZHQR_MAX : PSD_TRAIN_MAX;
    const int M = (tw >= 2) ? ((tw > Mcap) ? Mcap : tw) : 1;
    P.cst = nullptr;
    P.cep = nullptr;
    P.tshift = nullptr;
    P.tick = 0;
    if (M > 1 || tw == -2) {
        PSD_CHECK(c->zgtreserve(p));
        PSD_CHECK(psd_rt_memset(c->zgtcst, 0, sizeof(psd_zgstate) * PSD_TRAIN_MAX + sizeof(int) * (PSD_TRAIN_MAX + 8), c->stream));
        PSD_CHECK(psd_rt_memset(c->zgtdesc, 0, sizeof(psd_gapply_desc) * PSD_TRAIN_MAX, c->stream));
        P.cst = c->zgtcst;
        P.cep = (int*)(c->zgtcst + PSD_TRAIN_MAX);
        P.tshift = c->zgtshift;
        if (M > 1) {
            P.desc = c->zgtdesc;
            P.cnt = c->zgtcnt;
            P.tr = c->zgttr;
#ifndef PSD_HOSTSIM
            PSD_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(psd_zgq_step_train),
                                          hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_step));
#endif
        }
    }
    int train_oc = 50;
    if (const char* e = psd_env_diag("PSD_TRAIN_OC")) train_oc = atoi(e);  // (tuning hook)
    PSD_LAUNCH(psd_zgq_init, psd_dim3(1), 256, 0, c->stream, P, n, p, wantT, wantZ, W, maxitfac, maxlog, hessmode,
               (tw == -2) ? -2 : M, train_oc);
    const size_t lds_apply = sizeof(psd_ztr) * PSD_GTR_CAP + (size_t)32 * (PSD_ZAPPLY_NT + 1) * sizeof(psd_z);
    const int tiles = (n + PSD_ZAPPLY_NT - 1) / PSD_ZAPPLY_NT;
    const int dtiles = (n + 255) / 256;
    const int batch = 32;
    psd_zgstate hst;
    memset(&hst, 0, sizeof(hst));
    long long launched = 0;
    const int nbmin = (W - 3 > 0) ? ((W - 3 < 8) ? (W - 3) : 8) : 1;  // (trains run windows down to 8 positions)
    const long long cap = (long long)maxitfac * n * ((long long)n / nbmin + 8) + 8LL * n + 1024 + (hessmode ? (long long)n * n : 0);
#ifndef PSD_HOSTSIM
    double sample_ms = 0.0;
    int samples = 0;
    std::vector<std::pair<hipEvent_t, hipEvent_t>> pend;
    psd_poller poller(c, sample_ms, samples);
#endif
    for (;;) {
        for (int b = 0; b < batch; ++b) {
            P.tick = (int)launched;
            if (hess_pipe)
                PSD_LAUNCH(psd_zgq_hess_step, psd_dim3(1), 64 * hess_waves, lds_hess, c->stream, P, hess_links);
            else if (M > 1)
                PSD_LAUNCH(psd_zgq_step_train, psd_dim3(M), PSD_STEP_NT, lds_step, c->stream, P, p, p + 8);
            else
                PSD_LAUNCH(psd_zgq_step, psd_dim3(1), PSD_STEP_NT, lds_step, c->stream, P);
            if (M > 1) {
                PSD_LAUNCH(psd_zgq_apply_train, psd_dim3(tiles, p, 2 * M), PSD_ZAPPLY_NT, lds_apply, c->stream, P, n, p, p + 8, 0);
                PSD_LAUNCH(psd_zgq_apply_train, psd_dim3(tiles, p, M), PSD_ZAPPLY_NT, lds_apply, c->stream, P, n, p, p + 8, 1);
            } else {
                PSD_LAUNCH(psd_zgq_apply, psd_dim3(tiles, p, 3), PSD_ZAPPLY_NT, lds_apply, c->stream, P, n, p);
            }
            if (!hess_pipe) PSD_LAUNCH(psd_zgq_defer, psd_dim3(dtiles), 256, 0, c->stream, P, n);
            ++launched;
        }
#ifdef PSD_HOSTSIM
        PSD_CHECK(psd_rt_d2h(&hst, c->zgst, sizeof(hst), c->stream));
        PSD_CHECK(psd_rt_sync(c->stream));
#else
        PSD_CHECK(poller.poll(&hst, c->zgst, sizeof(hst), pend));
#endif
        if (hst.phase == PSD_GPH_DONE) break;
        if (launched > cap) {
            *st_out = hst;
            return PSD_INFO_RUNTIME + 0xfffc;
        }
    }
#ifndef PSD_HOSTSIM
    PSD_CHECK(poller.finish(pend));
#endif
    if (!hessmode && hst.info == 0 && wantT) {  // generalized.jl:860-908
        for (int l = p; l >= 2; --l) PSD_LAUNCH(psd_zgq_phase, psd_dim3(n), 64, 64, c->stream, P, n, l, wantZ);
    }
    PSD_CHECK(psd_rt_last_error());
    *st_out = hst;
    if (stats) {
        stats->nlaunch_step = (int32_t)launched;
        stats->window = W;
    }
    return 0;
}

int zgrun_iteration(psd_ctx* c, int n, int p, psd_z* dH, psd_z* dZ, const uint8_t* S, int wantT, int wantZ,
                    int maxitfac, double* alpha, double* beta, int32_t* ascale, psd_stats* stats, int* info) {
    const int maxlog = 2 * maxitfac * n + n + 16;
    psd_zgstate st;
    int rc = zgiterate_dev(c, n, p, dH, dZ, S, wantT, wantZ, maxitfac, &st, stats, maxlog, 0);
    if (rc != 0) return *info = rc;
    if (stats) {
        stats->niter = st.jiter;
        stats->nsweeps = st.nsweeps;
        stats->nrqpass = st.nzshift;
        stats->ndefl1 = st.nsplit;
        stats->ndefl2 = st.ncase2;
        stats->reserved = st.ncase2 + 1000 * st.ncase3;
        stats->maxits = st.ntrainsweeps;  // (signed path: sweeps that ran inside multishift trains)
        stats->nwindows = st.nwindows;
        stats->nlog = st.nlog;
    }
    PSD_CHECK(psd_rt_d2h(alpha, c->zalpha, sizeof(psd_z) * n, c->stream));
    PSD_CHECK(psd_rt_d2h(beta, c->zbeta, sizeof(double) * n, c->stream));
    std::vector<int> hsc(n, 0);
    PSD_CHECK(psd_rt_d2h(hsc.data(), c->zascale, sizeof(int) * n, c->stream));
    PSD_CHECK(psd_rt_sync(c->stream));
    for (int q = 0; q < n; ++q) ascale[q] = hsc[q];
    return *info = (st.info == PSD_LIST_OVERFLOW) ? (PSD_INFO_RUNTIME + 77) : ((st.info != 0) ? (PSD_INFO_NOCONV + st.info) : 0);
}

// _phessenberg!(A, S; wantQ) for ComplexF64 on device — generalized.jl:988-1082
int zsghess_dev(psd_ctx* c, int n, int p, psd_z* dA, psd_z* dQ, const uint8_t* S, psd_stats* stats) {
    const size_t nn = (size_t)n * n;
    const size_t lds_refl = PSD_HESS_NT * 8;
    const size_t lds_apply = (PSD_HESS_NT + (size_t)n + 8) * sizeof(psd_z);
    auto sg = [&](int l) { return !S || S[l - 1]; };
    if (dQ) PSD_LAUNCH(psd_zset_identity, psd_dim3(n, p), 64, 0, c->stream, dQ, n);
    Timer t;
    t.start(c->stream);
    for (int l = p; l >= 2; --l) {
        psd_z* Al = dA + (size_t)(l - 1) * nn;
        psd_z* Am = dA + (size_t)(l - 2) * nn;
        psd_z* Ql = dQ ? dQ + (size_t)(l - 1) * nn : nullptr;
        const bool rq = !sg(l);
        const int mrows = sg(l - 1) ? 0 : 1;
        if (rq) {
            PSD_LAUNCH(psd_zantitranspose, psd_dim3(n), 256, 0, c->stream, Al, n);
            PSD_LAUNCH(psd_zflip, psd_dim3(n), 256, 0, c->stream, Am, n, mrows);
            if (Ql) PSD_LAUNCH(psd_zflip, psd_dim3(n), 256, 0, c->stream, Ql, n, 0);
        }
        for (int i = 1; i <= n - 1; ++i) {
            psd_z* vcur = c->zvbuf + (size_t)(i & 1) * (n + 8);
            psd_z* vnext = (i < n - 1) ? (c->zvbuf + (size_t)((i + 1) & 1) * (n + 8)) : nullptr;
            if (i == 1) PSD_LAUNCH(psd_zhess_refl, psd_dim3(1), PSD_HESS_NT, lds_refl, c->stream, Al, n, i, i, vcur, (psd_z*)nullptr);
            const int nL = (n - i + 3) / 4;
            const int nR = (n + PSD_HESS_RS - 1) / PSD_HESS_RS;
            // one launch: A_l (+ Q_l) and the neighbour A_{l-1} (see sghess_dev)
            const int g1 = nL + (Ql ? nR : 0);
            const int nLm = (n + 3) / 4;
            if (mrows == 0) {
                PSD_LAUNCH(psd_zhess_apply2, psd_dim3(g1 + nR), PSD_HESS_NT, lds_apply, c->stream, Al, Ql, i + 1, nL, g1,
                           (psd_z*)nullptr, Am, 1, 0, n, i, (const psd_z*)vcur, vnext);
            } else {
                PSD_LAUNCH(psd_zhess_apply2, psd_dim3(g1 + nLm), PSD_HESS_NT, lds_apply, c->stream, Al, Ql, i + 1, nL, g1, Am,
                           (psd_z*)nullptr, 1, nLm, n, i, (const psd_z*)vcur, vnext);
            }
        }
        PSD_LAUNCH(psd_ztril_zero, psd_dim3(n), 256, 0, c->stream, Al, n);
        if (rq) {
            PSD_LAUNCH(psd_zantitranspose, psd_dim3(n), 256, 0, c->stream, Al, n);
            PSD_LAUNCH(psd_zflip, psd_dim3(n), 256, 0, c->stream, Am, n, mrows);
            if (Ql) PSD_LAUNCH(psd_zflip, psd_dim3(n), 256, 0, c->stream, Ql, n, 0);
        }
    }
    const double ms1 = t.stop(c->stream);
    t.start(c->stream);
    psd_zgstate st;
    int rc = zgiterate_dev(c, n, p, dA, dQ, S, 1, dQ ? 1 : 0, 1, &st, nullptr, 16, 1);
    const double ms2 = t.stop(c->stream);
    if (stats) {
        stats->ms_hess = ms1 + ms2;
        stats->ms_formq = ms1;
        stats->bytes_hess = 2.0 * 16.0 * p * (double)n * n * n;
    }
    return rc;
}

// pschur!(A, S, lr) for ComplexF64 with a signed S — generalized.jl:108-148
int zsigned_pschur_host(psd_ctx* c, int n, int p, double* const* A, const uint8_t* S, char orient, int wantT, int wantZ,
                        int maxitfac, double* const* Z, double* alpha, double* beta, int32_t* ascale, int* schurindex,
                        psd_stats* stats, int* info) {
    if (maxitfac < 1) return *info = -9;
    const bool left = orient == 'L';
    auto slotA = [&](int j) { return left ? (p + 1 - j) : j; };
    auto slotZ = [&](int j) { return (!left || j == 1) ? j : (p + 2 - j); };  // generalized.jl:910-927
    std::vector<uint8_t> Sarg(p, 1);
    for (int j = 1; j <= p; ++j) Sarg[j - 1] = S[slotA(j) - 1] ? 1 : 0;
    if (!Sarg[0]) return *info = -5;  // generalized.jl:140
    const int mlog = 2 * maxitfac * n + n + 16;
    if ((*info = c->zreserve(n, p, true, mlog)) != 0) return *info;
    if ((*info = c->zgreserve(n, p)) != 0) return *info;
    const size_t nn = (size_t)n * n;
    for (int j = 1; j <= p; ++j) PSD_CHECK(psd_rt_h2d(c->zH + (size_t)(j - 1) * nn, A[slotA(j) - 1], nn * 16, c->stream));
    psd_stats local;
    memset(&local, 0, sizeof(local));
    psd_stats* s = stats ? stats : &local;
    Timer tt;
    tt.start(c->stream);
    if ((*info = zsghess_dev(c, n, p, c->zH, wantZ ? c->zZ : nullptr, Sarg.data(), s)) != 0) return *info;
    const double keep_hess = s->ms_hess, keep_formq = s->ms_formq, keep_bh = s->bytes_hess;
    Timer ti;
    ti.start(c->stream);
    int rc = zgrun_iteration(c, n, p, c->zH, wantZ ? c->zZ : nullptr, Sarg.data(), wantT, wantZ, maxitfac, alpha, beta,
                             ascale, s, info);
    s->ms_hess = keep_hess;
    s->ms_formq = keep_formq;
    s->bytes_hess = keep_bh;
    s->ms_iter = ti.stop(c->stream);
    s->ms_total = tt.stop(c->stream);
    if (schurindex) *schurindex = left ? p : 1;
    if (rc < 0 || rc >= PSD_INFO_NOTIMPL) return rc;
    for (int j = 1; j <= p; ++j) PSD_CHECK(psd_rt_d2h(A[slotA(j) - 1], c->zH + (size_t)(j - 1) * nn, nn * 16, c->stream));
    if (wantZ)
        for (int j = 1; j <= p; ++j)
            PSD_CHECK(psd_rt_d2h(Z[slotZ(j) - 1], c->zZ + (size_t)(j - 1) * nn, nn * 16, c->stream));
    PSD_CHECK(psd_rt_sync(c->stream));
    return rc;
}

// pschur!(H1, Hs, S; ...) for ComplexF64 with a signed S — generalized.jl:166-931
int zsigned_hess_host(psd_ctx* c, int n, int p, double* const* H, const uint8_t* S, double* const* Q, int wantT,
                      int wantZ, int maxitfac, double* alpha, double* beta, int32_t* ascale, psd_stats* stats, int* info) {
    const int mlog = 2 * maxitfac * n + n + 16;
    if ((*info = c->zreserve(n, p, true, mlog)) != 0) return *info;
    if ((*info = c->zgreserve(n, p)) != 0) return *info;
    const size_t nn = (size_t)n * n;
    for (int j = 0; j < p; ++j) PSD_CHECK(psd_rt_h2d(c->zH + j * nn, H[j], nn * 16, c->stream));
    if (wantZ)
        for (int j = 0; j < p; ++j) PSD_CHECK(psd_rt_h2d(c->zZ + j * nn, Q[j], nn * 16, c->stream));
    psd_stats local;
    memset(&local, 0, sizeof(local));
    psd_stats* s = stats ? stats : &local;
    Timer t;
    t.start(c->stream);
    int rc = zgrun_iteration(c, n, p, c->zH, wantZ ? c->zZ : nullptr, S, wantT, wantZ, maxitfac, alpha, beta, ascale, s,
                             info);
    s->ms_iter = s->ms_total = t.stop(c->stream);
    if (rc < 0 || rc >= PSD_INFO_NOTIMPL) return rc;
    for (int j = 0; j < p; ++j) PSD_CHECK(psd_rt_d2h(H[j], c->zH + j * nn, nn * 16, c->stream));
    if (wantZ)
        for (int j = 0; j < p; ++j) PSD_CHECK(psd_rt_d2h(Q[j], c->zZ + j * nn, nn * 16, c->stream));
    PSD_CHECK(psd_rt_sync(c->stream));
    return rc;
}

}  // namespace

extern "C" {

// _phessenberg!(A, S) for ComplexF64 — generalized.jl:988-1082
int psd_z_gphessenberg(psd_ctx* c, int n, int p, double* const* A, const uint8_t* S, double* const* Q, psd_stats* stats,
                       int* info) {
    int dummy;
    if (!info) info = &dummy;
    if (stats) memset(stats, 0, sizeof(*stats));
    if (!c) return *info = -1;
    if ((*info = check_dims(n, p)) != 0) return *info;
    if (!A) return *info = -4;
    if (S && !S[0]) return *info = -5;
    if ((*info = c->zreserve(n, p, true, 16)) != 0) return *info;
    if ((*info = c->zgreserve(n, p)) != 0) return *info;
    const size_t nn = (size_t)n * n;
    for (int j = 0; j < p; ++j) PSD_CHECK(psd_rt_h2d(c->zH + j * nn, A[j], nn * 16, c->stream));
    *info = zsghess_dev(c, n, p, c->zH, Q ? c->zZ : nullptr, S, stats);
    if (*info != 0) return *info;
    for (int j = 0; j < p; ++j) PSD_CHECK(psd_rt_d2h(A[j], c->zH + j * nn, nn * 16, c->stream));
    if (Q)
        for (int j = 0; j < p; ++j) PSD_CHECK(psd_rt_d2h(Q[j], c->zZ + j * nn, nn * 16, c->stream));
    PSD_CHECK(psd_rt_sync(c->stream));
    return 0;
}

int psd_z_phessenberg(psd_ctx* c, int n, int p, double* const* A, double* tau, psd_stats* stats, int* info) {
    int dummy;
    if (!info) info = &dummy;
    if (stats) memset(stats, 0, sizeof(*stats));
    if (!c) return *info = -1;
    if ((*info = check_dims(n, p)) != 0) return *info;
    if (!A || !tau) return *info = -4;
    if ((*info = c->zreserve(n, p, true, 16)) != 0) return *info;
    const size_t nn = (size_t)n * n;
    for (int j = 0; j < p; ++j) PSD_CHECK(psd_rt_h2d(c->zH + j * nn, A[j], nn * 16, c->stream));
    Timer tk;
    tk.start(c->stream);
    if ((*info = zhessenberg_dev(c, n, p, c->zH, c->ztau)) != 0) return *info;
    const double ms = tk.stop(c->stream);
    for (int j = 0; j < p; ++j) PSD_CHECK(psd_rt_d2h(A[j], c->zH + j * nn, nn * 16, c->stream));
    PSD_CHECK(psd_rt_d2h(tau, c->ztau, sizeof(psd_z) * (size_t)n * p, c->stream));
    PSD_CHECK(psd_rt_sync(c->stream));
    PSD_CHECK(psd_rt_last_error());
    if (stats) {
        stats->ms_hess = stats->ms_total = ms;
        stats->bytes_hess = 2.0 * 16.0 * p * (5.0 / 6.0) * (double)n * n * n;
    }
    return *info = 0;
}

int psd_z_pschur_dev(psd_ctx* c, int n, int p, double* dA_, char orient, int wantT, int wantZ, int maxitfac, double* dZ_,
                     double* alpha, double* beta, int32_t* ascale, int* schurindex, psd_stats* stats,
                     int32_t* sweeplog, int64_t maxlog, int* info) {
    int dummy;
    if (!info) info = &dummy;
    if (stats) memset(stats, 0, sizeof(*stats));
    if (!c) return *info = -1;
    if ((*info = check_dims(n, p)) != 0) return *info;
    if (!dA_) return *info = -4;
    if (orient != 'R' && orient != 'L') return *info = -5;
    if (maxitfac < 1) return *info = -8;
    if (wantZ && !dZ_) return *info = -9;
    if (!alpha || !beta || !ascale) return *info = -10;
    psd_z* dA = reinterpret_cast<psd_z*>(dA_);
    psd_z* dZ = reinterpret_cast<psd_z*>(dZ_);
    const bool left = orient == 'L';
    const int mlog = 2 * maxitfac * n + n + 16;
    if ((*info = c->zreserve(n, p, false, mlog)) != 0) return *info;
    Timer tall, tph;
    tall.start(c->stream);
    if (left && p > 1) PSD_LAUNCH(psd_reverse_blocks, psd_dim3(n, p / 2), 64, 0, c->stream, dA_, 2 * n, n, 0, p);
    tph.start(c->stream);
    if ((*info = zhessenberg_dev(c, n, p, dA, c->ztau)) != 0) return *info;
    const double ms_hess = tph.stop(c->stream);
    tph.start(c->stream);
    if (wantZ) {
        if ((*info = zformq_dev(c, n, p, dA, c->ztau, dZ)) != 0) return *info;
    }
    PSD_LAUNCH(psd_ztriu, psd_dim3(n, p), 64, 0, c->stream, dA, n);
    const double ms_formq = tph.stop(c->stream);
    tph.start(c->stream);
    psd_stats local;
    memset(&local, 0, sizeof(local));
    psd_stats* s = stats ? stats : &local;
    int rc = zrun_iteration(c, n, p, dA, dZ, wantT, wantZ, maxitfac, alpha, beta, ascale, s, sweeplog, maxlog, info);
    const double ms_iter = tph.stop(c->stream);
    if (left && p > 1) {
        PSD_LAUNCH(psd_reverse_blocks, psd_dim3(n, p / 2), 64, 0, c->stream, dA_, 2 * n, n, 0, p);
        if (wantZ && p > 2)
            PSD_LAUNCH(psd_reverse_blocks, psd_dim3(n, (p - 1) / 2), 64, 0, c->stream, dZ_, 2 * n, n, 1, p - 1);
    }
    s->ms_hess = ms_hess;
    s->ms_formq = ms_formq;
    s->ms_iter = ms_iter;
    s->ms_total = tall.stop(c->stream);
    s->bytes_hess = 2.0 * 16.0 * p * (5.0 / 6.0) * (double)n * n * n;
    s->bytes_formq = wantZ ? 2.0 * 16.0 * p * (double)n * n * n / 3.0 : 0.0;
    if (schurindex) *schurindex = left ? p : 1;
    return rc;
}

int psd_z_pschur(psd_ctx* c, int n, int p, double* const* A, const uint8_t* S, char orient, int wantT, int wantZ,
                 int maxitfac, double* const* Z, double* alpha, double* beta, int32_t* ascale, int* schurindex,
                 psd_stats* stats, int32_t* sweeplog, int64_t maxlog, int* info) {
    int dummy;
    if (!info) info = &dummy;
    if (stats) memset(stats, 0, sizeof(*stats));
    if (!c) return *info = -1;
    if ((*info = check_dims(n, p)) != 0) return *info;
    if (!A) return *info = -4;
    if (orient != 'R' && orient != 'L') return *info = -6;
    if (wantZ && !Z) return *info = -10;
    if (S)
        for (int j = 0; j < p; ++j)
            if (!S[j])  // signed case (generalized.jl:138-146)
                return zsigned_pschur_host(c, n, p, A, S, orient, wantT, wantZ, maxitfac, Z, alpha, beta, ascale,
                                           schurindex, stats, info);
    const int mlog = 2 * (maxitfac > 0 ? maxitfac : 1) * n + n + 16;
    if ((*info = c->zreserve(n, p, true, mlog)) != 0) return *info;
    const size_t nn = (size_t)n * n;
    Timer tc;
    tc.start(c->stream);
    for (int j = 0; j < p; ++j) PSD_CHECK(psd_rt_h2d(c->zH + j * nn, A[j], nn * 16, c->stream));
    double ms_copy = tc.stop(c->stream);
    psd_stats local;
    psd_stats* s = stats ? stats : &local;
    int rc = psd_z_pschur_dev(c, n, p, reinterpret_cast<double*>(c->zH), orient, wantT, wantZ, maxitfac,
                              wantZ ? reinterpret_cast<double*>(c->zZ) : nullptr, alpha, beta, ascale, schurindex, s,
                              sweeplog, maxlog, info);
    if (rc < 0 || rc >= PSD_INFO_NOTIMPL) return rc;
    tc.start(c->stream);
    for (int j = 0; j < p; ++j) PSD_CHECK(psd_rt_d2h(A[j], c->zH + j * nn, nn * 16, c->stream));
    if (wantZ)
        for (int j = 0; j < p; ++j) PSD_CHECK(psd_rt_d2h(Z[j], c->zZ + j * nn, nn * 16, c->stream));
    PSD_CHECK(psd_rt_sync(c->stream));
    ms_copy += tc.stop(c->stream);
    s->ms_copy = ms_copy;
    return rc;
}

int psd_z_pschur_hess(psd_ctx* c, int n, int p, double* const* H, const uint8_t* S, double* const* Q, int wantT,
                      int wantZ, int maxitfac, double* alpha, double* beta, int32_t* ascale, psd_stats* stats,
                      int32_t* sweeplog, int64_t maxlog, int* info) {
    int dummy;
    if (!info) info = &dummy;
    if (stats) memset(stats, 0, sizeof(*stats));
    if (!c) return *info = -1;
    if ((*info = check_dims(n, p)) != 0) return *info;
    if (!H) return *info = -4;
    bool signedS = false;
    if (S) {
        if (!S[0]) return *info = -5;  // generalized.jl:182
        for (int j = 0; j < p; ++j) signedS = signedS || !S[j];
    }
    if (wantZ && !Q) return *info = -6;
    if (maxitfac < 1) return *info = -9;
    if (signedS) return zsigned_hess_host(c, n, p, H, S, Q, wantT, wantZ, maxitfac, alpha, beta, ascale, stats, info);
    const int mlog = 2 * maxitfac * n + n + 16;
    if ((*info = c->zreserve(n, p, true, mlog)) != 0) return *info;
    const size_t nn = (size_t)n * n;
    for (int j = 0; j < p; ++j) PSD_CHECK(psd_rt_h2d(c->zH + j * nn, H[j], nn * 16, c->stream));
    if (wantZ)
        for (int j = 0; j < p; ++j) PSD_CHECK(psd_rt_h2d(c->zZ + j * nn, Q[j], nn * 16, c->stream));
    psd_stats local;
    memset(&local, 0, sizeof(local));
    psd_stats* s = stats ? stats : &local;
    Timer t;
    t.start(c->stream);
    int rc = zrun_iteration(c, n, p, c->zH, wantZ ? c->zZ : nullptr, wantT, wantZ, maxitfac, alpha, beta, ascale, s,
                            sweeplog, maxlog, info);
    s->ms_iter = s->ms_total = t.stop(c->stream);
    if (rc < 0 || rc >= PSD_INFO_NOTIMPL) return rc;
    for (int j = 0; j < p; ++j) PSD_CHECK(psd_rt_d2h(H[j], c->zH + j * nn, nn * 16, c->stream));
    if (wantZ)
        for (int j = 0; j < p; ++j) PSD_CHECK(psd_rt_d2h(Q[j], c->zZ + j * nn, nn * 16, c->stream));
    PSD_CHECK(psd_rt_sync(c->stream));
    return rc;
}

}  // extern "C"

// =================================================================================================
// ordschur! (complex)
namespace {

size_t ord_lds_bytes(int p, int W) {
    size_t b = (size_t)p * W * (W + 1) * 16 + (size_t)13 * p * 16 + ((size_t)p + 2) * 8 + (size_t)p * 4 + 64;
    return (b + 15) & ~(size_t)15;
}
int choose_window_ord(int p) {
    const int cand[] = {32, 24, 20, 16, 12, 10, 8, 6, 4};
    for (int W : cand)
        if (ord_lds_bytes(p, W) <= 155 * 1024) return W;
    return 0;
}

// device arrays in internal right order; select on host
int zordschur_dev(psd_ctx* c, int n, int p, psd_z* dH, psd_z* dZ, const uint8_t* select, int wantZ, double* alpha,
                  double* beta, int32_t* ascale, psd_stats* stats, int* info) {
    int W = choose_window_ord(p);
    if (W == 0) return *info = PSD_INFO_NOTIMPL;
    // pipelined driver (psd_oslot): the selected eigenvalues travel a window apart; with many of them shorter windows
    // (the tick lasts as long as its longest window) are the faster schedule
    const bool pipe = c->ord_pipe != 0;
    if (pipe) {
        int nsel = 0;
        for (int q = 0; q < n; ++q) nsel += select[q] ? 1 : 0;
        if (nsel >= 8 && n >= 128 && W > 16) W = 16;
        PSD_CHECK(c->ztreserve(p));
    }
    PSD_CHECK(psd_rt_h2d(c->osel, select, (size_t)n, c->stream));
    psd_oparams O;
    O.z.H = dH;
    O.z.zlo = 1;  // (ordschur! is not sharded: every Z_m is this context's)
    O.z.zhi = p;
    O.z.zcoff = O.z.zc3off = 0;
    O.z.zcdefer = 0;
    O.z.zslG = 1;
    O.z.zslmem = nullptr;
    O.z.zslerr = nullptr;
    O.z.Z = wantZ ? dZ : nullptr;
    O.z.st = c->zst;
    O.z.desc = c->zdesc;
    O.z.tr = c->ztr;
    O.z.cnt = c->zcnt;
    O.z.dG = c->zdG;
    O.z.alpha = c->zalpha;
    O.z.beta = c->zbeta;
    O.z.ascale = c->zascale;
    O.z.log = c->zlog;
    O.st = c->ost;
    O.select = c->osel;
    const size_t lds_step = ord_lds_bytes(p, W);
#ifndef PSD_HOSTSIM
    if (lds_step > c->ostep_lds_set) {
        PSD_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(psd_zord_step),
                                      hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_step));
        c->ostep_lds_set = lds_step;
    }
#endif
    const size_t lds_apply = PSD_ZTR_LDS_BYTES + (size_t)32 * (PSD_ZAPPLY_NT + 1) * sizeof(psd_z);
    const int tiles = (n + PSD_ZAPPLY_NT - 1) / PSD_ZAPPLY_NT;
    psd_ostate hst;
    memset(&hst, 0, sizeof(hst));
    long long launched = 0;
    const long long cap = (long long)n * ((long long)n / (W > 1 ? W - 1 : 1) + 2) + 1024;
    Timer t;
    t.start(c->stream);
    if (pipe) {
        O.st = c->ombst;
        O.z.desc = c->ztdesc;
        O.z.cnt = c->ztcnt;
        O.z.tr = c->zttr;
#ifndef PSD_HOSTSIM
        if (lds_step > c->ostep_mb_lds_set) {
            PSD_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(psd_zord_step_mb),
                                          hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_step));
            c->ostep_mb_lds_set = lds_step;
        }
        const int wl_grid = 2048;
#else
        const int wl_grid = 6;
#endif
        const size_t lds_wl = lds_apply + sizeof(int) * 2 * (PSD_TRAIN_MAX + 2);
        PSD_CHECK(psd_rt_memset(c->ztdesc, 0, sizeof(psd_zapply_desc) * PSD_TRAIN_MAX, c->stream));
        PSD_LAUNCH(psd_ord1_init_mb, psd_dim3(1), 64, 0, c->stream, c->ombst, c->oslots, c->omb, n, p, wantZ, W);
        psd_omb hg;
        memset(&hg, 0, sizeof(hg));
        for (;;) {
            for (int b = 0; b < 16; ++b) {
                PSD_LAUNCH(psd_ord1_plan, psd_dim3(1), 64, 0, c->stream, c->ombst, c->oslots, c->omb, c->osel);
                PSD_LAUNCH(psd_zord_step_mb, psd_dim3(PSD_O_SLOTS), PSD_STEP_NT, lds_step, c->stream, O, p, p + 8);
                // rows of every window, then columns (a row operation of one window meets a column operation of another in
                // off-diagonal blocks), then the Schur vectors: the work-list form of the trains
                for (int pass : {2, 1, 3})
                    PSD_LAUNCH(psd_zq_apply_wl, psd_dim3(wl_grid), PSD_ZAPPLY_NT, lds_wl, c->stream, O.z, n, p, p + 8, pass, PSD_O_SLOTS, (int)lds_apply);
                ++launched;
            }
            PSD_CHECK(psd_rt_d2h(&hg, c->omb, sizeof(hg), c->stream));
            PSD_CHECK(psd_rt_sync(c->stream));
            if (hg.phase == PSD_OPH_DONE) break;
            if (launched > cap) return *info = PSD_INFO_RUNTIME + 0xfffd;
        }
        std::vector<psd_ostate> hs(PSD_O_SLOTS);
        PSD_CHECK(psd_rt_d2h(hs.data(), c->ombst, sizeof(psd_ostate) * PSD_O_SLOTS, c->stream));
        PSD_CHECK(psd_rt_sync(c->stream));
        hst.phase = PSD_OPH_DONE;
        hst.info = hg.info;
        for (const psd_ostate& q : hs) {
            hst.nswaps += q.nswaps;
            hst.nwindows += q.nwindows;
        }
    } else {
    PSD_LAUNCH(psd_zord_init, psd_dim3(1), 64, 0, c->stream, O, n, p, wantZ, W);
    for (;;) {
        for (int b = 0; b < 32; ++b) {
            PSD_LAUNCH(psd_zord_step, psd_dim3(1), PSD_STEP_NT, lds_step, c->stream, O);
            PSD_LAUNCH(psd_zq_apply, psd_dim3(tiles, p, 3), PSD_ZAPPLY_NT, lds_apply, c->stream, O.z, n, p);
            ++launched;
        }
        PSD_CHECK(psd_rt_d2h(&hst, c->ost, sizeof(hst), c->stream));
        PSD_CHECK(psd_rt_sync(c->stream));
        if (hst.phase == PSD_OPH_DONE) break;
        if (launched > cap) return *info = PSD_INFO_RUNTIME + 0xfffd;
    }
    }
    if (hst.info == 0) {
        PSD_LAUNCH(psd_zord_values, psd_dim3((n + 255) / 256), 256, 0, c->stream, O.z, n, p);
        PSD_CHECK(psd_rt_d2h(alpha, c->zalpha, sizeof(psd_z) * n, c->stream));
        PSD_CHECK(psd_rt_d2h(beta, c->zbeta, sizeof(double) * n, c->stream));
        std::vector<int> hsc(n, 0);
        PSD_CHECK(psd_rt_d2h(hsc.data(), c->zascale, sizeof(int) * n, c->stream));
        PSD_CHECK(psd_rt_sync(c->stream));
        for (int q = 0; q < n; ++q) ascale[q] = hsc[q];
    }
    const double ms = t.stop(c->stream);
    PSD_CHECK(psd_rt_last_error());
    if (stats) {
        stats->ms_iter = stats->ms_total = ms;
        stats->nsweeps = hst.nswaps;  // adjacent swaps performed
        stats->nwindows = hst.nwindows;
        stats->nlaunch_step = (int32_t)launched;
        stats->window = W;
    }
    return *info = hst.info;  // 0, 2000+j (IllConditionedException(j)), 3000 (SingularException)
}

}  // namespace

extern "C" {

int psd_z_ordschur(psd_ctx* c, int n, int p, double* const* T, double* const* Z, char orient, int schurindex,
                   const uint8_t* select, int wantZ, double* alpha, double* beta, int32_t* ascale, psd_stats* stats,
                   int* info) {
    int dummy;
    if (!info) info = &dummy;
    if (stats) memset(stats, 0, sizeof(*stats));
    if (!c) return *info = -1;
    if ((*info = check_dims(n, p)) != 0) return *info;
    if (!T) return *info = -4;
    if (wantZ && !Z) return *info = -5;
    if (orient != 'R' && orient != 'L') return *info = -6;
    if (!select) return *info = -8;
    std::vector<int> slotA, slotZ;
    if (!ord_slots(orient, schurindex, p, slotA, slotZ)) return *info = -7;  // ArgumentError, ordschur.jl:32
    if ((*info = c->zreserve(n, p, true, 16)) != 0) return *info;
    const size_t nn = (size_t)n * n;
    for (int j = 0; j < p; ++j) PSD_CHECK(psd_rt_h2d(c->zH + j * nn, T[slotA[j]], nn * 16, c->stream));
    if (wantZ)
        for (int j = 0; j < p; ++j) PSD_CHECK(psd_rt_h2d(c->zZ + j * nn, Z[slotZ[j]], nn * 16, c->stream));
    int rc = zordschur_dev(c, n, p, c->zH, c->zZ, select, wantZ, alpha, beta, ascale, stats, info);
    if (rc < 0 || rc >= PSD_INFO_NOTIMPL) return rc;
    for (int j = 0; j < p; ++j) PSD_CHECK(psd_rt_d2h(T[slotA[j]], c->zH + j * nn, nn * 16, c->stream));
    if (wantZ)
        for (int j = 0; j < p; ++j) PSD_CHECK(psd_rt_d2h(Z[slotZ[j]], c->zZ + j * nn, nn * 16, c->stream));
    PSD_CHECK(psd_rt_sync(c->stream));
    return rc;
}

}  // extern "C"

// =================================================================================================
// ordschur! (real)
namespace {

size_t rord_lds_bytes(int p, int W) {
    size_t b = ((size_t)p * W * (W + 1) + (size_t)p * (PSD_RORD_SCR + 52) + 4 + 192 + psd_rord_tree_doubles(p)) * 8 +
               (size_t)p * 5 + 80;
    return (b + 15) & ~(size_t)15;
}
int choose_window_rord(int p) {
    const int cand[] = {32, 24, 20, 16, 12, 10, 8, 6};
    for (int W : cand)
        if (rord_lds_bytes(p, W) <= 155 * 1024) return W;
    return 0;
}

// Sint != nullptr: GeneralizedPeriodicSchur (signed swaps), eigenvalues returned in the scaled form (alpha complex)
int rordschur_dev(psd_ctx* c, int n, int p, double* dH, double* dZ, const uint8_t* select, int wantZ, double* wr,
                  double* wi, psd_stats* stats, int* info, const uint8_t* Sint = nullptr, double* alpha = nullptr,
                  double* beta = nullptr, int32_t* ascale = nullptr) {
    int W = choose_window_rord(p);
    if (W == 0) return *info = PSD_INFO_NOTIMPL;
    if ((*info = c->roreserve(n, p)) != 0) return *info;
    // Pipelined driver (psd_roslot): the selected blocks travel one window apart.  The tick lasts as long as its longest
    // window, so with many blocks to move shorter windows (more blocks under way) are the faster schedule.
    const bool pipe = c->ord_pipe != 0;
    if (pipe) {
        int nsel = 0;
        for (int q = 0; q < n; ++q) nsel += select[q] ? 1 : 0;
        if (nsel >= 8 && n >= 128 && W > 16) W = 16;
    }
    if (Sint) {
        if ((*info = c->greserve(n, p, 16)) != 0) return *info;
        std::vector<unsigned char> hS(p, 1);
        for (int l = 0; l < p; ++l) hS[l] = Sint[l] ? 1 : 0;
        PSD_CHECK(psd_rt_h2d(c->gS, hS.data(), (size_t)p, c->stream));
        PSD_CHECK(psd_rt_sync(c->stream));
    }
    PSD_CHECK(psd_rt_h2d(c->rosel, select, (size_t)n, c->stream));
    psd_roparams P;
    P.mb = nullptr;
    P.slots = nullptr;
    P.H = dH;
    P.Z = wantZ ? dZ : nullptr;
    P.st = c->rost;
    P.desc = c->desc;
    P.tq = c->rotq;
    P.cnt = c->cnt;
    P.select = c->rosel;
    P.wr = c->wr;
    P.wi = c->wi;
    P.xscr = c->roxscr;
    P.S = Sint ? c->gS : nullptr;
    P.alpha = Sint ? c->galpha : nullptr;
    P.beta = Sint ? c->gbeta : nullptr;
    P.ascale = Sint ? c->gascale : nullptr;
    const size_t lds_step = rord_lds_bytes(p, W);
#ifndef PSD_HOSTSIM
    if (lds_step > c->rostep_lds_set) {
        PSD_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(psd_rord_step),
                                      hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_step));
        c->rostep_lds_set = lds_step;
    }
#endif
    const size_t lds_apply = sizeof(psd_tq) * PSD_RORD_CAP + (size_t)32 * (PSD_APPLY_NT + 1) * 8;
    const int tiles = (n + PSD_APPLY_NT - 1) / PSD_APPLY_NT;
    psd_rostate hst;
    memset(&hst, 0, sizeof(hst));
    long long launched = 0;
    const long long cap = 2LL * n * ((long long)n / (W > 4 ? W - 4 : 1) + 2) + 1024;
    Timer t;
    t.start(c->stream);
    if (pipe) {
        P.mb = c->romb;
        P.slots = c->roslots;
        P.tq = c->rotq_mb;
        P.cnt = c->rocnt_mb;
        P.desc = c->rodesc_mb;
#ifndef PSD_HOSTSIM
        if (lds_step > c->rostep_mb_lds_set) {
            PSD_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(psd_rord_step_mb),
                                          hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_step));
            c->rostep_mb_lds_set = lds_step;
        }
#endif
        PSD_LAUNCH(psd_rord_init_mb, psd_dim3(1), 64, 0, c->stream, P, n, p, wantZ, W);
        psd_romb hg;
        memset(&hg, 0, sizeof(hg));
        for (;;) {
            for (int b = 0; b < 16; ++b) {
                PSD_LAUNCH(psd_rord_plan, psd_dim3(1), 64, 0, c->stream, P);
                PSD_LAUNCH(psd_rord_step_mb, psd_dim3(PSD_RO_SLOTS), PSD_STEP_NT, lds_step, c->stream, P);
                PSD_LAUNCH(psd_rord_apply_mb, psd_dim3(tiles, p, 3 * PSD_RO_SLOTS), PSD_APPLY_NT, lds_apply, c->stream, P, n, p, 0);
                PSD_LAUNCH(psd_rord_apply_mb, psd_dim3(tiles, p, 3 * PSD_RO_SLOTS), PSD_APPLY_NT, lds_apply, c->stream, P, n, p, 1);
                ++launched;
            }
            PSD_CHECK(psd_rt_d2h(&hg, c->romb, sizeof(hg), c->stream));
            PSD_CHECK(psd_rt_sync(c->stream));
            if (hg.phase == PSD_ROPH_DONE) break;
            if (launched > cap) return *info = PSD_INFO_RUNTIME + 0xfffc;
        }
        std::vector<psd_roslot> hs(PSD_RO_SLOTS);
        PSD_CHECK(psd_rt_d2h(hs.data(), c->roslots, sizeof(psd_roslot) * PSD_RO_SLOTS, c->stream));
        PSD_CHECK(psd_rt_sync(c->stream));
        hst.phase = PSD_ROPH_DONE;
        hst.info = hg.info;
        for (const psd_roslot& q : hs) {
            hst.nswaps += q.nswaps;
            hst.nwindows += q.nwindows;
            for (int e = 0; e < 6; ++e) hst.cyc[e] += q.cyc[e];
        }
    } else {
    PSD_LAUNCH(psd_rord_init, psd_dim3(1), 64, 0, c->stream, P, n, p, wantZ, W);
    for (;;) {
        for (int b = 0; b < 32; ++b) {
            PSD_LAUNCH(psd_rord_step, psd_dim3(1), PSD_STEP_NT, lds_step, c->stream, P);
            PSD_LAUNCH(psd_rord_apply, psd_dim3(tiles, p, 3), PSD_APPLY_NT, lds_apply, c->stream, P, n, p);
            ++launched;
        }
        PSD_CHECK(psd_rt_d2h(&hst, c->rost, sizeof(hst), c->stream));
        PSD_CHECK(psd_rt_sync(c->stream));
        if (hst.phase == PSD_ROPH_DONE) break;
        if (launched > cap) return *info = PSD_INFO_RUNTIME + 0xfffc;
    }
    }
    if (hst.info == 0 && Sint) {
        PSD_LAUNCH(psd_grord_values, psd_dim3((n + 63) / 64), 64, 0, c->stream, P, n, p);
        PSD_LAUNCH(psd_grord_cleanup, psd_dim3(n), 64, 0, c->stream, P, n);
        PSD_CHECK(psd_rt_d2h(alpha, c->galpha, sizeof(psd_z) * n, c->stream));
        PSD_CHECK(psd_rt_d2h(beta, c->gbeta, sizeof(double) * n, c->stream));
        std::vector<int> hsc(n, 0);
        PSD_CHECK(psd_rt_d2h(hsc.data(), c->gascale, sizeof(int) * n, c->stream));
        PSD_CHECK(psd_rt_sync(c->stream));
        for (int q = 0; q < n; ++q) ascale[q] = hsc[q];
    } else if (hst.info == 0) {
        PSD_LAUNCH(psd_rord_values, psd_dim3((n + 63) / 64), 64, 0, c->stream, P, n, p);
        PSD_LAUNCH(psd_rord_cleanup, psd_dim3(n), 64, 0, c->stream, P, n);
        PSD_CHECK(psd_rt_d2h(wr, c->wr, sizeof(double) * n, c->stream));
        PSD_CHECK(psd_rt_d2h(wi, c->wi, sizeof(double) * n, c->stream));
        PSD_CHECK(psd_rt_sync(c->stream));
    }
    const double ms = t.stop(c->stream);
    PSD_CHECK(psd_rt_last_error());
    if (stats) {
        stats->ms_iter = stats->ms_total = ms;
        stats->nsweeps = hst.nswaps;
        stats->nwindows = hst.nwindows;
        for (int q = 0; q < 6; ++q) stats->step_cycles[q] = hst.cyc[q];
        stats->nlaunch_step = (int32_t)launched;
        stats->window = W;
    }
    return *info = hst.info;
}

}  // namespace

extern "C" {

int psd_d_ordschur(psd_ctx* c, int n, int p, double* const* T, double* const* Z, char orient, int schurindex,
                   const uint8_t* select, int wantZ, double* wr, double* wi, psd_stats* stats, int* info) {
    int dummy;
    if (!info) info = &dummy;
    if (stats) memset(stats, 0, sizeof(*stats));
    if (!c) return *info = -1;
    if ((*info = check_dims(n, p)) != 0) return *info;
    if (!T) return *info = -4;
    if (wantZ && !Z) return *info = -5;
    if (orient != 'R' && orient != 'L') return *info = -6;
    if (!select) return *info = -8;
    std::vector<int> slotA, slotZ;
    if (!ord_slots(orient, schurindex, p, slotA, slotZ)) return *info = -7;  // rordschur.jl:25
    if ((*info = c->reserve(n, p, true, 16)) != 0) return *info;
    const size_t nn = (size_t)n * n;
    for (int j = 0; j < p; ++j) PSD_CHECK(psd_rt_h2d(c->dH + j * nn, T[slotA[j]], nn * 8, c->stream));
    if (wantZ)
        for (int j = 0; j < p; ++j) PSD_CHECK(psd_rt_h2d(c->dZ + j * nn, Z[slotZ[j]], nn * 8, c->stream));
    int rc = rordschur_dev(c, n, p, c->dH, c->dZ, select, wantZ, wr, wi, stats, info);
    if (rc < 0 || rc >= PSD_INFO_NOTIMPL) return rc;
    for (int j = 0; j < p; ++j) PSD_CHECK(psd_rt_d2h(T[slotA[j]], c->dH + j * nn, nn * 8, c->stream));
    if (wantZ)
        for (int j = 0; j < p; ++j) PSD_CHECK(psd_rt_d2h(Z[slotZ[j]], c->dZ + j * nn, nn * 8, c->stream));
    PSD_CHECK(psd_rt_sync(c->stream));
    return rc;
}

}  // extern "C"


// =================================================================================================
// real generalized (signed) periodic QZ — rgeneralized.jl
namespace {

// dH [p][n][n] (H_1 Hessenberg), dZ [p][n][n] or nullptr; S host signature (S[0] true)
// hessmode: run stage 2 of the signed Hessenberg reduction instead of the QZ iteration (same step/apply machinery)
int giterate_dev(psd_ctx* c, int n, int p, double* dH, double* dZ, const uint8_t* S, int wantT, int wantZ, int maxitfac,
                 psd_gstate* st_out, psd_stats* stats, int maxlog, int hessmode = 0) {
    const int W = choose_window(p, 8);
    if (W == 0) return PSD_INFO_NOTIMPL;
    std::vector<unsigned char> hS(p, 1);
    for (int l = 0; l < p; ++l) hS[l] = (!S || S[l]) ? 1 : 0;
    PSD_CHECK(psd_rt_h2d(c->gS, hS.data(), (size_t)p, c->stream));
    PSD_CHECK(psd_rt_sync(c->stream));
    psd_gparams P;
    P.H = dH;
    {
        int zl = 0, zh = p;
        c->slice(p, zl, zh);
        P.zlo = zl + 1;
        P.zhi = zh;
    }
    P.Z = wantZ ? dZ : nullptr;
    P.S = c->gS;
    P.st = c->gst;
    P.desc = c->gdesc;
    P.tr = c->gtr;
    P.cnt = c->gcnt;
    P.dG = c->gdG;
    P.alpha = c->galpha;
    P.beta = c->gbeta;
    P.ascale = c->gascale;
    P.log = c->glog;
    P.xscr = c->gxscr;
    size_t lds_step = step_lds_bytes(p, W, 8);
    // scan form of the sweep windows (psd_gs3_run; PSD_GSCAN=0: the single-wave sweep): rotation tables and the helper
    // wavefronts' command block behind everything the state machine uses
    P.gcoff = P.gtaboff = 0;
    int gwaves = 1;
    (void)gwaves;
    {
        int gscan = 1;
        if (const char* e = psd_env("PSD_GSCAN")) gscan = atoi(e);
        const size_t need = lds_step + (size_t)p * 8 * sizeof(double) + 128;
        if (gscan && !hessmode && p >= 2 && p <= 64 && need <= (size_t)160 * 1024) {
            P.gtaboff = (int)lds_step;
            P.gcoff = (int)(lds_step + (size_t)p * 8 * sizeof(double));
            lds_step = need;
#ifndef PSD_HOSTSIM
            gwaves = 4;
#endif
        }
    }
#ifndef PSD_HOSTSIM
    if (lds_step > c->gstep_lds_set) {
        PSD_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(psd_gq_step),
                                      hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_step));
        c->gstep_lds_set = lds_step;
    }
#endif
    // stage 2 of the signed Hessenberg reduction runs as a pipeline over the factors (psd_gq_hess_step) when its LDS
    // (window + mailboxes) fits; PSD_HESS_SERIAL=1 (test hook) keeps the single-wave chase
    const size_t lds_hess = psd_ghess_lds_bytes(p, W);
    const int hess_links = psd_ghess_links(p), hess_waves = psd_ghess_waves(p);
    const char* hserial = psd_env("PSD_HESS_SERIAL");
    const bool hess_pipe = hessmode && lds_hess <= (size_t)160 * 1024 && !(hserial && hserial[0] == '1');
    // (scan form of the stage-2 kernel, the default; PSD_HESS_SCAN=0: the pipeline of beats over the factors of round 2)
    int hess_scan = 1;
    if (const char* e = psd_env("PSD_HESS_SCAN")) hess_scan = atoi(e);
#ifndef PSD_HOSTSIM
    if (hess_pipe && lds_hess > c->ghess_lds_set) {
        PSD_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(psd_gq_hess_step),
                                      hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_hess));
        PSD_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(psd_gq_hess_step_scan),
                                      hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_hess));
        c->ghess_lds_set = lds_hess;
    }
#endif
    // multishift trains (as iterate_dev): M cursors, cursor 0 is the ordinary state machine
    const int tw = hessmode ? 0 : c->gtrain_m;
    const int M = (tw >= 2) ? ((tw > PSD_TRAIN_MAX) ? PSD_TRAIN_MAX : tw) : 1;
    P.cst = nullptr;
    P.cep = nullptr;
    P.tshift = nullptr;
    P.tick = 0;
    if (M > 1 || tw == -2) {
        PSD_CHECK(c->gtreserve(p));
        PSD_CHECK(psd_rt_memset(c->gtcst, 0, sizeof(psd_gstate) * PSD_TRAIN_MAX + sizeof(int) * (PSD_TRAIN_MAX + 8), c->stream));
        PSD_CHECK(psd_rt_memset(c->gtdesc, 0, sizeof(psd_gapply_desc) * PSD_TRAIN_MAX, c->stream));
        P.cst = c->gtcst;
        P.cep = (int*)(c->gtcst + PSD_TRAIN_MAX);
        P.tshift = c->gtshift;
        if (M > 1) {
            P.desc = c->gtdesc;
            P.cnt = c->gtcnt;
            P.tr = c->gttr;
#ifndef PSD_HOSTSIM
            PSD_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(psd_gq_step_train),
                                          hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_step));
#endif
        }
    }
    int train_oc = 50;  // (a position of a signed factor costs about 1 us, a tick's overhead about 50 us)
    if (const char* e = psd_env_diag("PSD_TRAIN_OC")) train_oc = atoi(e);  // (tuning hook)
    PSD_LAUNCH(psd_gq_init, psd_dim3(1), 256, 0, c->stream, P, n, p, wantT, wantZ, W, maxitfac, maxlog, hessmode,
               (tw == -2) ? -2 : M, train_oc);
    const size_t lds_apply = PSD_GTR_LDS_BYTES + (size_t)32 * (PSD_GAPPLY_NT + 1) * sizeof(double);
    const int tiles = (n + PSD_GAPPLY_NT - 1) / PSD_GAPPLY_NT;
    const int dtiles = (n + 255) / 256;
    const int batch = 32;
    psd_gstate hst;
    memset(&hst, 0, sizeof(hst));
    long long launched = 0;
    const int nbmin = (W - 5 > 0) ? ((W - 5 < 8) ? (W - 5) : 8) : 1;  // (trains run windows down to 8 positions)
    const long long cap = (long long)maxitfac * n * ((long long)n / nbmin + 8) + 8LL * n + 1024 + (hessmode ? (long long)n * n : 0);
    double sample_ms = 0.0;
    int samples = 0;
#ifndef PSD_HOSTSIM
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
    std::vector<std::pair<hipEvent_t, hipEvent_t>> pend;
    psd_poller poller(c, sample_ms, samples);
#endif
    for (;;) {
        for (int b = 0; b < batch; ++b) {
#ifndef PSD_HOSTSIM
            const bool sample = c->profile && ((launched & 15) == 0);
            if (sample) {
                (void)hipEventCreate(&ev0);
                (void)hipEventCreate(&ev1);
                (void)hipEventRecord(ev0, c->stream);
            }
#endif
            P.tick = (int)launched;
            if (hess_pipe && hess_scan && p <= 64)
                PSD_LAUNCH(psd_gq_hess_step_scan, psd_dim3(1), 256, lds_hess, c->stream, P, hess_links);  // (scan form: four wavefronts, one per SIMD)
            else if (hess_pipe)
                PSD_LAUNCH(psd_gq_hess_step, psd_dim3(1), 64 * hess_waves, lds_hess, c->stream, P, hess_links, 0);
            else if (M > 1)
                PSD_LAUNCH2(psd_gq_step_train, psd_dim3(M), PSD_STEP_NT, gwaves, lds_step, c->stream, P, p, p + 8);
            else
                PSD_LAUNCH2(psd_gq_step, psd_dim3(1), PSD_STEP_NT, gwaves, lds_step, c->stream, P);
#ifndef PSD_HOSTSIM
            if (sample) {
                (void)hipEventRecord(ev1, c->stream);
                pend.emplace_back(ev0, ev1);
            }
#endif
            if (M > 1) {
                PSD_LAUNCH(psd_gq_apply_train, psd_dim3(tiles, p, 2 * M), PSD_GAPPLY_NT, lds_apply, c->stream, P, n, p, p + 8, 0);
                PSD_LAUNCH(psd_gq_apply_train, psd_dim3(tiles, p, M), PSD_GAPPLY_NT, lds_apply, c->stream, P, n, p, p + 8, 1);
            } else {
                PSD_LAUNCH(psd_gq_apply, psd_dim3(tiles, p, 3), PSD_GAPPLY_NT, lds_apply, c->stream, P, n, p);
            }
            if (!hess_pipe) PSD_LAUNCH(psd_gq_defer, psd_dim3(dtiles), 256, 0, c->stream, P, n);
            ++launched;
        }
#ifdef PSD_HOSTSIM
        PSD_CHECK(psd_rt_d2h(&hst, c->gst, sizeof(hst), c->stream));
        PSD_CHECK(psd_rt_sync(c->stream));
#else
        PSD_CHECK(poller.poll(&hst, c->gst, sizeof(hst), pend));
#endif
        if (hst.phase == PSD_GPH_DONE) break;
        if (launched > cap) {
            *st_out = hst;
            return PSD_INFO_RUNTIME + 0xfffd;
        }
    }
#ifndef PSD_HOSTSIM
    PSD_CHECK(poller.finish(pend));
#endif
    if (psd_env_diag("PSD_GDBG")) {
        fprintf(stderr, "psd signed check stages (cycles/calls: test 1, tests 2-3, start rotations, train shifts, explicit start, cursor states, 2x2 block, split):");
        for (int q = 0; q < 8; ++q) fprintf(stderr, " %lld/%d", hst.dbg[q], hst.dbgn[q]);
        fprintf(stderr, " | cyc decide %lld load %lld chase %lld store %lld total %lld\n", hst.cyc[0], hst.cyc[1], hst.cyc[2], hst.cyc[3], hst.cyc[4]);
    }
    PSD_CHECK(psd_rt_last_error());
    *st_out = hst;
    if (stats) {
        stats->nlaunch_step = (int32_t)launched;
        stats->window = W;
        stats->step_kernel_ms_avg = samples ? sample_ms / samples : 0.0;
        stats->step_kernel_samples = samples;
    }
    return 0;
}

int grun_iteration(psd_ctx* c, int n, int p, double* dH, double* dZ, const uint8_t* S, int wantT, int wantZ,
                   int maxitfac, double* alpha, double* beta, int32_t* ascale, psd_stats* stats, int32_t* sweeplog,
                   int64_t maxlog_user, int* info) {
    const int maxlog = 2 * maxitfac * n + n + 16;
    if ((*info = c->greserve(n, p, maxlog)) != 0) return *info;
    psd_gstate st;
    int rc = giterate_dev(c, n, p, dH, dZ, S, wantT, wantZ, maxitfac, &st, stats, maxlog);
    if (rc != 0) {
        *info = rc;
        return rc;
    }
    if (stats) {
        stats->niter = st.jiter;
        stats->nsweeps = st.nsweeps;
        stats->nrqpass = st.nzshift;
        stats->ndefl1 = st.nsplit;
        stats->ndefl2 = st.n2real + st.n2cplx;
        stats->nwindows = st.nwindows;
        stats->nlog = st.nlog;
        stats->reserved = st.ncase2 + 1000 * st.ncase3;
        stats->maxits = st.ntrainsweeps;  // (signed path: sweeps that ran inside multishift trains)
        for (int q = 0; q < 6; ++q) stats->step_cycles[q] = st.cyc[q];
    }
    PSD_CHECK(psd_rt_d2h(alpha, c->galpha, sizeof(psd_z) * n, c->stream));
    PSD_CHECK(psd_rt_d2h(beta, c->gbeta, sizeof(double) * n, c->stream));
    std::vector<int> hsc(n, 0);
    PSD_CHECK(psd_rt_d2h(hsc.data(), c->gascale, sizeof(int) * n, c->stream));
    const int nl = st.nlog < maxlog ? st.nlog : maxlog;
    std::vector<int> hlog((size_t)3 * nl + 3, 0);
    if (nl > 0) PSD_CHECK(psd_rt_d2h(hlog.data(), c->glog, sizeof(int) * 3 * (size_t)nl, c->stream));
    PSD_CHECK(psd_rt_sync(c->stream));
    for (int q = 0; q < n; ++q) ascale[q] = hsc[q];
    if (stats) {  // algorithmic bytes of the sweeps: two rotations per (j, l), E = 8
        double b = 0.0;
        for (int q = 0; q < nl; ++q)
            if (hlog[3 * q] == 0 || hlog[3 * q] == 4) {
                const double w = hlog[3 * q + 2] - hlog[3 * q + 1] + 1;
                const double rot = (hlog[3 * q] == 0) ? 2.0 : 1.0;
                if (!wantT) b += rot * 2 * 8.0 * p * w * w + (wantZ ? rot * 2 * 8.0 * p * w * n : 0.0);
                else b += rot * 2 * 8.0 * p * w * (wantZ ? (2.0 * n + 1) : (n + 1.0));
            }
        stats->bytes_sweeps = b;
    }
    if (sweeplog) {
        const int64_t m = nl < maxlog_user ? nl : maxlog_user;
        for (int64_t q = 0; q < 3 * m; ++q) sweeplog[q] = hlog[q];
    }
    *info = (st.info == PSD_LIST_OVERFLOW) ? (PSD_INFO_RUNTIME + 77) : ((st.info != 0) ? (PSD_INFO_NOCONV + st.info) : 0);
    return *info;
}

// _phessenberg!(A, S; wantQ) on device — generalized.jl:988-1082.  dA [p][n][n] internal order, overwritten by
// H_1 (Hessenberg), H_l (upper triangular); dQ [p][n][n] or nullptr.
int sghess_dev(psd_ctx* c, int n, int p, double* dA, double* dQ, const uint8_t* S, psd_stats* stats) {
    const size_t nn = (size_t)n * n;
    const size_t lds_refl = PSD_HESS_NT * 8;
    const size_t lds_apply = (PSD_HESS_NT + (size_t)n + 8) * 8;
    auto sg = [&](int l) { return !S || S[l - 1]; };
    if (dQ) PSD_LAUNCH(psd_set_identity, psd_dim3(n, p), 64, 0, c->stream, dQ, n);
    Timer t;
    t.start(c->stream);
    for (int l = p; l >= 2; --l) {  // stage 1 (:1009-1028)
        double* Al = dA + (size_t)(l - 1) * nn;
        double* Am = dA + (size_t)(l - 2) * nn;
        double* Ql = dQ ? dQ + (size_t)(l - 1) * nn : nullptr;
        const bool rq = !sg(l);
        const int mrows = sg(l - 1) ? 0 : 1;  // A_{l-1} takes U from the right (columns) or U' from the left (rows)
        if (rq) {
            PSD_LAUNCH(psd_antitranspose, psd_dim3(n), 256, 0, c->stream, Al, n);
            PSD_LAUNCH(psd_flip, psd_dim3(n), 256, 0, c->stream, Am, n, mrows);
            if (Ql) PSD_LAUNCH(psd_flip, psd_dim3(n), 256, 0, c->stream, Ql, n, 0);
        }
        for (int i = 1; i <= n - 1; ++i) {
            // (reflector i: by its own launch for the first column, behind the panel update of link i - 1 otherwise)
            double* vcur = c->vbuf + (size_t)(i & 1) * (n + 8);
            double* vnext = (i < n - 1) ? (c->vbuf + (size_t)((i + 1) & 1) * (n + 8)) : nullptr;
            if (i == 1) PSD_LAUNCH(psd_hess_refl, psd_dim3(1), PSD_HESS_NT, lds_refl, c->stream, Al, n, i, i, vcur, (double*)nullptr);
            const int nL = (n - i + 3) / 4;   // columns i+1..n of A_l
            const int nR = (n + PSD_HESS_RS - 1) / PSD_HESS_RS;
            // one launch: A_l (+ Q_l) and the neighbour A_{l-1} (different matrices, same reflector)
            const int g1 = nL + (Ql ? nR : 0);
            const int nLm = (n + 3) / 4;
            if (mrows == 0) {
                PSD_LAUNCH(psd_hess_apply2, psd_dim3(g1 + nR), PSD_HESS_NT, lds_apply, c->stream, Al, Ql, i + 1, nL, g1,
                           (double*)nullptr, Am, 1, 0, n, i, (const double*)vcur, vnext);
            } else {
                PSD_LAUNCH(psd_hess_apply2, psd_dim3(g1 + nLm), PSD_HESS_NT, lds_apply, c->stream, Al, Ql, i + 1, nL, g1, Am,
                           (double*)nullptr, 1, nLm, n, i, (const double*)vcur, vnext);
            }
        }
        PSD_LAUNCH(psd_tril_zero, psd_dim3(n), 256, 0, c->stream, Al, n);
        if (rq) {
            PSD_LAUNCH(psd_antitranspose, psd_dim3(n), 256, 0, c->stream, Al, n);
            PSD_LAUNCH(psd_flip, psd_dim3(n), 256, 0, c->stream, Am, n, mrows);
            if (Ql) PSD_LAUNCH(psd_flip, psd_dim3(n), 256, 0, c->stream, Ql, n, 0);
        }
    }
    const double ms1 = t.stop(c->stream);
    t.start(c->stream);
    // stage 2 (:1034-1079): same step/apply machinery as the QZ iteration
    if (int rc = c->greserve(n, p, 16)) return rc;
    psd_gstate st;
    int rc = giterate_dev(c, n, p, dA, dQ, S, 1, dQ ? 1 : 0, 1, &st, nullptr, 16, 1);
    const double ms2 = t.stop(c->stream);
    if (stats) {
        stats->ms_hess = ms1 + ms2;
        stats->ms_formq = ms1;  // stage 1 (QR/RQ sweeps incl. the accumulation of Q)
        // stage 2 moves about n^3 element pairs per factor incl. Q
        stats->bytes_hess = 2.0 * 8.0 * p * (double)n * n * n;
    }
    return rc;
}

}  // namespace

extern "C" {

// _phessenberg!(A, S) — generalized.jl:988-1082 (Float64).  A[l] overwritten by H_l, Q[l] = Qs[l].
int psd_d_gphessenberg(psd_ctx* c, int n, int p, double* const* A, const uint8_t* S, double* const* Q, psd_stats* stats,
                       int* info) {
    int dummy;
    if (!info) info = &dummy;
    if (stats) memset(stats, 0, sizeof(*stats));
    if (!c) return *info = -1;
    if ((*info = check_dims(n, p)) != 0) return *info;
    if (!A) return *info = -4;
    if (S && !S[0]) return *info = -5;  // generalized.jl:990
    if ((*info = c->reserve(n, p, true, 16)) != 0) return *info;
    const size_t nn = (size_t)n * n;
    for (int j = 0; j < p; ++j) PSD_CHECK(psd_rt_h2d(c->dH + j * nn, A[j], nn * 8, c->stream));
    *info = sghess_dev(c, n, p, c->dH, Q ? c->dZ : nullptr, S, stats);
    if (*info != 0) return *info;
    for (int j = 0; j < p; ++j) PSD_CHECK(psd_rt_d2h(A[j], c->dH + j * nn, nn * 8, c->stream));
    if (Q)
        for (int j = 0; j < p; ++j) PSD_CHECK(psd_rt_d2h(Q[j], c->dZ + j * nn, nn * 8, c->stream));
    PSD_CHECK(psd_rt_sync(c->stream));
    return 0;
}

// pschur!(A, S, lr; wantZ, wantT) for Float64 — rgeneralized.jl:3-45.  User-order in/out as psd_d_pschur.
int psd_d_gpschur(psd_ctx* c, int n, int p, double* const* A, const uint8_t* S, char orient, int wantT, int wantZ,
                  int maxitfac, double* const* Z, double* alpha, double* beta, int32_t* ascale, int* schurindex,
                  psd_stats* stats, int* info) {
    int dummy;
    if (!info) info = &dummy;
    if (stats) memset(stats, 0, sizeof(*stats));
    if (!c) return *info = -1;
    if ((*info = check_dims(n, p)) != 0) return *info;
    if (!A) return *info = -4;
    if (orient != 'R' && orient != 'L') return *info = -6;
    if (wantZ && !Z) return *info = -10;
    if (maxitfac < 1) return *info = -9;
    const bool left = orient == 'L';
    auto slotA = [&](int j) { return left ? (p + 1 - j) : j; };               // internal j <- user slot (1-based)
    auto slotZ = [&](int j) { return (!left || j == 1) ? j : (p + 2 - j); };  // rgeneralized.jl:1062-1071
    std::vector<uint8_t> Sarg(p, 1);
    bool alltrue = true;
    for (int j = 1; j <= p; ++j) {
        Sarg[j - 1] = (!S || S[slotA(j) - 1]) ? 1 : 0;
        alltrue = alltrue && Sarg[j - 1];
    }
    if (!Sarg[0]) return *info = -5;  // rgeneralized.jl:37
    if ((*info = c->reserve(n, p, true, 16)) != 0) return *info;
    const size_t nn = (size_t)n * n;
    Timer tc;
    tc.start(c->stream);
    for (int j = 1; j <= p; ++j) PSD_CHECK(psd_rt_h2d(c->dH + (size_t)(j - 1) * nn, A[slotA(j) - 1], nn * 8, c->stream));
    double ms_copy = tc.stop(c->stream);
    psd_stats local;
    memset(&local, 0, sizeof(local));
    psd_stats* s = stats ? stats : &local;
    Timer tt;
    tt.start(c->stream);
    if (alltrue) {  // rgeneralized.jl:21-34: Householder phessenberg!, explicit Q
        Timer t;
        t.start(c->stream);
        if ((*info = hessenberg_dev(c, n, p, c->dH, c->tau)) != 0) return *info;
        s->ms_hess = t.stop(c->stream);
        if (wantZ) {
            t.start(c->stream);
            if ((*info = formq_dev(c, n, p, c->dH, c->tau, c->dZ)) != 0) return *info;
            s->ms_formq = t.stop(c->stream);
        }
        PSD_LAUNCH(psd_triu, psd_dim3(n, p), 64, 0, c->stream, c->dH, n);
    } else {
        if ((*info = sghess_dev(c, n, p, c->dH, wantZ ? c->dZ : nullptr, Sarg.data(), s)) != 0) return *info;
    }
    Timer ti;
    ti.start(c->stream);
    const double keep_hess = s->ms_hess, keep_formq = s->ms_formq, keep_bh = s->bytes_hess;
    int rc = grun_iteration(c, n, p, c->dH, wantZ ? c->dZ : nullptr, Sarg.data(), wantT, wantZ, maxitfac, alpha, beta,
                            ascale, s, nullptr, 0, info);
    s->ms_hess = keep_hess;
    s->ms_formq = keep_formq;
    s->bytes_hess = keep_bh;
    s->ms_iter = ti.stop(c->stream);
    s->ms_total = tt.stop(c->stream);
    if (schurindex) *schurindex = left ? p : 1;
    if (rc < 0 || rc >= PSD_INFO_NOTIMPL) return rc;
    tc.start(c->stream);
    for (int j = 1; j <= p; ++j) PSD_CHECK(psd_rt_d2h(A[slotA(j) - 1], c->dH + (size_t)(j - 1) * nn, nn * 8, c->stream));
    if (wantZ)
        for (int j = 1; j <= p; ++j)
            PSD_CHECK(psd_rt_d2h(Z[slotZ(j) - 1], c->dZ + (size_t)(j - 1) * nn, nn * 8, c->stream));
    PSD_CHECK(psd_rt_sync(c->stream));
    s->ms_copy = ms_copy + tc.stop(c->stream);
    return rc;
}

int psd_d_gpschur_hess(psd_ctx* c, int n, int p, double* const* H, const uint8_t* S, double* const* Q, int wantT,
                       int wantZ, int maxitfac, double* alpha, double* beta, int32_t* ascale, psd_stats* stats,
                       int32_t* sweeplog, int64_t maxlog, int* info) {
    int dummy;
    if (!info) info = &dummy;
    if (stats) memset(stats, 0, sizeof(*stats));
    if (!c) return *info = -1;
    if ((*info = check_dims(n, p)) != 0) return *info;
    if (!H) return *info = -4;
    if (S && !S[0]) return *info = -5;  // rgeneralized.jl:73
    if (wantZ && !Q) return *info = -6;
    if (maxitfac < 1) return *info = -9;
    if ((*info = c->reserve(n, p, true, 16)) != 0) return *info;
    const size_t nn = (size_t)n * n;
    for (int j = 0; j < p; ++j) PSD_CHECK(psd_rt_h2d(c->dH + j * nn, H[j], nn * 8, c->stream));
    if (wantZ)
        for (int j = 0; j < p; ++j) PSD_CHECK(psd_rt_h2d(c->dZ + j * nn, Q[j], nn * 8, c->stream));
    psd_stats local;
    memset(&local, 0, sizeof(local));
    psd_stats* s = stats ? stats : &local;
    Timer t;
    t.start(c->stream);
    int rc = grun_iteration(c, n, p, c->dH, wantZ ? c->dZ : nullptr, S, wantT, wantZ, maxitfac, alpha, beta, ascale, s,
                            sweeplog, maxlog, info);
    s->ms_iter = s->ms_total = t.stop(c->stream);
    if (rc < 0 || rc >= PSD_INFO_NOTIMPL) return rc;
    for (int j = 0; j < p; ++j) PSD_CHECK(psd_rt_d2h(H[j], c->dH + j * nn, nn * 8, c->stream));
    if (wantZ)
        for (int j = 0; j < p; ++j) PSD_CHECK(psd_rt_d2h(Q[j], c->dZ + j * nn, nn * 8, c->stream));
    PSD_CHECK(psd_rt_sync(c->stream));
    return rc;
}

}  // extern "C"


// =================================================================================================
// ordschur! for GeneralizedPeriodicSchur (1x1 swaps)
namespace {

size_t zgord_lds_bytes(int p, int W) {
    size_t b = (size_t)p * W * (W + 1) * 16 + (size_t)15 * p * 16 + ((size_t)p + 2) * 8 + (size_t)p * 4 + 64;
    return (b + 15) & ~(size_t)15;
}

// dH/dZ in the internal right order, S internal signature
int zgordschur_dev(psd_ctx* c, int n, int p, psd_z* dH, psd_z* dZ, const uint8_t* S, const uint8_t* select, int wantZ,
                   double* alpha, double* beta, int32_t* ascale, psd_stats* stats, int* info) {
    int W = 0;
    {
        const int cand[] = {32, 24, 20, 16, 12, 10, 8, 6, 4};
        for (int Wc : cand)
            if (zgord_lds_bytes(p, Wc) <= 155 * 1024) {
                W = Wc;
                break;
            }
    }
    if (W == 0) return *info = PSD_INFO_NOTIMPL;
    const bool pipe = c->ord_pipe != 0;  // pipelined driver (psd_oslot), as zordschur_dev
    if (pipe) {
        int nsel = 0;
        for (int q = 0; q < n; ++q) nsel += select[q] ? 1 : 0;
        if (nsel >= 8 && n >= 128 && W > 16) W = 16;
        PSD_CHECK(c->zgtreserve(p));
    }
    PSD_CHECK(psd_rt_h2d(c->osel, select, (size_t)n, c->stream));
    std::vector<unsigned char> hS(p, 1);
    for (int l = 0; l < p; ++l) hS[l] = S[l] ? 1 : 0;
    PSD_CHECK(psd_rt_h2d(c->zgS, hS.data(), (size_t)p, c->stream));
    PSD_CHECK(psd_rt_sync(c->stream));
    psd_zgoparams O;
    O.z.H = dH;
    O.z.zlo = 1;  // (ordschur! is not sharded: every Z_m is this context's)
    O.z.zhi = p;
    O.z.Z = wantZ ? dZ : nullptr;
    O.z.S = c->zgS;
    O.z.st = c->zgst;
    O.z.desc = c->zgdesc;
    O.z.tr = c->zgtr;
    O.z.cnt = c->zgcnt;
    O.z.dG = c->zgdG;
    O.z.alpha = c->zalpha;
    O.z.beta = c->zbeta;
    O.z.ascale = c->zascale;
    O.z.log = c->zlog;
    O.st = c->ost;
    O.select = c->osel;
    const size_t lds_step = zgord_lds_bytes(p, W);
#ifndef PSD_HOSTSIM
    if (lds_step > c->zgostep_lds_set) {
        PSD_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(psd_zgord_step),
                                      hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_step));
        c->zgostep_lds_set = lds_step;
    }
#endif
    const size_t lds_apply = sizeof(psd_ztr) * PSD_GTR_CAP + (size_t)32 * (PSD_ZAPPLY_NT + 1) * sizeof(psd_z);
    const int tiles = (n + PSD_ZAPPLY_NT - 1) / PSD_ZAPPLY_NT;
    psd_ostate hst;
    memset(&hst, 0, sizeof(hst));
    long long launched = 0;
    const long long cap = (long long)n * ((long long)n / (W > 1 ? W - 1 : 1) + 2) + 1024;
    Timer t;
    t.start(c->stream);
    if (pipe) {
        O.st = c->ombst;
        O.z.desc = c->zgtdesc;
        O.z.cnt = c->zgtcnt;
        O.z.tr = c->zgttr;
#ifndef PSD_HOSTSIM
        if (lds_step > c->zgostep_mb_lds_set) {
            PSD_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(psd_zgord_step_mb),
                                          hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_step));
            c->zgostep_mb_lds_set = lds_step;
        }
#endif
        PSD_CHECK(psd_rt_memset(c->zgtdesc, 0, sizeof(psd_gapply_desc) * PSD_TRAIN_MAX, c->stream));
        PSD_LAUNCH(psd_ord1_init_mb, psd_dim3(1), 64, 0, c->stream, c->ombst, c->oslots, c->omb, n, p, wantZ, W);
        psd_omb hg;
        memset(&hg, 0, sizeof(hg));
        for (;;) {
            for (int b = 0; b < 16; ++b) {
                PSD_LAUNCH(psd_ord1_plan, psd_dim3(1), 64, 0, c->stream, c->ombst, c->oslots, c->omb, c->osel);
                PSD_LAUNCH(psd_zgord_step_mb, psd_dim3(PSD_O_SLOTS), PSD_STEP_NT, lds_step, c->stream, O, p, p + 8);
                PSD_LAUNCH(psd_zgq_apply_train, psd_dim3(tiles, p, 2 * PSD_O_SLOTS), PSD_ZAPPLY_NT, lds_apply, c->stream, O.z, n, p, p + 8, 0);
                PSD_LAUNCH(psd_zgq_apply_train, psd_dim3(tiles, p, PSD_O_SLOTS), PSD_ZAPPLY_NT, lds_apply, c->stream, O.z, n, p, p + 8, 1);
                ++launched;
            }
            PSD_CHECK(psd_rt_d2h(&hg, c->omb, sizeof(hg), c->stream));
            PSD_CHECK(psd_rt_sync(c->stream));
            if (hg.phase == PSD_OPH_DONE) break;
            if (launched > cap) return *info = PSD_INFO_RUNTIME + 0xfffb;
        }
        std::vector<psd_ostate> hs(PSD_O_SLOTS);
        PSD_CHECK(psd_rt_d2h(hs.data(), c->ombst, sizeof(psd_ostate) * PSD_O_SLOTS, c->stream));
        PSD_CHECK(psd_rt_sync(c->stream));
        hst.phase = PSD_OPH_DONE;
        hst.info = hg.info;
        for (const psd_ostate& q : hs) {
            hst.nswaps += q.nswaps;
            hst.nwindows += q.nwindows;
        }
    } else {
    PSD_LAUNCH(psd_zgord_init, psd_dim3(1), 64, 0, c->stream, O, n, p, wantZ, W);
    for (;;) {
        for (int b = 0; b < 32; ++b) {
            PSD_LAUNCH(psd_zgord_step, psd_dim3(1), PSD_STEP_NT, lds_step, c->stream, O);
            PSD_LAUNCH(psd_zgq_apply, psd_dim3(tiles, p, 3), PSD_ZAPPLY_NT, lds_apply, c->stream, O.z, n, p);
            ++launched;
        }
        PSD_CHECK(psd_rt_d2h(&hst, c->ost, sizeof(hst), c->stream));
        PSD_CHECK(psd_rt_sync(c->stream));
        if (hst.phase == PSD_OPH_DONE) break;
        if (launched > cap) return *info = PSD_INFO_RUNTIME + 0xfffb;
    }
    }
    if (hst.info == 0) {
        PSD_LAUNCH(psd_zgord_values, psd_dim3((n + 255) / 256), 256, 0, c->stream, O.z, n, p);
        PSD_CHECK(psd_rt_d2h(alpha, c->zalpha, sizeof(psd_z) * n, c->stream));
        PSD_CHECK(psd_rt_d2h(beta, c->zbeta, sizeof(double) * n, c->stream));
        std::vector<int> hsc(n, 0);
        PSD_CHECK(psd_rt_d2h(hsc.data(), c->zascale, sizeof(int) * n, c->stream));
        PSD_CHECK(psd_rt_sync(c->stream));
        for (int q = 0; q < n; ++q) ascale[q] = hsc[q];
    }
    const double ms = t.stop(c->stream);
    PSD_CHECK(psd_rt_last_error());
    if (stats) {
        stats->ms_iter = stats->ms_total = ms;
        stats->nsweeps = hst.nswaps;
        stats->nwindows = hst.nwindows;
        stats->nlaunch_step = (int32_t)launched;
        stats->window = W;
    }
    return *info = hst.info;
}

// host entry shared by the complex and the (promoted) real case; Tz/Zz: p host matrices of n*n complex
int gordschur_host(psd_ctx* c, int n, int p, std::vector<std::vector<psd_z>>& Tz, std::vector<std::vector<psd_z>>& Zz,
                   const uint8_t* S, char orient, int schurindex, const uint8_t* select, int wantZ, double* alpha,
                   double* beta, int32_t* ascale, psd_stats* stats, int* info) {
    std::vector<int> sA, sZ;
    if (!ord_slots(orient, schurindex, p, sA, sZ)) return *info = -7;  // ArgumentError, ordschur.jl:32
    if ((*info = c->zreserve(n, p, true, 16)) != 0) return *info;
    if ((*info = c->zgreserve(n, p)) != 0) return *info;
    const size_t nn = (size_t)n * n;
    auto slotA = [&](int j) { return sA[j - 1] + 1; };  // internal j <- user slot (1-based)
    auto slotZ = [&](int j) { return sZ[j - 1] + 1; };
    std::vector<uint8_t> Sint(p, 1);
    for (int j = 1; j <= p; ++j) Sint[j - 1] = S[slotA(j) - 1] ? 1 : 0;
    if (!Sint[0]) return *info = -5;
    for (int j = 1; j <= p; ++j)
        PSD_CHECK(psd_rt_h2d(c->zH + (size_t)(j - 1) * nn, Tz[slotA(j) - 1].data(), nn * 16, c->stream));
    if (wantZ)
        for (int j = 1; j <= p; ++j)
            PSD_CHECK(psd_rt_h2d(c->zZ + (size_t)(j - 1) * nn, Zz[slotZ(j) - 1].data(), nn * 16, c->stream));
    int rc = zgordschur_dev(c, n, p, c->zH, c->zZ, Sint.data(), select, wantZ, alpha, beta, ascale, stats, info);
    if (rc < 0 || rc >= PSD_INFO_NOTIMPL) return rc;
    for (int j = 1; j <= p; ++j)
        PSD_CHECK(psd_rt_d2h(Tz[slotA(j) - 1].data(), c->zH + (size_t)(j - 1) * nn, nn * 16, c->stream));
    if (wantZ)
        for (int j = 1; j <= p; ++j)
            PSD_CHECK(psd_rt_d2h(Zz[slotZ(j) - 1].data(), c->zZ + (size_t)(j - 1) * nn, nn * 16, c->stream));
    PSD_CHECK(psd_rt_sync(c->stream));
    return rc;
}

int gord_check_args(psd_ctx* c, int n, int p, const void* T, const void* Z, const uint8_t* S, char orient,
                    const uint8_t* select, int wantZ, int* info) {
    if (!c) return *info = -1;
    if ((*info = check_dims(n, p)) != 0) return *info;
    if (!T) return *info = -4;
    if (wantZ && !Z) return *info = -5;
    if (!S) return *info = -5;
    if (orient != 'R' && orient != 'L') return *info = -6;
    if (!select) return *info = -8;
    return 0;
}

}  // namespace

extern "C" {

int psd_z_gordschur(psd_ctx* c, int n, int p, double* const* T, double* const* Z, const uint8_t* S, char orient,
                    int schurindex, const uint8_t* select, int wantZ, double* alpha, double* beta, int32_t* ascale,
                    psd_stats* stats, int* info) {
    int dummy;
    if (!info) info = &dummy;
    if (stats) memset(stats, 0, sizeof(*stats));
    if (gord_check_args(c, n, p, T, Z, S, orient, select, wantZ, info) != 0) return *info;
    const size_t nn = (size_t)n * n;
    std::vector<std::vector<psd_z>> Tz(p), Zz(wantZ ? p : 0);
    for (int j = 0; j < p; ++j) {
        Tz[j].resize(nn);
        memcpy(Tz[j].data(), T[j], nn * 16);
        if (wantZ) {
            Zz[j].resize(nn);
            memcpy(Zz[j].data(), Z[j], nn * 16);
        }
    }
    int rc = gordschur_host(c, n, p, Tz, Zz, S, orient, schurindex, select, wantZ, alpha, beta, ascale, stats, info);
    if (rc < 0 || rc >= PSD_INFO_NOTIMPL) return rc;
    for (int j = 0; j < p; ++j) {
        memcpy(T[j], Tz[j].data(), nn * 16);
        if (wantZ) memcpy(Z[j], Zz[j].data(), nn * 16);
    }
    return rc;
}

// Float64 with a real spectrum (T1 triangular): promoted to the complex kernel, whose rotations stay real on real data.
// A 2x2 block in T1 (conjugate pair) needs the signed block swap of sylswap.jl:197-538: PSD_INFO_NOTIMPL.
int psd_d_gordschur(psd_ctx* c, int n, int p, double* const* T, double* const* Z, const uint8_t* S, char orient,
                    int schurindex, const uint8_t* select, int wantZ, double* alpha, double* beta, int32_t* ascale,
                    psd_stats* stats, int* info) {
    int dummy;
    if (!info) info = &dummy;
    if (stats) memset(stats, 0, sizeof(*stats));
    if (gord_check_args(c, n, p, T, Z, S, orient, select, wantZ, info) != 0) return *info;
    if (schurindex < 1 || schurindex > p) return *info = -7;
    const size_t nn = (size_t)n * n;
    bool pairs = false;
    {
        const double* T1 = T[schurindex - 1];
        for (int j = 0; j + 1 < n; ++j)
            if (T1[(size_t)j * n + (j + 1)] != 0.0) pairs = true;
    }
    if (pairs) {  // 2x2 blocks: signed block swaps in real arithmetic (sylswap.jl:197-538)
        std::vector<int> sA, sZ;
        if (!ord_slots(orient, schurindex, p, sA, sZ)) return *info = -7;
        if ((*info = c->reserve(n, p, true, 16)) != 0) return *info;
        auto slotA = [&](int j) { return sA[j - 1] + 1; };
        auto slotZ = [&](int j) { return sZ[j - 1] + 1; };
        std::vector<uint8_t> Sint(p, 1);
        for (int j = 1; j <= p; ++j) Sint[j - 1] = S[slotA(j) - 1] ? 1 : 0;
        if (!Sint[0]) return *info = -5;
        for (int j = 1; j <= p; ++j) PSD_CHECK(psd_rt_h2d(c->dH + (size_t)(j - 1) * nn, T[slotA(j) - 1], nn * 8, c->stream));
        if (wantZ)
            for (int j = 1; j <= p; ++j)
                PSD_CHECK(psd_rt_h2d(c->dZ + (size_t)(j - 1) * nn, Z[slotZ(j) - 1], nn * 8, c->stream));
        int rc = rordschur_dev(c, n, p, c->dH, c->dZ, select, wantZ, nullptr, nullptr, stats, info, Sint.data(), alpha, beta,
                               ascale);
        if (rc < 0 || rc >= PSD_INFO_NOTIMPL) return rc;
        for (int j = 1; j <= p; ++j) PSD_CHECK(psd_rt_d2h(T[slotA(j) - 1], c->dH + (size_t)(j - 1) * nn, nn * 8, c->stream));
        if (wantZ)
            for (int j = 1; j <= p; ++j)
                PSD_CHECK(psd_rt_d2h(Z[slotZ(j) - 1], c->dZ + (size_t)(j - 1) * nn, nn * 8, c->stream));
        PSD_CHECK(psd_rt_sync(c->stream));
        return rc;
    }
    std::vector<std::vector<psd_z>> Tz(p), Zz(wantZ ? p : 0);
    for (int j = 0; j < p; ++j) {
        Tz[j].resize(nn);
        for (size_t q = 0; q < nn; ++q) Tz[j][q] = zmk(T[j][q], 0.0);
        if (wantZ) {
            Zz[j].resize(nn);
            for (size_t q = 0; q < nn; ++q) Zz[j][q] = zmk(Z[j][q], 0.0);
        }
    }
    int rc = gordschur_host(c, n, p, Tz, Zz, S, orient, schurindex, select, wantZ, alpha, beta, ascale, stats, info);
    if (rc < 0 || rc >= PSD_INFO_NOTIMPL) return rc;
    for (int j = 0; j < p; ++j) {
        for (size_t q = 0; q < nn; ++q) T[j][q] = Tz[j][q].re;
        if (wantZ)
            for (size_t q = 0; q < nn; ++q) Z[j][q] = Zz[j][q].re;
    }
    return rc;
}

}  // extern "C"

// ------------------------------------------------------------------------------------------------
// _rphessenberg!(Ap, A, Q) — rhessx.jl:55-109
namespace {
template <class O, int ES>
int rphessenberg_host(psd_ctx* c, int m, int n, int p, double* Ap, double* const* A, double* const* Q, int nq, int nqc,
                      int* info) {
    typedef typename O::T T;
    int dummy;
    if (!info) info = &dummy;
    if (!c) return *info = -1;
    if (n < 1 || !(m == n || m == n + 1)) return *info = -2;  // ArgumentError, rhessx.jl:62
    if (p < 1) return *info = -3;
    if (!Ap) return *info = -4;
    if (p > 1 && !A) return *info = -5;
    if (Q && (nq < 1 || nqc < n)) return *info = -8;
    const size_t nap = (size_t)m * n, nn = (size_t)n * n, nqq = (size_t)nq * nqc;
    T *dAp = nullptr, *dA = nullptr, *dQ = nullptr;
    PSD_CHECK(psd_rt_malloc((void**)&dAp, nap * ES));
    PSD_CHECK(psd_rt_malloc((void**)&dA, (p > 1 ? (size_t)(p - 1) * nn : 1) * ES));
    if (Q) PSD_CHECK(psd_rt_malloc((void**)&dQ, (size_t)p * nqq * ES));
    PSD_CHECK(psd_rt_h2d(dAp, Ap, nap * ES, c->stream));
    for (int l = 0; l + 1 < p; ++l) PSD_CHECK(psd_rt_h2d(dA + (size_t)l * nn, A[l], nn * ES, c->stream));
    if (Q)
        for (int l = 0; l < p; ++l) PSD_CHECK(psd_rt_h2d(dQ + (size_t)l * nqq, Q[l], nqq * ES, c->stream));
    const size_t lds = sizeof(double) * PSD_RH_NT + (size_t)(n + 2) * ES;
    PSD_LAUNCH(psd_rphess_kernel<O>, psd_dim3(1), PSD_RH_NT, lds, c->stream, dAp, dA, dQ, m, n, p, nq, nqc);
    PSD_CHECK(psd_rt_d2h(Ap, dAp, nap * ES, c->stream));
    for (int l = 0; l + 1 < p; ++l) PSD_CHECK(psd_rt_d2h(A[l], dA + (size_t)l * nn, nn * ES, c->stream));
    if (Q)
        for (int l = 0; l < p; ++l) PSD_CHECK(psd_rt_d2h(Q[l], dQ + (size_t)l * nqq, nqq * ES, c->stream));
    PSD_CHECK(psd_rt_sync(c->stream));
    PSD_CHECK(psd_rt_last_error());
    psd_rt_free(dAp);
    psd_rt_free(dA);
    if (dQ) psd_rt_free(dQ);
    return *info = 0;
}
}  // namespace

extern "C" {
int psd_d_rphessenberg(psd_ctx* c, int m, int n, int p, double* Ap, double* const* A, double* const* Q, int nq, int nqc,
                       int* info) {
    return rphessenberg_host<psd_rh_real, 8>(c, m, n, p, Ap, A, Q, nq, nqc, info);
}
int psd_z_rphessenberg(psd_ctx* c, int m, int n, int p, double* Ap, double* const* A, double* const* Q, int nq, int nqc,
                       int* info) {
    return rphessenberg_host<psd_rh_cplx, 16>(c, m, n, p, Ap, A, Q, nq, nqc, info);
}
}  // extern "C"


// Diagnostic (tools/apply_bench.py; not part of the drop-in boundary): the bulk-apply kernel alone on synthetic windows.
// nwin windows of width W - 2 positions (two records per position and owner: the shape of a sweep window's lists)
// spread over the diagonal of p factors of order n; which: 1 = rows role, 2 = column role, 4 = Schur vectors (each
// timed on its own, reps launches, HIP events).  ms[3]: milliseconds per launch; bytes[3]: algorithmic bytes per launch.
extern "C" int psd_dbg_apply_bench(psd_ctx* c, int n, int p, int nwin, int W, int which, int reps, double* ms, double* bytes) {
#ifdef PSD_HOSTSIM
    (void)c; (void)n; (void)p; (void)nwin; (void)W; (void)which; (void)reps; (void)ms; (void)bytes;
    return PSD_INFO_NOTIMPL;
#else
    if (!c || nwin < 1 || nwin > PSD_SLOTS || W < 6 || W > 32) return -1;
    PSD_CHECK(c->treserve(p));
    const int nb = W - 4, S = nb + 2;
    std::vector<psd_apply_desc> hd(PSD_SLOTS);
    std::vector<int> hc((size_t)PSD_SLOTS * (p + 8), 0);
    std::vector<psd_tr> ht((size_t)PSD_SLOTS * p * PSD_TR_CAP);
    memset(hd.data(), 0, sizeof(psd_apply_desc) * PSD_SLOTS);
    const int step = (nwin > 1) ? (n - S - 2) / (nwin - 1) : 0;
    if (nwin > 1 && step < S + 2) return -2;
    double br = 0, bc = 0, bz = 0;
    for (int b = 0; b < nwin; ++b) {
        psd_apply_desc& d = hd[b];
        d.active = 1;
        d.prob = 0;
        d.plo = 2 + b * step;
        d.phi = d.plo + S - 1;
        d.lc0 = d.phi + 2; d.lc1 = n;
        d.rr0 = 1; d.rr1 = d.plo - 2;
        d.zr0 = 1; d.zr1 = n;
        d.cut = d.lc1 + 1; d.rcut = d.rr0; d.split = 0;
        br += 16.0 * p * S * (double)(d.lc1 >= d.lc0 ? d.lc1 - d.lc0 + 1 : 0);
        bc += 16.0 * p * S * (double)(d.rr1 >= d.rr0 ? d.rr1 - d.rr0 + 1 : 0);
        bz += 16.0 * p * S * (double)n;
        for (int m = 0; m < p; ++m) {
            int e = 0;
            psd_tr* L = ht.data() + ((size_t)b * p + m) * PSD_TR_CAP;
            const double v2 = 0.1, v3 = 0.05;
            for (int k = 0; k < nb; ++k) {
                L[e].pos = d.plo + k; L[e].kind = PSD_TR_R3; L[e].c0 = v2; L[e].c1 = v3; L[e].c2 = 2.0 / (1.0 + v2 * v2 + v3 * v3); ++e;
                L[e].pos = d.plo + k + 1; L[e].kind = PSD_TR_H2; L[e].c0 = v2; L[e].c1 = 0.0; L[e].c2 = 2.0 / (1.0 + v2 * v2); ++e;
            }
            hc[(size_t)b * (p + 8) + m] = e;
        }
    }
    double *dH = nullptr, *dZ = nullptr;
    const size_t bytesM = (size_t)p * n * n * sizeof(double);
    PSD_CHECK(psd_rt_malloc((void**)&dH, bytesM));
    PSD_CHECK(psd_rt_malloc((void**)&dZ, bytesM));
    PSD_CHECK(psd_rt_memset(dH, 0x3f, bytesM, c->stream));  // (every double 4.8e-4: orthogonal updates keep it bounded)
    PSD_CHECK(psd_rt_memset(dZ, 0x3f, bytesM, c->stream));
    PSD_CHECK(psd_rt_h2d(c->tdesc, hd.data(), sizeof(psd_apply_desc) * PSD_SLOTS, c->stream));
    PSD_CHECK(psd_rt_h2d(c->tcnt, hc.data(), sizeof(int) * hc.size(), c->stream));
    PSD_CHECK(psd_rt_h2d(c->ttr, ht.data(), sizeof(psd_tr) * ht.size(), c->stream));
    psd_rparams P;
    memset(&P, 0, sizeof(P));
    P.H = dH; P.Z = dZ; P.desc = c->tdesc; P.cnt = c->tcnt; P.tr = c->ttr; P.nprob = 1;
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0);
    (void)hipEventCreate(&e1);
    const int cfg[3][2] = {{0, 3}, {1, 3}, {0, 4}};  // (pass, mode): rows role, column role, Schur vectors
    const double by[3] = {br, bc, bz};
    for (int q = 0; q < 3; ++q) {
        ms[q] = 0.0;
        bytes[q] = by[q];
        if (!((which >> q) & 1)) continue;
        for (int r = 0; r < 2; ++r) PSD_CHECK(launch_apply_wl(c, c->stream, P, n, p, cfg[q][0], PSD_SLOTS, 1, p, cfg[q][1], W, c->apply_wl_grid));
        (void)hipEventRecord(e0, c->stream);
        for (int r = 0; r < reps; ++r) PSD_CHECK(launch_apply_wl(c, c->stream, P, n, p, cfg[q][0], PSD_SLOTS, 1, p, cfg[q][1], W, c->apply_wl_grid));
        (void)hipEventRecord(e1, c->stream);
        PSD_CHECK(hipEventSynchronize(e1));
        float t = 0;
        (void)hipEventElapsedTime(&t, e0, e1);
        ms[q] = (double)t / reps;
    }
    (void)hipEventDestroy(e0);
    (void)hipEventDestroy(e1);
    psd_rt_free(dH);
    psd_rt_free(dZ);
    PSD_CHECK(psd_rt_last_error());
    return 0;
#endif
}

#include "psd_check_host.inl"
