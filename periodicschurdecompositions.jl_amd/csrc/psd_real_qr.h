// Real periodic QR iteration on the GPU: device-resident state machine.
//
// Replaces pschur!(H1, Hs; wantT, wantZ, Q, maxitfac) — /root/reference/src/
// PeriodicSchurDecompositions.jl:322-1096 (MB03WD-type double-shift periodic QR).
//
// MI355X design (not a translation of the reference's loop nest):
//   * The reference applies every 2-/3-wide reflector immediately to full rows/columns.  Here the
//     serial dependency chain (w*p reflector generations per sweep) runs inside ONE wavefront on
//     LDS-resident *diagonal window blocks* of all p factors (`psd_rq_step`), emitting the
//     reflectors as per-factor transform lists; the O(n) off-window updates of H_m (rows),
//     H_{m-1} (columns) and Z_m (columns) are applied afterwards at bandwidth by a wide kernel
//     (`psd_rq_apply`), every element read once and written once per window.
//   * All control (deflation tests, shifts, window bookkeeping) lives in device memory, so the
//     host only enqueues {step, apply} pairs and polls a flag every few dozen launches.
//   * Ownership convention for a transform generated "at factor m": left on rows of H_m, right on
//     columns of H_{m-1 (cyclic)} and of Z_m.
#pragma once
#include "psd_scalar.h"
#include "psd_hqr.h"

enum {
    PSD_PH_DECIDE = 0,
    PSD_PH_RQ = 1,
    PSD_PH_SHIFT = 2,
    PSD_PH_QR = 3,
    PSD_PH_DEFLATE = 4,
    PSD_PH_NEXT = 5,
    PSD_PH_FINAL = 6,
    PSD_PH_DONE = 7,
    // multishift trains (on by default: up to 32 bulges in the single-range mode, 64 cursor slots and longer trains under
    // the multi-block scheduler; psd_set_train / PSD_TRAIN): the leader waits for the cursors
    // behind it (TWAIT); a cursor waits for its start (CWAIT), runs QR windows, and ends in CDONE
    PSD_PH_TWAIT = 8,
    PSD_PH_CWAIT = 9,
    PSD_PH_CDONE = 10
};
enum { PSD_TR_R3 = 3, PSD_TR_H2 = 2, PSD_TR_R2 = 4, PSD_TR_G = 5 };

#define PSD_TR_CAP 64     // transform-list capacity per owner and window
#define PSD_STEP_NT 64    // the chase runs in one wavefront
#define PSD_APPLY_NT 128  // threads (= tile rows / tile columns) of the bulk-apply kernel
#define PSD_TRAIN_MAX 64  // bulges (cursors) of a multishift train
#define PSD_SLOTS 64      // workgroup slots of the multi-block scheduler (leaders of independent active blocks + cursors)
enum { PSD_ROLE_FREE = 0, PSD_ROLE_CLAIMED = 1, PSD_ROLE_LEADER = 2, PSD_ROLE_CURSOR = 3 };
#define PSD_EPOCH_NEVER 0x7fffffff

struct psd_tr {
    int pos;   // first row/column index (1-based) the transform acts on
    int kind;  // PSD_TR_*
    double c0, c1, c2;
};

// transform acting on up to three values (identical formula from the left on a column and from
// the right on a row, because everything is real)
PSD_HD void psd_tr_apply(const psd_tr& t, double& a1, double& a2, double& a3) {
    if (t.kind == PSD_TR_R3) {  // householder.jl:207-237 with v = (1, c0, c1), tau = c2
        const double x = t.c2 * (a1 + t.c0 * a2 + t.c1 * a3);
        a1 -= x;
        a2 -= x * t.c0;
        a3 -= x * t.c1;
    } else if (t.kind == PSD_TR_H2) {  // v = (1, c0), tau = c2
        const double x = t.c2 * (a1 + t.c0 * a2);
        a1 -= x;
        a2 -= x * t.c0;
    } else if (t.kind == PSD_TR_R2) {  // householder.jl:281-304 HH2(v1 = c0, v2 = c1, tau = c2)
        const double s = a1 * t.c0 + a2 * t.c1;
        a1 -= s * (t.c2 * t.c0);
        a2 -= s * (t.c2 * t.c1);
    } else {  // Givens: lmul!(G, .) on rows == rmul!(., G') on columns
        const double b1 = t.c0 * a1 + t.c1 * a2;
        const double b2 = -t.c1 * a1 + t.c0 * a2;
        a1 = b1;
        a2 = b2;
    }
}
PSD_HD int psd_tr_len(const psd_tr& t) { return t.kind == PSD_TR_R3 ? 3 : 2; }
// a record to a list in device memory through a pointer that was read from a structure (generic to the compiler: a
// plain assignment would be FLAT stores, which count against the LDS counter the chain waits on)
PSD_D void psd_tr_store_global(psd_tr* dst, const psd_tr& tr) {
#ifdef PSD_HOSTSIM
    *dst = tr;
#else
    __attribute__((address_space(1))) double* q = (__attribute__((address_space(1))) double*)dst;
    q[0] = __hiloint2double(tr.kind, tr.pos);  // (pos, kind: the first eight bytes)
    q[1] = tr.c0;
    q[2] = tr.c1;
    q[3] = tr.c2;
#endif
}

struct psd_apply_desc {
    int active;
    int prob;      // batch: problem the window belongs to
    int cut;       // first far column of its rows role (columns lc0 .. cut - 1 are near: psd_rdefer_edge)
    int rcut;      // first near row of its column role (rows rr0 .. rcut - 1 are far)
    int split;     // 1: a window in the middle of a sweep: the far part of its bulk update (rows role beyond `cut`, column
                   // role above `rcut`, the whole Z role) may run while the next tick chases (psd_rq_apply_wl modes)
    int plo, phi;  // span of positions touched by the lists
    int lc0, lc1;  // left role: columns of H_m
    int rr0, rr1;  // right role: rows of H_{m-1}
    int zr0, zr1;  // Z role: rows of Z_m (empty when !wantZ)
};

struct psd_rstate {
    int n, p, wantT, wantZ, W;  // W: window width of the running sweep (<= Wmax, the width the LDS is laid out for)
    int Wmax;
    int train_oc;  // o / c of the window-width rule in psd_rq_shift
    int phase, info;
    int i, l, its, maxitleft;
    int i1, i2;
    int kcur;
    int maxits;
    long long niter;
    int nsweeps, nrqpass, ndefl1, ndefl2, nwindows, nlog, maxlog;
    double v[3];
    double smlnum, ulp, ulpx;
    // in-kernel cycle accounting (s_memtime): 0 decide, 1 window load, 2 chase, 3 window store, 4 total; 5 = total in
    // 100 MHz wall ticks (s_memrealtime) so that the shader clock can be derived
    long long cyc[6];
    // multishift train: bulges wanted / in the running train / train number / this state's cursor
    int train_want, train_n, train_id, cursor;
    int train_tick0;  // tick (launch index) of the leader's first window of the running train
    int ntrains, ntrainsweeps;
    int exc_dec;  // its / 10 at the last exceptional shift of the current block (a train advances its by several)
    // multi-block scheduler: this state's slot, the leader a cursor belongs to, the lower end of
    // the range this leader owns (the reference works bottom-up through ONE range 1..n, PSD.jl:1057-1060; here the part
    // above a negligible subdiagonal is handed to another workgroup as soon as it is found), the running train's key
    int mb, slot, parent, lo, train_key;
    int prob;  // batch: the problem (its own factors, Schur vectors, band arrays, eigenvalues) this range belongs to
    // ticks between the starts of consecutive cursors of a train: 2 = two whole windows apart; 1 = one tick apart, the
    // cursor's first window 4 positions short, so that its rows end above the window of the cursor ahead (cursors are
    // then nb + 4 positions apart instead of 2 nb: more bulges fit a block, the train fills and drains faster).
    // cgap: what the context asks for; tgap: what the running train uses (1 needs windows of >= 6 positions).
    // All cursors advance nb positions per tick, so the spacing s (2 nb, or nb + 4) holds for the whole sweep if cursor
    // b sits at K_0(t) - b s: it enters the block in the first tick in which that window reaches past l (cstart), with
    // the part of the window that lies inside the block (cfirst positions).
    int cgap, tgap, cstart, cfirst;
    int cslots[PSD_TRAIN_MAX];  // slots of the running train's cursors (entry 0 unused)
    // Long trains (multi-block): a train may have more bulges than cursor slots — a slot that has finished bulge b goes
    // on as bulge b + train_S (same schedule formula), so the pipeline fills and drains once per train_n sweeps instead of
    // once per slot count.  train_S: cursor slots of the train; train_ms: distinct shift pairs (bulge b uses pair
    // b mod train_ms); train_long: bulges wanted per train (psd_rq_init).
    int train_S, train_ms, train_long, train_wdiv;  // (train_wdiv: a long train has at most w / train_wdiv bulges)
    // slots of this leader's share of the tick's slot plan (psd_rq_plan) already taken in launch plan_tick
    int plan_tick, plan_used;
    // deferred column roles (psd_rparams::cdefer): tick in which a decision of this range found that it needs
    // opnorm(H_1[l:i, l:i]) and put itself off by one launch (see psd_rq_decide)
    int opn_tick;
};

// global words of the multi-block scheduler
struct psd_rglobal {
    int done, info, abort;
    int nactive;    // leaders alive
    int nlog;       // entries of the shared sweep log
    int train_seq;  // train keys
    int itbudget;   // (unused since the budget became per problem: pbudget)
    // batch (psd_d_pschur_hess_batch): leaders alive, sweep budget (the reference's maxitleft, PSD.jl:471,1057: ONE
    // budget of maxitfac * n sweeps for all ranges of a problem) and info per problem
    int pactive[PSD_SLOTS], pbudget[PSD_SLOTS], pinfo[PSD_SLOTS];
    int nsweeps, nrqpass, ndefl1, ndefl2, nwindows, ntrains, ntrainsweeps, maxits, nspawn, nslotmax;
    long long niter;
    long long cyc[6];
    // diagnostics (PSD_TICKLOG): shader cycles and calls of the phases of a launch that are not window work:
    // [0] decide, [1] split / spawn, [2] train shifts: staging + product, [3] small QR, [4] slot claims,
    // [5] cursor states, [6] deflate, [7] RQ window; dbgn: calls
    long long dbg[8];
    int dbgn[8];
    // diagnostics (PSD_TICKLOG): shader cycles of the scan chase (psd_chase3.h), wavefront 0, summed over all positions:
    // [0] loads + scan 1, [1] reflectors + B blocks, [2] scan 2, [3] 2-reflectors + table + records, [4] barrier wait,
    // [5] apply phase 0, [6] apply phase 1, [7] positions
    long long c3dbg[8];
};

struct psd_rparams {
    double* H;  // [p][n][n] column-major blocks, internal order (H_1 Hessenberg)
    double* Z;  // [p][n][n] or nullptr
    psd_rstate* st;
    psd_apply_desc* desc;
    psd_tr* tr;     // [p][PSD_TR_CAP]
    int* cnt;       // [p]
    double* hdiag;  // [n+2], 1-based
    double* hsub;
    double* hsup;
    double* Pd;     // [n+3] band of prod_{j>=2} H_j: diagonal, first and second superdiagonal
    double* Pe;
    double* Pf;
    double* hnorms;  // [p+1], 1-based
    double* wr;      // [n] eigenvalues (0-based)
    double* wi;
    int* log;  // [3*maxlog]
    psd_rstate* cst;   // [PSD_TRAIN_MAX] cursor states of a train (entry 0 unused) or nullptr
    double* tshift;    // [PSD_TRAIN_MAX][4] shift pairs of the train: rt1r, rt1i, rt2r, rt2i
    psd_rstate* lead;  // the main state (== st except in a cursor's parameter block)
    int tick;          // launch index (the driver counts ticks; cursors start at fixed tick offsets)
    // multi-block scheduler (nullptr / 0 otherwise): slot s has state cst[s], descriptor desc[s], lists tr + s p CAP,
    // shift pairs tshift + s PSD_TSHIFT_STRIDE
    psd_rglobal* gl;
    int* cep;    // single-range train mode: [PSD_TRAIN_MAX] epoch words of the cursor states (psd_pub_*), then the count
                 // of finished cursors
    int* role;   // [PSD_SLOTS] PSD_ROLE_*
    int* epoch;  // [PSD_SLOTS] tick at which a claimed slot's state was written (PSD_EPOCH_NEVER while free)
    int nprob;   // batch: problems in this call; problem q's arrays start q * (stride) behind problem 0's: factors and
                 // Schur vectors p n n, band arrays / eigenvalues n + 8, hnorms p + 8
    int* cdone;  // [PSD_SLOTS] per LEADER slot: cursors of its running train that have finished (a cursor's own slot
                 // may be reused by another train before the leader looks)
    // diagnostics (nullptr by default, PSD_TICKLOG): per tick the longest workgroup of the chase launch,
    // (100 MHz ticks << 12) | bit mask of the phases it ran (1 << PSD_PH_*)
    int* ticklog;
    int ticklog_n;
    // multi-block: [3][PSD_SLOTS] tick, first and last row of the product band psd_rq_band computed for a slot's
    // pending decision (nullptr otherwise)
    int* bandinfo;
    // multi-block, long trains: [2][PSD_SLOTS] per LEADER slot — 1 once a bulge of its running train has found the
    // bottom of the range converged (no further bulge of the train enters), and the number of bulges cancelled that way
    int* ccancel;
    // multi-block: the slot plan of the tick, written by psd_rq_plan in front of every chase launch from the slot words
    // as the previous launch left them (see PSD_PLAN_*): which free slots each leader may claim in this launch, and
    // the snapshots of cdone / ccancel the workgroups of the launch read.  Nothing a workgroup decides in a launch
    // depends on what another workgroup of the SAME launch has done so far: the schedule is reproducible run to run.
    int* plan;
    // byte offset of the two-wave chase's command block in dynamic LDS (the reduction scratch behind the window image,
    // idle while a window is chased); 0: the chase kernels run single wavefronts (psd_qr_micro3)
    int c2off;
    // scan chase (psd_chase3.h): byte offset of its reflector table in dynamic LDS (behind the scratch of the state
    // machine); 0: off.  The chase workgroups then have PSD_C3_WAVES wavefronts, and c2off names the command block.
    int c3off;
    // factor-sliced scan chase (psd_slice3.h): slices per slot (1: off); per slot PSD_SL_CMD_BYTES of command block and
    // slG inboxes of PSD_SL_BOX_BYTES in device memory; the error word of its bounded waits
    int slG;
    unsigned char* slmem;
    int* slerr;
    // 1: the far part of a tick's column roles (rows more than PSD_CDEFER_EDGE above the window, psd_apply_desc::rcut) runs
    // on the second stream beside the NEXT tick's chases (iterate_dev; psd_rq_apply_wl modes 5 / 6)
    int cdefer;
    int redge, cedge;  // psd_rdefer_edge / psd_cdefer_edge when > 0 (tuning: more of a role's lines in its near part)
};
// Deferred column roles.  The column role of a window (owner m's transformations on columns plo..phi of H_{m-1}) reaches
// from the top of the matrix down to the window.  What the next launch of chases, bands, shift blocks and decisions reads
// of it lies within a window's width of the diagonal (diagonal blocks of order <= W, the band of the product within two of
// the diagonal, trailing blocks of order <= 16 for the shifts), so rows more than max(W + 2, 16) above the window can wait
// until the next tick's ROWS roles need them (they cross these columns): that part runs on the second stream while the
// next tick chases.  Every element still sees the same sequence of operations (rows roles of tick t, column roles of tick
// t, rows roles of tick t + 1, ...), so the results are the same bits as with everything on one stream.  The one reader
// of far entries is psd_h1_opnorm (a fallback of the deflation tests when a diagonal pair of the product is exactly zero):
// a decision that needs it waits one launch (psd_rstate::opn_tick).
PSD_HD int psd_cdefer_edge(int Wmax) { return (Wmax + 2 > 16) ? (Wmax + 2) : 16; }
// The same for the rows roles (psd_rparams::rdefer): the columns of a window's rows role that lie psd_rdefer_edge or more
// to the right of the window.  Until that bulge's next window but one nothing reads them: a chase reads diagonal window
// blocks (at most Wmax columns past the window's end), a decision the product band and the trailing block of the shifts
// (at most 16 off the diagonal), and the near rows of any other window's column role (psd_cdefer_edge above that window)
// can only meet these rows in columns less than Wmax + psd_cdefer_edge past this window.  The far rows of other windows'
// column roles DO cross them, which is why both far parts run on one stream in the order rows, columns.
PSD_HD int psd_rdefer_edge(int Wmax) { return (2 * Wmax + 8 > 48) ? (2 * Wmax + 8) : 48; }
// layout of psd_rparams::plan (ints): share of leader slot s = free slots PSD_PLAN_FREE[first[s] .. first[s] + count[s])
#define PSD_PLAN_FIRST 0
#define PSD_PLAN_COUNT (PSD_SLOTS)
#define PSD_PLAN_FREE (2 * PSD_SLOTS)
#define PSD_PLAN_CDONE (3 * PSD_SLOTS)
#define PSD_PLAN_CCANCEL (4 * PSD_SLOTS)
#define PSD_PLAN_TICK (5 * PSD_SLOTS)
#define PSD_PLAN_INTS (5 * PSD_SLOTS + 8)
#define PSD_TSHIFT_STRIDE (4 * PSD_TRAIN_MAX + 8)
#define PSD_TRAIN_LONG_MINW 64  // narrower ranges keep one bulge per slot (a long train there is sweeps past convergence)
#define PSD_DECIDE_YIELD 128  // active-range width from which the band of a decision comes from psd_rq_band (multi-block mode)

// diagnostics: add shader cycles since t0 to counter k of psd_rglobal (multi-block mode with a tick log only)
#define PSD_DBG_T0() const long long _dbg0 = (P.ticklog != nullptr) ? psd_clock() : 0
#define PSD_DBG_ADD(k)                                                        \
    do {                                                                      \
        if (P.ticklog != nullptr && P.gl != nullptr) {                        \
            PSD_ONE {                                                         \
                psd_atomic_add_ll(&P.gl->dbg[k], psd_clock() - _dbg0);        \
                psd_atomic_add(&P.gl->dbgn[k], 1);                            \
            }                                                                 \
        }                                                                     \
    } while (0)
PSD_HD psd_mat<double> psd_fac(const psd_rparams& P, int n, int j) {
    return psd_mat<double>{P.H + (size_t)(j - 1) * n * n, n};
}

// ------------------------------------------------------------------------------------------------
// LDS window: blocks [bs..be]^2 of all p factors, column-major with leading dimension W+1
// Column pitch of a window image of width W inside an area laid out for Wmax x (Wmax + 1) per factor: W + 1, or W + 2
// when that makes it even and still fits — an even pitch is what the LDS-DMA window load needs (psd_win_load).
// (the area itself is laid out for an even pitch at Wmax: Wmax x (Wmax + 1) or Wmax x (Wmax + 2) doubles per factor)
PSD_HD int psd_win_area(int Wmax) { return Wmax * ((((Wmax + 1) & 1) == 0) ? (Wmax + 1) : (Wmax + 2)); }
PSD_HD int psd_win_pitch(int W, int Wmax) {
    if (((W + 1) & 1) == 0) return W + 1;
    return (W * (W + 2) <= psd_win_area(Wmax)) ? (W + 2) : (W + 1);
}
struct psd_win {
    double* b;
    int W, ld, bsz, bs, be;
    PSD_HD double& at(int j, int r, int c) const { return b[(j - 1) * bsz + (c - bs) * ld + (r - bs)]; }
};

// Window <-> HBM: 64 lanes, lane = (row pair, column group of 4); a lane moves two consecutive rows of a column as
// one 16-byte access (8-byte aligned: global_load/store_dwordx4 accept that), two factors per batch = 16 accesses of
// 16 bytes in flight per lane.  A single wavefront is bounded by the number of outstanding accesses, not by bandwidth,
// so the bytes per access are what counts.  The odd last row travels with the row above it (no access beyond the
// window's rows, which at the bottom of the matrix would leave the allocation).
// Entries more than PSD_WIN_BAND below the diagonal of a window block are structural zeros that no window pass reads or
// writes (Hessenberg / triangular factors plus a 3x3 bulge): they are neither loaded nor stored.  (The simulated tier
// poisons LDS with NaNs, so a read outside the band would surface there.)
#define PSD_WIN_BAND 3
#define PSD_WIN_LF 4  // factors whose window loads are in flight together (16-byte loads, 8 per factor and lane)
struct alignas(8) psd_pair {
    double a, b;
};
#ifdef PSD_HOSTSIM
PSD_D psd_pair psd_pair_load(const double* q) { return *(const psd_pair*)q; }
PSD_D void psd_pair_store(double* q, const psd_pair& x) { *(psd_pair*)q = x; }
#else
typedef double psd_v2u __attribute__((ext_vector_type(2), aligned(8)));
PSD_D psd_pair psd_pair_load(const double* q) {
    const psd_v2u v = *(const psd_v2u*)q;
    psd_pair x;
    x.a = v.x;
    x.b = v.y;
    return x;
}
PSD_D void psd_pair_store(double* q, const psd_pair& x) {
    psd_v2u v;
    v.x = x.a;
    v.y = x.b;
    *(psd_v2u*)q = v;
}
#endif
// (j0, jstep: the factors j0, j0 + jstep, ... only — the wavefronts of a scan-chase workgroup share the window)
PSD_D void psd_win_load(const psd_rparams& P, const psd_win& w, int n, int p, int j0 = 0, int jstep = 1) {
    const int m = w.be - w.bs + 1;
#ifndef PSD_HOSTSIM
    if ((w.ld & 1) == 0 && w.ld <= 64) {
        // LDS-DMA (global_load_lds_dwordx4): a lane's 16 bytes — two consecutive rows of a column — go straight to LDS
        // at wave-uniform base + 16 lane, no VGPRs, so every load of the window can be in flight at once (the register
        // path holds 32 per lane and is bounded by that: 42 us of a 376 us window at p = 64).  A column of the window
        // image is ld = W + 1 doubles = ld / 2 lanes, so one instruction fills 64 / (ld / 2) whole columns; needs ld even
        // (W = 17 at p = 64).  The odd last row at the bottom edge of the matrix (its pair would leave the allocation)
        // comes by an ordinary load.
        const int lpc = w.ld >> 1, cpi = 64 / lpc;
        const int lane = PSD_TID;
        const int cl = lane / lpc, r = 2 * (lane - cl * lpc);
        const bool on = cl < cpi && r < m;
        const bool pair_ok = (r + 1 < m) || (w.bs - 1 + r + 1 < n);  // (the second row exists in the matrix)
        const size_t fstride = (size_t)n * n;
        for (int c0 = 0; c0 < m; c0 += cpi) {  // (column group outside, factors inside: the inner loop is a pointer step and the load)
            const int c = c0 + cl;
            const bool act = on && c < m && c + PSD_WIN_BAND >= r;
            const double* q = P.H + (size_t)(w.bs - 1 + (act ? c : 0)) * n + (w.bs - 1 + (act ? r : 0)) + (size_t)j0 * fstride;
            double* dst = w.b + c0 * w.ld + j0 * w.bsz;
            if (act && pair_ok) {
                for (int j = j0; j < p; j += jstep) {
                    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)q,
                                                     (__attribute__((address_space(3))) void*)dst, 16, 0, 0);
                    q += fstride * jstep;
                    dst += w.bsz * jstep;
                }
            } else if (act) {
                for (int j = j0; j < p; j += jstep) w.b[j * w.bsz + c * w.ld + r] = q[(size_t)(j - j0) * fstride];
            }
        }
        PSD_SYNC();
        return;
    }
#endif
    PSD_PAR_FOR(t, PSD_STEP_NT) {
        const int r = 2 * (t & 15), g = t >> 4;
        if (r < m) {
            const bool pair = r + 1 < m;
            const int back = (pair || r == 0) ? 0 : 1;  // (m == 1: the single element twice, second copy dropped)
            const bool one = !pair && r == 0;
            for (int j = j0 * PSD_WIN_LF; j < p; j += jstep * PSD_WIN_LF) {
                const double* src = P.H + (size_t)j * n * n + (size_t)(w.bs - 1) * n + (w.bs - 1 + r - back);
                double* dst = w.b + j * w.bsz + r;
                psd_pair v[PSD_WIN_LF][8];
#pragma unroll
                for (int f = 0; f < PSD_WIN_LF; ++f) {
#pragma unroll
                    for (int u = 0; u < 8; ++u) {
                        const int c = g + 4 * u;
                        const double* q = src + (size_t)f * n * n + (size_t)c * n;
                        psd_pair x;
                        x.a = x.b = 0.0;
                        if (c < m && c + PSD_WIN_BAND >= r && j + f < p) {
                            if (one) x.a = q[0];
                            else x = psd_pair_load(q);
                        }
                        v[f][u] = x;
                    }
                }
#pragma unroll
                for (int f = 0; f < PSD_WIN_LF; ++f) {
#pragma unroll
                    for (int u = 0; u < 8; ++u) {
                        const int c = g + 4 * u;
                        if (c < m && c + PSD_WIN_BAND >= r && j + f < p) {
                            double* q = dst + f * w.bsz + c * w.ld;
                            q[0] = back ? v[f][u].b : v[f][u].a;
                            if (pair) q[1] = v[f][u].b;
                        }
                    }
                }
            }
        }
    }
    PSD_SYNC();
}
PSD_D void psd_win_store(const psd_rparams& P, const psd_win& w, int n, int p, int j0 = 0, int jstep = 1) {
    const int m = w.be - w.bs + 1;
    PSD_SYNC();
    PSD_PAR_FOR(t, PSD_STEP_NT) {
        const int r = 2 * (t & 15), g = t >> 4;
        if (r < m) {
            const bool pair = r + 1 < m;
            for (int j = j0; j < p; j += jstep) {
                double* dst = P.H + (size_t)j * n * n + (size_t)(w.bs - 1) * n + (w.bs - 1 + r);
                const double* src = w.b + j * w.bsz + r;
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    const int c = g + 4 * u;
                    if (c < m && c + PSD_WIN_BAND >= r) {
                        if (pair) {
                            psd_pair x;
                            x.a = src[c * w.ld];
                            x.b = src[c * w.ld + 1];
                            psd_pair_store(dst + (size_t)c * n, x);
                        } else {
                            dst[(size_t)c * n] = src[c * w.ld];
                        }
                    }
                }
            }
        }
    }
    PSD_SYNC();
}

// in-window application: from the left to rows tr.pos.. of factor jl over columns [c0,c1], and
// from the right to columns tr.pos.. of factor jr over rows [r0,r1] (ranges clipped to the window)
PSD_D void psd_win_apply(const psd_win& w, int jl, int jr, const psd_tr& tr, int c0, int c1, int r0, int r1) {
    if (c0 < w.bs) c0 = w.bs;
    if (c1 > w.be) c1 = w.be;
    if (r0 < w.bs) r0 = w.bs;
    if (r1 > w.be) r1 = w.be;
    const int nl = (jl > 0 && c1 >= c0) ? (c1 - c0 + 1) : 0;
    const int nr = (jr > 0 && r1 >= r0) ? (r1 - r0 + 1) : 0;
    const int len = psd_tr_len(tr);
    const int q = tr.pos;
    if (jl == jr) {  // p == 1: same matrix, left then right
        PSD_PAR_FOR(t, nl) {
            const int c = c0 + t;
            double a1 = w.at(jl, q, c), a2 = w.at(jl, q + 1, c), a3 = (len == 3) ? w.at(jl, q + 2, c) : 0.0;
            psd_tr_apply(tr, a1, a2, a3);
            w.at(jl, q, c) = a1;
            w.at(jl, q + 1, c) = a2;
            if (len == 3) w.at(jl, q + 2, c) = a3;
        }
        PSD_SYNC();
        PSD_PAR_FOR(t, nr) {
            const int r = r0 + t;
            double a1 = w.at(jr, r, q), a2 = w.at(jr, r, q + 1), a3 = (len == 3) ? w.at(jr, r, q + 2) : 0.0;
            psd_tr_apply(tr, a1, a2, a3);
            w.at(jr, r, q) = a1;
            w.at(jr, r, q + 1) = a2;
            if (len == 3) w.at(jr, r, q + 2) = a3;
        }
        PSD_SYNC();
    } else {
        PSD_PAR_FOR(t, nl + nr) {
            if (t < nl) {
                const int c = c0 + t;
                double a1 = w.at(jl, q, c), a2 = w.at(jl, q + 1, c), a3 = (len == 3) ? w.at(jl, q + 2, c) : 0.0;
                psd_tr_apply(tr, a1, a2, a3);
                w.at(jl, q, c) = a1;
                w.at(jl, q + 1, c) = a2;
                if (len == 3) w.at(jl, q + 2, c) = a3;
            } else {
                const int r = r0 + (t - nl);
                double a1 = w.at(jr, r, q), a2 = w.at(jr, r, q + 1), a3 = (len == 3) ? w.at(jr, r, q + 2) : 0.0;
                psd_tr_apply(tr, a1, a2, a3);
                w.at(jr, r, q) = a1;
                w.at(jr, r, q + 1) = a2;
                if (len == 3) w.at(jr, r, q + 2) = a3;
            }
        }
        PSD_SYNC();
    }
}

// append a transform to owner m's list (LDS counters, global list)
PSD_D void psd_record(const psd_rparams& P, int* lcnt, int m, const psd_tr& tr) {
    PSD_ONE {
        const int q = lcnt[m - 1];
        if (q < PSD_TR_CAP) P.tr[(size_t)(m - 1) * PSD_TR_CAP + q] = tr;
        lcnt[m - 1] = q + 1;
    }
}

PSD_D void psd_log(const psd_rparams& P, psd_rstate& st, int kind, int l, int i) {
    PSD_ONE {
        // (multi-block mode: several leaders share the log, entries land in the order the leaders reach them)
        const int at = P.gl ? psd_atomic_add(&P.gl->nlog, 1) : st.nlog;
        if (at < st.maxlog) {
            P.log[3 * at + 0] = kind;
            P.log[3 * at + 1] = l;
            P.log[3 * at + 2] = i;
        }
    }
    st.nlog += 1;
}

// ------------------------------------------------------------------------------------------------
// Multi-block scheduler.  PSD_SLOTS workgroup slots; a slot is FREE, CLAIMED (its state is being
// written, or was written in this launch), a LEADER (the state machine of one active range lo..i) or a CURSOR of some
// leader's train.  A slot's workgroup is the only one that reads its state, and it starts doing so in the launch AFTER
// the one that wrote it (epoch < tick), so state never crosses workgroups inside a launch; what does are the role
// words (agent-scope CAS), the per-leader counters of finished cursors and the counters of psd_rglobal.
// Claims come out of the leader's share of the tick's plan (psd_rq_plan): the slots of a share were FREE when the
// previous launch ended and belong to no other share, so no claim of this launch competes with another (round 2 took
// whatever a CAS scan found free at that moment: train sizes, and with them sweep counts, changed from run to run).
PSD_D int psd_mb_share_left(const psd_rparams& P, const psd_rstate& st, int& base) {  // (any lane; uniform)
    if (P.plan == nullptr || P.plan[PSD_PLAN_TICK] != P.tick) return 0;
    const int used = (st.plan_tick == P.tick) ? st.plan_used : 0;
    base = P.plan[PSD_PLAN_FIRST + st.slot] + used;
    const int left = P.plan[PSD_PLAN_COUNT + st.slot] - used;
    return (left > 0) ? left : 0;
}
PSD_D void psd_mb_share_take(const psd_rparams& P, psd_rstate& st, int k) {  // (every lane, uniform)
    if (k <= 0) return;
    st.plan_used = ((st.plan_tick == P.tick) ? st.plan_used : 0) + k;
    st.plan_tick = P.tick;
}
PSD_D int psd_mb_claim(const psd_rparams& P, const psd_rstate& st) {  // (one lane; the caller books it with psd_mb_share_take)
    int base = 0;
    if (psd_mb_share_left(P, st, base) < 1) return -1;
    const int s = P.plan[PSD_PLAN_FREE + base];
    psd_atomic_store(P.role + s, PSD_ROLE_CLAIMED);
    psd_atomic_max(&P.gl->nslotmax, s + 1);
    return s;
}
// Up to `want` slots of the share at once.  The slots go to list[0..], the count is returned in *count (both LDS).
PSD_D void psd_mb_claim_many(const psd_rparams& P, psd_rstate& st, int want, int* list, int* count) {
    PSD_SYNC();
    int base = 0;
    int k = psd_mb_share_left(P, st, base);
    if (k > want) k = want;
    if (k < 0) k = 0;
    PSD_PAR_FOR(q, k) {
        const int s = P.plan[PSD_PLAN_FREE + base + q];
        psd_atomic_store(P.role + s, PSD_ROLE_CLAIMED);
        list[q] = s;
        if (q == k - 1) psd_atomic_max(&P.gl->nslotmax, s + 1);  // (the free list ascends)
    }
    PSD_ONE { *count = k; }
    psd_mb_share_take(P, st, k);
    PSD_SYNC();
}
// hands a slot back: everything this workgroup stored must be out before another workgroup may reuse the slot
PSD_D void psd_mb_release_slot(const psd_rparams& P, int s) {  // (one lane)
    psd_release_fence();
    psd_atomic_store(P.epoch + s, PSD_EPOCH_NEVER);
    psd_atomic_store(P.role + s, PSD_ROLE_FREE);
}
// A negligible subdiagonal at l > lo splits the range: rows/columns lo..l-1 become an active range of their own with
// its own leader (if a slot is free; otherwise they wait their turn as in the reference, PSD.jl:1057-1060).
// bc: LDS broadcast cell.
PSD_D void psd_mb_spawn(const psd_rparams& P, psd_rstate& st, int* bc) {
    PSD_SYNC();
    PSD_ONE {
        const int s = psd_mb_claim(P, st);
        if (s >= 0) {
            psd_rstate cs = st;
            cs.slot = s;
            cs.parent = -1;
            cs.cursor = 0;
            cs.i = st.l - 1;
            cs.l = st.lo;
            cs.its = 1;
            cs.maxitleft = psd_atomic_load(&P.gl->pbudget[st.prob]);
            cs.exc_dec = 0;
            cs.kcur = 0;
            cs.phase = PSD_PH_DECIDE;
            cs.train_n = 1;
            cs.W = st.Wmax;
            cs.niter = 0;
            cs.maxits = 0;
            cs.nsweeps = cs.nrqpass = cs.ndefl1 = cs.ndefl2 = cs.nwindows = cs.nlog = 0;
            cs.ntrains = cs.ntrainsweeps = 0;
            cs.plan_tick = -1;
            cs.plan_used = 0;
            cs.opn_tick = -2;
            for (int q = 0; q < 6; ++q) cs.cyc[q] = 0;
            P.cst[s] = cs;
            psd_atomic_add(&P.gl->nactive, 1);
            psd_atomic_add(&P.gl->pactive[st.prob], 1);
            psd_atomic_add(&P.gl->nspawn, 1);
            psd_atomic_store(P.epoch + s, P.tick);
        }
        bc[0] = s;
    }
    PSD_SYNC();
    if (bc[0] >= 0) {
        st.lo = st.l;
        psd_mb_share_take(P, st, 1);
    }
    PSD_SYNC();
}
// a leader whose range is exhausted: totals to psd_rglobal, the last one alive finishes the decomposition
// (PSD.jl:1066-1073) and raises `done`
PSD_D void psd_mb_finish_leader(const psd_rparams& P, psd_rstate& st, int* bc) {
    PSD_SYNC();
    PSD_ONE {
        psd_rglobal* g = P.gl;
        psd_atomic_add(&g->nsweeps, st.nsweeps);
        psd_atomic_add(&g->nrqpass, st.nrqpass);
        psd_atomic_add(&g->ndefl1, st.ndefl1);
        psd_atomic_add(&g->ndefl2, st.ndefl2);
        psd_atomic_add(&g->nwindows, st.nwindows);
        psd_atomic_add(&g->ntrains, st.ntrains);
        psd_atomic_add(&g->ntrainsweeps, st.ntrainsweeps);
        psd_atomic_max(&g->maxits, st.maxits);
        psd_atomic_add_ll(&g->niter, st.niter);
        for (int q = 0; q < 6; ++q) psd_atomic_add_ll(&g->cyc[q], st.cyc[q]);
        psd_release_fence();  // (eigenvalues of this range are read by whoever finishes last)
        bc[1] = psd_atomic_add(&g->pactive[st.prob], -1) - 1;
        bc[0] = psd_atomic_add(&g->nactive, -1) - 1;
    }
    PSD_SYNC();
    const int left = bc[0], leftp = bc[1];
    PSD_SYNC();
    if (leftp == 0) {  // last range of this problem: PSD.jl:1066-1073
        psd_acquire_fence();
        const psd_mat<double> H1 = psd_fac(P, st.n, 1);
        PSD_PAR_FOR(q, st.n - 1) {
            if (P.wi[q] == 0.0) H1(q + 2, q + 1) = 0.0;
        }
        PSD_SYNC();
    }
    if (left == 0) {
        PSD_ONE { psd_atomic_store(&P.gl->done, 1); }
    }
    st.phase = PSD_PH_DONE;  // (psd_rq_step_body hands the slot back after its last store)
}

// Product band of the decisions pending at the start of a tick, all slots, 64 rows per workgroup (multi-block mode).
// A leader that computes the band of a wide range itself walks p * w scattered cache lines with one wavefront: 1 ms at
// w = 1024, p = 64, three windows' worth of chase with every other slot waiting for the tick to end.  Here the same
// rows are spread over the chip in front of the chase launch; psd_rq_decide finds them through P.bandinfo.  Same
// recurrence, same order of operations per row (PSD.jl:475-495,507-516).  grid = (ceil(n / 64), PSD_SLOTS), 64 threads.
// The slot plan of a tick (one workgroup of 64 lanes = slots, the last row of psd_rq_band's grid; runs between two chase
// launches, so the slot words and states it reads are at rest).  Candidates are the leaders that may start a train or
// hand a range over in the coming launch (any phase from which PSD_PH_SHIFT can be reached before the workgroup
// emits); each gets a share of the free slots sized by the widest range it can decide on, the widest range first.
#define PSD_PLAN_LDS_INTS (5 * PSD_SLOTS)
PSD_D void psd_rq_plan_body(const psd_rparams& P) {
    PSD_LDS_DECL;
    int* isfree = (int*)psd_lds;
    int* cw = isfree + PSD_SLOTS;     // width of a candidate's range (0: no candidate)
    int* need = cw + PSD_SLOTS;       // slots it may ask for
    int* order = need + PSD_SLOTS;    // candidates by rank
    int* nfree = order + PSD_SLOTS;   // [0] free slots
    PSD_PAR_FOR(s, PSD_SLOTS) {
        // (every word this lane needs is requested before the first is looked at: one memory round trip, not two —
        //  a free slot's state is stale, never unmapped)
        const psd_rstate* x = P.cst + s;
        const int role = psd_atomic_load(P.role + s);
        const int ep = psd_atomic_load(P.epoch + s);
        const int ph = x->phase, xcur = x->cursor, xi = x->i, xlo = x->lo, xW = x->Wmax, xwant = x->train_want;
        isfree[s] = (role == PSD_ROLE_FREE) ? 1 : 0;
        int w = 0, nd = 0;
        if (role == PSD_ROLE_LEADER || (role == PSD_ROLE_CLAIMED && ep < P.tick)) {
            if (xcur == 0 && (ph == PSD_PH_DECIDE || ph == PSD_PH_SHIFT || ph == PSD_PH_DEFLATE || ph == PSD_PH_NEXT)) {
                w = xi - xlo + 1;
                if (w < 1) w = 1;
                // most slots psd_rq_shift can ask for on a range of width <= w: the long-train pipeline at the
                // narrowest window it considers, plus one for a hand-over (psd_mb_spawn)
                int nbm = xW - 4;
                if (nbm > 8) nbm = 8;
                if (nbm < 1) nbm = 1;
                nd = (w + 4 * nbm + 3) / (nbm + 4) + 1;
                if (nd > xwant) nd = xwant;
                if (nd > PSD_TRAIN_MAX - 1) nd = PSD_TRAIN_MAX - 1;
                if (nd < 0) nd = 0;
                nd += 1;
            }
        }
        cw[s] = w;
        need[s] = nd;
        P.plan[PSD_PLAN_CDONE + s] = psd_atomic_load(P.cdone + s);
        P.plan[PSD_PLAN_CCANCEL + s] = (P.ccancel != nullptr) ? psd_atomic_load(P.ccancel + s) : 0;
    }
#ifndef PSD_HOSTSIM
    {
        // one wavefront, lane = slot: positions among the free slots by a ballot, ranks among the candidates by a
        // broadcast loop over the lanes' registers, the shares by a scan in rank order (the portable form below walks
        // LDS 64 times per lane in three passes: 9 us in front of every chase launch)
        const int s = PSD_TID;
        const int w = cw[s], fr = isfree[s], nd = need[s];
        const unsigned long long fmask = __ballot(fr != 0);
        const int fidx = __popcll(fmask & ((1ull << s) - 1ull));
        const int nf = __popcll(fmask);
        if (fr) P.plan[PSD_PLAN_FREE + fidx] = s;
        int rank = 0;
        for (int t = 0; t < PSD_SLOTS; ++t) {
            const int wt = __builtin_amdgcn_readlane(w, t);
            if (w > 0 && wt > 0 && (wt > w || (wt == w && t < s))) rank += 1;
        }
        if (w > 0) order[rank] = nd;  // (need, by rank)
        const unsigned long long cmask = __ballot(w > 0);
        const int nc = __popcll(cmask);
        PSD_SYNC();
        int incl = (s < nc) ? order[s] : 0;
        const int own = incl;
#pragma unroll
        for (int sft = 1; sft < 64; sft <<= 1) {
            const int up = __shfl_up(incl, sft, 64);
            if (s >= sft) incl += up;
        }
        PSD_SYNC();
        if (s < nc) order[s] = incl - own;  // (first, by rank)
        PSD_SYNC();
        int first = 0, cnt = 0;
        if (w > 0) {
            first = order[rank];
            if (first > nf) first = nf;
            cnt = nd;
            if (cnt > nf - first) cnt = nf - first;
        }
        P.plan[PSD_PLAN_FIRST + s] = first;
        P.plan[PSD_PLAN_COUNT + s] = cnt;
        (void)nfree;
    }
#else
    PSD_SYNC();
    PSD_PAR_FOR(s, PSD_SLOTS) {
        // position of slot s among the free slots (ascending) and among the candidates (widest first, then by slot)
        int fidx = 0, rank = 0;
        const int w = cw[s];
        for (int t = 0; t < PSD_SLOTS; ++t) {
            if (t < s && isfree[t]) fidx += 1;
            if (w > 0 && cw[t] > 0 && (cw[t] > w || (cw[t] == w && t < s))) rank += 1;
        }
        if (isfree[s]) P.plan[PSD_PLAN_FREE + fidx] = s;
        if (s == PSD_SLOTS - 1) nfree[0] = fidx + isfree[s];
        if (w > 0) order[rank] = s;
    }
    PSD_SYNC();
    PSD_PAR_FOR(s, PSD_SLOTS) {
        int first = 0, cnt = 0;
        const int w = cw[s];
        if (w > 0) {
            int rank = 0;
            for (int t = 0; t < PSD_SLOTS; ++t)
                if (cw[t] > 0 && (cw[t] > w || (cw[t] == w && t < s))) rank += 1;
            for (int r = 0; r < rank; ++r) first += need[order[r]];
            const int nf = nfree[0];
            if (first > nf) first = nf;
            cnt = need[s];
            if (cnt > nf - first) cnt = nf - first;
        }
        P.plan[PSD_PLAN_FIRST + s] = first;
        P.plan[PSD_PLAN_COUNT + s] = cnt;
    }
#endif
    PSD_ONE { P.plan[PSD_PLAN_TICK] = P.tick; }
}

PSD_KERNEL_B(64) psd_rq_band(psd_rparams P, int n, int p) {
    if (P.plan != nullptr && PSD_BLOCK_Y == PSD_GRID_Y - 1) {  // (grid.y = slots with a band row + 1)
        if (PSD_BLOCK_X == 0) psd_rq_plan_body(P);
        return;
    }
    const int s = PSD_BLOCK_Y, chunk = PSD_BLOCK_X;
    // (a leader, or a slot a leader was spawned into in an earlier launch: it goes live with the chase launch behind this one)
    const int role = psd_atomic_load(P.role + s);
    if (role != PSD_ROLE_LEADER && role != PSD_ROLE_CLAIMED) return;
    if (psd_atomic_load(P.epoch + s) >= P.tick) return;  // (state still being written / not this slot's turn yet)
    if (psd_atomic_load(&P.gl->done) || psd_atomic_load(&P.gl->abort)) return;
    const psd_rstate* st = P.cst + s;
    if (st->phase != PSD_PH_DECIDE || st->cursor != 0) return;
    const int lo = st->l, i = st->i;
    if (i - lo + 1 < PSD_DECIDE_YIELD) return;
    const int a0 = lo + 64 * chunk;
    if (a0 > i) return;
    const size_t sb = (size_t)st->prob * (n + 8);
    double* H = P.H + (size_t)st->prob * p * n * n;
    PSD_PAR_FOR(t, 64) {
        const int a = a0 + t;
        if (a <= i) {
            double d = 1.0, e = 0.0, f = 0.0;
            for (int j = 2; j <= p; ++j) {
                const psd_mat<double> Hj = psd_mat<double>{H + (size_t)(j - 1) * n * n, n};
                if (a + 2 <= i) f = d * Hj(a, a + 2) + e * Hj(a + 1, a + 2) + f * Hj(a + 2, a + 2);
                if (a + 1 <= i) e = d * Hj(a, a + 1) + e * Hj(a + 1, a + 1);
                d *= Hj(a, a);
            }
            P.Pd[sb + a] = d;
            P.Pe[sb + a] = e;
            P.Pf[sb + a] = f;
        }
    }
    if (chunk == 0) {
        PSD_ONE {
            P.bandinfo[s] = P.tick;
            P.bandinfo[PSD_SLOTS + s] = lo;
            P.bandinfo[2 * PSD_SLOTS + s] = i;
        }
    }
}

// opnorm(view(H1, lo:hi, lo:hi), 1) — PSD.jl:537,596 fallback when a diagonal pair is exactly zero
PSD_D double psd_h1_opnorm(const psd_rparams& P, const psd_rstate& st, double* red, int lo, int hi) {
    const psd_mat<double> H1 = psd_fac(P, st.n, 1);
    const int NT = PSD_NTHREADS;
    PSD_PAR_FOR(t, NT) {
        double best = 0.0;
        for (int c = lo + t; c <= hi; c += NT) {
            double s = 0.0;
            const int rmax = (c + 1 < hi) ? (c + 1) : hi;
            for (int r = lo; r <= rmax; ++r) s += fabs(H1(r, c));
            if (s > best) best = s;
        }
        red[t] = best;
    }
    PSD_SYNC();
    double best = 0.0;
    for (int t = 0; t < NT; ++t)
        if (red[t] > best) best = red[t];
    PSD_SYNC();
    return best;
}

// ------------------------------------------------------------------------------------------------
// PSD.jl:471-672: product band, deflation search, RQ decision
// stage: LDS free during the decision (the window area), stage_doubles of it
// Returns true when the decision put itself off by one launch (deferred column roles: psd_cdefer_edge).
PSD_D bool psd_rq_decide(const psd_rparams& P, psd_rstate& st, double* red, int* redi, double* stage, size_t stage_doubles) {
    const int n = st.n, p = st.p, i = st.i, lo = st.l;
    const int NT = PSD_NTHREADS;
    if (!(st.its < st.maxitleft)) {  // PSD.jl:471,891-893
        if (st.mb && P.nprob > 1) {
            // batch: only this problem has failed.  Its info word says so; this range ends like an exhausted one (the
            // leader counts itself out, the slot goes back), the other problems of the call run on.
            PSD_ONE { psd_atomic_store(&P.gl->pinfo[st.prob], i); }
            st.phase = PSD_PH_FINAL;
            return false;
        }
        st.info = i;
        st.phase = PSD_PH_DONE;
        if (st.mb) {
            PSD_ONE {
                psd_atomic_store(&P.gl->info, i);
                psd_atomic_store(&P.gl->pinfo[st.prob], i);
                psd_atomic_store(&P.gl->abort, 1);
                psd_atomic_store(&P.gl->done, 1);
            }
        }
        return false;
    }
    const psd_mat<double> H1 = psd_fac(P, n, 1);
    // band of P = H_2 H_3 ... H_p on rows lo..i (PSD.jl:475-495,507-516, evaluated per row).  The entries a row needs —
    // Hj(a, a..a+2), Hj(a+1, a+1..a+2), Hj(a+2, a+2) — are the last three of columns a, a+1, a+2 above the diagonal:
    // 24 contiguous bytes per (column, factor).  They are staged through LDS in chunks of R rows with ALL the loads of a
    // chunk independent of each other (a lane that walks j with the recurrence in front of every load pays the HBM
    // latency p times per row: 1 ms for a full-width block at p = 64), then the recurrence runs out of LDS.
    bool have = false;  // psd_rq_band has the band of rows lo..i (same H: nothing was emitted since)
    if (P.bandinfo != nullptr && st.mb) {
        have = P.bandinfo[st.slot] == P.tick && P.bandinfo[PSD_SLOTS + st.slot] <= lo && i <= P.bandinfo[2 * PSD_SLOTS + st.slot];
    }
    if (!have) {
        const int pj = p - 1;  // factors 2..p
        int R = (pj > 0) ? (int)(stage_doubles / (3 * (size_t)pj)) - 2 : NT;
        if (R > NT) R = NT;
        if (pj > 0 && R >= 1) {
            for (int a0 = lo; a0 <= i; a0 += R) {
                const int a1 = (a0 + R - 1 < i) ? (a0 + R - 1) : i;  // rows a0..a1, columns a0..min(a1 + 2, i)
                const int c1 = (a1 + 2 < i) ? (a1 + 2) : i;
                const int nc = c1 - a0 + 1;
                PSD_SYNC();
                PSD_PAR_FOR(t, nc * pj) {
                    const int cc = t % nc, jj = t / nc;  // (consecutive lanes: consecutive columns of one factor)
                    const int c = a0 + cc;
                    const psd_mat<double> Hj = psd_fac(P, n, jj + 2);
                    double* q = stage + 3 * (size_t)t;
                    q[0] = Hj(c, c);
                    q[1] = (c - 1 >= lo) ? Hj(c - 1, c) : 0.0;
                    q[2] = (c - 2 >= lo) ? Hj(c - 2, c) : 0.0;
                }
                PSD_SYNC();
                PSD_PAR_FOR(t, a1 - a0 + 1) {
                    const int a = a0 + t;
                    double d = 1.0, e = 0.0, f = 0.0;
                    for (int jj = 0; jj < pj; ++jj) {
                        const double* q0 = stage + 3 * ((size_t)jj * nc + t);  // column a: (a,a), (a-1,a), (a-2,a)
                        if (a + 2 <= i) f = d * q0[8] + e * q0[7] + f * q0[6];     // column a+2: (a,a+2), (a+1,a+2), (a+2,a+2)
                        if (a + 1 <= i) e = d * q0[4] + e * q0[3];                 // column a+1: (a,a+1), (a+1,a+1)
                        d *= q0[0];
                    }
                    P.Pd[a] = d;
                    P.Pe[a] = e;
                    P.Pf[a] = f;
                }
            }
        } else {
            PSD_PAR_FOR(t, i - lo + 1) {
                const int a = lo + t;
                double d = 1.0, e = 0.0, f = 0.0;
                for (int j = 2; j <= p; ++j) {
                    const psd_mat<double> Hj = psd_fac(P, n, j);
                    if (a + 2 <= i) f = d * Hj(a, a + 2) + e * Hj(a + 1, a + 2) + f * Hj(a + 2, a + 2);
                    if (a + 1 <= i) e = d * Hj(a, a + 1) + e * Hj(a + 1, a + 1);
                    d *= Hj(a, a);
                }
                P.Pd[a] = d;
                P.Pe[a] = e;
                P.Pf[a] = f;
            }
        }
    }
    PSD_SYNC();
    // tridiagonal band of the product (PSD.jl:485-488,517-528)
    PSD_PAR_FOR(t, i - lo + 1) {
        const int r = lo + t;
        if (i == lo) {
            P.hdiag[i] = H1(i, i) * P.Pd[i];
        } else if (r == i) {
            P.hsub[i] = H1(i, i - 1) * P.Pd[i - 1];
            P.hdiag[i] = H1(i, i - 1) * P.Pe[i - 1] + H1(i, i) * P.Pd[i];
        } else if (r > lo) {
            P.hsub[r] = H1(r, r - 1) * P.Pd[r - 1];
            P.hdiag[r] = H1(r, r - 1) * P.Pe[r - 1] + H1(r, r) * P.Pd[r];
            P.hsup[r] = H1(r, r - 1) * P.Pf[r - 1] + H1(r, r) * P.Pe[r] + H1(r, r + 1) * P.Pd[r + 1];
        } else {
            P.hdiag[r] = H1(r, r) * P.Pd[r];
            P.hsup[r] = H1(r, r) * P.Pe[r] + H1(r, r + 1) * P.Pd[r + 1];
        }
    }
    PSD_SYNC();
    // search for a negligible subdiagonal from the bottom (PSD.jl:504-576)
    int klast = 0;
    double h1norm = -1.0;
    for (int pass = 0; pass < 2; ++pass) {
        PSD_PAR_FOR(t, NT) {
            int best = 0, need = 0;
            for (int k = i - t; k >= lo + 1; k -= NT) {
                const double hh21 = P.hsub[k], hh22 = P.hdiag[k], hh11 = P.hdiag[k - 1], hh12 = P.hsup[k - 1];
                double tst1 = fabs(hh11) + fabs(hh22);
                if (tst1 == 0) {
                    if (h1norm < 0) {
                        need = 1;
                        continue;
                    }
                    tst1 = h1norm;
                }
                bool found = false;
                if (fabs(hh21) <= st.smlnum) {
                    found = true;
                } else if (fabs(hh21) <= st.ulp * tst1) {  // LAPACK + Ahues-Tisseur, PSD.jl:548-555
                    const double ab = fmax(fabs(hh21), fabs(hh12));
                    const double ba = fmin(fabs(hh21), fabs(hh12));
                    const double aa = fmax(fabs(hh22), fabs(hh11 - hh22));
                    const double bb = fmin(fabs(hh22), fabs(hh11 - hh22));
                    const double stmp = aa + ab;
                    found = ba * (ab / stmp) <= fmax(st.smlnum, st.ulpx * (bb * (aa / stmp)));
                }
                if (found) {
                    best = k;
                    break;
                }
            }
            redi[t] = best;
            redi[NT + t] = need;
        }
        PSD_SYNC();
        int need = 0;
        klast = 0;
        for (int t = 0; t < NT; ++t) {
            if (redi[t] > klast) klast = redi[t];
            need |= redi[NT + t];
        }
        PSD_SYNC();
        if (!need) break;
        if (P.cdefer && st.opn_tick != P.tick - 1) {  // far column updates of this range's last windows may be in flight
            st.opn_tick = P.tick;
            return true;
        }
        h1norm = psd_h1_opnorm(P, st, red, lo, i);
    }
    const bool found = klast > 0;
    const int l = (i > lo) ? (found ? klast : lo) : i;  // PSD.jl:585
    st.l = l;
    st.phase = PSD_PH_SHIFT;
    if (l > 1 && st.wantT) {  // PSD.jl:589-665
        double tst1 = fabs(H1(l - 1, l - 1)) + fabs(H1(l, l));
        if (tst1 == 0) {
            if (P.cdefer && st.opn_tick != P.tick - 1) {
                st.opn_tick = P.tick;
                st.l = lo;
                st.phase = PSD_PH_DECIDE;
                return true;
            }
            tst1 = psd_h1_opnorm(P, st, red, l, i);
        }
        if (fabs(H1(l, l - 1)) > fmax(st.ulp * tst1, st.smlnum)) {
            st.phase = PSD_PH_RQ;
            st.kcur = i;
            st.nrqpass += 1;
            psd_log(P, st, 1, l, i);
        } else {
            PSD_SYNC();
            PSD_ONE { H1(l, l - 1) = 0.0; }
            PSD_SYNC();
        }
    }
    return false;
}

// ------------------------------------------------------------------------------------------------
// Multishift trains (the default iteration strategy on large active blocks, psd_set_train / PSD_TRAIN).
// A train is m double-shift sweeps whose shift pairs are the eigenvalues of the trailing 2m x 2m block of the product,
// fixed before the first sweep starts.  The sweeps run as m cursors nb + 4 positions (or two windows) apart: cursor 0 is the ordinary state
// machine (the leader), cursors 1..m-1 are psd_rq_cursor_step launches with their own state, lists and descriptor.

// first column of (P - s1)(P - s2) at the top of the active block (PSD.jl:768-803) for the shift pair `sh`
PSD_D void psd_rq_startvec(const double h11, const double h12, const double h21, const double h22, const double h32,
                           const double* sh, double* v) {
    const double rt1r = sh[0], rt1i = sh[1], rt2r = sh[2], rt2i = sh[3];
    const double s = fabs(h11 - rt2r) + fabs(rt2i) + fabs(h21);
    const double h21s = h21 / s;
    const double v1 = h21s * h12 + (h11 - rt1r) * ((h11 - rt2r) / s) - rt1i * (rt2i / s);
    const double v2 = h21s * (h11 + h22 - rt1r - rt2r);
    const double v3 = h21s * h32;
    const double t = fabs(v1) + fabs(v2) + fabs(v3);
    v[0] = v1 / t;
    v[1] = v2 / t;
    v[2] = v3 / t;
}

// leading entries of the product band on rows l, l+1, l+2 straight from the factors (the recurrences of
// psd_rq_decide, PSD.jl:475-495): h11, h12, h21, h22, h32.  Evaluated redundantly by every lane.
PSD_D void psd_rq_topband(const psd_rparams& P, int n, int p, int l, int i, double& h11, double& h12, double& h21,
                          double& h22, double& h32) {
    double d0 = 1.0, e0 = 0.0, d1 = 1.0;
    for (int j = 2; j <= p; ++j) {
        const psd_mat<double> Hj = psd_fac(P, n, j);
        if (l + 1 <= i) e0 = d0 * Hj(l, l + 1) + e0 * Hj(l + 1, l + 1);
        d0 *= Hj(l, l);
        if (l + 1 <= i) d1 *= Hj(l + 1, l + 1);
    }
    const psd_mat<double> H1 = psd_fac(P, n, 1);
    h11 = H1(l, l) * d0;
    h12 = H1(l, l) * e0 + H1(l, l + 1) * d1;
    h21 = H1(l + 1, l) * d0;
    h22 = H1(l + 1, l) * e0 + H1(l + 1, l + 1) * d1;
    h32 = (l + 2 <= i) ? H1(l + 2, l + 1) * d1 : 0.0;
}

// The 2m shifts of a train: eigenvalues of the trailing K x K block (K = 2m) of H_1 H_2 ... H_p, which is
// H_1[t0:i, t0-1:i] * (prod_j H_j[t0-1:i, t0-1:i])[:, 2:end] because the other factors are triangular.
// Called by every lane: the (K+1) x (K+1) trailing blocks of all factors are staged in LDS by the whole wavefront (read
// one by one from HBM by a single lane they cost a millisecond per train), the triangular products run one entry per
// lane, only the small Hessenberg-QR is one lane's work.  `work`: LDS, psd_rq_train_doubles(p, m) doubles (the window
// area is free while the shifts are computed).  Pairs go to P.tshift; *okf = 1 on success.
PSD_HD size_t psd_rq_train_doubles(int p, int m) {
    const size_t K = 2 * (size_t)m, K1 = K + 1;
    return (size_t)p * K1 * K1 + 2 * K1 * K1 + K * K + 3 * PSD_HQR_MAX + 8;
}
PSD_D void psd_rq_train_shifts(const psd_rparams& P, int n, int p, int i, int m, double* work, int* okf) {
    const int K = 2 * m, K1 = K + 1, t0 = i - K + 1, KK = K1 * K1;
    double* B = work;                // [p][K1][K1] row-major trailing blocks (rows / columns t0-1 .. i)
    double* R0 = B + (size_t)p * KK;  // running product, double-buffered
    double* R1 = R0 + KK;
    double* T = R1 + KK;             // K x K
    double* wr = T + K * K;
    double* wi = wr + PSD_HQR_MAX;
    double* re = wi + PSD_HQR_MAX;
    PSD_SYNC();
    PSD_PAR_FOR(t, p * KK) {
        const int j = t / KK, q = t - j * KK, r = q / K1, c = q - r * K1;
        B[t] = psd_fac(P, n, j + 1)(t0 - 1 + r, t0 - 1 + c);
    }
    PSD_PAR_FOR(q, KK) { R0[q] = (q / K1 == q % K1) ? 1.0 : 0.0; }
    PSD_SYNC();
    double* cur = R0;
    double* nxt = R1;
    for (int j = 2; j <= p; ++j) {  // cur <- cur * B_j (both upper triangular), one entry per lane
        const double* Bj = B + (size_t)(j - 1) * KK;
        PSD_PAR_FOR(q, KK) {
            const int r = q / K1, c = q - r * K1;
            double acc = 0.0;
            for (int k = r; k <= c; ++k) acc += cur[r * K1 + k] * Bj[k * K1 + c];
            nxt[q] = acc;
        }
        PSD_SYNC();
        double* sw = cur;
        cur = nxt;
        nxt = sw;
    }
    PSD_PAR_FOR(q, K * K) {  // T[r, c] = sum_k H_1[t0 + r, t0 - 1 + k] R[k, c + 1]
        const int r = q / K, c = q - r * K;
        double acc = 0.0;
        for (int k = r; k <= c + 1; ++k) acc += B[(r + 1) * K1 + k] * cur[k * K1 + (c + 1)];
        T[q] = acc;
    }
    PSD_SYNC();
#ifndef PSD_HOSTSIM
    const long long _hq0 = psd_clock();
    const bool okw = psd_hqr_wave(T, K, K, wr, wi, PSD_TID);  // (the workgroup is one wavefront)
    if (P.ticklog != nullptr && P.gl != nullptr) {
        PSD_ONE {
            psd_atomic_add_ll(&P.gl->dbg[3], psd_clock() - _hq0);
            psd_atomic_add(&P.gl->dbgn[3], 1);
        }
    }
#endif
    PSD_ONE {
#ifndef PSD_HOSTSIM
        bool ok = okw;
#else
        bool ok = psd_hqr(T, K, K, wr, wi);
#endif
        int np = 0, nre = 0;
        // conjugate pairs first, then the real eigenvalues in ascending order two by two
        for (int q = 0; ok && q < K; ++q) {
            if (!(wr[q] == wr[q]) || !(wi[q] == wi[q])) ok = false;
            if (wi[q] > 0.0) {
                if (np < m) {
                    double* sh = P.tshift + 4 * np;
                    sh[0] = wr[q]; sh[1] = wi[q]; sh[2] = wr[q]; sh[3] = -wi[q];
                    ++np;
                }
            } else if (wi[q] == 0.0) {
                re[nre++] = wr[q];
            }
        }
        for (int a = 1; a < nre; ++a) {  // insertion sort
            const double x = re[a];
            int b = a - 1;
            while (b >= 0 && re[b] > x) {
                re[b + 1] = re[b];
                --b;
            }
            re[b + 1] = x;
        }
        for (int a = 0; ok && a < nre && np < m; a += 2) {
            double* sh = P.tshift + 4 * np;
            sh[0] = re[a]; sh[1] = 0.0; sh[2] = (a + 1 < nre) ? re[a + 1] : re[a]; sh[3] = 0.0;
            ++np;
        }
        *okf = (ok && np == m) ? 1 : 0;
    }
    PSD_SYNC();
}

// PSD.jl:668-803: split test, shifts, first column of the shifted product
// entry tick and first window of cursor b of the running train (see psd_rstate::cstart)
PSD_HD void psd_cursor_schedule(const psd_rstate& st, int b, int& cstart, int& cfirst) {
    const int nbw = st.W - 4;
    const int spc = (st.tgap == 1) ? (nbw + 4) : (2 * nbw);
    int d = (b * spc - nbw + 1 + nbw - 1) / nbw;  // ceil((b s - nb + 1) / nb)
    if (d < 1) d = 1;
    int x = nbw * d - b * spc + nbw;
    if (x > nbw) x = nbw;
    if (x < 1) x = 1;
    cstart = st.train_tick0 + d;
    cfirst = x;
}

// Returns true when the shift pairs of a multishift train were computed (trailing blocks of all factors, their product,
// a small Hessenberg-QR in one lane: as long as a window's chase) — the multi-block driver then ends this workgroup's
// launch there, so that the tick is not stretched for all the other slots; the sweep starts with the next launch.
PSD_D bool psd_rq_shift(const psd_rparams& P, psd_rstate& st, double* work, int* bc) {
    const int n = st.n, i = st.i, l = st.l;
    bool heavy = false;
    // (after the RQ clean-up at l, which still touches position l - 1; a block that deflates at once is not worth a
    //  hand-over: the range above it simply stays with this leader, as in the reference)
    if (st.mb && st.l > st.lo && st.l < i - 1) {
        PSD_DBG_T0();
        psd_mb_spawn(P, st, bc);
        PSD_DBG_ADD(1);
    }
    if (l >= i - 1) {
        st.phase = PSD_PH_DEFLATE;
        return false;
    }
    if (!st.wantT) {
        st.i1 = l;
        st.i2 = i;
    }
    psd_log(P, st, 0, l, i);
    st.nsweeps += 1;
    const double dat1 = 0.75, dat2 = -0.4375;
    const double* hdiag = P.hdiag;
    const double* hsub = P.hsub;
    const double* hsup = P.hsup;
    double h33 = 0, h44 = 0, h43h34 = 0;
    double rt1r = 0, rt2r = 0, rt1i = 0, rt2i = 0;
    bool exc = false;
    // (its == 10, then its % 10 == 0 in the reference; a train advances its by its number of bulges, so the test is
    //  "a new decade since the last exceptional shift" — the same thing for unit steps)
    const int dec = st.its / 10;
    if (dec > st.exc_dec && dec == 1) {  // PSD.jl:680-689
        st.exc_dec = dec;
        exc = true;
        const double s = fabs(hsub[l + 1]) + fabs(hsub[l + 2]);
        h44 = dat1 * s + hdiag[l];
        h33 = h44;
        h43h34 = dat2 * s * s;
    } else if (dec > st.exc_dec) {  // PSD.jl:690-699
        st.exc_dec = dec;
        exc = true;
        const double s = fabs(hsub[i]) + fabs(hsub[i - 1]);
        h44 = dat1 * s + hdiag[i];
        h33 = h44;
        h43h34 = dat2 * s * s;
    } else {  // PSD.jl:729-762 (dlahqr shifts; _slicot_shifts[] is false by default)
        h44 = hdiag[i];
        h33 = hdiag[i - 1];
        double h43 = hsub[i], h34 = hsup[i - 1];
        const double s = fabs(h33) + fabs(h34) + fabs(h43) + fabs(h44);
        if (s != 0) {
            h33 /= s; h44 /= s; h34 /= s; h43 /= s;
            const double trc = (h33 + h44) * 0.5;
            const double disc = (h33 - trc) * (h44 - trc) - h34 * h43;
            const double rtdisc = sqrt(fabs(disc));
            if (disc >= 0) {
                rt1r = trc * s;
                rt2r = rt1r;
                rt1i = rtdisc * s;
                rt2i = -rt1i;
            } else {
                rt1r = trc + rtdisc;
                rt2r = trc - rtdisc;
                rt1r = (fabs(rt1r - h44) <= fabs(rt2r - h44)) ? (rt1r * s) : (rt2r * s);
                rt2r = rt1r;
                rt1i = rt2i = 0.0;
            }
        }
        // multishift train: m bulges if the active block leaves room for cursors nb + 4 positions (cgap 2: two windows) apart
        st.train_n = 1;
        if (st.train_want >= 2 && P.cst != nullptr) {
            // Window width of this train: nb positions per window cost a tick of about nb p c + o (c: one position of one
            // factor, o: launches, window transfer and bulk updates of a tick; o / c = train_oc, about 104), a train of m
            // cursors two windows apart takes w / nb + 2 (m - 1) ticks, and m is limited by the room, m <= 1 + (w - nb) /
            // (2 nb).  The width with the least modelled time per sweep is taken (narrow windows: more, cheaper ticks and
            // more cursors).
            const int w = i - l + 1;
            // (round 4: with the scan chase a position costs little and hardly depends on p, so what a tick costs beside
            //  its positions weighs more the wider the active block is — the bulk updates grow with it.  Measured with fixed
            //  ratios (tools/r04/run_ff.sh): the best o / c is about 104 at w = 256, 400 at w = 512, 800 .. 3200 at w = 1024
            //  for p = 16 (iteration 394 -> 353 ms), and makes no difference at p = 64, where the LDS caps the width: o / c
            //  grows with (w / 256)^2.)
            double ocw = (double)st.train_oc;
            // (periods of 12 and more only: short periods keep the schedule their residual margins were measured with — the
            //  hard inputs of DESIGN section 6, n = 600 .. 700 with p <= 11, sit at 0.71 .. 0.94 of the gate with it and one of
            //  them at 1.03 with the wider windows)
            if (w > 256 && st.p >= 12) {
                double f = (double)w / 256.0;
                f *= f;
                if (f > 32.0) f = 32.0;
                ocw *= f;
            }
            int mt = st.train_want;
            if (mt > PSD_TRAIN_MAX) mt = PSD_TRAIN_MAX;
            int nb = st.Wmax - 4, m = 1;
            double best = 1e300;
            for (int nbc = (st.Wmax - 4 < 8) ? ((st.Wmax > 5) ? st.Wmax - 4 : 1) : 8; nbc <= st.Wmax - 4; ++nbc) {
                const int gap = (st.cgap == 1 && nbc >= 6) ? 1 : 2;
                int mc = 1 + (w - nbc) / ((gap == 1) ? (nbc + 4) : (2 * nbc));
                if (mc > mt) mc = mt;
                if (mc < 2) break;
                const int spc = (gap == 1) ? (nbc + 4) : (2 * nbc);
                const double cost = (double)((w + nbc - 1) / nbc + ((mc - 1) * spc + nbc - 1) / nbc) *
                                    ((double)(nbc * st.p) + ocw) / mc;
                if (cost < best) {
                    best = cost;
                    nb = nbc;
                    m = mc;
                }
            }
            // Long train (slots recycled, see psd_rstate::train_S): in its steady state S(nb) = the slots that keep the
            // pipeline full are all busy, S nb / w sweeps per tick at nb p c + o per tick.  The width with the most
            // sweeps per time among those whose S fits the slots is taken; none fits: the train above.
            if (st.mb && st.train_long > m && m >= 2 && st.cgap == 1 && w >= PSD_TRAIN_LONG_MINW && st.train_want >= PSD_TRAIN_MAX) {
                double bestthr = 0.0;
                int nbl = 0, ml = 0;
                for (int nbc = (st.Wmax - 4 < 8) ? ((st.Wmax - 4 >= 6) ? st.Wmax - 4 : 6) : 8; nbc <= st.Wmax - 4; ++nbc) {
                    if (nbc < 6) continue;
                    const int spx = nbc + 4;
                    const int sneed = (((w - 1 + nbc - 1) / nbc + 2) * nbc + spx - 1) / spx;
                    int mx = 1 + (w - nbc) / spx;
                    if (mx > mt) mx = mt;
                    if (sneed > PSD_TRAIN_MAX - 1 || mx < 2 || w / st.train_wdiv <= sneed + 1) continue;
                    const double thr = (double)sneed * nbc / ((double)(nbc * st.p) + ocw);
                    if (thr > bestthr) {
                        bestthr = thr;
                        nbl = nbc;
                        ml = mx;
                    }
                }
                if (nbl > 0) {
                    nb = nbl;
                    m = ml;
                }
            }
            // the shift pairs come from one lane's Hessenberg-QR of a block of order <= PSD_HQR_MAX: a longer train
            // runs through them twice
            int ms = (2 * m > PSD_HQR_MAX) ? PSD_HQR_MAX / 2 : m;
            while (ms >= 2 && psd_rq_train_doubles(st.p, ms) > (size_t)st.p * psd_win_area(st.Wmax)) --ms;  // LDS of the staging
            if (ms < m && ms < PSD_HQR_MAX / 2) m = ms;
            if (m >= 2 && 2 * m + 2 <= w) {
                int* okf = (int*)P.tshift + 8 * PSD_TRAIN_MAX;  // (flag word behind the pairs)
                {
                    PSD_DBG_T0();
                    psd_rq_train_shifts(P, n, st.p, i, ms, work, okf);
                    PSD_DBG_ADD(2);
                }
                heavy = st.mb != 0;
                if (*okf && m > ms) {
                    PSD_ONE {
                        for (int b = ms; b < m && b < PSD_TRAIN_MAX; ++b)
                            for (int q = 0; q < 4; ++q) P.tshift[4 * b + q] = P.tshift[4 * (b - ms) + q];
                    }
                    PSD_SYNC();
                }
                int mgot = m;
                if (*okf && st.mb) {  // the cursors need slots: as many as are free
                    PSD_DBG_T0();
                    // (a long train needs a few slots more than bulges fit the range at once: see below)
                    int wantslots = m - 1;
                    if (st.train_long > m && w >= PSD_TRAIN_LONG_MINW && st.train_want >= PSD_TRAIN_MAX) {  // (a caller's cap on the bulges in flight stands)
                        const int tg = (st.cgap == 1 && nb >= 6) ? 1 : 2, spc = (tg == 1) ? (nb + 4) : (2 * nb);
                        const int sneed = (((i - l + nb - 1) / nb + 2) * nb + spc - 1) / spc;
                        if (sneed > wantslots) wantslots = sneed;
                        if (wantslots > PSD_TRAIN_MAX - 1) wantslots = PSD_TRAIN_MAX - 1;
                    }
                    psd_mb_claim_many(P, st, wantslots, bc + 3, bc);
                    PSD_DBG_ADD(4);
                    PSD_ONE {
                        bc[0] += 1;  // (with the leader)
                        bc[1] = psd_atomic_add(&P.gl->train_seq, 1) + 1;
                        psd_atomic_store(P.cdone + st.slot, 0);  // (no cursor of an earlier train of this leader is left)
                        if (P.ccancel) {
                            psd_atomic_store(P.ccancel + st.slot, 0);
                            psd_atomic_store(P.ccancel + PSD_SLOTS + st.slot, 0);
                        }
                    }
                    PSD_SYNC();
                    mgot = bc[0];
                    st.train_key = bc[1];
                    for (int b = 1; b < mgot; ++b) st.cslots[b] = bc[2 + b];
                    PSD_SYNC();
                }
                if (*okf && mgot >= 2) {
                    m = mgot;
                    st.W = nb + 4;
                    st.tgap = (st.cgap == 1 && nb >= 6) ? 1 : 2;
                    st.train_S = m - 1;
                    st.train_ms = (ms < m) ? ms : m;
                    if (st.mb && st.train_long > m && w >= PSD_TRAIN_LONG_MINW && st.train_want >= PSD_TRAIN_MAX) {
                        // more bulges than slots: the slot of bulge b takes bulge b + S when b is through.  Bulge b + S
                        // enters S s / nb ticks after bulge b, which needs (i - l) / nb + 1 ticks for its sweep and one
                        // to turn around — with two ticks to spare, or the train stays at one bulge per slot
                        const int S = m - 1, spc = (st.tgap == 1) ? (nb + 4) : (2 * nb);
                        const int need = (i - l + nb - 1) / nb + 2;
                        if ((S * spc) / nb >= need) {
                            int want = st.train_long;
                            if (want > w / st.train_wdiv) want = w / st.train_wdiv;  // (a train deflates a fraction of its bulge count: keep several trains per range)
                            if (want > m) m = want;
                        }
                    }
                    st.train_n = m;
                    st.train_tick0 = P.tick + (heavy ? 1 : 0);  // (the leader's first window runs in the next launch)
                    st.train_id += 1;
                    st.ntrains += 1;
                    st.ntrainsweeps += m;
                    for (int b = 1; b < m; ++b) psd_log(P, st, 0, l, i);  // one log entry per bulge of the train
                    rt1r = P.tshift[0];
                    rt1i = P.tshift[1];
                    rt2r = P.tshift[2];
                    rt2i = P.tshift[3];
                }
                PSD_SYNC();
            }
        }
    }
    {  // PSD.jl:768-803 with mmax = l (_allow_early_QR[] is false by default)
        const int m = l;
        const double h11 = hdiag[m], h12 = hsup[m], h21 = hsub[m + 1], h22 = hdiag[m + 1];
        double v1, v2, v3;
        if (exc) {
            const double h44s = h44 - h11, h33s = h33 - h11;
            v1 = (h33s * h44s - h43h34) / h21 + h12;
            v2 = h22 - h11 - h33s - h44s;
            v3 = hsub[m + 2];
        } else {
            const double s = fabs(h11 - rt2r) + fabs(rt2i) + fabs(h21);
            const double h21s = h21 / s;
            v1 = h21s * h12 + (h11 - rt1r) * ((h11 - rt2r) / s) - rt1i * (rt2i / s);
            v2 = h21s * (h11 + h22 - rt1r - rt2r);
            v3 = h21s * hsub[m + 2];
        }
        const double s = fabs(v1) + fabs(v2) + fabs(v3);
        st.v[0] = v1 / s;
        st.v[1] = v2 / s;
        st.v[2] = v3 / s;
    }
    st.phase = PSD_PH_QR;
    st.kcur = l;
    if (st.train_n > 1) {  // the cursors behind the leader start from this state
        PSD_DBG_T0();
        PSD_SYNC();
        PSD_ONE {
            if (!st.mb) psd_atomic_store(P.cep + PSD_TRAIN_MAX, 0);  // finished cursors of this train
            const int ngen1 = (st.mb && st.train_S > 0 && st.train_S + 1 < st.train_n) ? (st.train_S + 1) : st.train_n;
            for (int b = 1; b < ngen1; ++b) {  // (a long train: the first bulge of every slot)
                psd_rstate cs = st;
                cs.cursor = b;
                cs.phase = PSD_PH_CWAIT;
                cs.kcur = 0;
                cs.nsweeps = cs.nwindows = cs.nlog = 0;
                cs.maxlog = 0;
                for (int q = 0; q < 6; ++q) cs.cyc[q] = 0;
                psd_cursor_schedule(st, b, cs.cstart, cs.cfirst);
                if (st.mb) {  // (the slot's workgroup picks the state up in the next launch: epoch < tick)
                    cs.parent = st.slot;
                    cs.slot = st.cslots[b];
                    P.cst[cs.slot] = cs;
                    psd_atomic_store(P.epoch + cs.slot, P.tick);
                } else {
                    psd_pub_begin(P.cep + b);
                    P.cst[b] = cs;
                    psd_pub_end(P.cep + b, P.tick);
                }
            }
        }
        PSD_SYNC();
        PSD_DBG_ADD(5);
    }
    return heavy;
}

PSD_D void psd_desc_write(const psd_rparams& P, psd_rstate& st, const int* lcnt, int plo, int phi, int lc0,
                          int lc1, int rr0, int rr1, int split = 0) {
    PSD_SYNC();
    const bool over = psd_list_overflow(lcnt, st.p, PSD_TR_CAP);
    if (over) {  // never apply truncated lists
        st.info = PSD_LIST_OVERFLOW;
        st.phase = PSD_PH_DONE;
        if (st.mb) {
            PSD_ONE {
                psd_atomic_store(&P.gl->info, PSD_LIST_OVERFLOW);
                psd_atomic_store(&P.gl->abort, 1);
                psd_atomic_store(&P.gl->done, 1);
            }
        }
    }
    PSD_PAR_FOR(m, st.p) { P.cnt[m] = lcnt[m]; }
    PSD_ONE {
        psd_apply_desc d;
        d.active = over ? 0 : 1;
        d.prob = st.prob;
        d.split = (split && st.wantT) ? 1 : 0;
        {   // first far column of the rows role (psd_rdefer_edge)
            int cc = phi + 1 + ((P.redge > 0) ? P.redge : psd_rdefer_edge(st.Wmax));
            if (cc < lc0) cc = lc0;
            if (cc > lc1 + 1) cc = lc1 + 1;
            d.cut = cc;
        }
        {   // first near row of the column role (psd_cdefer_edge): rows rr0 .. rcut - 1 may run one tick late
            int rc = plo - ((P.cedge > 0) ? P.cedge : psd_cdefer_edge(st.Wmax));
            if (rc < rr0) rc = rr0;
            if (rc > rr1 + 1) rc = rr1 + 1;
            d.rcut = rc;
        }
        d.plo = plo;
        d.phi = phi;
        d.lc0 = lc0;
        d.lc1 = lc1;
        d.rr0 = rr0;
        d.rr1 = rr1;
        d.zr0 = 1;
        d.zr1 = st.wantZ ? st.n : 0;
        *P.desc = d;
    }
    PSD_SYNC();
}

// Hot micro-step of the sweep, (k, factor j >= 2), 3-wide case (PSD.jl:844-883): 3-reflector from
// column k of H_j, then the 2-reflector from column k+1, both applied to rows of H_j (lanes =
// columns k+1..c1max) and to columns of H_{j-1} (lanes = rows r0..rlim) inside the window.
//   * one wave, ONE code path: a lane addresses its three operands as base + {0,1,2}*stride
//     (column lanes: stride 1 in H_j; row lanes: stride ld in H_{j-1}; the extra lane that writes
//     the annihilated column (beta,0,0) is folded in with selects) — no divergence, because with
//     a single resident wave every extra VALU instruction is ~8 issue cycles on the serial chain;
//   * operands are loaded from LDS once, BEFORE the reflector arithmetic (latency hidden), stay
//     in registers across both reflectors and are stored once;
//   * the values the chain needs next (column k+1 of H_j for the 2-reflector, column k of H_{j-1}
//     for the next factor) travel by v_readlane, not through LDS.
// (x0,x1,x2) in: H_j[k..k+2,k]; out: H_{j-1}[k..k+2,k].  lane_off/lane_str: per-lane operand
// offset inside a factor block / stride, lane_adj: 0 for column lanes, bsz for row lanes.
PSD_D void psd_qr_micro3(const psd_rparams& P, const psd_win& w, int j, int nl, int cnt, int lk, int k,
                         PSD_LANEVAR_REF(int, lane_off), PSD_LANEVAR_REF(int, lane_str),
                         PSD_LANEVAR_REF(int, lane_adj), double& x0, double& x1, double& x2, double& tau_io,
                         bool more, int slot) {
    // (x0, x1, x2, tau_io) in: the 3-reflector of this factor, already generated: (beta, v2, v3, tau);
    // out (if `more`): the 3-reflector of the next factor, generated here from H_{j-1}[k..k+2, k] while the
    // 2-reflector of this factor is being generated and applied — two independent dependency chains in one
    // basic block, which the scheduler interleaves (a lone wavefront is otherwise stalled on each chain's latency)
    PSD_LANEVAR(double, a1);
    PSD_LANEVAR(double, a2);
    PSD_LANEVAR(double, a3);
    const int boff = (j - 1) * w.bsz;
    PSD_PAR_ONCE(t, cnt) {
        const double* q = w.b + (boff - PSD_LV(lane_adj) + PSD_LV(lane_off));
        const int sd = PSD_LV(lane_str);
        PSD_LV(a1) = q[0];
        PSD_LV(a2) = q[sd];
        PSD_LV(a3) = q[2 * sd];
    }
    const double tau = tau_io;
    const double beta = x0, v2 = x1, v3 = x2;
    PSD_PAR_ONCE(t, cnt) {
        const double x = tau * (PSD_LV(a1) + v2 * PSD_LV(a2) + v3 * PSD_LV(a3));
        PSD_LV(a1) -= x;
        PSD_LV(a2) -= x * v2;
        PSD_LV(a3) -= x * v3;
    }
    double y0 = PSD_BCAST(a2, 0), y1 = PSD_BCAST(a3, 0);  // H_j[k+1..k+2, k+1]
    x0 = PSD_BCAST(a1, lk);                               // H_{j-1}[k..k+2, k] for the next factor
    x1 = PSD_BCAST(a1, lk + 1);
    x2 = PSD_BCAST(a1, lk + 2);
    double tau2;
    psd_refl32_pair(x0, x1, x2, tau_io, y0, y1, tau2);  // (for the last factor the 3-reflector is simply not used)
    const double beta2 = y0, w2 = y1;
    PSD_PAR_ONCE(t, cnt) {
        const double x = tau2 * (PSD_LV(a2) + w2 * PSD_LV(a3));
        double b1 = PSD_LV(a1), b2 = PSD_LV(a2) - x, b3 = PSD_LV(a3) - x * w2;
        if (t == 0) {  // column k+1 of H_j: annihilated by the 2-reflector
            b2 = beta2;
            b3 = 0.0;
        }
        if (t == cnt - 1) {  // column k of H_j: annihilated by the 3-reflector
            b1 = beta;
            b2 = 0.0;
            b3 = 0.0;
        }
        double* q = w.b + (boff - PSD_LV(lane_adj) + PSD_LV(lane_off));
        const int sd = PSD_LV(lane_str);
        q[0] = b1;
        q[sd] = b2;
        q[2 * sd] = b3;
        if (t == cnt - 1) {
            psd_tr tr;
            tr.pos = k;
            tr.kind = PSD_TR_R3;
            tr.c0 = v2;
            tr.c1 = v3;
            tr.c2 = tau;
            if (slot < PSD_TR_CAP) P.tr[(size_t)(j - 1) * PSD_TR_CAP + slot] = tr;
            tr.pos = k + 1;
            tr.kind = PSD_TR_H2;
            tr.c0 = w2;
            tr.c1 = 0.0;
            tr.c2 = tau2;
            if (slot + 1 < PSD_TR_CAP) P.tr[(size_t)(j - 1) * PSD_TR_CAP + slot + 1] = tr;
        }
    }
    PSD_WAVE_SYNC();
}

// H_1's 3-reflector at step k (PSD.jl:812-842), p > 1: rows of H_1 (columns k..c1max), columns of
// H_p (rows r0..rlim).  (x0,x1,x2) in: the vector to reflect; out: H_p[k..k+2, k].
PSD_D void psd_qr_micro_h1(const psd_rparams& P, const psd_win& w, int p, int k, int c1max, int r0, int rlim,
                           double& x0, double& x1, double& x2, bool fix, int slot) {
    const int nl = c1max - k + 1;  // columns k..c1max
    const int nrw = rlim - r0 + 1;
    const int cnt = nl + nrw + 1;
    PSD_LANEVAR(double, a1);
    PSD_LANEVAR(double, a2);
    PSD_LANEVAR(double, a3);
    PSD_PAR_ONCE(t, cnt) {
        if (t < nl) {
            const int c = k + t;
            PSD_LV(a1) = w.at(1, k, c);
            PSD_LV(a2) = w.at(1, k + 1, c);
            PSD_LV(a3) = w.at(1, k + 2, c);
        } else if (t < nl + nrw) {
            const int r = r0 + (t - nl);
            PSD_LV(a1) = w.at(p, r, k);
            PSD_LV(a2) = w.at(p, r, k + 1);
            PSD_LV(a3) = w.at(p, r, k + 2);
        } else {
            PSD_LV(a1) = 0.0;
            PSD_LV(a2) = 0.0;
            PSD_LV(a3) = 0.0;
        }
    }
    const double tau = psd_refl3(x0, x1, x2);
    const double beta = x0, v2 = x1, v3 = x2;
    PSD_PAR_ONCE(t, cnt) {
        if (t < nl + nrw) {
            const double x = tau * (PSD_LV(a1) + v2 * PSD_LV(a2) + v3 * PSD_LV(a3));
            PSD_LV(a1) -= x;
            PSD_LV(a2) -= x * v2;
            PSD_LV(a3) -= x * v3;
            if (t < nl) {
                const int c = k + t;
                w.at(1, k, c) = PSD_LV(a1);
                w.at(1, k + 1, c) = PSD_LV(a2);
                w.at(1, k + 2, c) = PSD_LV(a3);
            } else {
                const int r = r0 + (t - nl);
                w.at(p, r, k) = PSD_LV(a1);
                w.at(p, r, k + 1) = PSD_LV(a2);
                w.at(p, r, k + 2) = PSD_LV(a3);
            }
        } else {
            if (fix) {
                w.at(1, k, k - 1) = beta;
                w.at(1, k + 1, k - 1) = 0.0;
                w.at(1, k + 2, k - 1) = 0.0;
            }
            psd_tr tr;
            tr.pos = k;
            tr.kind = PSD_TR_R3;
            tr.c0 = v2;
            tr.c1 = v3;
            tr.c2 = tau;
            if (slot < PSD_TR_CAP) P.tr[slot] = tr;
        }
    }
    const int lk = nl + (k - r0);
    x0 = PSD_BCAST(a1, lk);
    x1 = PSD_BCAST(a1, lk + 1);
    x2 = PSD_BCAST(a1, lk + 2);
    PSD_WAVE_SYNC();
}

// In-window application of a 3- or 2-reflector with v = (1, v2, v3) (general, slower form used for
// the last step of a sweep and for p == 1): from the left to rows q.. of factor jl over columns
// [c0,c1], from the right to columns q.. of factor jr (!= jl) over rows [r0,r1]; one extra lane
// writes the annihilated column (beta, 0, 0) at (q.., fc) of factor jf and appends the transform
// to owner `own`'s list.
PSD_D void psd_win_reflect(const psd_rparams& P, const psd_win& w, int jl, int jr, int len, int q, double v2,
                           double v3, double tau, int c0, int c1, int r0, int r1, int jf, int fc, double beta,
                           bool fix, int own, int slot) {
    if (c1 > w.be) c1 = w.be;
    if (r0 < w.bs) r0 = w.bs;
    if (r1 > w.be) r1 = w.be;
    const int nl = (c1 >= c0) ? (c1 - c0 + 1) : 0;
    const int nr = (r1 >= r0) ? (r1 - r0 + 1) : 0;
    PSD_PAR_FOR(t, nl + nr + 1) {
        if (t < nl) {
            const int c = c0 + t;
            if (len == 3) {
                double a1 = w.at(jl, q, c), a2 = w.at(jl, q + 1, c), a3 = w.at(jl, q + 2, c);
                const double x = tau * (a1 + v2 * a2 + v3 * a3);
                w.at(jl, q, c) = a1 - x;
                w.at(jl, q + 1, c) = a2 - x * v2;
                w.at(jl, q + 2, c) = a3 - x * v3;
            } else {
                double a1 = w.at(jl, q, c), a2 = w.at(jl, q + 1, c);
                const double x = tau * (a1 + v2 * a2);
                w.at(jl, q, c) = a1 - x;
                w.at(jl, q + 1, c) = a2 - x * v2;
            }
        } else if (t < nl + nr) {
            const int r = r0 + (t - nl);
            if (len == 3) {
                double a1 = w.at(jr, r, q), a2 = w.at(jr, r, q + 1), a3 = w.at(jr, r, q + 2);
                const double x = tau * (a1 + v2 * a2 + v3 * a3);
                w.at(jr, r, q) = a1 - x;
                w.at(jr, r, q + 1) = a2 - x * v2;
                w.at(jr, r, q + 2) = a3 - x * v3;
            } else {
                double a1 = w.at(jr, r, q), a2 = w.at(jr, r, q + 1);
                const double x = tau * (a1 + v2 * a2);
                w.at(jr, r, q) = a1 - x;
                w.at(jr, r, q + 1) = a2 - x * v2;
            }
        } else {
            if (fix) {
                w.at(jf, q, fc) = beta;
                w.at(jf, q + 1, fc) = 0.0;
                if (len == 3) w.at(jf, q + 2, fc) = 0.0;
            }
            psd_tr tr;
            tr.pos = q;
            tr.kind = (len == 3) ? PSD_TR_R3 : PSD_TR_H2;
            tr.c0 = v2;
            tr.c1 = v3;
            tr.c2 = tau;
            if (slot < PSD_TR_CAP) P.tr[(size_t)(own - 1) * PSD_TR_CAP + slot] = tr;
        }
    }
    PSD_SYNC();
}

// ------------------------------------------------------------------------------------------------
// Two-wave chase (round 3).  The sweep's chain has two strands per position k (PSD.jl:844-883): the 3-reflectors
// (H_1, then factors p..2: each from column k of H_j as the previous one's right update left it) and the 2-reflectors
// (factors p..2: each from column k+1 of H_j after that factor's 3-reflector from the left and the previous factor's
// 2-reflector from the right).  The 3-strand never reads anything a 2-reflector of the same position writes (column k
// against columns k+1, k+2), so the two strands are two serial chains of the same length that only have to stay in
// order where they touch the same factor from the same side.  psd_qr_micro3 runs both in one wavefront (~165
// instructions per (position, factor), ~900 cycles: the wave is bound by its own issue rate, not by the dependency
// chain).  Here wavefront A of a chase workgroup runs the 3-strand and wavefront B the 2-strand THREE links behind, one
// hardware barrier per link:
//   link q = (position k, t): t = 0 the H_1 step, t >= 1 factor j = p + 1 - t.  Step s: A does link s, B link s - 3.
//   * same step, disjoint data: A(k,j) touches rows k..k+2 of H_j and columns k..k+2 of H_{j-1}, B(k,j+3) rows of
//     H_{j+3} and columns of H_{j+2}; across the position wrap the factors differ as long as p >= 6 (p >= 8 is asked);
//   * order where it matters (same factor, same side): A(k,j) before B(k,j) on rows of H_j and on columns of H_{j-1};
//     B(k,j) before A(k+1,j) on both (p - 3 steps later);
//   * a left and a right transformation of the same block commute, so B(k,j+1)'s column update of H_j may follow
//     A(k,j)'s row update of it (the reference has them the other way round): same result up to rounding;
//   * three links of lag (two would do for the hazards) let each wave fetch the operands of its NEXT link from LDS
//     before the barrier, so no LDS latency sits on either chain: what B(k,j) reads was last written by A(k,j-1), one
//     step before the step in which B prefetches it.
// The helper wavefront (threadIdx.y = 1 of a 64 x 2 block) parks at the pair barrier between runs; code written in terms
// of PSD_TID / PSD_SYNC never sees it.  The simulated tier runs A and B of a step one after the other.
struct psd_c2 {
    int cmd;                  // 0: the workgroup is done (helper leaves), 1: a run follows
    int ld, bsz, bs, be;      // window image
    int p, l, i, ks, npos;    // positions ks .. ks + npos - 1, all with three-row bulges (k <= i - 2)
    int c1max, r0;
    int n1, nj;               // list slots of the first position: owner 1, owners 2..p
    int wboff;                // window image: byte offset in dynamic LDS (a generic pointer would make the helper's
                              // LDS traffic FLAT instructions)
    psd_tr* tr;               // this slot's lists
    double v0, v1, v2;        // the sweep's start vector (used when ks == l)
    long long* dbg;           // diagnostics (scan chase): psd_rglobal::c3dbg or nullptr
    double* H;                // commands 3 / 4 (window load / store shared by the wavefronts): factors, order, window width
    int n, W;
    // factor-sliced scan chase (psd_slice3.h; command 5): slices, this workgroup's slice, the tick, the slot's inboxes,
    // the error word of the bounded waits
    int slG, slg, sltick;
    unsigned char* slbox;
    int* slerr;
};
#define PSD_C2_MINP 8
#ifndef PSD_C2_STAMP
#define PSD_C2_STAMP(i) ((void)0)  // (tools/micro/c2_bench.py: in-loop cycle stamps of the timing harness)
#endif
#define PSD_C2_LAG 3
#ifdef PSD_HOSTSIM
#define PSD_C2_UNI(x) (x)
#else
#define PSD_C2_UNI(x) __builtin_amdgcn_readfirstlane(x)
#endif

// roles: bit 0 = A, bit 1 = B (a wavefront runs one; the simulated tier both).  Lane classes of a link: 0 = an operand
// triple / pair that takes the reflector, 1 = the lane that writes the annihilated column and the record, 2 = idle.
// Operand addresses are fixed per position up to the factor's block: off + (j - 1) bsz (the right lanes carry -bsz).
//
// A lone wavefront hides nothing: every taken branch is a fetch bubble, every dependent instruction waits out its
// predecessor.  So the steps are straight-line blocks (selects and one masked store instead of per-class branches), a
// step starts with its LDS loads and generates its reflector while they are in flight, both strands hand the next
// link's vector over in registers (lane broadcasts from the lanes that just updated it) instead of reading it back from
// LDS, and the loops over the factors carry no position logic.  Steps: s = 0 .. L + LAG - 1, a barrier in front of each
// and one behind the last; A works in steps [0, L), B (link s - LAG) in [LAG, L + LAG).
PSD_D void psd_c2_run(const psd_c2& Cin, int roles_) {
    PSD_LDS_DECL;
    const int roles = PSD_C2_UNI(roles_);
    // (wave-uniform words: scalar registers on the device)
    const int p = PSD_C2_UNI(Cin.p), ld = PSD_C2_UNI(Cin.ld), bsz = PSD_C2_UNI(Cin.bsz), bs = PSD_C2_UNI(Cin.bs);
    const int l = PSD_C2_UNI(Cin.l), ie = PSD_C2_UNI(Cin.i), ks = PSD_C2_UNI(Cin.ks), npos = PSD_C2_UNI(Cin.npos);
    const int c1max = PSD_C2_UNI(Cin.c1max), r0 = PSD_C2_UNI(Cin.r0), n1 = PSD_C2_UNI(Cin.n1), nj = PSD_C2_UNI(Cin.nj);
    double* const wb = (double*)(psd_lds + PSD_C2_UNI(Cin.wboff));
    psd_tr* const trb = Cin.tr;
    const int L = npos * p;
    // ---- A
    double x0 = 0.0, x1 = 0.0, x2 = 0.0;
    PSD_LANEVAR(double, a1);
    PSD_LANEVAR(double, a2);
    PSD_LANEVAR(double, a3);
    PSD_LANEVAR(int, offH);  // H_1 step of the current position: absolute offset, stride, class
    PSD_LANEVAR(int, strH);
    PSD_LANEVAR(int, clsH);
    PSD_LANEVAR(int, offF);  // factor steps
    PSD_LANEVAR(int, strF);
    PSD_LANEVAR(int, clsF);
    int lkH = 0, lkF = 0;
    // ---- B
    double y0 = 0.0, y1 = 0.0;
    PSD_LANEVAR(double, b2);
    PSD_LANEVAR(double, b3);
    PSD_LANEVAR(int, offB);
    PSD_LANEVAR(int, strB);
    PSD_LANEVAR(int, clsB);
    int flB = 0, lyB = 0;
#define PSD_C2_SETUP_A(k_)                                                                             \
    do {                                                                                               \
        const int k__ = (k_);                                                                          \
        const int rlim__ = (k__ + 3 < ie) ? (k__ + 3) : ie;                                            \
        const int nrw__ = rlim__ - r0 + 1;                                                             \
        int nlh__ = c1max - k__ + 1, nlf__ = c1max - k__;                                              \
        if (nlh__ < 0) nlh__ = 0;                                                                      \
        if (nlf__ < 0) nlf__ = 0;                                                                      \
        lkH = nlh__ + (k__ - r0);                                                                      \
        lkF = nlf__ + (k__ - r0);                                                                      \
        PSD_PAR_ALL64(t) {                                                                 \
            if (t < nlh__) {                                                                           \
                PSD_LV(offH) = (k__ + t - bs) * ld + (k__ - bs);                                       \
                PSD_LV(strH) = 1;                                                                      \
                PSD_LV(clsH) = 0;                                                                      \
            } else if (t < nlh__ + nrw__) {                                                            \
                PSD_LV(offH) = (p - 1) * bsz + (k__ - bs) * ld + (r0 + (t - nlh__) - bs);              \
                PSD_LV(strH) = ld;                                                                     \
                PSD_LV(clsH) = 0;                                                                      \
            } else {                                                                                   \
                PSD_LV(offH) = (k__ - 1 - bs) * ld + (k__ - bs);                                       \
                PSD_LV(strH) = 1;                                                                      \
                PSD_LV(clsH) = (t == nlh__ + nrw__) ? 1 : 2;                                           \
            }                                                                                          \
            if (t < nlf__) {                                                                           \
                PSD_LV(offF) = (k__ + 1 + t - bs) * ld + (k__ - bs);                                   \
                PSD_LV(strF) = 1;                                                                      \
                PSD_LV(clsF) = 0;                                                                      \
            } else if (t < nlf__ + nrw__) {                                                            \
                PSD_LV(offF) = (k__ - bs) * ld + (r0 + (t - nlf__) - bs) - bsz;                        \
                PSD_LV(strF) = ld;                                                                     \
                PSD_LV(clsF) = 0;                                                                      \
            } else {                                                                                   \
                PSD_LV(offF) = (k__ - bs) * ld + (k__ - bs);                                           \
                PSD_LV(strF) = 1;                                                                      \
                PSD_LV(clsF) = (t == nlf__ + nrw__) ? 1 : 2;                                           \
            }                                                                                          \
        }                                                                                              \
    } while (0)
#define PSD_C2_SETUP_B(k_)                                                                             \
    do {                                                                                               \
        const int k__ = (k_);                                                                          \
        const int rlim__ = (k__ + 3 < ie) ? (k__ + 3) : ie;                                            \
        const int nrw__ = rlim__ - r0 + 1;                                                             \
        int nlb__ = c1max - (k__ + 2) + 1;                                                             \
        if (nlb__ < 0) nlb__ = 0;                                                                      \
        flB = nlb__ + nrw__;                                                                           \
        lyB = nlb__ + (k__ + 1 - r0);                                                                  \
        PSD_PAR_ALL64(t) {                                                                 \
            if (t < nlb__) {                                                                           \
                PSD_LV(offB) = (k__ + 2 + t - bs) * ld + (k__ + 1 - bs);                               \
                PSD_LV(strB) = 1;                                                                      \
                PSD_LV(clsB) = 0;                                                                      \
            } else if (t < nlb__ + nrw__) {                                                            \
                PSD_LV(offB) = (k__ + 1 - bs) * ld + (r0 + (t - nlb__) - bs) - bsz;                    \
                PSD_LV(strB) = ld;                                                                     \
                PSD_LV(clsB) = 0;                                                                      \
            } else {                                                                                   \
                PSD_LV(offB) = (k__ + 1 - bs) * ld + (k__ + 1 - bs);                                   \
                PSD_LV(strB) = 1;                                                                      \
                PSD_LV(clsB) = (t == nlb__ + nrw__) ? 1 : 2;                                           \
            }                                                                                          \
        }                                                                                              \
    } while (0)
    // one link of A.  OFF / STR / CLS: the lane variables of the layout; JB: block offset added to OFF; JIDX: the factor
    // that is updated from the left (list owner); FIXON: the class-1 lane writes its column; SLOT: list slot; LK: lane of
    // the first of the three rows of the next link's vector.  (a1, a2, a3) <- operands; (x0, x1, x2) in: the vector to
    // reflect, out: the next link's.
#define PSD_C2_A_STEP(OFF, STR, CLS, JB, JIDX, FIXON, KPOS, SLOT, LK)                                  \
    do {                                                                                               \
        PSD_PAR_ALL64(t) {                                                                 \
            PSD_LV(a1) = PSD_LV(a2) = PSD_LV(a3) = 0.0;                                                \
            if (PSD_LV(CLS) == 0) {                                                                    \
                const double* q = wb + (PSD_LV(OFF) + (JB));                                           \
                const int sd = PSD_LV(STR);                                                            \
                PSD_LV(a1) = q[0];                                                                     \
                PSD_LV(a2) = q[sd];                                                                    \
                PSD_LV(a3) = q[2 * sd];                                                                \
            }                                                                                          \
        }                                                                                              \
        PSD_C2_STAMP(0);                                                                               \
        const double tau__ = psd_refl3_lean(x0, x1, x2);                                               \
        PSD_C2_STAMP(1);                                                                               \
        const double beta__ = x0, v2__ = x1, v3__ = x2;                                                \
        PSD_PAR_ALL64(t) {                                                                 \
            const double xx = tau__ * (PSD_LV(a1) + v2__ * PSD_LV(a2) + v3__ * PSD_LV(a3));            \
            PSD_LV(a1) -= xx;                                                                          \
            PSD_LV(a2) -= xx * v2__;                                                                   \
            PSD_LV(a3) -= xx * v3__;                                                                   \
            if (PSD_LV(CLS) == 1) {                                                                    \
                PSD_LV(a1) = beta__;                                                                   \
                PSD_LV(a2) = 0.0;                                                                      \
                PSD_LV(a3) = 0.0;                                                                      \
            }                                                                                          \
        }                                                                                              \
        PSD_C2_STAMP(2);                                                                               \
        x0 = PSD_BCAST(a1, (LK));                                                                      \
        x1 = PSD_BCAST(a1, (LK) + 1);                                                                  \
        x2 = PSD_BCAST(a1, (LK) + 2);                                                                  \
        PSD_C2_STAMP(3);                                                                               \
        PSD_PAR_ALL64(t) {                                                                 \
            const int cls = PSD_LV(CLS);                                                               \
            double* q = wb + (PSD_LV(OFF) + (JB));                                                     \
            const int sd = PSD_LV(STR);                                                                \
            if (cls == 0 || (cls == 1 && (FIXON))) {                                                   \
                q[0] = PSD_LV(a1);                                                                     \
                q[sd] = PSD_LV(a2);                                                                    \
                q[2 * sd] = PSD_LV(a3);                                                                \
            }                                                                                          \
            if (cls == 1 && (SLOT) < PSD_TR_CAP) {                                                     \
                psd_tr tr;                                                                             \
                tr.pos = (KPOS);                                                                       \
                tr.kind = PSD_TR_R3;                                                                   \
                tr.c0 = v2__;                                                                          \
                tr.c1 = v3__;                                                                          \
                tr.c2 = tau__;                                                                         \
                psd_tr_store_global(trb + (size_t)((JIDX) - 1) * PSD_TR_CAP + (SLOT), tr);             \
            }                                                                                          \
        }                                                                                              \
        PSD_C2_STAMP(4);                                                                               \
    } while (0)
    // one link of B (factor JIDX).  FIRST: the vector comes from the class-1 lane's operands (first factor of a
    // position); otherwise (y0, y1) was handed over by the previous link.  Out: the next factor's vector, column k + 1
    // of H_{j-1} on rows k+1, k+2 as this link's column update left it (lanes LY, LY + 1 of the right lanes).
#define PSD_C2_B_STEP(JB, JIDX, FIRST, KPOS, SLOT)                                                     \
    do {                                                                                               \
        PSD_PAR_ALL64(t) {                                                                 \
            PSD_LV(b2) = PSD_LV(b3) = 0.0;                                                             \
            if (PSD_LV(clsB) != 2) {                                                                   \
                const double* q = wb + (PSD_LV(offB) + (JB));                                          \
                const int sd = PSD_LV(strB);                                                           \
                PSD_LV(b2) = q[0];                                                                     \
                PSD_LV(b3) = q[sd];                                                                    \
            }                                                                                          \
        }                                                                                              \
        if (FIRST) {                                                                                   \
            y0 = PSD_BCAST(b2, flB);                                                                   \
            y1 = PSD_BCAST(b3, flB);                                                                   \
        }                                                                                              \
        const double tau2__ = psd_refl2_lean(y0, y1);                                                  \
        const double beta2__ = y0, w2__ = y1;                                                          \
        PSD_PAR_ALL64(t) {                                                                 \
            const double xx = tau2__ * (PSD_LV(b2) + w2__ * PSD_LV(b3));                               \
            PSD_LV(b2) -= xx;                                                                          \
            PSD_LV(b3) -= xx * w2__;                                                                   \
            if (PSD_LV(clsB) == 1) {                                                                   \
                PSD_LV(b2) = beta2__;                                                                  \
                PSD_LV(b3) = 0.0;                                                                      \
            }                                                                                          \
        }                                                                                              \
        y0 = PSD_BCAST(b2, lyB);                                                                       \
        y1 = PSD_BCAST(b2, lyB + 1);                                                                   \
        PSD_PAR_ALL64(t) {                                                                 \
            const int cls = PSD_LV(clsB);                                                              \
            double* q = wb + (PSD_LV(offB) + (JB));                                                    \
            const int sd = PSD_LV(strB);                                                               \
            if (cls != 2) {                                                                            \
                q[0] = PSD_LV(b2);                                                                     \
                q[sd] = PSD_LV(b3);                                                                    \
            }                                                                                          \
            if (cls == 1 && (SLOT) < PSD_TR_CAP) {                                                     \
                psd_tr tr;                                                                             \
                tr.pos = (KPOS) + 1;                                                                   \
                tr.kind = PSD_TR_H2;                                                                   \
                tr.c0 = w2__;                                                                          \
                tr.c1 = 0.0;                                                                           \
                tr.c2 = tau2__;                                                                        \
                psd_tr_store_global(trb + (size_t)((JIDX) - 1) * PSD_TR_CAP + (SLOT), tr);             \
            }                                                                                          \
        }                                                                                              \
    } while (0)
    if (roles & 1) {
        if (ks > l) {
            x0 = wb[(ks - 1 - bs) * ld + (ks - bs)];
            x1 = wb[(ks - 1 - bs) * ld + (ks + 1 - bs)];
            x2 = wb[(ks - 1 - bs) * ld + (ks + 2 - bs)];
        } else {
            x0 = Cin.v0;
            x1 = Cin.v1;
            x2 = Cin.v2;
        }
    }
#ifndef PSD_HOSTSIM
    if (roles == 1) {
        for (int kk = 0; kk < npos; ++kk) {
            const int k = ks + kk;
            PSD_C2_SETUP_A(k);
            PSD_PAIR_BARRIER_BARE();
            PSD_C2_A_STEP(offH, strH, clsH, 0, 1, k > l, k, n1 + kk, lkH);
            int jb = (p - 1) * bsz;
            const int slot = nj + 2 * kk;
            for (int j = p; j > 2; --j) {
                PSD_PAIR_BARRIER_BARE();
                PSD_C2_A_STEP(offF, strF, clsF, jb, j, true, k, slot, lkF);
                jb -= bsz;
            }
            // (factor 2: the next vector is column k of H_1 on rows k+1 .. k+3, for the H_1 step of position k + 1;
            //  behind the last position of the run nobody uses it, and row k + 3 may not exist: the lane index stays
            //  inside the wavefront, the values are not looked at)
            PSD_PAIR_BARRIER_BARE();
            PSD_C2_A_STEP(offF, strF, clsF, jb, 2, true, k, slot, (lkF + 1));
        }
        for (int e = 0; e < PSD_C2_LAG + 1; ++e) PSD_PAIR_BARRIER();
    } else {
        for (int e = 0; e < PSD_C2_LAG; ++e) PSD_PAIR_BARRIER_BARE();
        for (int kk = 0; kk < npos; ++kk) {
            const int k = ks + kk;
            PSD_C2_SETUP_B(k);
            PSD_PAIR_BARRIER_BARE();  // (the slot of the H_1 step: no 2-reflector)
            int jb = (p - 1) * bsz;
            const int slot = nj + 2 * kk + 1;
            PSD_PAIR_BARRIER_BARE();
            PSD_C2_B_STEP(jb, p, true, k, slot);
            jb -= bsz;
            for (int j = p - 1; j >= 2; --j) {
                PSD_PAIR_BARRIER_BARE();
                PSD_C2_B_STEP(jb, j, false, k, slot);
                jb -= bsz;
            }
        }
        PSD_PAIR_BARRIER();
    }
#else
    // the simulated tier: the same steps in the order of the global step counter (A's link s, then B's link s - LAG)
    for (int s = 0; s < L + PSD_C2_LAG; ++s) {
        if ((roles & 1) && s < L) {
            const int kk = s / p, tq = s - kk * p, k = ks + kk;
            if (tq == 0) {
                PSD_C2_SETUP_A(k);
                PSD_C2_A_STEP(offH, strH, clsH, 0, 1, k > l, k, n1 + kk, lkH);
            } else {
                const int j = p + 1 - tq;
                PSD_C2_A_STEP(offF, strF, clsF, (j - 1) * bsz, j, true, k, nj + 2 * kk, (lkF + ((j == 2) ? 1 : 0)));
            }
        }
        const int q = s - PSD_C2_LAG;
        if ((roles & 2) && q >= 0) {
            const int kk = q / p, tq = q - kk * p, k = ks + kk;
            if (tq == 0) {
                PSD_C2_SETUP_B(k);
            } else {
                const int j = p + 1 - tq;
                PSD_C2_B_STEP((j - 1) * bsz, j, tq == 1, k, nj + 2 * kk + 1);
            }
        }
    }
#endif
    (void)L;
#undef PSD_C2_SETUP_A
#undef PSD_C2_SETUP_B
#undef PSD_C2_A_STEP
#undef PSD_C2_B_STEP
}

// wavefront A's side of a run: publish it, run it (the helper runs B), leave the helper parked
PSD_D void psd_c2_lead(const psd_rparams& P, psd_c2& C) {
#ifdef PSD_HOSTSIM
    (void)P;
    psd_c2_run(C, 3);
#else
    PSD_LDS_DECL;
    psd_c2* cmd = (psd_c2*)(psd_lds + P.c2off);
    C.cmd = 1;
    PSD_ONE { *cmd = C; }
    PSD_PAIR_BARRIER();
    psd_c2_run(C, 1);
    // (the helper is back at its command barrier or on its way there; the command word it finds if this wavefront
    //  ever left without a word is "done")
    PSD_ONE { cmd->cmd = 0; }
#endif
}
#ifndef PSD_HOSTSIM
// the helper wavefront of a chase workgroup (threadIdx.y == 1)
PSD_D void psd_c3_run(const psd_c2& Cin, int wv_, int nw_, int taboff_);
PSD_D void psd_c3s_run(const psd_c2& Cin, int wv_, int nw_, int taboff_);
PSD_D void psd_c2_helper(int c2off, int c3off) {
    PSD_LDS_DECL;
    const psd_c2* cmd = (const psd_c2*)(psd_lds + c2off);
    for (;;) {
        PSD_PAIR_BARRIER();
        const psd_c2 C = *cmd;
        if (C.cmd == 0) return;
        if (C.cmd == 2) {
            psd_c3_run(C, PSD_WAVE_ROLE, (int)blockDim.y, c3off);  // scan chase: every wavefront takes part
        } else if (C.cmd == 5) {
            psd_c3s_run(C, PSD_WAVE_ROLE, (int)blockDim.y, c3off);  // this workgroup's slice of a factor-sliced run
        } else if (C.cmd == 3 || C.cmd == 4) {  // this wavefront's share of a window load / store
            psd_rparams R;
            R.H = C.H;
            psd_win w;
            w.b = (double*)(psd_lds + C.wboff);
            w.W = C.W; w.ld = C.ld; w.bsz = C.bsz; w.bs = C.bs; w.be = C.be;
            if (C.cmd == 3) psd_win_load(R, w, C.n, C.p, PSD_WAVE_ROLE, (int)blockDim.y);
            else psd_win_store(R, w, C.n, C.p, PSD_WAVE_ROLE, (int)blockDim.y);
            PSD_PAIR_BARRIER();
        } else {
            psd_c2_run(C, 2);  // two-wave chase: 64 x 2 workgroups
        }
    }
}
// wavefront A, last thing before it leaves the kernel
PSD_D void psd_c2_release(int c2off) {
    PSD_LDS_DECL;
    psd_c2* cmd = (psd_c2*)(psd_lds + c2off);
    PSD_ONE { cmd->cmd = 0; }
    PSD_PAIR_BARRIER();
}
#endif

#include "psd_chase3.h"
#include "psd_slice3.h"
// wavefront 0's side of a scan-chase run (the other wavefronts of the workgroup wait at the command barrier)
PSD_D void psd_c3_lead(const psd_rparams& P, psd_c2& C) {
#ifdef PSD_HOSTSIM
    psd_c3_run(C, 0, 1, P.c3off);
#else
    PSD_LDS_DECL;
    psd_c2* cmd = (psd_c2*)(psd_lds + P.c2off);
    C.cmd = 2;
    PSD_ONE { *cmd = C; }
    PSD_PAIR_BARRIER();
    psd_c3_run(C, 0, (int)blockDim.y, P.c3off);
    PSD_ONE { cmd->cmd = 0; }
#endif
}

// a window load / store by all wavefronts of a scan-chase workgroup, each its share of the factors (wavefront 0's side)
PSD_D void psd_c3_winio(const psd_rparams& P, const psd_win& w, int n, int p, bool store) {
#ifndef PSD_HOSTSIM
    if (P.c3off != 0 && blockDim.y > 1) {
        PSD_LDS_DECL;
        psd_c2* cmd = (psd_c2*)(psd_lds + P.c2off);
        PSD_SYNC();
        PSD_ONE {
            psd_c2 C;
            C.cmd = store ? 4 : 3;
            C.ld = w.ld; C.bsz = w.bsz; C.bs = w.bs; C.be = w.be; C.W = w.W;
            C.p = p; C.n = n; C.H = P.H;
            C.wboff = (int)((char*)w.b - (char*)psd_lds);
            *cmd = C;
        }
        PSD_PAIR_BARRIER();
        if (store) psd_win_store(P, w, n, p, 0, (int)blockDim.y);
        else psd_win_load(P, w, n, p, 0, (int)blockDim.y);
        PSD_PAIR_BARRIER();
        PSD_ONE { cmd->cmd = 0; }
        return;
    }
#endif
    if (store) psd_win_store(P, w, n, p);
    else psd_win_load(P, w, n, p);
}

#ifndef PSD_HOSTSIM
// ---- factor-sliced runs (psd_slice3.h): a slot's command block and inboxes in device memory
PSD_D unsigned char* psd_sl_cmd(const psd_rparams& P, int slot) {
    return P.slmem + (size_t)slot * (PSD_SL_CMD_BYTES + PSD_SL_MAXG * PSD_SL_BOX_BYTES);
}
// slice 0 publishes the tick's command to the slot's workers: a run (the psd_c2 block behind a release fence) or nothing
PSD_D void psd_sl_publish(const psd_rparams& P, int slot, const psd_c2* C) {  // (one lane)
    unsigned char* cm = psd_sl_cmd(P, slot);
    if (C != nullptr) {
        *(psd_c2*)(cm + 64) = *C;
        psd_release_fence();
    }
    __hip_atomic_store((unsigned long long*)cm, psd_sl_tag(P.tick, 0, (C != nullptr) ? 1 : 4), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
// slice 0, at the end of its launch: workers of a slot that chased no sweep window in this tick are sent home
PSD_D void psd_sl_finish(const psd_rparams& P, int slot) {
    PSD_ONE {
        const unsigned long long v = __hip_atomic_load((unsigned long long*)psd_sl_cmd(P, slot), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (v != psd_sl_tag(P.tick, 0, 1)) psd_sl_publish(P, slot, nullptr);
    }
}
// slice 0 waits until the workers of the slot have stored their window blocks (the tail of a sweep is not sliced)
PSD_D void psd_sl_wait_stored(const psd_rparams& P, int slot) {
    const unsigned long long* fl = (const unsigned long long*)(psd_sl_cmd(P, slot) + 384);
    const unsigned long long want = psd_sl_tag(P.tick, 0, 6);
    for (int g = 1; g < P.slG; ++g) {
        int spins = 0;
        long long t0 = 0;
        while (__hip_atomic_load(fl + g, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != want) {
            if ((++spins & 255) == 0) {
                if (__hip_atomic_load(P.slerr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0) break;
                const long long now = (long long)__builtin_amdgcn_s_memrealtime();
                if (t0 == 0) t0 = now;
                else if (now - t0 > PSD_SL_WAIT_TICKS) {
                    __hip_atomic_store(P.slerr, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    break;
                }
            }
            __builtin_amdgcn_s_sleep(1);
        }
    }
    psd_acquire_fence();
}
// A worker workgroup (slice g >= 1 of the slot): every wavefront enters.  Waits for the tick's command, loads the window
// blocks of its own factors, runs its slice of the scan chase, stores its blocks, says so, leaves.
PSD_D void psd_sl_worker(const psd_rparams& P, int slot, int g) {
    PSD_LDS_DECL;
    psd_c2* cmd = (psd_c2*)(psd_lds + P.c2off);
    const int wv = PSD_WAVE_ROLE, nw = (int)blockDim.y;
    if (wv == 0) {
        unsigned char* cm = psd_sl_cmd(P, slot);
        const unsigned long long trun = psd_sl_tag(P.tick, 0, 1), tidle = psd_sl_tag(P.tick, 0, 4);
        int kind = 4, spins = 0;
        long long t0 = 0;
        for (;;) {
            const unsigned long long v = __hip_atomic_load((unsigned long long*)cm, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (v == trun) {
                kind = 1;
                break;
            }
            if (v == tidle) break;
            if ((++spins & 255) == 0) {
                if (__hip_atomic_load(P.slerr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0) break;
                const long long now = (long long)__builtin_amdgcn_s_memrealtime();
                if (t0 == 0) t0 = now;
                else if (now - t0 > PSD_SL_WAIT_TICKS) {
                    __hip_atomic_store(P.slerr, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    break;
                }
            }
            __builtin_amdgcn_s_sleep(2);
        }
        if (kind == 1) {
            psd_acquire_fence();
            psd_c2 C = *(const psd_c2*)(cm + 64);
            C.cmd = 5;
            C.slg = g;
            C.wboff = 0;
            PSD_ONE { *cmd = C; }
        } else {
            PSD_ONE { cmd->cmd = 0; }
        }
    }
    PSD_PAIR_BARRIER();
    const psd_c2 C = *cmd;
    if (C.cmd != 5) return;
    int jlo, jhi;
    psd_sl_range(C.p, C.slG, g, jlo, jhi);
    psd_rparams R;
    R.H = C.H + (size_t)(jlo - 1) * C.n * C.n;
    psd_win w;
    w.b = (double*)psd_lds;
    w.W = C.W; w.ld = C.ld; w.bsz = C.bsz; w.bs = C.bs; w.be = C.be;
    psd_win_load(R, w, C.n, jhi - jlo + 1, wv, nw);
    PSD_PAIR_BARRIER();
    psd_c3s_run(C, wv, nw, P.c3off);
    psd_win_store(R, w, C.n, jhi - jlo + 1, wv, nw);
    psd_release_fence();  // (this wavefront's stores are out)
    PSD_PAIR_BARRIER();
    if (wv == 0) {
        PSD_ONE {
            __hip_atomic_store((unsigned long long*)(psd_sl_cmd(P, slot) + 384) + g, psd_sl_tag(P.tick, 0, 6), __ATOMIC_RELAXED,
                               __HIP_MEMORY_SCOPE_AGENT);
        }
    }
}
// slice 0's side of a sliced run: the command to the slot's workers, then its own slice (with its helper wavefronts)
PSD_D void psd_c3s_lead(const psd_rparams& P, psd_c2& C, int slot) {
    PSD_LDS_DECL;
    psd_c2* cmd = (psd_c2*)(psd_lds + P.c2off);
    C.cmd = 5;
    C.slG = P.slG;
    C.slg = 0;
    C.sltick = P.tick;
    C.slbox = psd_sl_cmd(P, slot) + PSD_SL_CMD_BYTES;
    C.slerr = P.slerr;
    PSD_ONE {
        psd_sl_publish(P, slot, &C);
        *cmd = C;
    }
    PSD_PAIR_BARRIER();
    psd_c3s_run(C, 0, (int)blockDim.y, P.c3off);
    PSD_ONE { cmd->cmd = 0; }
}
#endif

// PSD.jl:806-886: one window (steps kcur .. kcur+nb-1) of the double-shift periodic QR sweep
PSD_D void psd_rq_qr_window(const psd_rparams& P, psd_rstate& st, double* ldsd, int* lcnt) {
    const int n = st.n, p = st.p, i = st.i, l = st.l, i1 = st.i1, i2 = st.i2;
    const int ks = st.kcur;
    // (a cursor one tick behind its predecessor chases 4 positions less in its first window: see psd_rstate::tgap)
    const int nb = (st.cursor > 0 && ks == l) ? st.cfirst : (st.W - 4);
    const int ke = (ks + nb - 1 < i - 1) ? (ks + nb - 1) : (i - 1);
    psd_win w;
    w.b = ldsd;
    w.W = st.W;
    w.ld = psd_win_pitch(st.W, st.Wmax);
    w.bsz = st.W * w.ld;
    w.bs = (ks > l) ? (ks - 1) : l;
    w.be = (ke + 3 < i) ? (ke + 3) : i;
    const long long tc0 = psd_clock();
    const bool scan3 = P.c3off != 0 && p >= PSD_C3_MINP && p <= PSD_C3_MAXP;
    const int klast = (ke < i - 2) ? ke : (i - 2);  // last position with a three-row bulge
    // factor-sliced window (psd_slice3.h): this workgroup is slice 0 and holds the blocks of its own factors 1..p0 only
    // while the sweep positions run; the slot's other workgroups hold theirs
    int p0 = p;
#ifndef PSD_HOSTSIM
    const bool sliced = scan3 && P.slG > 1 && st.mb && klast >= ks;
    if (sliced) {
        int jlo0;
        psd_sl_range(p, P.slG, 0, jlo0, p0);
    }
#else
    const bool sliced = false;
    (void)sliced;
#endif
    psd_c3_winio(P, w, n, p0, false);
    const long long tc1 = psd_clock();
    const int c1max = (w.be < i2) ? w.be : i2;
    const int r0 = (w.bs > i1) ? w.bs : i1;
    int n1 = 0, nj = 0;  // list lengths: owner 1, owners 2..p
    int kfirst = ks;
    if (scan3 || (P.c2off != 0 && P.c3off == 0 && p >= PSD_C2_MINP)) {
        // the positions with three-row bulges (all but possibly the last of a sweep) on the scan chase / two-wave chase
        if (klast >= ks) {
            psd_c2 C;
            C.cmd = 1;
            C.ld = w.ld; C.bsz = w.bsz; C.bs = w.bs; C.be = w.be;
            C.p = p; C.l = l; C.i = i; C.ks = ks; C.npos = klast - ks + 1;
            C.c1max = c1max; C.r0 = r0;
            C.n1 = 0; C.nj = 0;
            {
                PSD_LDS_DECL;
                C.wboff = (int)((char*)w.b - (char*)psd_lds);
            }
            C.tr = P.tr;
            C.dbg = (P.ticklog != nullptr && P.gl != nullptr) ? P.gl->c3dbg : nullptr;
            C.v0 = st.v[0]; C.v1 = st.v[1]; C.v2 = st.v[2];
            C.H = P.H; C.n = n; C.W = st.W;
            PSD_SYNC();
#ifndef PSD_HOSTSIM
            if (sliced) psd_c3s_lead(P, C, st.slot);
            else
#endif
            if (scan3) psd_c3_lead(P, C);
            else psd_c2_lead(P, C);
            PSD_SYNC();
            n1 = C.npos;
            nj = 2 * C.npos;
            kfirst = klast + 1;
        }
    }
#ifndef PSD_HOSTSIM
    if (sliced && kfirst <= ke) {
        // The last position of a sweep (a two-row bulge) is not sliced: the workers have stored their blocks (they say
        // so), this workgroup fetches them and finishes the window on all factors as the unsliced engine does.
        psd_sl_wait_stored(P, st.slot);
        psd_rparams R = P;
        R.H = P.H + (size_t)p0 * n * n;
        psd_win w2 = w;
        w2.b = w.b + (size_t)p0 * w.bsz;
        psd_c3_winio(R, w2, n, p - p0, false);
        p0 = p;
    }
#endif
    for (int k = kfirst; k <= ke; ++k) {
        const int nr = (3 < i - k + 1) ? 3 : (i - k + 1);
        const int rlim = (k + nr < i) ? (k + nr) : i;
        double x0, x1, x2 = 0.0;
        if (k > l) {
            x0 = w.at(1, k, k - 1);
            x1 = w.at(1, k + 1, k - 1);
            if (nr == 3) x2 = w.at(1, k + 2, k - 1);
        } else {
            x0 = st.v[0];
            x1 = st.v[1];
            x2 = (nr == 3) ? st.v[2] : 0.0;
        }
        if (nr == 3 && p > 1) {
            psd_qr_micro_h1(P, w, p, k, c1max, r0, rlim, x0, x1, x2, k > l, n1);
            // lane roles for the factors j = p..2 at this k: [0,nl) columns k+1.. of H_j, [nl,nl+nrw) rows
            // r0..rlim of H_{j-1}, last lane = annihilated column k of H_j
            const int nl = c1max - k, nrw = rlim - r0 + 1, cnt = nl + nrw + 1;
            PSD_LANEVAR(int, lane_off);
            PSD_LANEVAR(int, lane_str);
            PSD_LANEVAR(int, lane_adj);
            PSD_PAR_ONCE(t, cnt) {
                if (t < nl) {
                    PSD_LV(lane_off) = (k + 1 + t - w.bs) * w.ld + (k - w.bs);
                    PSD_LV(lane_str) = 1;
                    PSD_LV(lane_adj) = 0;
                } else if (t < nl + nrw) {
                    PSD_LV(lane_off) = (k - w.bs) * w.ld + (r0 + (t - nl) - w.bs);
                    PSD_LV(lane_str) = w.ld;
                    PSD_LV(lane_adj) = w.bsz;
                } else {
                    PSD_LV(lane_off) = (k - w.bs) * w.ld + (k - w.bs);
                    PSD_LV(lane_str) = 1;
                    PSD_LV(lane_adj) = 0;
                }
            }
            const int lk = nl + (k - r0);
            double tau3 = psd_refl3(x0, x1, x2);  // reflector of factor p; each micro-step generates the next one
            for (int j = p; j >= 2; --j)
                psd_qr_micro3(P, w, j, nl, cnt, lk, k, lane_off, lane_str, lane_adj, x0, x1, x2, tau3, j > 2, nj);
        } else {
            double tau = (nr == 3) ? psd_refl3(x0, x1, x2) : psd_refl2(x0, x1);
            if (p > 1) {
                psd_win_reflect(P, w, 1, p, nr, k, x1, x2, tau, k, i2, i1, rlim, 1, k - 1, x0, k > l, 1, n1);
            } else {  // same matrix: left, then right (PSD.jl:834-837)
                psd_win_reflect(P, w, 1, 1, nr, k, x1, x2, tau, k, i2, 1, 0, 1, k - 1, x0, k > l, 1, n1);
                psd_win_reflect(P, w, 1, 1, nr, k, x1, x2, tau, 1, 0, i1, rlim, 1, k - 1, x0, false, 1, n1);
            }
            for (int j = p; j >= 2; --j) {
                x0 = w.at(j, k, k);
                x1 = w.at(j, k + 1, k);
                x2 = (nr == 3) ? w.at(j, k + 2, k) : 0.0;
                tau = (nr == 3) ? psd_refl3(x0, x1, x2) : psd_refl2(x0, x1);
                psd_win_reflect(P, w, j, j - 1, nr, k, x1, x2, tau, k + 1, i2, i1, rlim, j, k, x0, true, j, nj);
                if (nr == 3) {
                    double y0 = w.at(j, k + 1, k + 1), y1 = w.at(j, k + 2, k + 1);
                    const double tau2 = psd_refl2(y0, y1);
                    psd_win_reflect(P, w, j, j - 1, 2, k + 1, y1, 0.0, tau2, k + 2, i2, i1, rlim, j, k + 1, y0, true,
                                    j, nj + 1);
                }
            }
        }
        n1 += 1;
        nj += (nr == 3) ? 2 : 1;
    }
    PSD_PAR_FOR(m, p) { lcnt[m] = (m == 0) ? n1 : nj; }
    const long long tc2 = psd_clock();
    psd_c3_winio(P, w, n, p0, true);  // (a sliced window: this slice's blocks; the workers store theirs)
    const long long tc3 = psd_clock();
    st.cyc[1] += tc1 - tc0;
    st.cyc[2] += tc2 - tc1;
    st.cyc[3] += tc3 - tc2;
    const int phi = (ke + 2 < i) ? (ke + 2) : i;
    // (the sweep goes on below this window: nothing reads the far part of this window's update before the sweep ends)
    psd_desc_write(P, st, lcnt, ks, phi, w.be + 1, i2, i1, w.bs - 1, (ke < i - 1) ? 1 : 0);
    st.nwindows += 1;
    st.kcur = ke + 1;
    if (ke >= i - 1) {  // sweep complete (PSD.jl:887)
        st.its += 1;
        st.phase = (st.cursor > 0) ? PSD_PH_CDONE : ((st.train_n > 1) ? PSD_PH_TWAIT : PSD_PH_DECIDE);
        // Long train: a bulge that leaves the bottom of the range converged — the last or the last but one subdiagonal
        // entry of the product negligible against its diagonal neighbours, from the window still in LDS — stops the
        // train: bulges that have not entered are cancelled (psd_rq_cursor_body), the leader decides sooner.  Only a hint:
        // the deflation itself is the leader's decision on the exact band (psd_rq_decide).
        if (st.mb && P.ccancel != nullptr && st.train_n > st.train_S + 1 && st.train_S > 0 && i - 2 >= w.bs && i - 2 >= l) {
            PSD_SYNC();
            PSD_ONE {
                double sub1 = w.at(1, i, i - 1), sub2 = w.at(1, i - 1, i - 2);
                double d0 = w.at(1, i, i), d1 = w.at(1, i - 1, i - 1), d2 = w.at(1, i - 2, i - 2);
                for (int j = 2; j <= p; ++j) {
                    const double t0 = w.at(j, i, i), t1 = w.at(j, i - 1, i - 1), t2 = w.at(j, i - 2, i - 2);
                    sub1 *= t1;
                    sub2 *= t2;
                    d0 *= t0;
                    d1 *= t1;
                    d2 *= t2;
                }
                const bool conv = fabs(sub1) <= st.ulp * (fabs(d0) + fabs(d1)) || fabs(sub2) <= st.ulp * (fabs(d1) + fabs(d2));
                if (conv) psd_atomic_store(P.ccancel + ((st.cursor > 0) ? st.parent : st.slot), 1);
            }
            PSD_SYNC();
        }
    }
}

// PSD.jl:602-663: one window of the RQ clean-up pass (k descending from kcur)
PSD_D void psd_rq_rq_window(const psd_rparams& P, psd_rstate& st, double* ldsd, int* lcnt) {
    const int n = st.n, p = st.p, i = st.i, l = st.l, i1 = st.i1, i2 = st.i2;
    const int nb = st.W - 3;
    const int ks = st.kcur;
    const int ke = (ks - nb + 1 > l) ? (ks - nb + 1) : l;
    psd_win w;
    w.b = ldsd;
    w.W = st.W;
    w.ld = psd_win_pitch(st.W, st.Wmax);
    w.bsz = st.W * w.ld;
    w.bs = ke - 1;
    w.be = (ks + 1 < i) ? (ks + 1) : i;
    PSD_PAR_FOR(m, p) { lcnt[m] = 0; }
    psd_c3_winio(P, w, n, p, false);
    for (int k = ks; k >= ke; --k) {
        for (int j = 1; j <= p - 1; ++j) {
            double x[2] = {w.at(j, k, k), w.at(j, k, k - 1)};
            PSD_SYNC();
            const double t = psd_reflector_small(x, 2);
            PSD_ONE {
                w.at(j, k, k - 1) = 0.0;
                w.at(j, k, k) = x[0];
            }
            PSD_SYNC();
            psd_tr tr;
            tr.pos = k - 1;
            tr.kind = PSD_TR_R2;
            tr.c0 = x[1];
            tr.c1 = 1.0;
            tr.c2 = t;
            psd_win_apply(w, j + 1, j, tr, k - 1, i2, i1, k - 1);
            psd_record(P, lcnt, j + 1, tr);
        }
        if (k < i) {
            double x[2] = {w.at(p, k + 1, k + 1), w.at(p, k + 1, k)};
            PSD_SYNC();
            const double t = psd_reflector_small(x, 2);
            PSD_ONE {
                w.at(p, k + 1, k) = 0.0;
                w.at(p, k + 1, k + 1) = x[0];
            }
            PSD_SYNC();
            psd_tr tr;
            tr.pos = k;
            tr.kind = PSD_TR_R2;
            tr.c0 = x[1];
            tr.c1 = 1.0;
            tr.c2 = t;
            if (p == 1) {
                // same matrix: the reference applies rmul! (rows i1:k) first, then lmul! (PSD.jl:627-629)
                psd_win_apply(w, 0, 1, tr, 0, -1, i1, k);
                psd_win_apply(w, 1, 0, tr, k, i2, 0, -1);
            } else {
                psd_win_apply(w, 1, p, tr, k, i2, i1, k);
            }
            psd_record(P, lcnt, 1, tr);
        }
    }
    psd_c3_winio(P, w, n, p, true);
    psd_desc_write(P, st, lcnt, w.bs, w.be, w.be + 1, i2, i1, w.bs - 1);
    st.nwindows += 1;
    st.kcur = ke - 1;
    if (ke <= l) {  // pass complete: PSD.jl:653-663
        const psd_mat<double> H1 = psd_fac(P, n, 1);
        const psd_mat<double> Hp = psd_fac(P, n, p);
        PSD_ONE {
            Hp(l, l - 1) = 0.0;
            H1(l, l - 1) = 0.0;
        }
        PSD_SYNC();
        st.phase = PSD_PH_SHIFT;
    }
}

// PSD.jl:896-1054: deflation of a 1x1 or 2x2 block at the bottom of the active window.
// Returns true if an apply descriptor was emitted.
PSD_D bool psd_rq_deflate(const psd_rparams& P, psd_rstate& st, double* ldsd, int* lcnt) {
    const int n = st.n, p = st.p, i = st.i, l = st.l, i1 = st.i1, i2 = st.i2;
    st.phase = PSD_PH_NEXT;
    if (l == i) {  // PSD.jl:896-899
        PSD_ONE {
            P.wr[i - 1] = P.hdiag[i];
            P.wi[i - 1] = 0.0;
        }
        st.ndefl1 += 1;
        return false;
    }
    st.ndefl2 += 1;
    psd_win w;
    w.b = ldsd;
    w.W = st.W;
    w.ld = psd_win_pitch(st.W, st.Wmax);
    w.bsz = st.W * w.ld;
    w.bs = i - 1;
    w.be = i;
    double hh11, hh12, hh21, hh22;
    if (st.wantT) {
        PSD_PAR_FOR(m, p) { lcnt[m] = 0; }
        psd_c3_winio(P, w, n, p, false);
        double hp22 = 1.0, hp12 = 0.0, hp11 = 1.0;  // PSD.jl:908-920
        for (int j = 2; j <= p; ++j) {
            hp22 *= w.at(j, i, i);
            hp12 = hp11 * w.at(j, i - 1, i) + hp12 * w.at(j, i, i);
            hp11 *= w.at(j, i - 1, i - 1);
        }
        hh21 = w.at(1, i, i - 1) * hp11;
        hh22 = w.at(1, i, i - 1) * hp12 + w.at(1, i, i) * hp22;
        hh11 = w.at(1, i - 1, i - 1) * hp11;
        hh12 = w.at(1, i - 1, i - 1) * hp12 + w.at(1, i - 1, i) * hp22;
    } else {  // PSD.jl:921-926
        hh11 = P.hdiag[i - 1];
        hh12 = P.hsup[i - 1];
        hh21 = P.hsub[i];
        hh22 = P.hdiag[i];
    }
    PSD_SYNC();
    double a = hh11, b = hh12, c = hh21, d = hh22, gcs, gsn, w1r, w1i, w2r, w2i;
    psd_gs2x2(a, b, c, d, gcs, gsn, w1r, w1i, w2r, w2i);  // PSD.jl:930
    const double hsub_im1 = w1i;                           // PSD.jl:934
    if (!st.wantT) {
        PSD_ONE {
            P.wr[i - 2] = w1r; P.wi[i - 2] = w1i;
            P.wr[i - 1] = w2r; P.wi[i - 1] = w2i;
        }
        return false;
    }
    // negligible diagonal entries of H_j, j > 1 (PSD.jl:937-958)
    int jmin = 0, jmax = 0;
    for (int j = 2; j <= p; ++j) {
        const double hn = P.hnorms[j];
        if (jmin == 0 && fabs(w.at(j, i - 1, i - 1)) <= hn) jmin = j;
        if (fabs(w.at(j, i, i)) <= hn) jmax = j;
    }
    if (jmin != 0 && jmax != 0) {
        if (jmin - 1 <= p - jmax + 1) jmax = 0;
        else jmin = 0;
    }
    if (jmin != 0) {  // PSD.jl:959-977 (with beta written back; the reference writes xi[2] — defect)
        for (int j = 1; j <= jmin - 1; ++j) {
            double x[2] = {w.at(j, i, i), w.at(j, i, i - 1)};
            PSD_SYNC();
            const double t = psd_reflector_small(x, 2);
            psd_tr tr;
            tr.pos = i - 1;
            tr.kind = PSD_TR_R2;
            tr.c0 = x[1];
            tr.c1 = 1.0;
            tr.c2 = t;
            PSD_ONE {
                w.at(j, i, i - 1) = 0.0;
                w.at(j, i, i) = x[0];
            }
            PSD_SYNC();
            psd_win_apply(w, j + 1, j, tr, i - 1, i2, i1, i - 1);
            psd_record(P, lcnt, j + 1, tr);
        }
    } else {  // PSD.jl:978-1052
        bool replaceG = (jmax > 0) && (hsub_im1 == 0);
        const double a1 = hypot(w1r, w1i), a2 = hypot(w2r, w2i);
        const double prr = w2r * w1r - w2i * w1i, pri = w2r * w1i + w2i * w1r;
        if (prr == 0 && pri == 0) {
            replaceG = true;
        } else if (hsub_im1 == 0) {
            if (fmin(a1, a2) / fmax(a1, a2) < PSD_DBL_EPS) replaceG = true;
        }
        bool guarded = false;  // the first pass left a subdiagonal entry that is not negligible in H_1 (see below)
        for (int its2 = 1; its2 <= 20; ++its2) {
            if (replaceG) {
                double rr;
                psd_givens(w.at(1, i - 1, i - 1), w.at(1, i, i - 1), gcs, gsn, rr);
            }
            PSD_SYNC();
            psd_tr tr;
            tr.pos = i - 1;
            tr.kind = PSD_TR_G;
            tr.c0 = gcs;
            tr.c1 = gsn;
            tr.c2 = 0.0;
            if (p == 1) {
                psd_win_apply(w, 1, 0, tr, i - 1, i2, 0, -1);
                psd_win_apply(w, 0, 1, tr, 0, -1, i1, i);
            } else {
                psd_win_apply(w, 1, p, tr, i - 1, i2, i1, i);
            }
            psd_record(P, lcnt, 1, tr);
            const int jstop = (2 > jmax + 1) ? 2 : (jmax + 1);
            for (int j = p; j >= jstop; --j) {
                double x[2] = {w.at(j, i - 1, i - 1), w.at(j, i, i - 1)};
                PSD_SYNC();
                const double t = psd_reflector_small(x, 2);
                PSD_ONE {
                    w.at(j, i - 1, i - 1) = x[0];
                    w.at(j, i, i - 1) = 0.0;
                }
                PSD_SYNC();
                tr.pos = i - 1;
                tr.kind = PSD_TR_H2;
                tr.c0 = x[1];
                tr.c1 = 0.0;
                tr.c2 = t;
                psd_win_apply(w, j, j - 1, tr, i, i2, i1, i);
                psd_record(P, lcnt, j, tr);
            }
            const double h21 = w.at(1, i, i - 1);
            // what zeroing H_1(i, i-1) would cost, measured in the factor that carries it (the reference measures against
            // the eigenvalues of the PRODUCT, which says nothing about H_1 when the pair is orders of magnitude apart)
            const double h1sc = fabs(w.at(1, i - 1, i - 1)) + fabs(w.at(1, i - 1, i)) + fabs(w.at(1, i, i));
            PSD_SYNC();
            if (replaceG) {
                if (fabs(h21) < fmax(st.smlnum, st.ulp * fmax(a1, a2))) break;
                if (guarded && fabs(h21) <= fmax(st.smlnum, 8.0 * st.ulp * h1sc)) break;
            } else {
                // Guard (not in the reference, which zeroes whatever the one pass leaves, PSD.jl:1031-1037,1066-1073): a
                // real pair whose rotation came from the explicitly formed product can leave a subdiagonal entry far
                // above rounding level when the two eigenvalues are many orders of magnitude apart (but less than
                // 1 / eps, so that :996-1003 does not switch to the rotation from H_1 itself).  Zeroing it would put that
                // entry into the residual; instead the block takes the reference's own fallback, rotations from
                // H_1's column (an unshifted periodic QR step on the 2 x 2 block, which converges with the ratio of the
                // pair per pass), until the entry is negligible in H_1.
                if (hsub_im1 != 0 || jmax > 0) break;
                if (fabs(h21) <= fmax(st.smlnum, 8.0 * st.ulp * h1sc)) break;
                guarded = true;
            }
            replaceG = true;
        }
        PSD_ONE {
            if (jmax > 0) {
                w.at(1, i, i - 1) = 0.0;
                if (jmax > 1) w.at(jmax, i, i - 1) = 0.0;
            } else if (hh21 == 0) {
                w.at(1, i, i - 1) = 0.0;
            }
        }
        PSD_SYNC();
        if (replaceG) {  // PSD.jl:1039-1051 (indices i-1, i; the reference's [1],[2] is a defect)
            double l1 = w.at(1, i - 1, i - 1);
            for (int j = 2; j <= p; ++j) l1 *= w.at(j, i - 1, i - 1);
            const double d1 = hypot(l1 - w1r, w1i), d2 = hypot(l1 - w2r, w2i);
            if (d1 > d2) {
                double tr_ = w1r, ti_ = w1i;
                w1r = w2r; w1i = w2i;
                w2r = tr_; w2i = ti_;
            }
        }
    }
    PSD_ONE {
        P.wr[i - 2] = w1r; P.wi[i - 2] = w1i;
        P.wr[i - 1] = w2r; P.wi[i - 1] = w2i;
    }
    psd_c3_winio(P, w, n, p, true);
    psd_desc_write(P, st, lcnt, i - 1, i, i + 1, i2, i1, i - 2);
    return true;
}

// One launch = state transitions until a window's worth of transforms has been emitted.
PSD_D void psd_rq_step_body(const psd_rparams& P) {
    PSD_LDS_DECL;
    psd_rstate st = *P.st;
    if (st.phase == PSD_PH_DONE) {
        PSD_ONE { P.desc->active = 0; }
        return;
    }
    const int NT = PSD_NTHREADS;
    double* ldsd = (double*)psd_lds;
    const size_t winb = (size_t)st.p * psd_win_area(st.Wmax);
    double* red = ldsd + winb;
    int* redi = (int*)(red + NT);
    int* lcnt = redi + 2 * NT;
    PSD_ONE { P.desc->active = 0; }
    const long long tk0 = psd_clock(), tw0 = psd_wallclock();
    bool emitted = false;
    int guard = 0;
    int phmask = 0;
    while (!emitted && st.phase != PSD_PH_DONE && guard < 4 * st.n + 16) {
        ++guard;
        phmask |= 1 << st.phase;
        switch (st.phase) {
            case PSD_PH_DECIDE: {
                const long long td0 = psd_clock();
                {
                    PSD_DBG_T0();
                    if (psd_rq_decide(P, st, red, redi, ldsd, winb)) emitted = true;  // (put off: nothing emitted, the descriptor stays inactive)
                    PSD_DBG_ADD(0);
                }
                st.cyc[0] += psd_clock() - td0;

                break;
            }
            case PSD_PH_RQ: {
                PSD_DBG_T0();
                psd_rq_rq_window(P, st, ldsd, lcnt);
                PSD_DBG_ADD(7);
                emitted = true;
                break;
            }
            case PSD_PH_SHIFT:
                if (psd_rq_shift(P, st, ldsd, redi)) emitted = true;  // (no transformations: the descriptor stays inactive)
                break;
            case PSD_PH_QR:
                psd_rq_qr_window(P, st, ldsd, lcnt);
                emitted = true;
                break;
            case PSD_PH_DEFLATE: {
                PSD_DBG_T0();
                emitted = psd_rq_deflate(P, st, ldsd, lcnt);
                PSD_DBG_ADD(6);
                break;
            }
            case PSD_PH_TWAIT: {  // the leader's sweep is done: wait for the cursors of the train, then go on
                bool all = true;
                if (st.mb) {  // a cursor adds one to its leader's counter after its last window (read as the previous launch left it)
                    all = ((P.plan != nullptr) ? P.plan[PSD_PLAN_CDONE + st.slot] : psd_atomic_load(P.cdone + st.slot)) == st.train_n - 1;
                } else {  // a cursor counts itself in after its last store; its slot's state is then complete
                    all = psd_atomic_load(P.cep + PSD_TRAIN_MAX) == st.train_n - 1;
                    if (all) psd_acquire_fence();
                }
                if (all) {
                    int swept = st.train_n;  // bulges that ran (a stopped long train drops the ones that had not entered)
                    if (st.mb && P.ccancel != nullptr) {
                        const int dropped = psd_atomic_load(P.ccancel + PSD_SLOTS + st.slot);
                        swept -= dropped;
                        st.ntrainsweeps -= dropped;
                    }
                    for (int b = 1; b < swept; ++b) {
                        if (!st.mb) st.nwindows += P.cst[b].nwindows;  // (multi-block: a cursor adds its own to psd_rglobal)
                        st.nsweeps += 1;
                        st.its += 1;
                    }
                    st.train_n = 1;
                    st.W = st.Wmax;
                    st.phase = PSD_PH_DECIDE;
                    emitted = true;  // (the decision runs in the next launch: the cursors' last bulk updates are
                                     //  ordered before it by the tick barrier of the driver)
                } else {
                    emitted = true;
                }
                break;
            }
            case PSD_PH_NEXT:  // PSD.jl:1057-1060
                if (st.mb) {  // the budget is shared by all ranges
                    PSD_SYNC();
                    PSD_ONE { redi[0] = psd_atomic_add(&P.gl->pbudget[st.prob], -st.its); }
                    PSD_SYNC();
                    st.maxitleft = redi[0];  // (value before the subtraction: the line below takes `its` off)
                    PSD_SYNC();
                }
                st.maxitleft -= st.its;
                st.niter += st.its;
                if (st.its > st.maxits) st.maxits = st.its;
                st.i = st.l - 1;
                st.l = st.mb ? st.lo : 1;
                st.its = 1;
                st.exc_dec = 0;
                st.phase = (st.i >= st.l) ? PSD_PH_DECIDE : PSD_PH_FINAL;
                // (a wide rest of the range: its product band is the job of psd_rq_band in front of the next launch)
                if (st.mb && P.bandinfo != nullptr && st.phase == PSD_PH_DECIDE && st.i - st.l + 1 >= PSD_DECIDE_YIELD) emitted = true;
                break;
            case PSD_PH_FINAL: {  // PSD.jl:1066-1073
                if (st.mb) {
                    psd_mb_finish_leader(P, st, redi);
                    break;
                }
                const psd_mat<double> H1 = psd_fac(P, st.n, 1);
                PSD_SYNC();
                PSD_PAR_FOR(q, st.n - 1) {
                    if (P.wi[q] == 0.0) H1(q + 2, q + 1) = 0.0;
                }
                PSD_SYNC();
                st.phase = PSD_PH_DONE;
                break;
            }
            default:
                st.phase = PSD_PH_DONE;
                break;
        }
    }
    st.cyc[4] += psd_clock() - tk0;
    st.cyc[5] += psd_wallclock() - tw0;
    if (P.ticklog && P.tick < P.ticklog_n) {
        PSD_ONE { psd_atomic_max(P.ticklog + P.tick, (int)(((psd_wallclock() - tw0) << 12) | (phmask & 0xfff))); }
    }
    if (st.info == PSD_LIST_OVERFLOW) st.phase = PSD_PH_DONE;  // (a window that overran a list ends the call)
    PSD_SYNC();
    PSD_ONE {
        *P.st = st;
        if (st.mb && st.phase == PSD_PH_DONE && st.info == 0) psd_mb_release_slot(P, st.slot);
    }
}

// One launch of cursor b >= 1 of a multishift train (P.st = the cursor's state, P.lead = the main state, P.desc / P.cnt /
// P.tr = the cursor's own descriptor and lists).  Starts two windows behind its predecessor, chases one window per
// launch, ends in PSD_PH_CDONE.
// The chase kernels run blocks of 64 x 1 threads, or 64 x 2 when P.c2off != 0: the second wavefront is the helper of
// the two-wave chase (psd_c2_helper) and takes no part in anything else.
#ifndef PSD_HOSTSIM
#define PSD_C2_ENTER(P)                                 \
    if ((P).c2off != 0 && PSD_WAVE_ROLE >= 1) {         \
        psd_c2_helper((P).c2off, (P).c3off);            \
        return;                                         \
    }
#define PSD_C2_LEAVE(P) \
    if ((P).c2off != 0) psd_c2_release((P).c2off)
#else
#define PSD_C2_ENTER(P) ((void)0)
#define PSD_C2_LEAVE(P) ((void)0)
#endif
PSD_KERNEL_B(PSD_C3_WAVES * PSD_STEP_NT) psd_rq_step(psd_rparams P) {
    PSD_C2_ENTER(P);
    psd_rq_step_body(P);
    PSD_C2_LEAVE(P);
}

PSD_D void psd_rq_cursor_body(const psd_rparams& P, int b) {
    PSD_LDS_DECL;
    PSD_ONE { P.desc->active = 0; }
    psd_rstate st;
    if (P.gl) {
        st = *P.st;  // (multi-block: a slot's state is only ever read by its own workgroup, see psd_rq_step_mb)
    } else if (!psd_pub_read(P.cep + b, P.tick, P.st, st)) {
        return;  // (published in an earlier launch, not being rewritten)
    }
    if (st.cursor != b) return;
    if (st.phase != PSD_PH_CWAIT && st.phase != PSD_PH_QR) return;
    double* ldsd = (double*)psd_lds;
    const size_t winb = (size_t)st.p * psd_win_area(st.Wmax);
    int* lcnt = (int*)(ldsd + winb + PSD_STEP_NT) + 2 * PSD_STEP_NT;
    // (the stop word as the previous launch left it: a bulge that stops the train in THIS launch is seen from the next one
    //  on by every workgroup alike, whatever the order the workgroups of a launch happen to run in)
    const bool stopped = st.mb && P.ccancel != nullptr &&
                         ((P.plan != nullptr) ? P.plan[PSD_PLAN_CCANCEL + st.parent] : psd_atomic_load(P.ccancel + st.parent)) != 0;
    if (st.phase == PSD_PH_CWAIT && st.train_S > 0 && stopped) {
        // the train was stopped (psd_rq_qr_window): this bulge and the later ones of this slot do not enter
        PSD_ONE {
            const int k = 1 + (st.train_n - 1 - b) / st.train_S;
            psd_atomic_add(P.ccancel + PSD_SLOTS + st.parent, k);
            psd_mb_release_slot(P, st.slot);
            psd_atomic_add(P.cdone + st.parent, k);
        }
        return;
    }
    if (st.phase == PSD_PH_CWAIT) {
        // Cursor b enters at the tick its schedule says (psd_cursor_schedule: nb + 4 positions, or two whole windows, behind
        // its predecessor for the whole sweep).  A fixed schedule, not a look at the predecessor's progress: the
        // predecessor's state is being written while this kernel runs.
        if (P.tick < st.cstart) return;
        double h11, h12, h21, h22, h32;
        psd_rq_topband(P, st.n, st.p, st.l, st.i, h11, h12, h21, h22, h32);
        psd_rq_startvec(h11, h12, h21, h22, h32, P.tshift + 4 * ((st.train_ms > 0) ? (b % st.train_ms) : b), st.v);
        st.kcur = st.l;
        st.phase = PSD_PH_QR;
    }
    psd_rq_qr_window(P, st, ldsd, lcnt);
    // long train: this slot goes on as bulge b + S (its report below counts bulge b in)
    bool rearm = st.mb && st.phase == PSD_PH_CDONE && st.train_S > 0 && b + st.train_S <= st.train_n - 1;
    int ncancel = 0;  // later bulges of this slot that a stopped train drops
    if (rearm && stopped) {
        ncancel = (st.train_n - 1 - b) / st.train_S;
        rearm = false;
    }
    const int nwin_done = st.nwindows;
    long long cyc_done[4] = {0, st.cyc[1], st.cyc[2], st.cyc[3]};
    if (rearm) {
        st.cursor = b + st.train_S;
        psd_cursor_schedule(st, st.cursor, st.cstart, st.cfirst);
        if (st.cstart <= P.tick) {  // (excluded by the margin psd_rq_shift keeps; never run a bulge off its schedule)
            st.info = PSD_LIST_OVERFLOW;
            PSD_ONE {
                psd_atomic_store(&P.gl->info, PSD_LIST_OVERFLOW);
                psd_atomic_store(&P.gl->abort, 1);
                psd_atomic_store(&P.gl->done, 1);
            }
        }
        st.phase = PSD_PH_CWAIT;
        st.kcur = 0;
        st.nwindows = 0;
        st.cyc[1] = st.cyc[2] = st.cyc[3] = 0;
    }
    PSD_SYNC();
    PSD_ONE { *P.st = st; }
    if (!st.mb && st.phase == PSD_PH_CDONE) {  // last window: count this cursor in (its state and lists are out first)
        PSD_ONE {
            psd_release_fence();
            psd_atomic_add(P.cep + PSD_TRAIN_MAX, 1);
        }
    }
    if (st.mb && (st.phase == PSD_PH_CDONE || rearm)) {  // last window of this bulge: report, hand the slot back (or keep it)
        PSD_ONE {
            psd_atomic_add(&P.gl->nwindows, nwin_done);
            for (int q = 1; q < 4; ++q) psd_atomic_add_ll(&P.gl->cyc[q], cyc_done[q]);
            if (!rearm) psd_mb_release_slot(P, st.slot);
            if (ncancel) psd_atomic_add(P.ccancel + PSD_SLOTS + st.parent, ncancel);
            psd_atomic_add(P.cdone + st.parent, 1 + ncancel);
        }
    }
}

// All cursors of a tick in ONE launch: workgroup 0 is the ordinary state machine (the leader), workgroup b >= 1 is cursor
// b.  P holds the leader's state and slot 0 of the cursor arrays (see psd_rq_apply_train).  No cross-stream events: the
// chases of a tick run side by side on different compute units, the tick ends with the kernel.
PSD_KERNEL_B(PSD_C3_WAVES * PSD_STEP_NT) psd_rq_step_train(psd_rparams P, int p, int cstride) {
    PSD_C2_ENTER(P);
    const int b = PSD_BLOCK_X;
    if (b == 0) {
        psd_rq_step_body(P);
    } else {
        psd_rparams Q = P;
        Q.lead = P.st;
        Q.st = P.cst + b;
        Q.desc = P.desc + b;
        Q.cnt = P.cnt + (size_t)b * cstride;
        Q.tr = P.tr + (size_t)b * p * PSD_TR_CAP;
        psd_rq_cursor_body(Q, b);
    }
    PSD_C2_LEAVE(P);
}

// Multi-block tick: workgroup s runs slot s (see psd_mb_claim).  P holds slot 0 of every per-slot array.
PSD_D void psd_rq_step_mb_body(const psd_rparams& P, int p, int cstride) {
    const int s = PSD_BLOCK_X;
    psd_rparams Q = P;
    Q.st = P.cst + s;
    Q.desc = P.desc + s;
    Q.cnt = P.cnt + (size_t)s * cstride;
    Q.tr = P.tr + (size_t)s * p * PSD_TR_CAP;
    PSD_ONE { Q.desc->active = 0; }
    if (psd_atomic_load(&P.gl->done) || psd_atomic_load(&P.gl->abort)) return;
    int role = psd_atomic_load(P.role + s);
    if (role == PSD_ROLE_CLAIMED && psd_atomic_load(P.epoch + s) < P.tick) {
        // written in an earlier launch: the slot goes live (only this workgroup ever changes a claimed slot's role)
        role = (P.cst[s].cursor > 0) ? PSD_ROLE_CURSOR : PSD_ROLE_LEADER;
        PSD_ONE { psd_atomic_store(P.role + s, role); }
    }
    if (P.nprob > 1 && (role == PSD_ROLE_LEADER || role == PSD_ROLE_CURSOR)) {
        const int pr = P.cst[s].prob, n_ = P.cst[s].n;
        const size_t sm = (size_t)p * n_ * n_, sb = (size_t)n_ + 8;
        Q.H = P.H + pr * sm;
        if (P.Z) Q.Z = P.Z + pr * sm;
        Q.hdiag = P.hdiag + pr * sb; Q.hsub = P.hsub + pr * sb; Q.hsup = P.hsup + pr * sb;
        Q.Pd = P.Pd + pr * sb; Q.Pe = P.Pe + pr * sb; Q.Pf = P.Pf + pr * sb;
        Q.wr = P.wr + pr * sb; Q.wi = P.wi + pr * sb;
        Q.hnorms = P.hnorms + (size_t)pr * (p + 8);
    }
    if (role == PSD_ROLE_LEADER) {
        Q.lead = Q.st;
        Q.tshift = P.tshift + (size_t)s * PSD_TSHIFT_STRIDE;
        psd_rq_step_body(Q);
    } else if (role == PSD_ROLE_CURSOR) {
        const int parent = P.cst[s].parent;
        Q.lead = P.cst + parent;
        Q.tshift = P.tshift + (size_t)parent * PSD_TSHIFT_STRIDE;
        psd_rq_cursor_body(Q, P.cst[s].cursor);
    }
}
PSD_KERNEL_B(PSD_C3_WAVES * PSD_STEP_NT) psd_rq_step_mb(psd_rparams P, int p, int cstride) {
#ifndef PSD_HOSTSIM
    if (P.slG > 1 && PSD_BLOCK_Y >= 1) {  // grid (slots, slices): a worker of slot blockIdx.x (psd_slice3.h)
        psd_sl_worker(P, PSD_BLOCK_X, PSD_BLOCK_Y);
        return;
    }
#endif
    PSD_C2_ENTER(P);
    psd_rq_step_mb_body(P, p, cstride);
#ifndef PSD_HOSTSIM
    if (P.slG > 1) psd_sl_finish(P, PSD_BLOCK_X);
#endif
    PSD_C2_LEAVE(P);
}

// ------------------------------------------------------------------------------------------------
// Bulk application of one window's transform lists.  grid = (tiles, p owners, 3 roles).
//   role 0: left on H_m,  rows [plo,phi] x columns [lc0,lc1]
//   role 1: right on H_{m-1}, rows [rr0,rr1] x columns [plo,phi]
//   role 2: right on Z_m, rows [zr0,zr1] x columns [plo,phi]
// Every element of the panel is read once and written once; the sequence runs out of LDS.
// Sweep and RQ-pass lists visit their positions monotonically (ascending resp. descending), so one line (a column of
// the rows panel, a row of the columns panel; at most 32 elements) stays in registers through the whole list: the loop
// over positions is fully unrolled (static register indices) and the list is read ahead from LDS, with two sentinel
// records behind its end.
#define PSD_TR_LDS_RECS (PSD_TR_CAP + 2)
#define PSD_TR_LDS_BYTES (sizeof(psd_tr) * PSD_TR_LDS_RECS + 16)
template <bool UP>
PSD_D void psd_tr_regline(const psd_tr* ltr, int plo, double (&a)[34]) {
    int e = 0;
    psd_tr cur = ltr[0], nxt = ltr[1];
#pragma unroll
    for (int q = 0; q < 32; ++q) {
        const int b = UP ? q : 31 - q;
        while (cur.pos - plo == b) {
            psd_tr_apply(cur, a[b], a[b + 1], a[b + 2]);
            cur = nxt;
            ++e;
            nxt = ltr[e + 1 < PSD_TR_LDS_RECS ? e + 1 : PSD_TR_LDS_RECS - 1];
        }
    }
}
// Stages the owner's list (plus sentinels) in LDS and classifies its order: +1 ascending, -1 descending, 0 neither.
// Contains block synchronisations; every thread of the block calls it.
PSD_D int psd_tr_stage(const psd_tr* gtr, int cnt, psd_tr* ltr, int* flags) {
    PSD_PAR_FOR(e, PSD_TR_LDS_RECS) {
        psd_tr tr;
        if (e < cnt) {
            tr = gtr[e];
        } else {
            tr.pos = 0x3fffffff;
            tr.kind = PSD_TR_G;
            tr.c0 = 1.0;
            tr.c1 = tr.c2 = 0.0;
        }
        ltr[e] = tr;
    }
    PSD_ONE { flags[0] = flags[1] = 0; }
    PSD_SYNC();
    PSD_PAR_FOR(e, cnt - 1) {
        if (ltr[e + 1].pos < ltr[e].pos) flags[0] = 1;
        if (ltr[e + 1].pos > ltr[e].pos) flags[1] = 1;
    }
    PSD_SYNC();
    return !flags[0] ? 1 : (!flags[1] ? -1 : 0);
}

PSD_D void psd_rq_apply_body(const psd_rparams& P, int n, int p, int role) {
    PSD_LDS_DECL;
    const psd_apply_desc d = *P.desc;
    if (!d.active) return;
    const int m = PSD_BLOCK_Y + 1;
    const int cnt = P.cnt[m - 1] < PSD_TR_CAP ? P.cnt[m - 1] : PSD_TR_CAP;
    if (cnt <= 0) return;
    const int T = PSD_APPLY_NT;
    const int S = d.phi - d.plo + 1;
    psd_tr* ltr = (psd_tr*)psd_lds;
    int* flags = (int*)(psd_lds + sizeof(psd_tr) * PSD_TR_LDS_RECS);
    double* tile = (double*)(psd_lds + PSD_TR_LDS_BYTES);
    const psd_tr* gtr = P.tr + (size_t)(m - 1) * PSD_TR_CAP;
    if (role == 0) {
        const int c0 = d.lc0 + PSD_BLOCK_X * T;
        if (c0 > d.lc1) return;
        const int nc = (d.lc1 - c0 + 1 < T) ? (d.lc1 - c0 + 1) : T;
        const psd_mat<double> M = psd_mat<double>{P.H + (size_t)(m - 1) * n * n, n};
        const int ldt = T + 1;
        // rows panel -> LDS (transposed access: a thread owns a column); eight loads in flight per thread
        PSD_PAR_FOR(t0, 32 * 4) {  // (S <= 32; thread t0 covers row t0 & 31 of the columns (t0 >> 5) + 4 k)
            const int r = t0 & 31, cb = t0 >> 5;
            if (r >= S) continue;
            for (int k0 = 0; k0 < T / 4; k0 += 8) {
                double v[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    const int c = cb + 4 * (k0 + u);
                    v[u] = (c < nc) ? M(d.plo + r, c0 + c) : 0.0;
                }
#pragma unroll
                for (int u = 0; u < 8; ++u) tile[r * ldt + cb + 4 * (k0 + u)] = v[u];
            }
        }
        const int order = psd_tr_stage(gtr, cnt, ltr, flags);
        PSD_PAR_FOR(c, nc) {
            if (order != 0) {
                double a[34];
#pragma unroll
                for (int r = 0; r < 34; ++r) a[r] = (r < S) ? tile[r * ldt + c] : 0.0;
                if (order > 0) psd_tr_regline<true>(ltr, d.plo, a);
                else psd_tr_regline<false>(ltr, d.plo, a);
#pragma unroll
                for (int r = 0; r < 32; ++r)
                    if (r < S) tile[r * ldt + c] = a[r];
                continue;
            }
            for (int e = 0; e < cnt; ++e) {
                const psd_tr tr = ltr[e];
                const int r = tr.pos - d.plo;
                const int len = psd_tr_len(tr);
                double a1 = tile[r * ldt + c], a2 = tile[(r + 1) * ldt + c], a3 = (len == 3) ? tile[(r + 2) * ldt + c] : 0.0;
                psd_tr_apply(tr, a1, a2, a3);
                tile[r * ldt + c] = a1;
                tile[(r + 1) * ldt + c] = a2;
                if (len == 3) tile[(r + 2) * ldt + c] = a3;
            }
        }
        PSD_SYNC();
        PSD_PAR_FOR(t0, 32 * 4) {
            const int r = t0 & 31, cb = t0 >> 5;
            if (r >= S) continue;
            for (int k0 = 0; k0 < T / 4; k0 += 8) {
                double v[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) v[u] = tile[r * ldt + cb + 4 * (k0 + u)];
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    const int c = cb + 4 * (k0 + u);
                    if (c < nc) M(d.plo + r, c0 + c) = v[u];
                }
            }
        }
    } else {
        const int lo = (role == 1) ? d.rr0 : d.zr0;
        const int hi = (role == 1) ? d.rr1 : d.zr1;
        const int r0 = lo + PSD_BLOCK_X * T;
        if (r0 > hi) return;
        const int nr = (hi - r0 + 1 < T) ? (hi - r0 + 1) : T;
        const int jm = (role == 1) ? ((m == 1) ? p : (m - 1)) : m;
        double* base = (role == 1) ? P.H : P.Z;
        const psd_mat<double> M = psd_mat<double>{base + (size_t)(jm - 1) * n * n, n};
        const int order = psd_tr_stage(gtr, cnt, ltr, flags);
        if (order != 0) {
            // a thread owns a row of the columns panel: coalesced loads straight into registers, no LDS tile
            PSD_PAR_FOR(r, nr) {
                double a[34];
#pragma unroll
                for (int c = 0; c < 34; ++c) a[c] = (c < S) ? M(r0 + r, d.plo + c) : 0.0;
                if (order > 0) psd_tr_regline<true>(ltr, d.plo, a);
                else psd_tr_regline<false>(ltr, d.plo, a);
#pragma unroll
                for (int c = 0; c < 32; ++c)
                    if (c < S) M(r0 + r, d.plo + c) = a[c];
            }
            return;
        }
        PSD_PAR_FOR(t, S * T) {
            const int r = t & (T - 1), c = t / T;
            if (r >= nr) continue;
            tile[c * T + r] = M(r0 + r, d.plo + c);
        }
        PSD_SYNC();
        PSD_PAR_FOR(r, nr) {
            for (int e = 0; e < cnt; ++e) {
                const psd_tr tr = ltr[e];
                const int c = tr.pos - d.plo;
                const int len = psd_tr_len(tr);
                double a1 = tile[c * T + r], a2 = tile[(c + 1) * T + r], a3 = (len == 3) ? tile[(c + 2) * T + r] : 0.0;
                psd_tr_apply(tr, a1, a2, a3);
                tile[c * T + r] = a1;
                tile[(c + 1) * T + r] = a2;
                if (len == 3) tile[(c + 2) * T + r] = a3;
            }
        }
        PSD_SYNC();
        PSD_PAR_FOR(t, S * T) {
            const int r = t & (T - 1), c = t / T;
            if (r >= nr) continue;
            M(r0 + r, d.plo + c) = tile[c * T + r];
        }
    }
}

PSD_KERNEL_B(PSD_APPLY_NT) psd_rq_apply(psd_rparams P, int n, int p) { psd_rq_apply_body(P, n, p, PSD_BLOCK_Z); }

// Bulk updates of all M cursors of a tick in two launches.  P holds cursor 0's descriptor / counts / lists; cursor b's
// are b entries (desc), b * cstride (counts) and b * p * PSD_TR_CAP (lists) further.  Updates of different cursors meet
// only where one cursor's rows cross another's columns, i.e. a rows-role block against a columns-role block of H:
// pass 0 runs the rows role and the Z role of every cursor (grid.z = 2 M), pass 1 the columns role (grid.z = M).
PSD_KERNEL_B(PSD_APPLY_NT) psd_rq_apply_train(psd_rparams P, int n, int p, int cstride, int pass) {
    const int z = PSD_BLOCK_Z;
    const int b = (pass == 0) ? (z >> 1) : z;
    const int role = (pass == 0) ? ((z & 1) ? 2 : 0) : 1;
    psd_rparams Q = P;
    Q.desc = P.desc + b;
    Q.cnt = P.cnt + (size_t)b * cstride;
    Q.tr = P.tr + (size_t)b * p * PSD_TR_CAP;
    psd_rq_apply_body(Q, n, p, role);
}

// ------------------------------------------------------------------------------------------------
// Bulk application, work-list form: ONE grid of single-wave workgroups loops over the (cursor, owner, role, tile)
// items of a tick, so that no workgroup is launched only to find its cursor idle (the grid-per-cursor form above starts
// tiles x p x 2M workgroups per launch, almost all of which read a descriptor and leave: at n = 1024, p = 64 that is
// 32 768 workgroups and most of the launch's 70-100 us).  An item = 64 lines (rows role: columns of H_m; column roles:
// rows of H_{m-1} / Z_m) x the window span S <= 32 (17 KiB of LDS + the list: eight workgroups per compute unit, so
// that one item's loads overlap another's arithmetic): the tile goes HBM -> LDS, every lane streams
// its line through a three-element register window (the lists of a sweep visit their positions monotonically, so an
// element is read from LDS once and written once), and the tile goes back.  Lists that are not monotone (2x2 deflation
// passes) are applied record by record in LDS.  pass 0: rows and Z roles of every cursor; pass 1: column roles (one
// cursor's rows cross another's columns, as in psd_rq_apply_train).
#define PSD_WL_NT 64
#define PSD_WL_LINES 64
#define PSD_WL_LD 33
// ld: pitch of a line in the tile (odd, >= the longest list span: W + 2 for windows of width W <= 30, PSD_WL_LD otherwise):
// at p = 64 (W = 17) the tile takes 9.5 instead of 16.5 KiB and twice as many workgroups fit a CU
PSD_HD int psd_wl_pitch(int W) { return (W + 2 < PSD_WL_LD) ? ((W + 2) | 1) : PSD_WL_LD; }
PSD_HD size_t psd_wl_lds_bytes(int ld = PSD_WL_LD) {
    return PSD_TR_LDS_BYTES + (size_t)PSD_WL_LINES * ld * sizeof(double) + (size_t)(3 * PSD_SLOTS + 4) * sizeof(int);
}

// the three-element window of one line moving up (UP) or down through positions; L: the line in LDS (S elements)
template <bool UP>
PSD_D void psd_wl_stream(double* L, int S, const psd_tr* ltr, int cnt, int plo) {
    if (UP) {
        int base = 0, nx = 3;
        double a1 = L[0], a2 = (1 < S) ? L[1] : 0.0, a3 = (2 < S) ? L[2] : 0.0;
        for (int e = 0; e < cnt; ++e) {
            const psd_tr tr = ltr[e];
            const int rp = tr.pos - plo;
            while (base < rp) {
                L[base] = a1;
                a1 = a2;
                a2 = a3;
                a3 = (nx < S) ? L[nx] : 0.0;
                ++nx;
                ++base;
            }
            psd_tr_apply(tr, a1, a2, a3);
        }
        L[base] = a1;
        if (base + 1 < S) L[base + 1] = a2;
        if (base + 2 < S) L[base + 2] = a3;
    } else {
        int base = ltr[0].pos - plo;  // window = elements base .. base + 2, starting at the first (highest) record
        double a1 = L[base], a2 = (base + 1 < S) ? L[base + 1] : 0.0, a3 = (base + 2 < S) ? L[base + 2] : 0.0;
        for (int e = 0; e < cnt; ++e) {
            const psd_tr tr = ltr[e];
            const int rp = tr.pos - plo;
            while (base > rp) {
                if (base + 2 < S) L[base + 2] = a3;
                a3 = a2;
                a2 = a1;
                --base;
                a1 = L[base];
            }
            psd_tr_apply(tr, a1, a2, a3);
        }
        L[base] = a1;
        if (base + 1 < S) L[base + 1] = a2;
        if (base + 2 < S) L[base + 2] = a3;
    }
}

// lane t's share of a 64-line tile, HBM <-> 32 registers <-> LDS.  Rows role (lines = columns l0.., elements = rows
// plo..phi, contiguous in memory): lane = (row pair 2 (t & 15), column t >> 4 of every group of four), sixteen 16-byte
// accesses.  Column roles (lines = rows l0.., elements = columns plo..): lane = row l0 + t, one 8-byte access per
// element, 512 contiguous bytes per wavefront and column.
// (spans of at most 16 rows — p = 64: 15 — take 8 row-pair lanes x 8 columns per access instead of 16 x 4: with 16 row-pair
//  lanes half of the wavefront idled in the rows role's transfers)
PSD_D void psd_wl_load(bool rowsrole, const psd_mat<double>& Mx, int plo, int S, int l0, int nl, int t, double (&v)[32]) {
    if (rowsrole) {
        const bool narrow = S <= 16;
        const int rr = narrow ? 2 * (t & 7) : 2 * (t & 15), cq = narrow ? (t >> 3) : (t >> 4), cs = narrow ? 8 : 4;
        const bool pair = rr + 1 < S;
#pragma unroll
        for (int u = 0; u < 16; ++u) {
            const int c = cs * u + cq;
            v[2 * u] = v[2 * u + 1] = 0.0;
            if (rr < S && c < nl && c < PSD_WL_LINES) {
                const double* src = &Mx(plo + rr, l0 + c);
                if (pair) {
                    const psd_pair x = psd_pair_load(src);
                    v[2 * u] = x.a;
                    v[2 * u + 1] = x.b;
                } else {
                    v[2 * u] = src[0];
                }
            }
        }
    } else {
#pragma unroll
        for (int u = 0; u < 32; ++u) v[u] = (t < nl && u < S) ? Mx(l0 + t, plo + u) : 0.0;
    }
}
PSD_D void psd_wl_to_tile(bool rowsrole, double* tile, int LD, int S, int nl, int t, const double (&v)[32]) {
    if (rowsrole) {
        const bool narrow = S <= 16;
        const int rr = narrow ? 2 * (t & 7) : 2 * (t & 15), cq = narrow ? (t >> 3) : (t >> 4), cs = narrow ? 8 : 4;
        if (rr < S) {
#pragma unroll
            for (int u = 0; u < 16; ++u) {
                const int c = cs * u + cq;
                if (c < PSD_WL_LINES) {
                    tile[c * LD + rr] = v[2 * u];
                    if (rr + 1 < S) tile[c * LD + rr + 1] = v[2 * u + 1];
                }
            }
        }
    } else if (t < nl) {
#pragma unroll
        for (int u = 0; u < 32; ++u)
            if (u < S) tile[t * LD + u] = v[u];
    }
}
PSD_D void psd_wl_store(bool rowsrole, const psd_mat<double>& Mx, const double* tile, int LD, int plo, int S, int l0, int nl, int t) {
    if (rowsrole) {
        const bool narrow = S <= 16;
        const int rr = narrow ? 2 * (t & 7) : 2 * (t & 15), cq = narrow ? (t >> 3) : (t >> 4), cs = narrow ? 8 : 4;
        if (rr < S) {
            const bool pair = rr + 1 < S;
#pragma unroll
            for (int u = 0; u < 16; ++u) {
                const int c = cs * u + cq;
                if (c < nl && c < PSD_WL_LINES) {
                    double* dst = &Mx(plo + rr, l0 + c);
                    if (pair) {
                        psd_pair x;
                        x.a = tile[c * LD + rr];
                        x.b = tile[c * LD + rr + 1];
                        psd_pair_store(dst, x);
                    } else {
                        dst[0] = tile[c * LD + rr];
                    }
                }
            }
        }
    } else if (t < nl) {
#pragma unroll
        for (int u = 0; u < 32; ++u)
            if (u < S) Mx(l0 + t, plo + u) = tile[t * LD + u];
    }
}
PSD_D void psd_wl_compute(double* tile, int LD, int S, int nl, int t, int order, const psd_tr* ltr, int cnt, int plo) {
    if (t >= nl) return;
    double* L = tile + t * LD;
    if (order > 0) {
        psd_wl_stream<true>(L, S, ltr, cnt, plo);
    } else if (order < 0) {
        psd_wl_stream<false>(L, S, ltr, cnt, plo);
    } else {
        for (int e = 0; e < cnt; ++e) {
            const psd_tr tr = ltr[e];
            const int r = tr.pos - plo;
            const int len = psd_tr_len(tr);
            double a1 = L[r], a2 = L[r + 1], a3 = (len == 3) ? L[r + 2] : 0.0;
            psd_tr_apply(tr, a1, a2, a3);
            L[r] = a1;
            L[r + 1] = a2;
            if (len == 3) L[r + 2] = a3;
        }
    }
}

// zlo..zhi: the owners m (1-based) whose Z_m this context holds (period-sharded contexts: the Z role of the others is
// some other rank's work; 1..p otherwise)
// line ranges of a window's three roles under an apply mode (see psd_rq_apply_wl)
PSD_HD void psd_wl_ranges(psd_apply_desc& d, int mode, int cut) {
    (void)cut;
    if (mode == 3) {  // everything but the Schur vectors
        d.zr1 = d.zr0 - 1;
    } else if (mode == 4) {  // the Schur vectors only (pass 0)
        d.lc1 = d.lc0 - 1;
        d.rr1 = d.rr0 - 1;
    } else if (mode == 5) {  // the two H roles, the column role on its near rows only (rcut .. rr1)
        d.zr1 = d.zr0 - 1;
        if (d.rr0 < d.rcut) d.rr0 = d.rcut;
    } else if (mode == 6) {  // the far rows of the column role only (pass 1): rr0 .. rcut - 1
        d.lc1 = d.lc0 - 1;
        d.zr1 = d.zr0 - 1;
        if (d.rr1 > d.rcut - 1) d.rr1 = d.rcut - 1;
    } else if (mode == 7) {  // pass 0: the rows role on its near columns only (lc0 .. cut - 1)
        d.zr1 = d.zr0 - 1;
        if (d.lc1 > d.cut - 1) d.lc1 = d.cut - 1;
    } else if (mode == 8) {  // pass 0: the far columns of the rows role only (cut .. lc1)
        d.zr1 = d.zr0 - 1;
        if (d.lc0 < d.cut) d.lc0 = d.cut;
    }
}

#define PSD_WL_GROUP 4  // 64-line tiles per item: the owner's list is staged once for all of them
// mode 0: everything.
// Modes 5 / 6 split the column roles by rows (psd_cdefer_edge): mode 5 = the H roles without the far rows of the column
// roles, mode 6 = those far rows (pass 1), which run on the second stream beside the next tick's chases.
// Modes 7 / 8 split the rows roles by columns (psd_rdefer_edge) in the same way: mode 7 = near columns, mode 8 = far columns.
// Modes 3 / 4 split off the Schur vectors alone: mode 3 = the two H roles (both passes), mode 4 = the Z role (pass 0).
// Nothing reads Z_m before the iteration ends and only owner m's lists touch it, so the Z updates of a tick only have to
// stay in tick order among themselves: they run on a second stream beside the following ticks' chases.
PSD_KERNEL_B(PSD_WL_NT) psd_rq_apply_wl(psd_rparams P, int n, int p, int cstride, int pass, int M, int zlo, int zhi,
                                        int mode, int ld) {
    PSD_LDS_DECL;
    psd_tr* ltr = (psd_tr*)psd_lds;
    int* flags = (int*)(psd_lds + sizeof(psd_tr) * PSD_TR_LDS_RECS);
    double* tile = (double*)(psd_lds + PSD_TR_LDS_BYTES);
    int* ioff = (int*)(tile + (size_t)PSD_WL_LINES * ld);  // [M + 1] item offsets, [M] groups A, [M] groups B
    int* tA = ioff + PSD_SLOTS + 2;
    int* tB = tA + PSD_SLOTS;
    const int TL = PSD_WL_LINES;
    // item table: slot b contributes p * (groups of role A + groups of role B) items (pass 0: A = rows, B = Z;
    // pass 1: A = columns, B = none); a group = up to PSD_WL_GROUP tiles of 64 lines.  The group size adapts to the
    // tick: whole groups only while they still leave every workgroup of the grid a few items (a tick of one small
    // window must spread over the chip, a tick of sixty windows must not stage a list per 8 KiB)
    int GL = TL;
#ifndef PSD_HOSTSIM
    {
        // lane b = slot b (M <= 64): one read of the descriptor, group counts for the chosen group size, offsets by a
        // wavefront scan (a serial prefix over 64 slots in LDS and a second pass over the descriptors cost every
        // workgroup of every launch several microseconds before its first item)
        const int b = PSD_TID;
        psd_apply_desc d;
        d.active = 0;
        if (b < M) d = P.desc[b];
        int la = 0, lz = 0;  // lines of role A / B
        if (b < M && d.active) {
            psd_wl_ranges(d, mode, d.cut);
            if (pass == 0) {
                la = (d.lc1 >= d.lc0) ? (d.lc1 - d.lc0 + 1) : 0;
                lz = (d.zr1 >= d.zr0) ? (d.zr1 - d.zr0 + 1) : 0;
            } else {
                la = (d.rr1 >= d.rr0) ? (d.rr1 - d.rr0 + 1) : 0;
            }
        }
        int a = (la + TL - 1) / TL, z = (lz + TL - 1) / TL;
        int tiles = p * (a + z);
#pragma unroll
        for (int sft = 32; sft > 0; sft >>= 1) tiles += __shfl_xor(tiles, sft, 64);
        int grp = 1;
        if (tiles >= 4 * PSD_WL_GROUP * PSD_GRID_X) grp = PSD_WL_GROUP;
        else if (tiles >= 8 * PSD_GRID_X) grp = 2;
        GL = TL * grp;
        a = (la + GL - 1) / GL;
        z = (lz + GL - 1) / GL;
        const int mine = p * (a + z);
        int incl = mine;
#pragma unroll
        for (int sft = 1; sft < 64; sft <<= 1) {
            const int up = __shfl_up(incl, sft, 64);
            if (b >= sft) incl += up;
        }
        if (b < M) {
            tA[b] = a;
            tB[b] = z;
            ioff[b] = incl - mine;
        }
        if (b == 63) ioff[M] = incl;  // (lanes >= M contribute nothing)
        PSD_SYNC();
    }
#else
    for (int trial = 0; trial < 2; ++trial) {
        PSD_PAR_FOR(b, M) {
            psd_apply_desc d = P.desc[b];
            int a = 0, z = 0;
            if (d.active) {
                psd_wl_ranges(d, mode, d.cut);
                if (pass == 0) {
                    a = (d.lc1 >= d.lc0) ? ((d.lc1 - d.lc0 + 1 + GL - 1) / GL) : 0;
                    z = (d.zr1 >= d.zr0) ? ((d.zr1 - d.zr0 + 1 + GL - 1) / GL) : 0;
                } else {
                    a = (d.rr1 >= d.rr0) ? ((d.rr1 - d.rr0 + 1 + GL - 1) / GL) : 0;
                }
            }
            tA[b] = a;
            tB[b] = z;
        }
        PSD_SYNC();
        PSD_ONE {
            int acc = 0;
            for (int b = 0; b < M; ++b) {
                ioff[b] = acc;
                acc += p * (tA[b] + tB[b]);
            }
            ioff[M] = acc;
        }
        PSD_SYNC();
        if (trial == 1) break;
        const int tiles = ioff[M];
        int grp = 1;
        if (tiles >= 4 * PSD_WL_GROUP * PSD_GRID_X) grp = PSD_WL_GROUP;
        else if (tiles >= 8 * PSD_GRID_X) grp = 2;
        PSD_SYNC();
        if (grp == 1) break;
        GL = TL * grp;
    }
#endif
    const int total = ioff[M];
    for (int item = PSD_BLOCK_X; item < total; item += PSD_GRID_X) {
        int b = 0;
        {  // the slot whose items contain `item`: last b with ioff[b] <= item (binary search; empty slots repeat an offset)
            int lo_ = 0, hi_ = M - 1;
            while (lo_ < hi_) {
                const int mid = (lo_ + hi_ + 1) >> 1;
                if (ioff[mid] <= item) lo_ = mid;
                else hi_ = mid - 1;
            }
            b = lo_;
        }
        const int per = tA[b] + tB[b];
        const int q = item - ioff[b];
        const int m = q / per + 1, tt = q - (m - 1) * per;
        const int role = (pass == 0) ? ((tt < tA[b]) ? 0 : 2) : 1;
        const int gix = (role == 2) ? (tt - tA[b]) : tt;
        if (role == 2 && (m < zlo || m > zhi)) continue;
        psd_apply_desc d = P.desc[b];
        psd_wl_ranges(d, mode, d.cut);
        int cnt = P.cnt[(size_t)b * cstride + (m - 1)];
        if (cnt > PSD_TR_CAP) cnt = PSD_TR_CAP;
        if (cnt <= 0) continue;
        const psd_tr* gtr = P.tr + ((size_t)b * p + (m - 1)) * PSD_TR_CAP;
        const int S = d.phi - d.plo + 1;
        int lo, hi;
        double* base;
        int jm;
        if (role == 0) {
            lo = d.lc0; hi = d.lc1; base = P.H; jm = m;
        } else if (role == 1) {
            lo = d.rr0; hi = d.rr1; base = P.H; jm = (m == 1) ? p : (m - 1);
        } else {
            lo = d.zr0; hi = d.zr1; base = P.Z; jm = m;
        }
        const int g0 = lo + gix * GL;  // first line of the group (1-based column for the rows role, row otherwise)
        const int gl = (hi - g0 + 1 < GL) ? (hi - g0 + 1) : GL;
        const int ns = (gl + TL - 1) / TL;
        const psd_mat<double> Mx = psd_mat<double>{base + ((size_t)d.prob * p + (jm - 1)) * n * n, n};
        const bool rowsrole = role == 0;
#ifndef PSD_HOSTSIM
        // the next tile's loads are in flight while this one is computed on: a lane keeps them in 32 registers
        const int t = PSD_TID;
        double v[32];
        psd_wl_load(rowsrole, Mx, d.plo, S, g0, (gl < TL) ? gl : TL, t, v);
        PSD_SYNC();  // (the previous item's tile and list are no longer in use)
        const int order = psd_tr_stage(gtr, cnt, ltr, flags);
        for (int k = 0; k < ns; ++k) {
            const int l0 = g0 + k * TL;
            const int nl = (gl - k * TL < TL) ? (gl - k * TL) : TL;
            psd_wl_to_tile(rowsrole, tile, ld, S, nl, t, v);
            PSD_SYNC();
            if (k + 1 < ns) {
                const int nl1 = (gl - (k + 1) * TL < TL) ? (gl - (k + 1) * TL) : TL;
                psd_wl_load(rowsrole, Mx, d.plo, S, l0 + TL, nl1, t, v);
            }
            psd_wl_compute(tile, ld, S, nl, t, order, ltr, cnt, d.plo);
            PSD_SYNC();
            psd_wl_store(rowsrole, Mx, tile, ld, d.plo, S, l0, nl, t);
            PSD_SYNC();
        }
#else
        PSD_SYNC();
        const int order = psd_tr_stage(gtr, cnt, ltr, flags);
        for (int k = 0; k < ns; ++k) {
            const int l0 = g0 + k * TL;
            const int nl = (gl - k * TL < TL) ? (gl - k * TL) : TL;
            PSD_PAR_FOR(t, PSD_WL_NT) {
                double v[32];
                psd_wl_load(rowsrole, Mx, d.plo, S, l0, nl, t, v);
                psd_wl_to_tile(rowsrole, tile, ld, S, nl, t, v);
            }
            PSD_SYNC();
            PSD_PAR_FOR(t, PSD_WL_NT) { psd_wl_compute(tile, ld, S, nl, t, order, ltr, cnt, d.plo); }
            PSD_SYNC();
            PSD_PAR_FOR(t, PSD_WL_NT) { psd_wl_store(rowsrole, Mx, tile, ld, d.plo, S, l0, nl, t); }
            PSD_SYNC();
        }
#endif
    }
}

// hnorms[j] = ulp*n*opnorm(H_j, 1), column-1 / sub-Hessenberg clean-up (PSD.jl:379-388,406), and
// state initialisation.  grid = p blocks.
PSD_KERNEL psd_rq_init(psd_rparams P, int n, int p, int wantT, int wantZ, int W, int maxitfac, int maxlog,
                       int train_want, int train_oc, int mb, int cgap, int train_long, int train_wdiv) {
    PSD_LDS_DECL;
    double* red = (double*)psd_lds;
    const int j = PSD_BLOCK_X + 1;
    const int pr = PSD_BLOCK_Y;  // problem of the batch (grid.y = nprob; 1 otherwise)
    const int NT = PSD_NTHREADS;
    const psd_mat<double> Hj = psd_mat<double>{P.H + ((size_t)pr * p + (j - 1)) * n * n, n};
    if (j == 1) {
        // _gethess!: zero below the first subdiagonal
        PSD_PAR_FOR(c, n) {
            for (int r = c + 3; r <= n; ++r) Hj(r, c + 1) = 0.0;
        }
        PSD_ONE {
            psd_rstate st;
            st.n = n; st.p = p; st.wantT = wantT; st.wantZ = wantZ; st.W = st.Wmax = W; st.train_oc = train_oc;
            st.phase = PSD_PH_DECIDE; st.info = 0;
            st.i = n; st.l = 1; st.its = 1; st.maxitleft = maxitfac * n;
            st.i1 = 1; st.i2 = n; st.kcur = 0; st.maxits = 0; st.niter = 0;
            st.nsweeps = st.nrqpass = st.ndefl1 = st.ndefl2 = st.nwindows = st.nlog = 0;
            st.maxlog = maxlog;
            st.v[0] = st.v[1] = st.v[2] = 0.0;
            for (int q = 0; q < 6; ++q) st.cyc[q] = 0;
            st.train_want = train_want; st.train_n = 1; st.train_id = 0; st.cursor = 0; st.train_tick0 = 0;
            st.train_S = 0; st.train_ms = 0; st.train_long = mb ? train_long : 0; st.train_wdiv = (train_wdiv > 0) ? train_wdiv : 8;
            st.ntrains = st.ntrainsweeps = 0; st.exc_dec = 0;
            st.ulp = PSD_DBL_EPS;
            st.smlnum = PSD_DBL_MIN * ((double)n / PSD_DBL_EPS);
            // PSD.jl:366-375 with _AT_pwr16[] = 4: ulpx = ulp^(1 + 4/16)
            double s = PSD_DBL_EPS, ulpx = PSD_DBL_EPS;
            s = sqrt(s);          // iu = 8
            s = sqrt(s);          // iu = 4 -> selected
            ulpx *= s;
            st.ulpx = ulpx;
            st.mb = mb;
            st.cgap = cgap;
            st.tgap = 2;
            st.cstart = st.cfirst = 0;
            st.slot = pr;
            st.parent = -1;
            st.lo = 1;
            st.train_key = 0;
            st.plan_tick = -1;
            st.plan_used = 0;
            st.opn_tick = -2;
            st.prob = pr;
            for (int q = 0; q < PSD_TRAIN_MAX; ++q) st.cslots[q] = 0;
            if (mb) {
                // (the host zeroed psd_rglobal and the slot words: roles FREE) problem pr starts as one range led by slot
                // pr; block 0 sets what is shared
                P.cst[pr] = st;
                P.gl->pactive[pr] = 1;
                P.gl->pbudget[pr] = maxitfac * n;
                P.role[pr] = PSD_ROLE_LEADER;
                if (pr == 0) {
                    P.gl->nactive = P.nprob;
                    P.gl->nslotmax = P.nprob;
                    for (int q = 0; q < PSD_SLOTS; ++q) {
                        P.epoch[q] = PSD_EPOCH_NEVER;
                        P.cdone[q] = 0;
                        P.desc[q].active = 0;
                    }
                }
            } else {
                *P.st = st;
                P.desc->active = 0;
            }
        }
    } else {
        PSD_PAR_FOR(r, n - 1) { Hj(r + 2, 1) = 0.0; }
        PSD_SYNC();
        PSD_PAR_FOR(t, NT) {
            double best = 0.0;
            for (int c = 1 + t; c <= n; c += NT) {
                double s = 0.0;
                for (int r = 1; r <= n; ++r) s += fabs(Hj(r, c));
                if (s > best) best = s;
            }
            red[t] = best;
        }
        PSD_SYNC();
        PSD_ONE {
            double best = 0.0;
            for (int t = 0; t < NT; ++t)
                if (red[t] > best) best = red[t];
            P.hnorms[(size_t)pr * (p + 8) + j] = PSD_DBL_EPS * n * best;
        }
    }
}
