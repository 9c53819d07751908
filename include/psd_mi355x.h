/* libpsd_mi355x — C ABI of the MI355X-native periodic Schur engine.
 *
 * The reference (RalphAS/PeriodicSchurDecompositions.jl v0.1.6) has no FFI layer: its operator API
 * is a set of Julia methods.  Each entry point below is what a thin Julia `ccall` wrapper binds
 * to replace one of those methods (file:line under /root/reference/src); INTEGRATION.md shows the
 * wrapper.  Plain pointers and sizes only; no exceptions cross the boundary.
 *
 * Conventions
 *   - matrices are column-major, ld = n, Float64;
 *   - host entry points take an array of p pointers (Julia's Vector{Matrix{Float64}}); the caller
 *     owns every buffer, results are written back into the same buffers (the reference's
 *     in-place contract, PeriodicSchurDecompositions.jl:120-152);
 *   - `_dev` entry points take one device allocation [p][n][n] (factor-major) already resident
 *     in HBM and leave the result there;
 *   - return value == *info: 0 ok; <0 argument -k invalid; >0 algorithmic:
 *       PSD_INFO_NOCONV + level  convergence failed at level i  (PSD.jl:892 ErrorException)
 *       PSD_INFO_NOTIMPL         not implemented                 (PSD.jl:30 NotImplemented)
 *       PSD_INFO_RUNTIME         HIP runtime failure / no device
 */
#ifndef PSD_MI355X_H
#define PSD_MI355X_H
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define PSD_INFO_NOCONV 1000000
#define PSD_INFO_NOTIMPL 2000000
#define PSD_INFO_RUNTIME 3000000

typedef struct psd_ctx psd_ctx;

/* per-call statistics: what bench.py needs to turn time into sweeps/s and algorithmic GB/s */
typedef struct psd_stats {
    int64_t niter;        /* sum of `its` over deflation levels (PSD.jl:1060 `niter`)            */
    int32_t maxits;       /* PSD.jl:1059                                                         */
    int32_t nsweeps;      /* double-shift QR sweeps performed (PSD.jl:806-887)                   */
    int32_t nrqpass;      /* RQ clean-up passes (PSD.jl:602-635)                                 */
    int32_t ndefl1;       /* 1x1 deflations                                                      */
    int32_t ndefl2;       /* 2x2 deflations                                                      */
    int32_t nwindows;     /* diagonal windows chased (= step-kernel launches that did work)      */
    int32_t nlaunch_step; /* step-kernel launches issued                                         */
    int32_t window;       /* window width W used by the chase kernel                             */
    int32_t nlog;         /* entries in the sweep log                                            */
    double ms_hess;       /* periodic Hessenberg-triangular reduction                            */
    double ms_formq;      /* Q formation                                                         */
    double ms_iter;       /* periodic QR iteration                                               */
    double ms_total;      /* device time of the whole call (excludes host<->device copies)       */
    double ms_copy;       /* host<->device copies (host entry points only)                       */
    double bytes_sweeps;  /* algorithmic bytes of all sweeps: sum 2*8*p*w*(2n+1) (wantZ) etc.    */
    double bytes_hess;    /* algorithmic bytes of the reduction: 2*8*p*(5/6)*n^3                 */
    double bytes_formq;   /* 2*8*p*n^3/3                                                         */
    double step_kernel_ms_avg; /* sampled HIP-event duration of the chase kernel (profile mode)  */
    int32_t step_kernel_samples;
    int32_t reserved;
    /* in-kernel cycle accounting of the chase kernel (s_memtime ticks): decide, window load, chase, window
     * store, total; [5] = total in 100 MHz s_memrealtime ticks (shader clock = cyc[4]/cyc[5]*100 MHz) */
    int64_t step_cycles[6];
} psd_stats;

/* context: device selection, stream, workspace cache. One call at a time per context. */
int psd_create(psd_ctx** ctx, int device);
int psd_destroy(psd_ctx* ctx);
/* profile != 0: sample the chase kernel's duration with HIP events (every 16th launch) */
int psd_set_profile(psd_ctx* ctx, int profile);
/* Multishift trains in the real and the complex periodic QR / QZ iteration.  bulges >= 2 (default and maximum 32 — the engines' own defaults: 32 real, 48 complex and real signed, 16 complex signed; PSD_TRAIN in the
 * environment presets it): a sweep of a large active block becomes a train of up to `bulges` double-shift sweeps whose
 * shifts are the eigenvalues of the trailing 2m x 2m block of the product (m <= 8; a longer train runs through the pairs
 * twice), chased as cursors two windows apart by one workgroup each, in windows whose width a cost model picks per train.  0 or 1: the reference's one-shift-one-sweep iteration (PSD.jl:729-763), sweep for sweep.  Same
 * decomposition up to rounding and iteration path; psd_stats.reserved counts the sweeps that ran inside trains. */
int psd_set_train(psd_ctx* ctx, int bulges);
int psd_get_train(psd_ctx* ctx);
/* psd_set_train sets every path (32 or more = every engine's own default); these two address the complex single-shift
 * path alone (its bulges take the eigenvalues of the trailing m x m block of the product as shifts, m <= 16, reused by
 * longer trains; default 48 bulges W positions apart, at most 64) */
int psd_set_train_z(psd_ctx* ctx, int bulges);
int psd_get_train_z(psd_ctx* ctx);
/* the signed paths psd_d_pschur(A, S) / psd_z_pschur(A, S) (double-shift sweeps of rgeneralized.jl:806-1054, single-shift
 * sweeps of generalized.jl:808-852; the complex one takes the eigenvalues of the trailing m x m block of H_1 T): trains with explicit shifts
 * taken from the trailing 2m x 2m block of prod_{l>=2} H_l^{s_l} * H_1 (default 32, the complex one stops at 16; psd_set_train sets this path too,
 * PSD_TRAIN_G presets it alone).  psd_stats.maxits counts the sweeps that ran inside trains.  -2 (test hook): single
 * sweeps started from explicit shifts instead of the implicit _qzrots start. */
int psd_set_train_g(psd_ctx* ctx, int bulges);
int psd_get_train_g(psd_ctx* ctx);
const char* psd_version(void);
/* 1 if the multi-stream periodic Hessenberg reductions of this context (phessenberg!, PSD.jl:213-259, for p >= 32,
 * n >= 512 or when forced) take the pipe form — consecutive chain launches overlapping on two streams —, 0 if they run
 * back to back.  Fixed for the life of the context at psd_create (a context created while no other context of the
 * process is alive gets the pipe form; PSD_H2_PIPE=0 / 2 in the environment forbids / forces it) and forced on by
 * psd_set_shard with world > 1, so that every rank of a period-sharded call runs the same form: the two forms round
 * differently (plain sum of squares against the scaled form of the norm). */
int psd_get_hess_pipe(psd_ctx* ctx);
/* Factor-sliced sweep windows of the real periodic QR iteration (PSD.jl:806-886): `slices` = G >= 2 workgroups chase one
 * window, workgroup g holding the diagonal window blocks of the factors of ITS contiguous slice of the period,
 * (g p/G, (g+1) p/G], and the chain vectors of a position handed from slice to slice as tagged records in the receiver's
 * inbox — the north_star partition (H_j by period, hand-off at slice boundaries) with the G compute
 * units of a slot as the ranks; an inbox is a plain device pointer, across GPUs it would be a peer mapping.  The
 * result does not depend on the number of slices (the reflectors are functions of the vectors handed over: G = 2 and
 * G = 4 agree to the bit) and agrees with slices = 1 to rounding.  1 (default): off.
 * Honoured for p <= 64 and at least two factors per slice, and clamped to what can be resident together (64 x G
 * workgroups, one per compute unit: G <= 4 on the 256 CUs of one MI355X); every wait is bounded, a hand-over that never arrives ends
 * the call with PSD_INFO_RUNTIME.  DESIGN.md section 7c. */
int psd_set_slices(psd_ctx* ctx, int slices);
int psd_get_slices(psd_ctx* ctx);
/* Period sharding over `world` contexts (one per GPU, one process each): every context runs the
 * latency-bound chains (Hessenberg links, window chases) and the updates of the factors H_j they read — identical on
 * all ranks, bit for bit (the tick schedule is reproducible: tests/test_gpu_headline.py, tests/test_gpu_shard.py) — while the
 * Schur vectors, half of all bulk bytes, are split by period: the context forms and updates only the Z_j of its
 * contiguous slice of the period.  No per-tick exchange; the caller all-gathers the slices at the end if it wants every
 * Z_j everywhere (periodicschurdecompositions.jl_amd/sharded.py does, over torch.distributed: RCCL on the GPUs, gloo in
 * the CPU test).  psd_shard_owned: owned[s] = 1 for the user slots s (0-based) of Z this context holds after a call with
 * `orient`.  Honoured by psd_d_pschur / _dev / _hess (Q formation and every Z update), by psd_z_pschur / _dev / _hess
 * (the same: BASELINE configs[2]) and by the iterations of psd_d_gpschur / the signed psd_z_pschur (their Hessenberg
 * stage accumulates Q on every rank); rank 0 of world 1 (default) = everything. */
int psd_set_shard(psd_ctx* ctx, int rank, int world);
int psd_shard_owned(psd_ctx* ctx, int p, char orient, uint8_t* owned);

/* phessenberg!(A)  — PeriodicSchurDecompositions.jl:213-259.
 * A[j] overwritten LAPACK-style (H_j in the upper part, reflectors below), tau is [p][n]
 * (tau[0][n-1] unused, tau[j>=1][n-1] = 0) exactly as the Hessenberg / QR objects the reference
 * returns are laid out. */
int psd_d_phessenberg(psd_ctx* ctx, int n, int p, double* const* A, double* tau, psd_stats* stats, int* info);

/* pschur!(A, lr; wantZ, wantT, maxitfac), Float64 — PeriodicSchurDecompositions.jl:120-152.
 * S must be NULL or all-true here (PSD_INFO_NOTIMPL otherwise): the signed case returns eigenvalues in the scaled
 * (alpha, beta, scale) form and has its own entry, psd_d_gpschur.
 * On exit A[s] holds the user-order factor T_s; the quasi-triangular one is A[*schurindex - 1]
 * (schurindex = 1 for 'R', p for 'L').  Z[s] (p pointers, may be NULL when !wantZ) receive Z_s.
 * wr/wi: n eigenvalues of the product.  sweeplog: optional [3*maxlog] (kind,l,i) per iteration. */
int psd_d_pschur(psd_ctx* ctx, int n, int p, double* const* A, const uint8_t* S, char orient, int wantT, int wantZ,
                 int maxitfac, double* const* Z, double* wr, double* wi, int* schurindex, psd_stats* stats,
                 int32_t* sweeplog, int64_t maxlog, int* info);

/* pschur!(H1, Hs; wantT, wantZ, Q, maxitfac, rev) — PeriodicSchurDecompositions.jl:322-330.
 * H[0] upper Hessenberg, H[1..p-1] upper triangular (internal order).  Q: p pointers holding Q_j
 * on entry (post-multiplied in place and returned as Z_j), or NULL for Z_j = accumulated
 * transformations only (then Zout receives them if wantZ).  With rev != 0 the caller permutes the
 * result as the reference does at :1078-1092 (the kernel work is identical). */
int psd_d_pschur_hess(psd_ctx* ctx, int n, int p, double* const* H, double* const* Q, int wantT, int wantZ,
                      int maxitfac, double* wr, double* wi, psd_stats* stats, int32_t* sweeplog, int64_t maxlog,
                      int* info);

/* nb Hessenberg-triangular problems of the same shape (n, p) in ONE call — what the Krylov driver of the reference issues
 * one by one (krylov.jl:575-592,645,710,800-829: projected problems of order <= 40): pschur!(H1, Hs; wantT, wantZ, Q,
 * maxitfac) for each.  H / Q: nb * p pointers, problem q's factor j at [q * p + j] (H[q * p] upper Hessenberg, the
 * others upper triangular); Q may be NULL when !wantZ, otherwise it holds Q_j on entry and Z_j on exit.  wr / wi: nb * n
 * eigenvalues, problem by problem.  infos[nb] (may be NULL): per-problem info; the return value is the first non-zero
 * one.  nb <= 32.  The problems run side by side on the slot scheduler of the real iteration. */
int psd_d_pschur_hess_batch(psd_ctx* ctx, int nb, int n, int p, double* const* H, double* const* Q, int wantT, int wantZ,
                            int maxitfac, double* wr, double* wi, int* infos, psd_stats* stats, int* info);

/* Device-resident variant of psd_d_pschur: dA, dZ are device pointers to [p][n][n] blocks in user
 * order (dZ may be NULL when !wantZ).  wr/wi/sweeplog are host buffers. */
int psd_d_pschur_dev(psd_ctx* ctx, int n, int p, double* dA, char orient, int wantT, int wantZ, int maxitfac,
                     double* dZ, double* wr, double* wi, int* schurindex, psd_stats* stats, int32_t* sweeplog,
                     int64_t maxlog, int* info);

/* ---- ComplexF64 (interleaved re,im; same conventions) ---------------------------------------------
 * Eigenvalues are returned in the reference's scaled form (generalized.jl:40-42,74-76):
 * values[k] = alpha[k] / beta[k] * 2^ascale[k], alpha interleaved complex. */

/* phessenberg!(A) for ComplexF64 — PeriodicSchurDecompositions.jl:213-259; tau is [p][n] complex */
int psd_z_phessenberg(psd_ctx* ctx, int n, int p, double* const* A, double* tau, psd_stats* stats, int* info);

/* pschur!(A::Vector{Matrix{ComplexF64}}, lr; wantZ, wantT) — PeriodicSchurDecompositions.jl:1106-1111, which runs
 * generalized.jl:108-137 with S = trues; with a signed S the signed Hessenberg reduction and the signed periodic QZ of
 * generalized.jl:138-146 run (the leftmost entry of S in working order must be true, info -5). */
int psd_z_pschur(psd_ctx* ctx, int n, int p, double* const* A, const uint8_t* S, char orient, int wantT, int wantZ,
                 int maxitfac, double* const* Z, double* alpha, double* beta, int32_t* ascale, int* schurindex,
                 psd_stats* stats, int32_t* sweeplog, int64_t maxlog, int* info);

/* pschur!(H1, Hs, S; wantT, wantZ, Q, maxitfac, rev) for ComplexF64 — generalized.jl:166-175 */
int psd_z_pschur_hess(psd_ctx* ctx, int n, int p, double* const* H, const uint8_t* S, double* const* Q, int wantT,
                      int wantZ, int maxitfac, double* alpha, double* beta, int32_t* ascale, psd_stats* stats,
                      int32_t* sweeplog, int64_t maxlog, int* info);

/* pschur!(H1, Hs, S; wantT, wantZ, Q, maxitfac) for Float64 with a signature — rgeneralized.jl:49-59 (MB03BD type
 * real periodic QZ).  H[0] Hessenberg, H[1..p-1] upper triangular, S[0] must be true (rgeneralized.jl:73, info -5);
 * Q (p matrices) is accumulated when wantZ; alpha is complex (2n doubles), values = alpha / beta * 2^ascale.
 * stats->reserved = ncase2 + 1000 * ncase3, ndefl2 = 2x2 blocks, nrqpass = zero-shift passes. */
int psd_d_gpschur_hess(psd_ctx* ctx, int n, int p, double* const* H, const uint8_t* S, double* const* Q, int wantT,
                       int wantZ, int maxitfac, double* alpha, double* beta, int32_t* ascale, psd_stats* stats,
                       int32_t* sweeplog, int64_t maxlog, int* info);

/* _phessenberg!(A, S; wantQ) for Float64 — generalized.jl:988-1082: signed periodic Hessenberg-triangular reduction.
 * A[l] is overwritten by H_l (H_1 Hessenberg, the others upper triangular), Q[l] (may be NULL) receives Qs[l]:
 * A_l = Q_l H_l Q_{l+1}' for S[l], A_l = Q_{l+1} H_l Q_l' for !S[l].  S[0] must be true (info -5). */
int psd_d_gphessenberg(psd_ctx* ctx, int n, int p, double* const* A, const uint8_t* S, double* const* Q,
                       psd_stats* stats, int* info);

/* pschur!(A, S, lr; wantZ, wantT) for Float64 — rgeneralized.jl:3-45.  User-order in/out as psd_d_pschur; the entry of
 * S that lands leftmost in the working order (S[0] for 'R', S[p-1] for 'L') must be true (info -5, :37). */
int psd_d_gpschur(psd_ctx* ctx, int n, int p, double* const* A, const uint8_t* S, char orient, int wantT, int wantZ,
                  int maxitfac, double* const* Z, double* alpha, double* beta, int32_t* ascale, int* schurindex,
                  psd_stats* stats, int* info);

/* _phessenberg!(A, S; wantQ) for ComplexF64 — generalized.jl:988-1082 (see psd_d_gphessenberg) */
int psd_z_gphessenberg(psd_ctx* ctx, int n, int p, double* const* A, const uint8_t* S, double* const* Q,
                       psd_stats* stats, int* info);

/* device-resident variant of psd_z_pschur */
int psd_z_pschur_dev(psd_ctx* ctx, int n, int p, double* dA, char orient, int wantT, int wantZ, int maxitfac,
                     double* dZ, double* alpha, double* beta, int32_t* ascale, int* schurindex, psd_stats* stats,
                     int32_t* sweeplog, int64_t maxlog, int* info);

/* LinearAlgebra.ordschur!(P::PeriodicSchur{ComplexF64}, select; wantZ) — ordschur.jl:11-73: move the selected
 * eigenvalues (select[j] != 0) and their subspace to the top by adjacent swaps (sylswap.jl:542-635).
 * T: p pointers, the full user-order factor list with T1 at position `schurindex`; Z: p pointers (or NULL with
 * !wantZ); both updated in place.  schurindex must be 1 or p (info -7 otherwise, ordschur.jl:32); all four
 * alignments of the reference (utils.jl:6-85: _rev_alias, _circshift) are mapped onto the same working form.
 * Eigenvalues are recomputed from the diagonals (ordschur.jl:97-120) in scaled form.
 * info: 0; PSD_INFO_ILLCOND + j  IllConditionedException(j) (ordschur.jl:61); PSD_INFO_SINGULAR (utils.jl:128). */
#define PSD_INFO_ILLCOND 2000
#define PSD_INFO_SINGULAR 3000
int psd_z_ordschur(psd_ctx* ctx, int n, int p, double* const* T, double* const* Z, char orient, int schurindex,
                   const uint8_t* select, int wantZ, double* alpha, double* beta, int32_t* ascale, psd_stats* stats,
                   int* info);

/* LinearAlgebra.ordschur!(P::PeriodicSchur{Float64}, select; wantZ) — rordschur.jl:3-132 (1x1 and 2x2 blocks;
 * a selected member of a conjugate pair takes its partner along, :46-75).  Same conventions as psd_z_ordschur;
 * eigenvalues are returned as wr + i*wi (ordschur.jl:122-204). */
int psd_d_ordschur(psd_ctx* ctx, int n, int p, double* const* T, double* const* Z, char orient, int schurindex,
                   const uint8_t* select, int wantZ, double* wr, double* wi, psd_stats* stats, int* info);

/* LinearAlgebra.ordschur!(P::GeneralizedPeriodicSchur, select; wantZ) by adjacent 1x1 swaps — ordschur.jl:11-96,
 * 323-328, signed swap sylswap.jl:638-764.  T/Z: p matrices in the user order of the decomposition (T1 at
 * `schurindex`), S the user-order signature; eigenvalues are recomputed in the scaled form.  info as psd_z_ordschur.
 * psd_d_gordschur (Float64): rordschur.jl:3-132,141-268 with the signed block swap sylswap.jl:197-538 for 2x2 blocks
 * (conjugate pairs; `select` is completed to pairs) and ordschur.jl:206-314 for the eigenvalues; a decomposition with
 * a real spectrum (T1 triangular) goes through the 1x1 kernel. */
int psd_z_gordschur(psd_ctx* ctx, int n, int p, double* const* T, double* const* Z, const uint8_t* S, char orient,
                    int schurindex, const uint8_t* select, int wantZ, double* alpha, double* beta, int32_t* ascale,
                    psd_stats* stats, int* info);
int psd_d_gordschur(psd_ctx* ctx, int n, int p, double* const* T, double* const* Z, const uint8_t* S, char orient,
                    int schurindex, const uint8_t* select, int wantZ, double* alpha, double* beta, int32_t* ascale,
                    psd_stats* stats, int* info);

/* _rphessenberg!(Ap, A, Q) — rhessx.jl:55-109 (called by the Krylov driver, krylov.jl:809): row-wise periodic Hessenberg
 * reduction for the left orientation.  Ap: m x n column-major (ld m), m = n or n + 1 (info -2 otherwise, :62), reduced to
 * upper Hessenberg form; A: p-1 pointers to n x n matrices, reduced to upper triangular; Q: p pointers to nq x nqc
 * matrices (nqc >= n), post-multiplied by the accumulated reflectors, or NULL.  All updated in place.
 * psd_z_*: ComplexF64 (interleaved). */
int psd_d_rphessenberg(psd_ctx* ctx, int m, int n, int p, double* Ap, double* const* A, double* const* Q, int nq, int nqc,
                       int* info);
int psd_z_rphessenberg(psd_ctx* ctx, int m, int n, int p, double* Ap, double* const* A, double* const* Q, int nq, int nqc,
                       int* info);

/* checkpsd(P, As; thresh, strict) — diagnostics.jl:190-263 — as a device-side verifier (matrix cores: three n x n x n
 * FP64 products per factor).  T, Z, A: p pointers each, USER order of the decomposition (T[schurindex-1] is the
 * quasi-triangular factor), S the user-order signature (NULL = all true), wi (real variant, may be NULL) the imaginary
 * parts of P.values (non-zero subdiagonals below real eigenvalues are only a warning in the reference, :223-230).
 * err[p]: ||Z_a T_l Z_b' - A_l||_F / eps / opnorm(A_l, 1) with (a, b) by the signature and orientation (:247-252);
 * orth[p] / tri[p] (may be NULL): ||Z_l Z_l' - I||_F and ||tril(T_l, -1 or -2)||_F; *ok: the reference's Bool
 * (tri <= (strict ? 0 : 10 eps n), orth <= 10 eps n, err <= thresh).  psd_z_*: ComplexF64 interleaved.
 * psd_d_checkpsd_dev: operands already in HBM as [p][n][n] blocks in user order. */
int psd_d_checkpsd(psd_ctx* ctx, int n, int p, double* const* T, double* const* Z, double* const* A, const uint8_t* S,
                   char orient, int schurindex, const double* wi, double thresh, int strict, double* err, double* orth,
                   double* tri, int* ok, int* info);
int psd_z_checkpsd(psd_ctx* ctx, int n, int p, double* const* T, double* const* Z, double* const* A, const uint8_t* S,
                   char orient, int schurindex, double thresh, int strict, double* err, double* orth, double* tri, int* ok,
                   int* info);
int psd_d_checkpsd_dev(psd_ctx* ctx, int n, int p, const double* dT, const double* dZ, const double* dA,
                       const uint8_t* S, char orient, int schurindex, double thresh, int strict, double* err,
                       double* orth, double* tri, int* ok, int* info);

#ifdef __cplusplus
}
#endif
#endif
