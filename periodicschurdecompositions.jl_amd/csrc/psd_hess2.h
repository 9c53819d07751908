// Periodic Hessenberg-triangular reduction, look-ahead form (real, p >= 3): ONE launch per chain link.
//
// Replaces the loop body of phessenberg!(A) — /root/reference/src/PeriodicSchurDecompositions.jl:229-247 — with the
// reflector algebra of /root/reference/src/householder.jl:66-108 (_xreflector!) and :207-237 (rmul!(A,H), lmul!(H',A)).
//
// The reduction is a serial chain of (n-1) p reflector generations: link q = (column i, factor j) needs column i of A_j
// AFTER the right update by the previous link's reflector — and only that column.  So the chain and the bandwidth work
// are separated (psd_hess.h runs two dependent launches per link and makes the chain wait for the whole panel update):
//
//   chain#q   (blocks [0, nC), 8-row strips of M_{q+1}, the next link's matrix): forms v_q from the unscaled column
//             and the partial norms the previous launch left in the ring, w = M_{q+1}[rows, r0_q:n] v_q (a GEMV: the
//             block is only READ), and from it the next link's column  M_{q+1}[rows, r0_q] - tau_q w  plus its partial
//             norms.  Block 0 also stores v_q LAPACK-style into A_j and publishes (v_q, tau_q) for the bulk.
//   bulk B(q-1)  (the remaining blocks): the deferred update of the matrix of link q-1,
//             M_{q-1} <- H(v_{q-1})' (M_{q-1} H(v_{q-2})), fused into ONE pass over the matrix: every element is read
//             once and written once for both reflectors (bottom rows: per column  a - tau w v_c,  then the left
//             reflector, column in registers; rows above the reflector: fused GEMV + rank-one update per 8-row strip).
//
// Dependencies: B(q-1) needs v_{q-1} (made by the previous launch) and w of link q-2; chain#q reads M_{q+1} = M_{q+1-p},
// last written by B(q+1-p) in launch q+2-p <= q-1 for p >= 3 (p <= 2 keeps the two-launch form of psd_hess.h).  All
// per-link data lives in a 4-slot ring indexed by q & 3, so a launch never writes a slot another part of it reads.
// HBM traffic per link: 8 m^2 (GEMV) + 16 n m (fused update) against the 16 (n m + m^2) of the two separate
// panel updates the algorithmic-bytes figure counts.
#pragma once
#include "psd_scalar.h"

#ifndef PSD_HOSTSIM
#define PSD_H2_NT 256
#define PSD_H2_ROWS 8
#define PSD_H2_RING 256  // slots of the per-link ring (power of two)

// The bounded wait of the pipe form (hand-over records that do not validate): bounded in TIME, not in poll rounds — a
// predecessor launch can be held back by a tracer that serialises dispatches, by another process on the GPU or by a long
// panel batch on the masked compute units, and a bound in rounds (2^17 of them, about a second when nothing else runs)
// would then give up on a healthy chain.  s_memrealtime counts at 100 MHz; the clock is looked at every 256 rounds only,
// the first look sets the start.  About three seconds, then the error word is set, every later launch stops waiting and
// the host returns PSD_INFO_RUNTIME: the factors passed in are destroyed at that point (the reduction works in place).
#define PSD_H2_WAIT_TICKS 300000000LL
__device__ __forceinline__ bool psd_h2_wait_expired(int& spins, long long& t0) {
    if ((++spins & 255) != 0) return false;
    const long long now = (long long)__builtin_amdgcn_s_memrealtime();
    if (t0 == 0) {
        t0 = now;
        return false;
    }
    return now - t0 > PSD_H2_WAIT_TICKS;
}

struct psd_hess2_args {
    double* H;     // [p][n][n]
    double* tau;   // [p][n]
    double* ring;  // ringmask + 1 slots of psd_h2_slot_doubles(n)
    int p;         // period
    int ringmask;  // slots - 1 (a power of two minus one: 3 when the panel updates ride one launch behind the chain)
    int xcd;       // 1: chain blocks whose strips share a 128-byte line of the matrix run on the same XCD (see psd_hess2_link)
    long long* trace;  // diagnostics (PSD_H2_TRACE): [1024][8] 100 MHz stamps of one chain block per link, or nullptr
    int trace_hi;      // only links with ring position below this are recorded (PSD_H2_TRACE=<links>)
    int pipe;          // 1: consecutive chain launches overlap (two streams); the staged column travels as self-validating records (see psd_h2_tag)
    int* err;          // pipe: set when a launch gave up waiting for its predecessor's records (the host reports a runtime error)
    int fault;         // test hook (PSD_H2_FAULT=<link>): the records for this link are written with a wrong tag; -1: off
};
// slot layout: v[n+8] | w[n+8] | col[n+8] | hdr[8] (tau, beta) | part[2 * (n/4 + 2)] | rec[2 (n+8)]
PSD_HD size_t psd_h2_slot_doubles(int n) { return 5 * (size_t)(n + 8) + 8 + 2 * (size_t)(n / 4 + 2); }
struct psd_h2_slot {
    double *v, *w, *col, *hdr, *part;
    unsigned long long* rec;  // pipe mode: the staged column as (bits(x), bits(x) ^ tag) pairs
};
// Pipe mode.  A chain launch needs of its predecessor only the staged column.  With the launches of consecutive links on two
// streams the next launch is resident and has its strip of the matrix in registers while the previous one still runs; it
// then polls the column.  Every entry is a record (bits(x), bits(x) ^ tag(link)) written and read with 8-byte agent-scope
// accesses (no fences, no read-modify-write; measured 3.7 us per hand-over against 10 us for a counter barrier,
// tools/micro/grid_barrier.cpp): whatever mixture of old and new halves a reader sees fails the check unless the value
// is the one the producer wrote (a slot is reused 256 links later, its old records carry another tag).
PSD_HD unsigned long long psd_h2_tag(int slot) { return ((unsigned long long)(unsigned)(slot + 3)) * 0x9E3779B97F4A7C15ull | 1ull; }
PSD_D psd_h2_slot psd_h2_get(const psd_hess2_args* G, int n, int q) {
    double* b = G->ring + (size_t)(q & G->ringmask) * psd_h2_slot_doubles(n);
    psd_h2_slot s;
    s.v = b;
    s.w = b + (n + 8);
    s.col = b + 2 * (size_t)(n + 8);
    s.hdr = b + 3 * (size_t)(n + 8);
    s.part = s.hdr + 8;
    s.rec = (unsigned long long*)(s.part + 2 * (size_t)(n / 4 + 2));
    return s;
}
struct psd_h2_link {
    int valid, i, j, r0;  // 1-based column, factor, first row of the reflector
};
// link (i, j) shifted by d in {-2, -1, 0, +1} chain positions (the order is j = p..1 inside a column, then i + 1);
// (i, j) itself may lie one or two positions outside the chain (staging launch, drain launches)
PSD_D psd_h2_link psd_h2_linkat(int i, int j, int d, int n, int p) {
    int jj = j - d, ii = i;
    if (jj < 1) {
        jj += p;
        ii += 1;
    } else if (jj > p) {
        jj -= p;
        ii -= 1;
    }
    psd_h2_link L;
    L.valid = (ii >= 1 && ii <= n - 1) ? 1 : 0;
    L.i = ii;
    L.j = jj;
    L.r0 = (jj == 1) ? (ii + 1) : ii;
    return L;
}

typedef double psd_h2_v2 __attribute__((ext_vector_type(2), aligned(8)));
typedef unsigned psd_h2_u4 __attribute__((ext_vector_type(4)));

// two consecutive rows (r, r+1) of column c (0-based); the second is masked at the bottom edge
PSD_D void psd_h2_ld2(const double* M, int n, int r, int c, bool ok0, bool ok1, double& x0, double& x1) {
    x0 = x1 = 0.0;
    if (ok0 && ok1) {
        const psd_h2_v2 t = *(const psd_h2_v2*)(M + (size_t)c * n + r);
        x0 = t.x;
        x1 = t.y;
    } else if (ok0) {
        x0 = M[(size_t)c * n + r];
    } else if (ok1) {
        x1 = M[(size_t)c * n + r + 1];
    }
}
PSD_D void psd_h2_st2(double* M, int n, int r, int c, bool ok0, bool ok1, double x0, double x1) {
    if (ok0 && ok1) {
        psd_h2_v2 t;
        t.x = x0;
        t.y = x1;
        *(psd_h2_v2*)(M + (size_t)c * n + r) = t;
    } else if (ok0) {
        M[(size_t)c * n + r] = x0;
    } else if (ok1) {
        M[(size_t)c * n + r + 1] = x1;
    }
}

PSD_D double psd_h2_wave_sum(double x) {
#pragma unroll
    for (int s = 32; s > 0; s >>= 1) x += __shfl_xor(x, s, 64);
    return x;
}
PSD_D double psd_h2_wave_max(double x) {
#pragma unroll
    for (int s = 32; s > 0; s >>= 1) x = fmax(x, __shfl_xor(x, s, 64));
    return x;
}

// (tau, beta, mult) of the reflector of x = (alpha; tail) from the tail's scaled norm (householder.jl:77-105, dlarfg);
// tau = 0: H = I (householder.jl:74-76)
PSD_D void psd_h2_larfg(double alpha, double xnorm, double& tau, double& beta, double& mult) {
    if (xnorm == 0.0) {
        tau = 0.0;
        beta = alpha;
        mult = 0.0;
        return;
    }
    const double sfmin = 2.0 * PSD_DBL_MIN / PSD_DBL_EPS;
    {
        // The common case without an IEEE sqrt and three IEEE divisions (0.36 us of every link: PSD_H2_TRACE): both
        // magnitudes far from the range limits, so the squares neither overflow nor vanish and the hardware reciprocal /
        // reciprocal square root seeds with two Newton steps (psd_scalar.h) are as accurate as the divisions.
        const double aa = fabs(alpha), ww = fmax(aa, xnorm), mn = fmin(aa, xnorm);
        if (ww < 1e140 && ww > 1e-140 && (mn == 0.0 || mn > 1e-140)) {
            double nrm, rn;
            psd_sqrt_pair_fast(alpha * alpha + xnorm * xnorm, nrm, rn);
            beta = -copysign(nrm, alpha);
            tau = 1.0 + aa * rn;                              // (beta - alpha) / beta
            mult = psd_rcp_fast(copysign(aa + nrm, alpha));   // 1 / (alpha - beta)
            return;
        }
        // dlapy2 (the library hypot costs several hundred cycles on the chain)
        const double zz = mn / ww;
        beta = -copysign(ww * sqrt(1.0 + zz * zz), alpha);
    }
    if (fabs(beta) >= sfmin) {
        tau = (beta - alpha) / beta;
        mult = 1.0 / (alpha - beta);
        return;
    }
    int kount = 0;
    double acc = 1.0;
    if (fabs(beta) < sfmin) {
        const double rsfmin = 1.0 / sfmin;
        bool smallb = true;
        while (smallb) {
            kount += 1;
            acc *= rsfmin;
            beta *= rsfmin;
            alpha *= rsfmin;
            smallb = (fabs(beta) < sfmin) && (kount < 20);
        }
        xnorm *= acc;
        beta = -copysign(hypot(alpha, xnorm), alpha);
    }
    tau = (beta - alpha) / beta;
    mult = acc * (1.0 / (alpha - beta));
    for (int q = 0; q < kount; ++q) beta *= sfmin;
}

// Bottom part of a panel update with CPW columns per wavefront (two-stream forms, NK <= 16): the two vectors of the link
// (w of the right reflector, v of the left one: 16 + 16 doubles per lane) stay in registers for all of the wave's columns,
// and the next column is requested before the current one is reduced.  One column per wave read both vectors again for
// every column (two thirds of its load instructions, all L2 traffic) and had one column's loads in flight per wave:
// the panel kernel alone ran at 2.3 TB/s on its 192 CUs (PSD_H2_BULKBENCH).
template <int NKC>
PSD_D void psd_h2_col_load(const double* M, int n, int R0, int mL, int c, int lane, double (&a0)[NKC], double (&a1)[NKC]) {
#pragma unroll
    for (int k = 0; k < NKC; ++k) {
        const int rr = 2 * lane + 128 * k;
        a0[k] = a1[k] = 0.0;
        if (rr < mL) psd_h2_ld2(M, n, R0 + rr, c, true, rr + 1 < mL, a0[k], a1[k]);
    }
}
template <int NKC>
PSD_D void psd_h2_col_update(double* M, int n, int R0, int mL, int c, int lane, double vc, double tauL, const double (&w0)[NKC],
                             const double (&w1)[NKC], const double (&v0)[NKC], const double (&v1)[NKC], double (&a0)[NKC], double (&a1)[NKC]) {
    double z = 0.0;
#pragma unroll
    for (int k = 0; k < NKC; ++k) {  // (entries outside the reflector are zero in a, w and v alike)
        a0[k] = __builtin_fma(-vc, w0[k], a0[k]);
        a1[k] = __builtin_fma(-vc, w1[k], a1[k]);
        z = __builtin_fma(a0[k], v0[k], z);
        z = __builtin_fma(a1[k], v1[k], z);
    }
    z = tauL * psd_h2_wave_sum(z);
#pragma unroll
    for (int k = 0; k < NKC; ++k) {
        const int rr = 2 * lane + 128 * k;
        if (rr < mL) psd_h2_st2(M, n, R0 + rr, c, true, rr + 1 < mL, __builtin_fma(-z, v0[k], a0[k]), __builtin_fma(-z, v1[k], a1[k]));
    }
}
template <int NK, int CPW>
PSD_D void psd_h2_bulk_bottom_multi(double* M, int n, const psd_h2_slot& SR, const psd_h2_slot& SL, double tauR, double tauL, int rR, int R0,
                                    int ci, int B, int lane, int wave) {
    constexpr int NKC = (NK + 1) / 2;
    const int mL = n - R0;
    const int c0 = ci + (B * 4 + wave) * CPW;
    if (c0 >= n) return;
    double a0[NKC], a1[NKC], b0[NKC], b1[NKC];
    psd_h2_col_load<NKC>(M, n, R0, mL, c0, lane, a0, a1);
    double w0[NKC], w1[NKC], v0[NKC], v1[NKC], vcs[CPW];
#pragma unroll
    for (int j = 0; j < CPW; ++j) {
        const int c = c0 + j;
        vcs[j] = (tauR != 0.0 && c < n && c >= rR) ? tauR * SR.v[c - rR] : 0.0;
    }
#pragma unroll
    for (int k = 0; k < NKC; ++k) {
        const int rr = 2 * lane + 128 * k;
        w0[k] = (tauR != 0.0 && rr < mL) ? SR.w[R0 + rr] : 0.0;
        w1[k] = (tauR != 0.0 && rr + 1 < mL) ? SR.w[R0 + rr + 1] : 0.0;
        v0[k] = (tauL != 0.0 && rr < mL) ? SL.v[rr] : 0.0;
        v1[k] = (tauL != 0.0 && rr + 1 < mL) ? SL.v[rr + 1] : 0.0;
    }
#pragma unroll
    for (int j = 0; j < CPW; j += 2) {  // (column c0 + j is in a0/a1 here; no early exit, so that the loop unrolls)
        const int c = c0 + j;
        if (c < n) {
            const bool more1 = j + 1 < CPW && c + 1 < n;
            if (more1) psd_h2_col_load<NKC>(M, n, R0, mL, c + 1, lane, b0, b1);
            psd_h2_col_update<NKC>(M, n, R0, mL, c, lane, vcs[j], tauL, w0, w1, v0, v1, a0, a1);
            if (more1) {
                if (j + 2 < CPW && c + 2 < n) psd_h2_col_load<NKC>(M, n, R0, mL, c + 2, lane, a0, a1);
                psd_h2_col_update<NKC>(M, n, R0, mL, c + 1, lane, vcs[(j + 1 < CPW) ? j + 1 : j], tauL, w0, w1, v0, v1, b0, b1);
            }
        }
    }
}

// The deferred update of the matrix of link Lb: M <- H(v_Lb)' (M H(v_La)), La the link before Lb (either may lie outside
// the chain: first link / drain).  bb: block index inside the update (nT row strips above the reflector, then the
// 4-column groups of the bottom part); slotb: ring position of Lb.
template <int NK, int CPW = 1>
PSD_D void psd_h2_bulk_body(const psd_hess2_args* G, int n, const psd_h2_link Lb, const psd_h2_link La, int slotb, int bb, int nT,
                            double* vs, double* red) {
    const int p = G->p;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    if (!Lb.valid && !La.valid) return;
    // matrix of link qb (after the last link: the matrix the last reflector acts on from the right, A_p)
    const int jb = Lb.valid ? Lb.j : p;
    double* M = G->H + (size_t)(jb - 1) * n * n;
    const psd_h2_slot SR = psd_h2_get(G, n, slotb - 1), SL = psd_h2_get(G, n, slotb);
    const double tauR = La.valid ? SR.hdr[0] : 0.0;
    const double tauL = Lb.valid ? SL.hdr[0] : 0.0;
    const int rR = La.valid ? (La.r0 - 1) : 0;      // right reflector acts on columns rR..n-1
    const int mR = n - rR;
    const int R0 = Lb.valid ? (Lb.r0 - 1) : n;      // left reflector acts on rows R0..n-1
    if (bb < nT) {
        // rows above the left reflector: fused GEMV + rank-one update, 8-row strips (right reflector only)
        if (tauR == 0.0) return;
        const int t = bb;
        if (PSD_H2_ROWS * t >= R0) return;
        const int rp = tid & 3, cl = tid >> 2;
        const int r = PSD_H2_ROWS * t + 2 * rp;
        const bool ok0 = r < R0, ok1 = r + 1 < R0;
        double a0[NK], a1[NK];
#pragma unroll
        for (int k = 0; k < NK; ++k) {  // the panel is requested before the reflector is staged
            const int cc = cl + 64 * k;
            a0[k] = a1[k] = 0.0;
            if (cc < mR) psd_h2_ld2(M, n, r, rR + cc, ok0, ok1, a0[k], a1[k]);
        }
        for (int k = tid; k < mR; k += PSD_H2_NT) vs[k] = SR.v[k];
        __syncthreads();
        double acc0 = 0.0, acc1 = 0.0;
#pragma unroll
        for (int k = 0; k < NK; ++k) {
            const int cc = cl + 64 * k;
            if (cc < mR) {
                const double vv = vs[cc];
                acc0 += a0[k] * vv;
                acc1 += a1[k] * vv;
            }
        }
#pragma unroll
        for (int s = 4; s < 64; s <<= 1) {
            acc0 += __shfl_xor(acc0, s, 64);
            acc1 += __shfl_xor(acc1, s, 64);
        }
        if (lane < 4) {
            red[(wave * 4 + lane) * 2] = acc0;
            red[(wave * 4 + lane) * 2 + 1] = acc1;
        }
        __syncthreads();
        const int k0 = rp * 2;
        const double w0 = tauR * (red[k0] + red[8 + k0] + red[16 + k0] + red[24 + k0]);
        const double w1 = tauR * (red[k0 + 1] + red[8 + k0 + 1] + red[16 + k0 + 1] + red[24 + k0 + 1]);
#pragma unroll
        for (int k = 0; k < NK; ++k) {
            const int cc = cl + 64 * k;
            if (cc < mR) {
                const double vv = vs[cc];
                psd_h2_st2(M, n, r, rR + cc, ok0, ok1, a0[k] - w0 * vv, a1[k] - w1 * vv);
            }
        }
        return;
    }
    // rows R0..n-1: one wavefront per column, the column in registers: a <- a - tauR w v_c, then the left reflector
    if (!Lb.valid) return;
    if (tauR == 0.0 && tauL == 0.0) return;
    if constexpr (CPW > 1) {
        psd_h2_bulk_bottom_multi<NK, CPW>(M, n, SR, SL, tauR, tauL, rR, R0, Lb.i, bb - nT, lane, wave);
        return;
    }
    const int c = Lb.i + 4 * (bb - nT) + wave;  // 0-based column: the columns right of the reflector's (Lb.i - 1)
    if (c >= n) return;
    const int mL = n - R0;
    constexpr int NKC = (NK + 1) / 2;  // row pairs per lane: 128 rows per step
    double a0[NKC], a1[NKC];
    const double vc = (tauR != 0.0 && c >= rR) ? tauR * SR.v[c - rR] : 0.0;
    double z = 0.0;
#pragma unroll
    for (int k = 0; k < NKC; ++k) {
        const int rr = 2 * lane + 128 * k;  // offset inside the reflector
        a0[k] = a1[k] = 0.0;
        if (rr < mL) {
            const bool ok1 = rr + 1 < mL;
            psd_h2_ld2(M, n, R0 + rr, c, true, ok1, a0[k], a1[k]);
            if (vc != 0.0) {
                a0[k] -= vc * SR.w[R0 + rr];
                if (ok1) a1[k] -= vc * SR.w[R0 + rr + 1];
            }
            if (tauL != 0.0) {
                z += a0[k] * SL.v[rr];
                if (ok1) z += a1[k] * SL.v[rr + 1];
            }
        }
    }
    z = tauL * psd_h2_wave_sum(z);
#pragma unroll
    for (int k = 0; k < NKC; ++k) {
        const int rr = 2 * lane + 128 * k;
        if (rr < mL) {
            const bool ok1 = rr + 1 < mL;
            double x0 = a0[k], x1 = a1[k];
            if (tauL != 0.0) {
                x0 -= z * SL.v[rr];
                if (ok1) x1 -= z * SL.v[rr + 1];
            }
            psd_h2_st2(M, n, R0 + rr, c, true, ok1, x0, x1);
        }
    }
}

// One launch = chain#q + bulk B(q-1).  The link index q is a launch argument (no dependent load in front of the data
// loads; the launches are issued one by one: at 5-15 us each the host keeps up).  NK: ceil(n / 64) rounded up
// (register panel depth); CR: rows per chain strip (8, or 4 for large n: twice the workgroups on the GEMV).
// grid = nC + nT + nB blocks with nC = ceil(n / CR), nT = ceil(n / 8), nB = ceil(n / 4);
// LDS: (n + 8 + 2 * PSD_H2_NT + 64) doubles.
template <int NK, int CR>
__global__ void __launch_bounds__(PSD_H2_NT, ((NK > 16 || NK * CR > 128) ? ((NK * CR > 256) ? 1 : 2) : 4)) psd_hess2_link(const psd_hess2_args Gv, int n, int qi, int qj, int nC, int nT) {
    extern __shared__ __attribute__((aligned(16))) char psd_lds[];
    double* vs = (double*)psd_lds;       // n + 8: the reflector (v[0] = 1) the strip kernels multiply with
    double* red = vs + (n + 8);          // 2 * NT + 64
    const psd_hess2_args* G = &Gv;
    const int p = G->p;
    // chain position relative to the ends: q = -1 (staging), 0 .. Q-1 (links), Q, Q+1 (drain), given as (column, factor)
    const psd_h2_link L = psd_h2_linkat(qi, qj, 0, n, p);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int b = blockIdx.x;
    if (b < nC) {
        // ------------------------------------------------------------------ chain#q
        const psd_h2_link Ln = psd_h2_linkat(qi, qj, 1, n, p);
        if (!L.valid && !Ln.valid) return;  // drain launches have no chain part
        if (!L.valid && qi > 1) return;
        const int q = L.valid ? 0 : -1;  // (q = -1: the very first column is only staged)
        const int slot = qi * p - qj;    // ring position: any number that advances by one per link
        long long* const trc = (G->trace != nullptr && b == 2 && tid == 0 && slot < G->trace_hi) ? (G->trace + (size_t)(slot & 1023) * 8) : nullptr;
#define PSD_H2_STAMP(i) do { if (trc) trc[i] = (long long)__builtin_amdgcn_s_memrealtime(); } while (0)
        PSD_H2_STAMP(0);
        // (one link, every block: first and last stamp, and the XCD it ran on)
        long long* const blk = (G->trace != nullptr && tid == 0 && slot == G->trace_hi - 1 && b < 1024) ? (G->trace + 1024 * 8 + (size_t)b * 4) : nullptr;
        if (blk) {
            blk[0] = (long long)__builtin_amdgcn_s_memrealtime();
            blk[2] = (long long)__builtin_amdgcn_s_getreg(63492);  // (XCC_ID: hwreg 20, offset 0, size 4... diagnostic only)
        }
        const psd_h2_slot S = psd_h2_get(G, n, slot), Sn = psd_h2_get(G, n, slot + 1);
        const int r0 = L.r0 - 1;  // 0-based first row of v_q
        const int m = (q >= 0) ? (n - r0) : 0;
        constexpr int RP = CR / 2;                  // row pairs of a strip = lanes per column
        constexpr int CL = PSD_H2_NT / RP;          // column lanes
        constexpr int NKS = (64 * NK + CL - 1) / CL;  // column steps of a strip
        const int ntileC = (n + CR - 1) / CR;
        // Everything this block reads from memory is requested up front, so that the strip of the next matrix (HBM) is
        // in flight while the reflector is being formed: (half of) the strip itself, this thread's entries of the
        // unscaled column, its partial norm.
        const double* M = Ln.valid ? (G->H + (size_t)(Ln.j - 1) * n * n) : G->H;
        const int r0n = Ln.r0 - 1;
        const int cfirst = (q >= 0) ? r0 : (Ln.i - 1);  // column that becomes the next reflector's
        // block 0 forms and publishes v_q only; block b >= 1 takes strip t.  A strip of CR rows is CR * 8 = 32 or 64 bytes of
        // every column: GS = 128 / (CR * 8) neighbouring strips share each 128-byte line, and blocks are dealt round-robin
        // over the 8 XCDs (each with its own L2), so in strip order every line is fetched into 2 or 4 different L2s.  With
        // G.xcd the strips of one line go to blocks b, b + 8, .. (same XCD, dispatched together).
        int t = r0n / CR + (b - 1);
        if (G->xcd) {
            constexpr int GS = (CR < 16) ? 16 / CR : 1;
            const int idx = b - 1, grp0 = (r0n / CR) / GS;
            t = GS * (grp0 + (idx % 8) + 8 * (idx / (8 * GS))) + (idx / 8) % GS;
        }
        const bool strip = Ln.valid && b > 0 && t < ntileC && t >= r0n / CR;
        const int rp = tid % RP, cl = tid / RP;
        const int r = CR * t + 2 * rp;
        const bool ok0 = strip && r < n && r >= r0n, ok1 = strip && r + 1 < n && r + 1 >= r0n;
        constexpr int NK1 = (NKS + 1) / 2;  // steps requested before the reflector is formed
        double a0[NKS], a1[NKS];
        const int rfin = CR * t + tid;  // the row thread tid < CR finishes
        const bool fin = strip && tid < CR && rfin < n && rfin >= r0n;
        constexpr int NV = (64 * NK + PSD_H2_NT - 1) / PSD_H2_NT;  // entries of v per thread
        double xcol[NV];
        double tau = 0.0, beta = 0.0, mult = 0.0, xnorm = 0.0;
        double am = 0.0, sq = 0.0, alpha = 0.0;
        // (vector memory operations return in issue order: the small ring reads go first, so that forming the reflector
        //  does not wait for the strip)
        const bool pipe = G->pipe != 0;
        if (q >= 0 && !pipe) {
            const int np_ = ntileC - r0 / CR;
            if (tid < np_) {
                am = S.part[2 * tid];
                sq = S.part[2 * tid + 1];
            }
            alpha = S.col[r0];
#pragma unroll
            for (int u = 0; u < NV; ++u) {
                const int k = tid + PSD_H2_NT * u;
                xcol[u] = (k < m) ? S.col[r0 + k] : 0.0;
            }
        }
        const double mfirst = fin ? M[(size_t)cfirst * n + rfin] : 0.0;
        if (q >= 0) {
#pragma unroll
            for (int k = 0; k < NKS; ++k) {
                const int cc = cl + CL * k;
                a0[k] = a1[k] = 0.0;
                if (cc < m) psd_h2_ld2(M, n, r, r0 + cc, ok0, ok1, a0[k], a1[k]);
            }
        }
        if (q >= 0 && pipe) {
            // the strip is on its way; now the column of the previous launch (which may still be running)
            const unsigned long long tag = psd_h2_tag(slot);
            const unsigned long long* rec = S.rec + 2 * (size_t)r0;
            unsigned long long xb[NV], ab = 0;
            int spins = 0;
            long long wait_t0 = 0;
            // pipe == 1: first a cheap watch on one record per strip of the previous launch (thread t: the first row of
            // strip t at or below r0), then the column.  With 129 workgroups of 8-row strips polling the whole column costs
            // less than the extra round trip, so this is off by default (PSD_H2_POLL=1 turns it on)
            if (G->pipe == 1) {
                const int rw = CR * (r0 / CR + tid);
                const int kw = (rw > r0) ? (rw - r0) : 0;
                const bool watch = kw < m && (tid == 0 || rw > r0);
                for (;;) {
                    bool okr = true;
                    if (watch) {
                        const unsigned long long x = __hip_atomic_load(rec + 2 * kw, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        const unsigned long long cx = __hip_atomic_load(rec + 2 * kw + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        okr = (x ^ cx) == tag;
                    }
                    const bool giveup = __hip_atomic_load(G->err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0;
                    if (__syncthreads_and((okr || giveup) ? 1 : 0)) break;
                    if (psd_h2_wait_expired(spins, wait_t0)) {  // (every wave reaches an exit; the results are then void and the host says so)
                        if (tid == 0) __hip_atomic_store(G->err, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    }
                    __builtin_amdgcn_s_sleep(2);
                }
            }
            // (a record per 16-byte load, sc1 = coherent at agent scope like the atomic loads: a wavefront's 64 records are
            //  one contiguous KiB instead of 128 separate 8-byte requests; a torn 16-byte read fails the check like any other
            //  mixture of old and new halves)
            const __amdgpu_buffer_rsrc_t rsr = __builtin_amdgcn_make_buffer_rsrc((void*)rec, 0, m * 16, 0x00020000);
            for (;;) {
                bool okr = true;
#pragma unroll
                for (int u = 0; u < NV; ++u) {
                    const int k = tid + PSD_H2_NT * u;
                    xb[u] = 0;
                    if (PSD_H2_NT * u < m) {
                        const psd_h2_u4 t4 = __builtin_amdgcn_raw_buffer_load_b128(rsr, (k < m) ? (unsigned)k * 16u : 0xfffffff0u, 0, 16);
                        const unsigned long long x = ((unsigned long long)t4.y << 32) | t4.x, cx = ((unsigned long long)t4.w << 32) | t4.z;
                        okr = okr && (k >= m || (x ^ cx) == tag);
                        xb[u] = x;
                    }
                }
                asm volatile("" ::: "memory");
                const bool giveup = __hip_atomic_load(G->err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0;
                if (__syncthreads_and((okr || giveup) ? 1 : 0)) break;
                if (psd_h2_wait_expired(spins, wait_t0)) {  // (every wave reaches an exit; the results are then void and the host says so)
                    if (tid == 0) __hip_atomic_store(G->err, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                }
                __builtin_amdgcn_s_sleep(2);
            }
            // this thread's share of the tail (entries k >= 1): largest magnitude and PLAIN sum of squares — the scaled form
            // (dlassq) is only run when the magnitudes call for it, see below; alpha is thread 0's first entry
            am = 0.0;
            sq = 0.0;
#pragma unroll
            for (int u = 0; u < NV; ++u) {
                const int k = tid + PSD_H2_NT * u;
                xcol[u] = (k < m) ? __longlong_as_double((long long)xb[u]) : 0.0;
                if (k >= 1 && k < m) {
                    am = fmax(am, fabs(xcol[u]));
                    sq = __builtin_fma(xcol[u], xcol[u], sq);
                }
                // the RAW column goes to LDS (its first entry as zero): w = M v = M[:, 0] + mult (M[:, 1:] x[1:]), so the
                // strip can be multiplied while the norm is reduced and the reflector's scalars are formed
                if (k < m) vs[k] = (k == 0) ? 0.0 : xcol[u];
            }
            (void)ab;
        }
        if (q >= 0) {
            // norm of the tail from the partials of the launch that staged the column: (amax, ssq) pairs combine as
            // amax = max, ssq = sum ssq_k (amax_k / amax)^2 (dlassq); one pair per thread, one LDS exchange
            PSD_H2_STAMP(1);
            if (pipe) {
                // two independent wave reductions (largest magnitude, plain sum of squares), one LDS exchange; the root
                // without the IEEE wrapper when every square is far from the range limits, the scaled sum otherwise
                const double amw = psd_h2_wave_max(am);
                const double s2w = psd_h2_wave_sum(sq);
                if (lane == 0) {
                    red[2 * wave] = amw;
                    red[2 * wave + 1] = s2w;
                }
                if (tid == 0) red[8] = xcol[0];
                __syncthreads();
                alpha = red[8];
                const double amax = fmax(fmax(red[0], red[2]), fmax(red[4], red[6]));
                const double s2 = (red[1] + red[3]) + (red[5] + red[7]);
                if (m > 1 && amax > 0.0) {
                    if (amax < 1e140 && amax > 1e-140) {
                        double g, rg;
                        psd_sqrt_pair_fast(s2, g, rg);
                        xnorm = g;
                    } else {
                        __syncthreads();
                        double ss = 0.0;
#pragma unroll
                        for (int u = 0; u < NV; ++u) {
                            const int k = tid + PSD_H2_NT * u;
                            if (k >= 1 && k < m) {
                                const double z = xcol[u] / amax;
                                ss += z * z;
                            }
                        }
                        ss = psd_h2_wave_sum(ss);
                        if (lane == 0) red[16 + wave] = ss;
                        __syncthreads();
                        xnorm = amax * sqrt((red[16] + red[17]) + (red[18] + red[19]));
                    }
                }
            } else {
                const double amw = psd_h2_wave_max(am);
                double ssw = 0.0;
                if (amw > 0.0) {
                    const double f = am / amw;
                    ssw = sq * (f * f);
                }
                ssw = psd_h2_wave_sum(ssw);
                if (lane == 0) {
                    red[2 * wave] = amw;
                    red[2 * wave + 1] = ssw;
                }
                __syncthreads();
                const double amax = fmax(fmax(red[0], red[2]), fmax(red[4], red[6]));
                double tot = 0.0;
                if (amax > 0.0) {
#pragma unroll
                    for (int wv = 0; wv < 4; ++wv) {
                        const double f = red[2 * wv] / amax;
                        tot += red[2 * wv + 1] * (f * f);
                    }
                }
                xnorm = (m > 1) ? amax * sqrt(tot) : 0.0;
            }
            PSD_H2_STAMP(2);
            if (!pipe || b == 0) psd_h2_larfg(alpha, xnorm, tau, beta, mult);  // (pipe: the strip workgroups form the scalars behind their GEMV, below)
            PSD_H2_STAMP(3);
            if (!pipe) {
#pragma unroll
                for (int u = 0; u < NV; ++u) {
                    const int k = tid + PSD_H2_NT * u;
                    if (k < m) vs[k] = (k == 0) ? 1.0 : xcol[u] * mult;
                }
                __syncthreads();
            }
            if (b == 0) {  // publish v_q, store it LAPACK-style (PSD.jl:232-236,241-244)
                double* Mq = G->H + (size_t)(L.j - 1) * n * n;
                const int c = L.i - 1;
#pragma unroll
                for (int u = 0; u < NV; ++u) {
                    const int k = tid + PSD_H2_NT * u;
                    if (k < m) {
                        const double vk = (k == 0) ? 1.0 : xcol[u] * mult;
                        S.v[k] = vk;
                        Mq[(size_t)c * n + r0 + k] = (tau != 0.0) ? ((k == 0) ? beta : vk) : xcol[u];
                    }
                }
                if (tid == 0) {
                    S.hdr[0] = tau;
                    S.hdr[1] = beta;
                    G->tau[(size_t)(L.j - 1) * n + (L.i - 1)] = tau;
                }
            }
        }
        if (!strip) return;
        // GEMV on the next link's matrix: rows >= r0n, columns r0.. (q = -1: plain staging of column 1 of A_p)
        PSD_H2_STAMP(4);
        double acc0 = 0.0, acc1 = 0.0;
        if (q >= 0 && (pipe || tau != 0.0)) {
#pragma unroll
            for (int k = 0; k < NKS; ++k) {
                const int cc = cl + CL * k;
                if (cc < m) {
                    const double vv = vs[cc];
                    acc0 += a0[k] * vv;
                    acc1 += a1[k] * vv;
                }
            }
        }
        // reduce over the column lanes of a row pair: inside a wave (lanes with equal lane % RP), then the 4 waves
        // (pipe: its own part of the exchange area — workgroup mates may still be reading the norm's)
        double* const redg = pipe ? (red + 64) : red;
#pragma unroll
        for (int sft = RP; sft < 64; sft <<= 1) {
            acc0 += __shfl_xor(acc0, sft, 64);
            acc1 += __shfl_xor(acc1, sft, 64);
        }
        if (lane < RP) {
            redg[(wave * RP + lane) * 2] = acc0;
            redg[(wave * RP + lane) * 2 + 1] = acc1;
        }
        if (pipe && q >= 0) psd_h2_larfg(alpha, xnorm, tau, beta, mult);
        __syncthreads();
        PSD_H2_STAMP(5);
        double amt = 0.0, y = 0.0;
        if (fin) {
            double w = redg[tid] + redg[CR + tid] + redg[2 * CR + tid] + redg[3 * CR + tid];
            if (pipe && q >= 0) w = (tau != 0.0) ? __builtin_fma(mult, w, mfirst) : 0.0;  // (the unit entry of v times column r0)
            y = mfirst - tau * w;
            if (pipe) {
                const unsigned long long yb = (unsigned long long)__double_as_longlong(y);
                __hip_atomic_store(Sn.rec + 2 * (size_t)rfin, yb, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                __hip_atomic_store(Sn.rec + 2 * (size_t)rfin + 1, yb ^ psd_h2_tag(slot + 1) ^ ((G->fault == slot + 1) ? 0x10ull : 0ull), __ATOMIC_RELAXED,
                                   __HIP_MEMORY_SCOPE_AGENT);
            } else {
                Sn.col[rfin] = y;
            }
            S.w[rfin] = w;
            if (rfin > r0n) amt = fabs(y);
            else y = 0.0;  // the reflector's first entry is not part of the tail
        }
        if (wave == 0 && !pipe) {
            // partial scaled sum of squares of this strip's tail entries (lanes 0..CR-1 carry them)
            const double amax = psd_h2_wave_max(amt);
            double sq = 0.0;
            if (amax > 0.0 && amt > 0.0) {
                const double z = y / amax;
                sq = z * z;
            }
            sq = psd_h2_wave_sum(sq);
            if (lane == 0) {
                Sn.part[2 * (t - r0n / CR)] = amax;  // (partial norms by strip, whichever block computed them)
                Sn.part[2 * (t - r0n / CR) + 1] = sq;
            }
        }
        PSD_H2_STAMP(6);
        if (blk) blk[1] = (long long)__builtin_amdgcn_s_memrealtime();
        return;
    }
    // ---------------------------------------------------------------------- bulk B(q-1) on M_{q-1}
    psd_h2_bulk_body<NK>(G, n, psd_h2_linkat(qi, qj, -1, n, p), psd_h2_linkat(qi, qj, -2, n, p), qi * p - qj - 1, b - nC, nT, vs, red);
}

// The updates of K consecutive links in one launch (K distinct matrices: K <= p), for the two-stream form in which the
// chain launches carry no bulk part.  idx0: chain index of the first link ((i - 1) p + (p - j)); Q = (n - 1) p itself is
// the drain position (the last right reflector on A_p).  grid = (nT + nB, K).
template <int NK, int CPW = (NK <= 16 ? 4 : 1)>
__global__ void __launch_bounds__(PSD_H2_NT, (NK > 16 ? 2 : (CPW > 1 ? 3 : 4))) psd_hess2_bulk(const psd_hess2_args Gv, int n, int idx0, int nT) {
    extern __shared__ __attribute__((aligned(16))) char psd_lds[];
    double* vs = (double*)psd_lds;
    double* red = vs + (n + 8);
    const int p = Gv.p;
    const int idx = idx0 + (int)blockIdx.y;
    const int Q = (n - 1) * p;
    if (idx > Q) return;
    psd_h2_link Lb, La;
    Lb.i = idx / p + 1;
    Lb.j = p - idx % p;
    Lb.valid = (idx < Q) ? 1 : 0;
    Lb.r0 = (Lb.j == 1) ? (Lb.i + 1) : Lb.i;
    La = psd_h2_linkat(Lb.i, Lb.j, -1, n, p);
    psd_h2_bulk_body<NK, CPW>(&Gv, n, Lb, La, idx, (int)blockIdx.x, nT, vs, red);
}

#endif
