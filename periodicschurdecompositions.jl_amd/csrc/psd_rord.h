// Eigenvalue reordering on the GPU, Float64: ordschur!(P, select) with 1x1 and 2x2 blocks.
//
// Replaces /root/reference/src/rordschur.jl:3-132 (driver, conjugate-pair aware scan), :141-251
// (_moveblock!), sylswap.jl:14-157 (_swapadjqr!: periodic Sylvester solution -> per-factor
// orthogonal m x m transformation from QR of [X; I], fill-in repair by a 2x2 periodic Hessenberg
// reduction :159-191 / rpschur2x2.jl:326-359, strong stability test), sylswap.jl:542-635 for the
// 1x1/1x1 case, sylvester.jl:170-193 + babd.jl (cyclic block-bidiagonal solve), ordschur.jl:122-204
// (_updateλ!) with rpschur2x2.jl:9-275 (_rpeigvals2x2) for the conjugate pairs.
//
// MI355X structure (as psd_zord.h): a block travelling upwards is a chase; one wavefront keeps the
// diagonal window of all p factors in LDS, performs as many adjacent swaps as fit, and emits one
// dense m x m (m <= 4) orthogonal block transform per factor and swap; psd_rord_apply updates the
// off-window rows of T_m, columns of T_{m-1} and Z_m at bandwidth.  The small dense algebra of a swap runs out of
// LDS: the periodic Sylvester system is reduced by a tree of stacked-pair Householder QRs (log2 p levels, see
// psd_rord_psylsolve_tree; the sequential block-cyclic QR remains for p < 4), the 2x2 Hessenberg repair is a chain
// over the factors, everything else (Q formation, block products, stability test) runs one factor per lane.
#pragma once
#include "psd_real_qr.h"
#include "psd_zord.h"

#define PSD_RORD_SCR 96    // doubles of per-factor swap scratch
#define PSD_RORD_CAP 24    // block transforms per owner and window

struct psd_tq {  // dense orthogonal block transform acting on indices pos..pos+m-1
    int pos, m;
    double q[16];  // column-major, ld 4:  right: row <- row * Q ;  left: column <- Q' * column
};

enum { PSD_ROPH_SCAN = 0, PSD_ROPH_MOVE = 1, PSD_ROPH_DONE = 7 };

struct psd_rostate {
    int n, p, wantZ, W;
    int phase, info;
    int j, jdest, pairskip;          // driver scan (rordschur.jl:77-110)
    int here, nbsrc, splitsrc, jtarget, jsrc0, pend1x1;  // _moveblock! state
    int nswaps, nwindows;
    // in-kernel cycle accounting: 0 step total, 1 window load/store, 2 Sylvester solve, 3 per-factor small algebra,
    // 4 in-window application + recording, 5 total in 100 MHz wall ticks
    long long cyc[6];
};

// Pipelined driver: several selected blocks travel upwards at once, one workgroup (slot) each, a window apart.
// rordschur.jl:77-110 moves one selected block at a time to the top; the blocks it passes are the unselected ones, and the
// block selected next starts below everything moved so far.  So block k + 1 may start while block k is still under way,
// as long as its windows stay below block k: the swaps of the two touch disjoint diagonal windows, their bulk updates
// meet only in off-diagonal blocks where a row operation of one meets a column operation of the other (two passes, as in
// the multishift trains).  Every block makes exactly the swaps of the serial order, in the same order; the target of a
// block is known once its predecessor has arrived, until then its windows end below the predecessor's last known bottom.
#define PSD_RO_SLOTS 64
struct psd_roslot {
    int active, seq;  // seq: number of the block in the order of selection (blocks arrive in that order)
    int here, nbsrc, splitsrc, pend1x1, jsrc0;  // _moveblock! state (as psd_rostate)
    int jtarget, tknown;  // target row, valid once every earlier block has arrived
    int lim;              // top row its window of the coming tick may reach (jtarget, or the row below the predecessor)
    int landed, fail, info;
    int nswaps, nwindows, pad;
    long long cyc[6];
};
struct psd_romb {
    int n, p, wantZ, W;
    int phase, info;
    int j, pairskip, jdest;  // driver scan (rordschur.jl:77-110)
    int scandone, nstarted, nfinished, nactive;
    int failseq, failinfo;   // lowest block number with a rejected swap: later blocks stop, earlier ones finish
    int nticks;
};

struct psd_roparams {
    double* H;
    double* Z;
    psd_rostate* st;
    psd_apply_desc* desc;
    psd_tq* tq;   // [p][PSD_RORD_CAP]
    int* cnt;     // [p]
    const unsigned char* select;
    double* wr;
    double* wi;
    double* xscr;  // [n][p][8] scratch for _rpeigvals2x2
    // GeneralizedPeriodicSchur (signed swaps, sylswap.jl:197-538): signature of the internal right-order factors
    // (nullptr: all true) and the scaled eigenvalue outputs of ordschur.jl:206-314
    const unsigned char* S;
    psd_z* alpha;
    double* beta;
    int* ascale;
    // pipelined driver (psd_rord_plan / psd_rord_step_mb): scheduler state and the slots of the blocks in flight; desc, tq
    // and cnt then hold PSD_RO_SLOTS sets, one per slot
    struct psd_romb* mb;
    struct psd_roslot* slots;
};
PSD_HD bool psd_rosig(const psd_roparams& P, int j) { return P.S == nullptr || P.S[j - 1] != 0; }

PSD_D psd_rparams P_as_r(const psd_roparams& P) {
    psd_rparams R;
    R.H = P.H;
    R.Z = P.Z;
    R.st = nullptr; R.desc = nullptr; R.tr = nullptr; R.cnt = nullptr;
    R.hdiag = R.hsub = R.hsup = R.Pd = R.Pe = R.Pf = R.hnorms = R.wr = R.wi = nullptr;
    R.log = nullptr;
    return R;
}

// ------------------------------------------------------------------------------------------------
// tiny dense helpers (column-major, explicit leading dimension)

// full m x m Q of the Householder QR of the m x nc matrix Xi (ld 4): Q' Xi = [R; 0].  Q is written as a full
// 4 x 4 matrix, identity outside m x m (so that fixed-size 4 x 4 products need no masks).  Fixed trip counts and
// predicates keep everything in registers.
PSD_D void psd_sm_fullq(const double* Xi, int m, int nc, double* Q /*ld 4*/) {
    double S[16], Qr[16];
#pragma unroll
    for (int q = 0; q < 16; ++q) {
        const int c = q >> 2, r = q & 3;
        S[q] = (c < nc && r < m) ? Xi[q] : 0.0;
        Qr[q] = (r == c) ? 1.0 : 0.0;
    }
    const int kmax = (m - 1 < nc) ? (m - 1) : nc;
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        if (k < kmax) {
            double v[4];
            double amax = 0.0;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                v[i] = (i >= k && i < m) ? S[k * 4 + i] : 0.0;
                amax = fmax(amax, fabs(v[i]));
            }
            if (amax != 0.0) {
                int ex;
                (void)frexp(amax, &ex);
                double ssq = 0.0;
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    v[i] = ldexp(v[i], -ex);
                    ssq += v[i] * v[i];
                }
                double nrm, rnrm;
                psd_sqrt_pair_fast(ssq, nrm, rnrm);
                const double alpha = v[k];
                const double beta = -copysign(nrm, alpha);
                const double tau2 = psd_rcp_fast(nrm * (nrm + fabs(alpha)));
                v[k] = alpha - beta;
#pragma unroll
                for (int c = 0; c < 2; ++c) {  // nc <= 2
                    if (c >= k && c < nc) {
                        double d = 0.0;
#pragma unroll
                        for (int i = 0; i < 4; ++i) d += v[i] * S[c * 4 + i];
                        d *= tau2;
#pragma unroll
                        for (int i = 0; i < 4; ++i) S[c * 4 + i] -= d * v[i];
                    }
                }
#pragma unroll
                for (int r = 0; r < 4; ++r) {  // Q <- Q H
                    double d = 0.0;
#pragma unroll
                    for (int i = 0; i < 4; ++i) d += Qr[i * 4 + r] * v[i];
                    d *= tau2;
#pragma unroll
                    for (int i = 0; i < 4; ++i) Qr[i * 4 + r] -= d * v[i];
                }
            }
        }
    }
#pragma unroll
    for (int q = 0; q < 16; ++q) Q[q] = Qr[q];
}

// C (4 x 4, ld 4) = op(A) * op(B) on zero-/identity-padded 4 x 4 operands (C may alias A or B)
PSD_D void psd_sm_mul(const double* A, bool ta, const double* B, bool tb, int m, double* C) {
    double a[16], b[16], t[16];
#pragma unroll
    for (int q = 0; q < 16; ++q) {
        a[q] = A[q];
        b[q] = B[q];
    }
#pragma unroll
    for (int c = 0; c < 4; ++c)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            double sum = 0.0;
#pragma unroll
            for (int k = 0; k < 4; ++k) sum += (ta ? a[r * 4 + k] : a[k * 4 + r]) * (tb ? b[k * 4 + c] : b[c * 4 + k]);
            t[c * 4 + r] = sum;
        }
#pragma unroll
    for (int q = 0; q < 16; ++q) C[q] = t[q];
}

// Householder QR of the first NC columns of the NR x ncols matrix S (LDS, ld 8), applied to all columns, by the whole
// wavefront: the reflector is evaluated redundantly by every lane (broadcast LDS reads), the trailing columns are
// updated one per lane.  NR, NC are compile-time so that every loop unrolls without predicates (a lone wavefront is
// bound by its instruction count).  Returns false (uniformly) if a diagonal entry of R is exactly zero.
template <int NR, int NC>
PSD_D bool psd_sm_qr_par_t(double* S, int ncols) {
    bool ok = true;
#pragma unroll
    for (int k = 0; k < NC; ++k) {
        PSD_WAVE_SYNC();
        double v[NR];
        double amax = 0.0;
#pragma unroll
        for (int i = k; i < NR; ++i) {
            v[i] = S[k * 8 + i];
            amax = fmax(amax, fabs(v[i]));
        }
        if (amax == 0.0) {
            ok = false;
        } else {
            // scale by a power of two (exact), so that the fast rsqrt / reciprocal forms are in range; the reflector
            // I - v v' / (nrm (nrm + |alpha|)) does not depend on the scaling of v
            int ex;
            (void)frexp(amax, &ex);
            double ssq = 0.0;
#pragma unroll
            for (int i = k; i < NR; ++i) {
                v[i] = ldexp(v[i], -ex);
                ssq += v[i] * v[i];
            }
            const double alpha = v[k];
            double nrm, rnrm;
            psd_sqrt_pair_fast(ssq, nrm, rnrm);
            const double beta = -copysign(nrm, alpha);
            const double tau2 = psd_rcp_fast(nrm * (nrm + fabs(alpha)));
            v[k] = alpha - beta;
            const double beta_out = ldexp(beta, ex);
            PSD_WAVE_SYNC();
            PSD_PAR_FOR(t, ncols - k) {
                const int c = k + t;
                if (t == 0) {
                    S[k * 8 + k] = beta_out;
#pragma unroll
                    for (int i = k + 1; i < NR; ++i) S[k * 8 + i] = 0.0;
                } else {
                    double x[NR];
                    double d = 0.0;
#pragma unroll
                    for (int i = k; i < NR; ++i) {
                        x[i] = S[c * 8 + i];
                        d += v[i] * x[i];
                    }
                    d *= tau2;
#pragma unroll
                    for (int i = k; i < NR; ++i) S[c * 8 + i] = x[i] - d * v[i];
                }
            }
        }
    }
    PSD_WAVE_SYNC();
#pragma unroll
    for (int k = 0; k < NC; ++k)
        if (S[k * 8 + k] == 0.0) ok = false;
    return ok;
}
// nr = 2 pp (stacked elimination step) or pp (square solve), nc = pp, pp in {1, 2, 4}
PSD_D bool psd_sm_qr_par(double* S, int nr, int ncols, int pp) {
    if (pp == 4) return (nr == 8) ? psd_sm_qr_par_t<8, 4>(S, ncols) : psd_sm_qr_par_t<4, 4>(S, ncols);
    if (pp == 2) return (nr == 4) ? psd_sm_qr_par_t<4, 2>(S, ncols) : psd_sm_qr_par_t<2, 2>(S, ncols);
    return (nr == 2) ? psd_sm_qr_par_t<2, 1>(S, ncols) : psd_sm_qr_par_t<1, 1>(S, ncols);
}

// Periodic Sylvester system A_k X_k - X_{k+1} B_k = -C_k (k = 1..K cyclic), blocks p1 x p1, p2 x p2,
// p1 x p2 (sylvester.jl:170-193).  Block-cyclic structured QR (the role of babd.jl:17-96): the
// bottom block row is eliminated against the diagonal by Householder QR of stacked 2pp x pp
// blocks, then back substitution.  Per-factor scratch layout (ld 2 blocks): scr[l] + 0: T11, +4: T12,
// +8: T22, +12: X.  wk: K x 52 doubles (D 16, E 16, F 16, rhs 4); ws: 192 doubles of LDS work space
// (S 8 x 13, Lo, Hi, rb, x).  Whole wavefront, uniform control.  Returns false if singular.
// SL[k] (k = 0..K-1): signature of the left-sequence factor k+1 (sylvester.jl:53-87): equation k is
//   SL[k]:  A_k X_k - X_{k+1} B_k = -C_k        !SL[k]:  A_k X_{k+1} - X_k B_k = -C_k
PSD_D bool psd_rord_psylsolve(int K, int p1, int p2, double* scr, double* wk, double* ws, const unsigned char* SL) {
    const int pp = p1 * p2;
    double* S = ws;          // 8 x 13
    double* Lo = ws + 104;   // 16
    double* Hi = ws + 120;   // 16
    double* rb = ws + 136;   // 4
    double* xs = ws + 140;   // 4
    double* xnext = ws + 144;
    double* xlast = ws + 148;
    // kron(I_p2, A)[j*p1+i, j*p1+k] = A[i,k];  kron(B^T, -I_p1)[j*p1+i, k*p1+i] = -B[k,j]
    auto fillA = [&](const double* A, double* M) {  // pp x pp, ld 4
        for (int c = 0; c < pp; ++c)
            for (int r = 0; r < pp; ++r) M[c * 4 + r] = 0.0;
        for (int j = 0; j < p2; ++j)
            for (int i = 0; i < p1; ++i)
                for (int k = 0; k < p1; ++k) M[(j * p1 + k) * 4 + (j * p1 + i)] = A[k * 2 + i];
    };
    auto fillB = [&](const double* B, double* M) {
        for (int c = 0; c < pp; ++c)
            for (int r = 0; r < pp; ++r) M[c * 4 + r] = 0.0;
        for (int j = 0; j < p2; ++j)
            for (int i = 0; i < p1; ++i)
                for (int k = 0; k < p2; ++k) M[(k * p1 + i) * 4 + (j * p1 + i)] = -B[j * 2 + k];
    };
    auto rhsC = [&](const double* C, double* y) {
        for (int j = 0; j < p2; ++j)
            for (int i = 0; i < p1; ++i) y[j * p1 + i] = -C[j * 2 + i];
    };
    PSD_SYNC();
    PSD_PAR_FOR(k, K) {
        double* w = wk + k * 52;
        if (SL[k]) {
            fillA(scr + k * PSD_RORD_SCR + 0, w);        // D_k = kron(I, A_k) multiplies x_k
            fillB(scr + k * PSD_RORD_SCR + 8, w + 16);   // E_k = kron(B_k^T, -I) multiplies x_{k+1}
        } else {
            fillB(scr + k * PSD_RORD_SCR + 8, w);
            fillA(scr + k * PSD_RORD_SCR + 0, w + 16);
        }
        for (int q = 0; q < 16; ++q) w[32 + q] = 0.0;
        rhsC(scr + k * PSD_RORD_SCR + 4, w + 48);
    }
    PSD_SYNC();
    if (K == 1) {
        const double* w = wk;
        PSD_PAR_FOR(t, 16) {
            const int c = t >> 2, r = t & 3;
            if (c < pp && r < pp) S[c * 8 + r] = w[c * 4 + r] + w[16 + c * 4 + r];
            if (c == 0 && r < pp) S[pp * 8 + r] = w[48 + r];
        }
        PSD_SYNC();
        if (!psd_sm_qr_par(S, pp, pp + 1, pp)) return false;
        double y[4];
        for (int k = pp - 1; k >= 0; --k) {
            double sum = S[pp * 8 + k];
            for (int c = k + 1; c < pp; ++c) sum -= S[c * 8 + k] * y[c];
            y[k] = sum / S[k * 8 + k];
        }
        PSD_ONE {
            for (int q = 0; q < pp; ++q) scr[12 + (q / p1) * 2 + (q % p1)] = y[q];
        }
        PSD_SYNC();
        return true;
    }
    // rows k = 0..K-2: D_k x_k + E_k x_{k+1} + F_k x_{K-1} = r_k ; bottom row: Lo x_k' + Hi x_{K-1} = rb
    {
        const double* w = wk + (K - 1) * 52;
        PSD_PAR_FOR(q, 16) {
            Lo[q] = w[16 + q];  // bottom equation: E_{K-1} multiplies x_0
            Hi[q] = w[q];       // D_{K-1} multiplies x_{K-1}
            if (q < 4) rb[q] = w[48 + q];
        }
    }
    PSD_SYNC();
    for (int k = 0; k < K - 1; ++k) {
        double* w = wk + k * 52;
        const bool lastcol = (k + 1 == K - 1);
        // stack: rows 0..pp-1 = row k, rows pp..2pp-1 = bottom; column blocks: [col k | col k+1 | col K-1 | rhs]
        const int nblk = lastcol ? 2 : 3;
        const int ncols = nblk * pp + 1;
        PSD_PAR_FOR(t, 16) {
            const int c = t >> 2, r = t & 3;
            if (c < pp && r < pp) {
                S[c * 8 + r] = w[c * 4 + r];
                S[c * 8 + pp + r] = Lo[c * 4 + r];
                if (lastcol) {
                    S[(pp + c) * 8 + r] = w[16 + c * 4 + r] + w[32 + c * 4 + r];
                    S[(pp + c) * 8 + pp + r] = Hi[c * 4 + r];
                } else {
                    S[(pp + c) * 8 + r] = w[16 + c * 4 + r];
                    S[(pp + c) * 8 + pp + r] = 0.0;
                    S[(2 * pp + c) * 8 + r] = w[32 + c * 4 + r];
                    S[(2 * pp + c) * 8 + pp + r] = Hi[c * 4 + r];
                }
            }
            if (c == 0 && r < pp) {
                S[(nblk * pp) * 8 + r] = w[48 + r];
                S[(nblk * pp) * 8 + pp + r] = rb[r];
            }
        }
        PSD_SYNC();
        if (!psd_sm_qr_par(S, 2 * pp, ncols, pp)) return false;
        PSD_PAR_FOR(t, 16) {
            const int c = t >> 2, r = t & 3;
            if (c < pp && r < pp) {
                w[c * 4 + r] = S[c * 8 + r];
                w[16 + c * 4 + r] = S[(pp + c) * 8 + r];
                if (lastcol) {
                    w[32 + c * 4 + r] = 0.0;
                    Hi[c * 4 + r] = S[(pp + c) * 8 + pp + r];
                } else {
                    Lo[c * 4 + r] = S[(pp + c) * 8 + pp + r];
                    w[32 + c * 4 + r] = S[(2 * pp + c) * 8 + r];
                    Hi[c * 4 + r] = S[(2 * pp + c) * 8 + pp + r];
                }
            }
            if (c == 0 && r < pp) {
                w[48 + r] = S[(nblk * pp) * 8 + r];
                rb[r] = S[(nblk * pp) * 8 + pp + r];
            }
        }
        PSD_SYNC();
    }
    // x_{K-1}: Hi x = rb
    PSD_PAR_FOR(t, 16) {
        const int c = t >> 2, r = t & 3;
        if (c < pp && r < pp) S[c * 8 + r] = Hi[c * 4 + r];
        if (c == 0 && r < pp) S[pp * 8 + r] = rb[r];
    }
    PSD_SYNC();
    if (!psd_sm_qr_par(S, pp, pp + 1, pp)) return false;
    {
        double y[4];
        for (int k = pp - 1; k >= 0; --k) {
            double sum = S[pp * 8 + k];
            for (int c = k + 1; c < pp; ++c) sum -= S[c * 8 + k] * y[c];
            y[k] = sum / S[k * 8 + k];
        }
        PSD_SYNC();
        PSD_ONE {
            for (int q = 0; q < pp; ++q) {
                xs[q] = y[q];
                xlast[q] = y[q];
                xnext[q] = y[q];
                scr[(K - 1) * PSD_RORD_SCR + 12 + (q / p1) * 2 + (q % p1)] = y[q];
            }
        }
        PSD_SYNC();
    }
    bool ok = true;
    for (int k = K - 2; k >= 0; --k) {
        const double* w = wk + k * 52;
        double y[4];
        for (int r = 0; r < pp; ++r) {
            double sum = w[48 + r];
            for (int c = 0; c < pp; ++c) sum -= w[16 + c * 4 + r] * xnext[c];
            if (k + 1 != K - 1)
                for (int c = 0; c < pp; ++c) sum -= w[32 + c * 4 + r] * xlast[c];
            y[r] = sum;
        }
        for (int r = 0; r < pp; ++r)
            if (w[r * 4 + r] == 0.0) ok = false;
        if (!ok) break;
        for (int kk = pp - 1; kk >= 0; --kk) {
            double sum = y[kk];
            for (int c = kk + 1; c < pp; ++c) sum -= w[c * 4 + kk] * y[c];
            y[kk] = sum / w[kk * 4 + kk];
        }
        PSD_SYNC();
        PSD_ONE {
            for (int q = 0; q < pp; ++q) {
                xnext[q] = y[q];
                scr[k * PSD_RORD_SCR + 12 + (q / p1) * 2 + (q % p1)] = y[q];
            }
        }
        PSD_SYNC();
    }
    return ok;
}

// Tree (cyclic-reduction) form of the same solve: the cyclic block-bidiagonal system
//     D_i x_self(i) + E_i x_next(i) = r_i ,   i = 0..m-1
// is reduced level by level: rows (2q, 2q+1) are stacked with the shared unknown first,
//     [ E_a  D_a  0    | r_a ]        columns: x_mid | x_left | x_right | rhs
//     [ D_b  0    E_b  | r_b ]
// a Householder QR of the first block column leaves [R a b | s] (kept for the back substitution) on top and the
// reduced row  c x_left + d x_right = s'  below; an odd last row is carried over.  All pairs of a level are independent,
// four of them are factored per pass (16 lanes per pair, one column per lane), so the chain is log2(K) levels deep
// instead of K-1 steps.  Orthogonal transformations only, as the reference's structured QR (babd.jl:17-96).
// tw: LDS work area of psd_rord_tree_doubles(K) doubles.
// (rows over all levels: m + ceil(m/2) + ... <= 2 K + log2 K)
PSD_HD int psd_rord_tree_doubles(int K) { return (2 * K + 8) * 36 + 4 * 104 + 4 * 12 + (7 * K + 16 + 32 + 1) / 2 + 8; }

template <int PP>
PSD_D bool psd_rord_tree_level_qr(double* SR, double* vb, int nact) {
    // nact (<= 4) stacked matrices of 2 PP x (3 PP + 1), ld 8, at SR + g * 104; QR of the first PP columns of each
    constexpr int NR = 2 * PP, NCOL = 3 * PP + 1;
    bool ok = true;
#pragma unroll
    for (int k = 0; k < PP; ++k) {
        PSD_WAVE_SYNC();
        PSD_PAR_FOR(g, nact) {  // reflector of column k of matrix g
            const double* col = SR + g * 104 + k * 8;
            double v[NR];
            double amax = 0.0;
#pragma unroll
            for (int i = k; i < NR; ++i) {
                v[i] = col[i];
                amax = fmax(amax, fabs(v[i]));
            }
            double* o = vb + g * 12;
            if (amax == 0.0) {
                o[10] = 0.0;  // singular
                o[8] = 0.0;   // tau2 = 0: no update
                o[9] = 0.0;
            } else {
                int ex;
                (void)frexp(amax, &ex);
                double ssq = 0.0;
#pragma unroll
                for (int i = k; i < NR; ++i) {
                    v[i] = ldexp(v[i], -ex);
                    ssq += v[i] * v[i];
                }
                const double alpha = v[k];
                double nrm, rnrm;
                psd_sqrt_pair_fast(ssq, nrm, rnrm);
                const double beta = -copysign(nrm, alpha);
                v[k] = alpha - beta;
#pragma unroll
                for (int i = k; i < NR; ++i) o[i] = v[i];
                o[8] = psd_rcp_fast(nrm * (nrm + fabs(alpha)));
                o[9] = ldexp(beta, ex);
                o[10] = 1.0;
            }
        }
        PSD_WAVE_SYNC();
        PSD_PAR_FOR(t, 64) {
            const int g = t >> 4, c = t & 15;
            if (g < nact && c >= k && c < NCOL) {
                const double* o = vb + g * 12;
                double* col = SR + g * 104 + c * 8;
                if (o[10] != 0.0) {
                    if (c == k) {
                        col[k] = o[9];
#pragma unroll
                        for (int i = k + 1; i < NR; ++i) col[i] = 0.0;
                    } else {
                        double x[NR];
                        double d = 0.0;
#pragma unroll
                        for (int i = k; i < NR; ++i) {
                            x[i] = col[i];
                            d += o[i] * x[i];
                        }
                        d *= o[8];
#pragma unroll
                        for (int i = k; i < NR; ++i) col[i] = x[i] - d * o[i];
                    }
                }
            }
        }
        PSD_WAVE_SYNC();
        for (int g = 0; g < nact; ++g)
            if (vb[g * 12 + 10] == 0.0) ok = false;
    }
    for (int g = 0; g < nact; ++g)
        for (int k = 0; k < PP; ++k)
            if (SR[g * 104 + k * 8 + k] == 0.0) ok = false;
    return ok;
}
template <int PP>
PSD_D bool psd_rord_psylsolve_tree_t(int K, int p1, int p2, double* scr, double* wk, double* tw, const unsigned char* SL) {
    constexpr int pp = PP;
    double* RW = tw;                      // [2K+8][36]: D 16 | E 16 | r 4
    double* SR = RW + (2 * K + 8) * 36;   // [4][104]
    double* vb = SR + 4 * 104;        // [4][12]
    int* selfi = (int*)(vb + 4 * 12);  // [2K+8]
    int* nexti = selfi + 2 * K + 8;    // [2K+8]
    int* meta = nexti + 2 * K + 8;     // [K][3]: mid, left, right of every pair
    int* lvl = meta + 3 * K;           // [32]: per level (npairs, first pair slot)
    auto fillA = [&](const double* A, double* M) {
        for (int c = 0; c < 4; ++c)
            for (int r = 0; r < 4; ++r) M[c * 4 + r] = 0.0;
        for (int j = 0; j < p2; ++j)
            for (int i = 0; i < p1; ++i)
                for (int k = 0; k < p1; ++k) M[(j * p1 + k) * 4 + (j * p1 + i)] = A[k * 2 + i];
    };
    auto fillB = [&](const double* B, double* M) {
        for (int c = 0; c < 4; ++c)
            for (int r = 0; r < 4; ++r) M[c * 4 + r] = 0.0;
        for (int j = 0; j < p2; ++j)
            for (int i = 0; i < p1; ++i)
                for (int k = 0; k < p2; ++k) M[(k * p1 + i) * 4 + (j * p1 + i)] = -B[j * 2 + k];
    };
    PSD_WAVE_SYNC();
    PSD_PAR_FOR(k, K) {  // level 0: equation k couples x_k (D) and x_{k+1} (E), sylvester.jl:53-87
        double* w = RW + k * 36;
        if (SL[k]) {
            fillA(scr + k * PSD_RORD_SCR + 0, w);
            fillB(scr + k * PSD_RORD_SCR + 8, w + 16);
        } else {
            fillB(scr + k * PSD_RORD_SCR + 8, w);
            fillA(scr + k * PSD_RORD_SCR + 0, w + 16);
        }
        const double* C = scr + k * PSD_RORD_SCR + 4;
        for (int q = 0; q < 4; ++q) w[32 + q] = 0.0;
        for (int j = 0; j < p2; ++j)
            for (int i = 0; i < p1; ++i) w[32 + j * p1 + i] = -C[j * 2 + i];
        selfi[k] = k;
        nexti[k] = (k + 1 == K) ? 0 : (k + 1);
    }
    PSD_WAVE_SYNC();
    int m = K, rowbase = 0, pairslot = 0, nlev = 0;
    bool ok = true;
    while (m > 1) {
        const int npairs = m / 2;
        const int newbase = rowbase + m;
        for (int b0 = 0; b0 < npairs; b0 += 4) {
            const int nact = (npairs - b0 < 4) ? (npairs - b0) : 4;
            PSD_WAVE_SYNC();
            PSD_PAR_FOR(t, 64) {  // stack: lane = (pair g, column c)
                const int g = t >> 4, c = t & 15;
                if (g < nact && c <= 3 * pp) {
                    const int q = b0 + g;
                    const double* ra = RW + (rowbase + 2 * q) * 36;
                    const double* rb = RW + (rowbase + 2 * q + 1) * 36;
                    const bool merged = selfi[rowbase + 2 * q] == nexti[rowbase + 2 * q + 1];  // m == 2
                    double* col = SR + g * 104 + c * 8;
                    const int blk = (c == 3 * pp) ? 3 : (c / pp), cc = (c == 3 * pp) ? 0 : (c % pp);
                    for (int i = 0; i < pp; ++i) {
                        double top, bot;
                        if (blk == 0) {         // x_mid: E_a over D_b
                            top = ra[16 + cc * 4 + i];
                            bot = rb[cc * 4 + i];
                        } else if (blk == 1) {  // x_left: D_a over 0 (or E_b if left == right)
                            top = ra[cc * 4 + i];
                            bot = merged ? rb[16 + cc * 4 + i] : 0.0;
                        } else if (blk == 2) {  // x_right: 0 over E_b
                            top = 0.0;
                            bot = merged ? 0.0 : rb[16 + cc * 4 + i];
                        } else {
                            top = ra[32 + i];
                            bot = rb[32 + i];
                        }
                        col[i] = top;
                        col[pp + i] = bot;
                    }
                }
            }
            if (!psd_rord_tree_level_qr<PP>(SR, vb, nact)) ok = false;
            PSD_PAR_FOR(t, 64) {  // unstack
                const int g = t >> 4, c = t & 15;
                if (g < nact && c <= 3 * pp) {
                    const int q = b0 + g;
                    const double* col = SR + g * 104 + c * 8;
                    double* top = wk + (pairslot + q) * 52;  // R 16 | a 16 | b 16 | s 4
                    double* nr = RW + (newbase + q) * 36;
                    const int blk = (c == 3 * pp) ? 3 : (c / pp), cc = (c == 3 * pp) ? 0 : (c % pp);
                    for (int i = 0; i < pp; ++i) {
                        if (blk < 3) top[blk * 16 + cc * 4 + i] = col[i];
                        else top[48 + i] = col[i];
                        if (blk == 1) nr[cc * 4 + i] = col[pp + i];
                        else if (blk == 2) nr[16 + cc * 4 + i] = col[pp + i];
                        else if (blk == 3) nr[32 + i] = col[pp + i];
                    }
                    if (c == 0) {
                        const int ia = rowbase + 2 * q, ib = ia + 1;
                        meta[3 * (pairslot + q) + 0] = nexti[ia];
                        meta[3 * (pairslot + q) + 1] = selfi[ia];
                        meta[3 * (pairslot + q) + 2] = nexti[ib];
                        selfi[newbase + q] = selfi[ia];
                        nexti[newbase + q] = nexti[ib];
                    }
                }
            }
            PSD_WAVE_SYNC();
        }
        if (m & 1) {  // odd: the last row is carried over
            PSD_PAR_FOR(t, 36) { RW[(newbase + npairs) * 36 + t] = RW[(rowbase + m - 1) * 36 + t]; }
            PSD_ONE {
                selfi[newbase + npairs] = selfi[rowbase + m - 1];
                nexti[newbase + npairs] = nexti[rowbase + m - 1];
            }
            PSD_WAVE_SYNC();
        }
        PSD_ONE {
            lvl[2 * nlev] = npairs;
            lvl[2 * nlev + 1] = pairslot;
        }
        nlev += 1;
        pairslot += npairs;
        rowbase = newbase;
        m = (m + 1) / 2;
    }
    if (!ok) return false;
    // one row left: (D + E) x = r
    {
        const double* w = RW + rowbase * 36;
        PSD_WAVE_SYNC();
        PSD_PAR_FOR(t, 16) {
            const int c = t >> 2, r = t & 3;
            if (c < pp && r < pp) SR[c * 8 + r] = w[c * 4 + r] + w[16 + c * 4 + r];
            if (c == 0 && r < pp) SR[pp * 8 + r] = w[32 + r];
        }
        PSD_WAVE_SYNC();
        if (!psd_sm_qr_par(SR, pp, pp + 1, pp)) return false;
        double y[4];
        for (int k = pp - 1; k >= 0; --k) {
            double sum = SR[pp * 8 + k];
            for (int c = k + 1; c < pp; ++c) sum -= SR[c * 8 + k] * y[c];
            y[k] = sum / SR[k * 8 + k];
        }
        const int u = selfi[rowbase];
        PSD_WAVE_SYNC();
        PSD_ONE {
            for (int q = 0; q < pp; ++q) scr[u * PSD_RORD_SCR + 12 + (q / p1) * 2 + (q % p1)] = y[q];
        }
        PSD_WAVE_SYNC();
    }
    // back substitution, level by level:  R x_mid = s - a x_left - b x_right
    for (int L = nlev - 1; L >= 0; --L) {
        const int npairs = lvl[2 * L], ps = lvl[2 * L + 1];
        PSD_PAR_FOR(q, npairs) {
            const double* top = wk + (ps + q) * 52;
            const int um = meta[3 * (ps + q) + 0], ul = meta[3 * (ps + q) + 1], ur = meta[3 * (ps + q) + 2];
            double xl[4], xr[4], y[4];
            for (int e = 0; e < 4; ++e) {
                xl[e] = (e < pp) ? scr[ul * PSD_RORD_SCR + 12 + (e / p1) * 2 + (e % p1)] : 0.0;
                xr[e] = (e < pp) ? scr[ur * PSD_RORD_SCR + 12 + (e / p1) * 2 + (e % p1)] : 0.0;
            }
            for (int r = 0; r < pp; ++r) {
                double sum = top[48 + r];
                for (int c = 0; c < pp; ++c) sum -= top[16 + c * 4 + r] * xl[c] + top[32 + c * 4 + r] * xr[c];
                y[r] = sum;
            }
            for (int k = pp - 1; k >= 0; --k) {
                double sum = y[k];
                for (int c = k + 1; c < pp; ++c) sum -= top[c * 4 + k] * y[c];
                y[k] = sum / top[k * 4 + k];
            }
            for (int e = 0; e < pp; ++e) scr[um * PSD_RORD_SCR + 12 + (e / p1) * 2 + (e % p1)] = y[e];
        }
        PSD_WAVE_SYNC();
    }
    return true;
}

PSD_D bool psd_rord_psylsolve_tree(int K, int p1, int p2, double* scr, double* wk, double* tw, const unsigned char* SL) {
    const int pp = p1 * p2;
    if (pp == 4) return psd_rord_psylsolve_tree_t<4>(K, p1, p2, scr, wk, tw, SL);
    if (pp == 2) return psd_rord_psylsolve_tree_t<2>(K, p1, p2, scr, wk, tw, SL);
    return psd_rord_psylsolve_tree_t<1>(K, p1, p2, scr, wk, tw, SL);
}

// The small dense part of one swap of adjacent blocks (p1, p2) — sylswap.jl:14-129 (and :542-617 via the same
// machinery for p1 = p2 = 1).  Per-factor scratch (index = position l-1 in the reference's left-oriented
// sequence X_l): +0 T11, +4 T12, +8 T22, +12 X (ld 2); +16 Q, +32 Txx, +48 Ws, +64 Qfin (ld 4); +80 orig block
// (ld 4).  Whole wavefront: the cyclic solve and the 2x2 Hessenberg repair are chains over the factors, everything
// else runs one factor per lane.  Returns (uniformly) 0 ok, 1 rejected (strong test), 2 singular.
PSD_D int psd_rord_swap_scalar(int K, int p1, int p2, double* scr, double* wk, double* ws, double tnrm, long long* cyc,
                                const unsigned char* SL, bool gen) {
    const int m = p1 + p2;
    const long long tq0 = psd_clock();
    const bool solved = (K >= 4) ? psd_rord_psylsolve_tree(K, p1, p2, scr, wk, ws + 192, SL)
                                 : psd_rord_psylsolve(K, p1, p2, scr, wk, ws, SL);
    cyc[2] += psd_clock() - tq0;
    if (!solved) return 2;
    const double thresh = fmax(PSD_DBL_MIN, 100.0 * PSD_DBL_EPS * tnrm);
    PSD_SYNC();
    PSD_PAR_FOR(l, K) {
        double* s = scr + l * PSD_RORD_SCR;
        const int lp = (l == 0) ? (K - 1) : (l - 1);
        if (SL[lp]) {  // Q_l from the QR of [X; I] (sylswap.jl:58-66, 243-256)
            double Xi[16];  // [X; I] (m x p2), ld 4
#pragma unroll
            for (int q = 0; q < 16; ++q) {
                const int c = q >> 2, r = q & 3;
                double x = 0.0;
                if (c < p2) {
                    if (r < p1) x = s[12 + (c & 1) * 2 + (r & 1)];
                    else if (r == p1 + c) x = 1.0;
                }
                Xi[q] = x;
            }
            psd_sm_fullq(Xi, m, p2, s + 16);
        } else {
            // Q_l = q' from the RQ of [I -X] = [0 R] q (sylswap.jl:258-268): q = J Qb' J with Qb from the QR of
            // B = J [I -X]' J (m x p1), so Q_l = J Qb J
            double Bm[16];
#pragma unroll
            for (int q = 0; q < 16; ++q) {
                const int c = q >> 2, r = q & 3;  // B(r, c) = Xi(p1-1-c, m-1-r),  Xi(a, b) = (b < p1) ? (a == b) : -X(a, b-p1)
                double x = 0.0;
                if (c < p1 && r < m) {
                    const int a = p1 - 1 - c, b = m - 1 - r;
                    if (b < p1) x = (a == b) ? 1.0 : 0.0;
                    else x = -s[12 + ((b - p1) & 1) * 2 + (a & 1)];
                }
                Bm[q] = x;
            }
            psd_sm_fullq(Bm, m, p1, s + 64);  // (the Qfin slot is free until the end of the swap)
            for (int q = 0; q < 16; ++q) {
                const int c = q >> 2, r = q & 3;
                s[16 + q] = (r < m && c < m) ? s[64 + (m - 1 - c) * 4 + (m - 1 - r)] : ((r == c) ? 1.0 : 0.0);
            }
        }
        for (int q = 0; q < 16; ++q) s[32 + q] = s[80 + q];  // Txx <- original [T11 T12; 0 T22]
        for (int c = 0; c < 4; ++c)
            for (int r = 0; r < 4; ++r) s[48 + c * 4 + r] = (r == c) ? 1.0 : 0.0;
        // X_l:  SL[l] ? Txx[l] Q_l : Q_l' Txx[l]
        if (SL[l]) psd_sm_mul(s + 32, false, s + 16, false, m, s + 32);
        else psd_sm_mul(s + 16, true, s + 32, false, m, s + 32);
    }
    PSD_SYNC();
    PSD_PAR_FOR(l, K) {  // X_{l-1}:  SL[l-1] ? Q_l' Txx[l-1] : Txx[l-1] Q_l
        double* s = scr + l * PSD_RORD_SCR;
        const int lp = (l == 0) ? (K - 1) : (l - 1);
        double* sp = scr + lp * PSD_RORD_SCR;
        if (SL[lp]) psd_sm_mul(s + 16, true, sp + 32, false, m, sp + 32);
        else psd_sm_mul(sp + 32, false, s + 16, false, m, sp + 32);
    }
    PSD_SYNC();
    bool weak_ok = true;
    if (gen) {  // weak test of the signed swap (sylswap.jl:303-313)
        double wsmax = 0.0;
        for (int l = 0; l < K; ++l) {
            double ssq = 0.0;
            for (int a = p2; a < m; ++a)
                for (int b = 0; b < p2; ++b) {
                    const double x = scr[l * PSD_RORD_SCR + 32 + b * 4 + a];
                    ssq += x * x;
                }
            wsmax = fmax(wsmax, sqrt(ssq));
        }
        weak_ok = !(wsmax > thresh);
    }
    bool fill1 = false, fill2 = false;
    if (p2 > 1)
        for (int l = 0; l < K; ++l) fill1 |= fabs(scr[l * PSD_RORD_SCR + 32 + 0 * 4 + 1]) > thresh;
    if (p1 > 1)
        for (int l = 0; l < K; ++l) fill2 |= fabs(scr[l * PSD_RORD_SCR + 32 + p2 * 4 + p2 + 1]) > thresh;
    const bool fillin = fill1 || fill2;
    for (int pass = 0; pass < 2; ++pass) {  // sylswap.jl:159-191 _filled2hess! at j0 = 0 and j0 = p2
        if (pass == 0 && !fill1) continue;
        if (pass == 1 && !fill2) continue;
        const int j0 = (pass == 0) ? 0 : p2, j1 = j0 + 1;
        // rpschur2x2.jl:326-359 on the 2x2 copies: Qs[lp] = hr' for l = 2..K
        // (the solver's work array is free again: Th at wk + l*52, Hq at wk + l*52 + 4)
        PSD_SYNC();
        PSD_PAR_FOR(l, K) {
            const double* t = scr + l * PSD_RORD_SCR + 32;
            double* Th = wk + l * 52;
            double* Hq = Th + 4;
            Th[0] = t[j0 * 4 + j0];
            Th[1] = t[j0 * 4 + j1];
            Th[2] = t[j1 * 4 + j0];
            Th[3] = t[j1 * 4 + j1];  // [a(0,0) a(1,0) a(0,1) a(1,1)] column-major 2x2
            Hq[0] = 1.0; Hq[1] = 0.0; Hq[2] = 0.0; Hq[3] = 1.0;
        }
        PSD_SYNC();
        PSD_ONE {  // rpschur2x2.jl:326-359 with the signature
            for (int l = 2; l <= K; ++l) {
                double* Al = wk + (l - 1) * 52;  // [a(0,0) a(1,0) a(0,1) a(1,1)]
                const int lp = (l % K) + 1;
                double* Ap = wk + (lp - 1) * 52;
                double v1, v2, tau;
                if (SL[l - 1]) {
                    double xv[2] = {Al[0], Al[1]};
                    tau = psd_reflector_small(xv, 2);
                    Al[0] = xv[0];
                    Al[1] = 0.0;
                    v1 = 1.0;
                    v2 = xv[1];
                    const double sdot = v1 * Al[2] + v2 * Al[3];  // lmul!(hr', Al[:, 2])
                    Al[2] -= sdot * tau * v1;
                    Al[3] -= sdot * tau * v2;
                } else {
                    double xv[2] = {Al[3], Al[1]};  // (a(1,1), a(1,0))
                    tau = psd_reflector_small(xv, 2);
                    Al[1] = 0.0;
                    Al[3] = xv[0];
                    v1 = xv[1];
                    v2 = 1.0;
                    const double sdot = Al[0] * v1 + Al[2] * v2;  // rmul!(Al[1:1, :], hr)
                    Al[0] -= sdot * tau * v1;
                    Al[2] -= sdot * tau * v2;
                }
                {  // Qs[lp] <- hr' Qs[lp]
                    double* Q = wk + (lp - 1) * 52 + 4;
                    for (int c = 0; c < 2; ++c) {
                        const double sdot = v1 * Q[c * 2 + 0] + v2 * Q[c * 2 + 1];
                        Q[c * 2 + 0] -= sdot * tau * v1;
                        Q[c * 2 + 1] -= sdot * tau * v2;
                    }
                }
                if (SL[lp - 1]) {
                    for (int r = 0; r < 2; ++r) {  // rmul!(Ap, hr)
                        const double sdot = Ap[0 * 2 + r] * v1 + Ap[1 * 2 + r] * v2;
                        Ap[0 * 2 + r] -= sdot * tau * v1;
                        Ap[1 * 2 + r] -= sdot * tau * v2;
                    }
                } else {
                    for (int c = 0; c < 2; ++c) {  // lmul!(hr', Ap)
                        const double sdot = v1 * Ap[c * 2 + 0] + v2 * Ap[c * 2 + 1];
                        Ap[c * 2 + 0] -= sdot * tau * v1;
                        Ap[c * 2 + 1] -= sdot * tau * v2;
                    }
                }
            }
        }
        PSD_SYNC();
        PSD_PAR_FOR(t, K) {  // sylswap.jl:166-190
            const int l = t + 1;
            const int lp = (l % K) + 1;
            const double* q = wk + (l - 1) * 52 + 4;
            const double* qp = wk + (lp - 1) * 52 + 4;
            double* Tl = scr + (l - 1) * PSD_RORD_SCR + 32;
            double* W = scr + (l - 1) * PSD_RORD_SCR + 48;
            const double* qc = SL[l - 1] ? q : qp;  // acts on the columns j0:j1
            const double* qr = SL[l - 1] ? qp : q;  // its transpose acts on the rows j0:j1
            if (SL[l - 1]) {
                for (int r = 0; r < m; ++r) {
                    const double a = Tl[j0 * 4 + r], b = Tl[j1 * 4 + r];
                    Tl[j0 * 4 + r] = a * qc[0] + b * qc[1];
                    Tl[j1 * 4 + r] = a * qc[2] + b * qc[3];
                }
                for (int c = 0; c < m; ++c) {
                    const double a = Tl[c * 4 + j0], b = Tl[c * 4 + j1];
                    Tl[c * 4 + j0] = qr[0] * a + qr[1] * b;
                    Tl[c * 4 + j1] = qr[2] * a + qr[3] * b;
                }
            } else {
                for (int c = 0; c < m; ++c) {
                    const double a = Tl[c * 4 + j0], b = Tl[c * 4 + j1];
                    Tl[c * 4 + j0] = qr[0] * a + qr[1] * b;
                    Tl[c * 4 + j1] = qr[2] * a + qr[3] * b;
                }
                for (int r = 0; r < m; ++r) {
                    const double a = Tl[j0 * 4 + r], b = Tl[j1 * 4 + r];
                    Tl[j0 * 4 + r] = a * qc[0] + b * qc[1];
                    Tl[j1 * 4 + r] = a * qc[2] + b * qc[3];
                }
            }
            for (int r = 0; r < m; ++r) {
                const double a = W[j0 * 4 + r], b = W[j1 * 4 + r];
                W[j0 * 4 + r] = a * q[0] + b * q[1];
                W[j1 * 4 + r] = a * q[2] + b * q[3];
            }
        }
        PSD_SYNC();
    }
    // final transform of factor l: Qfin = q_l W_l ; strong test: Qfin_{l+1} Txx[l] Qfin_l' ~ original block
    PSD_SYNC();
    PSD_PAR_FOR(l, K) {
        double* s = scr + l * PSD_RORD_SCR;
        if (fillin) psd_sm_mul(s + 16, false, s + 48, false, m, s + 64);
        else
            for (int q = 0; q < 16; ++q) s[64 + q] = s[16 + q];
    }
    PSD_SYNC();
    PSD_PAR_FOR(l, K) {
        const int l1 = (l + 1) % K;
        double* Tt = wk + l * 52 + 16;  // (the E block of the solver is free again)
        if (SL[l]) {  // Qfin_{l+1} Txx[l] Qfin_l'
            psd_sm_mul(scr + l1 * PSD_RORD_SCR + 64, false, scr + l * PSD_RORD_SCR + 32, false, m, Tt);
            psd_sm_mul(Tt, false, scr + l * PSD_RORD_SCR + 64, true, m, Tt);
        } else {  // Qfin_l Txx[l] Qfin_{l+1}'  (sylswap.jl:355-366)
            psd_sm_mul(scr + l * PSD_RORD_SCR + 64, false, scr + l * PSD_RORD_SCR + 32, false, m, Tt);
            psd_sm_mul(Tt, false, scr + l1 * PSD_RORD_SCR + 64, true, m, Tt);
        }
        double d = 0.0;
        for (int c = 0; c < m; ++c)
            for (int r = 0; r < m; ++r) {
                const double e = Tt[c * 4 + r] - scr[l * PSD_RORD_SCR + 80 + c * 4 + r];
                d += e * e;
            }
        wk[l * 52 + 12] = (sqrt(d) > thresh) ? 1.0 : 0.0;
    }
    PSD_SYNC();
    bool ok = weak_ok;
    for (int l = 0; l < K; ++l)
        if (wk[l * 52 + 12] != 0.0) ok = false;
    PSD_SYNC();
    return ok ? 0 : 1;
}

// ------------------------------------------------------------------------------------------------
// in-window application of the block transform Q (m x m, ld 4) of sequence index l to X_l = T_{sg} and
// X_{l-1} = T_{own}.  Unsigned (and S true): right on the columns i1.. of T_sg, left (Q') on the rows i1.. of T_own;
// a negative signature of a factor flips its side (sylswap.jl:396-445).
// lane t < nA works on T_sg, the others on T_own; `right`: row r = bs + k gets  row <- row Q  on the columns i1..;
// otherwise column c = i1 + k gets  column <- Q' column  on the rows i1..
PSD_D void psd_rord_win_apply_one(const psd_win& w, int fac, bool right, int i1, int m, const double* q, int k) {
    double a[4], b[4];
    if (right) {
        const int r = w.bs + k;
#pragma unroll
        for (int e = 0; e < 4; ++e) a[e] = (e < m) ? w.at(fac, r, i1 + e) : 0.0;
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            double sum = 0.0;
#pragma unroll
            for (int e = 0; e < 4; ++e) sum += a[e] * q[c * 4 + e];
            b[c] = sum;
        }
#pragma unroll
        for (int e = 0; e < 4; ++e)
            if (e < m) w.at(fac, r, i1 + e) = b[e];
    } else {
        const int c = i1 + k;
#pragma unroll
        for (int e = 0; e < 4; ++e) a[e] = (e < m) ? w.at(fac, i1 + e, c) : 0.0;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            double sum = 0.0;
#pragma unroll
            for (int e = 0; e < 4; ++e) sum += q[r * 4 + e] * a[e];
            b[r] = sum;
        }
#pragma unroll
        for (int e = 0; e < 4; ++e)
            if (e < m) w.at(fac, i1 + e, c) = b[e];
    }
}
PSD_D void psd_rord_win_apply(const psd_win& w, int sg, int own, int i1, int m, const double* Q, bool ssg, bool sown) {
    const int nrow = (i1 + m - 1) - w.bs + 1;  // rows bs..i1+m-1
    const int ncol = w.be - i1 + 1;            // columns i1..be
    const bool rightA = ssg, rightB = !sown;   // T_sg: columns if S[sg]; T_own: rows if S[own]
    const int nA = rightA ? nrow : ncol, nB = rightB ? nrow : ncol;
    double q[16];  // identity-padded 4 x 4
#pragma unroll
    for (int e = 0; e < 16; ++e) q[e] = Q[e];
    if (sg != own) {
        PSD_PAR_FOR(t, nA + nB) {
            if (t < nA) psd_rord_win_apply_one(w, sg, rightA, i1, m, q, t);
            else psd_rord_win_apply_one(w, own, rightB, i1, m, q, t - nA);
        }
    } else {
        // p == 1: both sides act on the same factor and overlap in the diagonal block -> two ordered passes
        PSD_PAR_FOR(t, nA) { psd_rord_win_apply_one(w, sg, rightA, i1, m, q, t); }
        PSD_SYNC();
        PSD_PAR_FOR(t, nB) { psd_rord_win_apply_one(w, own, rightB, i1, m, q, t); }
    }
    PSD_SYNC();
}

// one swap inside the window; returns 0 ok / 1 rejected / 2 singular
PSD_D int psd_rord_swap(const psd_roparams& P, const psd_rostate& st, const psd_win& w, double* scr, double* wk,
                        double* ws, double* flagbuf, int* lcnt, int i1, int p1, int p2, long long* cyc,
                        const unsigned char* SL) {
    const int p = st.p, m = p1 + p2;
    PSD_SYNC();
    PSD_PAR_FOR(t, p) {
        const int l = t + 1, sg = psd_ord_sigma(p, l);
        double* s = scr + t * PSD_RORD_SCR;
        for (int c = 0; c < 4; ++c)
            for (int r = 0; r < 4; ++r) s[80 + c * 4 + r] = (r < m && c < m) ? w.at(sg, i1 + r, i1 + c) : 0.0;
        for (int b = 0; b < 2; ++b)
            for (int a = 0; a < 2; ++a) {
                s[0 + b * 2 + a] = (a < p1 && b < p1) ? s[80 + b * 4 + a] : 0.0;
                s[4 + b * 2 + a] = (a < p1 && b < p2) ? s[80 + (p1 + b) * 4 + a] : 0.0;
                s[8 + b * 2 + a] = (a < p2 && b < p2) ? s[80 + (p1 + b) * 4 + p1 + a] : 0.0;
            }
        // the reference's working copy is [T11 T12; 0 T22]: the lower-left block is not carried
        for (int c = 0; c < p1; ++c)
            for (int r = p1; r < m; ++r) s[80 + c * 4 + r] = 0.0;
    }
    PSD_SYNC();
    PSD_PAR_FOR(t, p) {  // Frobenius norm of the window blocks (sylswap.jl:41), one factor per lane
        const int sg = psd_ord_sigma(p, t + 1);
        double ssq = 0.0;
        for (int c = 0; c < m; ++c)
            for (int r = 0; r < m; ++r) {
                const double x = w.at(sg, i1 + r, i1 + c);
                ssq += x * x;
            }
        wk[t * 52 + 12] = ssq;
    }
    PSD_SYNC();
    double tn = 0.0;
    for (int t = 0; t < p; ++t) tn += wk[t * 52 + 12];
    tn = sqrt(tn);
    PSD_SYNC();
    const long long ts0 = psd_clock();
    const int flag0 = psd_rord_swap_scalar(p, p1, p2, scr, wk, ws, tn, cyc, SL, P.S != nullptr);
    cyc[3] += psd_clock() - ts0;
    const long long ta0 = psd_clock();
    PSD_ONE { flagbuf[0] = (double)flag0; }
    PSD_SYNC();
    const int flag = (int)flagbuf[0];
    if (flag) return flag;
    for (int l = 1; l <= p; ++l) {
        const int own = psd_ord_owner(p, l), sg = psd_ord_sigma(p, l);
        const double* Q = scr + (l - 1) * PSD_RORD_SCR + 64;
        psd_rord_win_apply(w, sg, own, i1, m, Q, psd_rosig(P, sg), psd_rosig(P, own));
        {
            const int q = lcnt[own - 1];
            PSD_SYNC();
            if (q < PSD_RORD_CAP) {
                psd_tq* tr = P.tq + ((size_t)(own - 1) * PSD_RORD_CAP + q);
                PSD_PAR_FOR(e, 16) { tr->q[e] = Q[e]; }
                PSD_ONE {
                    tr->pos = i1;
                    tr->m = m;
                }
            }
            PSD_ONE { lcnt[own - 1] = q + 1; }
        }
    }
    PSD_SYNC();
    // sweep up the dust (sylswap.jl:150-154): T_1 keeps only the new block structure, the others are triangular
    PSD_PAR_FOR(t, p) {
        const int sg = psd_ord_sigma(p, t + 1);
        if (sg == 1) {
            for (int r = i1 + p2; r <= i1 + m - 1; ++r)
                for (int c = i1; c <= i1 + p2 - 1; ++c) w.at(1, r, c) = 0.0;
        } else {
            for (int c = i1; c <= i1 + m - 1; ++c)
                for (int r = c + 1; r <= i1 + m - 1; ++r) w.at(sg, r, c) = 0.0;
        }
    }
    PSD_SYNC();
    cyc[4] += psd_clock() - ta0;
    return 0;
}

// One window of _moveblock! (rordschur.jl:181-247 restricted to the blocks that fit the window): the travelling block(s)
// at st.here swap upwards until the window's top or st.jtarget; window store, transform counts and the descriptor of the
// bulk update.  st.phase becomes PSD_ROPH_SCAN when the block has arrived, PSD_ROPH_DONE (with st.info) on a rejected swap.
PSD_D void psd_rord_move(const psd_roparams& P, psd_rostate& st, double* ldsd, double* scr, double* wk, double* flagbuf,
                         double* ws, int* lcnt, const unsigned char* SL, long long* cyc) {
    const int n = st.n, p = st.p;
    // window: bottom = end of the travelling block(s); top as high as the LDS window allows
    const int width = st.splitsrc ? 2 : st.nbsrc;
    psd_win w;
    w.b = ldsd;
    w.W = st.W;
    w.ld = st.W + 1;
    w.bsz = st.W * (st.W + 1);
    w.be = st.here + width - 1;
    w.bs = (w.be - st.W + 1 > st.jtarget) ? (w.be - st.W + 1) : st.jtarget;
    PSD_PAR_FOR(m, p) { lcnt[m] = 0; }
    const long long tl0 = psd_clock();
    psd_win_load(P_as_r(P), w, n, p);
    cyc[1] += psd_clock() - tl0;
    int fail = 0;
    int here = st.here;
    const int top0 = here;
    // rordschur.jl:181-247 restricted to the blocks that fit the window
    while (here > st.jtarget && !fail) {
        if (st.pend1x1) {  // second 1x1 of a split pair follows its partner (rordschur.jl:207-215)
            fail = psd_rord_swap(P, st, w, scr, wk, ws, flagbuf, lcnt, here, 1, 1, cyc, SL);
            if (fail) break;
            st.nswaps += 1;
            st.pend1x1 = 0;
            here -= 1;
            continue;
        }
        int nbnext = 1;
        if (here >= 3 && here - 2 >= w.bs) {
            if (w.at(1, here - 1, here - 2) != 0) nbnext = 2;
        } else if (here >= 3 && here - 1 != st.jtarget) {
            break;  // the next block may start above the window (row jtarget itself is a block start)
        }
        if (here - nbnext < w.bs) break;
        if (!st.splitsrc) {
            fail = psd_rord_swap(P, st, w, scr, wk, ws, flagbuf, lcnt, here - nbnext, nbnext, st.nbsrc, cyc, SL);
            if (fail) break;
            st.nswaps += 1;
            here -= nbnext;
            if (st.nbsrc == 2 && w.at(1, here + 1, here) == 0) st.splitsrc = 1;
        } else {
            fail = psd_rord_swap(P, st, w, scr, wk, ws, flagbuf, lcnt, here - nbnext, nbnext, 1, cyc, SL);
            if (fail) break;
            st.nswaps += 1;
            if (nbnext == 1) {
                st.pend1x1 = 1;  // handled at the top of the loop (position `here`)
            } else {
                if (w.at(1, here, here - 1) == 0) nbnext = 1;
                if (nbnext == 2) {
                    fail = psd_rord_swap(P, st, w, scr, wk, ws, flagbuf, lcnt, here - 1, 2, 1, cyc, SL);
                    if (fail) break;
                    st.nswaps += 1;
                    here -= 2;
                } else {
                    fail = psd_rord_swap(P, st, w, scr, wk, ws, flagbuf, lcnt, here, 1, 1, cyc, SL);
                    if (fail) break;
                    st.nswaps += 1;
                    fail = psd_rord_swap(P, st, w, scr, wk, ws, flagbuf, lcnt, here - 1, 1, 1, cyc, SL);
                    if (fail) break;
                    st.nswaps += 1;
                    here -= 2;
                }
            }
        }
    }
    if (fail) {
        st.info = (fail == 2) ? PSD_INFO_SINGULAR : (PSD_INFO_ILLCOND_BASE + st.jsrc0);
        st.phase = PSD_ROPH_DONE;
    } else {
        const long long tl1 = psd_clock();
        psd_win_store(P_as_r(P), w, n, p);
        cyc[1] += psd_clock() - tl1;
        PSD_SYNC();
        PSD_PAR_FOR(m, p) { P.cnt[m] = lcnt[m]; }
        PSD_ONE {
            psd_apply_desc d;
            d.active = 1;
            d.plo = w.bs;
            d.phi = w.be;
            d.lc0 = w.be + 1;
            d.lc1 = n;
            d.rr0 = 1;
            d.rr1 = w.bs - 1;
            d.zr0 = 1;
            d.zr1 = st.wantZ ? n : 0;
            *P.desc = d;
        }
        st.nwindows += 1;
        st.here = here;
        (void)top0;
        if (here <= st.jtarget && !st.pend1x1) {
            st.jdest = here;  // _moveblock! returns jdest = here
            if (st.nbsrc == 2) st.jdest += 1;
            st.phase = PSD_ROPH_SCAN;
        }
    }
}

PSD_KERNEL_B(PSD_STEP_NT) psd_rord_step(psd_roparams P) {
    PSD_LDS_DECL;
    psd_rostate st = *P.st;
    PSD_ONE { P.desc->active = 0; }
    if (st.phase == PSD_ROPH_DONE) return;
    const int n = st.n, p = st.p;
    long long cyc[6] = {0, 0, 0, 0, 0, 0};
    const long long tk0 = psd_clock(), tw0 = psd_wallclock();
    const psd_mat<double> A1 = psd_mat<double>{P.H, n};
    double* ldsd = (double*)psd_lds;
    const size_t winb = (size_t)p * st.W * (st.W + 1);
    double* scr = ldsd + winb;
    double* wk = scr + (size_t)p * PSD_RORD_SCR;
    double* flagbuf = wk + (size_t)p * 52;
    double* ws = flagbuf + 4;
    int* lcnt = (int*)(ws + 192 + psd_rord_tree_doubles(p));
    unsigned char* SL = (unsigned char*)(lcnt + p);  // signature of the left-oriented sequence X_l = T_{sigma(l)}
    PSD_PAR_FOR(t, p) { SL[t] = psd_rosig(P, psd_ord_sigma(p, t + 1)) ? 1 : 0; }
    PSD_SYNC();
    // driver scan: rordschur.jl:77-110
    while (st.phase == PSD_ROPH_SCAN) {
        st.j += 1;
        if (st.j > n) {
            st.phase = PSD_ROPH_DONE;
            break;
        }
        if (st.pairskip) {
            st.pairskip = 0;
            continue;
        }
        const int j = st.j;
        bool swap = P.select[j - 1] != 0;
        bool pair = false;
        if (j < n && A1(j + 1, j) != 0) {
            pair = true;
            swap = swap || (P.select[j] != 0);
        }
        st.pairskip = pair ? 1 : 0;
        if (swap) {
            st.jdest += 1;
            if (j != st.jdest) {
                // _moveblock! prologue (rordschur.jl:149-172); jsrc already points at a block start
                int jd = st.jdest;
                if (jd > 1 && A1(jd, jd - 1) != 0) jd -= 1;
                st.nbsrc = pair ? 2 : 1;
                st.jsrc0 = j;
                if (jd < j) {
                    st.here = j;
                    st.jtarget = jd;
                    st.splitsrc = 0;
                    st.pend1x1 = 0;
                    st.phase = PSD_ROPH_MOVE;
                } else {
                    st.jdest = jd;
                    if (pair) st.jdest += 1;
                }
            } else if (pair) {
                st.jdest += 1;
            }
        }
    }
    if (st.phase == PSD_ROPH_MOVE) psd_rord_move(P, st, ldsd, scr, wk, flagbuf, ws, lcnt, SL, cyc);
    cyc[0] = psd_clock() - tk0;
    cyc[5] = psd_wallclock() - tw0;
    for (int q = 0; q < 6; ++q) st.cyc[q] += cyc[q];
    PSD_SYNC();
    PSD_ONE { *P.st = st; }
}

PSD_KERNEL psd_rord_init(psd_roparams P, int n, int p, int wantZ, int W) {
    PSD_ONE {
        psd_rostate st;
        st.n = n; st.p = p; st.wantZ = wantZ; st.W = W;
        st.phase = PSD_ROPH_SCAN; st.info = 0;
        st.j = 0; st.jdest = 0; st.pairskip = 0;
        st.here = 0; st.nbsrc = 1; st.splitsrc = 0; st.jtarget = 0; st.jsrc0 = 0; st.pend1x1 = 0;
        st.nswaps = 0; st.nwindows = 0;
        for (int q = 0; q < 6; ++q) st.cyc[q] = 0;
        *P.st = st;
        P.desc->active = 0;
    }
}

// Bulk application of the window's block transforms: grid = (tiles, p owners, 3 roles) as psd_rq_apply.
PSD_KERNEL_B(PSD_APPLY_NT) psd_rord_apply(psd_roparams P, int n, int p) {
    PSD_LDS_DECL;
    const psd_apply_desc d = *P.desc;
    if (!d.active) return;
    const int m = PSD_BLOCK_Y + 1;
    const int role = PSD_BLOCK_Z;
    const int cnt = P.cnt[m - 1] < PSD_RORD_CAP ? P.cnt[m - 1] : PSD_RORD_CAP;
    if (cnt <= 0) return;
    const int T = PSD_APPLY_NT;
    const int S = d.phi - d.plo + 1;
    psd_tq* ltr = (psd_tq*)psd_lds;
    double* tile = (double*)(psd_lds + sizeof(psd_tq) * PSD_RORD_CAP);
    PSD_PAR_FOR(e, cnt) { ltr[e] = P.tq[(size_t)(m - 1) * PSD_RORD_CAP + e]; }
    // owner m acts on T_m from the left if S[m] (else from the right), on T_{m-1} from the right if S[m-1] (else
    // from the left), on Z_m from the right
    const int mm1 = (m == 1) ? p : (m - 1);
    const int fac = (role == 0) ? m : ((role == 1) ? mm1 : m);
    const bool left = (role == 0) ? psd_rosig(P, m) : ((role == 1) ? !psd_rosig(P, mm1) : false);
    if (left) {
        const int c0 = d.lc0 + PSD_BLOCK_X * T;
        if (c0 > d.lc1) return;
        const int nc = (d.lc1 - c0 + 1 < T) ? (d.lc1 - c0 + 1) : T;
        const psd_mat<double> M = psd_mat<double>{P.H + (size_t)(fac - 1) * n * n, n};
        const int ldt = T + 1;
        PSD_PAR_FOR(t, 32 * nc) {  // (S <= 32; no index divisions)
            const int r = t & 31, c = t >> 5;
            if (r >= S) continue;
            tile[r * ldt + c] = M(d.plo + r, c0 + c);
        }
        PSD_SYNC();
        PSD_PAR_FOR(c, nc) {
            for (int e = 0; e < cnt; ++e) {
                const psd_tq& tr = ltr[e];
                const int r = tr.pos - d.plo, mm = tr.m;
                double a[4], b[4];
                for (int q = 0; q < mm; ++q) a[q] = tile[(r + q) * ldt + c];
                for (int rr = 0; rr < mm; ++rr) {
                    double s = 0.0;
                    for (int q = 0; q < mm; ++q) s += tr.q[rr * 4 + q] * a[q];
                    b[rr] = s;
                }
                for (int q = 0; q < mm; ++q) tile[(r + q) * ldt + c] = b[q];
            }
        }
        PSD_SYNC();
        PSD_PAR_FOR(t, 32 * nc) {  // (S <= 32; no index divisions)
            const int r = t & 31, c = t >> 5;
            if (r >= S) continue;
            M(d.plo + r, c0 + c) = tile[r * ldt + c];
        }
    } else {
        const int lo = (role == 2) ? d.zr0 : d.rr0;
        const int hi = (role == 2) ? d.zr1 : d.rr1;
        const int r0 = lo + PSD_BLOCK_X * T;
        if (r0 > hi) return;
        const int nr = (hi - r0 + 1 < T) ? (hi - r0 + 1) : T;
        double* base = (role == 2) ? P.Z : P.H;
        const psd_mat<double> M = psd_mat<double>{base + (size_t)(fac - 1) * n * n, n};
        PSD_PAR_FOR(t, S * T) {
            const int r = t & (T - 1), c = t / T;
            if (r >= nr) continue;
            tile[c * T + r] = M(r0 + r, d.plo + c);
        }
        PSD_SYNC();
        PSD_PAR_FOR(r, nr) {
            for (int e = 0; e < cnt; ++e) {
                const psd_tq& tr = ltr[e];
                const int c = tr.pos - d.plo, mm = tr.m;
                double a[4], b[4];
                for (int q = 0; q < mm; ++q) a[q] = tile[(c + q) * T + r];
                for (int cc = 0; cc < mm; ++cc) {
                    double s = 0.0;
                    for (int q = 0; q < mm; ++q) s += a[q] * tr.q[cc * 4 + q];
                    b[cc] = s;
                }
                for (int q = 0; q < mm; ++q) tile[(c + q) * T + r] = b[q];
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


// ------------------------------------------------------------------------------------------------
// pipelined driver (see psd_roslot)

PSD_D int psd_ro_find(const psd_roslot* S, int seq) {
    for (int q = 0; q < PSD_RO_SLOTS; ++q)
        if (S[q].active && S[q].seq == seq) return q;
    return -1;
}

// One lane plans the coming tick from the state the previous tick left: arrivals in order, the head block's target, the
// window limits of the followers, at most one new block, termination.  Nothing here races with the step launch: it runs
// between two of them on the same stream.
PSD_KERNEL psd_rord_plan(psd_roparams P) {
    PSD_ONE {
        psd_romb G = *P.mb;
        psd_roslot* S = P.slots;
        if (G.phase != PSD_ROPH_DONE) {
            const int n = G.n;
            const psd_mat<double> A1 = psd_mat<double>{P.H, n};
            for (int q = 0; q < PSD_RO_SLOTS; ++q)
                if (S[q].active && S[q].fail && S[q].seq < G.failseq) {
                    G.failseq = S[q].seq;
                    G.failinfo = S[q].info;
                }
            for (;;) {  // arrivals, in the order of selection (_moveblock! returns jdest = here, rordschur.jl:247)
                const int q = psd_ro_find(S, G.nfinished);
                if (q < 0 || !S[q].landed || S[q].fail) break;
                G.jdest = S[q].here + ((S[q].nbsrc == 2) ? 1 : 0);
                S[q].active = 0;
                G.nfinished += 1;
                G.nactive -= 1;
            }
            {   // the head block learns its target (rordschur.jl:100-104, 149-160)
                const int q = psd_ro_find(S, G.nfinished);
                if (q >= 0 && !S[q].tknown) {
                    G.jdest += 1;
                    int jd = G.jdest;
                    if (jd > 1 && A1(jd, jd - 1) != 0) jd -= 1;
                    S[q].jtarget = jd;
                    S[q].tknown = 1;
                }
            }
            for (int q = 0; q < PSD_RO_SLOTS; ++q) {
                if (!S[q].active) continue;
                if (S[q].tknown) {
                    S[q].lim = S[q].jtarget;
                } else {
                    const int r = psd_ro_find(S, S[q].seq - 1);  // (in flight: blocks arrive in order)
                    S[q].lim = (r >= 0) ? (S[r].here + (S[r].splitsrc ? 2 : S[r].nbsrc)) : S[q].here;
                }
            }
            if (G.failseq == 0x7fffffff && !G.scandone) {
                int f = -1;
                for (int q = 0; q < PSD_RO_SLOTS && f < 0; ++q)
                    if (!S[q].active) f = q;
                while (f >= 0) {  // driver scan: rordschur.jl:77-110
                    const int j = G.j + 1;
                    if (j > n) {
                        G.scandone = 1;
                        break;
                    }
                    if (G.pairskip) {
                        G.pairskip = 0;
                        G.j = j;
                        continue;
                    }
                    bool swap = P.select[j - 1] != 0;
                    bool pair = false;
                    if (j < n && A1(j + 1, j) != 0) {
                        pair = true;
                        swap = swap || (P.select[j] != 0);
                    }
                    if (!swap) {
                        G.j = j;
                        G.pairskip = pair ? 1 : 0;
                        continue;
                    }
                    psd_roslot ns;
                    ns.active = 1;
                    ns.here = j;
                    ns.nbsrc = pair ? 2 : 1;
                    ns.splitsrc = 0;
                    ns.pend1x1 = 0;
                    ns.jsrc0 = j;
                    ns.landed = ns.fail = ns.info = 0;
                    ns.nswaps = S[f].nswaps;  // (the counters of a slot run on over the blocks it carries)
                    ns.nwindows = S[f].nwindows;
                    ns.pad = 0;
                    for (int q = 0; q < 6; ++q) ns.cyc[q] = S[f].cyc[q];
                    if (G.nactive == 0) {  // nothing under way: the serial driver's step
                        G.j = j;
                        G.pairskip = pair ? 1 : 0;
                        G.jdest += 1;
                        if (j != G.jdest) {
                            int jd = G.jdest;
                            if (jd > 1 && A1(jd, jd - 1) != 0) jd -= 1;
                            if (jd < j) {
                                ns.seq = G.nstarted;
                                ns.jtarget = jd;
                                ns.tknown = 1;
                                ns.lim = jd;
                                S[f] = ns;
                                G.nstarted += 1;
                                G.nactive += 1;
                                break;
                            }
                            G.jdest = jd;
                            if (pair) G.jdest += 1;
                        } else if (pair) {
                            G.jdest += 1;
                        }
                        continue;
                    }
                    // blocks under way: this one starts behind the last of them once two rows lie between
                    const int r = psd_ro_find(S, G.nstarted - 1);
                    const int tb = (r >= 0) ? (S[r].here + (S[r].splitsrc ? 2 : S[r].nbsrc) - 1) : n;
                    if (j - 2 < tb + 1) break;  // (row j is looked at again in the next tick)
                    G.j = j;
                    G.pairskip = pair ? 1 : 0;
                    ns.seq = G.nstarted;
                    ns.jtarget = 0;
                    ns.tknown = 0;
                    ns.lim = tb + 1;
                    S[f] = ns;
                    G.nstarted += 1;
                    G.nactive += 1;
                    break;
                }
            }
            if (G.failseq != 0x7fffffff) {
                bool any = false;
                for (int q = 0; q < PSD_RO_SLOTS; ++q)
                    if (S[q].active && !S[q].fail && !S[q].landed && S[q].seq < G.failseq) any = true;
                if (!any) {
                    G.info = G.failinfo;
                    G.phase = PSD_ROPH_DONE;
                }
            } else if (G.scandone && G.nactive == 0) {
                G.phase = PSD_ROPH_DONE;
            }
            G.nticks += 1;
        }
        *P.mb = G;
    }
}

// grid = PSD_RO_SLOTS workgroups: the block of slot s moves up by one window
PSD_KERNEL_B(PSD_STEP_NT) psd_rord_step_mb(psd_roparams P) {
    PSD_LDS_DECL;
    const int s = PSD_BLOCK_X;
    const psd_romb G = *P.mb;
    const int n = G.n, p = G.p;
    psd_roparams Q = P;
    Q.tq = P.tq + (size_t)s * p * PSD_RORD_CAP;
    Q.cnt = P.cnt + (size_t)s * p;
    Q.desc = P.desc + s;
    PSD_ONE { Q.desc->active = 0; }
    if (G.phase == PSD_ROPH_DONE) return;
    psd_roslot sl = P.slots[s];
    if (!sl.active || sl.fail || sl.landed || sl.seq >= G.failseq) return;
    const int jt = sl.tknown ? sl.jtarget : sl.lim;
    if (sl.here <= jt && !sl.pend1x1) {
        if (sl.tknown) {
            PSD_SYNC();
            PSD_ONE { P.slots[s].landed = 1; }
        }
        return;
    }
    long long cyc[6] = {0, 0, 0, 0, 0, 0};
    const long long tk0 = psd_clock(), tw0 = psd_wallclock();
    double* ldsd = (double*)psd_lds;
    const size_t winb = (size_t)p * G.W * (G.W + 1);
    double* scr = ldsd + winb;
    double* wk = scr + (size_t)p * PSD_RORD_SCR;
    double* flagbuf = wk + (size_t)p * 52;
    double* ws = flagbuf + 4;
    int* lcnt = (int*)(ws + 192 + psd_rord_tree_doubles(p));
    unsigned char* SL = (unsigned char*)(lcnt + p);
    PSD_PAR_FOR(t, p) { SL[t] = psd_rosig(P, psd_ord_sigma(p, t + 1)) ? 1 : 0; }
    PSD_SYNC();
    psd_rostate st;
    st.n = n; st.p = p; st.wantZ = G.wantZ; st.W = G.W;
    st.phase = PSD_ROPH_MOVE; st.info = 0;
    st.j = 0; st.jdest = 0; st.pairskip = 0;
    st.here = sl.here; st.nbsrc = sl.nbsrc; st.splitsrc = sl.splitsrc; st.jtarget = jt; st.jsrc0 = sl.jsrc0;
    st.pend1x1 = sl.pend1x1;
    st.nswaps = sl.nswaps; st.nwindows = sl.nwindows;
    psd_rord_move(Q, st, ldsd, scr, wk, flagbuf, ws, lcnt, SL, cyc);
    cyc[0] = psd_clock() - tk0;
    cyc[5] = psd_wallclock() - tw0;
    PSD_SYNC();
    PSD_ONE {
        sl.here = st.here; sl.splitsrc = st.splitsrc; sl.pend1x1 = st.pend1x1;
        sl.nswaps = st.nswaps; sl.nwindows = st.nwindows;
        for (int q = 0; q < 6; ++q) sl.cyc[q] += cyc[q];
        if (st.phase == PSD_ROPH_DONE) {
            sl.fail = 1;
            sl.info = st.info;
        } else if (st.phase == PSD_ROPH_SCAN && sl.tknown) {
            sl.landed = 1;
        }
        P.slots[s] = sl;
    }
}

PSD_KERNEL psd_rord_init_mb(psd_roparams P, int n, int p, int wantZ, int W) {
    PSD_PAR_FOR(q, PSD_RO_SLOTS) {
        psd_roslot z;
        z.active = 0; z.seq = -1;
        z.here = z.nbsrc = z.splitsrc = z.pend1x1 = z.jsrc0 = 0;
        z.jtarget = z.tknown = z.lim = 0;
        z.landed = z.fail = z.info = 0;
        z.nswaps = z.nwindows = z.pad = 0;
        for (int e = 0; e < 6; ++e) z.cyc[e] = 0;
        P.slots[q] = z;
        P.desc[q].active = 0;
    }
    PSD_ONE {
        psd_romb G;
        G.n = n; G.p = p; G.wantZ = wantZ; G.W = W;
        G.phase = PSD_ROPH_SCAN; G.info = 0;
        G.j = 0; G.pairskip = 0; G.jdest = 0;
        G.scandone = 0; G.nstarted = 0; G.nfinished = 0; G.nactive = 0;
        G.failseq = 0x7fffffff; G.failinfo = 0;
        G.nticks = 0;
        *P.mb = G;
    }
}

// Bulk application for the pipelined driver: grid = (tiles, p owners, 3 roles x PSD_RO_SLOTS).  pass 0: the operations
// from the left (rows of a factor) and the Schur vectors, pass 1: the operations from the right on the factors — a row
// operation of one window and a column operation of another meet in off-diagonal blocks, so the two kinds are two launches.
PSD_KERNEL_B(PSD_APPLY_NT) psd_rord_apply_mb(psd_roparams P, int n, int p, int pass) {
    PSD_LDS_DECL;
    const int slot = PSD_BLOCK_Z / 3;
    const int role = PSD_BLOCK_Z - 3 * slot;
    const psd_apply_desc d = P.desc[slot];
    if (!d.active) return;
    const int m = PSD_BLOCK_Y + 1;
    const int mm1 = (m == 1) ? p : (m - 1);
    const int fac = (role == 0) ? m : ((role == 1) ? mm1 : m);
    const bool left = (role == 0) ? psd_rosig(P, m) : ((role == 1) ? !psd_rosig(P, mm1) : false);
    const bool inpass0 = left || role == 2;
    if ((pass == 0) != inpass0) return;
    const int* pcnt = P.cnt + (size_t)slot * p;
    const int cnt = pcnt[m - 1] < PSD_RORD_CAP ? pcnt[m - 1] : PSD_RORD_CAP;
    if (cnt <= 0) return;
    const psd_tq* gtq = P.tq + ((size_t)slot * p + (m - 1)) * PSD_RORD_CAP;
    const int T = PSD_APPLY_NT;
    const int S = d.phi - d.plo + 1;
    psd_tq* ltr = (psd_tq*)psd_lds;
    double* tile = (double*)(psd_lds + sizeof(psd_tq) * PSD_RORD_CAP);
    if (left) {
        const int c0 = d.lc0 + PSD_BLOCK_X * T;
        if (c0 > d.lc1) return;
        PSD_PAR_FOR(e, cnt) { ltr[e] = gtq[e]; }
        const int nc = (d.lc1 - c0 + 1 < T) ? (d.lc1 - c0 + 1) : T;
        const psd_mat<double> M = psd_mat<double>{P.H + (size_t)(fac - 1) * n * n, n};
        const int ldt = T + 1;
        PSD_PAR_FOR(t, 32 * nc) {
            const int r = t & 31, c = t >> 5;
            if (r >= S) continue;
            tile[r * ldt + c] = M(d.plo + r, c0 + c);
        }
        PSD_SYNC();
        PSD_PAR_FOR(c, nc) {
            for (int e = 0; e < cnt; ++e) {
                const psd_tq& tr = ltr[e];
                const int r = tr.pos - d.plo, mm = tr.m;
                double a[4], b[4];
                for (int q = 0; q < mm; ++q) a[q] = tile[(r + q) * ldt + c];
                for (int rr = 0; rr < mm; ++rr) {
                    double sacc = 0.0;
                    for (int q = 0; q < mm; ++q) sacc += tr.q[rr * 4 + q] * a[q];
                    b[rr] = sacc;
                }
                for (int q = 0; q < mm; ++q) tile[(r + q) * ldt + c] = b[q];
            }
        }
        PSD_SYNC();
        PSD_PAR_FOR(t, 32 * nc) {
            const int r = t & 31, c = t >> 5;
            if (r >= S) continue;
            M(d.plo + r, c0 + c) = tile[r * ldt + c];
        }
    } else {
        const int lo = (role == 2) ? d.zr0 : d.rr0;
        const int hi = (role == 2) ? d.zr1 : d.rr1;
        const int r0 = lo + PSD_BLOCK_X * T;
        if (r0 > hi) return;
        PSD_PAR_FOR(e, cnt) { ltr[e] = gtq[e]; }
        const int nr = (hi - r0 + 1 < T) ? (hi - r0 + 1) : T;
        double* base = (role == 2) ? P.Z : P.H;
        const psd_mat<double> M = psd_mat<double>{base + (size_t)(fac - 1) * n * n, n};
        PSD_PAR_FOR(t, S * T) {
            const int r = t & (T - 1), c = t / T;
            if (r >= nr) continue;
            tile[c * T + r] = M(r0 + r, d.plo + c);
        }
        PSD_SYNC();
        PSD_PAR_FOR(r, nr) {
            for (int e = 0; e < cnt; ++e) {
                const psd_tq& tr = ltr[e];
                const int c = tr.pos - d.plo, mm = tr.m;
                double a[4], b[4];
                for (int q = 0; q < mm; ++q) a[q] = tile[(c + q) * T + r];
                for (int cc = 0; cc < mm; ++cc) {
                    double sacc = 0.0;
                    for (int q = 0; q < mm; ++q) sacc += a[q] * tr.q[cc * 4 + q];
                    b[cc] = sacc;
                }
                for (int q = 0; q < mm; ++q) tile[(c + q) * T + r] = b[q];
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


// rpschur2x2.jl:9-275 (_rpeigvals2x2, S all true, schurindex 1) for the conjugate pair at rows j, j+1 of the
// internal right-order sequence T_1 ... T_p; X: [p][8] complex 2x2 scratch of this thread.
PSD_D void psd_rord_eigpair(const psd_roparams& P, int n, int p, int j, psd_z* X, double& l1r, double& l1i,
                            double& l2r, double& l2i) {
    for (int l = 0; l < p; ++l) {
        const psd_mat<double> M = psd_mat<double>{P.H + (size_t)l * n * n, n};
        X[4 * l + 0] = zmk(M(j, j), 0.0);
        X[4 * l + 1] = zmk(M(j, j + 1), 0.0);
        X[4 * l + 2] = zmk(M(j + 1, j), 0.0);
        X[4 * l + 3] = zmk(M(j + 1, j + 1), 0.0);  // [a b; c d] = X[0], X[1]; X[2], X[3]
    }
    const int k = p;
    for (int iter = 1; iter <= 80; ++iter) {
        const double lhs = zabs(X[2]);
        double rhs = fmax(zabs(X[0]), zabs(X[3]));
        if (rhs == 0) rhs = zabs(X[1]);
        if (lhs <= PSD_DBL_EPS * rhs) break;
        double c;
        psd_z s, r;
        if (iter == 1) {
            psd_zgivens(zmk(1.0, -2.0), zmk(2.0, 2.0), c, s, r);
        } else if (iter % 40 == 0) {
            psd_zgivens(zmk((double)k, 1.0), zmk(1.0, -2.0), c, s, r);
        } else {
            c = 1.0;
            s = zmk(0.0, 0.0);
            double ct;
            psd_z st;
            psd_zgivens(zmk(1.0, 0.0), zmk(1.0, 0.0), ct, st, r);
            for (int l = k; l >= 2; --l) {
                const psd_z* Xl = X + 4 * (l - 1);
                psd_z Z[3][3];
                for (int a = 0; a < 3; ++a)
                    for (int b = 0; b < 3; ++b) Z[a][b] = zmk(0.0, 0.0);
                Z[0][0] = Xl[0]; Z[1][1] = Xl[0]; Z[1][2] = Xl[1]; Z[2][1] = Xl[2]; Z[2][2] = Xl[3];
                for (int q = 0; q < 3; ++q) psd_zrot_right_adj(ct, st, Z[q][0], Z[q][2]);
                for (int q = 0; q < 3; ++q) psd_zrot_right_adj(c, s, Z[q][0], Z[q][1]);
                psd_zgivens(Z[0][0], Z[2][0], ct, st, r);
                psd_zgivens(Xl[0], Z[1][0], c, s, r);
            }
            psd_z Z[2][3];
            Z[0][0] = X[0]; Z[0][1] = zneg(X[2]); Z[0][2] = zneg(X[3]);
            Z[1][0] = X[2]; Z[1][1] = zmk(0.0, 0.0); Z[1][2] = zmk(0.0, 0.0);
            for (int q = 0; q < 2; ++q) psd_zrot_right_adj(ct, st, Z[q][0], Z[q][2]);
            for (int q = 0; q < 2; ++q) psd_zrot_right_adj(c, s, Z[q][0], Z[q][1]);
            psd_zgivens(Z[0][0], Z[1][0], c, s, r);
        }
        const double ct0 = c;
        const psd_z st0 = s;
        for (int l = k; l >= 2; --l) {
            psd_z* Y = X + 4 * (l - 1);
            psd_zrot_right_adj(c, s, Y[0], Y[1]);
            psd_zrot_right_adj(c, s, Y[2], Y[3]);
            psd_zgivens(Y[0], Y[2], c, s, r);
            Y[0] = r;
            Y[2] = zmk(0.0, 0.0);
            psd_zrot_left(c, s, Y[1], Y[3]);
        }
        psd_zrot_left(ct0, st0, X[0], X[2]);
        psd_zrot_left(ct0, st0, X[1], X[3]);
        psd_zrot_right_adj(c, s, X[0], X[1]);
        psd_zrot_right_adj(c, s, X[2], X[3]);
    }
    psd_z alpha[2];
    double scal[2];
    for (int jj = 0; jj < 2; ++jj) {
        psd_z aj = zmk(1.0, 0.0);
        scal[jj] = 0.0;
        for (int l = 1; l <= k; ++l) {
            psd_z z = X[4 * (l - 1) + (jj == 0 ? 0 : 3)];
            double rhs = zabs(z);
            if (rhs != 0) {
                const int sl = (int)floor(log2(rhs));
                z = zscal(exp2(-(double)sl), z);
                scal[jj] += sl;
            }
            aj = zmul(aj, z);
            if ((l % 10 == 0) || (l == k)) {
                rhs = zabs(aj);
                if (rhs == 0) {
                    scal[jj] = 0;
                } else {
                    const int sl = (int)floor(log2(rhs));
                    aj = zscal(exp2(-(double)sl), aj);
                    scal[jj] += sl;
                }
            }
        }
        alpha[jj] = aj;
    }
    if (alpha[1].im > 0) {
        const psd_z ta = alpha[0];
        alpha[0] = alpha[1];
        alpha[1] = ta;
        const double ts = scal[0];
        scal[0] = scal[1];
        scal[1] = ts;
    }
    if (alpha[0].im != 0 || alpha[1].im != 0) {  // rpschur2x2.jl:238-275 _sanitize_reigpair!
        const double sl = scal[0] - scal[1];
        psd_z zt1, zt2;
        double cst;
        if (sl >= 0) {
            zt1 = zscal(exp2(-sl), alpha[1]);
            zt2 = zsub(alpha[0], zconj(zt1));
            cst = alpha[0].im;
        } else {
            zt1 = zscal(exp2(sl), alpha[0]);
            zt2 = zsub(alpha[1], zconj(zt1));
            cst = alpha[1].im;
        }
        const double misr = hypot(cst, zt1.im);
        const double misc = zabs(zt2) / 2;
        if (misr > misc) {
            const int jx = (scal[0] >= scal[1]) ? 0 : 1;
            const psd_z at = zscal(0.5, zadd(alpha[jx], zconj(zt1)));
            alpha[0] = zmk(at.re, fabs(at.im));
            alpha[1] = zconj(alpha[0]);
        } else {
            alpha[0].im = 0.0;
            alpha[1].im = 0.0;
        }
    }
    l1r = alpha[0].re * exp2(scal[0]);
    l1i = alpha[0].im * exp2(scal[0]);
    l2r = alpha[1].re * exp2(scal[1]);
    l2i = alpha[1].im * exp2(scal[1]);
}

// ordschur.jl:122-204 _updateλ! (real): one thread per position j
PSD_KERNEL psd_rord_values(psd_roparams P, int n, int p) {
    const int NT = PSD_NTHREADS;
    const psd_mat<double> A1 = psd_mat<double>{P.H, n};
    PSD_PAR_FOR(t, NT) {
        const int j = 1 + PSD_BLOCK_X * NT + t;
        if (j <= n) {
            const bool second = (j > 1) && (A1(j, j - 1) != 0);
            const bool first = (j < n) && (A1(j + 1, j) != 0);
            if (first && !second) {
                double a, b, c, d;
                psd_rord_eigpair(P, n, p, j, (psd_z*)(P.xscr + (size_t)(j - 1) * p * 8), a, b, c, d);
                P.wr[j - 1] = a; P.wi[j - 1] = b;
                P.wr[j] = c; P.wi[j] = d;
            } else if (!second) {
                double v = A1(j, j);
                int sc = 0;
                for (int l = 2; l <= p; ++l) {
                    v *= P.H[(size_t)(l - 1) * n * n + (size_t)(j - 1) * n + (j - 1)];
                    if (v != 0) {
                        int e;
                        v = frexp(v, &e);
                        sc += e;
                    }
                }
                P.wr[j - 1] = ldexp(v, sc);
                P.wi[j - 1] = 0.0;
            }
        }
    }
}

// rordschur.jl:117-130: zero everything below the (1x1 / 2x2) diagonal blocks of T_1.  grid over columns.
PSD_KERNEL psd_rord_cleanup(psd_roparams P, int n) {
    const int c = PSD_BLOCK_X + 1;
    const psd_mat<double> A1 = psd_mat<double>{P.H, n};
    // the pair's first eigenvalue has positive imaginary part (rpschur2x2.jl:263-266), the second negative
    const int j0 = (P.wi[c - 1] > 0.0) ? (c + 2) : (c + 1);
    PSD_PAR_FOR(t, n) {
        const int r = j0 + t;
        if (r <= n) A1(r, c) = 0.0;
    }
}
