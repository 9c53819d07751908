// Blocked materialisation of the orthogonal factors Q_j of the periodic Hessenberg-triangular reduction (real).
//
// Replaces Matrix(H.Q) / Matrix(QR.Q) — /root/reference/src/PeriodicSchurDecompositions.jl:136-143,180-197 (LAPACK's
// dorghr / dorgqr behind them) — for the packed reflectors psd_hess2.h / psd_hess.h leave in HBM.
//
// Q_j = H_{j,1} H_{j,2} ... H_{j,n-1} accumulated backwards, KB = 32 reflectors at a time in compact-WY form
// (H_ib ... H_{ib+KB-1} = I - V T V', T upper triangular: dlarft 'F','C'):
//   psd_fq_tfactor   every T of every factor in ONE launch (they depend on V and tau only): Gram matrix V'V through
//                    LDS, then the 32-step triangular recurrence;
//   psd_fq_apply     one launch per block step, grid = (column tiles of 64, factors):  Y = V' Q[rows, tile],
//                    W = T Y, Q[rows, tile] -= V W on the matrix cores (v_mfma_f64_16x16x4_f64).  A workgroup owns
//                    its 64 columns for all rows, so there is no exchange between workgroups; the tile is read twice
//                    and written once per step: 24 m^2 bytes per step and factor against the 16 m^2 of ONE of the 32
//                    rank-one updates it replaces (round 1's psd_formq_step moved 16 m^2 per reflector).
// The rank-32 update is formed transposed (W' V'), so that the MFMA result fragment has 16 consecutive ROWS of Q per
// 16-lane group: 128-byte segments on the read-modify-write of Q.
#pragma once
#include "psd_platform.h"

#define PSD_FQ_KB 32  // reflectors per block
#define PSD_FQ_TN 64  // columns of Q per workgroup
#define PSD_FQ_NT 256

struct psd_fq_args {
    const double* Hp;   // packed reflectors (LAPACK style), p blocks of n x n
    const double* tau;  // p x n
    double* Q;          // p blocks of n x n (identity on entry of the first step)
    double* T;          // nblk x (factors of the launch) blocks of KB x KB, column-major
    int n;
    int j0;    // first factor of the slice this context forms (0-based)
    int nblk;  // blocks of KB reflectors
};

// element (r, i) of factor j's V (0-based row r, 0-based reflector i; j 1-based): unit at row r0 = i + (j == 1)
PSD_HD double psd_fq_v(const double* Hj, int n, int j, int r, int i) {
    if (r >= n || i >= n - 1) return 0.0;
    const int r0 = i + ((j == 1) ? 1 : 0);
    return (r > r0) ? Hj[(size_t)i * n + r] : ((r == r0) ? 1.0 : 0.0);
}
PSD_HD double psd_fq_tau(const double* tau, int n, int j, int i) {
    if (i >= n - 1) return 0.0;
    const int m = n - i - ((j == 1) ? 1 : 0);  // length of the reflector
    return (m >= 2) ? tau[(size_t)(j - 1) * n + i] : 0.0;
}

// T of block PSD_BLOCK_X of factor j0 + PSD_BLOCK_Y + 1.  grid = (nblk, factors), 256 threads.
PSD_KERNEL_B(PSD_FQ_NT) psd_fq_tfactor(psd_fq_args g) {
    PSD_LDS_DECL;
    double* Vs = (double*)psd_lds;   // [64][33]
    double* G = Vs + 64 * 33;        // [32][33]
    double* Tm = G + 32 * 33;        // [32][33]
    double* tmp = Tm + 32 * 33;      // [32]
    const int n = g.n, b = PSD_BLOCK_X, jy = PSD_BLOCK_Y, j = g.j0 + jy + 1;
    const double* Hj = g.Hp + (size_t)(j - 1) * n * n;
    const int i0 = b * PSD_FQ_KB;
    PSD_PAR_FOR(t, PSD_FQ_NT) {
        const int a = t & 31, c = t >> 5;
        for (int q = 0; q < 4; ++q) {
            G[a * 33 + c + 8 * q] = 0.0;
            Tm[a * 33 + c + 8 * q] = 0.0;
        }
    }
    for (int rb = i0; rb < n; rb += 64) {
        PSD_SYNC();
        PSD_PAR_FOR(t, PSD_FQ_NT) {
            for (int q = 0; q < 8; ++q) {
                const int e = t + PSD_FQ_NT * q;
                const int rr = e & 63, ii = e >> 6;
                Vs[rr * 33 + ii] = psd_fq_v(Hj, n, j, rb + rr, i0 + ii);
            }
        }
        PSD_SYNC();
        PSD_PAR_FOR(t, PSD_FQ_NT) {
            const int a = t & 31, c = t >> 5;  // entries (a, c), (a, c + 8), (a, c + 16), (a, c + 24): this thread's alone
            double s0 = 0.0, s1 = 0.0, s2 = 0.0, s3 = 0.0;
            for (int rr = 0; rr < 64; ++rr) {
                const double va = Vs[rr * 33 + a];
                s0 += va * Vs[rr * 33 + c];
                s1 += va * Vs[rr * 33 + c + 8];
                s2 += va * Vs[rr * 33 + c + 16];
                s3 += va * Vs[rr * 33 + c + 24];
            }
            G[a * 33 + c] += s0;
            G[a * 33 + c + 8] += s1;
            G[a * 33 + c + 16] += s2;
            G[a * 33 + c + 24] += s3;
        }
    }
    PSD_SYNC();
    // dlarft, forward / columnwise: T(0:i-1, i) = -tau_i T(0:i-1, 0:i-1) (V(:, 0:i-1)' v_i), T(i, i) = tau_i
    for (int i = 0; i < PSD_FQ_KB; ++i) {
        const double ti = psd_fq_tau(g.tau, n, j, i0 + i);
        PSD_PAR_FOR(t, PSD_FQ_NT) {
            if (t < i) tmp[t] = -ti * G[t * 33 + i];
        }
        PSD_SYNC();
        PSD_PAR_FOR(t, PSD_FQ_NT) {
            if (t < i) {
                double s = 0.0;
                for (int k = t; k < i; ++k) s += Tm[t * 33 + k] * tmp[k];
                Tm[t * 33 + i] = s;
            } else if (t == i) {
                Tm[i * 33 + i] = ti;
            }
        }
        PSD_SYNC();
    }
    double* Tg = g.T + ((size_t)jy * g.nblk + b) * (PSD_FQ_KB * PSD_FQ_KB);
    PSD_PAR_FOR(t, PSD_FQ_NT) {
        for (int q = 0; q < 4; ++q) {
            const int e = t + PSD_FQ_NT * q;
            const int a = e & 31, c = e >> 5;
            Tg[c * PSD_FQ_KB + a] = Tm[a * 33 + c];
        }
    }
}

#ifndef PSD_HOSTSIM
typedef double psd_fq_d4 __attribute__((ext_vector_type(4)));

// Block step b: Q[rs.., tile] <- (I - V T V') Q[rs.., tile], rs = KB b.  grid = (tiles, factors), 256 threads.
__global__ void __launch_bounds__(PSD_FQ_NT) psd_fq_apply(psd_fq_args g, int b) {
    __shared__ double Tk[PSD_FQ_KB][PSD_FQ_KB + 1];  // Tk[k][i] = T(i, k)
    __shared__ double Vs[32][33];                    // phase A: Vs[row][reflector]
    __shared__ double Ms[32][65];                    // phase A: Ms[row][column]
    __shared__ double Ys[PSD_FQ_KB][PSD_FQ_TN + 8];  // Y, then W: [reflector][column]
    __shared__ double Vc[PSD_FQ_KB][64 + 8];         // phase C: Vc[reflector][row]
    const int n = g.n, jy = blockIdx.y, j = g.j0 + jy + 1;
    const int rs = b * PSD_FQ_KB, cs = rs + blockIdx.x * PSD_FQ_TN;
    if (cs >= n) return;
    const double* Hj = g.Hp + (size_t)(j - 1) * n * n;
    double* Qj = g.Q + (size_t)(j - 1) * n * n;
    const double* Tg = g.T + ((size_t)jy * g.nblk + b) * (PSD_FQ_KB * PSD_FQ_KB);
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int l15 = lane & 15, l4 = lane >> 4;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const int e = tid + PSD_FQ_NT * q;
        const int i = e & 31, k = e >> 5;
        Tk[k][i] = Tg[k * PSD_FQ_KB + i];
    }
    // ---- phase A: Y = V' Q[rs.., tile]; this wavefront: columns 16 wave .. 16 wave + 15, both halves of the 32 rows of Y
    psd_fq_d4 ya[2];
    ya[0] = psd_fq_d4{0.0, 0.0, 0.0, 0.0};
    ya[1] = psd_fq_d4{0.0, 0.0, 0.0, 0.0};
    for (int rb = rs; rb < n; rb += 32) {
        __syncthreads();
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int e = tid + PSD_FQ_NT * q;
            const int k = e & 31, i = e >> 5;
            Vs[k][i] = psd_fq_v(Hj, n, j, rb + k, rs + i);
        }
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            const int e = tid + PSD_FQ_NT * q;
            const int k = e & 31, c = e >> 5;
            Ms[k][c] = (rb + k < n && cs + c < n) ? Qj[(size_t)(cs + c) * n + (rb + k)] : 0.0;
        }
        __syncthreads();
#pragma unroll
        for (int kk = 0; kk < 32; kk += 4) {
            const double bq = Ms[kk + l4][16 * wave + l15];
            const double a0 = Vs[kk + l4][l15], a1 = Vs[kk + l4][16 + l15];
            ya[0] = __builtin_amdgcn_mfma_f64_16x16x4f64(a0, bq, ya[0], 0, 0, 0);
            ya[1] = __builtin_amdgcn_mfma_f64_16x16x4f64(a1, bq, ya[1], 0, 0, 0);
        }
    }
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int r = 0; r < 4; ++r) Ys[16 * a + l4 + 4 * r][16 * wave + l15] = ya[a][r];
    __syncthreads();
    // ---- phase B: W = T Y (this wavefront's 16 columns)
    psd_fq_d4 wa[2];
    wa[0] = psd_fq_d4{0.0, 0.0, 0.0, 0.0};
    wa[1] = psd_fq_d4{0.0, 0.0, 0.0, 0.0};
#pragma unroll
    for (int kk = 0; kk < PSD_FQ_KB; kk += 4) {
        const double bq = Ys[kk + l4][16 * wave + l15];
        const double a0 = Tk[kk + l4][l15], a1 = Tk[kk + l4][16 + l15];
        wa[0] = __builtin_amdgcn_mfma_f64_16x16x4f64(a0, bq, wa[0], 0, 0, 0);
        wa[1] = __builtin_amdgcn_mfma_f64_16x16x4f64(a1, bq, wa[1], 0, 0, 0);
    }
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int r = 0; r < 4; ++r) Ys[16 * a + l4 + 4 * r][16 * wave + l15] = wa[a][r];
    // ---- phase C: Q[rows, tile] -= V W, as (W' V')[column][row]; this wavefront: columns wc.., rows wr.. of the 64 x 64 chunk
    const int wc = (wave & 1) * 32, wr = (wave >> 1) * 32;
    for (int rb = rs; rb < n; rb += 64) {
        __syncthreads();
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            const int e = tid + PSD_FQ_NT * q;
            const int r = e & 63, k = e >> 6;
            Vc[k][r] = psd_fq_v(Hj, n, j, rb + r, rs + k);
        }
        __syncthreads();
        psd_fq_d4 d[2][2];
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
            for (int c = 0; c < 2; ++c) d[a][c] = psd_fq_d4{0.0, 0.0, 0.0, 0.0};
#pragma unroll
        for (int kk = 0; kk < PSD_FQ_KB; kk += 4) {
            double av[2], bv[2];
#pragma unroll
            for (int a = 0; a < 2; ++a) {
                av[a] = Ys[kk + l4][wc + 16 * a + l15];  // A[i = column][k] = W(k, column)
                bv[a] = Vc[kk + l4][wr + 16 * a + l15];  // B[k][j = row] = V(row, k)
            }
#pragma unroll
            for (int a = 0; a < 2; ++a)
#pragma unroll
                for (int c = 0; c < 2; ++c) d[a][c] = __builtin_amdgcn_mfma_f64_16x16x4f64(av[a], bv[c], d[a][c], 0, 0, 0);
        }
        // fragment: column index (lane & 15) = row of Q, row index (lane >> 4) + 4 reg = column of Q
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
            for (int c = 0; c < 2; ++c)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int col = cs + wc + 16 * a + l4 + 4 * r;
                    const int row = rb + wr + 16 * c + l15;
                    if (col < n && row < n) Qj[(size_t)col * n + row] -= d[a][c][r];
                }
    }
}
#else
// test-tier stand-in for the matrix-core kernel (plain loops, same argument block)
static void psd_fq_apply_sim(const psd_fq_args& g, int b, int nfac) {
    const int n = g.n, rs = b * PSD_FQ_KB, KB = PSD_FQ_KB;
    double Y[PSD_FQ_KB], W[PSD_FQ_KB];
    for (int jy = 0; jy < nfac; ++jy) {
        const int j = g.j0 + jy + 1;
        const double* Hj = g.Hp + (size_t)(j - 1) * n * n;
        double* Qj = g.Q + (size_t)(j - 1) * n * n;
        const double* Tg = g.T + ((size_t)jy * g.nblk + b) * (KB * KB);
        for (int c = rs; c < n; ++c) {
            for (int i = 0; i < KB; ++i) {
                double s = 0.0;
                for (int r = rs; r < n; ++r) s += psd_fq_v(Hj, n, j, r, rs + i) * Qj[(size_t)c * n + r];
                Y[i] = s;
            }
            for (int i = 0; i < KB; ++i) {
                double s = 0.0;
                for (int k = 0; k < KB; ++k) s += Tg[k * KB + i] * Y[k];
                W[i] = s;
            }
            for (int r = rs; r < n; ++r) {
                double s = 0.0;
                for (int k = 0; k < KB; ++k) s += psd_fq_v(Hj, n, j, r, rs + k) * W[k];
                Qj[(size_t)c * n + r] -= s;
            }
        }
    }
}
#endif
