// Device-side verifier of a (generalized) periodic Schur decomposition.
//
// Replaces checkpsd(P, As; thresh, strict) — /root/reference/src/diagnostics.jl:190-263 — for operands that are
// resident in HBM: per factor l
//   * || tril(T_l, -1) ||_F (tril(., -2) for the quasi-triangular factor of a real decomposition) — triangularity,
//   * || Z_l Z_l' - I ||_F                                                                      — orthogonality,
//   * || Z_a T_l Z_b' - A_l ||_F / eps / opnorm(A_l, 1), (a, b) = (l, l+1) or (l+1, l)           — factorization error.
// This is the one place of the path with dense contractions (three n x n x n products per factor), so it runs on the
// matrix cores: v_mfma_f64_16x16x4_f64, 64 x 64 output tile per 256-thread workgroup (each wavefront a 32 x 32 quarter
// = 2 x 2 MFMA tiles), operands staged through LDS in K-steps of 16.  ComplexF64 = four real products per tile on the
// de-interleaved planes.  The residual / orthogonality norms are reduced in the epilogue (the n x n result of the
// second product is never written); only W = T_l Z_b' goes through HBM.
#pragma once
#include "psd_complex.h"

#define PSD_CK_TM 64
#define PSD_CK_TK 16
#define PSD_CK_LD (PSD_CK_TM + 16)  // LDS row stride: k rows of a fragment fall into different bank halves

struct psd_ck_args {
    const double* A;  // left operand, column-major, ld = n (complex: interleaved)
    const double* B;  // right operand
    double* C;        // output (mode 0) or nullptr
    const double* D;  // matrix subtracted before the norm (mode 1) or nullptr (mode 2: identity)
    double* acc;      // mode 1/2: sum of squares accumulated here (one double)
    int n;
    int mode;    // 0: C = A op(B); 1: acc += ||A op(B) - D||_F^2; 2: acc += ||A op(B) - I||_F^2
    int bconjt;  // op(B) = B' (conjugate transpose) if 1, B if 0
};

#ifndef PSD_HOSTSIM
typedef double psd_d4 __attribute__((ext_vector_type(4)));

template <bool CPLX>
__global__ void __launch_bounds__(256) psd_ck_gemm(psd_ck_args g) {
    constexpr int E = CPLX ? 2 : 1;
    __shared__ double As[E][PSD_CK_TK][PSD_CK_LD];
    __shared__ double Bs[E][PSD_CK_TK][PSD_CK_LD];
    __shared__ double red[4];
    const int n = g.n;
    const int i0 = blockIdx.x * PSD_CK_TM, j0 = blockIdx.y * PSD_CK_TM;
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int wi = (wave & 1) * 32, wj = (wave >> 1) * 32;  // this wavefront's quarter of the tile
    psd_d4 cr[2][2], ci[2][2];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b) {
            cr[a][b] = psd_d4{0.0, 0.0, 0.0, 0.0};
            ci[a][b] = psd_d4{0.0, 0.0, 0.0, 0.0};
        }
    for (int k0 = 0; k0 < n; k0 += PSD_CK_TK) {
        // stage A[i0.., k0..] as As[k][i] and op(B)[k0.., j0..] as Bs[k][j]
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int e = tid + 256 * q;  // 0..1023
            {
                const int i = e & 63, k = e >> 6;
                double re = 0.0, im = 0.0;
                if (i0 + i < n && k0 + k < n) {
                    const size_t o = (size_t)(k0 + k) * n + (i0 + i);
                    if (CPLX) {
                        re = g.A[2 * o];
                        im = g.A[2 * o + 1];
                    } else {
                        re = g.A[o];
                    }
                }
                As[0][k][i] = re;
                if (CPLX) As[E - 1][k][i] = im;
            }
            if (g.bconjt) {  // op(B)[k][j] = conj(B[j][k]): contiguous in j
                const int j = e & 63, k = e >> 6;
                double re = 0.0, im = 0.0;
                if (j0 + j < n && k0 + k < n) {
                    const size_t o = (size_t)(k0 + k) * n + (j0 + j);
                    if (CPLX) {
                        re = g.B[2 * o];
                        im = -g.B[2 * o + 1];
                    } else {
                        re = g.B[o];
                    }
                }
                Bs[0][k][j] = re;
                if (CPLX) Bs[E - 1][k][j] = im;
            } else {  // B[k][j]: contiguous in k
                const int k = e & 15, j = e >> 4;
                double re = 0.0, im = 0.0;
                if (j0 + j < n && k0 + k < n) {
                    const size_t o = (size_t)(j0 + j) * n + (k0 + k);
                    if (CPLX) {
                        re = g.B[2 * o];
                        im = g.B[2 * o + 1];
                    } else {
                        re = g.B[o];
                    }
                }
                Bs[0][k][j] = re;
                if (CPLX) Bs[E - 1][k][j] = im;
            }
        }
        __syncthreads();
#pragma unroll
        for (int kk = 0; kk < PSD_CK_TK; kk += 4) {
            const int kr = kk + (lane >> 4), col = lane & 15;
            double ar[2], br[2], ai[2], bi[2];
#pragma unroll
            for (int a = 0; a < 2; ++a) {
                ar[a] = As[0][kr][wi + 16 * a + col];
                br[a] = Bs[0][kr][wj + 16 * a + col];
                if (CPLX) {
                    ai[a] = As[E - 1][kr][wi + 16 * a + col];
                    bi[a] = Bs[E - 1][kr][wj + 16 * a + col];
                }
            }
#pragma unroll
            for (int a = 0; a < 2; ++a)
#pragma unroll
                for (int b = 0; b < 2; ++b) {
                    cr[a][b] = __builtin_amdgcn_mfma_f64_16x16x4f64(ar[a], br[b], cr[a][b], 0, 0, 0);
                    if (CPLX) {
                        cr[a][b] = __builtin_amdgcn_mfma_f64_16x16x4f64(-ai[a], bi[b], cr[a][b], 0, 0, 0);
                        ci[a][b] = __builtin_amdgcn_mfma_f64_16x16x4f64(ar[a], bi[b], ci[a][b], 0, 0, 0);
                        ci[a][b] = __builtin_amdgcn_mfma_f64_16x16x4f64(ai[a], br[b], ci[a][b], 0, 0, 0);
                    }
                }
        }
        __syncthreads();
    }
    // epilogue: C/D fragment of v_mfma_f64_16x16x4_f64: col = lane & 15, row = (lane >> 4) + 4 * reg
    double ss = 0.0;
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int i = i0 + wi + 16 * a + (lane >> 4) + 4 * r;
                const int j = j0 + wj + 16 * b + (lane & 15);
                if (i >= n || j >= n) continue;
                const size_t o = (size_t)j * n + i;
                double vr = cr[a][b][r], vi = CPLX ? ci[a][b][r] : 0.0;
                if (g.mode == 0) {
                    if (CPLX) {
                        g.C[2 * o] = vr;
                        g.C[2 * o + 1] = vi;
                    } else {
                        g.C[o] = vr;
                    }
                } else {
                    if (g.mode == 1) {
                        if (CPLX) {
                            vr -= g.D[2 * o];
                            vi -= g.D[2 * o + 1];
                        } else {
                            vr -= g.D[o];
                        }
                    } else if (i == j) {
                        vr -= 1.0;
                    }
                    ss += vr * vr + vi * vi;
                }
            }
    if (g.mode != 0) {
        for (int s = 32; s > 0; s >>= 1) ss += __shfl_down(ss, s, 64);
        if (lane == 0) red[wave] = ss;
        __syncthreads();
        if (tid == 0) atomicAdd(g.acc, red[0] + red[1] + red[2] + red[3]);
    }
}
#endif

// per-factor small reductions: out[0] = || tril(T, -(1 + sub)) ||_F^2, out[1] = opnorm(A, 1); out[2] = number of
// non-zero first-subdiagonal entries below a real eigenvalue (the reference only warns about those).
// grid = 1 block of 256 threads.  `wi`: imaginary parts of the eigenvalues (or nullptr).
template <bool CPLX>
PSD_D void psd_ck_small_body(const double* T, const double* A, int n, int sub, const double* wi, double* out) {
    PSD_LDS_DECL;
    double* red = (double*)psd_lds;
    const int NT = PSD_NTHREADS;
    constexpr int E = CPLX ? 2 : 1;
    PSD_PAR_FOR(t, NT) {
        double s = 0.0;
        for (int c = t; c < n; c += NT)
            for (int r = c + 1 + sub; r < n; ++r)
                for (int e = 0; e < E; ++e) {
                    const double v = T[E * ((size_t)c * n + r) + e];
                    s += v * v;
                }
        red[t] = s;
    }
    PSD_SYNC();
    PSD_ONE {
        double s = 0.0;
        for (int t = 0; t < NT; ++t) s += red[t];
        out[0] = s;
    }
    PSD_SYNC();
    PSD_PAR_FOR(t, NT) {
        double best = 0.0;
        for (int c = t; c < n; c += NT) {
            double s = 0.0;
            for (int r = 0; r < n; ++r) {
                const size_t o = (size_t)c * n + r;
                s += CPLX ? hypot(A[2 * o], A[2 * o + 1]) : fabs(A[o]);
            }
            if (s > best) best = s;
        }
        red[t] = best;
    }
    PSD_SYNC();
    PSD_ONE {
        double best = 0.0;
        for (int t = 0; t < NT; ++t)
            if (red[t] > best) best = red[t];
        out[1] = best;
        int bad = 0;
        if (sub && wi)
            for (int j = 0; j + 1 < n; ++j)
                if (wi[j] == 0.0 && T[(size_t)j * n + j + 1] != 0.0) ++bad;
        out[2] = (double)bad;
    }
}
PSD_KERNEL psd_ck_small_d(const double* T, const double* A, int n, int sub, const double* wi, double* out) {
    psd_ck_small_body<false>(T, A, n, sub, wi, out);
}
PSD_KERNEL psd_ck_small_z(const double* T, const double* A, int n, int sub, const double* wi, double* out) {
    psd_ck_small_body<true>(T, A, n, sub, wi, out);
}

#ifdef PSD_HOSTSIM
// test-tier stand-in for the matrix-core kernel (plain loops, same argument block)
template <bool CPLX>
static void psd_ck_gemm_sim(const psd_ck_args& g) {
    const int n = g.n;
    double ss = 0.0;
    for (int j = 0; j < n; ++j)
        for (int i = 0; i < n; ++i) {
            double vr = 0.0, vi = 0.0;
            for (int k = 0; k < n; ++k) {
                const size_t oa = (size_t)k * n + i;
                const size_t ob = g.bconjt ? ((size_t)k * n + j) : ((size_t)j * n + k);
                if (CPLX) {
                    const double ar = g.A[2 * oa], ai = g.A[2 * oa + 1];
                    const double br = g.B[2 * ob], bi = g.bconjt ? -g.B[2 * ob + 1] : g.B[2 * ob + 1];
                    vr += ar * br - ai * bi;
                    vi += ar * bi + ai * br;
                } else {
                    vr += g.A[oa] * g.B[ob];
                }
            }
            const size_t o = (size_t)j * n + i;
            if (g.mode == 0) {
                if (CPLX) {
                    g.C[2 * o] = vr;
                    g.C[2 * o + 1] = vi;
                } else {
                    g.C[o] = vr;
                }
            } else {
                if (g.mode == 1) {
                    if (CPLX) {
                        vr -= g.D[2 * o];
                        vi -= g.D[2 * o + 1];
                    } else {
                        vr -= g.D[o];
                    }
                } else if (i == j) {
                    vr -= 1.0;
                }
                ss += vr * vr + vi * vi;
            }
        }
    if (g.mode != 0) *g.acc += ss;
}
#endif
