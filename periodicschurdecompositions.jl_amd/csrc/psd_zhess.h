// Periodic Hessenberg-triangular reduction on the GPU, ComplexF64.
// Same structure as psd_hess.h; replaces phessenberg!(A) for complex eltype
// (PeriodicSchurDecompositions.jl:213-259) with the complex reflector of householder.jl:110-156
// (zlarfg: beta real, tau complex) and lmul!(H',A) / rmul!(A,H) / lmul!(H,A) of :190-266.
#pragma once
#include "psd_complex.h"
#include "psd_hess.h"

// x = A[r0:n, c] -> (beta, v); vbuf[0] = tau, vbuf[1..m-1] = v
PSD_D void psd_zhess_refl_body(psd_z* A, int n, int r0, int c, psd_z* vbuf, psd_z* tau_out) {
    PSD_LDS_DECL;
    double* red = (double*)psd_lds;
    const int NT = PSD_NTHREADS;
    const psd_mat<psd_z> M = psd_mat<psd_z>{A, n};
    const int m = n - r0 + 1;
    if (m < 1) return;
    // householder.jl:26-56: scaled 2-norm of the tail over real and imaginary parts
    PSD_PAR_FOR(t, NT) {
        double a = 0.0;
        for (int q = 1 + t; q < m; q += NT) a = fmax(a, zabs1(M(r0 + q, c)));
        red[t] = a;
    }
    PSD_SYNC();
    const double amax = psd_block_max(red, NT);
    double xnorm = 0.0;
    if (amax > 0.0) {
        PSD_PAR_FOR(t, NT) {
            double s = 0.0;
            for (int q = 1 + t; q < m; q += NT) {
                const psd_z y = M(r0 + q, c);
                const double yr = y.re / amax, yi = y.im / amax;
                s += yr * yr + yi * yi;
            }
            red[t] = s;
        }
        PSD_SYNC();
        xnorm = amax * sqrt(psd_block_sum(red, NT));
    }
    psd_z alpha = M(r0, c);
    double ar = alpha.re, ai = alpha.im;
    if (xnorm == 0.0 && ai == 0.0) {  // householder.jl:121-123
        PSD_SYNC();
        PSD_ONE {
            vbuf[0] = zmk(0.0, 0.0);
            if (tau_out) *tau_out = zmk(0.0, 0.0);
        }
        PSD_PAR_FOR(q, m - 1) { vbuf[1 + q] = M(r0 + 1 + q, c); }
        return;
    }
    // householder.jl:124-155, evaluated redundantly by every lane.  _hypot3 (:161-169)
    const double sfmin = PSD_DBL_MIN / PSD_DBL_EPS;
    double w = fmax(fabs(ar), fmax(fabs(ai), xnorm));
    double beta = -copysign(w * sqrt((ar / w) * (ar / w) + (ai / w) * (ai / w) + (xnorm / w) * (xnorm / w)), ar);
    int kount = 0;
    double acc = 1.0;
    if (fabs(beta) < sfmin) {
        const double rsfmin = 1.0 / sfmin;
        bool smallb = true;
        while (smallb) {
            kount += 1;
            acc *= rsfmin;
            beta *= rsfmin;
            ar *= rsfmin;
            ai *= rsfmin;
            smallb = (fabs(beta) < sfmin) && (kount < 20);
        }
        xnorm *= acc;
        w = fmax(fabs(ar), fmax(fabs(ai), xnorm));
        beta = -copysign(w * sqrt((ar / w) * (ar / w) + (ai / w) * (ai / w) + (xnorm / w) * (xnorm / w)), ar);
    }
    const psd_z tau = zmk((beta - ar) / beta, -ai / beta);
    const psd_z mult = zscal(acc, zdiv(zmk(1.0, 0.0), zmk(ar - beta, ai)));
    for (int q = 0; q < kount; ++q) beta *= sfmin;
    PSD_SYNC();
    PSD_PAR_FOR(q, m - 1) {
        const psd_z v = zmul(M(r0 + 1 + q, c), mult);
        M(r0 + 1 + q, c) = v;
        vbuf[1 + q] = v;
    }
    PSD_ONE {
        M(r0, c) = zmk(beta, 0.0);
        vbuf[0] = tau;
        if (tau_out) *tau_out = tau;
    }
}

PSD_KERNEL psd_zhess_refl(psd_z* A, int n, int r0, int c, psd_z* vbuf, psd_z* tau_out) { psd_zhess_refl_body(A, n, r0, c, vbuf, tau_out); }

// blocks [0,nL): AL[r0:n, lc0:n] <- H' AL ; blocks [nL,..): AR[:, r0:n] <- AR H
PSD_D void psd_zhess_apply_body(psd_z* AL, psd_z* AR, int n, int r0, int lc0, const psd_z* vbuf, int nL, int b) {
    PSD_LDS_DECL;
    const int NT = PSD_NTHREADS;  // 256
    const int m = n - r0 + 1;
    const psd_z tau = vbuf[0];
    if (ziszero(tau)) return;
    psd_z* red = (psd_z*)psd_lds;  // NT
    psd_z* vs = red + NT;          // m
    if (b < nL) {
        if (!AL) return;
        const psd_mat<psd_z> M = psd_mat<psd_z>{AL, n};
        const int cbase = lc0 + 4 * b;
        const psd_z tc = zconj(tau);
        PSD_PAR_FOR(t, NT) {
            const int wv = t >> 6, lane = t & 63;
            const int c = cbase + wv;
            psd_z s = zmk(0.0, 0.0);
            if (c <= n)
                for (int q = lane; q < m; q += 64) {
                    const psd_z a = M(r0 + q, c);
                    s = zadd(s, (q == 0) ? a : zmul(zconj(vbuf[q]), a));
                }
            red[t] = s;
        }
        PSD_SYNC();
        for (int s = 32; s > 0; s >>= 1) {
            PSD_PAR_FOR(t, NT) {
                if ((t & 63) < s) red[t] = zadd(red[t], red[t + s]);
            }
            PSD_SYNC();
        }
        PSD_PAR_FOR(t, NT) {
            const int wv = t >> 6, lane = t & 63;
            const int c = cbase + wv;
            if (c <= n) {
                const psd_z va = zmul(tc, red[wv << 6]);
                for (int q = lane; q < m; q += 64) {
                    const psd_z a = M(r0 + q, c);
                    M(r0 + q, c) = zsub(a, (q == 0) ? va : zmul(va, vbuf[q]));
                }
            }
        }
    } else {
        if (!AR) return;
        const psd_mat<psd_z> M = psd_mat<psd_z>{AR, n};
        const int rbase = 1 + PSD_HESS_RS * (b - nL);
        if (rbase > n) return;
        PSD_PAR_FOR(q, m) { vs[q] = (q == 0) ? zmk(1.0, 0.0) : vbuf[q]; }
        PSD_SYNC();
        PSD_PAR_FOR(t, NT) {
            const int ph = t / PSD_HESS_RS, r = rbase + (t & (PSD_HESS_RS - 1));
            psd_z s = zmk(0.0, 0.0);
            if (r <= n)
                for (int q = ph; q < m; q += PSD_HESS_NT / PSD_HESS_RS) s = zadd(s, zmul(M(r, r0 + q), vs[q]));
            red[t] = s;
        }
        PSD_SYNC();
        PSD_PAR_FOR(t, PSD_HESS_RS) {
            psd_z s = zmk(0.0, 0.0);
            for (int ph = 0; ph < PSD_HESS_NT / PSD_HESS_RS; ++ph) s = zadd(s, red[ph * PSD_HESS_RS + t]);
            red[t] = zmul(tau, s);
        }
        PSD_SYNC();
        PSD_PAR_FOR(t, NT) {
            const int ph = t / PSD_HESS_RS, r = rbase + (t & (PSD_HESS_RS - 1));
            if (r <= n) {
                const psd_z x = red[t & (PSD_HESS_RS - 1)];
                for (int q = ph; q < m; q += PSD_HESS_NT / PSD_HESS_RS) M(r, r0 + q) = zsub(M(r, r0 + q), zmul(x, zconj(vs[q])));
            }
        }
    }
}

PSD_KERNEL psd_zhess_apply(psd_z* AL, psd_z* AR, int n, int r0, int lc0, const psd_z* vbuf, int nL) {
    psd_zhess_apply_body(AL, AR, n, r0, lc0, vbuf, nL, PSD_BLOCK_X);
}
// two independent panel updates with the same reflector in one launch (see psd_hess_apply2)
PSD_KERNEL psd_zhess_apply2(psd_z* AL1, psd_z* AR1, int lc1, int nL1, int g1, psd_z* AL2, psd_z* AR2, int lc2, int nL2, int n,
                            int r0, const psd_z* vbuf, psd_z* vnext) {
    const int b = PSD_BLOCK_X;
    if (b < g1) psd_zhess_apply_body(AL1, AR1, n, r0, lc1, vbuf, nL1, b);
    else psd_zhess_apply_body(AL2, AR2, n, r0, lc2, vbuf, nL2, b - g1);
    if (b == 0 && vnext != nullptr) {  // (the next reflector behind the update of its column, see psd_hess_apply2)
        PSD_SYNC();
        psd_zhess_refl_body(AL1, n, r0 + 1, lc1, vnext, (psd_z*)nullptr);
    }
}

PSD_KERNEL psd_zset_identity(psd_z* Q, int n) {
    const int c = PSD_BLOCK_X + 1, j = PSD_BLOCK_Y + 1;
    const psd_mat<psd_z> M = psd_mat<psd_z>{Q + (size_t)(j - 1) * n * n, n};
    PSD_PAR_FOR(r, n) { M(r + 1, c) = zmk((r + 1 == c) ? 1.0 : 0.0, 0.0); }
}

// backward accumulation step of Q_j = H_{j,1} ... H_{j,n-1}: lmul!(H, Q), householder.jl:190-205
PSD_KERNEL psd_zformq_step(const psd_z* Hp, const psd_z* tau, psd_z* Q, int n, int i, int j0) {
    PSD_LDS_DECL;
    psd_z* red = (psd_z*)psd_lds;
    const int NT = PSD_NTHREADS;
    const int j = j0 + PSD_BLOCK_Y + 1;  // (j0: first factor of a period-sharded context's slice, 0-based)
    const int r0 = i + ((j == 1) ? 1 : 0);
    const int m = n - r0 + 1;
    if (m < 1) return;
    const psd_z tj = tau[(size_t)(j - 1) * n + (i - 1)];
    if (ziszero(tj)) return;
    const psd_mat<psd_z> V = psd_mat<psd_z>{const_cast<psd_z*>(Hp) + (size_t)(j - 1) * n * n, n};
    const psd_mat<psd_z> M = psd_mat<psd_z>{Q + (size_t)(j - 1) * n * n, n};
    const int cbase = r0 + 4 * PSD_BLOCK_X;
    if (cbase > n) return;
    PSD_PAR_FOR(t, NT) {
        const int wv = t >> 6, lane = t & 63;
        const int c = cbase + wv;
        psd_z s = zmk(0.0, 0.0);
        if (c <= n)
            for (int q = lane; q < m; q += 64) {
                const psd_z a = M(r0 + q, c);
                s = zadd(s, (q == 0) ? a : zmul(zconj(V(r0 + q, i)), a));
            }
        red[t] = s;
    }
    PSD_SYNC();
    for (int s = 32; s > 0; s >>= 1) {
        PSD_PAR_FOR(t, NT) {
            if ((t & 63) < s) red[t] = zadd(red[t], red[t + s]);
        }
        PSD_SYNC();
    }
    PSD_PAR_FOR(t, NT) {
        const int wv = t >> 6, lane = t & 63;
        const int c = cbase + wv;
        if (c <= n) {
            const psd_z va = zmul(tj, red[wv << 6]);
            for (int q = lane; q < m; q += 64) {
                const psd_z a = M(r0 + q, c);
                M(r0 + q, c) = zsub(a, (q == 0) ? va : zmul(va, V(r0 + q, i)));
            }
        }
    }
}

#ifndef PSD_HOSTSIM
// B consecutive steps of that accumulation in one pass over Q_j (n <= 64 NR): one wavefront per column, the column (rows
// cmin..n) in registers, reflectors i, i-1, .., i-B+1 applied to it one after the other (each read once).  The one-reflector
// launches read every column twice and wrote it once per reflector (3 x 16 bytes per element and reflector: 192 ms for
// Q_1..Q_64 at n = 1024, 3.8 TB/s); here the column moves once per B reflectors.  Rows of the column above a reflector's
// first row are left alone; a column left of it holds zeros there, so applying the reflector to it changes nothing.
template <int NR>
__global__ void __launch_bounds__(256) psd_zformq_blk(const psd_z* Hp, const psd_z* tau, psd_z* Q, int n, int i, int B, int j0) {
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int j = j0 + (int)blockIdx.y + 1;
    const int off1 = (j == 1) ? 1 : 0;
    const int ilow = (i - B + 1 >= 1) ? (i - B + 1) : 1;
    const int cmin = ilow + off1;  // first row (and column) any reflector of the block touches
    const int c = cmin + 4 * (int)blockIdx.x + wave;
    if (c > n) return;
    const psd_z* V = Hp + (size_t)(j - 1) * n * n;
    psd_z* M = Q + (size_t)(j - 1) * n * n;
    psd_z a[NR];
#pragma unroll
    for (int u = 0; u < NR; ++u) {
        const int r = cmin + lane + 64 * u;
        a[u] = (r <= n) ? M[(size_t)(c - 1) * n + (r - 1)] : zmk(0.0, 0.0);
    }
    for (int ik = i; ik >= ilow; --ik) {
        const int r0 = ik + off1;
        if (r0 > n) continue;
        const psd_z tj = tau[(size_t)(j - 1) * n + (ik - 1)];
        if (ziszero(tj)) continue;
        psd_z v[NR];
        psd_z dot = zmk(0.0, 0.0);
#pragma unroll
        for (int u = 0; u < NR; ++u) {
            const int r = cmin + lane + 64 * u;
            v[u] = zmk(0.0, 0.0);
            if (r == r0) v[u] = zmk(1.0, 0.0);
            else if (r > r0 && r <= n) v[u] = V[(size_t)(ik - 1) * n + (r - 1)];
            // (conj(v) a)
            dot.re += v[u].re * a[u].re + v[u].im * a[u].im;
            dot.im += v[u].re * a[u].im - v[u].im * a[u].re;
        }
#pragma unroll
        for (int sft = 32; sft > 0; sft >>= 1) {
            dot.re += __shfl_xor(dot.re, sft, 64);
            dot.im += __shfl_xor(dot.im, sft, 64);
        }
        const psd_z va = zmul(tj, dot);
#pragma unroll
        for (int u = 0; u < NR; ++u) a[u] = zsub(a[u], zmul(va, v[u]));
    }
#pragma unroll
    for (int u = 0; u < NR; ++u) {
        const int r = cmin + lane + 64 * u;
        if (r <= n) M[(size_t)(c - 1) * n + (r - 1)] = a[u];
    }
}
#endif

PSD_KERNEL psd_ztriu(psd_z* H, int n) {
    const int c = PSD_BLOCK_X + 1, j = PSD_BLOCK_Y + 1;
    const psd_mat<psd_z> M = psd_mat<psd_z>{H + (size_t)(j - 1) * n * n, n};
    const int first = c + 1 + ((j == 1) ? 1 : 0);
    PSD_PAR_FOR(t, n) {
        const int r = first + t;
        if (r <= n) M(r, c) = zmk(0.0, 0.0);
    }
}
