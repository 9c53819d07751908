// Periodic Hessenberg-triangular reduction on the GPU (real).
//
// Replaces phessenberg!(A) — /root/reference/src/PeriodicSchurDecompositions.jl:213-259 with the
// reflector kernels of /root/reference/src/householder.jl:66-108 (_xreflector!), :207-237
// (rmul!(A,H), lmul!(H',A)) — and the Q materialisation of PSD.jl:136-143,180-197.
//
// Structure: the reduction is a serial chain of n*p reflector generations (each needs the column
// produced by the previous right-update).  Per chain link (i, j):
//   psd_hess_refl   one workgroup: column norm (tree reduction in LDS), beta/tau, scaled v; the
//                   reflector is stored LAPACK-style below the diagonal of A_j and in a staging
//                   vector for the update kernel;
//   psd_hess_apply  wide kernel: blocks [0,nL) apply H' from the left to A_j (one wavefront per
//                   column, lanes down the column: coalesced), blocks [nL,..) apply H from the
//                   right to A_{j-1} (8-row strips, 32 column phases, v staged in LDS: many short strips
//                   keep all compute units busy).
// Q_j are formed afterwards by backward accumulation, all p factors per launch.
#pragma once
#include "psd_scalar.h"

#define PSD_HESS_NT 256
#define PSD_HESS_RS 8  // rows per strip of the right-hand update (NT / RS column phases per strip)

PSD_D double psd_block_max(double* red, int NT) {
    for (int s = NT / 2; s > 0; s >>= 1) {
        PSD_PAR_FOR(t, s) { red[t] = fmax(red[t], red[t + s]); }
        PSD_SYNC();
    }
    const double r = red[0];
    PSD_SYNC();
    return r;
}
PSD_D double psd_block_sum(double* red, int NT) {
    for (int s = NT / 2; s > 0; s >>= 1) {
        PSD_PAR_FOR(t, s) { red[t] += red[t + s]; }
        PSD_SYNC();
    }
    const double r = red[0];
    PSD_SYNC();
    return r;
}

// x = A[r0:n, c]  ->  (beta, v);  vbuf[0] = tau, vbuf[1..m-1] = v (v0 = 1 implicit); tau_out = tau
PSD_D void psd_hess_refl_body(double* A, int n, int r0, int c, double* vbuf, double* tau_out) {
    PSD_LDS_DECL;
    double* red = (double*)psd_lds;
    const int NT = PSD_NTHREADS;
    const psd_mat<double> M = psd_mat<double>{A, n};
    const int m = n - r0 + 1;
    if (m <= 1) {
        PSD_ONE {
            vbuf[0] = 0.0;
            if (tau_out) *tau_out = 0.0;
        }
        return;
    }
    // householder.jl:5-24: scaled 2-norm of the tail
    PSD_PAR_FOR(t, NT) {
        double a = 0.0;
        for (int q = 1 + t; q < m; q += NT) a = fmax(a, fabs(M(r0 + q, c)));
        red[t] = a;
    }
    PSD_SYNC();
    const double amax = psd_block_max(red, NT);
    double xnorm = 0.0;
    if (amax > 0.0) {
        PSD_PAR_FOR(t, NT) {
            double s = 0.0;
            for (int q = 1 + t; q < m; q += NT) {
                const double y = M(r0 + q, c) / amax;
                s += y * y;
            }
            red[t] = s;
        }
        PSD_SYNC();
        xnorm = amax * sqrt(psd_block_sum(red, NT));
    }
    if (xnorm == 0.0) {  // householder.jl:74-76: H = I
        PSD_ONE {
            vbuf[0] = 0.0;
            if (tau_out) *tau_out = 0.0;
        }
        PSD_PAR_FOR(q, m - 1) { vbuf[1 + q] = M(r0 + 1 + q, c); }
        return;
    }
    // householder.jl:77-105 (dlarfg), evaluated redundantly by every lane
    const double sfmin = 2.0 * PSD_DBL_MIN / PSD_DBL_EPS;
    double alpha = M(r0, c);
    double beta = -copysign(hypot(alpha, xnorm), alpha);
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
    const double tau = (beta - alpha) / beta;
    const double mult = acc * (1.0 / (alpha - beta));
    for (int q = 0; q < kount; ++q) beta *= sfmin;
    PSD_SYNC();
    PSD_PAR_FOR(q, m - 1) {
        const double v = M(r0 + 1 + q, c) * mult;
        M(r0 + 1 + q, c) = v;
        vbuf[1 + q] = v;
    }
    PSD_ONE {
        M(r0, c) = beta;
        vbuf[0] = tau;
        if (tau_out) *tau_out = tau;
    }
}

// H = I - tau [1;v][1;v]' of length m = n - r0 + 1.
//   blocks [0, nL):   AL[r0:n, lc0:n] <- H' AL[r0:n, lc0:n]      (PSD.jl:238,245)
//   blocks [nL, ..):  AR[:, r0:n]     <- AR[:, r0:n] H           (PSD.jl:239,246)
PSD_D void psd_hess_apply_body(double* AL, double* AR, int n, int r0, int lc0, const double* vbuf, int nL, int b) {
    PSD_LDS_DECL;
    const int NT = PSD_NTHREADS;  // 256
    const int m = n - r0 + 1;
    const double tau = vbuf[0];
    if (tau == 0.0) return;
    double* red = (double*)psd_lds;  // NT doubles
    double* vs = red + NT;           // m doubles (v0 = 1)
    if (b < nL) {
        if (!AL) return;
        const psd_mat<double> M = psd_mat<double>{AL, n};
        const int cbase = lc0 + 4 * b;
        PSD_PAR_FOR(t, NT) {
            const int wv = t >> 6, lane = t & 63;
            const int c = cbase + wv;
            double s = 0.0;
            if (c <= n)
                for (int q = lane; q < m; q += 64) s += ((q == 0) ? 1.0 : vbuf[q]) * M(r0 + q, c);
            red[t] = s;
        }
        PSD_SYNC();
        for (int s = 32; s > 0; s >>= 1) {  // per-wavefront tree
            PSD_PAR_FOR(t, NT) {
                if ((t & 63) < s) red[t] += red[t + s];
            }
            PSD_SYNC();
        }
        PSD_PAR_FOR(t, NT) {
            const int wv = t >> 6, lane = t & 63;
            const int c = cbase + wv;
            if (c <= n) {
                const double w = tau * red[wv << 6];
                for (int q = lane; q < m; q += 64) M(r0 + q, c) -= w * ((q == 0) ? 1.0 : vbuf[q]);
            }
        }
    } else {
        if (!AR) return;
        const psd_mat<double> M = psd_mat<double>{AR, n};
        const int rbase = 1 + PSD_HESS_RS * (b - nL);
        if (rbase > n) return;
        PSD_PAR_FOR(q, m) { vs[q] = (q == 0) ? 1.0 : vbuf[q]; }
        PSD_SYNC();
        PSD_PAR_FOR(t, NT) {
            const int ph = t / PSD_HESS_RS, r = rbase + (t & (PSD_HESS_RS - 1));
            double s = 0.0;
            if (r <= n)
                for (int q = ph; q < m; q += PSD_HESS_NT / PSD_HESS_RS) s += M(r, r0 + q) * vs[q];
            red[t] = s;
        }
        PSD_SYNC();
        PSD_PAR_FOR(t, PSD_HESS_RS) {
            double s = 0.0;
            for (int ph = 0; ph < PSD_HESS_NT / PSD_HESS_RS; ++ph) s += red[ph * PSD_HESS_RS + t];
            red[t] = tau * s;
        }
        PSD_SYNC();
        PSD_PAR_FOR(t, NT) {
            const int ph = t / PSD_HESS_RS, r = rbase + (t & (PSD_HESS_RS - 1));
            if (r <= n) {
                const double x = red[t & (PSD_HESS_RS - 1)];
                for (int q = ph; q < m; q += PSD_HESS_NT / PSD_HESS_RS) M(r, r0 + q) -= x * vs[q];
            }
        }
    }
}

PSD_KERNEL psd_hess_refl(double* A, int n, int r0, int c, double* vbuf, double* tau_out) {
    psd_hess_refl_body(A, n, r0, c, vbuf, tau_out);
}
PSD_KERNEL psd_hess_apply(double* AL, double* AR, int n, int r0, int lc0, const double* vbuf, int nL) {
    psd_hess_apply_body(AL, AR, n, r0, lc0, vbuf, nL, PSD_BLOCK_X);
}

// two independent panel updates with the same reflector in one launch (stage 1 of the signed reduction: the factor
// itself + its Q, and the neighbouring factor): blocks [0, g1) run the first operand set, the rest the second
// vnext != nullptr: workgroup 0 — the one that has just updated column lc1 of AL1, the next reflector's column — forms that
// reflector behind its update (rows r0 + 1.., into vnext; the launch of psd_hess_refl between two panel updates, one
// workgroup, cost 5 us plus a launch gap per link of the QR sweeps)
PSD_KERNEL psd_hess_apply2(double* AL1, double* AR1, int lc1, int nL1, int g1, double* AL2, double* AR2, int lc2, int nL2,
                           int n, int r0, const double* vbuf, double* vnext) {
    const int b = PSD_BLOCK_X;
    if (b < g1) psd_hess_apply_body(AL1, AR1, n, r0, lc1, vbuf, nL1, b);
    else psd_hess_apply_body(AL2, AR2, n, r0, lc2, vbuf, nL2, b - g1);
    if (b == 0 && vnext != nullptr) {
        PSD_SYNC();
        psd_hess_refl_body(AL1, n, r0 + 1, lc1, vnext, (double*)nullptr);
    }
}

// Graph-replay form of one column of the reduction.  The reduction is n-1 columns x p links x 2 launches of small
// kernels, and issuing them one by one leaves the GPU idle two thirds of the time (host launch rate).  The launch
// sequence of ONE column is captured into a hipGraph whose kernels take everything that changes from column to
// column (the column index i, and the operand pointers of this call) from a small argument block in device memory;
// the graph is then replayed n-1 times, psd_hess_next advancing i on the device.
struct psd_hess_args {
    double* H;
    double* tau;
    double* vbuf;
    int i, p;
};
PSD_KERNEL psd_hess_refl_g(const psd_hess_args* G, int n, int j) {
    const int i = G->i;
    const int r0 = (j == 1) ? (i + 1) : i;
    if (i > n - 1 || n - r0 + 1 < 2) return;
    psd_hess_refl_body(G->H + (size_t)(j - 1) * n * n, n, r0, i, G->vbuf, G->tau + (size_t)(j - 1) * n + (i - 1));
}
// grid = nLmax + nR blocks; mode 0: left on A_j and right on A_{j-1}; 1: left only; 2: right only (p == 1)
PSD_KERNEL psd_hess_apply_g(const psd_hess_args* G, int n, int j, int nLmax, int mode) {
    const int i = G->i;
    const int r0 = (j == 1) ? (i + 1) : i;
    if (i > n - 1 || n - r0 + 1 < 2) return;
    const int lc0 = i + 1;
    const int nL = (n - lc0 + 1 + 3) / 4;
    const int b = PSD_BLOCK_X;
    const int jm1 = (j == 1) ? G->p : (j - 1);
    double* Aj = G->H + (size_t)(j - 1) * n * n;
    double* Am = G->H + (size_t)(jm1 - 1) * n * n;
    if (b < nLmax) {
        if (b >= nL || mode == 2) return;
        psd_hess_apply_body(Aj, nullptr, n, r0, lc0, G->vbuf, nL, b);
    } else {
        if (mode == 1) return;
        psd_hess_apply_body(nullptr, Am, n, r0, lc0, G->vbuf, nL, nL + (b - nLmax));
    }
}
PSD_KERNEL psd_hess_next(psd_hess_args* G) {
    PSD_ONE { G->i += 1; }
}

// Q <- I for all p factors.  grid = (n, p)
PSD_KERNEL psd_set_identity(double* Q, int n) {
    const int c = PSD_BLOCK_X + 1, j = PSD_BLOCK_Y + 1;
    const psd_mat<double> M = psd_mat<double>{Q + (size_t)(j - 1) * n * n, n};
    PSD_PAR_FOR(r, n) { M(r + 1, c) = (r + 1 == c) ? 1.0 : 0.0; }
}

// One step (reflector index i) of the backward accumulation Q_j = H_{j,1} ... H_{j,n-1}, all
// factors at once (what Matrix(H.Q) / Matrix(QR.Q) produce, PSD.jl:136-143).  grid = (tiles, p).
// (j0: first factor of the slice this launch forms, 0-based: a period-sharded context forms the Q_j it owns)
PSD_KERNEL psd_formq_step(const double* Hp, const double* tau, double* Q, int n, int i, int j0) {
    PSD_LDS_DECL;
    double* red = (double*)psd_lds;
    const int NT = PSD_NTHREADS;
    const int j = j0 + PSD_BLOCK_Y + 1;
    const int r0 = i + ((j == 1) ? 1 : 0);
    const int m = n - r0 + 1;
    if (m < 2) return;
    const double tj = tau[(size_t)(j - 1) * n + (i - 1)];
    if (tj == 0.0) return;
    const psd_mat<double> V = psd_mat<double>{const_cast<double*>(Hp) + (size_t)(j - 1) * n * n, n};
    const psd_mat<double> M = psd_mat<double>{Q + (size_t)(j - 1) * n * n, n};
    const int cbase = r0 + 4 * PSD_BLOCK_X;
    if (cbase > n) return;
    PSD_PAR_FOR(t, NT) {
        const int wv = t >> 6, lane = t & 63;
        const int c = cbase + wv;
        double s = 0.0;
        if (c <= n)
            for (int q = lane; q < m; q += 64) s += ((q == 0) ? 1.0 : V(r0 + q, i)) * M(r0 + q, c);
        red[t] = s;
    }
    PSD_SYNC();
    for (int s = 32; s > 0; s >>= 1) {
        PSD_PAR_FOR(t, NT) {
            if ((t & 63) < s) red[t] += red[t + s];
        }
        PSD_SYNC();
    }
    PSD_PAR_FOR(t, NT) {
        const int wv = t >> 6, lane = t & 63;
        const int c = cbase + wv;
        if (c <= n) {
            const double w = tj * red[wv << 6];
            for (int q = lane; q < m; q += 64) M(r0 + q, c) -= w * ((q == 0) ? 1.0 : V(r0 + q, i));
        }
    }
}

// PSD.jl:147,149: keep R_j (j >= 2) / triu(H_1, -1): zero the reflector storage.  grid = (n, p)
PSD_KERNEL psd_triu(double* H, int n) {
    const int c = PSD_BLOCK_X + 1, j = PSD_BLOCK_Y + 1;
    const psd_mat<double> M = psd_mat<double>{H + (size_t)(j - 1) * n * n, n};
    const int first = c + 1 + ((j == 1) ? 1 : 0);
    PSD_PAR_FOR(t, n) {
        const int r = first + t;
        if (r <= n) M(r, c) = 0.0;
    }
}

// reverse the order of `cnt` consecutive blocks (each `ncols` columns of `collen` doubles) starting at
// block `first` (orientation 'L', PSD.jl:127-131,1078-1092).  grid = (ncols, cnt/2)
PSD_KERNEL psd_reverse_blocks(double* X, int collen, int ncols, int first, int cnt) {
    const int c = PSD_BLOCK_X, s = PSD_BLOCK_Y;
    const size_t blk = (size_t)collen * ncols;
    double* a = X + (size_t)(first + s) * blk + (size_t)c * collen;
    double* b = X + (size_t)(first + cnt - 1 - s) * blk + (size_t)c * collen;
    PSD_PAR_FOR(r, collen) {
        const double t = a[r];
        a[r] = b[r];
        b[r] = t;
    }
}

// n == 1 shortcut (PSD.jl:333-352)
PSD_KERNEL psd_scalar_product(const double* H, int p, double* wr, double* wi) {
    PSD_ONE {
        double l1 = H[0];
        for (int j = 1; j < p; ++j) l1 *= H[j];
        wr[0] = l1;
        wi[0] = 0.0;
    }
}
