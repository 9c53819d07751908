// ordschur! of a real GeneralizedPeriodicSchur with 1x1 and 2x2 blocks: the signed swaps live in psd_rord.h
// (signature-aware psd_rord_step / psd_rord_apply); this header adds the eigenvalue update of
// /root/reference/src/ordschur.jl:206-314 (scaled form; 2x2 blocks by `_rpeigvals2x2` with the signature) and the
// clean-up of rordschur.jl:117-130 for that result type.
#pragma once
#include "psd_rgz.h"
#include "psd_rord.h"

// one thread per position j that starts a block; X scratch: P.xscr + (j-1) * p * 8 doubles = p * 4 complex
PSD_KERNEL psd_grord_values(psd_roparams P, int n, int p) {
    const int NT = PSD_NTHREADS;
    psd_gparams G;
    G.H = P.H;
    G.S = P.S;
    const psd_mat<double> A1 = psd_mat<double>{P.H, n};
    PSD_PAR_FOR(t, NT) {
        const int j = 1 + PSD_BLOCK_X * NT + t;
        if (j <= n) {
            const bool second = (j > 1) && (A1(j, j - 1) != 0);
            const bool first = (j < n) && (A1(j + 1, j) != 0);
            if (first && !second) {
                psd_z* X = (psd_z*)(P.xscr + (size_t)(j - 1) * p * 8);
                for (int l = 0; l < p; ++l) {
                    const psd_mat<double> M = psd_mat<double>{P.H + (size_t)l * n * n, n};
                    X[4 * l + 0] = zmk(M(j, j), 0.0);
                    X[4 * l + 1] = zmk(M(j, j + 1), 0.0);
                    X[4 * l + 2] = zmk(M(j + 1, j), 0.0);
                    X[4 * l + 3] = zmk(M(j + 1, j + 1), 0.0);
                }
                psd_z a2[2];
                double b2[2], sc2[2];
                bool cvg, good;
                psd_g_eigpair(G, p, X, a2, b2, sc2, cvg, good);
                for (int q = 0; q < 2; ++q) {
                    P.alpha[j - 1 + q] = a2[q];
                    P.beta[j - 1 + q] = b2[q];
                    P.ascale[j - 1 + q] = (int)sc2[q];
                }
            } else if (!second) {
                double a, b;
                int sc;
                psd_g_safeprod(G, n, p, j, a, b, sc);
                P.alpha[j - 1] = zmk(a, 0.0);
                P.beta[j - 1] = b;
                P.ascale[j - 1] = sc;
            }
        }
    }
}

// rordschur.jl:117-130: zero everything below the (1x1 / 2x2) diagonal blocks of T_1.  grid over columns.
PSD_KERNEL psd_grord_cleanup(psd_roparams P, int n) {
    const int c = PSD_BLOCK_X + 1;
    const psd_mat<double> A1 = psd_mat<double>{P.H, n};
    const int j0 = (P.alpha[c - 1].im > 0.0) ? (c + 2) : (c + 1);
    PSD_PAR_FOR(t, n) {
        const int r = j0 + t;
        if (r <= n) A1(r, c) = 0.0;
    }
}
