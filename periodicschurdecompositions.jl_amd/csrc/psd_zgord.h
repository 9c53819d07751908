// Eigenvalue reordering of a GeneralizedPeriodicSchur on the GPU by adjacent 1x1 swaps (ComplexF64; Float64 with a
// real spectrum is promoted by the host entry).
//
// Replaces ordschur!(P::GeneralizedPeriodicSchur, select) — /root/reference/src/ordschur.jl:11-73,323-328 with the
// signed swap `_swapadj1x1g!(T1, Ts, Zs, S, i1)` (sylswap.jl:638-764), the generalized scalar periodic Sylvester
// system (sylvester.jl:141-168,235-245) and `_updateλ!` (ordschur.jl:75-96).
//
// Structure as psd_zord.h (window of up to W-1 swaps per launch, O(p) structured solve).  Index conventions: device
// arrays are in the engine's internal right order with signature S_int; the reference's left-oriented sequence is
// X_l = T_{sigma(l)} with S_L[l] = S_int[sigma(l)].  Every rotation of the signed swap is (equivalent to) a standard
// rotation on (i, i+1) — the "backwards" Givens(2, 1, c, s') of :674-676 is the standard one with (c, -s) — and the
// rotation G_l is owned by Z_m, m = psd_ord_owner(p, l): it acts on T_m from the left if S_int[m] else from the
// right, and on T_{m-1} from the right if S_int[m-1] else from the left, which is the ownership rule of the signed QZ
// engine, so psd_zgq_apply serves the off-window updates.
#pragma once
#include "psd_zgz.h"
#include "psd_zord.h"

struct psd_zgoparams {
    psd_zgparams z;
    psd_ostate* st;
    const unsigned char* select;  // [n]
};

PSD_D void psd_zgord_step_body(const psd_zgoparams& O) {
    PSD_LDS_DECL;
    const psd_zgparams& P = O.z;
    psd_ostate st = *O.st;
    PSD_ONE { P.desc->active = 0; P.desc->defer_run = 0; }
    if (st.phase == PSD_OPH_DONE || st.phase == PSD_OPH_IDLE) return;
    const int n = st.n, p = st.p;
    psd_z* ldsz = (psd_z*)psd_lds;
    const size_t winb = (size_t)p * st.W * (st.W + 1);
    psd_z* sc = ldsz + winb;  // scratch: 15 arrays of p
    psd_z *T11 = sc, *T12 = sc + p, *T22 = sc + 2 * p, *Xv = sc + 3 * p, *Gs = sc + 4 * p, *wd = sc + 5 * p,
          *we = sc + 6 * p, *wf = sc + 7 * p, *wr = sc + 8 * p, *ca = sc + 9 * p, *cb = sc + 10 * p;
    psd_z* Txx = sc + 11 * p;             // 4 p
    double* Gc = (double*)(sc + 15 * p);  // p + 1 doubles
    int* lcnt = (int*)(Gc + p + 2);
    while (st.phase == PSD_OPH_SCAN) {  // ordschur.jl:53-65
        st.j += 1;
        if (st.j > n) {
            st.phase = PSD_OPH_DONE;
            break;
        }
        if (O.select[st.j - 1]) {
            st.js += 1;
            if (st.j != st.js) {
                st.here = st.j;
                st.phase = PSD_OPH_MOVE;
            }
        }
    }
    if (st.phase == PSD_OPH_MOVE) {
        const int nb = st.W - 1;
        const int hi = st.here;
        const int ilo = (hi - nb > st.js) ? (hi - nb) : st.js;
        psd_zwin w;
        w.b = ldsz;
        w.W = st.W;
        w.ld = st.W + 1;
        w.bsz = st.W * (st.W + 1);
        w.bs = ilo;
        w.be = hi;
        PSD_PAR_FOR(m, p) { lcnt[m] = 0; }
        psd_zgwin_load(P, w, n, p);
        bool failed = false;
        for (int i = hi - 1; i >= ilo && !failed; --i) {
            PSD_SYNC();
            PSD_PAR_FOR(t, p) {
                const int l = t + 1, sg = psd_ord_sigma(p, l);
                T11[t] = w.at(sg, i, i);
                T12[t] = w.at(sg, i, i + 1);
                T22[t] = w.at(sg, i + 1, i + 1);
                // sylvester.jl:141-168: row of X_l:  ca x_l + cb' x_{l+1} = -C_l with (ca, cb') = (A, -B) or (-B, A);
                // psd_ord_cycsolve takes (A, B) of  A x_l - B x_{l+1} = -C_l
                if (psd_zgsig(P, sg)) {
                    ca[t] = T11[t];
                    cb[t] = T22[t];
                } else {
                    ca[t] = zneg(T22[t]);
                    cb[t] = zneg(T11[t]);
                }
            }
            PSD_SYNC();
            PSD_ONE {  // sylswap.jl:657-730, one lane
                double n11 = 0.0, n12 = 0.0, n22 = 0.0;
                for (int t = 0; t < p; ++t) {
                    n11 = hypot(n11, zabs(T11[t]));
                    n12 = hypot(n12, zabs(T12[t]));
                    n22 = hypot(n22, zabs(T22[t]));
                }
                const double wmax = fmax(n11, fmax(n12, n22));
                const double h3 = (wmax == 0.0) ? 0.0
                                                 : wmax * sqrt((n11 / wmax) * (n11 / wmax) + (n12 / wmax) * (n12 / wmax) +
                                                               (n22 / wmax) * (n22 / wmax));
                const double thresh = fmax(20.0 * h3 * PSD_DBL_EPS, PSD_DBL_MIN);
                int flag = 0;  // 0 ok, 1 rejected, 2 singular
                psd_z r;
                if (!psd_ord_cycsolve(p, ca, cb, T12, Xv, wd, we, wf, wr)) flag = 2;
                if (!flag) {
                    for (int t = 0; t < p; ++t) {
                        const bool sl = psd_zgsig(P, psd_ord_sigma(p, t + 1));
                        if (sl) {
                            psd_zgivens(Xv[t], zmk(1.0, 0.0), Gc[t], Gs[t], r);
                        } else {  // Givens(2, 1, c, s') from (-X_l, 1)  ==  standard rotation (c, -s)
                            psd_zgivens(zneg(Xv[t]), zmk(1.0, 0.0), Gc[t], Gs[t], r);
                            Gs[t] = zneg(Gs[t]);
                        }
                    }
                    for (int t = 0; t < p; ++t) {
                        Txx[4 * t + 0] = T11[t];
                        Txx[4 * t + 1] = T12[t];
                        Txx[4 * t + 2] = zmk(0.0, 0.0);
                        Txx[4 * t + 3] = T22[t];
                    }
                    // X_l takes G_l on its columns (S_L[l]) or rows; X_{l-1} on its rows (S_L[l-1]) or columns
                    for (int t = 0; t < p; ++t) {
                        const int tp = (t == 0) ? (p - 1) : (t - 1);
                        psd_z* m = Txx + 4 * t;
                        if (psd_zgsig(P, psd_ord_sigma(p, t + 1))) {
                            psd_zrot_right_adj(Gc[t], Gs[t], m[0], m[1]);
                            psd_zrot_right_adj(Gc[t], Gs[t], m[2], m[3]);
                        } else {
                            psd_zrot_left(Gc[t], Gs[t], m[0], m[2]);
                            psd_zrot_left(Gc[t], Gs[t], m[1], m[3]);
                        }
                        psd_z* q = Txx + 4 * tp;
                        if (psd_zgsig(P, psd_ord_sigma(p, tp + 1))) {
                            psd_zrot_left(Gc[t], Gs[t], q[0], q[2]);
                            psd_zrot_left(Gc[t], Gs[t], q[1], q[3]);
                        } else {
                            psd_zrot_right_adj(Gc[t], Gs[t], q[0], q[1]);
                            psd_zrot_right_adj(Gc[t], Gs[t], q[2], q[3]);
                        }
                    }
                    double ws = 0.0;
                    for (int t = 0; t < p; ++t) ws += zabs(Txx[4 * t + 2]);
                    if (ws > thresh) flag = 1;
                    // strong test (:700-730): W_l = [c -s; conj(s) c];  S_L[l]: W_{l+1} Txx[l] W_l'  else  W_l Txx[l] W_{l+1}'
                    double ss = 0.0;
                    for (int t = 0; t < p; ++t) {
                        const int t1 = (t == p - 1) ? 0 : (t + 1);
                        const bool sl = psd_zgsig(P, psd_ord_sigma(p, t + 1));
                        const int ta = sl ? t1 : t, tb = sl ? t : t1;  // A = W_ta, C = W_tb
                        const psd_z a = zmk(Gc[ta], 0.0), b = zneg(Gs[ta]), cc = zconj(Gs[ta]), dd = zmk(Gc[ta], 0.0);
                        const psd_z* m = Txx + 4 * t;
                        const psd_z p0 = zadd(zmul(a, m[0]), zmul(b, m[2])), p1 = zadd(zmul(a, m[1]), zmul(b, m[3]));
                        const psd_z p2 = zadd(zmul(cc, m[0]), zmul(dd, m[2])), p3 = zadd(zmul(cc, m[1]), zmul(dd, m[3]));
                        const psd_z e0 = zmk(Gc[tb], 0.0), e1 = Gs[tb], e2 = zneg(zconj(Gs[tb])), e3 = zmk(Gc[tb], 0.0);
                        const psd_z r0 = zadd(zmul(p0, e0), zmul(p1, e2)), r1 = zadd(zmul(p0, e1), zmul(p1, e3));
                        const psd_z r2 = zadd(zmul(p2, e0), zmul(p3, e2)), r3 = zadd(zmul(p2, e1), zmul(p3, e3));
                        double dsum = 0.0;
                        dsum = hypot(dsum, zabs(zsub(r0, T11[t])));
                        dsum = hypot(dsum, zabs(zsub(r1, T12[t])));
                        dsum = hypot(dsum, zabs(r2));
                        dsum = hypot(dsum, zabs(zsub(r3, T22[t])));
                        ss = hypot(ss, dsum);
                    }
                    if (ss > thresh) flag = 1;
                }
                Gc[p] = (double)flag;
            }
            PSD_SYNC();
            const int flag = (int)Gc[p];
            if (flag) {
                st.info = (flag == 2) ? PSD_INFO_SINGULAR : (PSD_INFO_ILLCOND_BASE + st.j);
                failed = true;
                break;
            }
            // sylswap.jl:731-756 inside the window
            for (int l = 1; l <= p; ++l) {
                const int m = psd_ord_owner(p, l);
                const int mm1 = (m == 1) ? p : (m - 1);
                const double c = Gc[l - 1];
                const psd_z s = Gs[l - 1];
                if (psd_zgsig(P, mm1)) psd_zwin_right(w, mm1, i, c, s, 1, i + 1);
                else psd_zwin_left(w, mm1, i, c, s, i, n);
                if (psd_zgsig(P, m)) psd_zwin_left(w, m, i, c, s, i, n);
                else psd_zwin_right(w, m, i, c, s, 1, i + 1);
                psd_zgrecord(P, lcnt, m, i, c, s);
            }
            PSD_PAR_FOR(t, p) { w.at(t + 1, i + 1, i) = zmk(0.0, 0.0); }
            PSD_SYNC();
            st.nswaps += 1;
        }
        if (failed) {
            st.phase = PSD_OPH_DONE;
        } else {
            psd_zgwin_store(P, w, n, p);
            PSD_SYNC();
            PSD_PAR_FOR(m, p) { P.cnt[m] = lcnt[m]; }
            PSD_ONE {
                psd_gapply_desc d;
                d.active = 1;
                d.plo = ilo;
                d.phi = hi;
                d.lc0 = hi + 1;
                d.lc1 = n;
                d.rr0 = 1;
                d.rr1 = ilo - 1;
                d.zr0 = 1;
                d.zr1 = st.wantZ ? n : 0;
                d.defer_h1 = 0;
                d.defer_run = 0;
                d.djlo = d.djhi = d.drow0 = 0;
                d.h1mode = 0;
                d.h1c0 = 0;
                *P.desc = d;
            }
            st.nwindows += 1;
            st.here = ilo;
            if (ilo <= st.js) st.phase = PSD_OPH_SCAN;
        }
    }
    PSD_SYNC();
    PSD_ONE { *O.st = st; }
}

PSD_KERNEL_B(PSD_STEP_NT) psd_zgord_step(psd_zgoparams O) { psd_zgord_step_body(O); }

// pipelined driver: see psd_oslot (psd_zord.h)
PSD_KERNEL_B(PSD_STEP_NT) psd_zgord_step_mb(psd_zgoparams O, int p, int cstride) {
    const int s = PSD_BLOCK_X;
    O.st += s;
    O.z.desc += s;
    O.z.cnt += (size_t)s * cstride;
    O.z.tr += (size_t)s * p * PSD_GTR_CAP;
    psd_zgord_step_body(O);
}

PSD_KERNEL psd_zgord_init(psd_zgoparams O, int n, int p, int wantZ, int W) {
    PSD_ONE {
        psd_ostate st;
        st.n = n; st.p = p; st.wantZ = wantZ; st.W = W;
        st.phase = PSD_OPH_SCAN; st.info = 0;
        st.j = 0; st.js = 0; st.here = 0; st.nswaps = 0; st.nwindows = 0;
        *O.st = st;
        O.z.desc->active = 0;
        O.z.desc->defer_run = 0;
    }
}

// ordschur.jl:75-96 _updateλ!(P::GeneralizedPeriodicSchur): scaled eigenvalue j from the diagonals.  grid over j
PSD_KERNEL psd_zgord_values(psd_zgparams P, int n, int p) {
    const int NT = PSD_NTHREADS;
    PSD_PAR_FOR(t, NT) {
        const int j = 1 + PSD_BLOCK_X * NT + t;
        if (j <= n) {
            psd_z a;
            double b;
            int sc;
            psd_zg_safeprod(P, n, p, j, a, b, sc);
            P.alpha[j - 1] = a;
            P.beta[j - 1] = b;
            P.ascale[j - 1] = sc;
        }
    }
}
