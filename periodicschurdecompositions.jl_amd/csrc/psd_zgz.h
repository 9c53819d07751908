// Complex generalized (signed) periodic QZ iteration and signed Hessenberg reduction on the GPU.
//
// Replaces, for ComplexF64 and a signature with negative entries,
//   pschur!(H1, Hs, S; wantT, wantZ, Q, maxitfac)   /root/reference/src/generalized.jl:166-931 (MB03BZ type)
//   _phessenberg!(A, S; wantQ), stage 2              generalized.jl:1034-1079
// (the all-true signature keeps the tuned path of psd_zqz.h).  Same structure as the real signed engine psd_rgz.h:
// one wavefront chases a diagonal window of all p factors in LDS and emits one rotation list per owner Z_m; the bulk
// kernel is organised by factor, because the side an owner acts on depends on the signature:
//     rows    of H_l  <- owner  l      if S[l]  else  l+1
//     columns of H_l  <- owner  l+1    if S[l]  else  l        (cyclic)
// Implemented: deflation tests 1-3 (:323-353), controlled zero shift (:356-448), Case II (:453-566) and Case III
// (:568-740) directly on HBM (exactly singular factors only), 1x1 split with `_safeprod` (:741-762), single-shift
// sweep with the signed shift chain (:770-852), phase normalisation with signature (:860-908).
#pragma once
#include "psd_rgz.h"
#include "psd_zhqr.h"

struct psd_zgstate {
    int n, p, wantT, wantZ, W;
    int phase, info;
    int ilast, ifirst, ifirstm, ilastm, iiter, ziter, jiter, maxit;
    int jlo, kcur, zflag, hj;
    int nsweeps, nzshift, nsplit, ncase2, ncase3, nwindows, nlog, maxlog;
    double c0;
    psd_z s0;
    double smlnum, ulp, safmin;
    long long cyc[6];
    // multishift train (as psd_gstate); -2: explicit-shift start without a train (test hook)
    int train_want, train_n, train_id, cursor, train_tick0, ntrainsweeps;
    int cstart, cfirst;  // cursor: the tick of its first window and that window's number of positions (cursors W positions apart)
    int Wmax, train_oc;  // LDS layout width (W is the running sweep's, <= Wmax); o / c of the width rule (psd_rq_shift)
    psd_z sh;  // this bulge's shift
};

struct psd_zgparams {
    psd_z* H;
    psd_z* Z;
    const unsigned char* S;
    psd_zgstate* st;
    psd_gapply_desc* desc;
    psd_ztr* tr;   // [p][PSD_GTR_CAP]
    int* cnt;      // [p]
    psd_ztr* dG;   // [n+2]
    psd_z* alpha;  // [n]
    double* beta;  // [n]
    int* ascale;   // [n]
    int* log;
    psd_zgstate* cst;  // [PSD_TRAIN_MAX] cursor states of a train (entry 0 unused) or nullptr
    int* cep;  // [PSD_TRAIN_MAX] epoch words of the cursor states (psd_pub_*), then the count of finished cursors
    psd_z* tshift;     // [PSD_TRAIN_MAX] shifts, then a flag word
    int tick;          // launch index
    // period sharding (psd_set_shard): the owners m (1-based, inclusive) whose Schur vectors Z_m this context holds;
    // the updates of the others are some other rank's work (1..p without sharding)
    int zlo, zhi;
};

PSD_HD psd_mat<psd_z> psd_zgfac(const psd_zgparams& P, int n, int l) {
    return psd_mat<psd_z>{P.H + (size_t)(l - 1) * n * n, n};
}
PSD_HD bool psd_zgsig(const psd_zgparams& P, int l) { return P.S[l - 1] != 0; }
PSD_HD int psd_zgrowner(const psd_zgparams& P, int l, int p) { return psd_zgsig(P, l) ? l : psd_gnext(l, p); }
PSD_HD int psd_zgcowner(const psd_zgparams& P, int l, int p) { return psd_zgsig(P, l) ? psd_gnext(l, p) : l; }

#define PSD_NS ::
#include "psd_zgz_chain.inl"
#undef PSD_NS

PSD_D void psd_zglog(const psd_zgparams& P, psd_zgstate& st, int kind, int lo, int hi) {
    PSD_ONE {
        if (st.nlog < st.maxlog) {
            P.log[3 * st.nlog + 0] = kind;
            P.log[3 * st.nlog + 1] = lo;
            P.log[3 * st.nlog + 2] = hi;
        }
    }
    st.nlog += 1;
}
PSD_D void psd_zgdesc_write(const psd_zgparams& P, psd_zgstate& st, const int* lcnt, int plo, int phi, int lc0,
                            int lc1, int rr0, int rr1, int defer_h1, int defer_run, int djlo, int djhi, int h1mode = 0,
                            int h1c0 = 0) {
    PSD_SYNC();
    const bool over = psd_list_overflow(lcnt, st.p, PSD_GTR_CAP);
    if (over) {  // never apply truncated lists
        st.info = PSD_LIST_OVERFLOW;
        st.phase = PSD_GPH_DONE;
    }
    PSD_PAR_FOR(m, st.p) { P.cnt[m] = lcnt[m]; }
    PSD_ONE {
        psd_gapply_desc d;
        d.active = over ? 0 : 1;
        d.plo = plo;
        d.phi = phi;
        d.lc0 = lc0;
        d.lc1 = lc1;
        d.rr0 = rr0;
        d.rr1 = rr1;
        d.zr0 = 1;
        d.zr1 = st.wantZ ? st.n : 0;
        d.defer_h1 = defer_h1;
        d.defer_run = defer_run;
        d.djlo = djlo;
        d.djhi = djhi;
        d.drow0 = st.ifirstm;
        d.h1mode = h1mode;
        d.h1c0 = h1c0;
        *P.desc = d;
    }
    PSD_SYNC();
}

// generalized.jl:939-976 `_safeprod` with a signature
PSD_D void psd_zg_safeprod(const psd_zgparams& P, int n, int p, int idx, psd_z& alpha, double& beta, int& scale) {
    alpha = zmk(1.0, 0.0);
    beta = 1.0;
    scale = 0;
    for (int l = 1; l <= p; ++l) {
        const psd_z xi = psd_zgfac(P, n, l)(idx, idx);
        if (psd_zgsig(P, l)) {
            alpha = zmul(alpha, xi);
        } else if (ziszero(xi)) {
            beta = 0.0;
        } else {
            alpha = zdiv(alpha, xi);
        }
        if (zabs(alpha) == 0) {
            alpha = zmk(0.0, 0.0);
            scale = 0;
            if (beta == 0.0) return;
        } else {
            int guard = 0;
            while (zabs(alpha) < 1.0 && guard < 2200) {
                alpha = zscal(2.0, alpha);
                scale -= 1;
                ++guard;
            }
            while (zabs(alpha) >= 2.0 && guard < 4400) {
                alpha = zscal(0.5, alpha);
                scale += 1;
                ++guard;
            }
        }
    }
}

// ---- rotations applied directly on HBM by the whole workgroup (Cases II / III only) --------------------------------
PSD_D void psd_zgg_left(const psd_mat<psd_z>& M, int j, double c, psd_z s, int c0, int c1) {
    PSD_SYNC();
    PSD_PAR_FOR(t, c1 - c0 + 1) {
        const int cc = c0 + t;
        psd_z a1 = M(j, cc), a2 = M(j + 1, cc);
        psd_zrot_left(c, s, a1, a2);
        M(j, cc) = a1;
        M(j + 1, cc) = a2;
    }
    PSD_SYNC();
}
PSD_D void psd_zgg_right(const psd_mat<psd_z>& M, int j, double c, psd_z s, int r0, int r1) {
    PSD_SYNC();
    PSD_PAR_FOR(t, r1 - r0 + 1) {
        const int r = r0 + t;
        psd_z a1 = M(r, j), a2 = M(r, j + 1);
        psd_zrot_right_adj(c, s, a1, a2);
        M(r, j) = a1;
        M(r, j + 1) = a2;
    }
    PSD_SYNC();
}
PSD_D void psd_zgg_z(const psd_zgparams& P, const psd_zgstate& st, int m, int j, double c, psd_z s) {
    if (!st.wantZ || m < P.zlo || m > P.zhi) return;
    psd_zgg_right(psd_mat<psd_z>{P.Z + (size_t)(m - 1) * st.n * st.n, st.n}, j, c, s, 1, st.n);
}
PSD_D void psd_zgg_set2(const psd_mat<psd_z>& M, int r1, int c1, psd_z v1, int r2, int c2, psd_z v2) {
    PSD_SYNC();
    PSD_ONE {
        M(r1, c1) = v1;
        M(r2, c2) = v2;
    }
    PSD_SYNC();
}
PSD_D void psd_zgg_link(const psd_mat<psd_z>& M, int q, bool cols_in, double& c, psd_z& s, int rlo, int chi) {
    psd_z r;
    const psd_z z0 = zmk(0.0, 0.0);
    if (cols_in) {
        psd_zgg_right(M, q, c, s, rlo, q + 1);
        psd_zgivens(M(q, q), M(q + 1, q), c, s, r);
        psd_zgg_set2(M, q, q, r, q + 1, q, z0);
        psd_zgg_left(M, q, c, s, q + 1, chi);
    } else {
        psd_zgg_left(M, q, c, s, q, chi);
        psd_zgivens(M(q + 1, q + 1), zneg(M(q + 1, q)), c, s, r);
        psd_zgg_set2(M, q + 1, q + 1, r, q + 1, q, z0);
        psd_zgg_right(M, q, c, s, rlo, q);
    }
}

// generalized.jl:453-566 Case II
PSD_D void psd_zgq_case2(const psd_zgparams& P, psd_zgstate& st, int ldeflate, int jdeflate) {
    const int n = st.n, p = st.p, jlo = st.jlo, ilast = st.ilast, ifirstm = st.ifirstm, ilastm = st.ilastm;
    const psd_mat<psd_z> H1 = psd_zgfac(P, n, 1);
    const psd_z z0 = zmk(0.0, 0.0);
    st.ncase2 += 1;
    psd_zglog(P, st, 2, jlo, ilast);
    for (int j = jlo; j <= jdeflate - 1; ++j) {
        double c;
        psd_z s, r;
        psd_zgivens(H1(j, j), H1(j + 1, j), c, s, r);
        psd_zgg_set2(H1, j, j, r, j + 1, j, z0);
        psd_zgg_left(H1, j, c, s, j + 1, ilastm);
        psd_zgg_z(P, st, 1, j, c, s);
        for (int l = p; l >= 2; --l) {
            const int ntra = (l < ldeflate) ? (jdeflate - 2) : (jdeflate - 1);
            if (j > ntra) break;
            psd_zgg_link(psd_zgfac(P, n, l), j, psd_zgsig(P, l), c, s, ifirstm, ilastm);
            psd_zgg_z(P, st, l, j, c, s);
        }
        PSD_ONE {
            psd_ztr tr;
            tr.pos = j;
            tr.pad = 0;
            tr.c = c;
            tr.s = s;
            P.dG[j] = tr;
        }
    }
    PSD_SYNC();
    for (int j = jlo; j <= jdeflate - 2; ++j) {
        const psd_ztr g = P.dG[j];
        psd_zgg_right(H1, j, g.c, g.s, ifirstm, j + 1);
    }
    for (int j = ilast; j >= jdeflate + 1; --j) {
        double c;
        psd_z s, r;
        psd_zgivens(H1(j, j), zneg(H1(j, j - 1)), c, s, r);  // Givens(j, j-1, c, s') == standard (j-1, j; c, -s)
        psd_zgg_set2(H1, j, j, r, j, j - 1, z0);
        psd_zgg_right(H1, j - 1, c, s, ifirstm, j - 1);
        psd_zgg_z(P, st, psd_gnext(1, p), j - 1, c, s);
        bool alive = true;
        for (int l = 2; l <= p; ++l) {
            const int ntra = (l > ldeflate) ? (jdeflate + 2) : (jdeflate + 1);
            if (j < ntra) {
                alive = false;
                break;
            }
            psd_zgg_link(psd_zgfac(P, n, l), j - 1, !psd_zgsig(P, l), c, s, ifirstm, ilastm);
            psd_zgg_z(P, st, psd_gnext(l, p), j - 1, c, s);
        }
        PSD_ONE {
            psd_ztr tr;
            tr.pos = alive ? (j - 1) : -1;
            tr.pad = 0;
            tr.c = c;
            tr.s = s;
            P.dG[j] = tr;
        }
    }
    PSD_SYNC();
    for (int j = ilast; j >= jdeflate + 2; --j) {
        const psd_ztr g = P.dG[j];
        if (g.pos > 0) psd_zgg_left(H1, j - 1, g.c, g.s, j - 1, ilastm);
    }
}

// generalized.jl:568-740 Case III
PSD_D void psd_zgq_case3(const psd_zgparams& P, psd_zgstate& st, int ldeflate, int jdeflate) {
    const int n = st.n, p = st.p, jlo = st.jlo, ilast = st.ilast, ifirstm = st.ifirstm, ilastm = st.ilastm;
    const psd_mat<psd_z> H1 = psd_zgfac(P, n, 1);
    const psd_mat<psd_z> Hd = psd_zgfac(P, n, ldeflate);
    const psd_z z0 = zmk(0.0, 0.0);
    st.ncase3 += 1;
    psd_zglog(P, st, 3, jlo, ilast);
    double c;
    psd_z s, r;
    if (jdeflate > (ilast - jlo + 1) / 2.0) {
        for (int j1 = jdeflate; j1 <= ilast - 1; ++j1) {
            int j = j1;
            psd_zgivens(Hd(j, j + 1), Hd(j + 1, j + 1), c, s, r);
            psd_zgg_set2(Hd, j, j + 1, r, j + 1, j + 1, z0);
            psd_zgg_left(Hd, j, c, s, j + 2, ilastm);
            int ln = psd_gnext(ldeflate, p);
            psd_zgg_z(P, st, ln, j, c, s);
            for (int l = 1; l <= p - 1; ++l) {
                if (ln == 1) {
                    psd_zgg_left(H1, j, c, s, j - 1, ilastm);
                    psd_zgivens(H1(j + 1, j), zneg(H1(j + 1, j - 1)), c, s, r);
                    psd_zgg_set2(H1, j + 1, j, r, j + 1, j - 1, z0);
                    psd_zgg_right(H1, j - 1, c, s, ifirstm, j);
                    j -= 1;
                } else {
                    psd_zgg_link(psd_zgfac(P, n, ln), j, !psd_zgsig(P, ln), c, s, ifirstm, ilastm);
                }
                ln = psd_gnext(ln, p);
                psd_zgg_z(P, st, ln, j, c, s);
            }
            psd_zgg_right(Hd, j, c, s, ifirstm, j);
        }
        const int j = ilast;
        psd_zgivens(H1(j, j), zneg(H1(j, j - 1)), c, s, r);
        psd_zgg_set2(H1, j, j, r, j, j - 1, z0);
        psd_zgg_right(H1, j - 1, c, s, ifirstm, j - 1);
        psd_zgg_z(P, st, psd_gnext(1, p), j - 1, c, s);
        for (int l = 2; l <= ldeflate - 1; ++l) {
            psd_zgg_link(psd_zgfac(P, n, l), j - 1, !psd_zgsig(P, l), c, s, ifirstm, ilastm);
            psd_zgg_z(P, st, psd_gnext(l, p), j - 1, c, s);
        }
        psd_zgg_right(Hd, j - 1, c, s, ifirstm, j);
    } else {
        for (int j1 = jdeflate; j1 >= jlo + 1; --j1) {
            int j = j1;
            psd_zgivens(Hd(j - 1, j), zneg(Hd(j - 1, j - 1)), c, s, r);
            psd_zgg_set2(Hd, j - 1, j, r, j - 1, j - 1, z0);
            psd_zgg_right(Hd, j - 1, c, s, ifirstm, j - 2);
            psd_zgg_z(P, st, ldeflate, j - 1, c, s);
            int ln = ldeflate - 1;
            for (int l = 1; l <= p - 1; ++l) {
                if (ln == 1) {
                    psd_zgg_right(H1, j - 1, c, s, ifirstm, j + 1);
                    psd_zgivens(H1(j, j - 1), H1(j + 1, j - 1), c, s, r);
                    psd_zgg_set2(H1, j, j - 1, r, j + 1, j - 1, z0);
                    psd_zgg_left(H1, j, c, s, j, ilastm);
                    j += 1;
                } else {
                    psd_zgg_link(psd_zgfac(P, n, ln), j - 1, psd_zgsig(P, ln), c, s, ifirstm, ilastm);
                }
                psd_zgg_z(P, st, ln, j - 1, c, s);
                ln = (ln == 1) ? p : (ln - 1);
            }
            psd_zgg_left(Hd, j - 1, c, s, j, ilastm);
        }
        const int j = jlo;
        psd_zgivens(H1(j, j), H1(j + 1, j), c, s, r);
        psd_zgg_set2(H1, j, j, r, j + 1, j, z0);
        psd_zgg_left(H1, j, c, s, j + 1, ilastm);
        psd_zgg_z(P, st, 1, j, c, s);
        for (int l = p; l >= ldeflate + 1; --l) {
            psd_zgg_link(psd_zgfac(P, n, l), j, psd_zgsig(P, l), c, s, ifirstm, ilastm);
            psd_zgg_z(P, st, l, j, c, s);
        }
        psd_zgg_left(Hd, j, c, s, j + 1, ilastm);
    }
}

// One position of a downward chain inside the window: rotation at (j, j+1) through H_1 (rows), H_p .. H_2, and the
// columns of H_1 (rows ifirstm..hmax).  mode 0: generated from column j-1 of H_1 (sweep, :812-816); mode 1: given
// (first position of a sweep); mode 2: generated from `side` = column hj of A_1 (signed Hessenberg, stage 2).
PSD_D void psd_zgq_chain(const psd_zgparams& P, const psd_zgstate& st, const psd_zwin& w, int* lcnt, int j, int mode,
                         double c, psd_z s, psd_z* side, int hmax) {
    const int p = st.p;
    const psd_z z0 = zmk(0.0, 0.0);
    psd_z r;
    if (mode == 0) {
        psd_zgivens(w.at(1, j, j - 1), w.at(1, j + 1, j - 1), c, s, r);
        psd_zgwin_set2(w, 1, j, j - 1, r, j + 1, j - 1, z0);
    } else if (mode == 2) {
        psd_zgivens(side[j - w.bs], side[j + 1 - w.bs], c, s, r);
        PSD_WAVE_SYNC();
        PSD_ONE {
            side[j - w.bs] = r;
            side[j + 1 - w.bs] = z0;
        }
        PSD_WAVE_SYNC();
    }
    psd_zwin_left(w, 1, j, c, s, (mode == 2) ? w.bs : j, st.ilastm);
    psd_zgrecord(P, lcnt, 1, j, c, s);
    for (int l = p; l >= 2; --l) {
        psd_zg_link(w, l, j, psd_zgsig(P, l), c, s, st.ifirstm, st.ilastm);
        psd_zgrecord(P, lcnt, l, j, c, s);
    }
    psd_zwin_right(w, 1, j, c, s, st.ifirstm, hmax);
}

// generalized.jl:808-852: one window of the single-shift sweep
PSD_D void psd_zgq_sweep_window(const psd_zgparams& P, psd_zgstate& st, psd_z* ldsz, int* lcnt) {
    const int n = st.n, p = st.p, ifirst = st.ifirst, ilast = st.ilast, ifirstm = st.ifirstm, ilastm = st.ilastm;
    const int nb = (st.cursor > 0 && st.cfirst > 0 && st.kcur == st.ifirst) ? st.cfirst : (st.W - 3);  // (a cursor's first window: the part inside the block)
    const int ks = st.kcur;
    const int ke = (ks + nb - 1 < ilast - 1) ? (ks + nb - 1) : (ilast - 1);
    psd_zwin w;
    w.b = ldsz;
    w.W = st.W;
    w.ld = st.W + 1;
    w.bsz = st.W * (st.W + 1);
    w.bs = (ks > ifirst) ? (ks - 1) : ifirst;
    w.be = (ke + 2 < ilast) ? (ke + 2) : ilast;
    PSD_PAR_FOR(m, p) { lcnt[m] = 0; }
    psd_zgwin_load(P, w, n, p);
    for (int j = ks; j <= ke; ++j) {
        const int itmp = (j + 2 < ilastm) ? (j + 2) : ilastm;
        psd_zgq_chain(P, st, w, lcnt, j, (j > ifirst) ? 0 : 1, st.c0, st.s0, nullptr, itmp);
    }
    psd_zgwin_store(P, w, n, p);
    psd_zgdesc_write(P, st, lcnt, w.bs, w.be, w.be + 1, ilastm, ifirstm, w.bs - 1, 0, 0, 0, 0);
    st.nwindows += 1;
    st.kcur = ke + 1;
    if (ke >= ilast - 1) st.phase = (st.cursor > 0) ? PSD_GPH_CDONE : ((st.train_n > 1) ? PSD_GPH_TWAIT : PSD_GPH_CHECK);
}

// generalized.jl:356-448: one window of the controlled zero shift
PSD_D void psd_zgq_zshift_window(const psd_zgparams& P, psd_zgstate& st, psd_z* ldsz, int* lcnt) {
    const int n = st.n, p = st.p, jlo = st.jlo, ilast = st.ilast, ifirstm = st.ifirstm, ilastm = st.ilastm;
    const int nb = st.W - 2;
    const int ks = st.kcur;
    const int jend = ilast - 1;
    const int ke = (ks + nb - 1 < jend) ? (ks + nb - 1) : jend;
    const psd_z z0 = zmk(0.0, 0.0);
    psd_zwin w;
    w.b = ldsz;
    w.W = st.W;
    w.ld = st.W + 1;
    w.bsz = st.W * (st.W + 1);
    w.bs = ks;
    w.be = ke + 1;
    PSD_PAR_FOR(m, p) { lcnt[m] = 0; }
    psd_zgwin_load(P, w, n, p);
    for (int j = ks; j <= ke; ++j) {
        double c;
        psd_z s, r;
        psd_zgivens(w.at(1, j, j), w.at(1, j + 1, j), c, s, r);
        psd_zgwin_set2(w, 1, j, j, r, j + 1, j, z0);
        psd_zwin_left(w, 1, j, c, s, j + 1, ilastm);
        psd_zgrecord(P, lcnt, 1, j, c, s);
        for (int l = p; l >= 2 && !ziszero(s); --l) {
            const bool sg = psd_zgsig(P, l);
            if (sg) psd_zwin_right(w, l, j, c, s, ifirstm, j + 1);
            else psd_zwin_left(w, l, j, c, s, j, ilastm);
            double tol = zabs(w.at(l, j, j)) + zabs(w.at(l, j + 1, j + 1));
            if (tol == 0) {
                for (int cc = w.bs; cc <= j + 1; ++cc) {
                    double cs = 0.0;
                    for (int rr = w.bs; rr <= j + 1; ++rr) cs += zabs(w.at(l, rr, cc));
                    tol = fmax(tol, cs);
                }
            }
            tol = fmax(st.ulp * tol, st.smlnum);
            const psd_z sub = w.at(l, j + 1, j);
            if (zabs(sub) <= tol) {
                c = 1.0;
                s = z0;
                psd_zgwin_set2(w, l, j + 1, j, z0, j + 1, j, z0);
            } else if (sg) {
                psd_zgivens(w.at(l, j, j), sub, c, s, r);
                psd_zgwin_set2(w, l, j, j, r, j + 1, j, z0);
                psd_zwin_left(w, l, j, c, s, j + 1, ilastm);
                psd_zgrecord(P, lcnt, l, j, c, s);
            } else {
                psd_zgivens(w.at(l, j + 1, j + 1), zneg(sub), c, s, r);
                psd_zgwin_set2(w, l, j + 1, j + 1, r, j + 1, j, z0);
                psd_zwin_right(w, l, j, c, s, ifirstm, j);
                psd_zgrecord(P, lcnt, l, j, c, s);
            }
        }
        PSD_ONE {
            psd_ztr tr;
            tr.pos = j;
            tr.pad = 0;
            tr.c = c;
            tr.s = s;
            P.dG[j] = tr;
        }
        if (ziszero(s)) st.zflag = 1;
    }
    psd_zgwin_store(P, w, n, p);
    const bool last = ke >= jend;
    psd_zgdesc_write(P, st, lcnt, ks, ke + 1, w.be + 1, ilastm, ifirstm, w.bs - 1, 1, last ? 1 : 0, jlo, jend);
    st.nwindows += 1;
    st.kcur = ke + 1;
    if (last) {
        st.ziter = st.zflag ? 1 : 0;
        st.phase = PSD_GPH_CHECK;
    }
}

// generalized.jl:1034-1079 (ComplexF64): one window of stage 2 of the signed Hessenberg reduction
PSD_D void psd_zgq_hess_window(const psd_zgparams& P, psd_zgstate& st, psd_z* ldsz, psd_z* side, int* lcnt) {
    const int n = st.n, p = st.p, hj = st.hj;
    const int nb = st.W - 1;
    const int qe = st.kcur;
    const int qs = (qe - nb + 1 > hj + 1) ? (qe - nb + 1) : (hj + 1);
    psd_zwin w;
    w.b = ldsz;
    w.W = st.W;
    w.ld = st.W + 1;
    w.bsz = st.W * (st.W + 1);
    w.bs = qs;
    w.be = qe + 1;
    const psd_mat<psd_z> A1 = psd_zgfac(P, n, 1);
    PSD_PAR_FOR(m, p) { lcnt[m] = 0; }
    PSD_PAR_FOR(t, w.be - w.bs + 1) { side[t] = A1(w.bs + t, hj); }
    psd_zgwin_load(P, w, n, p);
    for (int q = qe; q >= qs; --q) psd_zgq_chain(P, st, w, lcnt, q, 2, 0.0, zmk(0.0, 0.0), side, n);
    psd_zgwin_store(P, w, n, p);
    PSD_PAR_FOR(t, w.be - w.bs + 1) { A1(w.bs + t, hj) = side[t]; }
    psd_zgdesc_write(P, st, lcnt, w.bs, w.be, w.be + 1, n, 1, w.bs - 1, 0, 0, 0, 0, 1, hj + 1);
    st.nwindows += 1;
    st.kcur = qs - 1;
    if (st.kcur < hj + 1) {
        st.hj = hj + 1;
        st.kcur = n - 1;
        if (st.hj > n - 2) st.phase = PSD_GPH_DONE;
    }
}

// ------------------------------------------------------------------------------------------------
// Stage 2 as a pipeline over the factors (see psd_gq_hess_step in psd_rgz.h for the argument): wave-scoped copies of
// the window helpers and of the chain link, one beat function, one multi-wave kernel.
namespace psd_wv {
#ifndef PSD_HOSTSIM
#undef PSD_TID
#undef PSD_TSTRIDE
#define PSD_TID PSD_TID_WAVE
#define PSD_TSTRIDE PSD_TSTRIDE_WAVE
#endif
#include "psd_zqz_win.inl"
#define PSD_NS psd_wv::
#include "psd_zgz_chain.inl"
#undef PSD_NS

// one beat of wave g; mail: [(G + 1)][2 parities][c, s.re, s.im, -]
PSD_D void psd_zghess_beat(const psd_zgparams& P, const psd_zgstate& st, const psd_zwin& w, psd_z* side, double* mail,
                           int* lcnt, int g, int G, int b, int K, int L, int qe) {
    const int p = st.p;
    const int n0 = psd_ghess_n0(L, p);
    const psd_z z0 = zmk(0.0, 0.0);
    if (g == 0) {
        const int kc = b - G;
        if (kc >= 0 && kc < K) {
            const double* m = mail + (size_t)(G * 2 + (kc & 1)) * 4;
            psd_wv::psd_zwin_right(w, 1, qe - kc, m[0], zmk(m[1], m[2]), st.ifirstm, st.n);
        }
        if (b < K) {
            const int q = qe - b;
            double c;
            psd_z s, r;
            psd_zgivens(side[q - w.bs], side[q + 1 - w.bs], c, s, r);
            PSD_WAVE_SYNC();
            PSD_ONE {
                side[q - w.bs] = r;
                side[q + 1 - w.bs] = z0;
            }
            PSD_WAVE_SYNC();
            psd_wv::psd_zwin_left(w, 1, q, c, s, w.bs, st.ilastm);
            psd_wv::psd_zgrecord(P, lcnt, 1, q, c, s);
            for (int i = 0; i < n0; ++i) {
                const int l = p - i;
                psd_wv::psd_zg_link(w, l, q, psd_zgsig(P, l), c, s, st.ifirstm, st.ilastm);
                psd_wv::psd_zgrecord(P, lcnt, l, q, c, s);
            }
            double* m = mail + (size_t)(1 * 2 + (b & 1)) * 4;
            PSD_ONE {
                m[0] = c;
                m[1] = s.re;
                m[2] = s.im;
            }
            PSD_WAVE_SYNC();
        }
    } else {
        const int k = b - g;
        if (k >= 0 && k < K) {
            const double* mi = mail + (size_t)(g * 2 + (k & 1)) * 4;
            double c = mi[0];
            psd_z s = zmk(mi[1], mi[2]);
            const int q = qe - k;
            const int i0 = n0 + (g - 1) * L;
            const int i1 = (i0 + L < p - 1) ? (i0 + L) : (p - 1);
            for (int i = i0; i < i1; ++i) {
                const int l = p - i;
                psd_wv::psd_zg_link(w, l, q, psd_zgsig(P, l), c, s, st.ifirstm, st.ilastm);
                psd_wv::psd_zgrecord(P, lcnt, l, q, c, s);
            }
            double* mo = mail + (size_t)((g + 1) * 2 + (k & 1)) * 4;
            PSD_ONE {
                mo[0] = c;
                mo[1] = s.re;
                mo[2] = s.im;
            }
            PSD_WAVE_SYNC();
        }
    }
}
#ifndef PSD_HOSTSIM
#undef PSD_TID
#undef PSD_TSTRIDE
#define PSD_TID PSD_TID_BLOCK
#define PSD_TSTRIDE PSD_TSTRIDE_BLOCK
#endif
}  // namespace psd_wv

PSD_HD size_t psd_zghess_lds_bytes(int p, int W) {
    size_t b = (size_t)p * W * (W + 1) * 16 + 64 * 16 + (size_t)8 * (PSD_GHESS_MAXWAVES + 1) * 8 + (size_t)p * 4;
    return (b + 15) & ~(size_t)15;
}

PSD_KERNEL_B(64 * PSD_GHESS_MAXWAVES) psd_zgq_hess_step(psd_zgparams P, int L) {
    PSD_LDS_DECL;
    psd_zgstate st = *P.st;
    PSD_ONE { P.desc->active = 0; P.desc->defer_run = 0; }
    if (st.phase != PSD_GPH_HESS) return;
    const int G = PSD_NTHREADS >> 6;
    const int n = st.n, p = st.p, hj = st.hj;
    psd_z* ldsz = (psd_z*)psd_lds;
    psd_z* side = ldsz + (size_t)p * st.W * (st.W + 1);
    double* mail = (double*)(side + 64);
    int* lcnt = (int*)(mail + 8 * (PSD_GHESS_MAXWAVES + 1));
    const int nb = st.W - 1;
    const int qe = st.kcur;
    const int qs = (qe - nb + 1 > hj + 1) ? (qe - nb + 1) : (hj + 1);
    const int K = qe - qs + 1;
    psd_zwin w;
    w.b = ldsz;
    w.W = st.W;
    w.ld = st.W + 1;
    w.bsz = st.W * (st.W + 1);
    w.bs = qs;
    w.be = qe + 1;
    const psd_mat<psd_z> A1 = psd_zgfac(P, n, 1);
    PSD_PAR_FOR(m, p) { lcnt[m] = 0; }
    PSD_PAR_FOR(t, w.be - w.bs + 1) { side[t] = A1(w.bs + t, hj); }
    PSD_WAVES_FOR(g, G) { psd_wv::psd_zgwin_load(P, w, n, p, g, G); }
    for (int b = 0; b < K + G; ++b) {
        PSD_WAVES_FOR(g, G) { psd_wv::psd_zghess_beat(P, st, w, side, mail, lcnt, g, G, b, K, L, qe); }
        PSD_SYNC();
    }
    PSD_WAVES_FOR(g, G) { psd_wv::psd_zgwin_store(P, w, n, p, g, G); }
    PSD_PAR_FOR(t, w.be - w.bs + 1) { A1(w.bs + t, hj) = side[t]; }
    psd_zgdesc_write(P, st, lcnt, w.bs, w.be, w.be + 1, n, 1, w.bs - 1, 0, 0, 0, 0, 1, hj + 1);
    st.nwindows += 1;
    st.kcur = qs - 1;
    if (st.kcur < hj + 1) {
        st.hj = hj + 1;
        st.kcur = n - 1;
        if (st.hj > n - 2) st.phase = PSD_GPH_DONE;
    }
    PSD_SYNC();
    PSD_ONE { *P.st = st; }
}

PSD_D int psd_zgq_scan_diag(const psd_zgparams& P, const psd_zgstate& st, int* redi, int jlo, bool sign) {
    const int n = st.n, p = st.p, ilast = st.ilast;
    const int NT = PSD_NTHREADS;
    const int wd = ilast - jlo + 1;
    PSD_SYNC();
    PSD_PAR_FOR(t, NT) {
        int key = 0x7fffffff;
        for (int q = t; q < (p - 1) * wd; q += NT) {
            const int l = 2 + q / wd, j = jlo + q % wd;
            if (psd_zgsig(P, l) != sign) continue;
            const psd_mat<psd_z> Hl = psd_zgfac(P, n, l);
            double tol;
            if (j == ilast) tol = zabs(Hl(j - 1, j));
            else if (j == jlo) tol = zabs(Hl(j, j + 1));
            else tol = zabs(Hl(j - 1, j)) + zabs(Hl(j, j + 1));
            tol = fmax(st.ulp * tol, st.smlnum);
            if (zabs(Hl(j, j)) <= tol) {
                const int k = l * (n + 2) + (n + 1 - j);
                if (k < key) key = k;
            }
        }
        redi[t] = key;
    }
    PSD_SYNC();
    int key = 0x7fffffff;
    for (int t = 0; t < NT; ++t)
        if (redi[t] < key) key = redi[t];
    PSD_SYNC();
    return key;
}

// generalized.jl:302-449,741-806
// ------------------------------------------------------------------------------------------------
// Explicit shifts for multishift trains of the signed single-shift sweep (see psd_rgz.h): the sweep is a similarity of
// H_1 T in Z_1 space, T = prod_{l=2..p} H_l^{s_l} (upper triangular).
// start rotation for the shift mu: first column of H_1 T - mu I, i.e. (H_1[f,f] - mu / T[f,f], H_1[f+1,f]); mu / T[f,f] by
// successive divisions / multiplications.  false: not finite.
PSD_D bool psd_zgq_start_rot_mu(const psd_zgparams& P, int n, int p, int ifirst, psd_z mu, double& c, psd_z& s) {
    psd_z t = mu;
    for (int l = p; l >= 2; --l) {
        const psd_z d = psd_zgfac(P, n, l)(ifirst, ifirst);
        if (psd_zgsig(P, l)) t = zdiv(t, d);
        else t = zmul(t, d);
    }
    if (!(zabs1(t) < 1e300)) return false;  // (also false for NaN)
    const psd_mat<psd_z> H1 = psd_zgfac(P, n, 1);
    psd_z r;
    psd_zgivens(zsub(H1(ifirst, ifirst), t), H1(ifirst + 1, ifirst), c, s, r);
    return true;
}

// the m shifts of a train: eigenvalues of the trailing m x m block of H_1 T (exact: the blocks carry one extra leading
// row / column for the subdiagonal term); inverted factors by back-substitution, one row per lane.  `work`: LDS,
// psd_zgq_train_elems(p, m) complex elements; shifts to P.tshift, closest to the last diagonal entry first.
PSD_HD size_t psd_zgq_train_elems(int p, int m) {
    const size_t K1 = (size_t)m + 1;
    return (size_t)p * K1 * K1 + 2 * K1 * K1 + (size_t)m * m + PSD_ZHQR_MAX + 8;
}
PSD_D void psd_zgq_train_shifts(const psd_zgparams& P, int n, int p, int ilast, int m, psd_z* work, int* okf) {
    const int K = m, K1 = K + 1, t0 = ilast - K + 1, KK = K1 * K1;
    psd_z* B = work;
    psd_z* R0 = B + (size_t)p * KK;
    psd_z* R1 = R0 + KK;
    psd_z* T = R1 + KK;
    psd_z* w = T + K * K;
    PSD_SYNC();
    PSD_PAR_FOR(t, p * KK) {
        const int j = t / KK, q = t - j * KK, r = q / K1, c = q - r * K1;
        B[t] = psd_zgfac(P, n, j + 1)(t0 - 1 + r, t0 - 1 + c);
    }
    PSD_PAR_FOR(q, KK) { R0[q] = (q / K1 == q % K1) ? zmk(1.0, 0.0) : zmk(0.0, 0.0); }
    PSD_SYNC();
    psd_z* cur = R0;
    psd_z* nxt = R1;
    for (int j = 2; j <= p; ++j) {
        const psd_z* Bj = B + (size_t)(j - 1) * KK;
        if (psd_zgsig(P, j)) {
            PSD_PAR_FOR(q, KK) {
                const int r = q / K1, c = q - r * K1;
                psd_z acc = zmk(0.0, 0.0);
                for (int k = r; k <= c; ++k) acc = zadd(acc, zmul(cur[r * K1 + k], Bj[k * K1 + c]));
                nxt[q] = acc;
            }
        } else {  // X B_j = cur, row by row
            PSD_PAR_FOR(r, K1) {
                for (int c = 0; c < K1; ++c) {
                    psd_z x = zmk(0.0, 0.0);
                    if (c >= r) {
                        x = cur[r * K1 + c];
                        for (int k = r; k < c; ++k) x = zsub(x, zmul(nxt[r * K1 + k], Bj[k * K1 + c]));
                        x = zdiv(x, Bj[c * K1 + c]);
                    }
                    nxt[r * K1 + c] = x;
                }
            }
        }
        PSD_SYNC();
        psd_z* sw = cur;
        cur = nxt;
        nxt = sw;
    }
    PSD_PAR_FOR(q, K * K) {
        const int r = q / K, c = q - r * K;
        psd_z acc = zmk(0.0, 0.0);
        for (int k = r; k <= c + 1; ++k) acc = zadd(acc, zmul(B[(r + 1) * K1 + k], cur[k * K1 + (c + 1)]));
        T[q] = acc;
    }
    PSD_SYNC();
#ifndef PSD_HOSTSIM
    const psd_z lastw = T[(K - 1) * K + (K - 1)];
    const bool wave1 = PSD_NTHREADS == 64;  // (the workgroup is one wavefront: all of it runs the small QR)
    bool finw = true;
    for (int q = 0; q < K * K; ++q)
        if (!(zabs1(T[q]) < 1e300)) finw = false;
    bool okw = false;
    if (wave1 && finw) okw = psd_zhqr_wave(T, K, K, w, PSD_TID);
#endif
    PSD_ONE {
#ifndef PSD_HOSTSIM
        bool ok = finw;
        const psd_z last = lastw;
        ok = ok && (wave1 ? okw : psd_zhqr(T, K, K, w));
#else
        bool ok = true;
        for (int q = 0; q < K * K; ++q)
            if (!(zabs1(T[q]) < 1e300)) ok = false;
        const psd_z last = T[(K - 1) * K + (K - 1)];
        ok = ok && psd_zhqr(T, K, K, w);
#endif
        for (int a = 0; ok && a < K; ++a)
            if (!(zabs1(w[a]) < 1e300)) ok = false;
        if (ok) {
            for (int a = 1; a < K; ++a) {
                const psd_z x = w[a];
                const double dx = zabs1(zsub(x, last));
                int b = a - 1;
                while (b >= 0 && zabs1(zsub(w[b], last)) > dx) {
                    w[b + 1] = w[b];
                    --b;
                }
                w[b + 1] = x;
            }
            for (int a = 0; a < K; ++a) P.tshift[a] = w[a];
        }
        *okf = ok ? 1 : 0;
    }
    PSD_SYNC();
}

PSD_D void psd_zgq_check(const psd_zgparams& P, psd_zgstate& st, int* redi, psd_z* work) {
    const int n = st.n, p = st.p;
    const int NT = PSD_NTHREADS;
    st.jiter += 1;
    if (st.jiter > st.maxit) {
        st.info = st.ilast;
        st.phase = PSD_GPH_DONE;
        return;
    }
    const psd_mat<psd_z> H1 = psd_zgfac(P, n, 1);
    const int ilast = st.ilast;
    bool split = false;
    int jlo = 1;
    if (ilast == 1) {
        split = true;
    } else {
        PSD_SYNC();
        PSD_PAR_FOR(t, NT) {
            int best = 0;
            for (int j = ilast - t; j >= 2; j -= NT) {
                double tol = zabs(H1(j - 1, j - 1)) + zabs(H1(j, j));
                if (tol == 0) {
                    for (int cc = 1; cc <= j; ++cc) {
                        double cs = 0.0;
                        const int rmax = (cc + 1 < j) ? (cc + 1) : j;
                        for (int rr = 1; rr <= rmax; ++rr) cs += zabs(H1(rr, cc));
                        tol = fmax(tol, cs);
                    }
                }
                tol = fmax(st.ulp * tol, st.smlnum);
                if (zabs(H1(j, j - 1)) <= tol) {
                    best = j;
                    break;
                }
            }
            redi[t] = best;
        }
        PSD_SYNC();
        int jfound = 0;
        for (int t = 0; t < NT; ++t)
            if (redi[t] > jfound) jfound = redi[t];
        PSD_SYNC();
        if (jfound > 0) {
            PSD_ONE { H1(jfound, jfound - 1) = zmk(0.0, 0.0); }
            PSD_SYNC();
            jlo = jfound;
            if (jfound == ilast) split = true;
        }
    }
    if (split) {
        psd_z a;
        double b;
        int sc;
        psd_zg_safeprod(P, n, p, ilast, a, b, sc);
        PSD_ONE {
            P.alpha[ilast - 1] = a;
            P.beta[ilast - 1] = b;
            P.ascale[ilast - 1] = sc;
        }
        st.nsplit += 1;
        st.ilast -= 1;
        if (st.ilast < 1) {
            st.phase = PSD_GPH_DONE;
            return;
        }
        st.iiter = 0;
        if (st.ziter != -1) st.ziter = 0;
        if (!st.wantT) {
            st.ilastm = st.ilast;
            if (st.ifirstm > st.ilast) st.ifirstm = 1;
        }
        return;
    }
    st.jlo = jlo;
    const int key2 = psd_zgq_scan_diag(P, st, redi, jlo, true);
    const int key3 = (key2 == 0x7fffffff) ? psd_zgq_scan_diag(P, st, redi, jlo, false) : 0x7fffffff;
    if (st.ziter >= 7 || st.ziter < 0) {  // test 4 first; a pending zero is found again afterwards
        st.phase = PSD_GPH_ZSHIFT;
        st.kcur = jlo;
        st.zflag = 0;
        st.nzshift += 1;
        psd_zglog(P, st, 4, jlo, ilast);
        return;
    }
    if (key2 != 0x7fffffff || key3 != 0x7fffffff) {
        const int key = (key2 != 0x7fffffff) ? key2 : key3;
        const int l = key / (n + 2), j = (n + 1) - key % (n + 2);
        PSD_ONE { psd_zgfac(P, n, l)(j, j) = zmk(0.0, 0.0); }
        PSD_SYNC();
        if (key2 != 0x7fffffff) psd_zgq_case2(P, st, l, j);
        else psd_zgq_case3(P, st, l, j);
        return;
    }
    // QZ step (:763-806)
    st.ifirst = jlo;
    st.iiter += 1;
    st.ziter += 1;
    if (!st.wantT) st.ifirstm = st.ifirst;
    double c;
    psd_z s, r;
    PSD_SYNC();
    if (st.iiter % 10 == 0) {
        psd_zgivens(zmk(0.35, 0.62), zmk(0.81, 0.27), c, s, r);  // the reference draws rand(T, 2) (:782)
    } else {
        const int ifirst = st.ifirst;
        psd_zgivens(zmk(1.0, 0.0), zmk(1.0, 0.0), c, s, r);
        for (int l = p; l >= 2; --l) {
            const psd_mat<psd_z> Hl = psd_zgfac(P, n, l);
            if (psd_zgsig(P, l)) {
                psd_zgivens(zscal(c, Hl(ifirst, ifirst)), zmul(Hl(ilast, ilast), zconj(s)), c, s, r);
            } else {
                psd_zgivens(zscal(c, Hl(ilast, ilast)), zneg(zmul(Hl(ifirst, ifirst), zconj(s))), c, s, r);
                s = zneg(s);
            }
        }
        psd_zgivens(zsub(zscal(c, H1(ifirst, ifirst)), zmul(H1(ilast, ilast), zconj(s))),
                    zscal(c, H1(ifirst + 1, ifirst)), c, s, r);
    }
    st.train_n = 1;
    if ((st.train_want >= 2 || st.train_want == -2) && P.tshift != nullptr && st.iiter % 10 != 0) {
        // window width of the train by the cost model of psd_rq_shift
        const int w = ilast - st.ifirst + 1;
        int mt = (st.train_want == -2) ? 1 : st.train_want;
        if (mt > PSD_TRAIN_MAX) mt = PSD_TRAIN_MAX;
        if (mt > PSD_ZHQR_MAX) mt = PSD_ZHQR_MAX;
        int nb = st.Wmax - 3, m = 1;
        double best = 1e300;
        for (int nbc = (st.Wmax - 3 < 8) ? ((st.Wmax > 4) ? st.Wmax - 3 : 1) : 8; nbc <= st.Wmax - 3 && mt >= 2; ++nbc) {
            int mc = 1 + (w - nbc) / (nbc + 3);  // (cursors nbc + 3 = W positions apart, as psd_zqz.h; two windows apart before)
            if (mc > mt) mc = mt;
            if (mc < 2) break;
            const double cost = (double)((w + nbc - 1) / nbc + ((mc - 1) * (nbc + 3) + nbc - 1) / nbc) * (double)(nbc * p + st.train_oc) / mc;
            if (cost < best) {
                best = cost;
                nb = nbc;
                m = mc;
            }
        }
        while (m >= 1 && psd_zgq_train_elems(p, m) > (size_t)p * st.Wmax * (st.Wmax + 1)) --m;
        if ((m >= 2 || st.train_want == -2) && m >= 1 && m + 2 <= w) {
            int* okf = (int*)(P.tshift + PSD_TRAIN_MAX);
            psd_zgq_train_shifts(P, n, p, ilast, m, work, okf);
            double cm;
            psd_z sm;
            if (*okf && psd_zgq_start_rot_mu(P, n, p, st.ifirst, P.tshift[0], cm, sm)) {
                c = cm;
                s = sm;
                if (m >= 2) {
                    st.W = nb + 3;
                    st.train_n = m;
                    st.train_tick0 = P.tick;
                    st.train_id += 1;
                    st.ntrainsweeps += m;
                }
            }
            PSD_SYNC();
        }
    }
    st.c0 = c;
    st.s0 = s;
    st.phase = PSD_GPH_SWEEP;
    st.kcur = st.ifirst;
    st.nsweeps += 1;
    psd_zglog(P, st, 0, st.ifirst, ilast);
    if (st.train_n > 1) {
        for (int b = 1; b < st.train_n; ++b) psd_zglog(P, st, 0, st.ifirst, ilast);
        PSD_SYNC();
        PSD_ONE {
            psd_atomic_store(P.cep + PSD_TRAIN_MAX, 0);  // finished cursors of this train
            for (int b = 1; b < st.train_n; ++b) {
                psd_zgstate cs = st;
                cs.cursor = b;
                {
                    const int nbw = st.W - 3, spc = st.W;
                    const int d = (b * spc - nbw + 1 + nbw - 1) / nbw;  // ceil((b W - nb + 1) / nb) >= 1
                    cs.cstart = st.train_tick0 + d;
                    cs.cfirst = d * nbw - b * spc + nbw;                 // 1 .. nb positions
                }
                cs.phase = PSD_GPH_CWAIT;
                cs.sh = P.tshift[b];
                cs.kcur = 0;
                cs.nsweeps = cs.nwindows = cs.nlog = 0;
                cs.maxlog = 0;
                for (int q = 0; q < 6; ++q) cs.cyc[q] = 0;
                psd_pub_begin(P.cep + b);
                P.cst[b] = cs;
                psd_pub_end(P.cep + b, P.tick);
            }
        }
        PSD_SYNC();
    }
}

PSD_D void psd_zgq_step_body(const psd_zgparams& P) {
    PSD_LDS_DECL;
    psd_zgstate st = *P.st;
    if (st.phase == PSD_GPH_DONE) {
        PSD_ONE { P.desc->active = 0; P.desc->defer_run = 0; }
        return;
    }
    const int NT = PSD_NTHREADS;
    psd_z* ldsz = (psd_z*)psd_lds;
    const size_t winb = (size_t)st.p * st.Wmax * (st.Wmax + 1);
    psd_z* side = ldsz + winb;  // NT/2 complex = NT doubles
    int* redi = (int*)((double*)side + NT);
    int* lcnt = redi + 2 * NT;
    PSD_ONE { P.desc->active = 0; P.desc->defer_run = 0; }
    const long long tk0 = psd_clock(), tw0 = psd_wallclock();
    bool emitted = false;
    int guard = 0;
    while (!emitted && st.phase != PSD_GPH_DONE && guard < 64) {
        ++guard;
        if (st.phase == PSD_GPH_CHECK) {
            const int nc = st.ncase2 + st.ncase3;
            psd_zgq_check(P, st, redi, ldsz);
            if (st.ncase2 + st.ncase3 != nc) break;
        } else if (st.phase == PSD_GPH_SWEEP) {
            psd_zgq_sweep_window(P, st, ldsz, lcnt);
            emitted = true;
        } else if (st.phase == PSD_GPH_ZSHIFT) {
            psd_zgq_zshift_window(P, st, ldsz, lcnt);
            emitted = true;
        } else if (st.phase == PSD_GPH_HESS) {
            psd_zgq_hess_window(P, st, ldsz, side, lcnt);
            emitted = true;
        } else if (st.phase == PSD_GPH_TWAIT) {  // the leader's sweep is done: wait for the cursors of the train
            bool all = true;
            // (a cursor counts itself in after its last store; its slot's state is then complete)
            all = psd_atomic_load(P.cep + PSD_TRAIN_MAX) == st.train_n - 1;
            if (all) psd_acquire_fence();
            if (all) {
                for (int b = 1; b < st.train_n; ++b) {
                    st.nwindows += P.cst[b].nwindows;
                    st.nsweeps += 1;
                }
                st.train_n = 1;
                st.W = st.Wmax;
                st.phase = PSD_GPH_CHECK;  // (runs in the next launch, behind the cursors' last bulk updates)
            }
            emitted = true;
        } else {
            st.phase = PSD_GPH_DONE;
        }
    }
    st.cyc[4] += psd_clock() - tk0;
    st.cyc[5] += psd_wallclock() - tw0;
    if (st.info == PSD_LIST_OVERFLOW) st.phase = PSD_GPH_DONE;  // (a window that overran a list ends the call)
    PSD_SYNC();
    PSD_ONE { *P.st = st; }
}

PSD_KERNEL_B(PSD_STEP_NT) psd_zgq_step(psd_zgparams P) { psd_zgq_step_body(P); }

// cursor b of a multishift train (see psd_gq_cursor_body)
PSD_D void psd_zgq_cursor_body(const psd_zgparams& P, int b) {
    PSD_LDS_DECL;
    PSD_ONE { P.desc->active = 0; P.desc->defer_run = 0; }
    psd_zgstate st;
    if (!psd_pub_read(P.cep + b, P.tick, P.st, st)) return;  // (published in an earlier launch, not being rewritten)
    if (st.cursor != b) return;
    if (st.phase != PSD_GPH_CWAIT && st.phase != PSD_GPH_SWEEP) return;
    const int NT = PSD_NTHREADS;
    psd_z* ldsz = (psd_z*)psd_lds;
    const size_t winb = (size_t)st.p * st.Wmax * (st.Wmax + 1);
    int* lcnt = (int*)((double*)(ldsz + winb) + NT) + 2 * NT;
    if (st.phase == PSD_GPH_CWAIT) {
        if (P.tick < st.cstart) return;
        double c;
        psd_z s;
        if (!psd_zgq_start_rot_mu(P, st.n, st.p, st.ifirst, st.sh, c, s)) {
            st.phase = PSD_GPH_CDONE;  // (not finite: this bulge is dropped)
            PSD_SYNC();
            PSD_ONE {
                *P.st = st;
                psd_release_fence();
                psd_atomic_add(P.cep + PSD_TRAIN_MAX, 1);
            }
            return;
        }
        st.c0 = c;
        st.s0 = s;
        st.kcur = st.ifirst;
        st.phase = PSD_GPH_SWEEP;
    }
    psd_zgq_sweep_window(P, st, ldsz, lcnt);
    PSD_SYNC();
    PSD_ONE {
        *P.st = st;
        if (st.phase == PSD_GPH_CDONE) {  // last window: count this cursor in (its state and lists are out first)
            psd_release_fence();
            psd_atomic_add(P.cep + PSD_TRAIN_MAX, 1);
        }
    }
}

PSD_KERNEL_B(PSD_STEP_NT) psd_zgq_step_train(psd_zgparams P, int p, int cstride) {
    const int b = PSD_BLOCK_X;
    if (b == 0) {
        psd_zgq_step_body(P);
        return;
    }
    psd_zgparams Q = P;
    Q.st = P.cst + b;
    Q.desc = P.desc + b;
    Q.cnt = P.cnt + (size_t)b * cstride;
    Q.tr = P.tr + (size_t)b * p * PSD_GTR_CAP;
    psd_zgq_cursor_body(Q, b);
}

// Bulk application of one window's rotation lists, by factor: grid = (tiles, p factors, 3 roles); tiles 64 wide
PSD_D void psd_zgq_apply_body(const psd_zgparams& P, int n, int p, int role) {
    PSD_LDS_DECL;
    const psd_gapply_desc d = *P.desc;
    if (!d.active) return;
    const int l = PSD_BLOCK_Y + 1;
    const int own = (role == 0) ? psd_zgrowner(P, l, p) : (role == 1) ? psd_zgcowner(P, l, p) : l;
    const int cnt = P.cnt[own - 1] < PSD_GTR_CAP ? P.cnt[own - 1] : PSD_GTR_CAP;
    if (cnt <= 0) return;
    const int T = PSD_ZAPPLY_NT;
    const int S = d.phi - d.plo + 1;
    psd_ztr* ltr = (psd_ztr*)psd_lds;
    psd_z* tile = (psd_z*)(psd_lds + sizeof(psd_ztr) * PSD_GTR_CAP);
    const bool h1x = d.h1mode == 1 && l == 1;
    if (role == 0) {
        const int c0 = (h1x ? d.h1c0 : d.lc0) + PSD_BLOCK_X * T;
        if (c0 > d.lc1) return;
        const int nc = (d.lc1 - c0 + 1 < T) ? (d.lc1 - c0 + 1) : T;
        if (h1x && c0 >= d.plo && c0 + nc - 1 <= d.phi) return;
        const psd_mat<psd_z> M = psd_zgfac(P, n, l);
        const int ldt = T + 1;
        PSD_PAR_FOR(e, cnt) { ltr[e] = P.tr[(size_t)(own - 1) * PSD_GTR_CAP + e]; }
        PSD_PAR_FOR(t, 32 * nc) {  // (S <= 32; no index divisions)
            const int r = t & 31, c = t >> 5;
            if (r >= S) continue;
            tile[r * ldt + c] = M(d.plo + r, c0 + c);
        }
        PSD_SYNC();
        PSD_PAR_FOR(c, nc) {
            if (h1x && c0 + c >= d.plo && c0 + c <= d.phi) continue;
            for (int e = 0; e < cnt; ++e) {
                const psd_ztr tr = ltr[e];
                const int r = tr.pos - d.plo;
                psd_z a1 = tile[r * ldt + c], a2 = tile[(r + 1) * ldt + c];
                psd_zrot_left(tr.c, tr.s, a1, a2);
                tile[r * ldt + c] = a1;
                tile[(r + 1) * ldt + c] = a2;
            }
        }
        PSD_SYNC();
        PSD_PAR_FOR(t, 32 * nc) {  // (S <= 32; no index divisions)
            const int r = t & 31, c = t >> 5;
            if (r >= S) continue;
            if (h1x && c0 + c >= d.plo && c0 + c <= d.phi) continue;
            M(d.plo + r, c0 + c) = tile[r * ldt + c];
        }
    } else {
        if (role == 1 && d.defer_h1 == 1 && l == 1) return;
        if (role == 2 && (l < P.zlo || l > P.zhi)) return;  // (another rank's Schur vectors)
        const bool h1r = h1x && role == 1;
        const int lo = (role == 1) ? (h1r ? 1 : d.rr0) : d.zr0;
        const int hi = (role == 1) ? (h1r ? n : d.rr1) : d.zr1;
        const int r0 = lo + PSD_BLOCK_X * T;
        if (r0 > hi) return;
        const int nr = (hi - r0 + 1 < T) ? (hi - r0 + 1) : T;
        psd_z* base = (role == 1) ? P.H : P.Z;
        const psd_mat<psd_z> M = psd_mat<psd_z>{base + (size_t)(l - 1) * n * n, n};
        PSD_PAR_FOR(e, cnt) { ltr[e] = P.tr[(size_t)(own - 1) * PSD_GTR_CAP + e]; }
        PSD_PAR_FOR(t, S * T) {
            const int r = t & (T - 1), c = t / T;
            if (r >= nr) continue;
            tile[c * T + r] = M(r0 + r, d.plo + c);
        }
        PSD_SYNC();
        PSD_PAR_FOR(r, nr) {
            if (h1r && r0 + r >= d.plo && r0 + r <= d.phi) continue;
            for (int e = 0; e < cnt; ++e) {
                const psd_ztr tr = ltr[e];
                const int c = tr.pos - d.plo;
                psd_z a1 = tile[c * T + r], a2 = tile[(c + 1) * T + r];
                psd_zrot_right_adj(tr.c, tr.s, a1, a2);
                tile[c * T + r] = a1;
                tile[(c + 1) * T + r] = a2;
            }
        }
        PSD_SYNC();
        PSD_PAR_FOR(t, S * T) {
            const int r = t & (T - 1), c = t / T;
            if (r >= nr) continue;
            if (h1r && r0 + r >= d.plo && r0 + r <= d.phi) continue;
            M(r0 + r, d.plo + c) = tile[c * T + r];
        }
    }
}

PSD_KERNEL_B(PSD_ZAPPLY_NT) psd_zgq_apply(psd_zgparams P, int n, int p) { psd_zgq_apply_body(P, n, p, PSD_BLOCK_Z); }

// bulk updates of all cursors of a tick: pass 0 = rows and Z roles (grid.z = 2 M), pass 1 = columns role (grid.z = M)
PSD_KERNEL_B(PSD_ZAPPLY_NT) psd_zgq_apply_train(psd_zgparams P, int n, int p, int cstride, int pass) {
    const int z = PSD_BLOCK_Z;
    const int b = (pass == 0) ? (z >> 1) : z;
    const int role = (pass == 0) ? ((z & 1) ? 2 : 0) : 1;
    psd_zgparams Q = P;
    Q.desc = P.desc + b;
    Q.cnt = P.cnt + (size_t)b * cstride;
    Q.tr = P.tr + (size_t)b * p * PSD_GTR_CAP;
    psd_zgq_apply_body(Q, n, p, role);
}

// Deferred right side of H_1 after a zero-shift pass (generalized.jl:436-444)
PSD_KERNEL psd_zgq_defer(psd_zgparams P, int n) {
    const psd_gapply_desc d = *P.desc;
    if (!d.active || d.defer_run != 1) return;
    const psd_mat<psd_z> H1 = psd_mat<psd_z>{P.H, n};
    const int NT = PSD_NTHREADS;
    const int rbase = d.drow0 + PSD_BLOCK_X * NT;
    PSD_PAR_FOR(t, NT) {
        const int r = rbase + t;
        if (r <= d.djhi + 1 && d.djhi >= d.djlo) {
            int j = (r - 1 > d.djlo) ? (r - 1) : d.djlo;
            psd_z a1 = H1(r, j);
            for (; j <= d.djhi; ++j) {
                const psd_ztr g = P.dG[j];
                psd_z a2 = H1(r, j + 1);
                psd_zrot_right_adj(g.c, g.s, a1, a2);
                H1(r, j) = a1;
                a1 = a2;
            }
            H1(r, d.djhi + 1) = a1;
        }
    }
}

PSD_KERNEL psd_zgq_init(psd_zgparams P, int n, int p, int wantT, int wantZ, int W, int maxitfac, int maxlog,
                        int hessmode, int train_want, int train_oc) {
    const psd_mat<psd_z> H1 = psd_mat<psd_z>{P.H, n};
    if (!hessmode) PSD_PAR_FOR(c, n) {
        for (int r = c + 3; r <= n; ++r) H1(r, c + 1) = zmk(0.0, 0.0);  // _gethess!
    }
    PSD_ONE {
        psd_zgstate st;
        st.n = n; st.p = p; st.wantT = wantT; st.wantZ = wantZ; st.W = st.Wmax = W; st.train_oc = train_oc;
        st.phase = PSD_GPH_CHECK; st.info = 0;
        st.ilast = n; st.ifirst = 1; st.ifirstm = 1; st.ilastm = n; st.iiter = 1;
        st.ziter = (p >= 20) ? -1 : 0;
        st.jiter = 0; st.maxit = maxitfac * n;
        st.jlo = 1; st.kcur = 0; st.zflag = 0; st.hj = 0;
        st.nsweeps = st.nzshift = st.nsplit = st.ncase2 = st.ncase3 = st.nwindows = st.nlog = 0;
        st.maxlog = maxlog;
        st.c0 = 1.0; st.s0 = zmk(0.0, 0.0);
        st.train_want = hessmode ? 0 : train_want; st.train_n = 1; st.train_id = 0; st.cursor = 0; st.train_tick0 = 0; st.cstart = 0; st.cfirst = 0;
        st.ntrainsweeps = 0; st.sh = zmk(0.0, 0.0);
        st.ulp = PSD_DBL_EPS;
        st.safmin = PSD_DBL_MIN;
        st.smlnum = PSD_DBL_MIN * ((double)n / PSD_DBL_EPS);
        for (int q = 0; q < 6; ++q) st.cyc[q] = 0;
        if (n == 0) st.phase = PSD_GPH_DONE;
        if (hessmode) {
            st.phase = (n >= 3) ? PSD_GPH_HESS : PSD_GPH_DONE;
            st.hj = 1;
            st.kcur = n - 1;
            st.wantT = 1;
        }
        *P.st = st;
        P.desc->active = 0;
        P.desc->defer_run = 0;
    }
}

// generalized.jl:860-908 with a signature, factor l (launched for l = p..2): diag(T_l) real >= 0; the phases go into
// row j (S[l]) or column j (!S[l]) of T_l, column j of Z_l and column j (S[l-1]) or row j (!S[l-1]) of T_{l-1}.
// grid = n blocks (one per j)
PSD_KERNEL psd_zgq_phase(psd_zgparams P, int n, int l, int wantZ) {
    PSD_LDS_DECL;
    psd_z* zs = (psd_z*)psd_lds;
    const int j = PSD_BLOCK_X + 1;
    const psd_mat<psd_z> Hl = psd_zgfac(P, n, l);
    const psd_mat<psd_z> Hm = psd_zgfac(P, n, l - 1);
    const bool sl = psd_zgsig(P, l), sm = psd_zgsig(P, l - 1);
    PSD_ONE {
        const psd_z d = Hl(j, j);
        const double abst = zabs(d);
        psd_z z = zmk(1.0, 0.0);
        if (abst > PSD_DBL_MIN) {
            z = zconj(zmk(d.re / abst, d.im / abst));
            Hl(j, j) = zmk(abst, 0.0);
        }
        zs[0] = z;
    }
    PSD_SYNC();
    const psd_z z = zs[0];
    if (z.re == 1.0 && z.im == 0.0) return;
    // sf[j] = z (S[l]) or conj(z) (!S[l]);  Z_l[:, j] *= conj(sf);  T_{l-1}: column j *= conj(sf) (S) / row j *= sf (!S)
    const psd_z sf = sl ? z : zconj(z);
    const psd_z sfc = zconj(sf);
    if (sl) {
        PSD_PAR_FOR(t, n - j) { Hl(j, j + 1 + t) = zmul(Hl(j, j + 1 + t), z); }
    } else {
        PSD_PAR_FOR(t, j - 1) { Hl(t + 1, j) = zmul(Hl(t + 1, j), z); }
    }
    if (wantZ && l >= P.zlo && l <= P.zhi) {
        const psd_mat<psd_z> Zl = psd_mat<psd_z>{P.Z + (size_t)(l - 1) * n * n, n};
        PSD_PAR_FOR(r, n) { Zl(r + 1, j) = zmul(Zl(r + 1, j), sfc); }
    }
    if (sm) {
        PSD_PAR_FOR(r, j) { Hm(r + 1, j) = zmul(Hm(r + 1, j), sfc); }
    } else {
        PSD_PAR_FOR(t, n - j + 1) { Hm(j, j + t) = zmul(Hm(j, j + t), sf); }
    }
}

// ---- stage 1 helpers of the complex signed Hessenberg reduction ------------------------------------------------------
// in place B(r, c) = conj(A(n+1-c, n+1-r)).  grid = n (columns)
PSD_KERNEL psd_zantitranspose(psd_z* A, int n) {
    const psd_mat<psd_z> M = psd_mat<psd_z>{A, n};
    const int c = PSD_BLOCK_X + 1;
    PSD_PAR_FOR(t, n) {
        const int r = t + 1;
        const int r2 = n + 1 - c, c2 = n + 1 - r;
        if (r + c < n + 1) {
            const psd_z x = M(r, c);
            M(r, c) = zconj(M(r2, c2));
            M(r2, c2) = zconj(x);
        } else if (r + c == n + 1) {
            M(r, c) = zconj(M(r, c));
        }
    }
}
PSD_KERNEL psd_zflip(psd_z* A, int n, int rows) {
    const psd_mat<psd_z> M = psd_mat<psd_z>{A, n};
    const int k = PSD_BLOCK_X + 1;
    if (rows == 0) {
        if (k > n / 2) return;
        PSD_PAR_FOR(t, n) {
            const psd_z x = M(t + 1, k);
            M(t + 1, k) = M(t + 1, n + 1 - k);
            M(t + 1, n + 1 - k) = x;
        }
    } else {
        PSD_PAR_FOR(t, n / 2) {
            const psd_z x = M(t + 1, k);
            M(t + 1, k) = M(n - t, k);
            M(n - t, k) = x;
        }
    }
}
PSD_KERNEL psd_ztril_zero(psd_z* A, int n) {
    const psd_mat<psd_z> M = psd_mat<psd_z>{A, n};
    const int c = PSD_BLOCK_X + 1;
    PSD_PAR_FOR(t, n) {
        if (t + 1 > c) M(t + 1, c) = zmk(0.0, 0.0);
    }
}
