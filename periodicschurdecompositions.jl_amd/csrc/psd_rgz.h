// Real generalized (signed) periodic QZ iteration on the GPU: device-resident state machine.
//
// Replaces pschur!(H1, Hs, S; wantT, wantZ, Q, maxitfac) for Float64 —
// /root/reference/src/rgeneralized.jl:49-1083 (after SLICOT MB03BD): Z_l' H_l Z_{l+1} = T_l for S[l], and
// Z_{l+1}' H_l Z_l = T_l for !S[l]; T_1 quasi-triangular, the others upper triangular.
//
// MI355X structure as in psd_real_qr.h / psd_zqz.h: one wavefront chases a diagonal window of all p factors in
// LDS and emits one rotation list per owner (the orthogonal factor Z_m a rotation belongs to); a wide kernel applies
// the lists to the off-window rows/columns.  With a signature the side an owner acts on depends on the factor:
//     rows    of H_l  <- owner  l      if S[l]  else  l+1
//     columns of H_l  <- owner  l+1    if S[l]  else  l        (cyclic)
// so the bulk kernel is organised by factor (rows role, columns role, Z role).
//
// Implemented: deflation tests 1-3 (:192-226), controlled zero shift (test 4, :229-324), Case II (:329-442) and
// Case III (:444-616), 1x1 split (:617-642), 2x2 blocks (:661-790: real single-shift PQZ `_rp2x2ssr!` with
// perfect-shift deflation, else the conjugate pair by `_rpeigvals2x2`), implicit double-shift sweep with the
// `_qzrots` starting rotations (:796-803, :890-1054).  The reference's explicit-shift branch (:804-887) is defective
//  and is not reproduced; every sweep uses the implicit shift.  Cases II/III (exactly singular
// factors) are rare and run directly on HBM from the step kernel instead of through windows.
#pragma once
#include "psd_real_qr.h"
#include "psd_zqz.h"

enum {
    PSD_GPH_CHECK = 0, PSD_GPH_SWEEP = 1, PSD_GPH_ZSHIFT = 2, PSD_GPH_HESS = 3, PSD_GPH_DONE = 7,
    PSD_GPH_TWAIT = 8, PSD_GPH_CWAIT = 9, PSD_GPH_CDONE = 10  // multishift trains, as in psd_real_qr.h
};
#define PSD_GTR_CAP 80  // rotations per owner and window
#define PSD_GAPPLY_NT 128

struct psd_gtr {  // Givens rotation on (pos, pos+1): [c s; -s c]
    int pos, pad;
    double c, s;
};

struct psd_gapply_desc {
    int active;
    int plo, phi;
    int lc0, lc1;  // rows role: columns lc0..lc1
    int rr0, rr1;  // columns role: rows rr0..rr1
    int zr0, zr1;  // Z role
    int defer_h1;  // 1: the column updates of H_1 are deferred to the end of the (zero-shift) pass
    int defer_run;
    int djlo, djhi, drow0;
    int h1mode;  // 1 (signed Hessenberg stage 2): H_1 takes its row list on columns h1c0..lc1 and its column list on
    int h1c0;    //   rows 1..n, both outside the window plo..phi
};

struct psd_gstate {
    int n, p, wantT, wantZ, W;
    int phase, info;
    int ilast, ifirst, ifirstm, ilastm, ziter, jiter, maxit;
    int jlo, kcur, zflag, hj;
    int nsweeps, nzshift, nsplit, ncase2, ncase3, n2real, n2cplx, nwindows, nlog, maxlog, iwarn;
    double c1, s1, c2, s2;  // starting rotations of the current sweep
    double smlnum, ulp;
    long long cyc[6];
    long long dbg[8];  // cycles of the check's stages (PSD_GDBG: test 1, tests 2/3, start rotations, train shifts, explicit start, cursor states, 2x2 block, split)
    int dbgn[8];
    // multishift train (as psd_rstate): bulges wanted (-2: explicit-shift start without a train, test hook) / in the
    // running train / train number / this state's cursor / tick of the leader's first window / sweeps in trains
    int train_want, train_n, train_id, cursor, train_tick0, ntrainsweeps;
    int cstart, cfirst;  // cursor: the tick of its first window and that window's number of positions (cursors W positions apart)
    int Wmax, train_oc;  // LDS layout width (W is the running sweep's, <= Wmax); o / c of the width rule (psd_rq_shift)
    double sh[4];  // this bulge's shift pair: rt1r, rt1i, rt2r, rt2i
};

struct psd_gparams {
    double* H;                // [p][n][n], H_1 Hessenberg
    double* Z;                // [p][n][n] or nullptr
    const unsigned char* S;   // [p] signature
    psd_gstate* st;
    psd_gapply_desc* desc;
    psd_gtr* tr;   // [p][PSD_GTR_CAP]
    int* cnt;      // [p]
    psd_gtr* dG;   // [n+2]
    psd_z* alpha;  // [n]
    double* beta;  // [n]
    int* ascale;   // [n]
    int* log;
    double* xscr;  // [16 p] scratch of the 2x2 solvers
    psd_gstate* cst;  // [PSD_TRAIN_MAX] cursor states of a train (entry 0 unused) or nullptr
    int* cep;  // [PSD_TRAIN_MAX] epoch words of the cursor states (psd_pub_*), then the count of finished cursors
    double* tshift;   // [PSD_TRAIN_MAX][4] shift pairs, then a flag word
    int tick;         // launch index
    // period sharding (psd_set_shard): the owners m (1-based, inclusive) whose Schur vectors Z_m this context holds;
    // the updates of the others are some other rank's work (1..p without sharding)
    int zlo, zhi;
    // scan form of the sweep windows (psd_gs3_run): byte offsets in LDS of the command block the helper wavefronts watch
    // and of the rotation tables; 0: single-wave sweep (64 x 1 workgroups)
    int gcoff, gtaboff;
};

PSD_HD psd_mat<double> psd_gfac(const psd_gparams& P, int n, int l) {
    return psd_mat<double>{P.H + (size_t)(l - 1) * n * n, n};
}
PSD_HD bool psd_gsig(const psd_gparams& P, int l) { return P.S[l - 1] != 0; }
PSD_HD int psd_gnext(int l, int p) { return (l % p) + 1; }
PSD_HD int psd_growner(const psd_gparams& P, int l, int p) { return psd_gsig(P, l) ? l : psd_gnext(l, p); }
PSD_HD int psd_gcowner(const psd_gparams& P, int l, int p) { return psd_gsig(P, l) ? psd_gnext(l, p) : l; }

struct psd_gwin {
    double* b;
    int W, ld, bsz, bs, be;
    PSD_HD double& at(int l, int r, int c) const { return b[(l - 1) * bsz + (c - bs) * ld + (r - bs)]; }
};

#include "psd_rgz_chain.inl"

PSD_D void psd_grecord(const psd_gparams& P, int* lcnt, int m, int pos, double c, double s) {
    PSD_ONE {
        const int q = lcnt[m - 1];
        if (q < PSD_GTR_CAP) {
            psd_gtr tr;
            tr.pos = pos;
            tr.pad = 0;
            tr.c = c;
            tr.s = s;
            P.tr[(size_t)(m - 1) * PSD_GTR_CAP + q] = tr;
        }
        lcnt[m - 1] = q + 1;
    }
    PSD_WAVE_SYNC();
}

PSD_D void psd_glog(const psd_gparams& P, psd_gstate& st, int kind, int lo, int hi) {
    PSD_ONE {
        if (st.nlog < st.maxlog) {
            P.log[3 * st.nlog + 0] = kind;
            P.log[3 * st.nlog + 1] = lo;
            P.log[3 * st.nlog + 2] = hi;
        }
    }
    st.nlog += 1;
}

PSD_D void psd_gdesc_write(const psd_gparams& P, psd_gstate& st, const int* lcnt, int plo, int phi, int lc0,
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

// index of entry (a, b) (1-based inside the active block of nn rows, m = nn - 1) among the nine staged entries of a factor
// (psd_g_qzrots): the trailing triangle first — for nn = 3 it overlaps the leading one, same values either way
PSD_HD int psd_g_qsidx(int a, int b, int m, int nn) {
    if (a == nn && b == nn) return 8;
    if (a == m && b == nn) return 7;
    if (a == m && b == m) return 6;
    if (a == 1) return b - 1;
    if (a == 2) return 1 + b;
    return 5;
}

// rgeneralized.jl:1140-1359 `_qzrots` (MB03AF 'Double'): starting rotations of an implicit double-shift sweep
// on the active block i1..i1+nb-1, from HBM.
// scr: LDS, 12 p doubles: the entries of the factors the two passes read — the leading 3 x 3 upper triangle and the
// trailing 2 x 2 upper triangle of the active block — staged by one lane per factor (read one after the other from device
// memory inside the chains they cost a memory round trip per factor and pass: 59 us per call at p = 32).
PSD_D void psd_g_qzrots(const psd_gparams& P, int n, int p, int i1, int nb, double& c1o, double& s1o, double& c2o,
                        double& s2o, double* scr) {
    const psd_mat<double> H1 = psd_gfac(P, n, 1);
    PSD_SYNC();
    PSD_PAR_FOR(t, p) {
        const psd_mat<double> Hl = psd_gfac(P, n, t + 1);
        double* q = scr + 12 * t;
        const int e = i1 + nb - 1;
        q[0] = Hl(i1, i1); q[1] = Hl(i1, i1 + 1); q[2] = Hl(i1, i1 + 2);
        q[3] = Hl(i1 + 1, i1 + 1); q[4] = Hl(i1 + 1, i1 + 2); q[5] = Hl(i1 + 2, i1 + 2);
        q[6] = Hl(e - 1, e - 1); q[7] = Hl(e - 1, e); q[8] = Hl(e, e);
    }
    PSD_SYNC();
    double c1, s1, c2, s2, r, al, be, ga, de;
    psd_givens(H1(i1, i1), H1(i1 + 1, i1), c1, s1, r);
    psd_givens(r, 1.0, c2, s2, r);
    const int i2 = i1 + nb - 1;
    for (int l = p; l >= 2; --l) {
        const double* q = scr + 12 * (l - 1);  // 0..5: (1,1) (1,2) (1,3) (2,2) (2,3) (3,3) of the active block; 6..8: (m,m) (m,nn) (nn,nn)
        if (psd_gsig(P, l)) {
            al = c2 * (c1 * q[0] + s1 * q[1]);
            be = s1 * c2 * q[3];
            ga = s2 * q[8];
            psd_givens(al, be, c1, s1, r);
            double v;
            psd_givens(r, ga, c2, s2, v);
        } else {
            al = c1 * s2 * q[0];
            ga = s1 * q[0];
            be = s2 * (c1 * q[1] + s1 * q[3]);
            de = c1 * q[3] - s1 * q[1];
            psd_givens(de, ga, c1, s1, r);
            al = c1 * al + s1 * be;
            be = c2 * q[8];
            psd_givens(be, al, c2, s2, r);
        }
    }
    al = s2 * H1(i2, i2) - c1 * c2;
    be = -s1 * c2;
    const int nn = nb, m = nb - 1;
#define PSD_V1(a, b) H1(i1 - 1 + (a), i1 - 1 + (b))
    ga = -s2 * PSD_V1(nn, m);
    psd_givens(al, ga, c2, s2, r);
    psd_givens(r, be, c1, s1, r);
    const double cx = c1 * c2, sx = c1 * s2;
    be = s1 * PSD_V1(nn, m);
    al = cx * PSD_V1(nn, m) + sx * PSD_V1(nn, nn);
    ga = s1 * PSD_V1(m, m);
    de = cx * PSD_V1(m, m) + sx * PSD_V1(m, nn);
    double val1 = s1 * PSD_V1(3, 2), val2 = cx * PSD_V1(2, 1) + s1 * PSD_V1(2, 2),
           val3 = cx * PSD_V1(1, 1) + s1 * PSD_V1(1, 2);
#undef PSD_V1
    double c3, s3, c4, s4, c5, s5, c6, s6;
    psd_givens(al, be, c1, s1, r);
    psd_givens(ga, r, c2, s2, r);
    psd_givens(de, r, c3, s3, r);
    psd_givens(val1, r, c4, s4, r);
    psd_givens(val2, r, c5, s5, r);
    psd_givens(val3, r, c6, s6, r);
    for (int i = p; i >= 2; --i) {
        const double* q = scr + 12 * (i - 1);
        // (the staged entries by name: PSD_V(a, b) with a, b in 1..3 from the leading triangle, m / nn from the trailing one)
#define PSD_V(a, b) q[psd_g_qsidx((a), (b), m, nn)]
        if (psd_gsig(P, i)) {
            double ss = s3 * s4;
            const double sss = s2 * ss, ssss = s1 * sss;
            val1 = c4 * PSD_V(1, 3);
            val2 = c4 * PSD_V(2, 3);
            val3 = c4 * PSD_V(3, 3);
            al = s4 * c3 * PSD_V(m, m) + sss * c1 * PSD_V(m, nn);
            be = ss * c2 * PSD_V(m, m) + ssss * PSD_V(m, nn);
            ga = sss * c1 * PSD_V(nn, nn);
            de = ssss * PSD_V(nn, nn);
            ss = s5 * s6;
            const double cs = c5 * s6;
            val1 = ss * val1 + cs * PSD_V(1, 2) + c6 * PSD_V(1, 1);
            val2 = ss * val2 + cs * PSD_V(2, 2);
            val3 = ss * val3;
            al = ss * al;
            be = ss * be;
            ga = ss * ga;
            de = ss * de;
            psd_givens(ga, de, c1, s1, r);
            psd_givens(be, r, c2, s2, r);
            psd_givens(al, r, c3, s3, r);
            psd_givens(val3, r, c4, s4, r);
            psd_givens(val2, r, c5, s5, r);
            psd_givens(val1, r, c6, s6, r);
        } else {
            double ep, ze, et, th, val4, val5;
            double c2R, s2R, c3R, s3R, c4R, s4R, c5R, s5R, c6R, s6R;
            de = c1 * PSD_V(nn, nn);
            ep = s1 * PSD_V(nn, nn);
            al = c2 * PSD_V(m, m);
            be = s2 * de;
            ga = -s2 * PSD_V(m, m);
            ze = c2 * PSD_V(m, nn) + s2 * ep;
            et = -s2 * PSD_V(m, nn) + c2 * ep;
            de = c1 * c2 * de + s1 * et;
            psd_givens(de, -ga, c2R, s2R, r);
            de = c3 * PSD_V(m, m);
            ep = s3 * al;
            et = c3 * PSD_V(m, nn) + s3 * be;
            th = s3 * ze;
            ga = -s3 * PSD_V(m, m);
            be = -s3 * PSD_V(m, nn) + c3 * be;
            al = c2R * c3 * al + s2R * (c1 * be + s1 * c3 * ze);
            psd_givens(al, -ga, c3R, s3R, r);
            val1 = c4 * PSD_V(3, 3);
            val2 = s4 * de;
            val3 = s4 * ep;
            val4 = s4 * et;
            val5 = s4 * th;
            be = -s4 * PSD_V(3, 3);
            de = c4 * de;
            ep = c4 * ep;
            ze = c4 * et;
            et = c4 * th;
            al = c3R * de + s3R * (c2R * ep + s2R * (c1 * ze + s1 * et));
            psd_givens(al, -be, c4R, s4R, r);
            be = c5 * PSD_V(2, 2);
            de = c5 * PSD_V(2, 3) + s5 * val1;
            ep = s5 * val2;
            ze = s5 * val3;
            et = s5 * val4;
            th = s5 * val5;
            ga = -s5 * PSD_V(2, 2);
            val1 = c5 * val1 - s5 * PSD_V(2, 3);
            val2 = c5 * val2;
            val3 = c5 * val3;
            val4 = c5 * val4;
            val5 = c5 * val5;
            al = c4R * val1 + s4R * (c3R * val2 + s3R * (c2R * val3 + s2R * (c1 * val4 + s1 * val5)));
            psd_givens(al, -ga, c5R, s5R, r);
            ga = -s6 * PSD_V(1, 1);
            be = c6 * be - s6 * PSD_V(1, 2);
            de = c6 * de - s6 * PSD_V(1, 3);
            ep = c6 * ep;
            ze = c6 * ze;
            et = c6 * et;
            th = c6 * th;
            al = c5R * be + s5R * (c4R * de + s4R * (c3R * ep + s3R * (c2R * ze + s2R * (c1 * et + s1 * th))));
            psd_givens(al, -ga, c6R, s6R, r);
            c2 = c2R; s2 = s2R; c3 = c3R; s3 = s3R; c4 = c4R; s4 = s4R; c5 = c5R; s5 = s5R; c6 = c6R; s6 = s6R;
        }
#undef PSD_V
    }
    val1 = s5 * s6;
    val2 = s4 * val1;
    val3 = s3 * val2;
    al = c3 * val2 - c6;
    be = c2 * val3 - c5 * s6;
    ga = -c4 * val1;
    psd_givens(be, ga, c2, s2, r);
    psd_givens(al, r, c1, s1, r);
    c1o = c1; s1o = s1; c2o = c2; s2o = s2;
}

// ---- 2x2 solvers: serial, called by one lane on HBM scratch ------------------------------------------------------
// X[4 l + (0..3)] = [a b; c d] of block l (0-based), Hessenberg (full) block LAST; sg2[l] its signature

// rgeneralized.jl:1364-1396 `_qzrot2x2`
PSD_D void psd_g_qzrot2x2(int p, const double* X, const psd_gparams& P, double& c1, double& s1) {
    // order of the blocks: factors 2, 3, ..., p, 1  ->  signature of block l (0-based) is S[l+2] (S[1] for the last)
    double c2, s2, r, al, be, ga, de;
    const double* Hp = X + 4 * (p - 1);
    psd_givens(Hp[0], Hp[2], c1, s1, r);
    psd_givens(r, 1.0, c2, s2, r);
    for (int l = p - 2; l >= 0; --l) {
        const double* Hl = X + 4 * l;
        if (psd_gsig(P, l + 2)) {
            al = c2 * (c1 * Hl[0] + s1 * Hl[1]);
            be = s1 * c2 * Hl[3];
            ga = s2 * Hl[3];
            psd_givens(al, be, c1, s1, r);
            double v;
            psd_givens(r, ga, c2, s2, v);
        } else {
            al = c1 * s2 * Hl[0];
            ga = s1 * Hl[0];
            be = s2 * (c1 * Hl[1] + s1 * Hl[3]);
            de = c1 * Hl[3] - s1 * Hl[1];
            psd_givens(de, ga, c1, s1, r);
            al = c1 * al + s1 * be;
            be = c2 * Hl[3];
            psd_givens(be, al, c2, s2, r);
        }
    }
    al = s2 * Hp[3] - c1 * c2;
    be = -s1 * c2;
    psd_givens(al, be, c1, s1, r);
}

// rpschur2x2.jl:280-318 `_rp2x2ssr!`
PSD_D bool psd_g_rp2x2ssr(int p, double* X, const psd_gparams& P) {
    bool done = false;
    for (int iter = 1; iter <= 20; ++iter) {
        double c, s, r;
        psd_g_qzrot2x2(p, X, P, c, s);
        {
            double* Y = X + 4 * (p - 1);  // rmul!(H2s[p], G')
            double a1 = Y[0], a2 = Y[1];
            Y[0] = a1 * c + a2 * s;
            Y[1] = -a1 * s + a2 * c;
            a1 = Y[2]; a2 = Y[3];
            Y[2] = a1 * c + a2 * s;
            Y[3] = -a1 * s + a2 * c;
        }
        for (int l = 0; l < p - 1; ++l) {
            double* Y = X + 4 * l;
            if (psd_gsig(P, l + 2)) {
                double a1 = Y[0], a2 = Y[2];
                Y[0] = c * a1 + s * a2;
                Y[2] = -s * a1 + c * a2;
                a1 = Y[1]; a2 = Y[3];
                Y[1] = c * a1 + s * a2;
                Y[3] = -s * a1 + c * a2;
                psd_givens(Y[3], -Y[2], c, s, r);
                Y[3] = r;
                Y[2] = 0.0;
                const double t1 = c * Y[0] + s * Y[1], t2 = c * Y[1] - s * Y[0];
                Y[0] = t1;
                Y[1] = t2;
            } else {
                double a1 = Y[0], a2 = Y[1];
                Y[0] = a1 * c + a2 * s;
                Y[1] = -a1 * s + a2 * c;
                a1 = Y[2]; a2 = Y[3];
                Y[2] = a1 * c + a2 * s;
                Y[3] = -a1 * s + a2 * c;
                psd_givens(Y[0], Y[2], c, s, r);
                Y[0] = r;
                Y[2] = 0.0;
                const double t1 = c * Y[1] + s * Y[3], t2 = c * Y[3] - s * Y[1];
                Y[1] = t1;
                Y[3] = t2;
            }
        }
        double* Y = X + 4 * (p - 1);
        double a1 = Y[0], a2 = Y[2];
        Y[0] = c * a1 + s * a2;
        Y[2] = -s * a1 + c * a2;
        a1 = Y[1]; a2 = Y[3];
        Y[1] = c * a1 + s * a2;
        Y[3] = -s * a1 + c * a2;
        done = fabs(Y[2]) < PSD_DBL_EPS * fmax(fabs(Y[0]), fmax(fabs(Y[1]), fabs(Y[3])));
        if (done) break;
    }
    return done;
}

// rpschur2x2.jl:9-275 `_rpeigvals2x2` + `_sanitize_reigpair!` with a signature (Aord = 1:p, schurindex 1,
// recip = false).  X: [p][4] complex scratch holding the blocks in natural order (block 1 full).
PSD_D void psd_g_eigpair(const psd_gparams& P, int p, psd_z* X, psd_z* alpha, double* beta, double* scal,
                         bool& converged, bool& good) {
    const int k = p;
    converged = false;
    for (int iter = 1; iter <= 80; ++iter) {
        const double lhs = zabs(X[2]);
        double rhs = fmax(zabs(X[0]), zabs(X[3]));
        if (rhs == 0) rhs = zabs(X[1]);
        if (lhs <= PSD_DBL_EPS * rhs) {
            converged = true;
            break;
        }
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
                if (psd_gsig(P, l)) {
                    for (int q = 0; q < 3; ++q) psd_zrot_right_adj(ct, st, Z[q][0], Z[q][2]);
                    for (int q = 0; q < 3; ++q) psd_zrot_right_adj(c, s, Z[q][0], Z[q][1]);
                    psd_zgivens(Z[0][0], Z[2][0], ct, st, r);
                    psd_zgivens(Xl[0], Z[1][0], c, s, r);
                } else {
                    for (int q = 0; q < 3; ++q) psd_zrot_left(ct, st, Z[0][q], Z[2][q]);
                    for (int q = 0; q < 3; ++q) psd_zrot_left(c, s, Z[0][q], Z[1][q]);
                    psd_zgivens(Z[2][2], Z[2][0], ct, st, r);
                    Z[2][2] = r;
                    st = zneg(st);
                    for (int q = 0; q < 2; ++q) psd_zrot_right_adj(ct, st, Z[q][0], Z[q][2]);
                    psd_zgivens(Z[1][1], Z[1][0], c, s, r);
                    s = zneg(s);
                }
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
            if (psd_gsig(P, l)) {
                psd_zrot_right_adj(c, s, Y[0], Y[1]);
                psd_zrot_right_adj(c, s, Y[2], Y[3]);
                psd_zgivens(Y[0], Y[2], c, s, r);
                Y[0] = r;
                Y[2] = zmk(0.0, 0.0);
                psd_zrot_left(c, s, Y[1], Y[3]);
            } else {
                psd_zrot_left(c, s, Y[0], Y[2]);
                psd_zrot_left(c, s, Y[1], Y[3]);
                psd_zgivens(Y[3], Y[2], c, s, r);
                Y[3] = r;
                Y[2] = zmk(0.0, 0.0);
                s = zneg(s);
                psd_zrot_right_adj(c, s, Y[0], Y[1]);
            }
        }
        psd_zrot_left(ct0, st0, X[0], X[2]);
        psd_zrot_left(ct0, st0, X[1], X[3]);
        psd_zrot_right_adj(c, s, X[0], X[1]);
        psd_zrot_right_adj(c, s, X[2], X[3]);
    }
    for (int jj = 0; jj < 2; ++jj) {
        psd_z aj = zmk(1.0, 0.0);
        scal[jj] = 0.0;
        beta[jj] = 1.0;
        for (int l = 1; l <= k; ++l) {
            psd_z z = X[4 * (l - 1) + (jj == 0 ? 0 : 3)];
            double rhs = zabs(z);
            int sl = 0;
            if (rhs != 0) {
                sl = (int)floor(log2(rhs));
                z = zscal(exp2(-(double)sl), z);
            }
            if (psd_gsig(P, l)) {
                aj = zmul(aj, z);
                scal[jj] += sl;
            } else if (rhs == 0) {
                beta[jj] = 0.0;
            } else {
                aj = zdiv(aj, z);
                scal[jj] -= sl;
            }
            if ((l % 10 == 0) || (l == k)) {
                rhs = zabs(aj);
                if (rhs == 0) {
                    scal[jj] = 0;
                } else {
                    const int s2 = (int)floor(log2(rhs));
                    aj = zscal(exp2(-(double)s2), aj);
                    scal[jj] += s2;
                }
            }
        }
        alpha[jj] = aj;
    }
    if (alpha[1].im > 0) {
        const psd_z ta = alpha[0];
        alpha[0] = alpha[1];
        alpha[1] = ta;
        double ts = scal[0];
        scal[0] = scal[1];
        scal[1] = ts;
        ts = beta[0];
        beta[0] = beta[1];
        beta[1] = ts;
    }
    good = true;
    if (alpha[0].im != 0 || alpha[1].im != 0) {
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
        const double cs = fmax(zabs(alpha[0]), fmax(1.0, zabs(alpha[1])));
        good = fmin(misr, misc) <= cs * sqrt(PSD_DBL_EPS);
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
}

// generalized.jl:939-976 `_safeprod` with a signature, for the diagonal entry idx of all factors
PSD_D void psd_g_safeprod(const psd_gparams& P, int n, int p, int idx, double& alpha, double& beta, int& scale) {
    alpha = 1.0;
    beta = 1.0;
    scale = 0;
    for (int l = 1; l <= p; ++l) {
        const double xi = psd_gfac(P, n, l)(idx, idx);
        if (psd_gsig(P, l)) {
            alpha *= xi;
        } else if (xi == 0) {
            beta = 0.0;
        } else {
            alpha /= xi;
        }
        if (alpha == 0) {
            alpha = 0.0;
            scale = 0;
            if (beta == 0.0) return;
        } else {
            int guard = 0;
            while (fabs(alpha) < 1.0 && guard < 2200) {
                alpha *= 2.0;
                scale -= 1;
                ++guard;
            }
            while (fabs(alpha) >= 2.0 && guard < 4400) {
                alpha *= 0.5;
                scale += 1;
                ++guard;
            }
        }
    }
}

// ---- rotations applied directly on HBM by the whole workgroup (Cases II / III only) --------------------------------
PSD_D void psd_gg_left(const psd_mat<double>& M, int j, double c, double s, int c0, int c1) {
    PSD_SYNC();
    PSD_PAR_FOR(t, c1 - c0 + 1) {
        const int cc = c0 + t;
        const double a1 = M(j, cc), a2 = M(j + 1, cc);
        M(j, cc) = c * a1 + s * a2;
        M(j + 1, cc) = c * a2 - s * a1;
    }
    PSD_SYNC();
}
PSD_D void psd_gg_right(const psd_mat<double>& M, int j, double c, double s, int r0, int r1) {
    PSD_SYNC();
    PSD_PAR_FOR(t, r1 - r0 + 1) {
        const int r = r0 + t;
        const double a1 = M(r, j), a2 = M(r, j + 1);
        M(r, j) = c * a1 + s * a2;
        M(r, j + 1) = c * a2 - s * a1;
    }
    PSD_SYNC();
}
PSD_D void psd_gg_z(const psd_gparams& P, const psd_gstate& st, int m, int j, double c, double s) {
    if (!st.wantZ || m < P.zlo || m > P.zhi) return;
    psd_gg_right(psd_mat<double>{P.Z + (size_t)(m - 1) * st.n * st.n, st.n}, j, c, s, 1, st.n);
}
PSD_D void psd_gg_set2(const psd_mat<double>& M, int r1, int c1, double v1, int r2, int c2, double v2) {
    PSD_SYNC();
    PSD_ONE {
        M(r1, c1) = v1;
        M(r2, c2) = v2;
    }
    PSD_SYNC();
}
// one factor of a chain on HBM (same roles as psd_g_link)
PSD_D void psd_gg_link(const psd_mat<double>& M, int q, bool cols_in, double& c, double& s, int rlo, int chi) {
    double r;
    if (cols_in) {
        psd_gg_right(M, q, c, s, rlo, q + 1);
        psd_givens(M(q, q), M(q + 1, q), c, s, r);
        psd_gg_set2(M, q, q, r, q + 1, q, 0.0);
        psd_gg_left(M, q, c, s, q + 1, chi);
    } else {
        psd_gg_left(M, q, c, s, q, chi);
        psd_givens(M(q + 1, q + 1), -M(q + 1, q), c, s, r);
        psd_gg_set2(M, q + 1, q + 1, r, q + 1, q, 0.0);
        psd_gg_right(M, q, c, s, rlo, q);
    }
}

// rgeneralized.jl:329-442 Case II: zero on the diagonal of a positive factor
PSD_D void psd_gq_case2(const psd_gparams& P, psd_gstate& st, int ldeflate, int jdeflate) {
    const int n = st.n, p = st.p, jlo = st.jlo, ilast = st.ilast, ifirstm = st.ifirstm, ilastm = st.ilastm;
    const psd_mat<double> H1 = psd_gfac(P, n, 1);
    st.ncase2 += 1;
    psd_glog(P, st, 2, jlo, ilast);
    // first unshifted step, from the top, up to the zero
    for (int j = jlo; j <= jdeflate - 1; ++j) {
        double c, s, r;
        psd_givens(H1(j, j), H1(j + 1, j), c, s, r);
        psd_gg_set2(H1, j, j, r, j + 1, j, 0.0);
        psd_gg_left(H1, j, c, s, j + 1, ilastm);
        psd_gg_z(P, st, 1, j, c, s);
        for (int l = p; l >= 2; --l) {
            const int ntra = (l < ldeflate) ? (jdeflate - 2) : (jdeflate - 1);
            if (j > ntra) break;  // every later factor has the smaller count as well
            psd_gg_link(psd_gfac(P, n, l), j, psd_gsig(P, l), c, s, ifirstm, ilastm);
            psd_gg_z(P, st, l, j, c, s);
        }
        PSD_ONE {  // right side of H_1 only after every row rotation has been generated (:383-386)
            psd_gtr tr;
            tr.pos = j;
            tr.pad = 0;
            tr.c = c;
            tr.s = s;
            P.dG[j] = tr;
        }
    }
    PSD_SYNC();
    for (int j = jlo; j <= jdeflate - 2; ++j) {
        const psd_gtr g = P.dG[j];
        psd_gg_right(H1, j, g.c, g.s, ifirstm, j + 1);
    }
    // second unshifted step, from the bottom
    for (int j = ilast; j >= jdeflate + 1; --j) {
        double c, s, r;
        psd_givens(H1(j, j), -H1(j, j - 1), c, s, r);  // Givens(j, j-1, c, s') == standard (j-1, j; c, -s)
        psd_gg_set2(H1, j, j, r, j, j - 1, 0.0);
        psd_gg_right(H1, j - 1, c, s, ifirstm, j - 1);
        psd_gg_z(P, st, psd_gnext(1, p), j - 1, c, s);
        bool alive = true;
        for (int l = 2; l <= p; ++l) {
            const int ntra = (l > ldeflate) ? (jdeflate + 2) : (jdeflate + 1);
            if (j < ntra) {
                alive = false;
                break;
            }
            psd_gg_link(psd_gfac(P, n, l), j - 1, !psd_gsig(P, l), c, s, ifirstm, ilastm);
            psd_gg_z(P, st, psd_gnext(l, p), j - 1, c, s);
        }
        PSD_ONE {  // left side of H_1 after the whole pass (:437-440)
            psd_gtr tr;
            tr.pos = alive ? (j - 1) : -1;
            tr.pad = 0;
            tr.c = c;
            tr.s = s;
            P.dG[j] = tr;
        }
    }
    PSD_SYNC();
    for (int j = ilast; j >= jdeflate + 2; --j) {
        const psd_gtr g = P.dG[j];
        if (g.pos > 0) psd_gg_left(H1, j - 1, g.c, g.s, j - 1, ilastm);
    }
}

// rgeneralized.jl:444-616 Case III: zero on the diagonal of a negative factor
PSD_D void psd_gq_case3(const psd_gparams& P, psd_gstate& st, int ldeflate, int jdeflate) {
    const int n = st.n, p = st.p, jlo = st.jlo, ilast = st.ilast, ifirstm = st.ifirstm, ilastm = st.ilastm;
    const psd_mat<double> H1 = psd_gfac(P, n, 1);
    const psd_mat<double> Hd = psd_gfac(P, n, ldeflate);
    st.ncase3 += 1;
    psd_glog(P, st, 3, jlo, ilast);
    double c, s, r;
    if (jdeflate > (ilast - jlo + 1) / 2.0) {
        for (int j1 = jdeflate; j1 <= ilast - 1; ++j1) {  // chase the zero down
            int j = j1;
            psd_givens(Hd(j, j + 1), Hd(j + 1, j + 1), c, s, r);
            psd_gg_set2(Hd, j, j + 1, r, j + 1, j + 1, 0.0);
            psd_gg_left(Hd, j, c, s, j + 2, ilastm);
            int ln = psd_gnext(ldeflate, p);
            psd_gg_z(P, st, ln, j, c, s);
            for (int l = 1; l <= p - 1; ++l) {
                if (ln == 1) {
                    psd_gg_left(H1, j, c, s, j - 1, ilastm);
                    psd_givens(H1(j + 1, j), -H1(j + 1, j - 1), c, s, r);  // standard (j-1, j; c, s)
                    psd_gg_set2(H1, j + 1, j, r, j + 1, j - 1, 0.0);
                    psd_gg_right(H1, j - 1, c, s, ifirstm, j);
                    j -= 1;
                } else {
                    psd_gg_link(psd_gfac(P, n, ln), j, !psd_gsig(P, ln), c, s, ifirstm, ilastm);
                }
                ln = psd_gnext(ln, p);
                psd_gg_z(P, st, ln, j, c, s);
            }
            psd_gg_right(Hd, j, c, s, ifirstm, j);
        }
        const int j = ilast;  // deflate the last element of the Hessenberg factor
        psd_givens(H1(j, j), -H1(j, j - 1), c, s, r);
        psd_gg_set2(H1, j, j, r, j, j - 1, 0.0);
        psd_gg_right(H1, j - 1, c, s, ifirstm, j - 1);
        psd_gg_z(P, st, psd_gnext(1, p), j - 1, c, s);
        for (int l = 2; l <= ldeflate - 1; ++l) {
            psd_gg_link(psd_gfac(P, n, l), j - 1, !psd_gsig(P, l), c, s, ifirstm, ilastm);
            psd_gg_z(P, st, psd_gnext(l, p), j - 1, c, s);
        }
        psd_gg_right(Hd, j - 1, c, s, ifirstm, j);
    } else {
        for (int j1 = jdeflate; j1 >= jlo + 1; --j1) {  // chase the zero up
            int j = j1;
            psd_givens(Hd(j - 1, j), -Hd(j - 1, j - 1), c, s, r);  // standard (j-1, j; c, s)
            psd_gg_set2(Hd, j - 1, j, r, j - 1, j - 1, 0.0);
            psd_gg_right(Hd, j - 1, c, s, ifirstm, j - 2);
            psd_gg_z(P, st, ldeflate, j - 1, c, s);
            int ln = ldeflate - 1;
            for (int l = 1; l <= p - 1; ++l) {
                if (ln == 1) {
                    psd_gg_right(H1, j - 1, c, s, ifirstm, j + 1);
                    psd_givens(H1(j, j - 1), H1(j + 1, j - 1), c, s, r);
                    psd_gg_set2(H1, j, j - 1, r, j + 1, j - 1, 0.0);
                    psd_gg_left(H1, j, c, s, j, ilastm);
                    j += 1;
                } else {
                    psd_gg_link(psd_gfac(P, n, ln), j - 1, psd_gsig(P, ln), c, s, ifirstm, ilastm);
                }
                psd_gg_z(P, st, ln, j - 1, c, s);
                ln = (ln == 1) ? p : (ln - 1);
            }
            psd_gg_left(Hd, j - 1, c, s, j, ilastm);
        }
        const int j = jlo;  // deflate the first element of the Hessenberg factor
        psd_givens(H1(j, j), H1(j + 1, j), c, s, r);
        psd_gg_set2(H1, j, j, r, j + 1, j, 0.0);
        psd_gg_left(H1, j, c, s, j + 1, ilastm);
        psd_gg_z(P, st, 1, j, c, s);
        for (int l = p; l >= ldeflate + 1; --l) {
            psd_gg_link(psd_gfac(P, n, l), j, psd_gsig(P, l), c, s, ifirstm, ilastm);
            psd_gg_z(P, st, l, j, c, s);
        }
        psd_gg_left(Hd, j, c, s, j + 1, ilastm);
    }
}

// rgeneralized.jl:1015-1048 (section 510 of MB03BD): a single rotation at (j, j+1) through all factors, inside the
// window.  mode 0: generated from column j-1 of H_1 (tail of a sweep); mode 1: (c, s) given (perfect-shift rotation
// of a 2x2 deflation, :717-742); mode 2: generated from `side` = column hj of H_1 outside the window (stage 2 of the
// signed Hessenberg reduction, generalized.jl:1036-1044).
PSD_D void psd_gq_tail(const psd_gparams& P, const psd_gstate& st, const psd_gwin& w, int slot, int j, int mode,
                       double c, double s, double* side) {
    const int p = st.p;
    if (mode == 0) {
        double r;
        psd_givens(w.at(1, j, j - 1), w.at(1, j + 1, j - 1), c, s, r);
        psd_gwin_set2(w, 1, j, j - 1, r, j + 1, j - 1, 0.0);
    } else if (mode == 2) {
        double r;
        psd_givens(side[j - w.bs], side[j + 1 - w.bs], c, s, r);
        PSD_WAVE_SYNC();
        PSD_ONE {
            side[j - w.bs] = r;
            side[j + 1 - w.bs] = 0.0;
        }
        PSD_WAVE_SYNC();
    }
    // (stage 2: A_1 is still full to the left of the position, the rows take the rotation on every window column)
    psd_gwin_left(w, 1, j, c, s, (mode == 2) ? w.bs : j, st.ilastm);
    psd_gstore_tr(P, 1, slot, j, c, s);
    for (int l = p; l >= 2; --l)  // owner: rows owner of l if S[l], columns owner of l if !S[l]: both are l
        psd_g_link(w, l, j, psd_gsig(P, l), c, s, st.ifirstm, st.ilastm, P.tr, l, slot);
    psd_gwin_right(w, 1, j, c, s, st.ifirstm, st.ilastm);
}

// generalized.jl:1034-1079: one window of stage 2 of the signed Hessenberg reduction: column hj of A_1 is
// annihilated below the subdiagonal by row rotations (bottom-up), every rotation travels through all factors.
// Positions qs..qe (processed downwards from qe), window rows/columns qs..qe+1.
PSD_D void psd_gq_hess_window(const psd_gparams& P, psd_gstate& st, double* ldsd, double* side, int* lcnt) {
    const int n = st.n, p = st.p, hj = st.hj;
    const int nb = st.W - 1;
    const int qe = st.kcur;
    const int qs = (qe - nb + 1 > hj + 1) ? (qe - nb + 1) : (hj + 1);
    psd_gwin w;
    w.b = ldsd;
    w.W = st.W;
    w.ld = st.W + 1;
    w.bsz = st.W * (st.W + 1);
    w.bs = qs;
    w.be = qe + 1;
    const psd_mat<double> A1 = psd_gfac(P, n, 1);
    PSD_PAR_FOR(m, p) { lcnt[m] = 0; }
    PSD_PAR_FOR(t, w.be - w.bs + 1) { side[t] = A1(w.bs + t, hj); }
    psd_gwin_load(P, w, n, p);
    for (int q = qe; q >= qs; --q) psd_gq_tail(P, st, w, qe - q, q, 2, 0.0, 0.0, side);
    PSD_WAVE_SYNC();
    PSD_PAR_FOR(m, p) { lcnt[m] = qe - qs + 1; }
    psd_gwin_store(P, w, n, p);
    PSD_PAR_FOR(t, w.be - w.bs + 1) { A1(w.bs + t, hj) = side[t]; }
    psd_gdesc_write(P, st, lcnt, w.bs, w.be, w.be + 1, n, 1, w.bs - 1, 0, 0, 0, 0, 1, hj + 1);
    st.nwindows += 1;
    st.kcur = qs - 1;
    if (st.kcur < hj + 1) {
        st.hj = hj + 1;
        st.kcur = n - 1;
        if (st.hj > n - 2) st.phase = PSD_GPH_DONE;
    }
}

// ------------------------------------------------------------------------------------------------
// Stage 2 of the signed Hessenberg reduction as a pipeline over the factors.
// Within one column hj the rotation at position q only depends on the rotation at q+1 THROUGH THE SAME FACTOR (they
// share index q+1); its starting values come from column hj of A_1, which the lap-closing column rotation of A_1
// (columns q, q+1 >= hj+1) never touches.  So the laps of consecutive positions overlap: the factors of the lap
// (A_1 rows, A_p, ..., A_2, A_1 columns) are dealt out to the G wavefronts of one workgroup, wave g works on rotation
// b - g in beat b, the rotation (c, s) moves to the next wave through an LDS mailbox, one barrier per beat.  Wave 0
// owns A_1: it generates the rotation, applies it to the rows, and applies the column rotation that closes the lap
// G beats later (a fixed delay, so the result does not depend on timing; left and right multiplications commute).
// A window of K positions takes K + G beats of L links instead of K (p + 1) links.
#define PSD_GHESS_MAXWAVES 16
PSD_HD int psd_ghess_n0(int L, int p) {  // links of wave 0 next to its two A_1 updates
    int n0 = L - 2;
    if (n0 < 0) n0 = 0;
    if (n0 > p - 1) n0 = p - 1;
    return n0;
}
PSD_HD int psd_ghess_links(int p) { return (p + 1 + PSD_GHESS_MAXWAVES - 1) / PSD_GHESS_MAXWAVES; }
PSD_HD int psd_ghess_waves(int p) {
    const int L = psd_ghess_links(p), rest = p - 1 - psd_ghess_n0(L, p);
    return 1 + (rest + L - 1) / L;
}

namespace psd_wv {
#ifndef PSD_HOSTSIM
#undef PSD_TID
#undef PSD_TSTRIDE
#define PSD_TID PSD_TID_WAVE
#define PSD_TSTRIDE PSD_TSTRIDE_WAVE
#endif
#include "psd_rgz_chain.inl"

// one beat of wave g (called by all lanes of that wave); mail: [(G + 1)][2 parities][c, s]
PSD_D void psd_ghess_beat(const psd_gparams& P, const psd_gstate& st, const psd_gwin& w, double* side, double* mail, int g,
                          int G, int b, int K, int L, int qe) {
    const int p = st.p;
    const int n0 = psd_ghess_n0(L, p);
    if (g == 0) {
        const int kc = b - G;  // the lap that closes on the columns of A_1 in this beat
        if (kc >= 0 && kc < K) {
            const double* m = mail + (size_t)(G * 2 + (kc & 1)) * 2;
            psd_wv::psd_gwin_right(w, 1, qe - kc, m[0], m[1], st.ifirstm, st.ilastm);
        }
        if (b < K) {
            const int q = qe - b;
            double c, s, r;
            psd_givens(side[q - w.bs], side[q + 1 - w.bs], c, s, r);
            PSD_WAVE_SYNC();
            PSD_ONE {
                side[q - w.bs] = r;
                side[q + 1 - w.bs] = 0.0;
            }
            PSD_WAVE_SYNC();
            psd_wv::psd_gwin_left(w, 1, q, c, s, w.bs, st.ilastm);
            psd_wv::psd_gstore_tr(P, 1, b, q, c, s);
            for (int i = 0; i < n0; ++i) {
                const int l = p - i;
                psd_wv::psd_g_link(w, l, q, psd_gsig(P, l), c, s, st.ifirstm, st.ilastm, P.tr, l, b);
            }
            double* m = mail + (size_t)(1 * 2 + (b & 1)) * 2;
            PSD_ONE {
                m[0] = c;
                m[1] = s;
            }
            PSD_WAVE_SYNC();
        }
    } else {
        const int k = b - g;
        if (k >= 0 && k < K) {
            const double* mi = mail + (size_t)(g * 2 + (k & 1)) * 2;
            double c = mi[0], s = mi[1];
            const int q = qe - k;
            const int i0 = n0 + (g - 1) * L;
            const int i1 = (i0 + L < p - 1) ? (i0 + L) : (p - 1);
            for (int i = i0; i < i1; ++i) {
                const int l = p - i;
                psd_wv::psd_g_link(w, l, q, psd_gsig(P, l), c, s, st.ifirstm, st.ilastm, P.tr, l, k);
            }
            double* mo = mail + (size_t)((g + 1) * 2 + (k & 1)) * 2;
            PSD_ONE {
                mo[0] = c;
                mo[1] = s;
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

// LDS of the pipelined kernel: window blocks, column hj of A_1 (<= 64 entries), mailboxes, list lengths, and the rotation
// table of the scan form (4 doubles per factor)
PSD_HD size_t psd_ghess_lds_bytes(int p, int W) {
    size_t b = (size_t)p * W * (W + 1) * 8 + 64 * 8 + (size_t)4 * (PSD_GHESS_MAXWAVES + 1) * 8 + (size_t)p * 4;
    b = (b + 15) & ~(size_t)15;
    b += (size_t)p * 4 * 8;
    return (b + 15) & ~(size_t)15;
}

// ------------------------------------------------------------------------------------------------
// Stage 2 of the signed Hessenberg reduction, scan form (round 4; generalized.jl:1034-1079).  One rotation per position
// travels through all factors (psd_gq_tail, mode 2).  As in the scan chases of the sweeps (psd_chase3.h, psd_zchase3.h)
// the pairs the p rotations are made from form a chain of 2 x 2 triangular matrix-vector products that needs no rotation
// and no update: with U_l = A_l[q:q+1, q:q+1] and the incoming rotation (c, s),
//     S[l] true  (the rotation acts on the columns, rgeneralized.jl:980-991):   (f, g) = U_l (c, s)',
//     S[l] false (it acts on the rows, the fill goes by a column rotation, :993-1004):
//                (p11, -p10) = adj(U_l) (c, s)',   adj(U) = [u11 -u01; 0 u00]   (the inverted factor),
// and the rotation made from a pair is its normalisation (up to the sign convention of givensAlgorithm, which does not
// matter: whichever of the two rotations comes out is applied to both sides and recorded).  So: z_p = M_p (c_1, s_1)',
// z_{l-1} = M_{l-1} z_l on the chain lanes, every rotation at once (one lane per factor), then all factors updated side by
// side by every wavefront of the workgroup: first each factor's INCOMING rotation (columns for S true, rows for S
// false; A_1: its own rotation on the rows), then the rotation made AT the factor (rows resp. columns, the annihilated
// entry set to zero; A_1: the rotation that closes the lap, on its columns).  K positions = K rounds of (scan, two update
// phases) instead of K + G beats of L dependent links each.
// tab: [p][4] = (c, s) of the rotation made at factor l (l = 1: from column hj of A_1), then its pair (scratch).
// The rendezvous between the chain and the update phases of the scan windows: LDS traffic has to be complete, the
// rotation records on their way to device memory have not (the next kernel reads them) — __syncthreads() would wait for
// their acknowledge at every position.
#ifndef PSD_HOSTSIM
#define PSD_GH_BARRIER() PSD_PAIR_BARRIER()
#else
#define PSD_GH_BARRIER() PSD_SYNC()
#endif
PSD_D void psd_ghess_scan_window(const psd_gparams& P, const psd_gstate& st, const psd_gwin& w, double* side, double* tab,
                                 int qe, int K) {
    const int p = st.p;
    const int NT = PSD_NTHREADS;
    const int r0 = (st.ifirstm > w.bs) ? st.ifirstm : w.bs;
    const int c1 = (st.ilastm < w.be) ? st.ilastm : w.be;
    const int tpf = (NT / p > 0) ? (NT / p) : 1;  // threads per factor in the update phases
#ifndef PSD_HOSTSIM
    // (per thread, once per window: its factor, its place among the factor's threads, the factor's signature)
    const int myf = PSD_TID / tpf, myq = PSD_TID - myf * tpf;
    const bool mysg = (myf >= 1 && myf < p) ? psd_gsig(P, myf + 1) : false;
    // chain lanes: the signatures of their four links
    bool lsg[4] = {true, true, true, true};
    if (PSD_TID < 16) {
#pragma unroll
        for (int q4 = 0; q4 < 4; ++q4) {
            const int c = 4 * PSD_TID + q4;
            if (c < p - 1) lsg[q4] = psd_gsig(P, p - c);
        }
    }
#endif
    for (int b = 0; b < K; ++b) {
        const int q = qe - b, slot = b;
#ifndef PSD_HOSTSIM
        if (PSD_TID < 64) {
            // ---- the rotation of A_1 (generalized.jl:1036-1044) and the chain of pairs
            const int lane = PSD_TID;
            double c1r, s1r, r1;
            psd_givens(side[q - w.bs], side[q + 1 - w.bs], c1r, s1r, r1);
            PSD_WAVE_SYNC();
            if (lane == 0) {
                side[q - w.bs] = r1;
                side[q + 1 - w.bs] = 0.0;
            }
            double M0[4], M1[4], M2[4], zq0[4], zq1[4];
            const bool chl = lane < 16;
#pragma unroll
            for (int q4 = 0; q4 < 4; ++q4) {
                const int c = 4 * lane + q4, lf = p - c;
                M0[q4] = 1.0; M1[q4] = 0.0; M2[q4] = 1.0;
                if (chl && c < p - 1) {
                    const double* u = w.b + (lf - 1) * w.bsz + (q - w.bs) * w.ld + (q - w.bs);
                    const double u00 = u[0], u01 = u[w.ld], u11 = u[w.ld + 1];
                    if (lsg[q4]) { M0[q4] = u00; M1[q4] = u01; M2[q4] = u11; }
                    else { M0[q4] = u11; M1[q4] = -u01; M2[q4] = u00; }
                }
                zq0[q4] = zq1[q4] = 0.0;
            }
            const int nsteps = (p - 1 + 3) / 4;
            double z0 = 0.0, z1 = 0.0;
            for (int s = 0; s < nsteps; ++s) {
                double w0 = psd_c3_shr(z0, c1r), w1 = psd_c3_shr(z1, s1r);
                if (s <= lane) {
#pragma unroll
                    for (int q4 = 0; q4 < 4; ++q4) {
                        const double n0 = __builtin_fma(M0[q4], w0, M1[q4] * w1);
                        const double n1 = M2[q4] * w1;
                        w0 = n0;
                        w1 = n1;
                        zq0[q4] = n0;
                        zq1[q4] = n1;
                    }
                    const int e = psd_c3_expo(fmax(fabs(w0), fabs(w1)));
                    z0 = psd_c3_ldexp(w0, -e);
                    z1 = psd_c3_ldexp(w1, -e);
                }
            }
#pragma unroll
            for (int q4 = 0; q4 < 4; ++q4) {
                const int c = 4 * lane + q4, lf = p - c;
                if (chl && c < p - 1) {
                    tab[(lf - 1) * 4 + 2] = zq0[q4];
                    tab[(lf - 1) * 4 + 3] = zq1[q4];
                }
            }
            // The pair the reference sees at factor l is M_l applied to the ROTATION of factor l + 1, (c, s) = pair / r, and
            // givensAlgorithm's r is negative when |f| > |g| and f < 0: the chain's vector is that pair up to the signs of
            // the r's above it.  sigma_l = (the factor above with |z_0| > |z_1| nearest to l decides: sign of its z_0; none:
            // +), so that the rotations come out with the reference's signs (two ballots instead of a serial product).
            double zf = 0.0, zg = 0.0;
            if (lane >= 1 && lane < p) {
                zf = tab[lane * 4 + 2];
                zg = tab[lane * 4 + 3];
            }
            const unsigned long long bigm = __ballot(lane >= 1 && lane < p && fabs(zf) > fabs(zg));
            const unsigned long long negm = __ballot(zf < 0.0);
            if (lane < p) {
                double c = c1r, s = s1r;
                if (lane >= 1) {
                    const unsigned long long above = (lane < 63) ? (bigm & ~((2ull << lane) - 1ull)) : 0ull;
                    double sg = 1.0;
                    if (above != 0ull) {
                        const int bpos = __ffsll((long long)above) - 1;
                        if ((negm >> bpos) & 1ull) sg = -1.0;
                    }
                    double r;
                    psd_givens(sg * zf, sg * zg, c, s, r);
                }
                tab[lane * 4 + 0] = c;
                tab[lane * 4 + 1] = s;
                if (slot < PSD_GTR_CAP) {
                    psd_gtr tr;
                    tr.pos = q;
                    tr.pad = 0;
                    tr.c = c;
                    tr.s = s;
                    P.tr[(size_t)lane * PSD_GTR_CAP + slot] = tr;
                }
            }
        }
#else
        {
            double c1r, s1r, r1;
            psd_givens(side[q - w.bs], side[q + 1 - w.bs], c1r, s1r, r1);
            side[q - w.bs] = r1;
            side[q + 1 - w.bs] = 0.0;
            double z0 = c1r, z1 = s1r, sigma = 1.0;  // (sigma: see the device form)
            int since = 0;
            tab[0] = c1r;
            tab[1] = s1r;
            for (int lf = p; lf >= 2; --lf) {
                const double u00 = w.at(lf, q, q), u01 = w.at(lf, q, q + 1), u11 = w.at(lf, q + 1, q + 1);
                const bool sg = psd_gsig(P, lf);
                const double m0 = sg ? u00 : u11, m1 = sg ? u01 : -u01, m2 = sg ? u11 : u00;
                const double n0 = m0 * z0 + m1 * z1, n1 = m2 * z1;
                double c, s, r;
                psd_givens(sigma * n0, sigma * n1, c, s, r);
                if (fabs(n0) > fabs(n1)) sigma = (n0 < 0.0) ? -1.0 : 1.0;
                tab[(lf - 1) * 4 + 0] = c;
                tab[(lf - 1) * 4 + 1] = s;
                z0 = n0;
                z1 = n1;
                if (++since == 4) {
                    since = 0;
                    const int e = psd_c3_expo(fmax(fabs(z0), fabs(z1)));
                    z0 = psd_c3_ldexp(z0, -e);
                    z1 = psd_c3_ldexp(z1, -e);
                }
            }
            for (int l = 1; l <= p; ++l) {
                if (slot < PSD_GTR_CAP) {
                    psd_gtr tr;
                    tr.pos = q;
                    tr.pad = 0;
                    tr.c = tab[(l - 1) * 4 + 0];
                    tr.s = tab[(l - 1) * 4 + 1];
                    P.tr[(size_t)(l - 1) * PSD_GTR_CAP + slot] = tr;
                }
            }
        }
#endif
        PSD_GH_BARRIER();
        // ---- the two update phases: thread (f, qq) is the qq-th of the tpf threads of factor f + 1
        for (int sub = 0; sub < 2; ++sub) {
            PSD_PAR_FOR(t, NT) {
#ifndef PSD_HOSTSIM
                const int f = myf, qq = myq;
                (void)t;
#else
                const int f = t / tpf, qq = t - f * tpf;
#endif
                if (f < p) {
                    const int l = f + 1;
#ifndef PSD_HOSTSIM
                    const bool sg = mysg;
#else
                    const bool sg = (l == 1) ? false : psd_gsig(P, l);
#endif
                    // which rotation, from which side: sub 0 = the incoming one, sub 1 = the one made at this factor (A_1:
                    // sub 0 its own on the rows, sub 1 the one of factor 2 — the end of the chain — on the columns)
                    int src;
                    bool rows;  // true: lmul on rows (q, q+1); false: rmul on columns (q, q+1)
                    if (l == 1) {
                        src = (sub == 0) ? 0 : ((p >= 2) ? 1 : 0);
                        rows = sub == 0;
                    } else {
                        src = (sub == 0) ? ((l == p) ? 0 : l) : (l - 1);
                        rows = (sub == 0) ? !sg : sg;
                    }
                    const double c = tab[src * 4 + 0], s = tab[src * 4 + 1];
                    double* const blk = w.b + (l - 1) * w.bsz;
                    if (rows) {
                        // columns: A_1 takes its rotation on every window column (it is still full to the left of the
                        // position), a factor from q on
                        const int ca = (l == 1) ? w.bs : q;
                        const int cb = (l == 1) ? ((st.ilastm < w.be) ? st.ilastm : w.be) : c1;
                        const bool fix = l >= 2 && sub == 1;  // (S true: column q becomes (r, 0), rgeneralized.jl:985-987)
                        for (int cc = ca + qq; cc <= cb; cc += tpf) {
                            double* e = blk + (cc - w.bs) * w.ld + (q - w.bs);
                            const double a1 = e[0], a2 = e[1];
                            e[0] = c * a1 + s * a2;
                            e[1] = (fix && cc == q) ? 0.0 : (c * a2 - s * a1);
                        }
                    } else {
                        const int ra = (l == 1) ? ((st.ifirstm > w.bs) ? st.ifirstm : w.bs) : r0;
                        const int rb = (l == 1) ? ((st.ilastm < w.be) ? st.ilastm : w.be) : (q + 1);
                        const bool fix = l >= 2 && sub == 1;  // (S false: (q+1, q) becomes 0, (q+1, q+1) = r, :998-1000)
                        for (int r = ra + qq; r <= rb; r += tpf) {
                            double* e = blk + (q - w.bs) * w.ld + (r - w.bs);
                            const double a1 = e[0], a2 = e[w.ld];
                            e[0] = (fix && r == q + 1) ? 0.0 : (c * a1 + s * a2);
                            e[w.ld] = c * a2 - s * a1;
                        }
                    }
                }
            }
            PSD_GH_BARRIER();
        }
    }
}

// ------------------------------------------------------------------------------------------------
// The implicit double-shift sweep in scan form (round 4; rgeneralized.jl:955-1050).  Per position j two rotations leave
// H_1 — A on (j+1, j+2), B on (j, j+1) — and each travels down the factors p..2 exactly as the single rotation of
// stage 2 above (psd_ghess_scan_window): the pairs the rotations are made from are a chain of 2 x 2 triangular
// matrix-vector products.  Chain B's blocks U_l[j:j+1, j:j+1] are the ones A's passage has left behind; the lane of
// factor l forms the two entries that changed from its own 3 x 3 block and A's two rotations (a dozen operations), so both
// chains are done before any factor is updated.  Then every factor takes its four rotations in two phases — every
// column rotation (A then B), then every row rotation (A then B) — each over the full three rows / columns of the
// position: U' = R_B R_A U C_A C_B whatever order the serial sweep applies them in.  H_1 is factor 1 of both phases (its
// own rotations on the rows, the ends of the chains on the columns).  Four wavefronts: wavefront 0 owns the chains, all
// update.  tab: [p][8] = A's (c, s) made at factor l, B's (c, s), chain B's (m0, m1, m2), spare.
struct psd_gc {
    int cmd;  // 0: leave, 1: run
    int p, W, ld, bsz, bs, be;
    int j0, j1, slot0;
    int ifirstm, ilastm;
    int wboff, taboff;
    psd_gtr* tr;
    const unsigned char* S;
};

#ifndef PSD_HOSTSIM
// chain lanes 0..15 carry four links each (factors p - 4 lane - q4): pairs into tab[.][6..7], then the rotations with the
// reference's signs (see psd_ghess_scan_window) into tab[.][col], tab[.][col + 1]; factor 1 gets (cs, ss) itself.
PSD_D void psd_gs3_chain(double* tab, int p, int lane, const double (&M0)[4], const double (&M1)[4], const double (&M2)[4],
                         double cs, double ss, int col) {
    const bool chl = lane < 16;
    double zq0[4] = {0.0, 0.0, 0.0, 0.0}, zq1[4] = {0.0, 0.0, 0.0, 0.0};
    const int nsteps = (p - 1 + 3) / 4;
    double z0 = 0.0, z1 = 0.0;
    for (int s = 0; s < nsteps; ++s) {
        double w0 = psd_c3_shr(z0, cs), w1 = psd_c3_shr(z1, ss);
        if (s <= lane) {
#pragma unroll
            for (int q4 = 0; q4 < 4; ++q4) {
                const double n0 = __builtin_fma(M0[q4], w0, M1[q4] * w1);
                const double n1 = M2[q4] * w1;
                w0 = n0;
                w1 = n1;
                zq0[q4] = n0;
                zq1[q4] = n1;
            }
            const int e = psd_c3_expo(fmax(fabs(w0), fabs(w1)));
            z0 = psd_c3_ldexp(w0, -e);
            z1 = psd_c3_ldexp(w1, -e);
        }
    }
#pragma unroll
    for (int q4 = 0; q4 < 4; ++q4) {
        const int c = 4 * lane + q4, lf = p - c;
        if (chl && c < p - 1) {
            tab[(lf - 1) * 8 + 6] = zq0[q4];
            tab[(lf - 1) * 8 + 7] = zq1[q4];
        }
    }
    PSD_WAVE_SYNC();
    double zf = 0.0, zg = 0.0;
    if (lane >= 1 && lane < p) {
        zf = tab[lane * 8 + 6];
        zg = tab[lane * 8 + 7];
    }
    const unsigned long long bigm = __ballot(lane >= 1 && lane < p && fabs(zf) > fabs(zg));
    const unsigned long long negm = __ballot(zf < 0.0);
    if (lane < p) {
        double c = cs, s = ss;
        if (lane >= 1) {
            const unsigned long long above = (lane < 63) ? (bigm & ~((2ull << lane) - 1ull)) : 0ull;
            double sg = 1.0;
            if (above != 0ull) {
                const int bpos = __ffsll((long long)above) - 1;
                if ((negm >> bpos) & 1ull) sg = -1.0;
            }
            double r;
            psd_givens(sg * zf, sg * zg, c, s, r);
        }
        tab[lane * 8 + col] = c;
        tab[lane * 8 + col + 1] = s;
    }
    PSD_WAVE_SYNC();
}
#endif

// wv: this wavefront (0 = the one that runs the state machine), nw: wavefronts of the workgroup
PSD_D void psd_gs3_run(const psd_gc& C, int wv, int nw) {
    PSD_LDS_DECL;
    const int p = C.p;
    psd_gwin w;
    w.b = (double*)(psd_lds + C.wboff);
    w.W = C.W; w.ld = C.ld; w.bsz = C.bsz; w.bs = C.bs; w.be = C.be;
    double* tab = (double*)(psd_lds + C.taboff);
    const int r0 = (C.ifirstm > w.bs) ? C.ifirstm : w.bs;
    const int c1 = (C.ilastm < w.be) ? C.ilastm : w.be;
#ifndef PSD_HOSTSIM
    const int NT = 64 * nw;
    const int tid = (int)threadIdx.x + 64 * wv;
    const int lane = (int)threadIdx.x;
#else
    const int NT = PSD_NTHREADS;
    (void)wv; (void)nw;
#endif
    const int tpf = (NT / p > 0) ? (NT / p) : 1;
#ifndef PSD_HOSTSIM
    const int myf = tid / tpf, myq = tid - myf * tpf;
    const bool mysg = (myf >= 1 && myf < p) ? (C.S[myf] != 0) : false;
    bool lsg[4] = {true, true, true, true};
    bool lanesg = true;  // signature of factor lane + 1
    if (wv == 0) {
        if (lane < 16) {
#pragma unroll
            for (int q4 = 0; q4 < 4; ++q4) {
                const int c = 4 * lane + q4;
                if (c < p - 1) lsg[q4] = C.S[p - c - 1] != 0;
            }
        }
        if (lane >= 1 && lane < p) lanesg = C.S[lane] != 0;
    }
#endif
    for (int j = C.j0; j <= C.j1; ++j) {
        const int slot = C.slot0 + 2 * (j - C.j0);
        double* const h1 = w.b;  // factor 1
#ifndef PSD_HOSTSIM
        if (wv == 0) {
            // ---- rgeneralized.jl:960-968: the two rotations from column j - 1 of H_1
            double* cj = h1 + (j - 1 - w.bs) * w.ld + (j - w.bs);
            double c2, s2, r2, c1r, s1r, r1;
            psd_givens(cj[1], cj[2], c2, s2, r2);
            psd_givens(cj[0], r2, c1r, s1r, r1);
            PSD_WAVE_SYNC();
            if (lane == 0) {
                cj[0] = r1;
                cj[1] = 0.0;
                cj[2] = 0.0;
            }
            // ---- chain A at q = j + 1
            double M0[4], M1[4], M2[4];
#pragma unroll
            for (int q4 = 0; q4 < 4; ++q4) {
                const int c = 4 * lane + q4, lf = p - c;
                M0[q4] = 1.0; M1[q4] = 0.0; M2[q4] = 1.0;
                if (lane < 16 && c < p - 1) {
                    const double* u = w.b + (lf - 1) * w.bsz + (j + 1 - w.bs) * w.ld + (j + 1 - w.bs);
                    const double u00 = u[0], u01 = u[w.ld], u11 = u[w.ld + 1];
                    if (lsg[q4]) { M0[q4] = u00; M1[q4] = u01; M2[q4] = u11; }
                    else { M0[q4] = u11; M1[q4] = -u01; M2[q4] = u00; }
                }
            }
            psd_gs3_chain(tab, p, lane, M0, M1, M2, c2, s2, 0);
            // ---- what A's passage leaves of U_l[j:j+1, j:j+1] (lane l - 1, l >= 2), as chain B's matrix
            if (lane >= 1 && lane < p) {
                const int l = lane + 1;
                const double* u = w.b + (l - 1) * w.bsz + (j - w.bs) * w.ld + (j - w.bs);  // (j, j)
                const double a = u[0], b1 = u[w.ld], b2 = u[2 * w.ld], d11 = u[w.ld + 1], d12 = u[2 * w.ld + 1], d22 = u[2 * w.ld + 2];
                const int in = (l == p) ? 0 : l;  // (table row of the incoming rotation: made at factor l + 1, or H_1's own)
                const double ci = tab[in * 8 + 0], si = tab[in * 8 + 1], co = tab[lane * 8 + 0], so = tab[lane * 8 + 1];
                double u01, u11;
                if (lanesg) {
                    u01 = ci * b1 + si * b2;
                    u11 = co * (ci * d11 + si * d12) + so * (si * d22);
                } else {
                    u01 = co * b1 + so * b2;
                    u11 = co * (ci * d11) + so * (ci * d12 + si * d22);
                }
                tab[lane * 8 + 4] = lanesg ? a : u11;
                tab[lane * 8 + 5] = lanesg ? u01 : -u01;
                tab[lane * 8 + 6] = lanesg ? u11 : a;
            }
            PSD_WAVE_SYNC();
#pragma unroll
            for (int q4 = 0; q4 < 4; ++q4) {
                const int c = 4 * lane + q4, lf = p - c;
                M0[q4] = 1.0; M1[q4] = 0.0; M2[q4] = 1.0;
                if (lane < 16 && c < p - 1) {
                    M0[q4] = tab[(lf - 1) * 8 + 4];
                    M1[q4] = tab[(lf - 1) * 8 + 5];
                    M2[q4] = tab[(lf - 1) * 8 + 6];
                }
            }
            PSD_WAVE_SYNC();
            psd_gs3_chain(tab, p, lane, M0, M1, M2, c1r, s1r, 2);
            if (lane < p && slot + 1 < PSD_GTR_CAP) {
                psd_gtr tr;
                tr.pos = j + 1;
                tr.pad = 0;
                tr.c = tab[lane * 8 + 0];
                tr.s = tab[lane * 8 + 1];
                C.tr[(size_t)lane * PSD_GTR_CAP + slot] = tr;
                tr.pos = j;
                tr.c = tab[lane * 8 + 2];
                tr.s = tab[lane * 8 + 3];
                C.tr[(size_t)lane * PSD_GTR_CAP + slot + 1] = tr;
            }
        }
        PSD_PAIR_BARRIER();
#else
        {
            double* cj = h1 + (j - 1 - w.bs) * w.ld + (j - w.bs);
            double c2, s2, r2, c1r, s1r, r1;
            psd_givens(cj[1], cj[2], c2, s2, r2);
            psd_givens(cj[0], r2, c1r, s1r, r1);
            cj[0] = r1;
            cj[1] = 0.0;
            cj[2] = 0.0;
            for (int ch = 0; ch < 2; ++ch) {
                const int col = 2 * ch;
                double z0 = ch ? c1r : c2, z1 = ch ? s1r : s2, sigma = 1.0;
                int since = 0;
                tab[col] = z0;
                tab[col + 1] = z1;
                if (ch == 1)  // chain B's matrices: what A's passage leaves of the blocks at (j, j + 1)
                    for (int l = 2; l <= p; ++l) {
                        const double* u = w.b + (l - 1) * w.bsz + (j - w.bs) * w.ld + (j - w.bs);
                        const double a = u[0], b1 = u[w.ld], b2 = u[2 * w.ld], d11 = u[w.ld + 1], d12 = u[2 * w.ld + 1], d22 = u[2 * w.ld + 2];
                        const int in = (l == p) ? 0 : l;
                        const double ci = tab[in * 8 + 0], si = tab[in * 8 + 1], co = tab[(l - 1) * 8 + 0], so = tab[(l - 1) * 8 + 1];
                        const bool sg = C.S[l - 1] != 0;
                        double u01, u11;
                        if (sg) {
                            u01 = ci * b1 + si * b2;
                            u11 = co * (ci * d11 + si * d12) + so * (si * d22);
                        } else {
                            u01 = co * b1 + so * b2;
                            u11 = co * (ci * d11) + so * (ci * d12 + si * d22);
                        }
                        tab[(l - 1) * 8 + 4] = sg ? a : u11;
                        tab[(l - 1) * 8 + 5] = sg ? u01 : -u01;
                        tab[(l - 1) * 8 + 6] = sg ? u11 : a;
                    }
                for (int lf = p; lf >= 2; --lf) {
                    double m0, m1, m2;
                    if (ch == 0) {
                        const double* u = w.b + (lf - 1) * w.bsz + (j + 1 - w.bs) * w.ld + (j + 1 - w.bs);
                        const double u00 = u[0], u01 = u[w.ld], u11 = u[w.ld + 1];
                        const bool sg = C.S[lf - 1] != 0;
                        m0 = sg ? u00 : u11; m1 = sg ? u01 : -u01; m2 = sg ? u11 : u00;
                    } else {
                        m0 = tab[(lf - 1) * 8 + 4]; m1 = tab[(lf - 1) * 8 + 5]; m2 = tab[(lf - 1) * 8 + 6];
                    }
                    const double n0 = m0 * z0 + m1 * z1, n1 = m2 * z1;
                    double c, s, r;
                    psd_givens(sigma * n0, sigma * n1, c, s, r);
                    if (fabs(n0) > fabs(n1)) sigma = (n0 < 0.0) ? -1.0 : 1.0;
                    tab[(lf - 1) * 8 + col] = c;
                    tab[(lf - 1) * 8 + col + 1] = s;
                    z0 = n0;
                    z1 = n1;
                    if (++since == 4) {
                        since = 0;
                        const int e = psd_c3_expo(fmax(fabs(z0), fabs(z1)));
                        z0 = psd_c3_ldexp(z0, -e);
                        z1 = psd_c3_ldexp(z1, -e);
                    }
                }
            }
            for (int l = 1; l <= p; ++l)
                if (slot + 1 < PSD_GTR_CAP) {
                    psd_gtr tr;
                    tr.pos = j + 1;
                    tr.pad = 0;
                    tr.c = tab[(l - 1) * 8 + 0];
                    tr.s = tab[(l - 1) * 8 + 1];
                    C.tr[(size_t)(l - 1) * PSD_GTR_CAP + slot] = tr;
                    tr.pos = j;
                    tr.c = tab[(l - 1) * 8 + 2];
                    tr.s = tab[(l - 1) * 8 + 3];
                    C.tr[(size_t)(l - 1) * PSD_GTR_CAP + slot + 1] = tr;
                }
        }
        PSD_SYNC();
#endif
        // ---- the two update phases: thread (f, qq) is the qq-th of the tpf threads of factor f + 1
        for (int sub = 0; sub < 2; ++sub) {
#ifndef PSD_HOSTSIM
            {  // (blockDim.x = 64: the data-parallel macros would run over one wavefront's lanes only)
                const int f = myf, qq = myq;
#else
            PSD_PAR_FOR(t, NT) {
                const int f = t / tpf, qq = t - f * tpf;
#endif
                if (f < p) {
                    const int l = f + 1;
#ifndef PSD_HOSTSIM
                    const bool sg = mysg;
#else
                    const bool sg = (l == 1) ? false : (C.S[l - 1] != 0);
#endif
                    const int in = (l == p) ? 0 : l;  // table row of the incoming rotations
                    // sub 0: the column rotations, sub 1: the row rotations.  S true: incoming on the columns, made on the
                    // rows; S false the other way round; H_1: the chains' ends (made at factor 2) on the columns, its own
                    // on the rows
                    int src;
                    if (l == 1) src = (sub == 0) ? ((p >= 2) ? 1 : 0) : 0;
                    else src = ((sub == 0) == sg) ? in : (l - 1);
                    const double ca = tab[src * 8 + 0], sa = tab[src * 8 + 1], cb = tab[src * 8 + 2], sb = tab[src * 8 + 3];
                    double* const blk = w.b + (l - 1) * w.bsz;
                    if (sub == 0) {
                        const int ra = r0;
                        int rb = (l == 1) ? (j + 3) : (j + 2);
                        if (l == 1 && rb > C.ilastm) rb = C.ilastm;
                        if (rb > w.be) rb = w.be;
                        for (int r = ra + qq; r <= rb; r += tpf) {
                            double* e = blk + (j - w.bs) * w.ld + (r - w.bs);
                            double a0 = e[0], a1 = e[w.ld], a2 = e[2 * w.ld];
                            const double t1 = ca * a1 + sa * a2;
                            a2 = ca * a2 - sa * a1;
                            a1 = t1;
                            const double t0 = cb * a0 + sb * a1;
                            a1 = cb * a1 - sb * a0;
                            a0 = t0;
                            e[0] = a0;
                            e[w.ld] = a1;
                            e[2 * w.ld] = a2;
                        }
                    } else {
                        const int cA = j;
                        for (int cc = cA + qq; cc <= c1; cc += tpf) {
                            double* e = blk + (cc - w.bs) * w.ld + (j - w.bs);
                            double a0 = e[0], a1 = e[1], a2 = e[2];
                            const double t1 = ca * a1 + sa * a2;
                            a2 = ca * a2 - sa * a1;
                            a1 = t1;
                            const double t0 = cb * a0 + sb * a1;
                            a1 = cb * a1 - sb * a0;
                            a0 = t0;
                            if (l >= 2) {  // (the factors stay triangular: what the rotations annihilate is set to zero)
                                if (cc == j) { a1 = 0.0; a2 = 0.0; }
                                if (cc == j + 1) a2 = 0.0;
                            }
                            e[0] = a0;
                            e[1] = a1;
                            e[2] = a2;
                        }
                    }
                }
            }
#ifndef PSD_HOSTSIM
            PSD_PAIR_BARRIER();
#else
            PSD_SYNC();
#endif
        }
    }
}

#ifndef PSD_HOSTSIM
// the helper wavefronts of a sweep workgroup (threadIdx.y >= 1): they take part in the scan runs and in nothing else
PSD_D void psd_gs3_helper(int gcoff) {
    PSD_LDS_DECL;
    const psd_gc* cmd = (const psd_gc*)(psd_lds + gcoff);
    for (;;) {
        PSD_PAIR_BARRIER();
        const psd_gc C = *cmd;
        if (C.cmd == 0) return;
        psd_gs3_run(C, PSD_WAVE_ROLE, (int)blockDim.y);
    }
}
PSD_D void psd_gs3_release(int gcoff) {
    PSD_LDS_DECL;
    psd_gc* cmd = (psd_gc*)(psd_lds + gcoff);
    PSD_ONE { cmd->cmd = 0; }
    PSD_PAIR_BARRIER();
}
#define PSD_GS3_ENTER(P)                          \
    if ((P).gcoff != 0 && PSD_WAVE_ROLE >= 1) {   \
        psd_gs3_helper((P).gcoff);                \
        return;                                   \
    }
#define PSD_GS3_LEAVE(P) \
    if ((P).gcoff != 0) psd_gs3_release((P).gcoff)
#else
#define PSD_GS3_ENTER(P) ((void)0)
#define PSD_GS3_LEAVE(P) ((void)0)
#endif

// wavefront 0's side of a scan run over the positions j0..j1 of a sweep window
PSD_D void psd_gs3_lead(const psd_gparams& P, const psd_gstate& st, const psd_gwin& w, int j0, int j1, int slot0) {
    PSD_LDS_DECL;
    psd_gc C;
    C.cmd = 1;
    C.p = st.p; C.W = w.W; C.ld = w.ld; C.bsz = w.bsz; C.bs = w.bs; C.be = w.be;
    C.j0 = j0; C.j1 = j1; C.slot0 = slot0;
    C.ifirstm = st.ifirstm; C.ilastm = st.ilastm;
    C.wboff = (int)((char*)w.b - (char*)psd_lds);
    C.taboff = P.gtaboff;
    C.tr = P.tr;
    C.S = P.S;
    PSD_SYNC();
#ifdef PSD_HOSTSIM
    psd_gs3_run(C, 0, 1);
#else
    psd_gc* cmd = (psd_gc*)(psd_lds + P.gcoff);
    PSD_ONE { *cmd = C; }
    PSD_PAIR_BARRIER();
    psd_gs3_run(C, 0, (int)blockDim.y);
    PSD_ONE { cmd->cmd = 0; }
#endif
}

// One window (positions qs..qe of column hj, processed downwards) of stage 2; blockDim = 64 G, L links per wave
// (psd_ghess_waves / psd_ghess_links).  Same state, lists and descriptor as psd_gq_hess_window.
PSD_D void psd_gq_hess_step_body(const psd_gparams& P, int L, int scan) {
    PSD_LDS_DECL;
    psd_gstate st = *P.st;
    PSD_ONE { P.desc->active = 0; P.desc->defer_run = 0; }
    if (st.phase != PSD_GPH_HESS) return;
    const long long tk0 = psd_clock(), tw0 = psd_wallclock();
    const int G = PSD_NTHREADS >> 6;
    const int n = st.n, p = st.p, hj = st.hj;
    double* ldsd = (double*)psd_lds;
    double* side = ldsd + (size_t)p * st.W * (st.W + 1);
    double* mail = side + 64;
    int* lcnt = (int*)(mail + 4 * (PSD_GHESS_MAXWAVES + 1));
    const int nb = st.W - 1;
    const int qe = st.kcur;
    const int qs = (qe - nb + 1 > hj + 1) ? (qe - nb + 1) : (hj + 1);
    const int K = qe - qs + 1;
    psd_gwin w;
    w.b = ldsd;
    w.W = st.W;
    w.ld = st.W + 1;
    w.bsz = st.W * (st.W + 1);
    w.bs = qs;
    w.be = qe + 1;
    const psd_mat<double> A1 = psd_gfac(P, n, 1);
    PSD_PAR_FOR(m, p) { lcnt[m] = K; }
    PSD_PAR_FOR(t, w.be - w.bs + 1) { side[t] = A1(w.bs + t, hj); }
    PSD_WAVES_FOR(g, G) { psd_wv::psd_gwin_load(P, w, n, p, g, G); }
    const long long tk1 = psd_clock();
    if (scan && p <= 64) {  // scan form: K rounds of (chain of pairs, two update phases)
        double* tab = (double*)((char*)psd_lds + ((((size_t)((char*)(lcnt + p) - (char*)psd_lds)) + 15) & ~(size_t)15));
        PSD_SYNC();
        psd_ghess_scan_window(P, st, w, side, tab, qe, K);
    } else {
        for (int b = 0; b < K + G; ++b) {
            PSD_WAVES_FOR(g, G) { psd_wv::psd_ghess_beat(P, st, w, side, mail, g, G, b, K, L, qe); }
            PSD_SYNC();
        }
    }
    const long long tk2 = psd_clock();
    PSD_WAVES_FOR(g, G) { psd_wv::psd_gwin_store(P, w, n, p, g, G); }
    PSD_PAR_FOR(t, w.be - w.bs + 1) { A1(w.bs + t, hj) = side[t]; }
    psd_gdesc_write(P, st, lcnt, w.bs, w.be, w.be + 1, n, 1, w.bs - 1, 0, 0, 0, 0, 1, hj + 1);
    st.cyc[1] += tk1 - tk0;  // state + window load
    st.cyc[2] += tk2 - tk1;  // the K positions
    st.cyc[3] += psd_clock() - tk2;  // window store, descriptor
    st.nwindows += 1;
    st.kcur = qs - 1;
    if (st.kcur < hj + 1) {
        st.hj = hj + 1;
        st.kcur = n - 1;
        if (st.hj > n - 2) st.phase = PSD_GPH_DONE;
    }
    st.cyc[4] += psd_clock() - tk0;
    st.cyc[5] += psd_wallclock() - tw0;
    if (st.info == PSD_LIST_OVERFLOW) st.phase = PSD_GPH_DONE;  // (a window that overran a list ends the call)
    PSD_SYNC();
    PSD_ONE { *P.st = st; }
}

PSD_KERNEL_B(64 * PSD_GHESS_MAXWAVES) psd_gq_hess_step(psd_gparams P, int L, int scan) { psd_gq_hess_step_body(P, L, scan); }
// the scan form runs four wavefronts: its own entry, so that the register budget is that of 256 threads, not of 1024
PSD_KERNEL_B(256) psd_gq_hess_step_scan(psd_gparams P, int L) { psd_gq_hess_step_body(P, L, 1); }

// rgeneralized.jl:890-1054: one window of the implicit double-shift sweep
PSD_D void psd_gq_sweep_window(const psd_gparams& P, psd_gstate& st, double* ldsd, int* lcnt) {
    const int n = st.n, p = st.p, ifirst = st.ifirst, ilast = st.ilast, ifirstm = st.ifirstm, ilastm = st.ilastm;
    const int ks = st.kcur;
    const bool first = (ks == ifirst);
    int ke = first ? (ifirst + ((st.cursor > 0 && st.cfirst > 0) ? st.cfirst : (st.W - 3)) - 1) : (ks + st.W - 5);  // (a cursor's first window: its part of the schedule)
    if (ke > ilast - 2) ke = ilast - 2;
    psd_gwin w;
    w.b = ldsd;
    w.W = st.W;
    w.ld = st.W + 1;
    w.bsz = st.W * (st.W + 1);
    w.bs = first ? ifirst : (ks - 1);
    w.be = (ke + 3 < ilast) ? (ke + 3) : ilast;
    const long long tc0 = psd_clock();
    PSD_PAR_FOR(m, p) { lcnt[m] = 0; }
    psd_gwin_load(P, w, n, p);
    const long long tc1 = psd_clock();
    int jstart = ks;
    int slot = 0;  // every owner receives the same number of list entries: 2 (initial pass), 2 per position, 1 (tail)
    if (first && p > 1) {
        // initial transformation (:890-943): the shift rotations enter H_1 from the right and travel forward
        const int j = ifirst;
        double c2 = st.c2, s2 = st.s2, c1 = st.c1, s1 = st.s1;
        const int own = psd_gnext(1, p);
        psd_gwin_right(w, 1, j + 1, c2, s2, ifirstm, ilast);
        psd_gwin_right(w, 1, j, c1, s1, ifirstm, ilast);
        psd_gstore_tr(P, own, 0, j + 1, c2, s2);
        psd_gstore_tr(P, own, 1, j, c1, s1);
        for (int l = 2; l <= p; ++l) {
            const bool sg = psd_gsig(P, l);
            const int ownl = psd_gnext(l, p);
            psd_g_link(w, l, j + 1, !sg, c2, s2, ifirstm, ilastm, P.tr, ownl, 0);
            psd_g_link(w, l, j, !sg, c1, s1, ifirstm, ilastm, P.tr, ownl, 1);
        }
        psd_gwin_left(w, 1, j + 1, c2, s2, ifirst, ilastm);
        psd_gwin_left(w, 1, j, c1, s1, ifirst, ilastm);
        jstart = ifirst + 1;
        slot = 2;
    }
    const bool scan = P.gtaboff != 0 && p >= 2 && p <= 64 && jstart <= ke;
    if (scan) {  // scan form: both chains of a position before any update (psd_gs3_run)
        psd_gs3_lead(P, st, w, jstart, ke, slot);
        slot += 2 * (ke - jstart + 1);
    }
    for (int j = jstart; j <= ke && !scan; ++j) {
        double c1, s1, c2, s2;
        if (first && p == 1 && j == ifirst) {  // :955-958
            c1 = st.c1; s1 = st.s1; c2 = st.c2; s2 = st.s2;
        } else {  // :960-968
            double r2, r1;
            psd_givens(w.at(1, j + 1, j - 1), w.at(1, j + 2, j - 1), c2, s2, r2);
            psd_givens(w.at(1, j, j - 1), r2, c1, s1, r1);
            PSD_WAVE_SYNC();
            PSD_ONE {
                w.at(1, j, j - 1) = r1;
                w.at(1, j + 1, j - 1) = 0.0;
                w.at(1, j + 2, j - 1) = 0.0;
            }
            PSD_WAVE_SYNC();
        }
        psd_gwin_left(w, 1, j + 1, c2, s2, j, ilastm);
        psd_gwin_left(w, 1, j, c1, s1, j, ilastm);
        psd_gstore_tr(P, 1, slot, j + 1, c2, s2);
        psd_gstore_tr(P, 1, slot + 1, j, c1, s1);
        for (int l = p; l >= 2; --l) {
            const bool sg = psd_gsig(P, l);
            psd_g_link(w, l, j + 1, sg, c2, s2, ifirstm, ilastm, P.tr, l, slot);
            psd_g_link(w, l, j, sg, c1, s1, ifirstm, ilastm, P.tr, l, slot + 1);
        }
        const int lm = (j + 3 < ilastm) ? (j + 3) : ilastm;
        psd_gwin_right(w, 1, j + 1, c2, s2, ifirstm, lm);
        psd_gwin_right(w, 1, j, c1, s1, ifirstm, lm);
        slot += 2;
    }
    const bool last = ke >= ilast - 2;
    if (last) {
        psd_gq_tail(P, st, w, slot, ilast - 1, 0, 0.0, 0.0, nullptr);
        slot += 1;
    }
    PSD_WAVE_SYNC();
    PSD_PAR_FOR(m, p) { lcnt[m] = slot; }
    const long long tc2 = psd_clock();
    psd_gwin_store(P, w, n, p);
    st.cyc[1] += tc1 - tc0;
    st.cyc[2] += tc2 - tc1;
    st.cyc[3] += psd_clock() - tc2;
    psd_gdesc_write(P, st, lcnt, w.bs, w.be, w.be + 1, ilastm, ifirstm, w.bs - 1, 0, 0, 0, 0);
    st.nwindows += 1;
    st.kcur = ke + 1;
    if (last) st.phase = (st.cursor > 0) ? PSD_GPH_CDONE : ((st.train_n > 1) ? PSD_GPH_TWAIT : PSD_GPH_CHECK);
}

// rgeneralized.jl:229-324: one window of the controlled zero shift (positions kcur..)
PSD_D void psd_gq_zshift_window(const psd_gparams& P, psd_gstate& st, double* ldsd, int* lcnt) {
    const int n = st.n, p = st.p, jlo = st.jlo, ilast = st.ilast, ifirstm = st.ifirstm, ilastm = st.ilastm;
    const int nb = st.W - 2;
    const int ks = st.kcur;
    const int jend = ilast - 1;
    const int ke = (ks + nb - 1 < jend) ? (ks + nb - 1) : jend;
    psd_gwin w;
    w.b = ldsd;
    w.W = st.W;
    w.ld = st.W + 1;
    w.bsz = st.W * (st.W + 1);
    w.bs = ks;
    w.be = ke + 1;
    PSD_PAR_FOR(m, p) { lcnt[m] = 0; }
    psd_gwin_load(P, w, n, p);
    for (int j = ks; j <= ke; ++j) {
        double c, s, r;
        psd_givens(w.at(1, j, j), w.at(1, j + 1, j), c, s, r);
        psd_gwin_set2(w, 1, j, j, r, j + 1, j, 0.0);
        psd_gwin_left(w, 1, j, c, s, j + 1, ilastm);
        psd_grecord(P, lcnt, 1, j, c, s);
        for (int l = p; l >= 2 && s != 0.0; --l) {
            const bool sg = psd_gsig(P, l);
            if (sg) psd_gwin_right(w, l, j, c, s, ifirstm, j + 1);
            else psd_gwin_left(w, l, j, c, s, j, ilastm);
            double tol = fabs(w.at(l, j, j)) + fabs(w.at(l, j + 1, j + 1));
            if (tol == 0) {  // opnorm(view(Hl, jlo:j+1, jlo:j+1), 1) restricted to the window
                for (int cc = w.bs; cc <= j + 1; ++cc) {
                    double cs = 0.0;
                    for (int rr = w.bs; rr <= j + 1; ++rr) cs += fabs(w.at(l, rr, cc));
                    tol = fmax(tol, cs);
                }
            }
            tol = fmax(st.ulp * tol, st.smlnum);
            const double sub = w.at(l, j + 1, j);
            if (fabs(sub) <= tol) {
                c = 1.0;
                s = 0.0;
                psd_gwin_set2(w, l, j + 1, j, 0.0, j + 1, j, 0.0);
            } else if (sg) {
                psd_givens(w.at(l, j, j), sub, c, s, r);
                psd_gwin_set2(w, l, j, j, r, j + 1, j, 0.0);
                psd_gwin_left(w, l, j, c, s, j + 1, ilastm);
                psd_grecord(P, lcnt, l, j, c, s);
            } else {
                psd_givens(w.at(l, j + 1, j + 1), -sub, c, s, r);
                psd_gwin_set2(w, l, j + 1, j + 1, r, j + 1, j, 0.0);
                psd_gwin_right(w, l, j, c, s, ifirstm, j);
                psd_grecord(P, lcnt, l, j, c, s);
            }
        }
        PSD_ONE {  // the right side of H_1 is applied after the whole pass (:312-320)
            psd_gtr tr;
            tr.pos = j;
            tr.pad = 0;
            tr.c = c;
            tr.s = s;
            P.dG[j] = tr;
        }
        if (s == 0.0) st.zflag = 1;
    }
    psd_gwin_store(P, w, n, p);
    const bool last = ke >= jend;
    psd_gdesc_write(P, st, lcnt, ks, ke + 1, w.be + 1, ilastm, ifirstm, w.bs - 1, 1, last ? 1 : 0, jlo, jend);
    st.nwindows += 1;
    st.kcur = ke + 1;
    if (last) {
        st.ziter = st.zflag ? 1 : 0;
        st.phase = PSD_GPH_CHECK;
    }
}

// first (smallest l) factor of the given sign with a negligible diagonal entry in jlo..ilast, largest j
// (rgeneralized.jl:198-226, 1114-1136); returns l * (n + 2) + (n + 1 - j) or INT_MAX
PSD_D int psd_gq_scan_diag(const psd_gparams& P, const psd_gstate& st, int* redi, int jlo, bool sign) {
    const int n = st.n, p = st.p, ilast = st.ilast;
    const int NT = PSD_NTHREADS;
    const int wd = ilast - jlo + 1;
    PSD_SYNC();
    PSD_PAR_FOR(t, NT) {
        int key = 0x7fffffff;
        for (int q = t; q < (p - 1) * wd; q += NT) {
            const int l = 2 + q / wd, j = jlo + q % wd;
            if (psd_gsig(P, l) != sign) continue;
            const psd_mat<double> Hl = psd_gfac(P, n, l);
            double tol;
            if (j == ilast) tol = fabs(Hl(j - 1, j));
            else if (j == jlo) tol = fabs(Hl(j, j + 1));
            else tol = fabs(Hl(j - 1, j)) + fabs(Hl(j, j + 1));
            tol = fmax(st.ulp * tol, st.smlnum);  // (tol == 0 fallback of the reference: smlnum floor)
            if (fabs(Hl(j, j)) <= tol) {
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

// rgeneralized.jl:169-803: deflation tests, split, zero-shift decision, 2x2 blocks, starting rotations.
// Returns true when a window was emitted (2x2 real deflation).
// ------------------------------------------------------------------------------------------------
// Explicit shifts for the signed double-shift sweep (multishift trains).  The sweep's starting
// rotations act on the columns of H_1, i.e. they are a similarity of P' = T H_1 with T = prod_{l=2..p} H_l^{s_l} (upper
// triangular; a diagonal block of T is the product of the factors' diagonal blocks, inverted where s_l is false).

// cur <- prod_{l=2..p} (H_l[r0:r0+K-1, r0:r0+K-1])^{s_l}; B: LDS staging [p][K][K], R0/R1: K x K each.  Called by every
// lane; returns the buffer that holds the product.  A zero diagonal entry of an inverted factor gives non-finite entries
// (the callers test their results).
PSD_D double* psd_gq_tprod(const psd_gparams& P, int n, int p, int r0, int K, double* B, double* R0, double* R1) {
    const int KK = K * K;
    PSD_SYNC();
    PSD_PAR_FOR(t, p * KK) {
        const int j = t / KK, q = t - j * KK, r = q / K, c = q - r * K;
        B[t] = psd_gfac(P, n, j + 1)(r0 + r, r0 + c);
    }
    PSD_PAR_FOR(q, KK) { R0[q] = (q / K == q % K) ? 1.0 : 0.0; }
    PSD_SYNC();
    double* cur = R0;
    double* nxt = R1;
    for (int j = 2; j <= p; ++j) {
        const double* Bj = B + (size_t)(j - 1) * KK;
        if (psd_gsig(P, j)) {
            PSD_PAR_FOR(q, KK) {
                const int r = q / K, c = q - r * K;
                double acc = 0.0;
                for (int k = r; k <= c; ++k) acc += cur[r * K + k] * Bj[k * K + c];
                nxt[q] = acc;
            }
        } else {  // X B_j = cur, row by row (one lane per row)
            PSD_PAR_FOR(r, K) {
                for (int c = 0; c < K; ++c) {
                    double x = 0.0;
                    if (c >= r) {
                        x = cur[r * K + c];
                        for (int k = r; k < c; ++k) x -= nxt[r * K + k] * Bj[k * K + c];
                        x /= Bj[c * K + c];
                    }
                    nxt[r * K + c] = x;
                }
            }
        }
        PSD_SYNC();
        double* sw = cur;
        cur = nxt;
        nxt = sw;
    }
    return cur;
}

// starting rotations of a double-shift sweep on the active block ifirst.. for the shift pair sh: first column of
// (P' - s1)(P' - s2) from the leading 3x3 blocks; false if not finite.  `work`: LDS, >= 9 p + 18 doubles.
PSD_D bool psd_gq_start_explicit(const psd_gparams& P, int n, int p, int ifirst, const double* sh, double* work, double& c1,
                                 double& s1, double& c2, double& s2) {
    double* T3 = psd_gq_tprod(P, n, p, ifirst, 3, work, work + 9 * (size_t)p, work + 9 * (size_t)p + 9);
    const psd_mat<double> H1 = psd_gfac(P, n, 1);
    const double h11 = H1(ifirst, ifirst), h21 = H1(ifirst + 1, ifirst), h12 = H1(ifirst, ifirst + 1),
                 h22 = H1(ifirst + 1, ifirst + 1), h32 = H1(ifirst + 2, ifirst + 1);
    const double t11 = T3[0], t12 = T3[1], t13 = T3[2], t22 = T3[4], t23 = T3[5], t33 = T3[8];
    const double a1 = t11 * h11 + t12 * h21, a2 = t22 * h21;             // P' e1
    const double b1 = h11 * a1 + h12 * a2, b2 = h21 * a1 + h22 * a2, b3 = h32 * a2;  // H_1 (P' e1)
    const double q1 = t11 * b1 + t12 * b2 + t13 * b3, q2 = t22 * b2 + t23 * b3, q3 = t33 * b3;  // P'^2 e1
    const double tr = sh[0] + sh[2], det = sh[0] * sh[2] - sh[1] * sh[3];
    double v1 = q1 - tr * a1 + det, v2 = q2 - tr * a2, v3 = q3;
    const double sc = fabs(v1) + fabs(v2) + fabs(v3);
    PSD_SYNC();
    if (!(sc > 0.0) || !(sc < 1e300)) return false;
    v1 /= sc;
    v2 /= sc;
    v3 /= sc;
    double r, rr;
    psd_givens(v2, v3, c2, s2, r);
    psd_givens(v1, r, c1, s1, rr);
    return true;
}

// the 2m shifts of a train: eigenvalues of the trailing K x K block (K = 2m) of P' = T H_1; pairs to P.tshift, *okf = 1 on
// success.  `work`: LDS, psd_gq_train_doubles(p, m) doubles.
PSD_HD size_t psd_gq_train_doubles(int p, int m) {
    const size_t K = 2 * (size_t)m;
    return (size_t)p * K * K + 3 * K * K + 3 * PSD_HQR_MAX + 8;
}
PSD_D void psd_gq_train_shifts(const psd_gparams& P, int n, int p, int ilast, int m, double* work, int* okf) {
    const int K = 2 * m, KK = K * K, t0 = ilast - K + 1;
    double* B = work;
    double* R0 = B + (size_t)p * KK;
    double* R1 = R0 + KK;
    double* T = R1 + KK;
    double* wr = T + KK;
    double* wi = wr + PSD_HQR_MAX;
    double* re = wi + PSD_HQR_MAX;
    const double* Tt = psd_gq_tprod(P, n, p, t0, K, B, R0, R1);
    PSD_PAR_FOR(q, KK) {  // T = Tt * H_1[t0.., t0..]  (B's first block is H_1's)
        const int r = q / K, c = q - r * K;
        double acc = 0.0;
        const int kmax = (c + 1 < K - 1) ? (c + 1) : (K - 1);
        for (int k = r; k <= kmax; ++k) acc += Tt[r * K + k] * B[k * K + c];
        T[q] = acc;
    }
    PSD_SYNC();
#ifndef PSD_HOSTSIM
    // (one lane alone pays an LDS round trip per operand — 1.7 ms per train at K = 16, psd_hqr.h; the workgroup is one wavefront)
    bool finite = true;
    for (int q = PSD_TID; q < KK; q += 64)
        if (!(fabs(T[q]) < 1e300)) finite = false;
    finite = __all(finite) != 0;
    const bool okw = finite && psd_hqr_wave(T, K, K, wr, wi, PSD_TID);
#endif
    PSD_ONE {
#ifndef PSD_HOSTSIM
        bool ok = okw;
#else
        bool ok = true;
        for (int q = 0; q < KK; ++q)
            if (!(fabs(T[q]) < 1e300)) ok = false;
        ok = ok && psd_hqr(T, K, K, wr, wi);
#endif
        int np = 0, nre = 0;
        for (int q = 0; ok && q < K; ++q) {
            if (!(wr[q] == wr[q]) || !(wi[q] == wi[q])) ok = false;
            if (wi[q] > 0.0) {
                if (np < m) {
                    double* sh = P.tshift + 4 * np;
                    sh[0] = wr[q]; sh[1] = wi[q]; sh[2] = wr[q]; sh[3] = -wi[q];
                    ++np;
                }
            } else if (wi[q] == 0.0) {
                re[nre++] = wr[q];
            }
        }
        for (int a = 1; a < nre; ++a) {
            const double x = re[a];
            int b = a - 1;
            while (b >= 0 && re[b] > x) {
                re[b + 1] = re[b];
                --b;
            }
            re[b + 1] = x;
        }
        for (int a = 0; ok && a < nre && np < m; a += 2) {
            double* sh = P.tshift + 4 * np;
            sh[0] = re[a]; sh[1] = 0.0; sh[2] = (a + 1 < nre) ? re[a + 1] : re[a]; sh[3] = 0.0;
            ++np;
        }
        *okf = (ok && np == m) ? 1 : 0;
    }
    PSD_SYNC();
}

PSD_D bool psd_gq_check(const psd_gparams& P, psd_gstate& st, double* ldsd, double* red, int* redi, int* lcnt) {
    const int n = st.n, p = st.p;
    const int NT = PSD_NTHREADS;
    st.jiter += 1;
    if (st.jiter > st.maxit) {  // :1057-1059
        st.info = st.ilast;
        st.phase = PSD_GPH_DONE;
        return false;
    }
    const psd_mat<double> H1 = psd_gfac(P, n, 1);
    const int ilast = st.ilast;
    bool split = false;
    int jlo = 1;
    long long tq = psd_clock();
#define PSD_GDBG_STAMP(i) do { const long long tn_ = psd_clock(); st.dbg[i] += tn_ - tq; st.dbgn[i] += 1; tq = tn_; } while (0)
    if (ilast == 1) {
        split = true;
    } else {
        // Test 1 (:1086-1112): first negligible subdiagonal of H_1 from the bottom
        PSD_SYNC();
        PSD_PAR_FOR(t, NT) {
            int best = 0;
            for (int j = ilast - t; j >= 2; j -= NT) {
                double tol = fabs(H1(j - 1, j - 1)) + fabs(H1(j, j));
                if (tol == 0) {  // opnorm(view(H1, 1:j, 1:j), 1): columns j-1, j carry the only candidates > 0 nearby
                    for (int cc = 1; cc <= j; ++cc) {
                        double cs = 0.0;
                        const int rmax = (cc + 1 < j) ? (cc + 1) : j;
                        for (int rr = 1; rr <= rmax; ++rr) cs += fabs(H1(rr, cc));
                        tol = fmax(tol, cs);
                    }
                }
                tol = fmax(st.ulp * tol, st.smlnum);
                if (fabs(H1(j, j - 1)) <= tol) {
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
            PSD_ONE { H1(jfound, jfound - 1) = 0.0; }
            PSD_SYNC();
            jlo = jfound;
            if (jfound == ilast) split = true;
        }
    }
    PSD_GDBG_STAMP(0);
    if (split) {  // :617-642
        double a, b;
        int sc;
        psd_g_safeprod(P, n, p, ilast, a, b, sc);
        PSD_GDBG_STAMP(7);
        PSD_ONE {
            P.alpha[ilast - 1] = zmk(a, 0.0);
            P.beta[ilast - 1] = b;
            P.ascale[ilast - 1] = sc;
        }
        st.nsplit += 1;
        st.ilast -= 1;
        if (st.ilast < 1) {
            st.phase = PSD_GPH_DONE;
            return false;
        }
        if (st.ziter != -1) st.ziter = 0;
        if (!st.wantT) {
            st.ilastm = st.ilast;
            if (st.ifirstm > st.ilast) st.ifirstm = 1;
        }
        return false;
    }
    st.jlo = jlo;
    // Tests 2 and 3 (:198-226)
    const int key2 = psd_gq_scan_diag(P, st, redi, jlo, true);
    const int key3 = (key2 == 0x7fffffff) ? psd_gq_scan_diag(P, st, redi, jlo, false) : 0x7fffffff;
    PSD_GDBG_STAMP(1);
    // Test 4 (:229): controlled zero shift; a pending zero diagonal entry is found again afterwards (the
    // reference's fall-through into Case II/III with stale indices is a defect)
    if (st.ziter >= 7 || st.ziter < 0) {
        st.phase = PSD_GPH_ZSHIFT;
        st.kcur = jlo;
        st.zflag = 0;
        st.nzshift += 1;
        psd_glog(P, st, 4, jlo, ilast);
        return false;
    }
    if (key2 != 0x7fffffff || key3 != 0x7fffffff) {
        const int key = (key2 != 0x7fffffff) ? key2 : key3;
        const int l = key / (n + 2), j = (n + 1) - key % (n + 2);
        PSD_ONE { psd_gfac(P, n, l)(j, j) = 0.0; }
        PSD_SYNC();
        if (key2 != 0x7fffffff) psd_gq_case2(P, st, l, j);
        else psd_gq_case3(P, st, l, j);
        return false;
    }
    // QZ step (:644-803)
    st.ifirst = jlo;
    st.ziter += 1;
    if (!st.wantT) st.ifirstm = st.ifirst;
    const int ifirst = st.ifirst;
    if (ifirst + 1 == ilast) {  // 2x2 block (:661-790)
        const int j = ilast - 1;
        PSD_SYNC();
        PSD_ONE {
            double* X = ldsd;  // (in LDS: one lane walks the 2x2 blocks up to 20 + 80 times; from global scratch that was 1.4 ms per block at p = 32)
            for (int l = 1; l <= p; ++l) {  // order 2, 3, ..., p, 1 (:669-672)
                const psd_mat<double> M = psd_gfac(P, n, (l == p) ? 1 : (l + 1));
                X[4 * (l - 1) + 0] = M(j, j);
                X[4 * (l - 1) + 1] = M(j, j + 1);
                X[4 * (l - 1) + 2] = M(j + 1, j);
                X[4 * (l - 1) + 3] = M(j + 1, j + 1);
            }
            bool done2 = false;
            // A block whose product has a clearly complex pair cannot be brought to real triangular form: the two
            // attempts below (up to 40 single-shift passes over the p factors by one lane, 1.2 ms at p = 32) would end
            // with done2 = false and their X discarded (:748-771 reloads the blocks).  The sign of the discriminant of
            // the explicitly formed 2x2 product (inverses by adjugates: a scalar multiple does not change the sign;
            // rescaled at every factor) tells; anything near zero or not finite takes the reference's route.
            bool clearly_complex = false;
            {
                double m11 = X[4 * (p - 1) + 0], m12 = X[4 * (p - 1) + 1], m21 = X[4 * (p - 1) + 2], m22 = X[4 * (p - 1) + 3];
                bool okp = true;
                for (int l = 2; l <= p && okp; ++l) {
                    const double* Y = X + 4 * (l - 2);
                    double b11, b12, b21, b22;
                    if (psd_gsig(P, l)) {
                        b11 = Y[0]; b12 = Y[1]; b21 = Y[2]; b22 = Y[3];
                    } else {
                        b11 = Y[3]; b12 = -Y[1]; b21 = -Y[2]; b22 = Y[0];
                    }
                    const double n11 = m11 * b11 + m12 * b21, n12 = m11 * b12 + m12 * b22;
                    const double n21 = m21 * b11 + m22 * b21, n22 = m21 * b12 + m22 * b22;
                    const double sc = fmax(fmax(fabs(n11), fabs(n12)), fmax(fabs(n21), fabs(n22)));
                    if (!(sc > 0.0) || !(sc < 1e300)) okp = false;
                    const double rs = 1.0 / sc;
                    m11 = n11 * rs; m12 = n12 * rs; m21 = n21 * rs; m22 = n22 * rs;
                }
                if (okp) {
                    const double hd = 0.5 * (m11 - m22), od = m12 * m21;
                    const double disc = hd * hd + od;
                    clearly_complex = disc < -1e-6 * fmax(hd * hd, fabs(od));
                }
            }
            for (int titer = 1; titer <= 2 && !done2 && !clearly_complex; ++titer) {
                psd_g_rp2x2ssr(p, X, P);
                const double* Xp = X + 4 * (p - 1);
                done2 = fabs(Xp[2]) < PSD_DBL_EPS * fmax(fabs(Xp[0]), fmax(fabs(Xp[1]), fabs(Xp[3])));
            }
            red[0] = done2 ? 1.0 : 0.0;
            if (done2) {  // perfect-shift rotation (:694-709)
                double c1 = 1.0, s1 = 1.0, r;
                for (int l = p; l >= 2; --l) {
                    const double rr = X[4 * (l - 2) + 3];
                    const double hjj = psd_gfac(P, n, l)(j, j);
                    if (psd_gsig(P, l)) psd_givens(c1 * hjj, s1 * rr, c1, s1, r);
                    else psd_givens(c1 * rr, s1 * hjj, c1, s1, r);
                }
                const double rr = X[4 * (p - 1) + 3];
                psd_givens(c1 * H1(j, j) - rr * s1, c1 * H1(j + 1, j), c1, s1, r);
                red[1] = c1;
                red[2] = s1;
            } else {  // conjugate pair (:748-771)
                psd_z* Xc = (psd_z*)(ldsd + 4 * p);
                for (int l = 1; l <= p; ++l) {
                    const psd_mat<double> M = psd_gfac(P, n, l);
                    Xc[4 * (l - 1) + 0] = zmk(M(j, j), 0.0);
                    Xc[4 * (l - 1) + 1] = zmk(M(j, j + 1), 0.0);
                    Xc[4 * (l - 1) + 2] = zmk(M(j + 1, j), 0.0);
                    Xc[4 * (l - 1) + 3] = zmk(M(j + 1, j + 1), 0.0);
                }
                psd_z a2[2];
                double b2[2], sc2[2];
                bool cvg, good;
                psd_g_eigpair(P, p, Xc, a2, b2, sc2, cvg, good);
                for (int q = 0; q < 2; ++q) {
                    P.alpha[j - 1 + q] = a2[q];
                    P.beta[j - 1 + q] = b2[q];
                    P.ascale[j - 1 + q] = (int)sc2[q];
                }
                red[1] = cvg ? 1.0 : 0.0;
                red[2] = good ? 1.0 : 0.0;
            }
        }
        PSD_SYNC();
        const bool done2 = red[0] != 0.0;
        const double r1 = red[1], r2 = red[2];
        PSD_SYNC();
        PSD_GDBG_STAMP(6);
        if (done2) {
            st.n2real += 1;
            psd_glog(P, st, 5, j, ilast);
            psd_gwin w;
            w.b = ldsd;
            w.W = st.W;
            w.ld = st.W + 1;
            w.bsz = st.W * (st.W + 1);
            w.bs = j;
            w.be = ilast;
            PSD_PAR_FOR(m, p) { lcnt[m] = 0; }
            psd_gwin_load(P, w, n, p);
            psd_gq_tail(P, st, w, 0, j, 1, r1, r2, nullptr);
            PSD_WAVE_SYNC();
            PSD_PAR_FOR(m, p) { lcnt[m] = 1; }
            psd_gwin_store(P, w, n, p);
            psd_gdesc_write(P, st, lcnt, w.bs, w.be, w.be + 1, st.ilastm, st.ifirstm, w.bs - 1, 0, 0, 0, 0);
            st.nwindows += 1;
            return true;
        }
        st.n2cplx += 1;
        psd_glog(P, st, 6, j, ilast);
        if (r1 == 0.0) st.iwarn = (st.iwarn > j) ? st.iwarn : j;
        else if (r2 == 0.0 && st.iwarn == 0) st.iwarn = n;
        st.ilast = ifirst - 1;
        if (st.ilast < 1) {
            st.phase = PSD_GPH_DONE;
            return false;
        }
        if (st.ziter != -1) st.ziter = 0;
        if (!st.wantT) {
            st.ilastm = st.ilast;
            if (st.ifirstm > st.ilast) st.ifirstm = 1;
        }
        return false;
    }
    PSD_SYNC();
    psd_g_qzrots(P, n, p, ifirst, ilast - ifirst + 1, st.c1, st.s1, st.c2, st.s2, ldsd);
    PSD_GDBG_STAMP(2);
    st.train_n = 1;
    if ((st.train_want >= 2 || st.train_want == -2) && P.tshift != nullptr && ilast - ifirst + 1 >= 4) {
        // window width of the train by the cost model of psd_rq_shift
        const int w = ilast - ifirst + 1;
        int mt = (st.train_want == -2) ? 1 : st.train_want;
        if (mt > PSD_TRAIN_MAX) mt = PSD_TRAIN_MAX;
        int nb = st.Wmax - 4, m = 1;
        double best = 1e300;
        for (int nbc = (st.Wmax - 4 < 8) ? ((st.Wmax > 5) ? st.Wmax - 4 : 1) : 8; nbc <= st.Wmax - 4 && mt >= 2; ++nbc) {
            // (cursors nbc + 4 = W positions apart — their windows only have to be disjoint; two windows apart before)
            int mc = 1 + (w - nbc) / (nbc + 4);
            if (mc > mt) mc = mt;
            if (mc < 2) break;
            const double cost = (double)((w + nbc - 1) / nbc + ((mc - 1) * (nbc + 4) + nbc - 1) / nbc) * (double)(nbc * p + st.train_oc) / mc;
            if (cost < best) {
                best = cost;
                nb = nbc;
                m = mc;
            }
        }
        // (a train longer than PSD_HQR_MAX / 2 runs through its shift pairs twice, as in psd_rq_shift)
        int ms = (2 * m > PSD_HQR_MAX) ? PSD_HQR_MAX / 2 : m;
        while (ms >= 1 && psd_gq_train_doubles(p, ms) > (size_t)p * st.Wmax * (st.Wmax + 1)) --ms;
        if (ms < m && ms < PSD_HQR_MAX / 2) m = ms;
        if ((m >= 2 || st.train_want == -2) && m >= 1 && 2 * m + 2 <= w) {
            int* okf = (int*)P.tshift + 8 * PSD_TRAIN_MAX;
            psd_gq_train_shifts(P, n, p, ilast, ms, ldsd, okf);
            PSD_GDBG_STAMP(3);
            if (*okf && m > ms) {
                PSD_ONE {
                    for (int b = ms; b < m; ++b)
                        for (int q = 0; q < 4; ++q) P.tshift[4 * b + q] = P.tshift[4 * (b - ms) + q];
                }
                PSD_SYNC();
            }
            double e1, f1, e2, f2;
            if (*okf && psd_gq_start_explicit(P, n, p, ifirst, P.tshift, ldsd, e1, f1, e2, f2)) {
                st.c1 = e1; st.s1 = f1; st.c2 = e2; st.s2 = f2;
                if (m >= 2) {
                    st.W = nb + 4;
                    st.train_n = m;
                    st.train_tick0 = P.tick;
                    st.train_id += 1;
                    st.ntrainsweeps += m;
                }
            }
            PSD_SYNC();
            PSD_GDBG_STAMP(4);
        }
    }
    st.phase = PSD_GPH_SWEEP;
    st.kcur = ifirst;
    st.nsweeps += 1;
    psd_glog(P, st, 0, ifirst, ilast);
    if (st.train_n > 1) {
        for (int b = 1; b < st.train_n; ++b) psd_glog(P, st, 0, ifirst, ilast);
        PSD_SYNC();
        PSD_ONE {
            psd_atomic_store(P.cep + PSD_TRAIN_MAX, 0);  // finished cursors of this train
            for (int b = 1; b < st.train_n; ++b) {
                psd_gstate cs = st;
                cs.cursor = b;
                {   // bulge b follows b W positions behind the leader, whose window starts at ifirst + 1 + d nb in tick tick0 + d (its
                    // first window has nb + 1 positions): b's first window, D ticks after the leader's, has f positions
                    const int nbw = st.W - 4, spc = st.W;
                    const int D = (b * spc + nbw - 1) / nbw - 1;  // ceil(b W / nb) - 1
                    cs.cstart = st.train_tick0 + D;
                    cs.cfirst = 1 + (D + 1) * nbw - b * spc;      // 1 .. nb
                }
                cs.phase = PSD_GPH_CWAIT;
                for (int q = 0; q < 4; ++q) cs.sh[q] = P.tshift[4 * b + q];
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
        PSD_GDBG_STAMP(5);
    }
    return false;
}

PSD_D void psd_gq_step_body(const psd_gparams& P) {
    PSD_LDS_DECL;
    psd_gstate st = *P.st;
    if (st.phase == PSD_GPH_DONE) {
        PSD_ONE { P.desc->active = 0; P.desc->defer_run = 0; }
        return;
    }
    const int NT = PSD_NTHREADS;
    double* ldsd = (double*)psd_lds;
    const size_t winb = (size_t)st.p * st.Wmax * (st.Wmax + 1);
    double* red = ldsd + winb;
    int* redi = (int*)(red + NT);
    int* lcnt = redi + 2 * NT;
    PSD_ONE { P.desc->active = 0; P.desc->defer_run = 0; }
    const long long tk0 = psd_clock(), tw0 = psd_wallclock();
    bool emitted = false;
    int guard = 0;
    while (!emitted && st.phase != PSD_GPH_DONE && guard < 64) {
        ++guard;
        if (st.phase == PSD_GPH_CHECK) {
            const long long td0 = psd_clock();
            const int nc = st.ncase2 + st.ncase3;
            emitted = psd_gq_check(P, st, ldsd, red, redi, lcnt);
            st.cyc[0] += psd_clock() - td0;
            if (st.ncase2 + st.ncase3 != nc) break;  // a Case II/III pass ran on HBM: end this launch
        } else if (st.phase == PSD_GPH_SWEEP) {
            psd_gq_sweep_window(P, st, ldsd, lcnt);
            emitted = true;
        } else if (st.phase == PSD_GPH_ZSHIFT) {
            psd_gq_zshift_window(P, st, ldsd, lcnt);
            emitted = true;
        } else if (st.phase == PSD_GPH_HESS) {
            psd_gq_hess_window(P, st, ldsd, red, lcnt);
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

PSD_KERNEL_B(4 * PSD_STEP_NT) psd_gq_step(psd_gparams P) {
    PSD_GS3_ENTER(P);
    psd_gq_step_body(P);
    PSD_GS3_LEAVE(P);
}

// cursor b of a multishift train (see psd_rq_cursor_body): starts 2 b ticks behind the leader with its own shift pair
PSD_D void psd_gq_cursor_body(const psd_gparams& P, int b) {
    PSD_LDS_DECL;
    PSD_ONE { P.desc->active = 0; P.desc->defer_run = 0; }
    psd_gstate st;
    if (!psd_pub_read(P.cep + b, P.tick, P.st, st)) return;  // (published in an earlier launch, not being rewritten)
    if (st.cursor != b) return;
    if (st.phase != PSD_GPH_CWAIT && st.phase != PSD_GPH_SWEEP) return;
    double* ldsd = (double*)psd_lds;
    const size_t winb = (size_t)st.p * st.Wmax * (st.Wmax + 1);
    int* lcnt = (int*)(ldsd + winb + PSD_NTHREADS) + 2 * PSD_NTHREADS;
    if (st.phase == PSD_GPH_CWAIT) {
        if (P.tick < st.cstart) return;
        double c1, s1, c2, s2;
        if (!psd_gq_start_explicit(P, st.n, st.p, st.ifirst, st.sh, ldsd, c1, s1, c2, s2)) {
            st.phase = PSD_GPH_CDONE;  // (not finite: this bulge is dropped)
            PSD_SYNC();
            PSD_ONE {
                *P.st = st;
                psd_release_fence();
                psd_atomic_add(P.cep + PSD_TRAIN_MAX, 1);
            }
            return;
        }
        st.c1 = c1; st.s1 = s1; st.c2 = c2; st.s2 = s2;
        st.kcur = st.ifirst;
        st.phase = PSD_GPH_SWEEP;
    }
    psd_gq_sweep_window(P, st, ldsd, lcnt);
    PSD_SYNC();
    PSD_ONE {
        *P.st = st;
        if (st.phase == PSD_GPH_CDONE) {  // last window: count this cursor in (its state and lists are out first)
            psd_release_fence();
            psd_atomic_add(P.cep + PSD_TRAIN_MAX, 1);
        }
    }
}

// all cursors of a tick in one launch, one workgroup each (as psd_rq_step_train)
PSD_KERNEL_B(4 * PSD_STEP_NT) psd_gq_step_train(psd_gparams P, int p, int cstride) {
    PSD_GS3_ENTER(P);
    const int b = PSD_BLOCK_X;
    if (b == 0) {
        psd_gq_step_body(P);
    } else {
        psd_gparams Q = P;
        Q.st = P.cst + b;
        Q.desc = P.desc + b;
        Q.cnt = P.cnt + (size_t)b * cstride;
        Q.tr = P.tr + (size_t)b * p * PSD_GTR_CAP;
        psd_gq_cursor_body(Q, b);
    }
    PSD_GS3_LEAVE(P);
}

// Bulk application of one window's rotation lists, by factor: grid = (tiles, p factors, 3 roles).
// As in psd_rq_apply a thread keeps its line (<= 32 elements) in registers through the whole list when the list visits
// its positions monotonically (sweeps ascending, Hessenberg stage 2 descending): fully unrolled position loop with
// static register indices, list read ahead from LDS behind two sentinel records.
#define PSD_GTR_LDS_RECS (PSD_GTR_CAP + 2)
#define PSD_GTR_LDS_BYTES (sizeof(psd_gtr) * PSD_GTR_LDS_RECS + 16)
template <bool UP>
PSD_D void psd_gtr_regline(const psd_gtr* ltr, int plo, double (&a)[33]) {
    int e = 0;
    psd_gtr cur = ltr[0], nxt = ltr[1];
#pragma unroll
    for (int q = 0; q < 32; ++q) {
        const int b = UP ? q : 31 - q;
        while (cur.pos - plo == b) {
            const double a1 = a[b], a2 = a[b + 1];
            a[b] = cur.c * a1 + cur.s * a2;
            a[b + 1] = cur.c * a2 - cur.s * a1;
            cur = nxt;
            ++e;
            nxt = ltr[e + 1 < PSD_GTR_LDS_RECS ? e + 1 : PSD_GTR_LDS_RECS - 1];
        }
    }
}
// stages the list (plus sentinels) in LDS and classifies its order: +1 ascending, -1 descending, 0 neither; contains
// block synchronisations
PSD_D int psd_gtr_stage(const psd_gtr* gtr, int cnt, psd_gtr* ltr, int* flags) {
    PSD_PAR_FOR(e, PSD_GTR_LDS_RECS) {
        psd_gtr tr;
        if (e < cnt) {
            tr = gtr[e];
        } else {
            tr.pos = 0x3fffffff;
            tr.pad = 0;
            tr.c = 1.0;
            tr.s = 0.0;
        }
        ltr[e] = tr;
    }
    PSD_ONE { flags[0] = flags[1] = 0; }
    PSD_SYNC();
    PSD_PAR_FOR(e, cnt - 1) {
        if (ltr[e + 1].pos < ltr[e].pos) flags[0] = 1;
        if (ltr[e + 1].pos > ltr[e].pos) flags[1] = 1;
    }
    PSD_SYNC();
    return !flags[0] ? 1 : (!flags[1] ? -1 : 0);
}

PSD_D void psd_gq_apply_body(const psd_gparams& P, int n, int p, int role) {
    PSD_LDS_DECL;
    const psd_gapply_desc d = *P.desc;
    if (!d.active) return;
    const int l = PSD_BLOCK_Y + 1;
    const int own = (role == 0) ? psd_growner(P, l, p) : (role == 1) ? psd_gcowner(P, l, p) : l;
    const int cnt = P.cnt[own - 1] < PSD_GTR_CAP ? P.cnt[own - 1] : PSD_GTR_CAP;
    if (cnt <= 0) return;
    const int T = PSD_GAPPLY_NT;
    const int S = d.phi - d.plo + 1;
    psd_gtr* ltr = (psd_gtr*)psd_lds;
    int* flags = (int*)(psd_lds + sizeof(psd_gtr) * PSD_GTR_LDS_RECS);
    double* tile = (double*)(psd_lds + PSD_GTR_LDS_BYTES);
    const psd_gtr* gtr = P.tr + (size_t)(own - 1) * PSD_GTR_CAP;
    const bool h1x = d.h1mode == 1 && l == 1;  // stage 2 of the signed Hessenberg reduction: H_1 outside the window
    if (role == 0) {
        const int c0 = (h1x ? d.h1c0 : d.lc0) + PSD_BLOCK_X * T;
        if (c0 > d.lc1) return;
        const int nc = (d.lc1 - c0 + 1 < T) ? (d.lc1 - c0 + 1) : T;
        if (h1x && c0 >= d.plo && c0 + nc - 1 <= d.phi) return;
        const psd_mat<double> M = psd_gfac(P, n, l);
        const int ldt = T + 1;
        // rows panel -> LDS (a thread owns a column); eight loads in flight per thread
        PSD_PAR_FOR(t0, 32 * 4) {  // (S <= 32; thread t0 covers row t0 & 31 of the columns (t0 >> 5) + 4 k)
            const int r = t0 & 31, cb = t0 >> 5;
            if (r >= S) continue;
            for (int k0 = 0; k0 < T / 4; k0 += 8) {
                double v[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    const int c = cb + 4 * (k0 + u);
                    v[u] = (c < nc) ? M(d.plo + r, c0 + c) : 0.0;
                }
#pragma unroll
                for (int u = 0; u < 8; ++u) tile[r * ldt + cb + 4 * (k0 + u)] = v[u];
            }
        }
        const int order = psd_gtr_stage(gtr, cnt, ltr, flags);
        PSD_PAR_FOR(c, nc) {
            if (h1x && c0 + c >= d.plo && c0 + c <= d.phi) continue;
            if (order != 0) {
                double a[33];
#pragma unroll
                for (int r = 0; r < 33; ++r) a[r] = (r < S) ? tile[r * ldt + c] : 0.0;
                if (order > 0) psd_gtr_regline<true>(ltr, d.plo, a);
                else psd_gtr_regline<false>(ltr, d.plo, a);
#pragma unroll
                for (int r = 0; r < 32; ++r)
                    if (r < S) tile[r * ldt + c] = a[r];
                continue;
            }
            for (int e = 0; e < cnt; ++e) {
                const psd_gtr tr = ltr[e];
                const int r = tr.pos - d.plo;
                const double a1 = tile[r * ldt + c], a2 = tile[(r + 1) * ldt + c];
                tile[r * ldt + c] = tr.c * a1 + tr.s * a2;
                tile[(r + 1) * ldt + c] = tr.c * a2 - tr.s * a1;
            }
        }
        PSD_SYNC();
        PSD_PAR_FOR(t0, 32 * 4) {
            const int r = t0 & 31, cb = t0 >> 5;
            if (r >= S) continue;
            for (int k0 = 0; k0 < T / 4; k0 += 8) {
                double v[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) v[u] = tile[r * ldt + cb + 4 * (k0 + u)];
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    const int c = cb + 4 * (k0 + u);
                    if (c < nc && !(h1x && c0 + c >= d.plo && c0 + c <= d.phi)) M(d.plo + r, c0 + c) = v[u];
                }
            }
        }
    } else {
        if (role == 1 && d.defer_h1 == 1 && l == 1) return;  // H_1's column updates are deferred (zero-shift pass)
        if (role == 2 && (l < P.zlo || l > P.zhi)) return;  // (another rank's Schur vectors)
        const bool h1r = h1x && role == 1;
        const int lo = (role == 1) ? (h1r ? 1 : d.rr0) : d.zr0;
        const int hi = (role == 1) ? (h1r ? n : d.rr1) : d.zr1;
        const int r0 = lo + PSD_BLOCK_X * T;
        if (r0 > hi) return;
        const int nr = (hi - r0 + 1 < T) ? (hi - r0 + 1) : T;
        double* base = (role == 1) ? P.H : P.Z;
        const psd_mat<double> M = psd_mat<double>{base + (size_t)(l - 1) * n * n, n};
        const int order = psd_gtr_stage(gtr, cnt, ltr, flags);
        if (order != 0) {
            // a thread owns a row of the columns panel: coalesced loads straight into registers, no LDS tile
            PSD_PAR_FOR(r, nr) {
                if (h1r && r0 + r >= d.plo && r0 + r <= d.phi) continue;
                double a[33];
#pragma unroll
                for (int c = 0; c < 33; ++c) a[c] = (c < S) ? M(r0 + r, d.plo + c) : 0.0;
                if (order > 0) psd_gtr_regline<true>(ltr, d.plo, a);
                else psd_gtr_regline<false>(ltr, d.plo, a);
#pragma unroll
                for (int c = 0; c < 32; ++c)
                    if (c < S) M(r0 + r, d.plo + c) = a[c];
            }
            return;
        }
        PSD_PAR_FOR(t, S * T) {
            const int r = t & (T - 1), c = t / T;
            if (r >= nr) continue;
            tile[c * T + r] = M(r0 + r, d.plo + c);
        }
        PSD_SYNC();
        PSD_PAR_FOR(r, nr) {
            if (h1r && r0 + r >= d.plo && r0 + r <= d.phi) continue;
            for (int e = 0; e < cnt; ++e) {
                const psd_gtr tr = ltr[e];
                const int c = tr.pos - d.plo;
                const double a1 = tile[c * T + r], a2 = tile[(c + 1) * T + r];
                tile[c * T + r] = tr.c * a1 + tr.s * a2;
                tile[(c + 1) * T + r] = tr.c * a2 - tr.s * a1;
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

PSD_KERNEL_B(PSD_GAPPLY_NT) psd_gq_apply(psd_gparams P, int n, int p) { psd_gq_apply_body(P, n, p, PSD_BLOCK_Z); }

// bulk updates of all cursors of a tick (as psd_rq_apply_train): pass 0 = rows and Z roles (grid.z = 2 M), pass 1 =
// columns role (grid.z = M)
PSD_KERNEL_B(PSD_GAPPLY_NT) psd_gq_apply_train(psd_gparams P, int n, int p, int cstride, int pass) {
    const int z = PSD_BLOCK_Z;
    const int b = (pass == 0) ? (z >> 1) : z;
    const int role = (pass == 0) ? ((z & 1) ? 2 : 0) : 1;
    psd_gparams Q = P;
    Q.desc = P.desc + b;
    Q.cnt = P.cnt + (size_t)b * cstride;
    Q.tr = P.tr + (size_t)b * p * PSD_GTR_CAP;
    psd_gq_apply_body(Q, n, p, role);
}

// Deferred right side of H_1 after a zero-shift pass (rgeneralized.jl:312-320):
// for j = djlo..djhi: rmul!(view(H1, drow0:(j+1), :), G_j').  One thread per row.
PSD_KERNEL psd_gq_defer(psd_gparams P, int n) {
    const psd_gapply_desc d = *P.desc;
    if (!d.active || d.defer_run != 1) return;
    const psd_mat<double> H1 = psd_mat<double>{P.H, n};
    const int NT = PSD_NTHREADS;
    const int rbase = d.drow0 + PSD_BLOCK_X * NT;
    PSD_PAR_FOR(t, NT) {
        const int r = rbase + t;
        if (r <= d.djhi + 1 && d.djhi >= d.djlo) {
            int j = (r - 1 > d.djlo) ? (r - 1) : d.djlo;
            double a1 = H1(r, j);
            for (; j <= d.djhi; ++j) {
                const psd_gtr g = P.dG[j];
                const double a2 = H1(r, j + 1);
                H1(r, j) = g.c * a1 + g.s * a2;
                a1 = g.c * a2 - g.s * a1;
            }
            H1(r, d.djhi + 1) = a1;
        }
    }
}

PSD_KERNEL psd_gq_init(psd_gparams P, int n, int p, int wantT, int wantZ, int W, int maxitfac, int maxlog,
                       int hessmode, int train_want, int train_oc) {
    const psd_mat<double> H1 = psd_mat<double>{P.H, n};
    if (!hessmode) PSD_PAR_FOR(c, n) {
        for (int r = c + 3; r <= n; ++r) H1(r, c + 1) = 0.0;  // _gethess!
        for (int l = 2; l <= p; ++l) {                          // :112 triu!(Hs[j-1], -1), then treated as triangular
            const psd_mat<double> Hl = psd_gfac(P, n, l);
            for (int r = c + 2; r <= n; ++r) Hl(r, c + 1) = 0.0;
        }
    }
    PSD_ONE {
        psd_gstate st;
        st.n = n; st.p = p; st.wantT = wantT; st.wantZ = wantZ; st.W = st.Wmax = W; st.train_oc = train_oc;
        st.phase = PSD_GPH_CHECK; st.info = 0;
        st.ilast = n; st.ifirst = 1; st.ifirstm = 1; st.ilastm = n;
        st.ziter = (p >= 20) ? -1 : 0;  // :107: p >= log2(floatmin)/log2(eps) = 19.65
        st.jiter = 0; st.maxit = maxitfac * n;
        st.jlo = 1; st.kcur = 0; st.zflag = 0;
        st.nsweeps = st.nzshift = st.nsplit = st.ncase2 = st.ncase3 = st.n2real = st.n2cplx = 0;
        st.nwindows = st.nlog = 0; st.maxlog = maxlog; st.iwarn = 0; st.hj = 0;
        st.c1 = st.c2 = 1.0; st.s1 = st.s2 = 0.0;
        st.train_want = hessmode ? 0 : train_want; st.train_n = 1; st.train_id = 0; st.cursor = 0; st.train_tick0 = 0;
        st.ntrainsweeps = 0;
        for (int q = 0; q < 4; ++q) st.sh[q] = 0.0;
        st.ulp = PSD_DBL_EPS;
        st.smlnum = PSD_DBL_MIN * ((double)n / PSD_DBL_EPS);
        for (int q = 0; q < 6; ++q) st.cyc[q] = 0;
        for (int q = 0; q < 8; ++q) { st.dbg[q] = 0; st.dbgn[q] = 0; }
        st.cstart = 0; st.cfirst = 0;
        if (n == 0) st.phase = PSD_GPH_DONE;
        if (hessmode) {  // stage 2 of _phessenberg!(A, S): columns 1..n-2, positions n-1 down to hj+1
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

// ---- stage 1 helpers of the signed Hessenberg reduction (generalized.jl:1009-1028) -------------------------------------
// RQ of A is taken as the QR of B = J A' J (reflection about the anti-diagonal), so the QR kernels of psd_hess.h serve
// both signs; the neighbours that receive the orthogonal factor are flipped to the same index space and back.
// in place B(r, c) = A(n+1-c, n+1-r).  grid = n (columns)
PSD_KERNEL psd_antitranspose(double* A, int n) {
    const psd_mat<double> M = psd_mat<double>{A, n};
    const int c = PSD_BLOCK_X + 1;
    PSD_PAR_FOR(t, n) {
        const int r = t + 1;
        const int r2 = n + 1 - c, c2 = n + 1 - r;
        if (r + c < n + 1) {  // strictly above the anti-diagonal: swap with the mirror image
            const double x = M(r, c);
            M(r, c) = M(r2, c2);
            M(r2, c2) = x;
        }
    }
}
// reverse the column order (rows == 0) or the row order (rows == 1).  grid = n
PSD_KERNEL psd_flip(double* A, int n, int rows) {
    const psd_mat<double> M = psd_mat<double>{A, n};
    const int k = PSD_BLOCK_X + 1;
    if (rows == 0) {
        if (k > n / 2) return;
        PSD_PAR_FOR(t, n) {
            const double x = M(t + 1, k);
            M(t + 1, k) = M(t + 1, n + 1 - k);
            M(t + 1, n + 1 - k) = x;
        }
    } else {
        PSD_PAR_FOR(t, n / 2) {
            const double x = M(t + 1, k);
            M(t + 1, k) = M(n - t, k);
            M(n - t, k) = x;
        }
    }
}
// zero everything strictly below the diagonal.  grid = n
PSD_KERNEL psd_tril_zero(double* A, int n) {
    const psd_mat<double> M = psd_mat<double>{A, n};
    const int c = PSD_BLOCK_X + 1;
    PSD_PAR_FOR(t, n) {
        if (t + 1 > c) M(t + 1, c) = 0.0;
    }
}
