// Scan chase of the complex single-shift periodic QZ sweep (generalized.jl:808-852, all signatures true): one position
// for ALL factors at once — the ComplexF64 counterpart of psd_chase3.h.
//
// At position j the reference hands ONE rotation from factor to factor: G' from the right on columns (j, j+1) of H_l, a
// new G from (H_l[j,j], H_l[j+1,j]), G from the left on rows (j, j+1) (:826-831) — p dependent (update, rotation) links,
// ~1 200 cycles each in the one-wave chase.  The first column of G' = [c -s; conj(s) c] is (c, conj(s)) = beta (f, g)
// for the pair (f, g) the rotation was made from (zlartg: c = |f|/rho, s = (f/|f|) conj(g)/rho, so beta = conj(f)/(|f| rho)),
// and a rotation does not change when its pair is multiplied by a complex scalar.  So the pairs of all factors are
//     z_p = U_p (c_1, conj(s_1)),   z_{l-1} = U_{l-1} z_l,      U_l = H_l[j:j+1, j:j+1] (upper triangular),
// a chain of 2 x 2 triangular complex matrix-vector products with no rotation and no update on it; then every rotation
// is generated at once (one lane per factor) and all factors are updated side by side by every wavefront of the
// workgroup.  Numerics as in psd_chase3.h: the pair a rotation is made from is U z instead of the updated column read
// back — both are U_l times the same computed column of G'_{l+1}, they agree to rounding in the scale of U_l — and the
// annihilated entry is set to zero as the reference does (:829).
//
// Chain lanes: lane i < 16 does the links 4 i .. 4 i + 3 of the chain p -> p - 1 -> ... -> 2 in registers and hands its
// last pair to lane i + 1 by row_shr:1 (psd_chase3.h); links beyond the chain are identities.
#pragma once

#define PSD_ZC3_TAB 4   // doubles per factor in the rotation table: c, Re s, Im s, -
#define PSD_ZC3_MAXP 64
#define PSD_ZC3_MINP 2
#define PSD_ZC3_WAVES 4
#define PSD_ZC3_FPL 4

struct psd_zc {
    int cmd;  // 0: the workgroup is done (helpers leave), 2: a scan-chase run, 3 / 4: this wavefront's share of a window load / store
    int ld, bsz, bs, be, W;
    int p, n, ifirst, ifirstm, ilast, ilastm, ks, npos;
    int wboff;     // window image: byte offset in dynamic LDS
    psd_ztr* tr;   // this slot's lists
    psd_z* H;      // factors (commands 3 / 4)
    double c0;     // the sweep's start rotation (used when ks == ifirst)
    psd_z s0;
    // factor-sliced runs (psd_zslice3.h; command 5): slices, this workgroup's slice, the tick, the slot's inboxes, the
    // error word of the bounded waits
    int slG, slg, sltick;
    unsigned char* slbox;
    int* slerr;
};

// the pair (x1, x2) under a rotation from the right (columns) or from the left (rows); fix: the second entry becomes 0
PSD_D void psd_zc3_item(psd_z* q, int sd, double c, psd_z s, bool left, bool fix) {
    psd_z a1 = q[0], a2 = q[sd];
    if (left) psd_zrot_left(c, s, a1, a2);
    else psd_zrot_right_adj(c, s, a1, a2);
    if (fix) a2 = zmk(0.0, 0.0);
    q[0] = a1;
    q[sd] = a2;
}

// The apply phase of one position: thread (f, q) is the q-th of the tpf threads of factor f + 1.  sub = 0: right updates
// of H_2..H_p (by the rotation of the factor behind them in the chain: owner l + 1, or H_1's for factor p), left update
// of H_1; sub = 1: left updates of H_2..H_p with column j becoming (r, 0) (generalized.jl:828-829), right update of H_1.
PSD_D void psd_zc3_apply(psd_z* wb, const double* tab, int sub, int f, int q, int tpf, int p, int ld, int bsz, int bs,
                         int be, int j, int r0, int c1max, int h1r0, int h1r1, int h1c1) {
    if (f >= p) return;
    const int l = f + 1;
    psd_z* const blk = wb + f * bsz;
    const bool right = (l >= 2) == (sub == 0);
    if (right) {
        const int lo = (l == 1) ? 2 : ((l == p) ? 1 : (l + 1));
        const double* t = tab + (lo - 1) * PSD_ZC3_TAB;
        const double c = t[0];
        const psd_z s = zmk(t[1], t[2]);
        const int ra = (l == 1) ? h1r0 : r0, rb = (l == 1) ? h1r1 : (j + 1);
        psd_z* const col = blk + (j - bs) * ld + (ra - bs);
        for (int r = q; r <= rb - ra; r += tpf) psd_zc3_item(col + r, ld, c, s, false, false);
    } else {
        const double* t = tab + f * PSD_ZC3_TAB;
        const double c = t[0];
        const psd_z s = zmk(t[1], t[2]);
        const int cb = (l == 1) ? h1c1 : c1max;  // columns j .. cb
        psd_z* const row = blk + (j - bs) * ld + (j - bs);
        for (int cc = q; cc <= cb - j; cc += tpf) psd_zc3_item(row + cc * ld, 1, c, s, true, l >= 2 && cc == 0);
    }
    (void)be;
}

// 2 x 2 upper triangular complex block times pair
PSD_D void psd_zc3_link(const psd_z u00, const psd_z u01, const psd_z u11, psd_z& z0, psd_z& z1) {
    const psd_z n0 = zadd(zmul(u00, z0), zmul(u01, z1));
    const psd_z n1 = zmul(u11, z1);
    z0 = n0;
    z1 = n1;
}
PSD_D void psd_zc3_loadu(const psd_z* q, int ld, psd_z& u00, psd_z& u01, psd_z& u11) {
    u00 = q[0];
    u01 = q[ld];
    u11 = q[ld + 1];
    const double um = fmax(fmax(zabs1(u00), zabs1(u01)), zabs1(u11));
    if (!(um > 1e-18 && um < 1e18)) {  // (far from unit scale: the pair is rescaled after every lane's four links anyway)
        const int eu = psd_c3_expo(um);
        u00 = zmk(psd_c3_ldexp(u00.re, -eu), psd_c3_ldexp(u00.im, -eu));
        u01 = zmk(psd_c3_ldexp(u01.re, -eu), psd_c3_ldexp(u01.im, -eu));
        u11 = zmk(psd_c3_ldexp(u11.re, -eu), psd_c3_ldexp(u11.im, -eu));
    }
}

// One run = positions ks .. ks + npos - 1 of a window.  Called by every wavefront of the workgroup (wv = its index, nw
// their number; the simulated tier: one call, wv = 0, nw = 1) with the same command block.
PSD_D void psd_zc3_run(const psd_zc& Cin, int wv_, int nw_, int taboff_) {
    PSD_LDS_DECL;
#ifndef PSD_HOSTSIM
    const int wv = PSD_C2_UNI(wv_), nw = PSD_C2_UNI(nw_), taboff = PSD_C2_UNI(taboff_);
    const int p = PSD_C2_UNI(Cin.p), ld = PSD_C2_UNI(Cin.ld), bsz = PSD_C2_UNI(Cin.bsz), bs = PSD_C2_UNI(Cin.bs), be = PSD_C2_UNI(Cin.be);
    const int ifirst = PSD_C2_UNI(Cin.ifirst), ifirstm = PSD_C2_UNI(Cin.ifirstm), ilastm = PSD_C2_UNI(Cin.ilastm);
    const int ks = PSD_C2_UNI(Cin.ks), npos = PSD_C2_UNI(Cin.npos);
    psd_z* const wb = (psd_z*)(psd_lds + PSD_C2_UNI(Cin.wboff));
#else
    const int wv = wv_, nw = nw_, taboff = taboff_;
    const int p = Cin.p, ld = Cin.ld, bsz = Cin.bsz, bs = Cin.bs, be = Cin.be;
    const int ifirst = Cin.ifirst, ifirstm = Cin.ifirstm, ilastm = Cin.ilastm, ks = Cin.ks, npos = Cin.npos;
    psd_z* const wb = (psd_z*)(psd_lds + Cin.wboff);
#endif
    double* const tab = (double*)(psd_lds + taboff);
    psd_ztr* const trb = Cin.tr;
    const int r0 = (bs > ifirstm) ? bs : ifirstm;
    const int c1max = (be < ilastm) ? be : ilastm;
#ifndef PSD_HOSTSIM
    const int lane = (int)threadIdx.x;
    const int tid = wv * 64 + lane, NT = nw * 64;
    const int tpf = (NT / p > 0) ? (NT / p) : 1;
    const int af = tid / tpf, aq = tid - af * tpf;
#endif
    for (int kk = 0; kk < npos; ++kk) {
        const int j = ks + kk;
        const int itmp = (j + 2 < ilastm) ? (j + 2) : ilastm;
        const int h1r1 = (itmp < be) ? itmp : be;  // rows of H_1's right update: r0 .. h1r1
        if (wv == 0) {
            // ---- H_1's rotation (generalized.jl:811-816)
            double c1;
            psd_z s1;
            if (j > ifirst) {
                psd_z* q = wb + (j - 1 - bs) * ld + (j - bs);
                psd_z r;
                psd_zgivens(q[0], q[1], c1, s1, r);
                PSD_WAVE_SYNC();
                PSD_ONE {
                    q[0] = r;
                    q[1] = zmk(0.0, 0.0);
                }
            } else {
                c1 = Cin.c0;
                s1 = Cin.s0;
            }
            const psd_z w0s = zmk(c1, 0.0), w1s = zconj(s1);  // first column of G_1'
#ifndef PSD_HOSTSIM
            // ---- scan on the chain lanes (psd_chase3.h): lane i < 16 does links 4 i .. 4 i + 3 (factors p - 4 i, ...)
            psd_z U0[PSD_ZC3_FPL], U1[PSD_ZC3_FPL], U2[PSD_ZC3_FPL], zq0[PSD_ZC3_FPL], zq1[PSD_ZC3_FPL];
            const bool chl = lane < 16;
#pragma unroll
            for (int q4 = 0; q4 < PSD_ZC3_FPL; ++q4) {
                const int c = PSD_ZC3_FPL * lane + q4, lf = p - c;
                U0[q4] = zmk(1.0, 0.0);
                U1[q4] = zmk(0.0, 0.0);
                U2[q4] = zmk(1.0, 0.0);
                if (chl && c < p - 1) psd_zc3_loadu(wb + (lf - 1) * bsz + (j - bs) * ld + (j - bs), ld, U0[q4], U1[q4], U2[q4]);
                zq0[q4] = zq1[q4] = zmk(0.0, 0.0);
            }
            const int nsteps = (p - 1 + PSD_ZC3_FPL - 1) / PSD_ZC3_FPL;
            psd_z z0 = zmk(0.0, 0.0), z1 = zmk(0.0, 0.0);
            for (int s = 0; s < nsteps; ++s) {
                psd_z w0 = zmk(psd_c3_shr(z0.re, w0s.re), psd_c3_shr(z0.im, w0s.im));
                psd_z w1 = zmk(psd_c3_shr(z1.re, w1s.re), psd_c3_shr(z1.im, w1s.im));
                if (s <= lane) {
#pragma unroll
                    for (int q4 = 0; q4 < PSD_ZC3_FPL; ++q4) {
                        psd_zc3_link(U0[q4], U1[q4], U2[q4], w0, w1);
                        zq0[q4] = w0;
                        zq1[q4] = w1;
                    }
                    const int e = psd_c3_expo(fmax(zabs1(w0), zabs1(w1)));
                    z0 = zmk(psd_c3_ldexp(w0.re, -e), psd_c3_ldexp(w0.im, -e));
                    z1 = zmk(psd_c3_ldexp(w1.re, -e), psd_c3_ldexp(w1.im, -e));
                }
            }
#pragma unroll
            for (int q4 = 0; q4 < PSD_ZC3_FPL; ++q4) {
                const int c = PSD_ZC3_FPL * lane + q4, lf = p - c;
                if (chl && c < p - 1) {
                    double* t = tab + (lf - 1) * PSD_ZC3_TAB;
                    t[0] = zq0[q4].re;
                    t[1] = zq0[q4].im;
                    t[2] = zq1[q4].re;
                    t[3] = zq1[q4].im;
                }
            }
            // ---- every rotation at once: lane l - 1 = factor l
            if (lane < p) {
                double c = c1;
                psd_z s = s1;
                double* t = tab + lane * PSD_ZC3_TAB;
                if (lane >= 1) {
                    const psd_z f = zmk(t[0], t[1]), g = zmk(t[2], t[3]);
                    psd_z r;
                    psd_zgivens(f, g, c, s, r);
                }
                t[0] = c;
                t[1] = s.re;
                t[2] = s.im;
                psd_ztr tr;
                tr.pos = j;
                tr.pad = 0;
                tr.c = c;
                tr.s = s;
                if (kk < PSD_ZTR_CAP) trb[(size_t)lane * PSD_ZTR_CAP + kk] = tr;
            }
#else
            {
                psd_z z0 = w0s, z1 = w1s;
                int since = 0;
                for (int lf = p; lf >= 2; --lf) {
                    psd_z u00, u01, u11;
                    psd_zc3_loadu(wb + (lf - 1) * bsz + (j - bs) * ld + (j - bs), ld, u00, u01, u11);
                    psd_zc3_link(u00, u01, u11, z0, z1);
                    double* t = tab + (lf - 1) * PSD_ZC3_TAB;
                    t[0] = z0.re; t[1] = z0.im; t[2] = z1.re; t[3] = z1.im;
                    if (++since == PSD_ZC3_FPL) {
                        since = 0;
                        const int e = psd_c3_expo(fmax(zabs1(z0), zabs1(z1)));
                        z0 = zmk(psd_c3_ldexp(z0.re, -e), psd_c3_ldexp(z0.im, -e));
                        z1 = zmk(psd_c3_ldexp(z1.re, -e), psd_c3_ldexp(z1.im, -e));
                    }
                }
                for (int lane = 0; lane < p; ++lane) {
                    double c = c1;
                    psd_z s = s1;
                    double* t = tab + lane * PSD_ZC3_TAB;
                    if (lane >= 1) {
                        psd_z r;
                        psd_zgivens(zmk(t[0], t[1]), zmk(t[2], t[3]), c, s, r);
                    }
                    t[0] = c; t[1] = s.re; t[2] = s.im;
                    psd_ztr tr;
                    tr.pos = j; tr.pad = 0; tr.c = c; tr.s = s;
                    if (kk < PSD_ZTR_CAP) trb[(size_t)lane * PSD_ZTR_CAP + kk] = tr;
                }
            }
#endif
        }
#ifndef PSD_HOSTSIM
        PSD_PAIR_BARRIER();
        psd_zc3_apply(wb, tab, 0, af, aq, tpf, p, ld, bsz, bs, be, j, r0, c1max, r0, h1r1, c1max);
        PSD_PAIR_BARRIER();
        psd_zc3_apply(wb, tab, 1, af, aq, tpf, p, ld, bsz, bs, be, j, r0, c1max, r0, h1r1, c1max);
        PSD_PAIR_BARRIER();
#else
        {
            const int NT = 64 * PSD_ZC3_WAVES, tpf = (NT / p > 0) ? (NT / p) : 1;
            for (int sub = 0; sub < 2; ++sub)
                for (int tid = 0; tid < NT; ++tid)
                    psd_zc3_apply(wb, tab, sub, tid / tpf, tid % tpf, tpf, p, ld, bsz, bs, be, j, r0, c1max, r0, h1r1, c1max);
        }
#endif
    }
    (void)nw;
}
