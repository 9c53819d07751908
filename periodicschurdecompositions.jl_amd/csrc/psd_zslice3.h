// Factor-sliced scan chase of the ComplexF64 engine: psd_slice3.h for the single-shift sweep of psd_zchase3.h.
// G workgroups chase one window, workgroup g with the diagonal window blocks of the factors of its contiguous slice of the
// period only.  One chain crosses the slices per position: slice 0 (the owner of H_1) makes H_1's rotation and hands the
// first column of its adjoint, (c_1, conj s_1), to slice G - 1 (the owner of H_p); every slice takes the pair that comes in
// through its factors (z <- U_l z) and hands the last one on to the slice below as a tagged record of four doubles; the
// rotation of a neighbour's factor is a function of the pair that was handed over (for H_1: the pair itself).  Everything
// else — window load and store, rotations, updates, the owners' lists — is local to a slice.  Inboxes, records, bounded
// waits and the command block of a slot: psd_slice3.h.  HIP only.
#pragma once
#ifndef PSD_HOSTSIM

PSD_D unsigned char* psd_zsl_cmd(unsigned char* slmem, int slot) {
    return slmem + (size_t)slot * (PSD_SL_CMD_BYTES + PSD_SL_MAXG * PSD_SL_BOX_BYTES);
}

// local apply phase (psd_zc3_apply with local factor indices: the right update of factor f takes table entry f + 1)
PSD_D void psd_zc3s_apply(psd_z* wb, const double* tab, int sub, int f, int q, int tpf, int nloc, bool lead, int ld, int bsz, int bs,
                          int j, int r0, int c1max, int h1r1) {
    if (f >= nloc) return;
    const bool h1 = lead && f == 0;
    psd_z* const blk = wb + f * bsz;
    const bool right = (!h1) == (sub == 0);
    if (right) {
        const double* t = tab + (f + 1) * PSD_ZC3_TAB;
        const double c = t[0];
        const psd_z s = zmk(t[1], t[2]);
        const int rb = h1 ? h1r1 : (j + 1);
        psd_z* const col = blk + (j - bs) * ld + (r0 - bs);
        for (int r = q; r <= rb - r0; r += tpf) psd_zc3_item(col + r, ld, c, s, false, false);
    } else {
        const double* t = tab + f * PSD_ZC3_TAB;
        const double c = t[0];
        const psd_z s = zmk(t[1], t[2]);
        psd_z* const row = blk + (j - bs) * ld + (j - bs);
        for (int cc = q; cc <= c1max - j; cc += tpf) psd_zc3_item(row + cc * ld, 1, c, s, true, !h1 && cc == 0);
    }
}

// One run of slice C.slg of C.slG (positions ks .. ks + npos - 1); every wavefront of the slice's workgroup calls it.
// Window image: the blocks of the slice's factors only, block f = factor jlo + f.
PSD_D void psd_zc3s_run(const psd_zc& Cin, int wv_, int nw_, int taboff_) {
    PSD_LDS_DECL;
    const int wv = PSD_C2_UNI(wv_), nw = PSD_C2_UNI(nw_), taboff = PSD_C2_UNI(taboff_);
    const int p = PSD_C2_UNI(Cin.p), ld = PSD_C2_UNI(Cin.ld), bsz = PSD_C2_UNI(Cin.bsz), bs = PSD_C2_UNI(Cin.bs), be = PSD_C2_UNI(Cin.be);
    const int ifirst = PSD_C2_UNI(Cin.ifirst), ifirstm = PSD_C2_UNI(Cin.ifirstm), ilastm = PSD_C2_UNI(Cin.ilastm);
    const int ks = PSD_C2_UNI(Cin.ks), npos = PSD_C2_UNI(Cin.npos);
    const int G = PSD_C2_UNI(Cin.slG), g = PSD_C2_UNI(Cin.slg), tick = PSD_C2_UNI(Cin.sltick);
    psd_z* const wb = (psd_z*)(psd_lds + PSD_C2_UNI(Cin.wboff));
    double* const tab = (double*)(psd_lds + taboff);
    psd_ztr* const trb = Cin.tr;
    unsigned char* const box = Cin.slbox;
    int* const err = Cin.slerr;
    int jlo, jhi;
    psd_sl_range(p, G, g, jlo, jhi);
    const bool lead = g == 0, top = jhi == p;
    const int nloc = jhi - jlo + 1;
    const int jmin = lead ? 2 : jlo;
    const int nlinks = jhi - jmin + 1;
    psd_vrec* const inz = (psd_vrec*)(box + (size_t)g * PSD_SL_BOX_BYTES);
    psd_vrec* const outz = (psd_vrec*)(box + (size_t)((g == 0) ? (G - 1) : (g - 1)) * PSD_SL_BOX_BYTES);
    const int r0 = (bs > ifirstm) ? bs : ifirstm;
    const int c1max = (be < ilastm) ? be : ilastm;
    const int lane = (int)threadIdx.x;
    const int tid = wv * 64 + lane, NT = nw * 64;
    const int tpf = (NT / nloc > 0) ? (NT / nloc) : 1;
    const int af = tid / tpf, aq = tid - af * tpf;
    for (int kk = 0; kk < npos; ++kk) {
        const int j = ks + kk;
        const int itmp = (j + 2 < ilastm) ? (j + 2) : ilastm;
        const int h1r1 = (itmp < be) ? itmp : be;
        if (wv == 0) {
            const unsigned long long tagz = psd_sl_tag(tick, kk, 2);
            // ---- slice 0: H_1's rotation (generalized.jl:811-816); the first column of its adjoint starts the lap
            double c1 = 1.0;
            psd_z s1 = zmk(0.0, 0.0);
            if (lead) {
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
                if (lane == 0) {
                    psd_sl_put(outz + 0, c1, tagz);
                    psd_sl_put(outz + 1, 0.0, tagz);
                    psd_sl_put(outz + 2, s1.re, tagz);
                    psd_sl_put(outz + 3, -s1.im, tagz);
                }
            }
            // ---- the chain lanes' blocks: lane i < 16 does links 4 i .. 4 i + 3 = factors jhi - 4 i, ...
            psd_z U0[PSD_ZC3_FPL], U1[PSD_ZC3_FPL], U2[PSD_ZC3_FPL], zq0[PSD_ZC3_FPL], zq1[PSD_ZC3_FPL];
            const bool chl = lane < 16;
#pragma unroll
            for (int q4 = 0; q4 < PSD_ZC3_FPL; ++q4) {
                const int c = PSD_ZC3_FPL * lane + q4, lf = jhi - c;
                U0[q4] = zmk(1.0, 0.0);
                U1[q4] = zmk(0.0, 0.0);
                U2[q4] = zmk(1.0, 0.0);
                if (chl && c < nlinks) psd_zc3_loadu(wb + (lf - jlo) * bsz + (j - bs) * ld + (j - bs), ld, U0[q4], U1[q4], U2[q4]);
                zq0[q4] = zq1[q4] = zmk(0.0, 0.0);
            }
            // ---- the pair that enters this slice
            double zin[4];
            (void)psd_sl_get(inz, 4, tagz, zin, err);
            const int ez = psd_c3_expo(fmax(fmax(fabs(zin[0]), fabs(zin[1])), fmax(fabs(zin[2]), fabs(zin[3]))));
            const psd_z w0s = zmk(psd_c3_ldexp(zin[0], -ez), psd_c3_ldexp(zin[1], -ez)), w1s = zmk(psd_c3_ldexp(zin[2], -ez), psd_c3_ldexp(zin[3], -ez));
            const int nsteps = (nlinks + PSD_ZC3_FPL - 1) / PSD_ZC3_FPL;
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
                const int c = PSD_ZC3_FPL * lane + q4, lf = jhi - c;
                if (chl && c < nlinks) {
                    double* t = tab + (lf - jlo) * PSD_ZC3_TAB;
                    t[0] = zq0[q4].re;
                    t[1] = zq0[q4].im;
                    t[2] = zq1[q4].re;
                    t[3] = zq1[q4].im;
                }
            }
            // ---- every rotation at once: lane f = local factor f; lane nloc the neighbour's, from the pair that came in
            const bool h1 = lead && lane == 0;
            if (lane <= nloc) {
                double c = c1;
                psd_z s = s1;
                double* t = tab + lane * PSD_ZC3_TAB;
                if (lane == nloc) {
                    if (top) {  // (H_1's rotation: the pair IS the first column of its adjoint)
                        c = zin[0];
                        s = zmk(zin[2], -zin[3]);
                    } else {
                        psd_z r;
                        psd_zgivens(zmk(zin[0], zin[1]), zmk(zin[2], zin[3]), c, s, r);
                    }
                } else if (!h1) {
                    const psd_z f = zmk(t[0], t[1]), gg = zmk(t[2], t[3]);
                    if (!lead && lane == 0) {  // the slice's last pair goes on to the slice below
                        psd_sl_put(outz + 0, f.re, tagz);
                        psd_sl_put(outz + 1, f.im, tagz);
                        psd_sl_put(outz + 2, gg.re, tagz);
                        psd_sl_put(outz + 3, gg.im, tagz);
                    }
                    psd_z r;
                    psd_zgivens(f, gg, c, s, r);
                }
                t[0] = c;
                t[1] = s.re;
                t[2] = s.im;
                if (lane < nloc) {  // the owners of this slice write their own lists
                    psd_ztr tr;
                    tr.pos = j;
                    tr.pad = 0;
                    tr.c = c;
                    tr.s = s;
                    if (kk < PSD_ZTR_CAP) trb[(size_t)(jlo + lane - 1) * PSD_ZTR_CAP + kk] = tr;
                }
            }
        }
        PSD_PAIR_BARRIER();
        psd_zc3s_apply(wb, tab, 0, af, aq, tpf, nloc, lead, ld, bsz, bs, j, r0, c1max, h1r1);
        PSD_PAIR_BARRIER();
        psd_zc3s_apply(wb, tab, 1, af, aq, tpf, nloc, lead, ld, bsz, bs, j, r0, c1max, h1r1);
        PSD_PAIR_BARRIER();
    }
}
#endif
