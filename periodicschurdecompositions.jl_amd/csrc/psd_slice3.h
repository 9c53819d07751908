// Factor-sliced scan chase: the north_star partition as device code.
//
// G workgroups chase ONE window: workgroup g of a slot holds the diagonal window blocks of the factors of ITS contiguous
// slice of the period, (g p/G, (g + 1) p/G], and nothing of the others'.  A transformation generated at factor j touches
// H_j (rows), H_{j-1} (columns) and Z_j only (PSD.jl:855-864), so everything a slice does is local to it — its window
// load and store, its reflectors, its updates, its owners' transformation lists — except the two chain vectors of the
// scan chase (psd_chase3.h) that cross a slice boundary once per position: the 3-vector of scan 1 and the 2-vector of
// scan 2, handed from slice g + 1 to slice g as tagged records in the receiver's inbox, and x(1), which slice 0 (the
// owner of H_1) hands to slice G - 1 (the owner of H_p: the cyclic wrap, PSD.jl:837,858) to start the lap.  The
// reflector of a neighbour's factor — slice g needs the one of factor j_hi + 1 for the columns of its own top factor —
// is not sent: it is a function of the vector that was (same code, same bits).
//
//   lap of one position:  slice 0: x(1) -> inbox of slice G-1;  slice G-1: scan 1 over its factors -> z -> slice G-2 ...
//                         -> slice 0: scan 1 over factors p/G .. 2.  Scan 2 the same way behind it, started by slice G-1.
//                         Then every slice updates its own blocks; slice 0 closes the lap on H_1's columns and starts the
//                         next position.
//
// An inbox is a plain device pointer: on one GPU the G workgroups of a slot are G compute units and the inboxes lie in
// that GPU's memory; across GPUs the same pointer would be a peer mapping of the downstream GPU's inbox (xGMI), nothing
// else changes.  A record is self-validating, the protocol of the Hessenberg pipe form (psd_hess2.h): every double x
// travels as the pair (bits(x), bits(x) XOR tag) in two agent-scope stores, the tag unique per (tick, position, kind);
// whatever mixture of old and new halves a reader sees fails the check unless the value is the one the producer wrote.
// No flag, no fence, no read-modify-write.  Every wait is bounded in time (about two seconds), after which an error
// word is set, every later wait ends at once and the host returns PSD_INFO_RUNTIME.
//
// Slice 0 is the ordinary chase workgroup of the slot: it runs the state machine, and between sweeps windows it works on
// all p factors as before (decisions, RQ clean-up and deflation windows are not sliced).  Slices g >= 1 are workers: they
// wait for the slot's command of the tick (the window geometry; published behind a release fence, picked up behind an
// acquire fence), run the window, store their blocks and leave.  HIP only: the serial simulation cannot run workgroups
// that wait for each other, there the engine stays unsliced.
#pragma once
#ifndef PSD_HOSTSIM

#define PSD_SL_MAXG 8
#define PSD_SL_WAIT_TICKS 200000000LL  // bound of every wait, in s_memrealtime ticks (100 MHz)
#define PSD_SL_BOX_BYTES 256           // inbox of one slice: z (3 records), t (2 records)
#define PSD_SL_CMD_BYTES 512           // command block of one slot in global memory: flag word + psd_c2

struct psd_vrec {
    unsigned long long a, b;  // bits(x), bits(x) ^ tag
};

PSD_D void psd_sl_put(psd_vrec* q, double x, unsigned long long tag) {
    const unsigned long long bits = (unsigned long long)__double_as_longlong(x);
    __hip_atomic_store(&q->a, bits, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __hip_atomic_store(&q->b, bits ^ tag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
// nval records from `q` (every lane of the wavefront polls the same words); false: gave up
PSD_D bool psd_sl_get(const psd_vrec* q, int nval, unsigned long long tag, double* out, int* err) {
    int spins = 0;
    long long t0 = 0;
    for (;;) {
        bool ok = true;
        for (int v = 0; v < nval; ++v) {
            const unsigned long long a = __hip_atomic_load(&q[v].a, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            const unsigned long long b = __hip_atomic_load(&q[v].b, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            ok = ok && ((a ^ b) == tag);
            out[v] = __longlong_as_double((long long)a);
        }
        if (ok) return true;
        if ((++spins & 255) == 0) {
            if (__hip_atomic_load(err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0) break;
            const long long now = (long long)__builtin_amdgcn_s_memrealtime();
            if (t0 == 0) t0 = now;
            else if (now - t0 > PSD_SL_WAIT_TICKS) {
                __hip_atomic_store(err, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                break;
            }
        }
        __builtin_amdgcn_s_sleep(1);
    }
    for (int v = 0; v < nval; ++v) out[v] = 0.0;
    return false;
}
PSD_D unsigned long long psd_sl_tag(int tick, int kk, int kind) {
    return ((unsigned long long)(unsigned)(tick + 1) << 24) | ((unsigned long long)(unsigned)kk << 4) | (unsigned long long)kind;
}
// first and last factor (1-based, inclusive) of slice g of G (the rule of psd_ctx::slice and sharded.period_slice)
PSD_HD void psd_sl_range(int p, int G, int g, int& jlo, int& jhi) {
    const int base = p / G, rem = p % G;
    const int lo = g * base + (g < rem ? g : rem);
    jlo = lo + 1;
    jhi = lo + base + (g < rem ? 1 : 0);
}

// Local apply phase: thread (f, q) is the q-th of the tpf threads of local factor f (global factor jbase + f).  The right
// update of a factor takes the reflectors of table entry f + 1 — the next factor of the slice, or the neighbour slice's
// bottom factor (entry nloc; H_1's for the top slice, which has no 2-reflector) —, the left update entry f.
PSD_D void psd_c3s_apply(double* wb, const double* tab, int sub, int f, int q, int tpf, int nloc, bool lead, bool top, int ld,
                         int bsz, int bs, int k, int l, int r0, int nrw, int ncl) {
    if (f >= nloc) return;
    const bool h1 = lead && f == 0;
    double* const blk = wb + f * bsz;
    const bool right = (!h1) == (sub == 0);
    if (right) {
        const double* t = tab + (f + 1) * PSD_C3_TAB;
        const double v1 = t[0], v2 = t[1], tau = t[2], w2 = t[3], tau2 = t[4];
        const bool with2 = !(top && f == nloc - 1);
        double* const col = blk + (k - bs) * ld + (r0 - bs);
        for (int r = q; r < nrw; r += tpf) psd_c3_item(col + r, ld, v1, v2, tau, with2, w2, tau2, 0);
    } else {
        const double* t = tab + f * PSD_C3_TAB;
        const double v1 = t[0], v2 = t[1], tau = t[2], w2 = t[3], tau2 = t[4];
        double* const row = blk + (k - bs) * ld + (k - bs);
        if (h1) {
            for (int cc = q; cc < ncl; cc += tpf) psd_c3_item(row + cc * ld, 1, v1, v2, tau, false, 0.0, 0.0, 0);
            if (k > l && q == tpf - 1) {  // PSD.jl:822-827
                double* c = blk + (k - 1 - bs) * ld + (k - bs);
                c[0] = t[5];
                c[1] = 0.0;
                c[2] = 0.0;
            }
        } else {
            const double b3 = t[5], b2 = t[6];
            for (int cc = q; cc < ncl; cc += tpf)
                psd_c3_item(row + cc * ld, 1, v1, v2, tau, cc >= 1, w2, tau2, (cc == 0) ? 1 : ((cc == 1) ? 2 : 0), (cc == 0) ? b3 : b2);
        }
    }
}

// One run of slice C.slg of C.slG (positions ks .. ks + npos - 1), called by every wavefront of the slice's workgroup.
// The window image holds the blocks of the slice's factors only, block f = factor jlo + f.
PSD_D void psd_c3s_run(const psd_c2& Cin, int wv_, int nw_, int taboff_) {
    PSD_LDS_DECL;
    const int wv = PSD_C2_UNI(wv_), nw = PSD_C2_UNI(nw_), taboff = PSD_C2_UNI(taboff_);
    const int p = PSD_C2_UNI(Cin.p), ld = PSD_C2_UNI(Cin.ld), bsz = PSD_C2_UNI(Cin.bsz), bs = PSD_C2_UNI(Cin.bs);
    const int l = PSD_C2_UNI(Cin.l), ie = PSD_C2_UNI(Cin.i), ks = PSD_C2_UNI(Cin.ks), npos = PSD_C2_UNI(Cin.npos);
    const int c1max = PSD_C2_UNI(Cin.c1max), r0 = PSD_C2_UNI(Cin.r0), n1 = PSD_C2_UNI(Cin.n1), nj = PSD_C2_UNI(Cin.nj);
    const int G = PSD_C2_UNI(Cin.slG), g = PSD_C2_UNI(Cin.slg), tick = PSD_C2_UNI(Cin.sltick);
    double* const wb = (double*)(psd_lds + PSD_C2_UNI(Cin.wboff));
    double* const tab = (double*)(psd_lds + taboff);
    psd_tr* const trb = Cin.tr;
    unsigned char* const box = Cin.slbox;  // this slot's inboxes: slice g's at box + g * PSD_SL_BOX_BYTES
    int* const err = Cin.slerr;
    int jlo, jhi;
    psd_sl_range(p, G, g, jlo, jhi);
    const bool lead = g == 0, top = jhi == p;
    const int nloc = jhi - jlo + 1;              // factors of this slice (block f = factor jlo + f)
    const int jmin = lead ? 2 : jlo;             // lowest factor on the chain
    const int nlinks = jhi - jmin + 1;           // chain links of this slice: factors jhi, jhi - 1, ..., jmin
    psd_vrec* const inz = (psd_vrec*)(box + (size_t)g * PSD_SL_BOX_BYTES);
    psd_vrec* const intv = inz + 4;
    psd_vrec* const outz = (psd_vrec*)(box + (size_t)((g == 0) ? (G - 1) : (g - 1)) * PSD_SL_BOX_BYTES);  // (slice 0 sends x(1) to slice G - 1)
    psd_vrec* const outt = outz + 4;
    const int lane = (int)threadIdx.x;
    const int tid = wv * 64 + lane, NT = nw * 64;
    const int tpf = (NT / nloc > 0) ? (NT / nloc) : 1;
    const int af = tid / tpf, aq = tid - af * tpf;
    for (int kk = 0; kk < npos; ++kk) {
        const int k = ks + kk;
        const int rlim = (k + 3 < ie) ? (k + 3) : ie;
        const int nrw = rlim - r0 + 1;
        int ncl = c1max - k + 1;
        if (ncl < 0) ncl = 0;
        if (wv == 0) {
            const unsigned long long tagz = psd_sl_tag(tick, kk, 2), tagt = psd_sl_tag(tick, kk, 3);
            const bool h1 = lead && lane == 0;
            const bool fac = lane < nloc && !h1;  // a factor of the chain
            // ---- slice 0: x(1) (PSD.jl:813-816) starts the lap at slice G - 1
            double x0 = 0.0, x1 = 0.0, x2 = 0.0;
            if (lead) {
                if (k > l) {
                    const double* q = wb + (k - 1 - bs) * ld + (k - bs);
                    x0 = q[0];
                    x1 = q[1];
                    x2 = q[2];
                } else {
                    x0 = Cin.v0;
                    x1 = Cin.v1;
                    x2 = Cin.v2;
                }
                if (lane == 0) {
                    psd_sl_put(outz + 0, x0, tagz);
                    psd_sl_put(outz + 1, x1, tagz);
                    psd_sl_put(outz + 2, x2, tagz);
                }
            }
            // this lane's own factor (for its B block)
            double u00 = 0.0, u01 = 0.0, u02 = 0.0, u11 = 0.0, u12 = 0.0, u22 = 0.0;
            int euo = 0;
            if (fac) {
                const double* q = wb + lane * bsz + (k - bs) * ld + (k - bs);
                u00 = q[0];
                u01 = q[ld];
                u11 = q[ld + 1];
                u02 = q[2 * ld];
                u12 = q[2 * ld + 1];
                u22 = q[2 * ld + 2];
                const double um = fmax(psd_c3_max3(u00, u01, u02), psd_c3_max3(u11, u12, u22));
                if (!(um > 1e-18 && um < 1e18)) {
                    const int eu = psd_c3_expo(um);
                    euo = eu;
                    u00 = psd_c3_ldexp(u00, -eu);
                    u01 = psd_c3_ldexp(u01, -eu);
                    u02 = psd_c3_ldexp(u02, -eu);
                    u11 = psd_c3_ldexp(u11, -eu);
                    u12 = psd_c3_ldexp(u12, -eu);
                    u22 = psd_c3_ldexp(u22, -eu);
                }
            }
            // the chain lanes' blocks: lane i < 16 does links 4 i .. 4 i + 3 = factors jhi - 4 i, ...
            double U[PSD_C3_FPL][6], zq[PSD_C3_FPL][3];
            int Ue[PSD_C3_FPL];
            const bool chl = lane < 16;
#pragma unroll
            for (int q4 = 0; q4 < PSD_C3_FPL; ++q4) {
                const int c = PSD_C3_FPL * lane + q4, jf = jhi - c;
                double a00 = 1.0, a01 = 0.0, a02 = 0.0, a11 = 1.0, a12 = 0.0, a22 = 1.0;
                Ue[q4] = 0;
                if (chl && c < nlinks) {
                    const double* q = wb + (jf - jlo) * bsz + (k - bs) * ld + (k - bs);
                    a00 = q[0];
                    a01 = q[ld];
                    a11 = q[ld + 1];
                    a02 = q[2 * ld];
                    a12 = q[2 * ld + 1];
                    a22 = q[2 * ld + 2];
                    const double um = fmax(psd_c3_max3(a00, a01, a02), psd_c3_max3(a11, a12, a22));
                    if (!(um > 1e-18 && um < 1e18)) {
                        const int eu = psd_c3_expo(um);
                        Ue[q4] = eu;
                        a00 = psd_c3_ldexp(a00, -eu);
                        a01 = psd_c3_ldexp(a01, -eu);
                        a02 = psd_c3_ldexp(a02, -eu);
                        a11 = psd_c3_ldexp(a11, -eu);
                        a12 = psd_c3_ldexp(a12, -eu);
                        a22 = psd_c3_ldexp(a22, -eu);
                    }
                }
                U[q4][0] = a00; U[q4][1] = a01; U[q4][2] = a02; U[q4][3] = a11; U[q4][4] = a12; U[q4][5] = a22;
                zq[q4][0] = zq[q4][1] = zq[q4][2] = 0.0;
            }
            // ---- the chain vector that enters this slice: from the slice above (x(1) for the top slice)
            double zin[3];
            (void)psd_sl_get(inz, 3, tagz, zin, err);
            const int ez = psd_c3_expo(psd_c3_max3(zin[0], zin[1], zin[2]));
            const double xs0 = psd_c3_ldexp(zin[0], -ez), xs1 = psd_c3_ldexp(zin[1], -ez), xs2 = psd_c3_ldexp(zin[2], -ez);
            // ---- scan 1 over this slice's links (psd_chase3.h)
            const int nsteps = (nlinks + PSD_C3_FPL - 1) / PSD_C3_FPL;
            double z0 = 0.0, z1 = 0.0, z2 = 0.0;
            int eout = 0;
            for (int s = 0; s < nsteps; ++s) {
                double w0 = psd_c3_shr(z0, xs0), w1 = psd_c3_shr(z1, xs1), w2 = psd_c3_shr(z2, xs2);
                if (s <= lane) {
#pragma unroll
                    for (int q4 = 0; q4 < PSD_C3_FPL; ++q4) {
                        const double n0 = __builtin_fma(U[q4][0], w0, __builtin_fma(U[q4][1], w1, U[q4][2] * w2));
                        const double n1_ = __builtin_fma(U[q4][3], w1, U[q4][4] * w2);
                        const double n2 = U[q4][5] * w2;
                        w0 = n0;
                        w1 = n1_;
                        w2 = n2;
                        zq[q4][0] = n0;
                        zq[q4][1] = n1_;
                        zq[q4][2] = n2;
                    }
                    eout = psd_c3_expo(psd_c3_max3(w0, w1, w2));
                    z0 = psd_c3_ldexp(w0, -eout);
                    z1 = psd_c3_ldexp(w1, -eout);
                    z2 = psd_c3_ldexp(w2, -eout);
                }
            }
            // (slots 3 / 4: the powers of two between a factor's vector and the product it stands for, as in psd_c3_run)
#pragma unroll
            for (int q4 = 0; q4 < PSD_C3_FPL; ++q4) {
                const int c = PSD_C3_FPL * lane + q4, jf = jhi - c;
                if (chl && c < nlinks) {
                    double* t = tab + (jf - jlo) * PSD_C3_TAB;
                    t[5] = zq[q4][0];
                    t[6] = zq[q4][1];
                    t[7] = zq[q4][2];
                    t[4] = (double)Ue[q4];
                    if (q4 > 0) t[3] = 0.0;
                    else if (c == 0) t[3] = (double)ez;
                }
            }
            if (chl) {
                const int cn = PSD_C3_FPL * (lane + 1);
                if (cn < nlinks) tab[(jhi - cn - jlo) * PSD_C3_TAB + 3] = (double)eout;
            }
            double zo0 = 0.0, zo1 = 0.0, zo2 = 0.0;
            int ecor = 0;
            if (fac) {
                const double* t = tab + lane * PSD_C3_TAB;
                zo0 = t[5];
                zo1 = t[6];
                zo2 = t[7];
                ecor = (int)t[3] + (int)t[4];
            }
            // the slice's last chain vector goes on to the slice below
            if (!lead && lane == 0) {
                psd_sl_put(outz + 0, zo0, tagz);
                psd_sl_put(outz + 1, zo1, tagz);
                psd_sl_put(outz + 2, zo2, tagz);
            }
            // ---- 3-reflectors: the chain's factors from their vectors, H_1 from x(1), lane nloc the neighbour's from the
            // vector that came in (the reflector of factor jhi + 1, or Q_1 for the top slice)
            double a0 = fac ? zo0 : (h1 ? x0 : zin[0]), a1 = fac ? zo1 : (h1 ? x1 : zin[1]), a2 = fac ? zo2 : (h1 ? x2 : zin[2]);
            const double tau = psd_refl3(a0, a1, a2);
            const double v1 = a1, v2 = a2;
            const double v1n = psd_c3_rol(v1), v2n = psd_c3_rol(v2), taun = psd_c3_rol(tau);
            const double a0n = psd_c3_rol(a0);  // (lane nloc: beta of the vector that came in)
            const double beta3 = fac ? psd_c3_beta(a0, a0n, ecor) : a0;
            if (fac) {
                double b00, b01, b10, b11;
                psd_c3_bblock(u00, u01, u02, u11, u12, u22, v1, v2, tau, v1n, v2n, taun, b00, b01, b10, b11);
                double* t = tab + lane * PSD_C3_TAB;
                t[3] = b00;
                t[4] = b01;
                t[6] = b10;
                t[7] = b11;
            }
            double Bq[PSD_C3_FPL][4], tq[PSD_C3_FPL][2];
#pragma unroll
            for (int q4 = 0; q4 < PSD_C3_FPL; ++q4) {
                const int c = PSD_C3_FPL * lane + q4, jf = jhi - c;
                double b00 = 1.0, b01 = 0.0, b10 = 0.0, b11 = 1.0;
                if (chl && c < nlinks) {
                    const double* t = tab + (jf - jlo) * PSD_C3_TAB;
                    b00 = t[3];
                    b01 = t[4];
                    b10 = t[6];
                    b11 = t[7];
                }
                Bq[q4][0] = b00; Bq[q4][1] = b01; Bq[q4][2] = b10; Bq[q4][3] = b11;
                tq[q4][0] = tq[q4][1] = 0.0;
            }
            // ---- scan 2: the 2-vector that enters this slice (e_1 for the top slice)
            double tin[2] = {1.0, 0.0};
            if (!top) (void)psd_sl_get(intv, 2, tagt, tin, err);
            const int et = psd_c3_expo(fmax(fabs(tin[0]), fabs(tin[1])));
            const double ts0 = top ? 1.0 : psd_c3_ldexp(tin[0], -et), ts1 = top ? 0.0 : psd_c3_ldexp(tin[1], -et);
            double t0 = 0.0, t1 = 0.0;
            int eout2 = 0;
            for (int s = 0; s < nsteps; ++s) {
                double w0 = psd_c3_shr(t0, ts0), w1 = psd_c3_shr(t1, ts1);
                if (s <= lane) {
#pragma unroll
                    for (int q4 = 0; q4 < PSD_C3_FPL; ++q4) {
                        const double n0 = __builtin_fma(Bq[q4][0], w0, Bq[q4][1] * w1);
                        const double n1_ = __builtin_fma(Bq[q4][2], w0, Bq[q4][3] * w1);
                        w0 = n0;
                        w1 = n1_;
                        tq[q4][0] = n0;
                        tq[q4][1] = n1_;
                    }
                    eout2 = psd_c3_expo(fmax(fabs(w0), fabs(w1)));
                    t0 = psd_c3_ldexp(w0, -eout2);
                    t1 = psd_c3_ldexp(w1, -eout2);
                }
            }
#pragma unroll
            for (int q4 = 0; q4 < PSD_C3_FPL; ++q4) {
                const int c = PSD_C3_FPL * lane + q4, jf = jhi - c;
                if (chl && c < nlinks) {
                    double* t = tab + (jf - jlo) * PSD_C3_TAB;
                    t[6] = tq[q4][0];
                    t[7] = tq[q4][1];
                    if (q4 > 0) t[3] = 0.0;
                    else if (c == 0) t[3] = top ? 0.0 : (double)et;
                }
            }
            if (chl) {
                const int cn = PSD_C3_FPL * (lane + 1);
                if (cn < nlinks) tab[(jhi - cn - jlo) * PSD_C3_TAB + 3] = (double)eout2;
            }
            double to0 = 0.0, to1 = 0.0;
            int ecor2 = 0;
            if (fac) {
                const double* t = tab + lane * PSD_C3_TAB;
                to0 = t[6];
                to1 = t[7];
                ecor2 = (int)t[3] + euo;
            }
            if (!lead && lane == 0) {
                psd_sl_put(outt + 0, to0, tagt);
                psd_sl_put(outt + 1, to1, tagt);
            }
            // ---- 2-reflectors: the chain's factors; lane nloc the neighbour's (none when the neighbour is H_1)
            const bool nb2 = lane == nloc && !top;
            double y0 = fac ? to0 : tin[0], y1 = fac ? to1 : tin[1];
            const double tau2 = (fac || nb2) ? psd_refl2(y0, y1) : 0.0;
            const double w2v = (fac || nb2) ? y1 : 0.0;
            // (the 2-vector that enters the top slice is e_1 itself: its "beta" is 1)
            const double y0n = psd_c3_rol((fac || nb2) ? y0 : 1.0);
            const double beta2 = fac ? psd_c3_beta(y0, y0n, ecor2) : 0.0;
            if (lane <= nloc) {
                double* t = tab + lane * PSD_C3_TAB;
                t[0] = v1;
                t[1] = v2;
                t[2] = tau;
                t[3] = w2v;
                t[4] = tau2;
                t[5] = beta3;
                t[6] = beta2;
            }
            if (lane < nloc) {  // the owners of this slice write their own lists
                const int j = jlo + lane;
                psd_tr tr;
                tr.pos = k;
                tr.kind = PSD_TR_R3;
                tr.c0 = v1;
                tr.c1 = v2;
                tr.c2 = tau;
                const int slot = (j == 1) ? (n1 + kk) : (nj + 2 * kk);
                if (slot < PSD_TR_CAP) psd_tr_store_global(trb + (size_t)(j - 1) * PSD_TR_CAP + slot, tr);
                if (j >= 2 && slot + 1 < PSD_TR_CAP) {
                    tr.pos = k + 1;
                    tr.kind = PSD_TR_H2;
                    tr.c0 = w2v;
                    tr.c1 = 0.0;
                    tr.c2 = tau2;
                    psd_tr_store_global(trb + (size_t)(j - 1) * PSD_TR_CAP + slot + 1, tr);
                }
            }
        }
        PSD_C3_BARRIER();
        psd_c3s_apply(wb, tab, 0, af, aq, tpf, nloc, lead, top, ld, bsz, bs, k, l, r0, nrw, ncl);
        PSD_C3_BARRIER();
        psd_c3s_apply(wb, tab, 1, af, aq, tpf, nloc, lead, top, ld, bsz, bs, k, l, r0, nrw, ncl);
        PSD_C3_BARRIER();
    }
}
#endif
