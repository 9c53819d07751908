// Scan chase (round 4): one position of the double-shift periodic QR sweep (PSD.jl:806-886) for ALL factors at once.
//
// The reference walks the factors one after the other: the reflector of factor j - 1 comes from column k of H_{j-1} as
// the right update by the reflector of factor j left it (PSD.jl:846-861) — a chain of p dependent (reflector, update)
// links per position, which is what the one- and two-wave chases of rounds 1-3 execute (~650 cycles per link).
// But a reflector Q = I - tau v v' that maps x to beta e_1 has Q e_1 = x / beta: the column the next factor sees is
//     H_{j-1}[k:k+2, k:k+2] Q_j e_1  =  U_{j-1} x(j) / beta_j ,     U_{j-1} the (upper triangular) 3 x 3 diagonal block,
// and a reflector does not change when its vector is scaled.  So the DIRECTIONS of all p bulge vectors of a position are
//     z(p) = U_p x(1),   z(j-1) = U_{j-1} z(j):   a chain of 3 x 3 triangular matrix-vector products,
// three dependent multiply-adds per factor, with no reflector, no square root and no update on it.  The same holds for
// the 2-reflectors (PSD.jl:865-881): y(j-1) = B_{j-1} y(j) / beta'_j with B_{j-1} the trailing 2 x 2 block of
// Q_{j-1}' U_{j-1} Q_j, which each factor forms from its own and its neighbour's 3-reflector.  Then every reflector is
// generated at once (one lane per factor), and all p factors are updated side by side by every wavefront of the workgroup:
// a left and a right transformation of one block commute, and different factors share nothing.
//
// Per position: scan 1 (p - 1 steps) -> 3-reflectors -> B blocks -> scan 2 (p - 1 steps) -> 2-reflectors   [wavefront 0]
//               barrier; right updates of H_2..H_p and the left update of H_1; barrier; left updates of H_2..H_p and the
//               right update of H_1; barrier                                                               [all waves]
// Numerics: the vector a reflector is generated from is U z (a product) instead of the updated column read back.  Both are
// U_{j-1} times the SAME computed first column of Q_j, so they agree to rounding in the scale of U_{j-1} — the size of the
// rounding errors of the reference's own update — and the entries the reflector is meant to annihilate are set to zero
// exactly as the reference does (:851-854,869).  Every transformation is orthogonal to working precision whatever its
// vector, so a loss of accuracy in a direction can cost convergence, never the decomposition.  The chain vectors are kept
// in range by powers of two (blocks scaled to unit maximum, the vector every 16 steps).
//
// Lanes of the scan wavefront: lane j - 1 = factor j.  The chain runs 1 -> p -> p - 1 -> ... -> 2, so the scan is a
// systolic array over lanes p - 1 .. 1 fed by wave_rol:1 (lane i reads lane i + 1): the lanes that are not factors 2..p
// (lane 0 and lanes >= p) are constant sources that emit x(1), which lane p - 1 picks up; step s is valid in lane
// p - 1 - s, and every lane keeps what it received and produced in ITS step.
#pragma once

#define PSD_C3_TAB 8      // doubles per factor in the reflector table: v2, v3, tau, w2, tau2, beta (factor 1), -, -
#define PSD_C3_MAXP 64    // one lane per factor
#define PSD_C3_MINP 2
#define PSD_C3_WAVES 4    // wavefronts of a chase workgroup under the scan chase (one per SIMD)
#define PSD_C3_FPL 4      // links of the factor chain per lane of the scan (16 lanes x 4 >= 63)

// exponent e with m 2^-e in [0.5, 1) for finite m > 0, else 0
PSD_D int psd_c3_expo(double m) {
    if (!(m > 0.0) || !(m < 1.7e308)) return 0;
#ifdef PSD_HOSTSIM
    int e = 0;
    (void)frexp(m, &e);
    return e;
#else
    return __builtin_amdgcn_frexp_exp(m);
#endif
}
PSD_D double psd_c3_ldexp(double x, int e) {
#ifdef PSD_HOSTSIM
    return ldexp(x, e);
#else
    return __builtin_amdgcn_ldexp(x, e);
#endif
}
PSD_D double psd_c3_max3(double a, double b, double c) { return fmax(fabs(a), fmax(fabs(b), fabs(c))); }

// trailing 2 x 2 block of Q' U Qn: Q = I - tau [1;v1;v2][1;v1;v2]' this factor's 3-reflector, Qn the one that acts on
// its columns (the next factor's, or H_1's for factor p), U upper triangular
PSD_D void psd_c3_bblock(double u00, double u01, double u02, double u11, double u12, double u22, double v1, double v2,
                         double tau, double v1n, double v2n, double taun, double& b00, double& b01, double& b10,
                         double& b11) {
    // R = U Qn = U - taun (U vn) vn'
    const double s0 = taun * (u00 + u01 * v1n + u02 * v2n);
    const double s1 = taun * (u11 * v1n + u12 * v2n);
    const double s2 = taun * (u22 * v2n);
    const double r01 = u01 - s0 * v1n, r02 = u02 - s0 * v2n;
    const double r11 = u11 - s1 * v1n, r12 = u12 - s1 * v2n;
    const double r21 = -s2 * v1n, r22 = u22 - s2 * v2n;
    // C = Q' R = R - tau v (v' R), rows 1..2, columns 1..2
    const double g1 = tau * (r01 + v1 * r11 + v2 * r21);
    const double g2 = tau * (r02 + v1 * r12 + v2 * r22);
    b00 = r11 - v1 * g1;
    b01 = r12 - v1 * g2;
    b10 = r21 - v2 * g1;
    b11 = r22 - v2 * g2;
}

// the two kinds of update items of the apply phase (window image in LDS; q points at the first of the three elements,
// sd is their stride: 1 along a column, ld along a row)
//   three elements under a 3-reflector, then (with2) the last two under a 2-reflector
//   fix 1: the column the 3-reflector was made for becomes (bfix, 0, 0); fix 2: the column of the 2-reflector becomes
//   (., bfix, 0).  bfix is the scan's own value of that entry (see "beta" in psd_c3_run): the reflector annihilates the
//   scan's vector exactly, so (beta, 0, 0) is the transformed column to the accuracy of one matrix-vector product; what
//   the update formulas leave in these three entries differs from it by the representation error of the neighbour's
//   reflector (several eps) and is not used.
PSD_D void psd_c3_item(double* q, int sd, double v1, double v2, double tau, bool with2, double w2, double tau2,
                       int fix /* 0 none, 1: (bfix, 0, 0), 2: (a1, bfix, 0) */, double bfix = 0.0) {
    double a1 = q[0], a2 = q[sd], a3 = q[2 * sd];
    const double xx = tau * (a1 + v1 * a2 + v2 * a3);
    a1 -= xx;
    a2 -= xx * v1;
    a3 -= xx * v2;
    if (with2) {
        const double yy = tau2 * (a2 + w2 * a3);
        a2 -= yy;
        a3 -= yy * w2;
    }
    if (fix == 1) {
        a1 = bfix;
        a2 = 0.0;
    }
    if (fix == 2) a2 = bfix;
    if (fix >= 1) a3 = 0.0;
    q[0] = a1;
    q[sd] = a2;
    q[2 * sd] = a3;
}

// The apply phase of one position.  The threads of the workgroup are dealt to the factors once per run: thread (f, q) is
// the q-th of the tpf threads of factor f + 1 and takes that factor's items q, q + tpf, ... — no index arithmetic per item,
// and a factor's reflectors are read from the table once per phase.  sub = 0: right updates of H_2..H_p (by the
// reflectors of the factor behind them in the chain: owner j + 1, or H_1's for factor p) and the left update of H_1 with
// its annihilated column k - 1; sub = 1: left updates of H_2..H_p, right update of H_1 (owner 2).  A left and a right
// update of one factor never share a phase, and within a phase the items of a factor are disjoint rows / columns.
PSD_D void psd_c3_apply(double* wb, const double* tab, int sub, int f, int q, int tpf, int p, int ld, int bsz, int bs,
                        int k, int l, int r0, int nrw, int ncl) {
    if (f >= p) return;
    const int j = f + 1;
    double* const blk = wb + f * bsz;
    const bool right = (j >= 2) == (sub == 0);
    if (right) {
        const int jo = (j == p) ? 1 : (j + 1);
        const double* t = tab + (jo - 1) * PSD_C3_TAB;
        const double v1 = t[0], v2 = t[1], tau = t[2], w2 = t[3], tau2 = t[4];
        double* const col = blk + (k - bs) * ld + (r0 - bs);
        for (int r = q; r < nrw; r += tpf) psd_c3_item(col + r, ld, v1, v2, tau, jo != 1, w2, tau2, 0);
    } else {
        const double* t = tab + f * PSD_C3_TAB;
        const double v1 = t[0], v2 = t[1], tau = t[2], w2 = t[3], tau2 = t[4];
        double* const row = blk + (k - bs) * ld + (k - bs);
        if (j == 1) {
            for (int cc = q; cc < ncl; cc += tpf) psd_c3_item(row + cc * ld, 1, v1, v2, tau, false, 0.0, 0.0, 0);
            if (k > l && q == tpf - 1) {  // PSD.jl:822-827
                double* c = blk + (k - 1 - bs) * ld + (k - bs);
                c[0] = t[5];
                c[1] = 0.0;
                c[2] = 0.0;
            }
        } else {
            // column k: (beta, 0, 0) (PSD.jl:851-854); column k + 1: (., beta', 0) (:869); the others: both reflectors
            const double b3 = t[5], b2 = t[6];
            for (int cc = q; cc < ncl; cc += tpf)
                psd_c3_item(row + cc * ld, 1, v1, v2, tau, cc >= 1, w2, tau2, (cc == 0) ? 1 : ((cc == 1) ? 2 : 0), (cc == 0) ? b3 : b2);
        }
    }
}

// beta of a chain vector from the beta of the vector that produced it: the column a reflector is made for is
// U z_in / beta(z_in) (first column of the exact reflector of z_in), the scan computed z = 2^-e U z_in, so the transformed
// column's leading entry is beta(z) 2^e / beta(z_in)
PSD_D double psd_c3_beta(double bz, double bin, int e) {
    if (bin == 0.0) return 0.0;
    return psd_c3_ldexp(bz / bin, e);
}

#ifndef PSD_HOSTSIM
// lane i <- lane (i + 1) mod 64.  (Inline assembly: through the builtin the compiler first copies the old value into the
// destination — a rotation has no lane without a source —, which doubled the moves on the scan's critical path.)
PSD_D double psd_c3_rol(double v) {
    const int lo = __double2loint(v), hi = __double2hiint(v);
    int rlo, rhi;
    // (s_nop 1: a DPP read of a register a vector instruction has just written needs two wait states, and the
    //  compiler's hazard pass does not look into inline assembly)
    asm volatile("s_nop 1\n\tv_mov_b32_dpp %0, %2 wave_rol:1 row_mask:0xf bank_mask:0xf\n\tv_mov_b32_dpp %1, %3 wave_rol:1 row_mask:0xf bank_mask:0xf"
                 : "=&v"(rlo), "=&v"(rhi)
                 : "v"(lo), "v"(hi));
    return __hiloint2double(rhi, rlo);
}
// lane i <- lane i - 1 inside a row of 16 lanes; lane 0 of a row (no source) gets `first`
PSD_D double psd_c3_shr(double v, double first) {
    int lo = __double2loint(v), hi = __double2hiint(v);
    lo = __builtin_amdgcn_update_dpp(__double2loint(first), lo, 0x111, 0xf, 0xf, false);
    hi = __builtin_amdgcn_update_dpp(__double2hiint(first), hi, 0x111, 0xf, 0xf, false);
    return __hiloint2double(hi, lo);
}
// full barrier of the chase workgroup: every LDS operation of every wavefront is done behind it
#define PSD_C3_BARRIER() PSD_PAIR_BARRIER()

// One run = positions ks .. ks + npos - 1 of a window, all with three-row bulges.  Called by every wavefront of the
// workgroup (wv = its index, nw their number) with the same command block.
PSD_D void psd_c3_run(const psd_c2& Cin, int wv_, int nw_, int taboff_) {
    PSD_LDS_DECL;
    const int wv = PSD_C2_UNI(wv_), nw = PSD_C2_UNI(nw_), taboff = PSD_C2_UNI(taboff_);
    const int p = PSD_C2_UNI(Cin.p), ld = PSD_C2_UNI(Cin.ld), bsz = PSD_C2_UNI(Cin.bsz), bs = PSD_C2_UNI(Cin.bs);
    const int l = PSD_C2_UNI(Cin.l), ie = PSD_C2_UNI(Cin.i), ks = PSD_C2_UNI(Cin.ks), npos = PSD_C2_UNI(Cin.npos);
    const int c1max = PSD_C2_UNI(Cin.c1max), r0 = PSD_C2_UNI(Cin.r0), n1 = PSD_C2_UNI(Cin.n1), nj = PSD_C2_UNI(Cin.nj);
    double* const wb = (double*)(psd_lds + PSD_C2_UNI(Cin.wboff));
    double* const tab = (double*)(psd_lds + taboff);
    psd_tr* const trb = Cin.tr;
    const int lane = (int)threadIdx.x;
    const int tid = wv * 64 + lane, NT = nw * 64;
    const int tpf = (NT / p > 0) ? (NT / p) : 1;  // threads per factor in the apply phases (p <= 64 <= NT)
    const int af = tid / tpf, aq = tid - af * tpf;
    long long* const dbg = Cin.dbg;
    long long dacc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#define PSD_C3_T(i) do { if (dbg) { const long long t__ = psd_clock(); dacc[i] += t__ - dt0; dt0 = t__; } } while (0)
    for (int kk = 0; kk < npos; ++kk) {
        const int k = ks + kk;
        const int rlim = (k + 3 < ie) ? (k + 3) : ie;
        const int nrw = rlim - r0 + 1;
        int ncl = c1max - k + 1;
        if (ncl < 0) ncl = 0;
        long long dt0 = dbg ? psd_clock() : 0;
        if (wv == 0) {
            const bool fac = lane >= 1 && lane < p;  // factors 2..p; the other lanes emit x(1)
            double x0, x1, x2;
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
            const int ex = psd_c3_expo(psd_c3_max3(x0, x1, x2));
            const double xs0 = psd_c3_ldexp(x0, -ex), xs1 = psd_c3_ldexp(x1, -ex), xs2 = psd_c3_ldexp(x2, -ex);
            // this lane's own factor (the B block below needs it)
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
                if (!(um > 1e-18 && um < 1e18)) {  // (far from unit scale: see the chain lanes' blocks below)
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
            // ---- scan 1 on the chain lanes.  The chain 1 -> p -> p - 1 -> ... -> 2 has p - 1 links; chain lane i < 16 does
            // links 4 i .. 4 i + 3 (factors p - 4 i, ...) one after the other in registers and hands its last vector to lane
            // i + 1 by row_shr:1 — a full-rate DPP move inside a row of 16 lanes (a wave-wide shift measured four times
            // slower: with one factor per lane the six moves of a step were most of it).  Lane i is fed valid data in step
            // i; it recomputes while s <= i and keeps its four vectors from then on.  Links beyond the chain are identities.
            double U[PSD_C3_FPL][6], zq[PSD_C3_FPL][3];
            int Ue[PSD_C3_FPL];
            const bool chl = lane < 16;
#pragma unroll
            for (int q4 = 0; q4 < PSD_C3_FPL; ++q4) {
                const int c = PSD_C3_FPL * lane + q4, jf = p - c;  // link c = factor jf
                double a00 = 1.0, a01 = 0.0, a02 = 0.0, a11 = 1.0, a12 = 0.0, a22 = 1.0;
                Ue[q4] = 0;
                if (chl && c < p - 1) {
                    const double* q = wb + (jf - 1) * bsz + (k - bs) * ld + (k - bs);
                    a00 = q[0];
                    a01 = q[ld];
                    a11 = q[ld + 1];
                    a02 = q[2 * ld];
                    a12 = q[2 * ld + 1];
                    a22 = q[2 * ld + 2];
                    const double um = fmax(psd_c3_max3(a00, a01, a02), psd_c3_max3(a11, a12, a22));
                    // (blocks are brought to unit scale only when they are far from it: the vector is rescaled after
                    //  every lane's four links anyway, and the scaling is a dozen instructions per block)
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
            const int nsteps = (p - 1 + PSD_C3_FPL - 1) / PSD_C3_FPL;
            double z0 = 0.0, z1 = 0.0, z2 = 0.0;
            int eout = 0;  // power of two this lane took out of the vector it handed on
            for (int s = 0; s < nsteps; ++s) {
                // (lane 0 of a row has no lane to its right... left: it keeps the "old" operand, the start vector)
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
            // the chain vectors to their factors' lanes through the table (slots 5..7 of a factor: free until the end),
            // with the powers of two between a factor's vector and the product it stands for: slot 4 the scaling of its
            // block, slot 3 what the vector that ENTERED its link had been scaled by (the start vector's ex for factor p,
            // a lane's hand-over scaling for the first link of the next lane, nothing inside a lane)
#pragma unroll
            for (int q4 = 0; q4 < PSD_C3_FPL; ++q4) {
                const int c = PSD_C3_FPL * lane + q4, jf = p - c;
                if (chl && c < p - 1) {
                    double* t = tab + (jf - 1) * PSD_C3_TAB;
                    t[5] = zq[q4][0];
                    t[6] = zq[q4][1];
                    t[7] = zq[q4][2];
                    t[4] = (double)Ue[q4];
                    if (q4 > 0) t[3] = 0.0;
                    else if (c == 0) t[3] = (double)ex;
                }
            }
            if (chl) {
                const int cn = PSD_C3_FPL * (lane + 1);  // first link of the next lane
                if (cn < p - 1) tab[(p - cn - 1) * PSD_C3_TAB + 3] = (double)eout;
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
            PSD_C3_T(0);
            // ---- 3-reflectors: factor lanes from their chain vector, the others Q_1 from x(1)
            double a0 = fac ? zo0 : x0, a1 = fac ? zo1 : x1, a2 = fac ? zo2 : x2;
            const double tau = psd_refl3(a0, a1, a2);  // (a0, a1, a2) <- (beta, v2, v3)
            const double v1 = a1, v2 = a2;
            const double v1n = psd_c3_rol(v1), v2n = psd_c3_rol(v2), taun = psd_c3_rol(tau);
            // the entry the reflector leaves in the annihilated column (psd_c3_beta): from this lane's and its chain
            // neighbour's beta (lane + 1: the factor above, H_1's true x(1) above factor p)
            const double a0n = psd_c3_rol(a0);
            const double beta3 = fac ? psd_c3_beta(a0, a0n, ecor) : a0;
            // ---- scan 2 on the trailing 2 x 2 blocks: B blocks by the factors' lanes, through the table (slots 3, 4, 6, 7)
            // to the chain lanes, the same systolic chain with 2-vectors, the results back through slots 6, 7
            if (fac) {
                double b00, b01, b10, b11;
                psd_c3_bblock(u00, u01, u02, u11, u12, u22, v1, v2, tau, v1n, v2n, taun, b00, b01, b10, b11);
                double* t = tab + lane * PSD_C3_TAB;
                t[3] = b00;
                t[4] = b01;
                t[6] = b10;
                t[7] = b11;
            }
            PSD_C3_T(1);
            double Bq[PSD_C3_FPL][4], tq[PSD_C3_FPL][2];
#pragma unroll
            for (int q4 = 0; q4 < PSD_C3_FPL; ++q4) {
                const int c = PSD_C3_FPL * lane + q4, jf = p - c;
                double b00 = 1.0, b01 = 0.0, b10 = 0.0, b11 = 1.0;
                if (chl && c < p - 1) {
                    const double* t = tab + (jf - 1) * PSD_C3_TAB;
                    b00 = t[3];
                    b01 = t[4];
                    b10 = t[6];
                    b11 = t[7];
                }
                Bq[q4][0] = b00; Bq[q4][1] = b01; Bq[q4][2] = b10; Bq[q4][3] = b11;
                tq[q4][0] = tq[q4][1] = 0.0;
            }
            double t0 = 0.0, t1 = 0.0;
            int eout2 = 0;
            for (int s = 0; s < nsteps; ++s) {
                double w0 = psd_c3_shr(t0, 1.0), w1 = psd_c3_shr(t1, 0.0);
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
                const int c = PSD_C3_FPL * lane + q4, jf = p - c;
                if (chl && c < p - 1) {
                    double* t = tab + (jf - 1) * PSD_C3_TAB;
                    t[6] = tq[q4][0];
                    t[7] = tq[q4][1];
                    if (q4 > 0 || c == 0) t[3] = 0.0;  // (scaling of the vector that entered the link: slot 3 as in scan 1)
                }
            }
            if (chl) {
                const int cn = PSD_C3_FPL * (lane + 1);
                if (cn < p - 1) tab[(p - cn - 1) * PSD_C3_TAB + 3] = (double)eout2;
            }
            double to0 = 0.0, to1 = 0.0;
            int ecor2 = 0;
            if (fac) {
                const double* t = tab + lane * PSD_C3_TAB;
                to0 = t[6];
                to1 = t[7];
                ecor2 = (int)t[3] + euo;  // (the B block was formed from this lane's own copy of the factor's block)
            }
            PSD_C3_T(2);
            double y0 = to0, y1 = to1;
            const double tau2 = fac ? psd_refl2(y0, y1) : 0.0;  // (y0, y1) <- (beta', w2)
            const double w2v = fac ? y1 : 0.0;
            // (the 2-vector that entered factor p's link is e_1 itself: its "beta" is 1)
            const double y0n = psd_c3_rol(fac ? y0 : 1.0);
            const double beta2 = fac ? psd_c3_beta(y0, y0n, ecor2) : 0.0;
            if (lane < p) {
                double* t = tab + lane * PSD_C3_TAB;
                t[0] = v1;
                t[1] = v2;
                t[2] = tau;
                t[3] = w2v;
                t[4] = tau2;
                t[5] = beta3;  // (H_1: beta of the true x(1))
                t[6] = beta2;
                psd_tr tr;
                tr.pos = k;
                tr.kind = PSD_TR_R3;
                tr.c0 = v1;
                tr.c1 = v2;
                tr.c2 = tau;
                const int slot = (lane == 0) ? (n1 + kk) : (nj + 2 * kk);
                if (slot < PSD_TR_CAP) psd_tr_store_global(trb + (size_t)lane * PSD_TR_CAP + slot, tr);
                if (lane >= 1 && slot + 1 < PSD_TR_CAP) {
                    tr.pos = k + 1;
                    tr.kind = PSD_TR_H2;
                    tr.c0 = w2v;
                    tr.c1 = 0.0;
                    tr.c2 = tau2;
                    psd_tr_store_global(trb + (size_t)lane * PSD_TR_CAP + slot + 1, tr);
                }
            }
        }
        if (wv == 0) PSD_C3_T(3);
        PSD_C3_BARRIER();
        if (wv == 0) PSD_C3_T(4);
        psd_c3_apply(wb, tab, 0, af, aq, tpf, p, ld, bsz, bs, k, l, r0, nrw, ncl);
        PSD_C3_BARRIER();
        if (wv == 0) PSD_C3_T(5);
        psd_c3_apply(wb, tab, 1, af, aq, tpf, p, ld, bsz, bs, k, l, r0, nrw, ncl);
        PSD_C3_BARRIER();
        if (wv == 0) PSD_C3_T(6);
    }
    if (dbg && wv == 0 && lane == 0) {
        for (int q = 0; q < 7; ++q) psd_atomic_add_ll(dbg + q, dacc[q]);
        psd_atomic_add_ll(dbg + 7, (long long)npos);
    }
#undef PSD_C3_T
}
#else
// The simulated tier: the same steps with the chain as a plain loop over the factors (rescaled after every PSD_C3_FPL
// links, where the device's chain lanes hand over).
PSD_D void psd_c3_run(const psd_c2& Cin, int, int, int taboff) {
    PSD_LDS_DECL;
    const int p = Cin.p, ld = Cin.ld, bsz = Cin.bsz, bs = Cin.bs, l = Cin.l, ie = Cin.i, ks = Cin.ks, npos = Cin.npos;
    const int c1max = Cin.c1max, r0 = Cin.r0, n1 = Cin.n1, nj = Cin.nj;
    double* const wb = (double*)(psd_lds + Cin.wboff);
    double* const tab = (double*)(psd_lds + taboff);
    psd_tr* const trb = Cin.tr;
    for (int kk = 0; kk < npos; ++kk) {
        const int k = ks + kk;
        const int rlim = (k + 3 < ie) ? (k + 3) : ie;
        const int nrw = rlim - r0 + 1;
        int ncl = c1max - k + 1;
        if (ncl < 0) ncl = 0;
        double x0, x1, x2;
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
        const int ex = psd_c3_expo(psd_c3_max3(x0, x1, x2));
        // factor j at index j (1..p); index 1 = H_1
        double U[65][6], zo[65][3], v1[65], v2[65], tau[65], beta[65], bz[65] = {0.0};
        int ecor[65], eu[65];
        for (int j = 2; j <= p; ++j) {
            const double* q = wb + (j - 1) * bsz + (k - bs) * ld + (k - bs);
            double u[6] = {q[0], q[ld], q[2 * ld], q[ld + 1], q[2 * ld + 1], q[2 * ld + 2]};  // u00 u01 u02 u11 u12 u22
            const double um = fmax(psd_c3_max3(u[0], u[1], u[2]), psd_c3_max3(u[3], u[4], u[5]));
            eu[j] = (um > 1e-18 && um < 1e18) ? 0 : psd_c3_expo(um);
            for (int t = 0; t < 6; ++t) U[j][t] = psd_c3_ldexp(u[t], -eu[j]);
        }
        {   // scan 1
            double w[3] = {psd_c3_ldexp(x0, -ex), psd_c3_ldexp(x1, -ex), psd_c3_ldexp(x2, -ex)};
            int ein = ex;
            for (int c = 0; c < p - 1; ++c) {
                const int j = p - c;
                const double* u = U[j];
                const double n0 = u[0] * w[0] + (u[1] * w[1] + u[2] * w[2]);
                const double n1_ = u[3] * w[1] + u[4] * w[2];
                const double n2 = u[5] * w[2];
                zo[j][0] = n0; zo[j][1] = n1_; zo[j][2] = n2;
                ecor[j] = ein + eu[j];
                if ((c % PSD_C3_FPL) == PSD_C3_FPL - 1) {
                    ein = psd_c3_expo(psd_c3_max3(n0, n1_, n2));
                    w[0] = psd_c3_ldexp(n0, -ein); w[1] = psd_c3_ldexp(n1_, -ein); w[2] = psd_c3_ldexp(n2, -ein);
                } else {
                    ein = 0;
                    w[0] = n0; w[1] = n1_; w[2] = n2;
                }
            }
        }
        for (int j = 1; j <= p; ++j) {
            double a0 = (j == 1) ? x0 : zo[j][0], a1 = (j == 1) ? x1 : zo[j][1], a2 = (j == 1) ? x2 : zo[j][2];
            tau[j] = psd_refl3(a0, a1, a2);
            v1[j] = a1;
            v2[j] = a2;
            bz[j] = a0;
        }
        beta[1] = bz[1];
        for (int j = 2; j <= p; ++j) beta[j] = psd_c3_beta(bz[j], bz[(j == p) ? 1 : (j + 1)], ecor[j]);
        double B[65][4], to[65][2], w2v[65], tau2[65], beta2[65], by[65];
        int ecor2[65];
        for (int j = 2; j <= p; ++j) {
            const int nb = (j == p) ? 1 : (j + 1);
            const double* u = U[j];
            psd_c3_bblock(u[0], u[1], u[2], u[3], u[4], u[5], v1[j], v2[j], tau[j], v1[nb], v2[nb], tau[nb], B[j][0], B[j][1],
                          B[j][2], B[j][3]);
        }
        {   // scan 2
            double w[2] = {1.0, 0.0};
            int ein = 0;
            for (int c = 0; c < p - 1; ++c) {
                const int j = p - c;
                const double n0 = B[j][0] * w[0] + B[j][1] * w[1];
                const double n1_ = B[j][2] * w[0] + B[j][3] * w[1];
                to[j][0] = n0; to[j][1] = n1_;
                ecor2[j] = ein + eu[j];
                if ((c % PSD_C3_FPL) == PSD_C3_FPL - 1) {
                    ein = psd_c3_expo(fmax(fabs(n0), fabs(n1_)));
                    w[0] = psd_c3_ldexp(n0, -ein); w[1] = psd_c3_ldexp(n1_, -ein);
                } else {
                    ein = 0;
                    w[0] = n0; w[1] = n1_;
                }
            }
        }
        w2v[1] = 0.0; tau2[1] = 0.0; beta2[1] = 0.0;
        for (int j = 2; j <= p; ++j) {
            double y0 = to[j][0], y1 = to[j][1];
            tau2[j] = psd_refl2(y0, y1);
            w2v[j] = y1;
            by[j] = y0;
        }
        for (int j = 2; j <= p; ++j) beta2[j] = psd_c3_beta(by[j], (j == p) ? 1.0 : by[j + 1], ecor2[j]);
        for (int j = 1; j <= p; ++j) {
            const int lane = j - 1;
            double* t = tab + lane * PSD_C3_TAB;
            t[0] = v1[j];
            t[1] = v2[j];
            t[2] = tau[j];
            t[3] = w2v[j];
            t[4] = tau2[j];
            t[5] = beta[j];
            t[6] = beta2[j];
            psd_tr tr;
            tr.pos = k;
            tr.kind = PSD_TR_R3;
            tr.c0 = v1[j];
            tr.c1 = v2[j];
            tr.c2 = tau[j];
            const int slot = (lane == 0) ? (n1 + kk) : (nj + 2 * kk);
            if (slot < PSD_TR_CAP) psd_tr_store_global(trb + (size_t)lane * PSD_TR_CAP + slot, tr);
            if (lane >= 1 && slot + 1 < PSD_TR_CAP) {
                tr.pos = k + 1;
                tr.kind = PSD_TR_H2;
                tr.c0 = w2v[j];
                tr.c1 = 0.0;
                tr.c2 = tau2[j];
                psd_tr_store_global(trb + (size_t)lane * PSD_TR_CAP + slot + 1, tr);
            }
        }
        {
            const int NT = 64 * PSD_C3_WAVES, tpf = (NT / p > 0) ? (NT / p) : 1;
            for (int sub = 0; sub < 2; ++sub)
                for (int tid = 0; tid < NT; ++tid)
                    psd_c3_apply(wb, tab, sub, tid / tpf, tid % tpf, tpf, p, ld, bsz, bs, k, l, r0, nrw, ncl);
        }
    }
}
#endif
