// Eigenvalue reordering on the GPU, ComplexF64: ordschur!(P, select) by adjacent 1x1 swaps.
//
// Replaces /root/reference/src/ordschur.jl:11-73 (driver: bubble every selected eigenvalue to the
// top), :317-322 (_swapschur1!), sylswap.jl:542-635 (_swapadj1x1g!: periodic Sylvester solution ->
// one Givens rotation per factor, weak + strong stability tests, application to full rows/columns),
// sylvester.jl:124-138,196-205 (_psyl1rep/_psylsolve1), ordschur.jl:97-120 (_updateλ!).
//
// MI355X structure: an eigenvalue moving up from position `here` is a chase.  One wavefront takes
// the 2x2 diagonal blocks of a window of up to W-1 consecutive swaps from LDS, solves each cyclic
// p x p Sylvester system by a structured Givens QR in O(p) (the reference forms the dense p x p
// matrix and calls qr!, O(p^3) per swap), runs the stability tests, updates the window and emits
// the rotations; psd_zq_apply then updates the off-window rows/columns of T_m, T_{m-1}, Z_m.
//
// Index conventions: device arrays are in the engine's internal right order (T_1 = Schur factor,
// T_j = Z_j' A_j Z_{j+1}).  The reference works on the aliased left-oriented sequence X_1..X_p with
// X_1 = T_1 (utils.jl:25-85); X_l = T_{sigma(l)}, sigma(1) = 1, sigma(l) = p+2-l, and the rotation
// G_l of sylswap.jl:618-628 (right on X_l, left on X_{l-1}, right on its Z) is owned by
// m(l) = 2 for l = 1, 1 for l = 2, p+3-l otherwise: left on T_m, right on T_{m-1}, Z_m.
#pragma once
#include "psd_zqz.h"

enum { PSD_OPH_SCAN = 0, PSD_OPH_MOVE = 1, PSD_OPH_IDLE = 5, PSD_OPH_DONE = 7 };
#define PSD_INFO_ILLCOND_BASE 2000
#define PSD_INFO_SINGULAR 3000

struct psd_ostate {
    int n, p, wantZ, W;
    int phase, info;
    int j, js, here;
    int nswaps, nwindows;
};

struct psd_oparams {
    psd_zparams z;            // H, Z, desc, tr, cnt (shared with the QZ kernels), alpha/beta/ascale
    psd_ostate* st;
    const unsigned char* select;  // [n]
};

PSD_HD int psd_ord_sigma(int p, int l) { return (l == 1) ? 1 : (p + 2 - l); }
PSD_HD int psd_ord_owner(int p, int l) {
    if (p == 1) return 1;
    if (l == 1) return 2;
    if (l == 2) return 1;
    return p + 3 - l;
}

// Solve A_k x_k - B_k x_{k+1} = -C_k (k = 1..K, x_{K+1} = x_1): cyclic bidiagonal system, Givens QR on
// rows (k, K) eliminating the corner row, then back substitution.  d/e/f: diagonal, superdiagonal and
// last-column work arrays.  Returns false on an exactly zero pivot (utils.jl:123-131).
PSD_D bool psd_ord_cycsolve(int K, const psd_z* A, const psd_z* B, const psd_z* C, psd_z* x, psd_z* d, psd_z* e,
                            psd_z* f, psd_z* rhs) {
    // rows k = 1..K-1: d[k] x_k + e[k] x_{k+1} (+ f[k] x_K) = rhs[k];  row K: bottom row (sparse: lo at column k, hi at K)
    for (int k = 0; k < K; ++k) {
        d[k] = A[k];
        e[k] = zneg(B[k]);
        f[k] = zmk(0.0, 0.0);
        rhs[k] = zneg(C[k]);
    }
    // bottom row K: entry at column 1 is -B_K, at column K is A_K
    psd_z lo = zneg(B[K - 1]);  // current leading nonzero of the bottom row, at column k
    psd_z hi = A[K - 1];        // entry of the bottom row in column K
    psd_z rb = rhs[K - 1];
    for (int k = 0; k < K - 1; ++k) {
        // rotate rows k and K to annihilate `lo` (column k) against d[k]
        double c;
        psd_z s, r;
        psd_zgivens(d[k], lo, c, s, r);
        // row k: (d[k], e[k], f[k]) ; bottom: (lo, nxt, hi) where nxt is its entry in column k+1 (currently 0)
        const bool lastcol = (k + 1 == K - 1);  // column k+1 is column K
        psd_z ek = e[k], fk = f[k];
        psd_z nxt = zmk(0.0, 0.0);
        if (lastcol) {  // e[k] and f[k] refer to the same column K: fold
            ek = zadd(ek, fk);
            fk = zmk(0.0, 0.0);
            nxt = hi;
        }
        d[k] = r;
        // column k+1
        psd_z t1 = ek, t2 = nxt;
        psd_zrot_left(c, s, t1, t2);
        e[k] = t1;
        psd_z newlo = t2;
        // column K (only when distinct from column k+1)
        if (!lastcol) {
            psd_z u1 = fk, u2 = hi;
            psd_zrot_left(c, s, u1, u2);
            f[k] = u1;
            hi = u2;
        } else {
            f[k] = zmk(0.0, 0.0);
            hi = newlo;
        }
        psd_z q1 = rhs[k], q2 = rb;
        psd_zrot_left(c, s, q1, q2);
        rhs[k] = q1;
        rb = q2;
        lo = newlo;
    }
    // now the bottom row has a single entry `hi` in column K (for K == 1: A_1 - B_1)
    if (K == 1) hi = zsub(A[0], B[0]);
    if (ziszero(hi)) return false;
    x[K - 1] = zdiv(rb, hi);
    for (int k = K - 2; k >= 0; --k) {
        if (ziszero(d[k])) return false;
        psd_z t = zsub(rhs[k], zmul(e[k], x[k + 1]));
        if (k + 1 != K - 1) t = zsub(t, zmul(f[k], x[K - 1]));
        x[k] = zdiv(t, d[k]);
    }
    return true;
}

PSD_D void psd_zord_step_body(const psd_oparams& O) {
    PSD_LDS_DECL;
    const psd_zparams& P = O.z;
    psd_ostate st = *O.st;
    PSD_ONE { P.desc->active = 0; P.desc->defer_run = 0; }
    if (st.phase == PSD_OPH_DONE || st.phase == PSD_OPH_IDLE) return;
    const int n = st.n, p = st.p;
    psd_z* ldsz = (psd_z*)psd_lds;
    const size_t winb = (size_t)p * st.W * (st.W + 1);
    psd_z* sc = ldsz + winb;  // scratch: 13 arrays of p
    psd_z *T11 = sc, *T12 = sc + p, *T22 = sc + 2 * p, *Xv = sc + 3 * p, *Gs = sc + 4 * p, *wd = sc + 5 * p,
          *we = sc + 6 * p, *wf = sc + 7 * p, *wr = sc + 8 * p;
    psd_z* Txx = sc + 9 * p;  // 4 p
    double* Gc = (double*)(sc + 13 * p);  // p doubles
    int* lcnt = (int*)(Gc + p + 2);  // Gc[p] carries the swap verdict
    // SCAN (ordschur.jl:53-65): next selected eigenvalue and its destination
    while (st.phase == PSD_OPH_SCAN) {
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
        const int ilo = (hi - nb > st.js) ? (hi - nb) : st.js;  // swaps at i = hi-1 .. ilo
        psd_zwin w;
        w.b = ldsz;
        w.W = st.W;
        w.ld = st.W + 1;
        w.bsz = st.W * (st.W + 1);
        w.bs = ilo;
        w.be = hi;
        PSD_PAR_FOR(m, p) { lcnt[m] = 0; }
        psd_zwin_load(P, w, n, p);
        bool failed = false;
        for (int i = hi - 1; i >= ilo && !failed; --i) {
            PSD_SYNC();
            PSD_PAR_FOR(t, p) {
                const int l = t + 1, sg = psd_ord_sigma(p, l);
                T11[t] = w.at(sg, i, i);
                T12[t] = w.at(sg, i, i + 1);
                T22[t] = w.at(sg, i + 1, i + 1);
            }
            PSD_SYNC();
            // sylswap.jl:559-617: scalar part, one lane
            PSD_ONE {
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
                if (p > 1) {
                    if (!psd_ord_cycsolve(p, T11, T22, T12, Xv, wd, we, wf, wr)) flag = 2;
                    if (!flag)
                        for (int t = 0; t < p; ++t) psd_zgivens(Xv[t], zmk(1.0, 0.0), Gc[t], Gs[t], r);
                } else {
                    psd_zgivens(T12[0], zsub(T22[0], T11[0]), Gc[0], Gs[0], r);
                }
                if (!flag) {
                    // 2x2 working copies: Txx[l] <- rmul!(Txx[l], G_l'); Txx[l-1] <- lmul!(G_l, Txx[l-1])
                    for (int t = 0; t < p; ++t) {
                        Txx[4 * t + 0] = T11[t];
                        Txx[4 * t + 1] = T12[t];
                        Txx[4 * t + 2] = zmk(0.0, 0.0);
                        Txx[4 * t + 3] = T22[t];
                    }
                    for (int t = 0; t < p; ++t) {
                        psd_z* m = Txx + 4 * t;
                        psd_zrot_right_adj(Gc[t], Gs[t], m[0], m[1]);
                        psd_zrot_right_adj(Gc[t], Gs[t], m[2], m[3]);
                        psd_z* q = Txx + 4 * ((t == 0) ? (p - 1) : (t - 1));
                        psd_zrot_left(Gc[t], Gs[t], q[0], q[2]);
                        psd_zrot_left(Gc[t], Gs[t], q[1], q[3]);
                    }
                    double ws = 0.0;
                    for (int t = 0; t < p; ++t) ws += zabs(Txx[4 * t + 2]);
                    if (ws > thresh) flag = 1;
                    // strong test: W_{l+1} Txx[l] W_l' against the original blocks, W_l = [c -s; conj(s) c]
                    double ss = 0.0;
                    for (int t = 0; t < p; ++t) {
                        const int t1 = (t == p - 1) ? 0 : (t + 1);
                        const psd_z a = zmk(Gc[t1], 0.0), b = zneg(Gs[t1]), cc = zconj(Gs[t1]), dd = zmk(Gc[t1], 0.0);
                        const psd_z* m = Txx + 4 * t;
                        const psd_z p0 = zadd(zmul(a, m[0]), zmul(b, m[2])), p1 = zadd(zmul(a, m[1]), zmul(b, m[3]));
                        const psd_z p2 = zadd(zmul(cc, m[0]), zmul(dd, m[2])), p3 = zadd(zmul(cc, m[1]), zmul(dd, m[3]));
                        // times W_l' = [c s; -conj(s) c]
                        const psd_z e0 = zmk(Gc[t], 0.0), e1 = Gs[t], e2 = zneg(zconj(Gs[t])), e3 = zmk(Gc[t], 0.0);
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
                Gc[p] = (double)flag;  // Gc has p+1 slots
            }
            PSD_SYNC();
            const int flag = (int)Gc[p];
            if (flag) {
                st.info = (flag == 2) ? PSD_INFO_SINGULAR : (PSD_INFO_ILLCOND_BASE + st.j);
                failed = true;
                break;
            }
            // sylswap.jl:618-633 inside the window; owner m(l): left on T_m, right on T_{m-1}
            for (int l = 1; l <= p; ++l) {
                const int m = psd_ord_owner(p, l);
                const int mm1 = (m == 1) ? p : (m - 1);
                const double c = Gc[l - 1];
                const psd_z s = Gs[l - 1];
                psd_zwin_right(w, mm1, i, c, s, 1, i + 1);
                psd_zwin_left(w, m, i, c, s, i, n);
                psd_zrecord(P, lcnt, m, i, c, s);
            }
            PSD_PAR_FOR(t, p) { w.at(t + 1, i + 1, i) = zmk(0.0, 0.0); }
            PSD_SYNC();
            st.nswaps += 1;
        }
        if (failed) {
            st.phase = PSD_OPH_DONE;
        } else {
            psd_zwin_store(P, w, n, p);
            PSD_SYNC();
            PSD_PAR_FOR(m, p) { P.cnt[m] = lcnt[m]; }
            PSD_ONE {
                psd_zapply_desc d;
                d.active = 1;
                d.plo = ilo;
                d.phi = hi;
                d.lc0 = hi + 1;
                d.lc1 = n;
                d.rr0 = 1;
                d.rr1 = ilo - 1;
                d.rcut = d.rr0;
                d.zr0 = 1;
                d.zr1 = st.wantZ ? n : 0;
                d.defer_h1 = 0;
                d.defer_run = 0;
                d.djlo = d.djhi = d.drow0 = 0;
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

PSD_KERNEL_B(PSD_STEP_NT) psd_zord_step(psd_oparams O) { psd_zord_step_body(O); }

// ------------------------------------------------------------------------------------------------
// Pipelined driver (as psd_roslot of the real engine, simpler: blocks are 1 x 1 and the k-th selected eigenvalue goes to
// row k, known when it starts).  Up to PSD_O_SLOTS selected eigenvalues travel upwards at once, one workgroup each, a
// follower's window ending below its predecessor's last known row.  The step kernel is the serial one: slot s runs it on
// its own state (phase MOVE, js = the top row its window may reach, j = its source row), descriptor, counts and lists; a
// one-lane plan kernel in front of every tick reads what the slots did, retires arrivals in order, starts at most one
// new eigenvalue and writes the slots' states for the coming tick.  Shared by the signed driver (psd_zgord_step_mb).
#define PSD_O_SLOTS 64
struct psd_oslot {
    int active, seq, target, jsrc, landed, fail, info, pad;
};
struct psd_omb {
    int n, phase, info;
    int j, js;  // scan cursor and count of selected eigenvalues seen (ordschur.jl:53-65)
    int scandone, nstarted, nfinished, nactive;
    int failseq, failinfo;
    int nticks;
};

PSD_D int psd_o_find(const psd_oslot* S, int seq) {
    for (int q = 0; q < PSD_O_SLOTS; ++q)
        if (S[q].active && S[q].seq == seq) return q;
    return -1;
}

PSD_KERNEL psd_ord1_plan(psd_ostate* sts, psd_oslot* S, psd_omb* Gp, const unsigned char* select) {
    PSD_ONE {
        psd_omb G = *Gp;
        if (G.phase != PSD_OPH_DONE) {
            const int n = G.n;
            for (int q = 0; q < PSD_O_SLOTS; ++q) {  // what the last tick did
                if (!S[q].active || S[q].fail || S[q].landed) continue;
                if (sts[q].phase == PSD_OPH_DONE) {
                    S[q].fail = 1;
                    S[q].info = sts[q].info;
                    if (S[q].seq < G.failseq) {
                        G.failseq = S[q].seq;
                        G.failinfo = S[q].info;
                    }
                } else if (sts[q].here <= S[q].target) {
                    S[q].landed = 1;
                }
            }
            for (;;) {  // arrivals in the order of selection
                const int q = psd_o_find(S, G.nfinished);
                if (q < 0 || !S[q].landed) break;
                S[q].active = 0;
                G.nfinished += 1;
                G.nactive -= 1;
            }
            if (G.failseq == 0x7fffffff && !G.scandone) {
                int f = -1;
                for (int q = 0; q < PSD_O_SLOTS && f < 0; ++q)
                    if (!S[q].active) f = q;
                while (f >= 0) {  // ordschur.jl:53-65
                    const int j = G.j + 1;
                    if (j > n) {
                        G.scandone = 1;
                        break;
                    }
                    if (!select[j - 1]) {
                        G.j = j;
                        continue;
                    }
                    if (j == G.js + 1) {  // already in place
                        G.j = j;
                        G.js += 1;
                        continue;
                    }
                    if (G.nactive > 0) {  // behind the eigenvalue started last, once a row lies between
                        const int r = psd_o_find(S, G.nstarted - 1);
                        if (r >= 0 && j < sts[r].here + 2) break;  // (row j is looked at again in the next tick)
                    }
                    G.j = j;
                    G.js += 1;
                    psd_oslot ns;
                    ns.active = 1;
                    ns.seq = G.nstarted;
                    ns.target = G.js;
                    ns.jsrc = j;
                    ns.landed = ns.fail = ns.info = ns.pad = 0;
                    S[f] = ns;
                    sts[f].here = j;
                    sts[f].info = 0;
                    G.nstarted += 1;
                    G.nactive += 1;
                    break;
                }
            }
            for (int q = 0; q < PSD_O_SLOTS; ++q) {  // the coming tick
                int ph = PSD_OPH_IDLE;
                if (S[q].active && !S[q].fail && !S[q].landed && S[q].seq < G.failseq) {
                    int lim = S[q].target;
                    const int r = psd_o_find(S, S[q].seq - 1);
                    if (r >= 0 && sts[r].here + 1 > lim) lim = sts[r].here + 1;
                    if (sts[q].here > lim) {
                        ph = PSD_OPH_MOVE;
                        sts[q].js = lim;
                        sts[q].j = S[q].jsrc;
                    }
                }
                sts[q].phase = ph;
            }
            if (G.failseq != 0x7fffffff) {
                bool any = false;
                for (int q = 0; q < PSD_O_SLOTS; ++q)
                    if (S[q].active && !S[q].fail && !S[q].landed && S[q].seq < G.failseq) any = true;
                if (!any) {
                    G.info = G.failinfo;
                    G.phase = PSD_OPH_DONE;
                }
            } else if (G.scandone && G.nactive == 0) {
                G.phase = PSD_OPH_DONE;
            }
            G.nticks += 1;
        }
        *Gp = G;
    }
}

PSD_KERNEL psd_ord1_init_mb(psd_ostate* sts, psd_oslot* S, psd_omb* Gp, int n, int p, int wantZ, int W) {
    PSD_PAR_FOR(q, PSD_O_SLOTS) {
        psd_ostate st;
        st.n = n; st.p = p; st.wantZ = wantZ; st.W = W;
        st.phase = PSD_OPH_IDLE; st.info = 0;
        st.j = 0; st.js = 0; st.here = 0; st.nswaps = 0; st.nwindows = 0;
        sts[q] = st;
        psd_oslot z;
        z.active = 0; z.seq = -1; z.target = z.jsrc = z.landed = z.fail = z.info = z.pad = 0;
        S[q] = z;
    }
    PSD_ONE {
        psd_omb G;
        G.n = n; G.phase = PSD_OPH_SCAN; G.info = 0;
        G.j = 0; G.js = 0;
        G.scandone = 0; G.nstarted = 0; G.nfinished = 0; G.nactive = 0;
        G.failseq = 0x7fffffff; G.failinfo = 0;
        G.nticks = 0;
        *Gp = G;
    }
}

// grid = PSD_O_SLOTS: slot s with its own state, descriptor, counts and lists (the layout of the trains' cursor arrays)
PSD_KERNEL_B(PSD_STEP_NT) psd_zord_step_mb(psd_oparams O, int p, int cstride) {
    const int s = PSD_BLOCK_X;
    O.st += s;
    O.z.desc += s;
    O.z.cnt += (size_t)s * cstride;
    O.z.tr += (size_t)s * p * PSD_ZTR_CAP;
    psd_zord_step_body(O);
}

PSD_KERNEL psd_zord_init(psd_oparams O, int n, int p, int wantZ, int W) {
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

// ordschur.jl:97-120 _updateλ!: eigenvalue j from the diagonals (T_1 first), scaled form.  grid over j.
PSD_KERNEL psd_zord_values(psd_zparams P, int n, int p) {
    const int NT = PSD_NTHREADS;
    PSD_PAR_FOR(t, NT) {
        const int j = 1 + PSD_BLOCK_X * NT + t;
        if (j <= n) {
            psd_z alpha = zmk(1.0, 0.0);
            int scale = 0;
            for (int l = 1; l <= p; ++l) {
                alpha = zmul(alpha, psd_zfac(P, n, l)(j, j));
                if (zabs(alpha) == 0) {
                    alpha = zmk(0.0, 0.0);
                    scale = 0;
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
            P.alpha[j - 1] = alpha;
            P.beta[j - 1] = 1.0;
            P.ascale[j - 1] = scale;
        }
    }
}
