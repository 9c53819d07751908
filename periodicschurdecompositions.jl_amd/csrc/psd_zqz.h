// Complex periodic QZ iteration on the GPU (all signatures +1): device-resident state machine.
//
// Replaces pschur!(H1, Hs, S; wantT, wantZ, Q, maxitfac) for ComplexF64 with S all true —
// /root/reference/src/generalized.jl:166-931 (SLICOT MB03BZ type single-shift periodic QZ with
// Givens rotations), which is what pschur!(A::Vector{Matrix{ComplexF64}}, lr) runs
// (PeriodicSchurDecompositions.jl:1106-1111, generalized.jl:108-137).
//
// Same MI355X structure as the real path (psd_real_qr.h): one wavefront chases a diagonal window of
// all p factors in LDS and emits per-factor rotation lists; a wide kernel applies them to the
// off-window rows/columns of H_m, H_{m-1} and Z_m.  Implemented state: deflation tests 1 and 2
// (:323-339), controlled zero shift (test 4, :356-448), Case II (a zero on the diagonal of a
// triangular factor: two unshifted half-passes, :453-566), 1x1 split with `_safeprod` (:741-762),
// single-shift sweep with the reference's shift chain (:770-852), final phase normalisation
// (:860-908).  Case III needs a negative signature (not in this build).
#pragma once
#include "psd_zhqr.h"
#include "psd_complex.h"
#include "psd_real_qr.h"

enum {
    PSD_ZPH_CHECK = 0,
    PSD_ZPH_SWEEP = 1,
    PSD_ZPH_ZSHIFT = 2,
    PSD_ZPH_CASE2A = 3,
    PSD_ZPH_CASE2B = 4,
    PSD_ZPH_DONE = 7,
    // multishift trains (as in psd_real_qr.h): leader waits for its cursors / cursor waits for its start / cursor done
    PSD_ZPH_TWAIT = 8,
    PSD_ZPH_CWAIT = 9,
    PSD_ZPH_CDONE = 10
};
#define PSD_ZTR_CAP 32  // rotations per owner and window

struct psd_ztr {  // Givens rotation on (pos, pos+1): c real, s complex
    int pos, pad;
    double c;
    psd_z s;
};

struct psd_zapply_desc {
    int active;
    int plo, phi;
    int lc0, lc1;   // left role: columns of H_m
    int rr0, rr1;   // right role: rows of H_{m-1}
    int rcut;       // first near row of the right role (rows rr0 .. rcut - 1 may run one tick late: psd_zparams::zcdefer); rr0: none are far (the last window of a sweep, every other kind of window)
    int zr0, zr1;   // Z role
    int defer_h1;   // 1: right-updates of H_1 are deferred (downward passes); 2: left-updates (upward pass)
    int defer_run;  // set with the last window of such a pass: run psd_zq_defer now (1 right, 2 left)
    int djlo, djhi, drow0;  // deferred list covers positions djlo..djhi; rows drow0.. / columns ..drow0
};

struct psd_zstate {
    int n, p, wantT, wantZ, W;
    int Wmax, train_oc;  // LDS layout width (W is the running sweep's, <= Wmax); o / c of the width rule (psd_rq_shift)
    int phase, info;
    int ilast, ifirst, ifirstm, ilastm, iiter, ziter, jiter, maxit;
    int jlo, kcur, zflag;
    int settle;  // (unused)
    int ldeflate, jdeflate, ncase2, pend2;
    int nsweeps, nzshift, nsplit, nwindows, nlog, maxlog;
    double c0;
    psd_z s0;
    double smlnum, ulp, safmin;
    long long cyc[6];
    // multishift train: bulges wanted / in the running train / train number / this state's cursor / tick of the leader's
    // first window / row whose diagonal entries are this bulge's shift / exceptional-shift bookkeeping / sweeps in trains
    int train_want, train_n, train_id, cursor, train_tick0, shidx, exc_dec, ntrainsweeps;
    int cstart, cfirst;  // cursor: the tick of its first window and that window's number of positions (cursors W positions apart)
    psd_z shift;  // this bulge's shift (an eigenvalue of the trailing block of the product)
};

struct psd_zparams {
    psd_z* H;
    psd_z* Z;
    psd_zstate* st;
    psd_zapply_desc* desc;
    psd_ztr* tr;   // [p][PSD_ZTR_CAP]
    int* cnt;      // [p]
    psd_ztr* dG;   // [n+2] deferred rotations of a zero-shift pass (indexed by position)
    psd_z* alpha;  // [n]
    double* beta;  // [n]
    int* ascale;   // [n]
    int* log;
    psd_zstate* cst;  // [PSD_TRAIN_MAX] cursor states of a train (entry 0 unused) or nullptr
    int* cep;  // [PSD_TRAIN_MAX] epoch words of the cursor states (psd_pub_*), then the count of finished cursors
    psd_z* tshift;    // [PSD_TRAIN_MAX + 1] shifts of the train, then a flag word
    int tick;         // launch index
    // period sharding (psd_set_shard): the owners m (1-based, inclusive) whose Schur vectors Z_m this context holds;
    // the updates of the others are some other rank's work (1..p without sharding)
    int zlo, zhi;
    // scan chase (psd_zchase3.h): byte offsets of the command block and of the rotation table in dynamic LDS; 0: off
    // (one wavefront per chase workgroup).  With it the chase workgroups have PSD_ZC3_WAVES wavefronts.
    int zcoff, zc3off;
    // factor-sliced sweep windows (psd_zslice3.h): slices per window (1: off), the slots' command blocks and inboxes in
    // device memory, the error word of the bounded waits
    int zslG;
    unsigned char* zslmem;
    int* zslerr;
    // 1: the far rows of the sweep windows' column roles (more than psd_cdefer_edge above the window) run on the second
    // stream beside the next tick's chases (as psd_rparams::cdefer of the real engine)
    int zcdefer;
};

PSD_HD psd_mat<psd_z> psd_zfac(const psd_zparams& P, int n, int j) {
    return psd_mat<psd_z>{P.H + (size_t)(j - 1) * n * n, n};
}

struct psd_zwin {
    psd_z* b;
    int W, ld, bsz, bs, be;
    PSD_HD psd_z& at(int j, int r, int c) const { return b[(j - 1) * bsz + (c - bs) * ld + (r - bs)]; }
};

#include "psd_zqz_win.inl"
#include "psd_zchase3.h"
#include "psd_zslice3.h"

// The helper wavefronts of a scan-chase workgroup (threadIdx.y >= 1) park at the command barrier between runs; code
// written in terms of PSD_TID / PSD_SYNC never sees them (as the two-wave chase of the real engine).
#ifndef PSD_HOSTSIM
PSD_D void psd_zc_helper(int zcoff, int zc3off) {
    PSD_LDS_DECL;
    const psd_zc* cmd = (const psd_zc*)(psd_lds + zcoff);
    for (;;) {
        PSD_PAIR_BARRIER();
        const psd_zc C = *cmd;
        if (C.cmd == 0) return;
        if (C.cmd == 2) {
            psd_zc3_run(C, PSD_WAVE_ROLE, (int)blockDim.y, zc3off);
        } else if (C.cmd == 5) {
            psd_zc3s_run(C, PSD_WAVE_ROLE, (int)blockDim.y, zc3off);
        } else {  // this wavefront's share of a window load (3) / store (4)
            psd_zparams R;
            R.H = C.H;
            psd_zwin w;
            w.b = (psd_z*)(psd_lds + C.wboff);
            w.W = C.W; w.ld = C.ld; w.bsz = C.bsz; w.bs = C.bs; w.be = C.be;
            if (C.cmd == 3) psd_zwin_load(R, w, C.n, C.p, PSD_WAVE_ROLE, (int)blockDim.y);
            else psd_zwin_store(R, w, C.n, C.p, PSD_WAVE_ROLE, (int)blockDim.y);
            PSD_PAIR_BARRIER();
        }
    }
}
#define PSD_ZC_ENTER(P)                                 \
    if ((P).zcoff != 0 && PSD_WAVE_ROLE >= 1) {         \
        psd_zc_helper((P).zcoff, (P).zc3off);           \
        return;                                         \
    }
#define PSD_ZC_LEAVE(P)                                         \
    if ((P).zcoff != 0) {                                       \
        PSD_LDS_DECL;                                           \
        psd_zc* cmd__ = (psd_zc*)(psd_lds + (P).zcoff);         \
        PSD_ONE { cmd__->cmd = 0; }                             \
        PSD_PAIR_BARRIER();                                     \
    }
#else
#define PSD_ZC_ENTER(P) ((void)0)
#define PSD_ZC_LEAVE(P) ((void)0)
#endif
// wavefront 0's side of a run / of a shared window transfer
PSD_D void psd_zc3_lead(const psd_zparams& P, psd_zc& C) {
#ifdef PSD_HOSTSIM
    psd_zc3_run(C, 0, 1, P.zc3off);
#else
    PSD_LDS_DECL;
    psd_zc* cmd = (psd_zc*)(psd_lds + P.zcoff);
    C.cmd = 2;
    PSD_SYNC();
    PSD_ONE { *cmd = C; }
    PSD_PAIR_BARRIER();
    psd_zc3_run(C, 0, (int)blockDim.y, P.zc3off);
    PSD_ONE { cmd->cmd = 0; }
#endif
}
PSD_D void psd_zc3_winio(const psd_zparams& P, const psd_zwin& w, int n, int p, bool store) {
#ifndef PSD_HOSTSIM
    if (P.zcoff != 0 && blockDim.y > 1) {
        PSD_LDS_DECL;
        psd_zc* cmd = (psd_zc*)(psd_lds + P.zcoff);
        PSD_SYNC();
        PSD_ONE {
            psd_zc C;
            C.cmd = store ? 4 : 3;
            C.ld = w.ld; C.bsz = w.bsz; C.bs = w.bs; C.be = w.be; C.W = w.W;
            C.p = p; C.n = n; C.H = P.H;
            C.wboff = (int)((char*)w.b - (char*)psd_lds);
            *cmd = C;
        }
        PSD_PAIR_BARRIER();
        if (store) psd_zwin_store(P, w, n, p, 0, (int)blockDim.y);
        else psd_zwin_load(P, w, n, p, 0, (int)blockDim.y);
        PSD_PAIR_BARRIER();
        PSD_ONE { cmd->cmd = 0; }
        return;
    }
#endif
    if (store) psd_zwin_store(P, w, n, p);
    else psd_zwin_load(P, w, n, p);
}

#ifndef PSD_HOSTSIM
// ---- factor-sliced sweep windows (psd_zslice3.h): slice 0's command to the slot's workers, the workers themselves
PSD_D void psd_zsl_publish(const psd_zparams& P, int slot, const psd_zc* C) {  // (one lane)
    unsigned char* cm = psd_zsl_cmd(P.zslmem, slot);
    if (C != nullptr) {
        *(psd_zc*)(cm + 64) = *C;
        psd_release_fence();
    }
    __hip_atomic_store((unsigned long long*)cm, psd_sl_tag(P.tick, 0, (C != nullptr) ? 1 : 4), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
PSD_D void psd_zsl_finish(const psd_zparams& P, int slot) {  // slice 0, at the end of its launch
    PSD_ONE {
        const unsigned long long v = __hip_atomic_load((unsigned long long*)psd_zsl_cmd(P.zslmem, slot), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (v != psd_sl_tag(P.tick, 0, 1)) psd_zsl_publish(P, slot, nullptr);
    }
}
PSD_D void psd_zsl_worker(const psd_zparams& P, int slot, int g) {  // every wavefront of a worker workgroup
    PSD_LDS_DECL;
    psd_zc* cmd = (psd_zc*)(psd_lds + P.zcoff);
    const int wv = PSD_WAVE_ROLE, nw = (int)blockDim.y;
    if (wv == 0) {
        unsigned char* cm = psd_zsl_cmd(P.zslmem, slot);
        const unsigned long long trun = psd_sl_tag(P.tick, 0, 1), tidle = psd_sl_tag(P.tick, 0, 4);
        int kind = 4, spins = 0;
        long long t0 = 0;
        for (;;) {
            const unsigned long long v = __hip_atomic_load((unsigned long long*)cm, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (v == trun) {
                kind = 1;
                break;
            }
            if (v == tidle) break;
            if ((++spins & 255) == 0) {
                if (__hip_atomic_load(P.zslerr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0) break;
                const long long now = (long long)__builtin_amdgcn_s_memrealtime();
                if (t0 == 0) t0 = now;
                else if (now - t0 > PSD_SL_WAIT_TICKS) {
                    __hip_atomic_store(P.zslerr, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    break;
                }
            }
            __builtin_amdgcn_s_sleep(2);
        }
        if (kind == 1) {
            psd_acquire_fence();
            psd_zc C = *(const psd_zc*)(cm + 64);
            C.cmd = 5;
            C.slg = g;
            C.wboff = 0;
            PSD_ONE { *cmd = C; }
        } else {
            PSD_ONE { cmd->cmd = 0; }
        }
    }
    PSD_PAIR_BARRIER();
    const psd_zc C = *cmd;
    if (C.cmd != 5) return;
    int jlo, jhi;
    psd_sl_range(C.p, C.slG, g, jlo, jhi);
    psd_zparams R;
    R.H = C.H + (size_t)(jlo - 1) * C.n * C.n;
    psd_zwin w;
    w.b = (psd_z*)psd_lds;
    w.W = C.W; w.ld = C.ld; w.bsz = C.bsz; w.bs = C.bs; w.be = C.be;
    psd_zwin_load(R, w, C.n, jhi - jlo + 1, wv, nw);
    PSD_PAIR_BARRIER();
    psd_zc3s_run(C, wv, nw, P.zc3off);
    psd_zwin_store(R, w, C.n, jhi - jlo + 1, wv, nw);
}
PSD_D void psd_zc3s_lead(const psd_zparams& P, psd_zc& C, int slot) {
    PSD_LDS_DECL;
    psd_zc* cmd = (psd_zc*)(psd_lds + P.zcoff);
    C.cmd = 5;
    C.slG = P.zslG;
    C.slg = 0;
    C.sltick = P.tick;
    C.slbox = psd_zsl_cmd(P.zslmem, slot) + PSD_SL_CMD_BYTES;
    C.slerr = P.zslerr;
    PSD_SYNC();
    PSD_ONE {
        psd_zsl_publish(P, slot, &C);
        *cmd = C;
    }
    PSD_PAIR_BARRIER();
    psd_zc3s_run(C, 0, (int)blockDim.y, P.zc3off);
    PSD_ONE { cmd->cmd = 0; }
}
#endif

PSD_D void psd_zrecord(const psd_zparams& P, int* lcnt, int m, int pos, double c, psd_z s) {
    PSD_ONE {
        const int q = lcnt[m - 1];
        if (q < PSD_ZTR_CAP) {
            psd_ztr tr;
            tr.pos = pos;
            tr.pad = 0;
            tr.c = c;
            tr.s = s;
            P.tr[(size_t)(m - 1) * PSD_ZTR_CAP + q] = tr;
        }
        lcnt[m - 1] = q + 1;
    }
    PSD_WAVE_SYNC();
}

PSD_D void psd_zlog(const psd_zparams& P, psd_zstate& st, int kind, int lo, int hi) {
    PSD_ONE {
        if (st.nlog < st.maxlog) {
            P.log[3 * st.nlog + 0] = kind;
            P.log[3 * st.nlog + 1] = lo;
            P.log[3 * st.nlog + 2] = hi;
        }
    }
    st.nlog += 1;
}

PSD_D void psd_zdesc_write(const psd_zparams& P, psd_zstate& st, const int* lcnt, int plo, int phi, int lc0,
                           int lc1, int rr0, int rr1, int defer_h1, int defer_run, int djlo, int djhi, bool nofar = true) {
    PSD_SYNC();
    const bool over = psd_list_overflow(lcnt, st.p, PSD_ZTR_CAP);
    if (over) {  // never apply truncated lists
        st.info = PSD_LIST_OVERFLOW;
        st.phase = PSD_ZPH_DONE;
    }
    PSD_PAR_FOR(m, st.p) { P.cnt[m] = lcnt[m]; }
    PSD_ONE {
        psd_zapply_desc d;
        d.active = over ? 0 : 1;
        d.plo = plo;
        d.phi = phi;
        d.lc0 = lc0;
        d.lc1 = lc1;
        d.rr0 = rr0;
        d.rr1 = rr1;
        {   // Plain sweep windows only, and not the last window of a sweep: the passes that defer a side of H_1 keep
            // their column roles whole, and the check behind a sweep's (a train's) last windows — it may read or rewrite
            // whole rows and columns: Case II, the norm of H_1 — runs in the very next launch, with nothing under way.
            int rc = rr0;
            if (P.zcdefer && !nofar && defer_h1 == 0 && defer_run == 0) {
                rc = plo - psd_cdefer_edge(st.Wmax);
                if (rc < rr0) rc = rr0;
                if (rc > rr1 + 1) rc = rr1 + 1;
            }
            d.rcut = rc;
        }
        d.zr0 = 1;
        d.zr1 = st.wantZ ? st.n : 0;
        d.defer_h1 = defer_h1;
        d.defer_run = defer_run;
        d.djlo = djlo;
        d.djhi = djhi;
        d.drow0 = st.ifirstm;
        *P.desc = d;
    }
    PSD_SYNC();
}

// opnorm(view(M, lo:hi, lo:hi), 1) from global memory (rare fallback when a tolerance base is zero)
PSD_D double psd_zopnorm(const psd_mat<psd_z>& M, double* red, int lo, int hi, bool upper) {
    const int NT = PSD_NTHREADS;
    PSD_SYNC();
    PSD_PAR_FOR(t, NT) {
        double best = 0.0;
        for (int c = lo + t; c <= hi; c += NT) {
            double s = 0.0;
            const int rmax = upper ? c : ((c + 1 < hi) ? (c + 1) : hi);
            for (int r = lo; r <= rmax; ++r) s += zabs(M(r, c));
            if (s > best) best = s;
        }
        red[t] = best;
    }
    PSD_SYNC();
    double best = 0.0;
    for (int t = 0; t < NT; ++t)
        if (red[t] > best) best = red[t];
    PSD_SYNC();
    return best;
}

// generalized.jl:939-976 with all signatures true: prod x_l = alpha * 2^scale, |alpha| in [1,2)
PSD_D void psd_zsafeprod(const psd_zparams& P, const psd_zstate& st, int idx, psd_z& alpha, double& beta, int& scale) {
    alpha = zmk(1.0, 0.0);
    beta = 1.0;
    scale = 0;
    for (int l = 1; l <= st.p; ++l) {
        const psd_z xi = psd_zfac(P, st.n, l)(idx, idx);
        alpha = zmul(alpha, xi);
        const double a = zabs(alpha);
        if (a == 0) {
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
}

PSD_D void psd_zq_start_case2(const psd_zparams& P, psd_zstate& st) {
    st.pend2 = 0;
    st.ncase2 += 1;
    psd_zlog(P, st, 2, st.jlo, st.ilast);
    if (st.jdeflate > st.jlo) {
        st.phase = PSD_ZPH_CASE2A;
        st.kcur = st.jlo;
    } else if (st.jdeflate < st.ilast) {
        st.phase = PSD_ZPH_CASE2B;
        st.kcur = st.ilast;
    } else {
        st.phase = PSD_ZPH_CHECK;
    }
}

// generalized.jl:302-449,741-806: deflation tests, split, zero-shift decision, shift chain
// Starting rotation of a single-shift sweep on the active block ifirst.. whose shift is the product of the diagonal
// entries in row `sh` of the factors (generalized.jl:786-805 with sh = ilast), as a chain of rotations over the factors.
PSD_D void psd_zq_start_rot(const psd_zparams& P, int n, int p, int ifirst, int sh, double& c, psd_z& s) {
    const psd_mat<psd_z> H1 = psd_zfac(P, n, 1);
    psd_z r;
    psd_zgivens(zmk(1.0, 0.0), zmk(1.0, 0.0), c, s, r);
    for (int l = p; l >= 2; --l) {
        const psd_mat<psd_z> Hl = psd_zfac(P, n, l);
        psd_zgivens(zscal(c, Hl(ifirst, ifirst)), zmul(Hl(sh, sh), zconj(s)), c, s, r);
    }
    psd_zgivens(zsub(zscal(c, H1(ifirst, ifirst)), zmul(H1(sh, sh), zconj(s))), zscal(c, H1(ifirst + 1, ifirst)), c, s,
                r);
}

// Starting rotation for an explicit shift mu: first column of (P - mu I) on the active block is
// D (H_1[f,f] - mu / D, H_1[f+1,f]) with D = prod_{l>=2} H_l[f,f]; mu / D by successive divisions (it stays bounded when
// mu is of the size of the product's entries).  false: not finite (the caller falls back to the reference's shift).
PSD_D bool psd_zq_start_rot_mu(const psd_zparams& P, int n, int p, int ifirst, psd_z mu, double& c, psd_z& s) {
    psd_z t = mu;
    for (int l = p; l >= 2; --l) {
        const psd_mat<psd_z> Hl = psd_zfac(P, n, l);
        t = zdiv(t, Hl(ifirst, ifirst));
    }
    if (!(zabs1(t) < 1e300)) return false;  // (also false for NaN)
    const psd_mat<psd_z> H1 = psd_zfac(P, n, 1);
    psd_z r;
    psd_zgivens(zsub(H1(ifirst, ifirst), t), H1(ifirst + 1, ifirst), c, s, r);
    return true;
}

// The m shifts of a train: eigenvalues of the trailing m x m block of H_1 H_2 ... H_p (see psd_rq_train_shifts: blocks
// staged in LDS by the whole wavefront, products one entry per lane, the small QR iteration by one lane).  `work`: LDS,
// psd_zq_train_elems(p, m) complex elements.  Shifts go to P.tshift ordered by distance from the last diagonal entry of
// the block (closest first: the leader's); *okf = 1 on success.
PSD_HD size_t psd_zq_train_elems(int p, int m) {
    const size_t K1 = (size_t)m + 1;
    return (size_t)p * K1 * K1 + 2 * K1 * K1 + (size_t)m * m + PSD_ZHQR_MAX + 8;
}
PSD_D void psd_zq_train_shifts(const psd_zparams& P, int n, int p, int ilast, int m, psd_z* work, int* okf) {
    const int K = m, K1 = K + 1, t0 = ilast - K + 1, KK = K1 * K1;
    psd_z* B = work;
    psd_z* R0 = B + (size_t)p * KK;
    psd_z* R1 = R0 + KK;
    psd_z* T = R1 + KK;
    psd_z* w = T + K * K;
    PSD_SYNC();
    PSD_PAR_FOR(t, p * KK) {
        const int j = t / KK, q = t - j * KK, r = q / K1, c = q - r * K1;
        B[t] = psd_zfac(P, n, j + 1)(t0 - 1 + r, t0 - 1 + c);
    }
    PSD_PAR_FOR(q, KK) { R0[q] = (q / K1 == q % K1) ? zmk(1.0, 0.0) : zmk(0.0, 0.0); }
    PSD_SYNC();
    psd_z* cur = R0;
    psd_z* nxt = R1;
    for (int j = 2; j <= p; ++j) {
        const psd_z* Bj = B + (size_t)(j - 1) * KK;
        PSD_PAR_FOR(q, KK) {
            const int r = q / K1, c = q - r * K1;
            psd_z acc = zmk(0.0, 0.0);
            for (int k = r; k <= c; ++k) acc = zadd(acc, zmul(cur[r * K1 + k], Bj[k * K1 + c]));
            nxt[q] = acc;
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
    bool okw = false;
    if (wave1) okw = psd_zhqr_wave(T, K, K, w, PSD_TID);
#endif
    PSD_ONE {
#ifndef PSD_HOSTSIM
        const psd_z last = lastw;
        bool ok = wave1 ? okw : psd_zhqr(T, K, K, w);
#else
        const psd_z last = T[(K - 1) * K + (K - 1)];
        bool ok = psd_zhqr(T, K, K, w);
#endif
        for (int a = 0; ok && a < K; ++a)
            if (!(zabs1(w[a]) < 1e300)) ok = false;
        if (ok) {
            for (int a = 1; a < K; ++a) {  // insertion sort by distance from the last diagonal entry
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

PSD_D void psd_zq_check(const psd_zparams& P, psd_zstate& st, double* red, int* redi, psd_z* work) {
    const int n = st.n, p = st.p;
    const int NT = PSD_NTHREADS;
    st.jiter += 1;
    if (st.jiter > st.maxit) {  // generalized.jl:856-858
        st.info = st.ilast;
        st.phase = PSD_ZPH_DONE;
        return;
    }
    const psd_mat<psd_z> H1 = psd_zfac(P, n, 1);
    const int ilast = st.ilast;
    bool split = false;
    int jlo = 1;
    if (ilast == 1) {
        split = true;
    } else {
        // Test 1 (:260-278,324): first negligible subdiagonal of H_1 from the bottom
        double h1norm = -1.0;
        int jfound = 0;
        for (int pass = 0; pass < 2; ++pass) {
            PSD_PAR_FOR(t, NT) {
                int best = 0, need = 0;
                for (int j = ilast - t; j >= 2; j -= NT) {
                    double tol = zabs(H1(j - 1, j - 1)) + zabs(H1(j, j));
                    if (tol == 0) {
                        if (h1norm < 0) {
                            need = 1;
                            continue;
                        }
                        tol = h1norm;  // opnorm over 1:j approximated by the window 1:ilast (doubly degenerate case)
                    }
                    tol = fmax(st.ulp * tol, st.smlnum);
                    if (zabs(H1(j, j - 1)) <= tol) {
                        best = j;
                        break;
                    }
                }
                redi[t] = best;
                redi[NT + t] = need;
            }
            PSD_SYNC();
            int need = 0;
            jfound = 0;
            for (int t = 0; t < NT; ++t) {
                if (redi[t] > jfound) jfound = redi[t];
                need |= redi[NT + t];
            }
            PSD_SYNC();
            if (!need) break;
            h1norm = psd_zopnorm(H1, red, 1, ilast, false);
        }
        if (jfound > 0) {
            PSD_ONE { H1(jfound, jfound - 1) = zmk(0.0, 0.0); }
            PSD_SYNC();
            jlo = jfound;
            if (jfound == ilast) split = true;
        }
    }
    if (split) {  // :741-762
        psd_z alpha;
        double beta;
        int scale;
        psd_zsafeprod(P, st, ilast, alpha, beta, scale);
        PSD_ONE {
            P.alpha[ilast - 1] = alpha;
            P.beta[ilast - 1] = beta;
            P.ascale[ilast - 1] = scale;
        }
        st.nsplit += 1;
        st.ilast -= 1;
        if (st.ilast < 1) {
            st.phase = PSD_ZPH_DONE;
            return;
        }
        st.iiter = 0;
        st.exc_dec = 0;
        if (st.ziter != -1) st.ziter = 0;
        if (!st.wantT) {
            st.ilastm = st.ilast;
            if (st.ifirstm > st.ilast) st.ifirstm = 1;
        }
        return;  // next jiter
    }
    // Test 2 (:280-299,328-339): zero on the diagonal of a triangular factor -> Case II.
    // First factor l (ascending) with a hit, and for it the largest j.
    st.jlo = jlo;
    {
        const int wd = ilast - jlo + 1;
        PSD_PAR_FOR(t, NT) {
            int key = 0x7fffffff;
            for (int q = t; q < (p - 1) * wd; q += NT) {
                const int l = 2 + q / wd, j = jlo + q % wd;
                const psd_mat<psd_z> Hl = psd_zfac(P, n, l);
                double tol;
                if (j == ilast) tol = zabs(Hl(j - 1, j));
                else if (j == jlo) tol = zabs(Hl(j, j + 1));
                else tol = zabs(Hl(j - 1, j)) + zabs(Hl(j, j + 1));
                tol = fmax(st.ulp * tol, st.smlnum);  // (tol == 0 fallback of the reference: smlnum floor)
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
        if (key != 0x7fffffff) {  // Case II (:453-566)
            const int l = key / (n + 2), j = (n + 1) - key % (n + 2);
            const psd_mat<psd_z> Hl = psd_zfac(P, n, l);
            PSD_ONE { Hl(j, j) = zmk(0.0, 0.0); }
            PSD_SYNC();
            st.ldeflate = l;
            st.jdeflate = j;
            st.pend2 = 1;
        }
    }
    // Test 4 (:356): controlled zero shift.  The reference would still run Case II afterwards with the
    // now stale position (:446-453) and produce an invalid decomposition; the zero is re-detected in
    // the next iteration instead.
    if (st.ziter >= 7 || st.ziter < 0) {
        st.pend2 = 0;
        st.phase = PSD_ZPH_ZSHIFT;
        st.kcur = jlo;
        st.zflag = 0;
        st.nzshift += 1;
        psd_zlog(P, st, 4, jlo, ilast);
        return;
    }
    if (st.pend2) {
        psd_zq_start_case2(P, st);
        return;
    }
    // QZ step (:763-806)
    st.ifirst = jlo;
    st.iiter += 1;
    st.ziter += 1;
    if (!st.wantT) st.ifirstm = st.ifirst;
    double c;
    psd_z s, r;
    st.train_n = 1;
    const int dec = st.iiter / 10;  // (iiter % 10 == 0 in the reference; a train advances iiter by several)
    if (dec > st.exc_dec) {
        st.exc_dec = dec;
        // exceptional shift: the reference draws rand(T, 2) (:782); fixed pair for determinism
        psd_zgivens(zmk(0.35, 0.62), zmk(0.81, 0.27), c, s, r);
    } else {
        psd_zq_start_rot(P, n, p, st.ifirst, ilast, c, s);
        // multishift train: bulge b takes the diagonal entries of row ilast - b of the factors as its shift (the
        // reference's shift is b = 0, :786-805); cursors two windows apart (see psd_rq_step_train)
        if (st.train_want >= 2 && P.cst != nullptr) {
            // window width of the train by the cost model of psd_rq_shift
            const int w = ilast - st.ifirst + 1;
            int mt = st.train_want;
            if (mt > PSD_TRAIN_MAX) mt = PSD_TRAIN_MAX;
            int nb = st.Wmax - 3, m = 1;
            double best = 1e300;
            for (int nbc = (st.Wmax - 3 < 8) ? ((st.Wmax > 4) ? st.Wmax - 3 : 1) : 8; nbc <= st.Wmax - 3; ++nbc) {
                // (cursors nbc + 3 positions apart — their windows only have to be disjoint — as in the real engine since round 2;
                //  two whole windows apart before: a train of m bulges fills and drains in ceil((m-1)(nbc+3)/nbc) ticks, not 2(m-1))
                int mc = 1 + (w - nbc) / (nbc + 3);
                if (mc > mt) mc = mt;
                if (mc < 2) break;
                const double cost = (double)((w + nbc - 1) / nbc + ((mc - 1) * (nbc + 3) + nbc - 1) / nbc) * (double)(nbc * p + st.train_oc) / mc;
                if (cost < best) {
                    best = cost;
                    nb = nbc;
                    m = mc;
                }
            }
            // (a train longer than the block the small QR and its LDS staging take runs through the shifts again)
            int ms = (m > PSD_ZHQR_MAX) ? PSD_ZHQR_MAX : m;
            while (ms >= 2 && psd_zq_train_elems(p, ms) > (size_t)p * st.Wmax * (st.Wmax + 1)) --ms;  // LDS of the staging
            if (ms < 2) m = 1;
            if (m >= 2 && 2 * m + 2 <= w) {
                int* okf = (int*)(P.tshift + PSD_TRAIN_MAX);  // (flag word behind the shifts)
                psd_zq_train_shifts(P, n, p, ilast, ms, work, okf);
                if (*okf && m > ms) {
                    PSD_ONE {
                        for (int b = ms; b < m; ++b) P.tshift[b] = P.tshift[b - ms];
                    }
                    PSD_SYNC();
                }
                double cm;
                psd_z sm;
                if (*okf && psd_zq_start_rot_mu(P, n, p, st.ifirst, P.tshift[0], cm, sm)) {
                    st.W = nb + 3;
                    st.train_n = m;
                    st.train_tick0 = P.tick;
                    st.train_id += 1;
                    st.ntrainsweeps += m;
                    c = cm;
                    s = sm;
                }
                PSD_SYNC();
            }
        }
    }
    st.c0 = c;
    st.s0 = s;
    st.phase = PSD_ZPH_SWEEP;
    st.kcur = st.ifirst;
    st.nsweeps += 1;
    psd_zlog(P, st, 0, st.ifirst, ilast);
    if (st.train_n > 1) {
        for (int b = 1; b < st.train_n; ++b) psd_zlog(P, st, 0, st.ifirst, ilast);  // one log entry per bulge
        PSD_SYNC();
        PSD_ONE {
            psd_atomic_store(P.cep + PSD_TRAIN_MAX, 0);  // finished cursors of this train
            for (int b = 1; b < st.train_n; ++b) {
                psd_zstate cs = st;
                cs.cursor = b;
                {  // bulge b runs b W positions behind the leader: first tick in which its window reaches into the block, and the part inside
                    const int nbw = st.W - 3, spc = st.W;
                    const int d = (b * spc - nbw + 1 + nbw - 1) / nbw;  // ceil((b W - nb + 1) / nb) >= 1
                    cs.cstart = st.train_tick0 + d;
                    cs.cfirst = d * nbw - b * spc + nbw;                 // 1 .. nb positions
                }
                cs.phase = PSD_ZPH_CWAIT;
                cs.shidx = ilast;
                cs.shift = P.tshift[b];
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

// Hot chain of the complex sweep (generalized.jl:823-845, S[l] true) at position j, factors l = p..2.  Per factor:
// incoming rotation G' from the right on columns (j, j+1) of H_l (lanes = rows r0..j+1), new rotation
// from (H_l[j,j], H_l[j+1,j]), applied from the left to rows (j, j+1) (lanes = columns j+1..c1max).
// One pass per factor: operands loaded once, the chain values (f, g and the 2 corner entries of column j+1)
// travel by v_readlane.  A step touches H_l only, so the operands of factor l - 1 are requested BEFORE the step of
// factor l runs (their LDS round trip, 120-130 cycles, is off the chain), and nothing waits for a step's stores: one
// wavefront's DS operations complete in order, and what a step wrote is read again a whole lap later.
// (c, s) in: rotation from H_1's rows; out: the one that goes to H_1's columns.
PSD_D void psd_zq_chain(const psd_zparams& P, const psd_zwin& w, int p, int j, int nr, int nl,
                        PSD_LANEVAR_REF(int, lane_off), PSD_LANEVAR_REF(int, lane_str), double& c, psd_z& s,
                        int slot) {
    const int cnt = nr + nl;
    PSD_LANEVAR(psd_z, x1);
    PSD_LANEVAR(psd_z, x2);
    PSD_LANEVAR(psd_z, y1);
    PSD_LANEVAR(psd_z, y2);
    PSD_PAR_ONCE(t, cnt) {
        const psd_z* q = w.b + ((p - 1) * w.bsz + PSD_LV(lane_off));
        PSD_LV(x1) = q[0];
        PSD_LV(x2) = q[PSD_LV(lane_str)];
    }
    for (int l = p; l >= 2; --l) {
        const int boff = (l - 1) * w.bsz;
        PSD_PAR_ONCE(t, cnt) {
            if (t < nr) psd_zrot_right_adj(c, s, PSD_LV(x1), PSD_LV(x2));
        }
        // rows j and j+1 are the last two row-lanes
        const psd_z f = PSD_BCASTZ(x1, nr - 2), g = PSD_BCASTZ(x1, nr - 1);
        const psd_z top = PSD_BCASTZ(x2, nr - 2), bot = PSD_BCASTZ(x2, nr - 1);
        if (l > 2) {  // (requested here, behind the first use of this step's operands: the wait in front of that use then covers the previous step's stores only, and the round trip runs beside the rotation's arithmetic)
            PSD_PAR_ONCE(t, cnt) {
                const psd_z* q = w.b + (boff - w.bsz + PSD_LV(lane_off));
                PSD_LV(y1) = q[0];
                PSD_LV(y2) = q[PSD_LV(lane_str)];
            }
        }
        psd_z r;
        psd_zgivens_lean(f, g, c, s, r);
        PSD_PAR_ONCE(t, cnt) {
            psd_z* q = w.b + (boff + PSD_LV(lane_off));
            if (t < nr) {
                if (t >= nr - 2) {  // rows j, j+1: column j becomes (r, 0); their column j+1 belongs to the left pass
                    q[0] = (t == nr - 2) ? r : zmk(0.0, 0.0);
                } else {
                    q[0] = PSD_LV(x1);
                    q[PSD_LV(lane_str)] = PSD_LV(x2);
                }
            } else {
                psd_z a1 = PSD_LV(x1), a2 = PSD_LV(x2);
                if (t == nr) {  // column j+1: entries already touched by the right pass
                    a1 = top;
                    a2 = bot;
                }
                psd_zrot_left(c, s, a1, a2);
                q[0] = a1;
                q[PSD_LV(lane_str)] = a2;
            }
            if (t == 0) {
                psd_ztr tr;
                tr.pos = j;
                tr.pad = 0;
                tr.c = c;
                tr.s = s;
                if (slot < PSD_ZTR_CAP) P.tr[(size_t)(l - 1) * PSD_ZTR_CAP + slot] = tr;
            }
            PSD_LV(x1) = PSD_LV(y1);
            PSD_LV(x2) = PSD_LV(y2);
        }
    }
    PSD_WAVE_SYNC();
}

// generalized.jl:808-852: one window of the single-shift sweep (positions kcur..kcur+nb-1)
PSD_D void psd_zq_sweep_window(const psd_zparams& P, psd_zstate& st, psd_z* ldsz, int* lcnt) {
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
    const long long tc0 = psd_clock();
    PSD_PAR_FOR(m, p) { lcnt[m] = 0; }
    const bool scan3 = P.zc3off != 0 && p >= PSD_ZC3_MINP && p <= PSD_ZC3_MAXP && ke >= ks;
    // factor-sliced window (psd_zslice3.h): this workgroup is slice 0 with the blocks of its own factors 1..p0
    int p0 = p;
#ifndef PSD_HOSTSIM
    const bool sliced = scan3 && P.zslG > 1;
    if (sliced) {
        int jlo0;
        psd_sl_range(p, P.zslG, 0, jlo0, p0);
    }
#else
    const bool sliced = false;
    (void)sliced;
#endif
    psd_zc3_winio(P, w, n, p0, false);
    const long long tc1 = psd_clock();
    if (scan3) {  // every factor of a position at once (psd_zchase3.h)
        psd_zc C;
        C.ld = w.ld; C.bsz = w.bsz; C.bs = w.bs; C.be = w.be; C.W = w.W;
        C.p = p; C.n = n; C.ifirst = ifirst; C.ifirstm = ifirstm; C.ilast = ilast; C.ilastm = ilastm;
        C.ks = ks; C.npos = ke - ks + 1;
        {
            PSD_LDS_DECL;
            C.wboff = (int)((char*)w.b - (char*)psd_lds);
        }
        C.tr = P.tr;
        C.H = P.H;
        C.c0 = st.c0;
        C.s0 = st.s0;
#ifndef PSD_HOSTSIM
        if (sliced) psd_zc3s_lead(P, C, (st.cursor > 0) ? st.cursor : 0);
        else
#endif
        psd_zc3_lead(P, C);
        PSD_SYNC();
        PSD_ONE { lcnt[0] = ke - ks + 1; }
        PSD_SYNC();
    }
    for (int j = scan3 ? (ke + 1) : ks; j <= ke; ++j) {
        double c;
        psd_z s, r;
        if (j > ifirst) {
            psd_zgivens(w.at(1, j, j - 1), w.at(1, j + 1, j - 1), c, s, r);
            PSD_WAVE_SYNC();
            PSD_ONE {
                w.at(1, j, j - 1) = r;
                w.at(1, j + 1, j - 1) = zmk(0.0, 0.0);
            }
            PSD_WAVE_SYNC();
        } else {
            c = st.c0;
            s = st.s0;
        }
        psd_zwin_left(w, 1, j, c, s, j, ilastm);
        psd_zrecord(P, lcnt, 1, j, c, s);
        if (p >= 2) {
            // lane roles for the factors l = p..2 at this position: [0,nr) rows r0..j+1 (columns j, j+1),
            // [nr,nr+nl) columns j+1..c1max (rows j, j+1)
            const int r0 = (w.bs > ifirstm) ? w.bs : ifirstm;
            const int c1max = (w.be < ilastm) ? w.be : ilastm;
            const int nr = j + 1 - r0 + 1, nl = c1max - j;
            PSD_LANEVAR(int, lane_off);
            PSD_LANEVAR(int, lane_str);
            PSD_PAR_ONCE(t, nr + nl) {
                if (t < nr) {
                    PSD_LV(lane_off) = (j - w.bs) * w.ld + (r0 + t - w.bs);
                    PSD_LV(lane_str) = w.ld;
                } else {
                    PSD_LV(lane_off) = (j + 1 + (t - nr) - w.bs) * w.ld + (j - w.bs);
                    PSD_LV(lane_str) = 1;
                }
            }
            const int slot = j - ks;
            psd_zq_chain(P, w, p, j, nr, nl, lane_off, lane_str, c, s, slot);
        }
        const int itmp = (j + 2 < ilastm) ? (j + 2) : ilastm;
        psd_zwin_right(w, 1, j, c, s, ifirstm, itmp);
    }
    if (p >= 2) {
        PSD_WAVE_SYNC();
        PSD_PAR_FOR(m, p) {
            if (m >= 1) lcnt[m] = ke - ks + 1;
        }
        PSD_WAVE_SYNC();
    }
    const long long tc2 = psd_clock();
    psd_zc3_winio(P, w, n, p0, true);  // (a sliced window: this slice's blocks; the workers store theirs)
    st.cyc[1] += tc1 - tc0;
    st.cyc[2] += tc2 - tc1;
    st.cyc[3] += psd_clock() - tc2;
    psd_zdesc_write(P, st, lcnt, ks, ke + 1, w.be + 1, ilastm, ifirstm, w.bs - 1, 0, 0, 0, 0, ke >= ilast - 1);
    st.nwindows += 1;
    st.kcur = ke + 1;
    if (ke >= ilast - 1)
        st.phase = (st.cursor > 0) ? PSD_ZPH_CDONE : ((st.train_n > 1) ? PSD_ZPH_TWAIT : PSD_ZPH_CHECK);
}

// One window of a downward unshifted pass (positions kcur..): the controlled zero shift
// (generalized.jl:356-448, `plain` = false, positions jlo..ilast-1, with the in-pass deflation
// test) or the first half of Case II (:460-510, `plain` = true, positions jlo..jdeflate-1; the
// rotations past the zero diagonal entry are exact identities, so the full chain is run).
PSD_D void psd_zq_zshift_window(const psd_zparams& P, psd_zstate& st, psd_z* ldsz, int* lcnt, bool plain) {
    const int n = st.n, p = st.p, jlo = st.jlo, ilast = st.ilast, ifirstm = st.ifirstm, ilastm = st.ilastm;
    const int nb = st.W - 2;
    const int ks = st.kcur;
    const int jend = plain ? (st.jdeflate - 1) : (ilast - 1);
    const int ke = (ks + nb - 1 < jend) ? (ks + nb - 1) : jend;
    psd_zwin w;
    w.b = ldsz;
    w.W = st.W;
    w.ld = st.W + 1;
    w.bsz = st.W * (st.W + 1);
    w.bs = ks;
    w.be = ke + 1;
    PSD_PAR_FOR(m, p) { lcnt[m] = 0; }
    psd_zwin_load(P, w, n, p);
    for (int j = ks; j <= ke; ++j) {
        double c;
        psd_z s, r;
        psd_zgivens(w.at(1, j, j), w.at(1, j + 1, j), c, s, r);
        PSD_WAVE_SYNC();
        PSD_ONE {
            w.at(1, j, j) = r;
            w.at(1, j + 1, j) = zmk(0.0, 0.0);
        }
        PSD_WAVE_SYNC();
        psd_zwin_left(w, 1, j, c, s, j + 1, ilastm);
        psd_zrecord(P, lcnt, 1, j, c, s);
        for (int l = p; l >= 2; --l) {
            if (plain || !ziszero(s)) {
                psd_zwin_right(w, l, j, c, s, ifirstm, j + 1);
                double tol = zabs(w.at(l, j, j)) + zabs(w.at(l, j + 1, j + 1));
                if (tol == 0) {  // opnorm(view(Hl, jlo:j+1, jlo:j+1), 1) restricted to the window
                    for (int cc = w.bs; cc <= j + 1; ++cc) {
                        double cs = 0.0;
                        for (int rr = w.bs; rr <= j + 1; ++rr) cs += zabs(w.at(l, rr, cc));
                        tol = fmax(tol, cs);
                    }
                }
                tol = fmax(st.ulp * tol, st.smlnum);
                const psd_z sub = w.at(l, j + 1, j);
                if (!plain && zabs(sub) <= tol) {
                    c = 1.0;
                    s = zmk(0.0, 0.0);
                    PSD_WAVE_SYNC();
                    PSD_ONE { w.at(l, j + 1, j) = zmk(0.0, 0.0); }
                    PSD_WAVE_SYNC();
                } else {
                    psd_zgivens(w.at(l, j, j), sub, c, s, r);
                    PSD_WAVE_SYNC();
                    PSD_ONE {
                        w.at(l, j, j) = r;
                        w.at(l, j + 1, j) = zmk(0.0, 0.0);
                    }
                    PSD_WAVE_SYNC();
                    psd_zwin_left(w, l, j, c, s, j + 1, ilastm);
                }
            }
            psd_zrecord(P, lcnt, l, j, c, s);
        }
        // right side of the Hessenberg factor is applied after the whole pass (:436-444)
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
    psd_zwin_store(P, w, n, p);
    const bool last = ke >= jend;
    psd_zdesc_write(P, st, lcnt, ks, ke + 1, w.be + 1, ilastm, ifirstm, w.bs - 1, 1, last ? 1 : 0, jlo, jend);
    st.nwindows += 1;
    st.kcur = ke + 1;
    if (last) {
        if (plain) {
            if (st.jdeflate < ilast) {
                st.phase = PSD_ZPH_CASE2B;
                st.kcur = ilast;
            } else {
                st.phase = PSD_ZPH_CHECK;
            }
        } else {
            st.ziter = st.zflag ? 1 : 0;
            st.phase = PSD_ZPH_CHECK;
            if (st.pend2) psd_zq_start_case2(P, st);
        }
    }
}

// generalized.jl:512-564: one window of the second (upward) half of Case II: positions j = kcur
// downwards to jdeflate+1; rotations are generated by annihilating H[j,j-1] from the right, the chain
// runs H_1 -> H_2 -> ... -> H_p, and the left side of H_1 is deferred to psd_zq_defer.
// A rotation generated at H_l is owned by m = l+1 (cyclic): right on H_l = H_{m-1}, Z_m, left on H_m.
PSD_D void psd_zq_case2b_window(const psd_zparams& P, psd_zstate& st, psd_z* ldsz, int* lcnt) {
    const int n = st.n, p = st.p, ilast = st.ilast, ifirstm = st.ifirstm, ilastm = st.ilastm;
    const int jend = st.jdeflate + 1;
    const int nb = st.W - 2;
    const int ks = st.kcur;
    const int ke = (ks - nb + 1 > jend) ? (ks - nb + 1) : jend;
    psd_zwin w;
    w.b = ldsz;
    w.W = st.W;
    w.ld = st.W + 1;
    w.bsz = st.W * (st.W + 1);
    w.bs = ke - 1;
    w.be = ks;
    PSD_PAR_FOR(m, p) { lcnt[m] = 0; }
    psd_zwin_load(P, w, n, p);
    for (int j = ks; j >= ke; --j) {
        double c;
        psd_z s, r;
        psd_zgivens(w.at(1, j, j), w.at(1, j, j - 1), c, s, r);
        PSD_WAVE_SYNC();
        PSD_ONE {
            w.at(1, j, j) = r;
            w.at(1, j, j - 1) = zmk(0.0, 0.0);
        }
        PSD_WAVE_SYNC();
        psd_z sr = zneg(s);  // Gtmp[j] = Givens(j-1, j, c, -s)
        psd_zwin_right(w, 1, j - 1, c, sr, ifirstm, j - 1);
        psd_zrecord(P, lcnt, (p >= 2) ? 2 : 1, j - 1, c, sr);
        for (int l = 2; l <= p; ++l) {
            psd_zwin_left(w, l, j - 1, c, sr, j - 1, ilastm);
            psd_zgivens(w.at(l, j, j), w.at(l, j, j - 1), c, s, r);
            PSD_WAVE_SYNC();
            PSD_ONE {
                w.at(l, j, j) = r;
                w.at(l, j, j - 1) = zmk(0.0, 0.0);
            }
            PSD_WAVE_SYNC();
            sr = zneg(s);
            psd_zwin_right(w, l, j - 1, c, sr, ifirstm, j - 1);
            psd_zrecord(P, lcnt, (l == p) ? 1 : (l + 1), j - 1, c, sr);
        }
        PSD_ONE {
            psd_ztr tr;
            tr.pos = j - 1;
            tr.pad = 0;
            tr.c = c;
            tr.s = sr;
            P.dG[j] = tr;
        }
    }
    psd_zwin_store(P, w, n, p);
    const bool last = ke <= jend;
    // deferred left side of H_1 covers j = ilast..jdeflate+2 (:561-564)
    psd_zdesc_write(P, st, lcnt, w.bs, w.be, w.be + 1, ilastm, ifirstm, w.bs - 1, 2, last ? 2 : 0, st.jdeflate + 2, ilast);
    st.nwindows += 1;
    st.kcur = ke - 1;
    if (last) st.phase = PSD_ZPH_CHECK;
}

PSD_D void psd_zq_step_body(const psd_zparams& P) {
    PSD_LDS_DECL;
    psd_zstate st = *P.st;
    if (st.phase == PSD_ZPH_DONE) {
        PSD_ONE { P.desc->active = 0; P.desc->defer_run = 0; }
        return;
    }
    const int NT = PSD_NTHREADS;
    psd_z* ldsz = (psd_z*)psd_lds;
    const size_t winb = (size_t)st.p * st.Wmax * (st.Wmax + 1);
    double* red = (double*)(ldsz + winb);
    int* redi = (int*)(red + NT);
    int* lcnt = redi + 2 * NT;
    PSD_ONE { P.desc->active = 0; P.desc->defer_run = 0; }
    const long long tk0 = psd_clock(), tw0 = psd_wallclock();
    bool emitted = false;
    int guard = 0;
    while (!emitted && st.phase != PSD_ZPH_DONE && guard < 4 * st.n + 16) {
        ++guard;
        if (st.phase == PSD_ZPH_CHECK) {
            const long long td0 = psd_clock();
            psd_zq_check(P, st, red, redi, ldsz);
            st.cyc[0] += psd_clock() - td0;
        } else if (st.phase == PSD_ZPH_SWEEP) {
            psd_zq_sweep_window(P, st, ldsz, lcnt);
            emitted = true;
        } else if (st.phase == PSD_ZPH_ZSHIFT) {
            psd_zq_zshift_window(P, st, ldsz, lcnt, false);
            emitted = true;
        } else if (st.phase == PSD_ZPH_CASE2A) {
            psd_zq_zshift_window(P, st, ldsz, lcnt, true);
            emitted = true;
        } else if (st.phase == PSD_ZPH_CASE2B) {
            psd_zq_case2b_window(P, st, ldsz, lcnt);
            emitted = true;
        } else if (st.phase == PSD_ZPH_TWAIT) {  // the leader's sweep is done: wait for the cursors of the train
            bool all = true;
            // (a cursor counts itself in after its last store; its slot's state is then complete)
            all = psd_atomic_load(P.cep + PSD_TRAIN_MAX) == st.train_n - 1;
            if (all) psd_acquire_fence();
            if (all) {
                for (int b = 1; b < st.train_n; ++b) {
                    st.nwindows += P.cst[b].nwindows;
                    st.nsweeps += 1;
                    st.iiter += 1;
                    st.jiter += 1;
                    // (cycle counters: the cursors' windows count like the leader's; what a cursor spends outside its windows — state
                    //  read, start rotation — goes with the decisions)
                    for (int q = 1; q <= 3; ++q) st.cyc[q] += P.cst[b].cyc[q];
                    st.cyc[0] += P.cst[b].cyc[4] - (P.cst[b].cyc[1] + P.cst[b].cyc[2] + P.cst[b].cyc[3]);
                }
                st.train_n = 1;
                st.W = st.Wmax;
                st.phase = PSD_ZPH_CHECK;
            }
            emitted = true;  // (the check runs in the next launch, behind the cursors' last bulk updates)
        } else {
            st.phase = PSD_ZPH_DONE;
        }
    }
    st.cyc[4] += psd_clock() - tk0;
    st.cyc[5] += psd_wallclock() - tw0;
    if (st.info == PSD_LIST_OVERFLOW) st.phase = PSD_ZPH_DONE;  // (a window that overran a list ends the call)
    PSD_SYNC();
    PSD_ONE { *P.st = st; }
}
PSD_KERNEL_B(PSD_ZC3_WAVES * PSD_STEP_NT) psd_zq_step(psd_zparams P) {
    PSD_ZC_ENTER(P);
    psd_zq_step_body(P);
    PSD_ZC_LEAVE(P);
}

// cursor b >= 1 of a multishift train (see psd_rq_cursor_body)
PSD_D void psd_zq_cursor_body(const psd_zparams& P, int b) {
    PSD_LDS_DECL;
    PSD_ONE { P.desc->active = 0; P.desc->defer_run = 0; }
    const long long tb0 = psd_clock();
    psd_zstate st;
    if (!psd_pub_read(P.cep + b, P.tick, P.st, st)) return;  // (published in an earlier launch, not being rewritten)
    if (st.cursor != b) return;
    if (st.phase != PSD_ZPH_CWAIT && st.phase != PSD_ZPH_SWEEP) return;
    psd_z* ldsz = (psd_z*)psd_lds;
    const size_t winb = (size_t)st.p * st.Wmax * (st.Wmax + 1);
    int* lcnt = (int*)((double*)(ldsz + winb) + PSD_STEP_NT) + 2 * PSD_STEP_NT;
    if (st.phase == PSD_ZPH_CWAIT) {
        if (P.tick < st.cstart) return;
        if (!psd_zq_start_rot_mu(P, st.n, st.p, st.ifirst, st.shift, st.c0, st.s0))
            psd_zq_start_rot(P, st.n, st.p, st.ifirst, st.shidx, st.c0, st.s0);
        st.kcur = st.ifirst;
        st.phase = PSD_ZPH_SWEEP;
    }
    psd_zq_sweep_window(P, st, ldsz, lcnt);
    st.cyc[4] += psd_clock() - tb0;
    PSD_SYNC();
    PSD_ONE {
        *P.st = st;
        if (st.phase == PSD_ZPH_CDONE) {  // last window: count this cursor in (its state and lists are out first)
            psd_release_fence();
            psd_atomic_add(P.cep + PSD_TRAIN_MAX, 1);
        }
    }
}

// all cursors of a tick in one launch (see psd_rq_step_train)
PSD_KERNEL_B(PSD_ZC3_WAVES * PSD_STEP_NT) psd_zq_step_train(psd_zparams P, int p, int cstride) {
#ifndef PSD_HOSTSIM
    if (P.zslG > 1 && PSD_BLOCK_Y >= 1) {  // grid (cursors, slices): a worker of cursor slot blockIdx.x (psd_zslice3.h)
        psd_zsl_worker(P, PSD_BLOCK_X, PSD_BLOCK_Y);
        return;
    }
#endif
    PSD_ZC_ENTER(P);
    const int b = PSD_BLOCK_X;
    if (b == 0) {
        psd_zq_step_body(P);
#ifndef PSD_HOSTSIM
        if (P.zslG > 1) psd_zsl_finish(P, 0);
#endif
        PSD_ZC_LEAVE(P);
        return;
    }
    psd_zparams Q = P;
    Q.st = P.cst + b;
    Q.desc = P.desc + b;
    Q.cnt = P.cnt + (size_t)b * cstride;
    Q.tr = P.tr + (size_t)b * p * PSD_ZTR_CAP;
    psd_zq_cursor_body(Q, b);
#ifndef PSD_HOSTSIM
    if (P.zslG > 1) psd_zsl_finish(P, b);
#endif
    PSD_ZC_LEAVE(P);
}

// Bulk application of one window's rotation lists.  grid = (tiles, p owners, 3 roles) as in the
// real path; tiles are 64 wide (16-byte elements).
#define PSD_ZAPPLY_NT 64
// As in psd_rq_apply: a thread keeps its line (<= 32 complex elements) in registers through the whole list when the list
// visits its positions monotonically; fully unrolled position loop, list read ahead from LDS behind two sentinels.
#define PSD_ZTR_LDS_RECS (PSD_ZTR_CAP + 2)
#define PSD_ZTR_LDS_BYTES (sizeof(psd_ztr) * PSD_ZTR_LDS_RECS + 16)
template <bool UP, bool LEFT>
PSD_D void psd_ztr_regline(const psd_ztr* ltr, int plo, psd_z (&a)[33]) {
    int e = 0;
    psd_ztr cur = ltr[0], nxt = ltr[1];
#pragma unroll
    for (int q = 0; q < 32; ++q) {
        const int b = UP ? q : 31 - q;
        while (cur.pos - plo == b) {
            if (LEFT) psd_zrot_left(cur.c, cur.s, a[b], a[b + 1]);
            else psd_zrot_right_adj(cur.c, cur.s, a[b], a[b + 1]);
            cur = nxt;
            ++e;
            nxt = ltr[e + 1 < PSD_ZTR_LDS_RECS ? e + 1 : PSD_ZTR_LDS_RECS - 1];
        }
    }
}
PSD_D int psd_ztr_stage(const psd_ztr* gtr, int cnt, psd_ztr* ltr, int* flags) {
    PSD_PAR_FOR(e, PSD_ZTR_LDS_RECS) {
        psd_ztr tr;
        if (e < cnt) {
            tr = gtr[e];
        } else {
            tr.pos = 0x3fffffff;
            tr.pad = 0;
            tr.c = 1.0;
            tr.s = zmk(0.0, 0.0);
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

// part (right role only): 0 all its rows, 1 the near rows (rcut .. rr1), 2 the far rows (rr0 .. rcut - 1)
PSD_D void psd_zq_apply_item(const psd_zparams& P, int n, int p, int role, int bx, int m, int part = 0) {
    PSD_LDS_DECL;
    const psd_zapply_desc d = *P.desc;
    if (!d.active) return;
    const int cnt = P.cnt[m - 1] < PSD_ZTR_CAP ? P.cnt[m - 1] : PSD_ZTR_CAP;
    if (cnt <= 0) return;
    const int T = PSD_ZAPPLY_NT;
    const int S = d.phi - d.plo + 1;
    psd_ztr* ltr = (psd_ztr*)psd_lds;
    int* flags = (int*)(psd_lds + sizeof(psd_ztr) * PSD_ZTR_LDS_RECS);
    psd_z* tile = (psd_z*)(psd_lds + PSD_ZTR_LDS_BYTES);
    const psd_ztr* gtr = P.tr + (size_t)(m - 1) * PSD_ZTR_CAP;
    const psd_z z0 = zmk(0.0, 0.0);
    if (role == 0) {
        if (d.defer_h1 == 2 && m == 1) return;  // H_1's left side is deferred (upward pass of Case II)
        const int c0 = d.lc0 + bx * T;
        if (c0 > d.lc1) return;
        const int nc = (d.lc1 - c0 + 1 < T) ? (d.lc1 - c0 + 1) : T;
        const psd_mat<psd_z> M = psd_mat<psd_z>{P.H + (size_t)(m - 1) * n * n, n};
        const int ldt = T + 1;
        // rows panel -> LDS (a thread owns a column); eight loads in flight per thread
        // (S <= 32 rows; RW = 16 or 32 row lanes, ncb = 64 / RW column phases: thread t0 covers row t0 % RW of the columns
        //  t0 / RW + ncb k — at p = 64 a window has 11 rows: with 32 row lanes two thirds of the wavefront sat idle here)
        const int RW = (S > 16) ? 32 : 16, ncb = PSD_ZAPPLY_NT / RW;
        PSD_PAR_FOR(t0, PSD_ZAPPLY_NT) {
            const int r = t0 & (RW - 1), cb = t0 / RW;
            if (r >= S) continue;
            for (int k0 = 0; k0 < T / ncb; k0 += 8) {
                psd_z v[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    const int c = cb + ncb * (k0 + u);
                    v[u] = z0;
                    if (c < nc) v[u] = M(d.plo + r, c0 + c);
                }
#pragma unroll
                for (int u = 0; u < 8; ++u) tile[r * ldt + cb + ncb * (k0 + u)] = v[u];
            }
        }
        const int order = psd_ztr_stage(gtr, cnt, ltr, flags);
        PSD_PAR_FOR(c, nc) {
            if (order != 0) {
                psd_z a[33];
#pragma unroll
                for (int r = 0; r < 33; ++r) {
                    a[r] = z0;
                    if (r < S) a[r] = tile[r * ldt + c];
                }
                if (order > 0) psd_ztr_regline<true, true>(ltr, d.plo, a);
                else psd_ztr_regline<false, true>(ltr, d.plo, a);
#pragma unroll
                for (int r = 0; r < 32; ++r)
                    if (r < S) tile[r * ldt + c] = a[r];
                continue;
            }
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
        PSD_PAR_FOR(t0, PSD_ZAPPLY_NT) {
            const int r = t0 & (RW - 1), cb = t0 / RW;
            if (r >= S) continue;
            for (int k0 = 0; k0 < T / ncb; k0 += 8) {
                psd_z v[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) v[u] = tile[r * ldt + cb + ncb * (k0 + u)];
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    const int c = cb + ncb * (k0 + u);
                    if (c < nc) M(d.plo + r, c0 + c) = v[u];
                }
            }
        }
    } else {
        const int jm = (role == 1) ? ((m == 1) ? p : (m - 1)) : m;
        if (role == 1 && d.defer_h1 == 1 && jm == 1) return;  // H_1's right side is deferred (downward passes)
        if (role == 2 && (m < P.zlo || m > P.zhi)) return;  // (another rank's Schur vectors)
        int lo = (role == 1) ? d.rr0 : d.zr0;
        int hi = (role == 1) ? d.rr1 : d.zr1;
        if (role == 1 && part == 1 && lo < d.rcut) lo = d.rcut;
        if (role == 1 && part == 2 && hi > d.rcut - 1) hi = d.rcut - 1;
        const int r0 = lo + bx * T;
        if (r0 > hi) return;
        const int nr = (hi - r0 + 1 < T) ? (hi - r0 + 1) : T;
        psd_z* base = (role == 1) ? P.H : P.Z;
        const psd_mat<psd_z> M = psd_mat<psd_z>{base + (size_t)(jm - 1) * n * n, n};
        const int order = psd_ztr_stage(gtr, cnt, ltr, flags);
        if (order != 0) {
            // a thread owns a row of the columns panel: coalesced loads straight into registers, no LDS tile
            PSD_PAR_FOR(r, nr) {
                psd_z a[33];
#pragma unroll
                for (int c = 0; c < 33; ++c) {
                    a[c] = z0;
                    if (c < S) a[c] = M(r0 + r, d.plo + c);
                }
                if (order > 0) psd_ztr_regline<true, false>(ltr, d.plo, a);
                else psd_ztr_regline<false, false>(ltr, d.plo, a);
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
            M(r0 + r, d.plo + c) = tile[c * T + r];
        }
    }
}

PSD_D void psd_zq_apply_body(const psd_zparams& P, int n, int p, int role) { psd_zq_apply_item(P, n, p, role, PSD_BLOCK_X, PSD_BLOCK_Y + 1); }

PSD_KERNEL_B(PSD_ZAPPLY_NT) psd_zq_apply(psd_zparams P, int n, int p) { psd_zq_apply_body(P, n, p, PSD_BLOCK_Z); }

// bulk updates of all cursors of a tick in two launches (see psd_rq_apply_train)
PSD_KERNEL_B(PSD_ZAPPLY_NT) psd_zq_apply_train(psd_zparams P, int n, int p, int cstride, int pass) {
    // pass 0: rows of H_m and Schur vectors (grid.z = 2 M), 1: columns of H_{m-1}, 2: rows alone, 3: Schur vectors alone
    // (grid.z = M; the Z updates of a tick as their own launch on the second stream, see ziterate_dev)
    const int z = PSD_BLOCK_Z;
    const int b = (pass == 0) ? (z >> 1) : z;
    const int role = (pass == 0) ? ((z & 1) ? 2 : 0) : ((pass == 1) ? 1 : ((pass == 2) ? 0 : 2));
    psd_zparams Q = P;
    Q.desc = P.desc + b;
    Q.cnt = P.cnt + (size_t)b * cstride;
    Q.tr = P.tr + (size_t)b * p * PSD_ZTR_CAP;
    psd_zq_apply_body(Q, n, p, role);
}

// The same updates as a work list (as psd_rq_apply_wl of the real engine): a fixed grid of single-wave workgroups loops over
// the items (cursor, owner, 64-line tile) of the tick's ACTIVE cursors.  The grid-per-cursor launches start tiles x p x M
// workgroups of which most find their cursor idle (8 of 48 cursors have a window in an average tick at n = 1024) and all pay
// the descriptor and list staging before their one tile.  pass: 1 columns of H_{m-1}, 2 rows of H_m, 3 Schur vectors.
// tbl: byte offset of the item table behind the tile in LDS (2 (M + 1) ints).
PSD_KERNEL_B(PSD_ZAPPLY_NT) psd_zq_apply_wl(psd_zparams P, int n, int p, int cstride, int pass, int M, int tbl) {
    PSD_LDS_DECL;
    int* ioff = (int*)(psd_lds + tbl);  // [M + 1]
    int* nt = ioff + (M + 1);           // [M]
    const int T = PSD_ZAPPLY_NT;
    // (pass 4 / 5: the near / far rows of the column roles, psd_zparams::zcdefer)
    const int role = (pass == 1 || pass == 4 || pass == 5) ? 1 : ((pass == 2) ? 0 : 2);
    const int part = (pass == 4) ? 1 : ((pass == 5) ? 2 : 0);
    PSD_PAR_FOR(b, M) {
        const psd_zapply_desc d = P.desc[b];
        int q = 0;
        if (d.active) {
            int lo = (role == 0) ? d.lc0 : ((role == 1) ? d.rr0 : d.zr0);
            int hi = (role == 0) ? d.lc1 : ((role == 1) ? d.rr1 : d.zr1);
            if (part == 1 && lo < d.rcut) lo = d.rcut;
            if (part == 2 && hi > d.rcut - 1) hi = d.rcut - 1;
            if (hi >= lo) q = (hi - lo + T) / T;
        }
        nt[b] = q;
    }
    PSD_SYNC();
    PSD_ONE {
        int acc = 0;
        for (int b = 0; b < M; ++b) {
            ioff[b] = acc;
            acc += p * nt[b];
        }
        ioff[M] = acc;
    }
    PSD_SYNC();
    const int total = ioff[M];
    for (int item = PSD_BLOCK_X; item < total; item += PSD_GRID_X) {
        int b = 0;
        {  // last b with ioff[b] <= item (idle cursors repeat an offset)
            int lo_ = 0, hi_ = M - 1;
            while (lo_ < hi_) {
                const int mid = (lo_ + hi_ + 1) >> 1;
                if (ioff[mid] <= item) lo_ = mid;
                else hi_ = mid - 1;
            }
            b = lo_;
        }
        const int q = item - ioff[b];
        const int m = q / nt[b] + 1, bx = q - (m - 1) * nt[b];
        psd_zparams Q = P;
        Q.desc = P.desc + b;
        Q.cnt = P.cnt + (size_t)b * cstride;
        Q.tr = P.tr + (size_t)b * p * PSD_ZTR_CAP;
        psd_zq_apply_item(Q, n, p, role, bx, m, part);
        PSD_SYNC();  // (the list and the tile in LDS are reused by the next item)
    }
}

// Deferred right side of H_1 after a zero-shift pass (generalized.jl:436-444):
// for j = djlo..djhi: rmul!(view(H1, drow0:(j+1), :), G_j').  One thread per row.
PSD_KERNEL psd_zq_defer(psd_zparams P, int n) {
    const psd_zapply_desc d = *P.desc;
    if (!d.active || !d.defer_run) return;
    const psd_mat<psd_z> H1 = psd_mat<psd_z>{P.H, n};
    const int NT = PSD_NTHREADS;
    if (d.defer_run == 1) {
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
    } else {
        // generalized.jl:561-564: for j = djhi:-1:djlo  lmul!(G_j, view(H1, :, (j-1):ilastm)), rows (j-1, j);
        // one thread per column c (>= djlo-1), rotations j <= c+1, streamed up the column
        const int cbase = (d.djlo - 1) + PSD_BLOCK_X * NT;
        PSD_PAR_FOR(t, NT) {
            const int c = cbase + t;
            if (c <= d.lc1 && d.djhi >= d.djlo) {
                int j = (c + 1 < d.djhi) ? (c + 1) : d.djhi;
                if (j >= d.djlo) {
                    psd_z a2 = H1(j, c);
                    for (; j >= d.djlo; --j) {
                        const psd_ztr g = P.dG[j];
                        psd_z a1 = H1(j - 1, c);
                        psd_zrot_left(g.c, g.s, a1, a2);
                        H1(j, c) = a2;
                        a2 = a1;
                    }
                    H1(d.djlo - 1, c) = a2;
                }
            }
        }
    }
}

PSD_KERNEL psd_zq_init(psd_zparams P, int n, int p, int wantT, int wantZ, int W, int maxitfac, int maxlog,
                       int train_want, int train_oc) {
    const psd_mat<psd_z> H1 = psd_mat<psd_z>{P.H, n};
    PSD_PAR_FOR(c, n) {
        for (int r = c + 3; r <= n; ++r) H1(r, c + 1) = zmk(0.0, 0.0);  // _gethess!
    }
    PSD_ONE {
        psd_zstate st;
        st.n = n; st.p = p; st.wantT = wantT; st.wantZ = wantZ; st.W = st.Wmax = W; st.train_oc = train_oc;
        st.phase = PSD_ZPH_CHECK; st.info = 0;
        st.ilast = n; st.ifirst = -1; st.ifirstm = 1; st.ilastm = n; st.iiter = 1; st.settle = -4;
        // generalized.jl:199: p >= log2(floatmin)/log2(eps) = 19.65
        st.ziter = (p >= 20) ? -1 : 0;
        st.jiter = 0; st.maxit = maxitfac * n;
        st.jlo = 1; st.kcur = 0; st.zflag = 0;
        st.ldeflate = st.jdeflate = -1; st.ncase2 = 0; st.pend2 = 0;
        st.nsweeps = st.nzshift = st.nsplit = st.nwindows = st.nlog = 0;
        st.maxlog = maxlog;
        st.c0 = 1.0; st.s0 = zmk(0.0, 0.0);
        st.ulp = PSD_DBL_EPS;
        st.safmin = PSD_DBL_MIN;
        st.smlnum = PSD_DBL_MIN * ((double)n / PSD_DBL_EPS);
        for (int q = 0; q < 6; ++q) st.cyc[q] = 0;
        st.train_want = train_want; st.train_n = 1; st.train_id = 0; st.cursor = 0; st.train_tick0 = 0; st.cstart = 0; st.cfirst = 0;
        st.shidx = n; st.exc_dec = 0; st.ntrainsweeps = 0; st.shift = zmk(0.0, 0.0);
        *P.st = st;
        P.desc->active = 0;
        P.desc->defer_run = 0;
    }
}

// generalized.jl:860-908 for factor l (S all true): diag(T_l) real >= 0, phases folded into row j of
// T_l, column j of Z_l and column j of T_{l-1}.  grid = n blocks (one per j); launched for l = p..2.
PSD_KERNEL psd_zq_phase(psd_zparams P, int n, int l, int wantZ) {
    PSD_LDS_DECL;
    psd_z* zs = (psd_z*)psd_lds;
    const int j = PSD_BLOCK_X + 1;
    const psd_mat<psd_z> Hl = psd_mat<psd_z>{P.H + (size_t)(l - 1) * n * n, n};
    const psd_mat<psd_z> Hm = psd_mat<psd_z>{P.H + (size_t)(l - 2) * n * n, n};
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
    const psd_z zc = zconj(z);
    PSD_PAR_FOR(t, n - j) { Hl(j, j + 1 + t) = zmul(Hl(j, j + 1 + t), z); }
    if (wantZ && l >= P.zlo && l <= P.zhi) {
        const psd_mat<psd_z> Zl = psd_mat<psd_z>{P.Z + (size_t)(l - 1) * n * n, n};
        PSD_PAR_FOR(r, n) { Zl(r + 1, j) = zmul(Zl(r + 1, j), zc); }
    }
    PSD_PAR_FOR(r, j) { Hm(r + 1, j) = zmul(Hm(r + 1, j), zc); }
}
