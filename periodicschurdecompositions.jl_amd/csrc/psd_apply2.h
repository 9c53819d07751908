// Bulk application of a tick's transform lists, register-line form (round 3; HIP only — the simulated tier keeps the
// serial psd_rq_apply_wl, which applies the same records to the same lines in the same order).
//
// What it replaces: the off-window updates of the periodic QR sweep — lmul!(H', view(H_m, k:k+2, k+1:n)),
// rmul!(view(H_{m-1}, 1:k+3, k:k+2), H), rmul!(view(Z_m, :, k:k+2), H) and their two-wide companions,
// /root/reference/src/PeriodicSchurDecompositions.jl:834-842,855-881 — applied per window from the lists the chase
// kernel emitted (psd_real_qr.h), every element read once and written once (2 8 p w (2n + 1) bytes
// per sweep).
//
// Why a second form.  psd_rq_apply_wl (round 2) runs single-wave workgroups that stage a 17 KiB tile in LDS and stream
// every line through it: eight waves per compute unit, one tile's loads in flight per wave, 8-byte accesses on the
// column roles — 2.7 TB/s by the counters at n = 1024, p = 64.  This kernel is bounded by HBM latency x bytes in flight,
// not by arithmetic (ten multiply-adds per position and line), so:
//   * column roles (right transforms on rows of H_{m-1} and Z_m: three quarters of the bytes): a lane owns TWO
//     consecutive rows and keeps both lines in registers for the whole list — S <= SP 16-byte loads per lane, all in
//     flight at once, no LDS tile, the position loop unrolled so that every register index is static;
//   * rows role (left transforms on columns of H_m): 16-byte loads of row pairs as before, transposed through a
//     per-wave LDS tile of pitch S + 1 (not 33), then the same register-line routine;
//   * the unit of work is the wavefront (four per workgroup, each with its own staged list and tile: no workgroup
//     barrier per item); the record that is being applied is wave-uniform and lives in scalar registers
//     (v_readfirstlane), so the dispatch is scalar branches and the multiply-adds take a scalar operand;
//   * occupancy follows from registers (SP = 17: four waves per SIMD), not from a 17 KiB tile per wave.
#pragma once
#ifndef PSD_HOSTSIM

#define PSD_WL2_NT 256
#define PSD_WL2_WAVES 4
#define PSD_WL2_TPI 8  // tiles per item at most
#define PSD_WL2_UNI(x) __builtin_amdgcn_readfirstlane(x)
// pitch of a wave's LDS tile in doubles: odd, so that 32 lanes reading 8 bytes at a stride of one pitch meet 32 bank pairs
#define PSD_WL2_LD(SP) (((SP) + 1) | 1)

template <int SP>
PSD_HD size_t psd_wl2_lds_bytes() {  // per wave: its copy of the list and its tile; per workgroup: the item table
    return (size_t)PSD_WL2_WAVES * (PSD_TR_LDS_BYTES + (size_t)64 * PSD_WL2_LD(SP) * sizeof(double)) + (size_t)(3 * PSD_SLOTS + 8) * sizeof(int);
}

// The record being applied is the same for every lane: all of it lives in scalar registers (v_readfirstlane after the
// LDS read), the multiply-adds take one scalar operand each — twelve vector registers less than two records in flight.
struct psd_wl2_rec {
    int pos, kind;
    double c0, c1, c2;
};
PSD_D double psd_wl2_uni_f64(double x) {
    return __hiloint2double(__builtin_amdgcn_readfirstlane(__double2hiint(x)), __builtin_amdgcn_readfirstlane(__double2loint(x)));
}
PSD_D psd_wl2_rec psd_wl2_fetch(const psd_tr* ltr, int e) {
    const psd_tr t = ltr[e < PSD_TR_LDS_RECS ? e : PSD_TR_LDS_RECS - 1];
    psd_wl2_rec r;
    r.pos = __builtin_amdgcn_readfirstlane(t.pos);
    r.kind = __builtin_amdgcn_readfirstlane(t.kind);
    r.c0 = psd_wl2_uni_f64(t.c0);
    r.c1 = psd_wl2_uni_f64(t.c1);
    r.c2 = psd_wl2_uni_f64(t.c2);
    return r;
}

// NL lines per lane, each SP + 2 registers; positions ascend (UP) or descend through the list
template <int SP, int NL, bool UP>
PSD_D void psd_wl2_regline(const psd_tr* ltr, int plo, double (&a)[NL][SP + 2]) {
    int e = 0;
    psd_wl2_rec cur = psd_wl2_fetch(ltr, 0), nxt = psd_wl2_fetch(ltr, 1);
#pragma unroll
    for (int q = 0; q < SP; ++q) {
        const int b = UP ? q : SP - 1 - q;
        while (cur.pos - plo == b) {
            if (cur.kind == PSD_TR_R3) {
#pragma unroll
                for (int l = 0; l < NL; ++l) {
                    const double x = cur.c2 * (a[l][b] + cur.c0 * a[l][b + 1] + cur.c1 * a[l][b + 2]);
                    a[l][b] -= x;
                    a[l][b + 1] -= x * cur.c0;
                    a[l][b + 2] -= x * cur.c1;
                }
            } else if (cur.kind == PSD_TR_H2) {
#pragma unroll
                for (int l = 0; l < NL; ++l) {
                    const double x = cur.c2 * (a[l][b] + cur.c0 * a[l][b + 1]);
                    a[l][b] -= x;
                    a[l][b + 1] -= x * cur.c0;
                }
            } else if (cur.kind == PSD_TR_R2) {
#pragma unroll
                for (int l = 0; l < NL; ++l) {
                    const double s = a[l][b] * cur.c0 + a[l][b + 1] * cur.c1;
                    a[l][b] -= s * (cur.c2 * cur.c0);
                    a[l][b + 1] -= s * (cur.c2 * cur.c1);
                }
            } else {
#pragma unroll
                for (int l = 0; l < NL; ++l) {
                    const double b1 = cur.c0 * a[l][b] + cur.c1 * a[l][b + 1];
                    const double b2 = -cur.c1 * a[l][b] + cur.c0 * a[l][b + 1];
                    a[l][b] = b1;
                    a[l][b + 1] = b2;
                }
            }
            cur = nxt;
            ++e;
            nxt = psd_wl2_fetch(ltr, e + 1);
        }
    }
}

typedef double psd_wl2_v2 __attribute__((ext_vector_type(2), aligned(8)));
typedef unsigned psd_wl2_u4 __attribute__((ext_vector_type(4)));
typedef unsigned psd_wl2_u2 __attribute__((ext_vector_type(2)));
#define PSD_WL2_OOR 0xfffffff0u  // a byte offset beyond any matrix: the buffer load returns zeros, the store is dropped

// Buffer addressing: ONE 32-bit offset register per lane for the whole tile, the column step in a scalar register
// (flat addressing kept one 64-bit address per column and lane alive: 34 more registers, a wave less per SIMD).  Lanes
// outside the tile get an out-of-range offset instead of a branch.  n x n doubles must stay below 4 GiB (n <= 23 000).
PSD_D __amdgpu_buffer_rsrc_t psd_wl2_rsrc(const psd_mat<double>& Mx) {
    return __builtin_amdgcn_make_buffer_rsrc(Mx.a, 0, (int)((size_t)Mx.ld * Mx.ld * 8), 0x00020000);
}
PSD_D void psd_wl2_ld16(__amdgpu_buffer_rsrc_t rs, unsigned voff, unsigned soff, double& x, double& y) {
    const psd_wl2_u4 t = __builtin_amdgcn_raw_buffer_load_b128(rs, voff, soff, 0);
    x = __hiloint2double((int)t.y, (int)t.x);
    y = __hiloint2double((int)t.w, (int)t.z);
}
PSD_D void psd_wl2_st16(__amdgpu_buffer_rsrc_t rs, unsigned voff, unsigned soff, double x, double y) {
    psd_wl2_u4 t;
    t.x = (unsigned)__double2loint(x);
    t.y = (unsigned)__double2hiint(x);
    t.z = (unsigned)__double2loint(y);
    t.w = (unsigned)__double2hiint(y);
    __builtin_amdgcn_raw_buffer_store_b128(t, rs, voff, soff, 0);
}
PSD_D double psd_wl2_ld8(__amdgpu_buffer_rsrc_t rs, unsigned voff, unsigned soff) {
    const psd_wl2_u2 t = __builtin_amdgcn_raw_buffer_load_b64(rs, voff, soff, 0);
    return __hiloint2double((int)t.y, (int)t.x);
}
PSD_D void psd_wl2_st8(__amdgpu_buffer_rsrc_t rs, unsigned voff, unsigned soff, double x) {
    psd_wl2_u2 t;
    t.x = (unsigned)__double2loint(x);
    t.y = (unsigned)__double2hiint(x);
    __builtin_amdgcn_raw_buffer_store_b64(t, rs, voff, soff, 0);
}

// one tile of a column role: rows l0 .. l0 + nl - 1 (1-based) of Mx, columns plo .. plo + S - 1; lane = row pair.
// FULL: nl = 128 (no tail lane).  Otherwise the lane that holds the odd last row reads the pair one row up and takes
// its second half (no second load to keep alive), and stores 8 bytes; not for the one-row tile at the top of the
// matrix (l0 = 1, nl = 1), which has no row above.
template <int SP, bool FULL, bool UP>
PSD_D void psd_wl2_cols_tile(const psd_mat<double>& Mx, int plo, int S, int l0, int nl, int lane, const psd_tr* ltr) {
    const int r = 2 * lane;
    const bool ok1 = FULL || r + 1 < nl, tail = !FULL && r + 1 == nl;
    const __amdgpu_buffer_rsrc_t rs = psd_wl2_rsrc(Mx);
    const unsigned off = (unsigned)(l0 - 1 + r) * 8u;
    const unsigned vld = ok1 ? off : (tail ? off - 8u : PSD_WL2_OOR);
    const unsigned v16 = ok1 ? off : PSD_WL2_OOR, v8 = tail ? off : PSD_WL2_OOR;
    const unsigned cstep = (unsigned)Mx.ld * 8u;
    const unsigned s0 = (unsigned)(plo - 1) * cstep;
    const bool anytail = !FULL && (nl & 1);
    double a[2][SP + 2];
#pragma unroll
    for (int u = 0; u < SP + 2; ++u) {
        a[0][u] = a[1][u] = 0.0;
        if (u < SP && u < S) {
            psd_wl2_ld16(rs, vld, s0 + u * cstep, a[0][u], a[1][u]);
            if (!FULL && tail) a[0][u] = a[1][u];
        }
    }
    psd_wl2_regline<SP, 2, UP>(ltr, plo, a);
#pragma unroll
    for (int u = 0; u < SP; ++u) {
        if (u < S) {
            psd_wl2_st16(rs, v16, s0 + u * cstep, a[0][u], a[1][u]);
            if (anytail) psd_wl2_st8(rs, v8, s0 + u * cstep, a[0][u]);
        }
    }
}

// the LDS round trips of a single wavefront: DS operations execute in order, only the compiler has to be fenced
#define PSD_WL2_WAVE_FENCE() asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory")

// one tile of the rows role: columns l0 .. l0 + nl - 1 of Mx (nl <= 64), rows plo .. plo + S - 1 (contiguous in memory).
// Memory side: lane = (row pair 2 (lane & 15), column lane >> 4 of every group of four), 16-byte accesses (the lane
// with the odd last row: 8 bytes); LDS side: lane = line (column), odd pitch (conflict-free both ways).
template <int SP, bool UP>
PSD_D void psd_wl2_rows_tile(const psd_mat<double>& Mx, double* tile, int plo, int S, int l0, int nl, int lane, const psd_tr* ltr) {
    constexpr int LD = PSD_WL2_LD(SP);
    constexpr int NU = 16;  // column groups of four
    const int rr = 2 * (lane & 15), cq = lane >> 4;
    const __amdgpu_buffer_rsrc_t rs = psd_wl2_rsrc(Mx);
    const unsigned cstep = (unsigned)Mx.ld * 8u;
    const unsigned off = (unsigned)(l0 - 1 + cq) * cstep + (unsigned)(plo - 1 + rr) * 8u;
    const bool pair = rr + 1 < S, tail = rr + 1 == S;  // (tail: the odd last row; S >= 2, so it has a row above)
    const bool anytail = (S & 1) != 0;
    double v0[NU], v1[NU];
#pragma unroll
    for (int u = 0; u < NU; ++u) {
        v0[u] = v1[u] = 0.0;
        if (4 * u < nl) {
            const bool in = 4 * u + cq < nl;
            psd_wl2_ld16(rs, in ? (pair ? off : (tail ? off - 8u : PSD_WL2_OOR)) : PSD_WL2_OOR, 4u * u * cstep, v0[u], v1[u]);
            if (tail) v0[u] = v1[u];
        }
    }
    if (rr < S) {
#pragma unroll
        for (int u = 0; u < NU; ++u) {
            const int c = 4 * u + cq;
            tile[c * LD + rr] = v0[u];
            if (pair) tile[c * LD + rr + 1] = v1[u];
        }
    }
    PSD_WL2_WAVE_FENCE();
    if (lane < nl) {
        double a[1][SP + 2];
#pragma unroll
        for (int u = 0; u < SP + 2; ++u) a[0][u] = (u < SP && u < S) ? tile[lane * LD + u] : 0.0;
        psd_wl2_regline<SP, 1, UP>(ltr, plo, a);
#pragma unroll
        for (int u = 0; u < SP; ++u)
            if (u < S) tile[lane * LD + u] = a[0][u];
    }
    PSD_WL2_WAVE_FENCE();
#pragma unroll
    for (int u = 0; u < NU; ++u) {
        if (4 * u < nl) {
            const int c = 4 * u + cq;
            const bool in = c < nl;
            const double x0 = (rr < S) ? tile[c * LD + rr] : 0.0;
            const double x1 = pair ? tile[c * LD + rr + 1] : 0.0;
            psd_wl2_st16(rs, (in && pair) ? off : PSD_WL2_OOR, 4u * u * cstep, x0, x1);
            if (anytail) psd_wl2_st8(rs, (in && tail) ? off : PSD_WL2_OOR, 4u * u * cstep, x0);
        }
    }
    PSD_WL2_WAVE_FENCE();
}

// 64 lines x S elements between memory and the wave's LDS tile (pitch LD), generic path: lane = line
PSD_D void psd_wl2_tile_move(bool store, bool rowsrole, const psd_mat<double>& Mx, double* tile, int LD, int plo, int S, int l0, int nl, int lane) {
    if (lane >= nl) return;
    for (int u = 0; u < S; ++u) {
        double& g = rowsrole ? Mx(plo + u, l0 + lane) : Mx(l0 + lane, plo + u);
        if (store) g = tile[lane * LD + u];
        else tile[lane * LD + u] = g;
    }
}

// lists that are not monotone (2x2 deflation passes): record by record in the LDS tile, 64 lines at a time
PSD_D void psd_wl2_generic_tile(bool rowsrole, const psd_mat<double>& Mx, double* tile, int LD, int plo, int S, int l0, int nl, int lane,
                                const psd_tr* ltr, int cnt) {
    psd_wl2_tile_move(false, rowsrole, Mx, tile, LD, plo, S, l0, nl, lane);
    PSD_WL2_WAVE_FENCE();
    if (lane < nl) {
        double* L = tile + lane * LD;
        for (int e = 0; e < cnt; ++e) {
            const psd_tr tr = ltr[e];
            const int r = tr.pos - plo;
            const int len = psd_tr_len(tr);
            double a1 = L[r], a2 = L[r + 1], a3 = (len == 3) ? L[r + 2] : 0.0;
            psd_tr_apply(tr, a1, a2, a3);
            L[r] = a1;
            L[r + 1] = a2;
            if (len == 3) L[r + 2] = a3;
        }
    }
    PSD_WL2_WAVE_FENCE();
    psd_wl2_tile_move(true, rowsrole, Mx, tile, LD, plo, S, l0, nl, lane);
    PSD_WL2_WAVE_FENCE();
}

// The owner's list into this wavefront's LDS copy (two sentinel records behind its end), classified: +1 positions
// ascend, -1 descend, 0 neither.  One record per lane (cnt <= 64), no workgroup barrier.
PSD_D int psd_wl2_stage_wave(const psd_tr* gtr, int cnt, psd_tr* ltr, int lane) {
    psd_tr tr;
    if (lane < cnt) {
        tr = gtr[lane];
    } else {
        tr.pos = 0x3fffffff;
        tr.kind = PSD_TR_G;
        tr.c0 = 1.0;
        tr.c1 = tr.c2 = 0.0;
    }
    ltr[lane] = tr;
    if (lane < PSD_TR_LDS_RECS - 64) {
        psd_tr se = tr;
        se.pos = 0x3fffffff;
        se.kind = PSD_TR_G;
        se.c0 = 1.0;
        se.c1 = se.c2 = 0.0;
        ltr[64 + lane] = se;
    }
    const int nxt = __shfl_down(tr.pos, 1, 64);
    const bool in = lane + 1 < cnt;
    const unsigned long long down = __ballot(in && nxt < tr.pos), up = __ballot(in && nxt > tr.pos);
    PSD_WL2_WAVE_FENCE();
    return (down == 0ull) ? 1 : ((up == 0ull) ? -1 : 0);
}

// Same contract as psd_rq_apply_wl (passes, modes, zlo..zhi: see there).  SP: the largest window span the launch can
// meet (17 or 32, from the window width the LDS of the chase kernel is laid out for); WPE: waves per SIMD the register
// allocation is held to.  The unit of work is the WAVEFRONT: each of the four waves of a workgroup takes its own items
// (up to `tpi` tiles of one (slot, owner, role)), stages the owner's list in its own LDS copy and never waits for the
// others — the workgroup only shares the item table.  grid: any; 256 threads.
template <int SP, int WPE>
__global__ void __launch_bounds__(PSD_WL2_NT, WPE) psd_rq_apply_wl2(psd_rparams P, int n, int p, int cstride, int pass, int M, int zlo, int zhi,
                                                               int mode) {
    extern __shared__ __attribute__((aligned(16))) char psd_lds[];
    const int tid = (int)threadIdx.x, lane = tid & 63, wv = PSD_WL2_UNI(tid >> 6);
    constexpr size_t WAVE_BYTES = PSD_TR_LDS_BYTES + (size_t)64 * PSD_WL2_LD(SP) * sizeof(double);
    psd_tr* ltr = (psd_tr*)(psd_lds + (size_t)wv * WAVE_BYTES);
    double* tile = (double*)(psd_lds + (size_t)wv * WAVE_BYTES + PSD_TR_LDS_BYTES);
    int* ioff = (int*)(psd_lds + (size_t)PSD_WL2_WAVES * WAVE_BYTES);  // [M + 1] item offsets, [M] items of role A, [M] of role B
    int* tA = ioff + PSD_SLOTS + 2;
    int* tB = tA + PSD_SLOTS;
    int* tpi_ = tB + PSD_SLOTS;
    const int TLA = (pass == 0) ? 64 : 128;  // lines per tile of role A (pass 0: rows role; pass 1: column role)
    const int TLB = 128;                      // role B: the Schur vectors
    const int workers = (int)gridDim.x * PSD_WL2_WAVES;
    if (wv == 0) {
        // item table (lane b = slot b): an item = up to `tpi` tiles of one (slot, owner, role); tpi adapts to the tick —
        // a tick of one small window must spread over the chip, a tick of sixty windows must not stage a list per tile
        const int b = lane;
        psd_apply_desc d;
        d.active = 0;
        if (b < M) d = P.desc[b];
        int la = 0, lz = 0;
        if (b < M && d.active) {
            psd_wl_ranges(d, mode, d.cut);
            if (pass == 0) {
                la = (d.lc1 >= d.lc0) ? (d.lc1 - d.lc0 + 1) : 0;
                lz = (d.zr1 >= d.zr0) ? (d.zr1 - d.zr0 + 1) : 0;
            } else {
                la = (d.rr1 >= d.rr0) ? (d.rr1 - d.rr0 + 1) : 0;
            }
        }
        int tl = p * ((la + TLA - 1) / TLA + (lz + TLB - 1) / TLB);
#pragma unroll
        for (int sft = 32; sft > 0; sft >>= 1) tl += __shfl_xor(tl, sft, 64);
        int tpi = 1;
        while (tpi < PSD_WL2_TPI && tl >= 3 * (2 * tpi) * workers) tpi *= 2;
        const int a = (la + TLA * tpi - 1) / (TLA * tpi), z = (lz + TLB * tpi - 1) / (TLB * tpi);
        const int mine = p * (a + z);
        int incl = mine;
#pragma unroll
        for (int sft = 1; sft < 64; sft <<= 1) {
            const int up = __shfl_up(incl, sft, 64);
            if (b >= sft) incl += up;
        }
        if (b < M) {
            tA[b] = a;
            tB[b] = z;
            ioff[b] = incl - mine;
        }
        if (b == 63) {
            ioff[M] = incl;
            tpi_[0] = tpi;
        }
    }
    __syncthreads();
    const int total = PSD_WL2_UNI(ioff[M]), tpi = PSD_WL2_UNI(tpi_[0]);
    for (int item = (int)blockIdx.x * PSD_WL2_WAVES + wv; item < total; item += workers) {
        int b = 0;
        {
            int lo_ = 0, hi_ = M - 1;
            while (lo_ < hi_) {
                const int mid = (lo_ + hi_ + 1) >> 1;
                if (ioff[mid] <= item) lo_ = mid;
                else hi_ = mid - 1;
            }
            b = PSD_WL2_UNI(lo_);
        }
        const int nA = PSD_WL2_UNI(tA[b]);
        const int per = nA + PSD_WL2_UNI(tB[b]);
        const int q = item - PSD_WL2_UNI(ioff[b]);
        const int m = q / per + 1, tt = q - (m - 1) * per;
        const int role = (pass == 0) ? ((tt < nA) ? 0 : 2) : 1;
        const int gix = (role == 2) ? (tt - nA) : tt;
        if (role == 2 && (m < zlo || m > zhi)) continue;
        psd_apply_desc d = P.desc[b];
        psd_wl_ranges(d, mode, d.cut);
        // (every lane read the same words; say so, or every address derived from them lives in vector registers)
        d.prob = PSD_WL2_UNI(d.prob); d.plo = PSD_WL2_UNI(d.plo); d.phi = PSD_WL2_UNI(d.phi);
        d.lc0 = PSD_WL2_UNI(d.lc0); d.lc1 = PSD_WL2_UNI(d.lc1); d.rr0 = PSD_WL2_UNI(d.rr0); d.rr1 = PSD_WL2_UNI(d.rr1);
        d.zr0 = PSD_WL2_UNI(d.zr0); d.zr1 = PSD_WL2_UNI(d.zr1);
        int cnt = PSD_WL2_UNI(P.cnt[(size_t)b * cstride + (m - 1)]);
        if (cnt > PSD_TR_CAP) cnt = PSD_TR_CAP;
        if (cnt <= 0) continue;
        const psd_tr* gtr = P.tr + ((size_t)b * p + (m - 1)) * PSD_TR_CAP;
        const int S = d.phi - d.plo + 1;
        if (S > SP) {  // (a window's span is at most the width this kernel was chosen for: fail loudly, never skip an update)
            if (lane == 0 && P.gl != nullptr) {
                psd_atomic_store(&P.gl->info, PSD_LIST_OVERFLOW);
                psd_atomic_store(&P.gl->abort, 1);
                psd_atomic_store(&P.gl->done, 1);
            }
            continue;
        }
        int lo, hi, jm;
        double* base;
        if (role == 0) {
            lo = d.lc0; hi = d.lc1; base = P.H; jm = m;
        } else if (role == 1) {
            lo = d.rr0; hi = d.rr1; base = P.H; jm = (m == 1) ? p : (m - 1);
        } else {
            lo = d.zr0; hi = d.zr1; base = P.Z; jm = m;
        }
        const int TL = (role == 0) ? 64 : 128;
        const int GL = TL * tpi;
        const int g0 = lo + gix * GL;
        const int gl = (hi - g0 + 1 < GL) ? (hi - g0 + 1) : GL;
        const psd_mat<double> Mx = psd_mat<double>{base + ((size_t)d.prob * p + (jm - 1)) * n * n, n};
        PSD_WL2_WAVE_FENCE();  // (this wave's reads of the previous item's list are done)
        const int order = PSD_WL2_UNI(psd_wl2_stage_wave(gtr, cnt, ltr, lane));
        if (order != 0) {
            const int ns = (gl + TL - 1) / TL;
            for (int k = 0; k < ns; ++k) {
                const int l0 = g0 + k * TL;
                const int nl = (gl - k * TL < TL) ? (gl - k * TL) : TL;
                // (one instantiation per direction and tile kind: joined behind a branch, the two directions' copies of
                //  the line registers do not share an allocation)
                if (role == 0) {
                    if (order > 0) psd_wl2_rows_tile<SP, true>(Mx, tile, d.plo, S, l0, nl, lane, ltr);
                    else psd_wl2_rows_tile<SP, false>(Mx, tile, d.plo, S, l0, nl, lane, ltr);
                } else if (nl == 128) {
                    if (order > 0) psd_wl2_cols_tile<SP, true, true>(Mx, d.plo, S, l0, nl, lane, ltr);
                    else psd_wl2_cols_tile<SP, true, false>(Mx, d.plo, S, l0, nl, lane, ltr);
                } else if (nl > 1 || l0 > 1) {
                    if (order > 0) psd_wl2_cols_tile<SP, false, true>(Mx, d.plo, S, l0, nl, lane, ltr);
                    else psd_wl2_cols_tile<SP, false, false>(Mx, d.plo, S, l0, nl, lane, ltr);
                } else {
                    psd_wl2_generic_tile(false, Mx, tile, PSD_WL2_LD(SP), d.plo, S, l0, nl, lane, ltr, cnt);
                }
            }
        } else {
            const int ns = (gl + 63) / 64;
            for (int k = 0; k < ns; ++k) {
                const int l0 = g0 + k * 64;
                const int nl = (gl - k * 64 < 64) ? (gl - k * 64) : 64;
                psd_wl2_generic_tile(role == 0, Mx, tile, PSD_WL2_LD(SP), d.plo, S, l0, nl, lane, ltr, cnt);
            }
        }
    }
}

#endif
