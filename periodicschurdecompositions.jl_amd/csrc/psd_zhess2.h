// Periodic Hessenberg-triangular reduction, look-ahead form, ComplexF64 (p >= 3): ONE launch per chain link.
//
// The complex counterpart of psd_hess2.h (same structure, same ring, same two-stream driver): replaces the loop body of
// phessenberg!(A) for complex eltype — /root/reference/src/PeriodicSchurDecompositions.jl:229-247 — with the complex
// reflector of /root/reference/src/householder.jl:110-156 (zlarfg: beta real, tau complex, H = I - tau [1;v][1;v]^H) and
// lmul!(H', A) / rmul!(A, H) of :207-237.  Round 1's form (psd_zhess.h) ran two dependent launches per link — the
// reflector, then BOTH panel updates — and made the chain wait for the whole panel: 5.5 + 16.6 us per link at n = 1024.
//
//   chain#q  (blocks [0, nC): strips of CR rows of M_{q+1}, the next link's matrix): forms v_q from the staged column and
//            the partial norms of the previous launch, w = M_{q+1}[rows, r0_q:n] v_q (the block is only READ), the next
//            link's column  M_{q+1}[rows, r0_q] - tau_q w  and its partial norms.  Block 0 stores v_q LAPACK-style.
//   bulk     M <- H(v_Lb)^H (M H(v_La)) in ONE pass per matrix (psd_zhess2_bulk: K links per launch, second stream):
//            right: a[r, c] -= (tau_R w_r) conj(v_c);  left: z_c = conj(tau_L) sum_r conj(v_r) a[r, c], a[r, c] -= z_c v_r.
// A strip of 4 rows is 64 bytes of every column (the real form's is 32).
#pragma once
#include "psd_complex.h"
#include "psd_hess2.h"

#ifndef PSD_HOSTSIM
#define PSD_ZH2_NT 256
#define PSD_ZH2_ROWS 4  // rows of a strip above the left reflector: 64 bytes of every column

struct psd_zhess2_args {
    psd_z* H;      // [p][n][n]
    psd_z* tau;    // [p][n]
    double* ring;  // ringmask + 1 slots of psd_zh2_slot_doubles(n)
    int p;
    int ringmask;
    int xcd;       // chain strips that share a 128-byte line on one XCD (CR = 4: two strips per line)
    int pipe;      // 1: consecutive chain launches overlap on two streams, the staged column travels as self-validating records (psd_hess2.h)
    int* err;      // pipe: set when a launch gave up waiting
};
// slot layout (doubles): v[2 (n+8)] | w[2 (n+8)] | col[2 (n+8)] | hdr[8] (tau.re, tau.im, beta) | part[2 (n/4 + 2)] | rec[4 (n+8)]
PSD_HD size_t psd_zh2_slot_doubles(int n) { return 10 * (size_t)(n + 8) + 8 + 2 * (size_t)(n / 4 + 2); }
struct psd_zh2_slot {
    psd_z *v, *w, *col;
    double *hdr, *part;
    unsigned long long* rec;  // pipe mode: entry k as (bits(re), bits(re) ^ tag, bits(im), bits(im) ^ tag)
};
PSD_D psd_zh2_slot psd_zh2_get(const psd_zhess2_args* G, int n, int q) {
    double* b = G->ring + (size_t)(q & G->ringmask) * psd_zh2_slot_doubles(n);
    psd_zh2_slot s;
    s.v = (psd_z*)b;
    s.w = (psd_z*)(b + 2 * (size_t)(n + 8));
    s.col = (psd_z*)(b + 4 * (size_t)(n + 8));
    s.hdr = b + 6 * (size_t)(n + 8);
    s.part = s.hdr + 8;
    s.rec = (unsigned long long*)(s.part + 2 * (size_t)(n / 4 + 2));
    return s;
}

PSD_D psd_z psd_zh2_fma(psd_z acc, psd_z a, psd_z b) {  // acc + a b
    acc.re += a.re * b.re - a.im * b.im;
    acc.im += a.re * b.im + a.im * b.re;
    return acc;
}
PSD_D psd_z psd_zh2_shfl_xor(psd_z x, int s) { return zmk(__shfl_xor(x.re, s, 64), __shfl_xor(x.im, s, 64)); }
PSD_D psd_z psd_zh2_wave_sum(psd_z x) {
#pragma unroll
    for (int s = 32; s > 0; s >>= 1) x = zadd(x, psd_zh2_shfl_xor(x, s));
    return x;
}

// rows (r, r+1) of column c (0-based); each an element of 16 bytes
PSD_D void psd_zh2_ld2(const psd_z* M, int n, int r, int c, bool ok0, bool ok1, psd_z& x0, psd_z& x1) {
    x0 = x1 = zmk(0.0, 0.0);
    const psd_z* q = M + (size_t)c * n + r;
    if (ok0) x0 = q[0];
    if (ok1) x1 = q[1];
}
PSD_D void psd_zh2_st2(psd_z* M, int n, int r, int c, bool ok0, bool ok1, psd_z x0, psd_z x1) {
    psd_z* q = M + (size_t)c * n + r;
    if (ok0) q[0] = x0;
    if (ok1) q[1] = x1;
}

// householder.jl:110-156 from the tail's norm: tau (complex), beta (real), mult = 1 / (alpha - beta); tau = 0: H = I
PSD_D void psd_zh2_larfg(psd_z alpha, double xnorm, psd_z& tau, double& beta, psd_z& mult) {
    double ar = alpha.re, ai = alpha.im;
    if (xnorm == 0.0 && ai == 0.0) {  // :121-123
        tau = zmk(0.0, 0.0);
        beta = ar;
        mult = zmk(0.0, 0.0);
        return;
    }
    const double sfmin = PSD_DBL_MIN / PSD_DBL_EPS;
    double w = fmax(fabs(ar), fmax(fabs(ai), xnorm));
    beta = -copysign(w * sqrt((ar / w) * (ar / w) + (ai / w) * (ai / w) + (xnorm / w) * (xnorm / w)), ar);
    int kount = 0;
    double acc = 1.0;
    if (fabs(beta) < sfmin) {
        const double rsfmin = 1.0 / sfmin;
        bool smallb = true;
        while (smallb) {
            kount += 1;
            acc *= rsfmin;
            beta *= rsfmin;
            ar *= rsfmin;
            ai *= rsfmin;
            smallb = (fabs(beta) < sfmin) && (kount < 20);
        }
        xnorm *= acc;
        w = fmax(fabs(ar), fmax(fabs(ai), xnorm));
        beta = -copysign(w * sqrt((ar / w) * (ar / w) + (ai / w) * (ai / w) + (xnorm / w) * (xnorm / w)), ar);
    }
    tau = zmk((beta - ar) / beta, -ai / beta);
    mult = zscal(acc, zdiv(zmk(1.0, 0.0), zmk(ar - beta, ai)));
    for (int q = 0; q < kount; ++q) beta *= sfmin;
}

// The deferred update of the matrix of link Lb: M <- H(v_Lb)^H (M H(v_La)) (see psd_h2_bulk_body)
template <int NK>
PSD_D void psd_zh2_bulk_body(const psd_zhess2_args* G, int n, const psd_h2_link Lb, const psd_h2_link La, int slotb, int bb, int nT,
                             psd_z* vs, psd_z* red) {
    const int p = G->p;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    if (!Lb.valid && !La.valid) return;
    const int jb = Lb.valid ? Lb.j : p;
    psd_z* M = G->H + (size_t)(jb - 1) * n * n;
    const psd_zh2_slot SR = psd_zh2_get(G, n, slotb - 1), SL = psd_zh2_get(G, n, slotb);
    const psd_z tauR = La.valid ? zmk(SR.hdr[0], SR.hdr[1]) : zmk(0.0, 0.0);
    const psd_z tauL = Lb.valid ? zmk(SL.hdr[0], SL.hdr[1]) : zmk(0.0, 0.0);
    const int rR = La.valid ? (La.r0 - 1) : 0;
    const int mR = n - rR;
    const int R0 = Lb.valid ? (Lb.r0 - 1) : n;
    if (bb < nT) {
        // rows above the left reflector: fused GEMV + rank-one update, 4-row strips (right reflector only)
        if (ziszero(tauR)) return;
        const int t = bb;
        if (PSD_ZH2_ROWS * t >= R0) return;
        constexpr int NKT = (NK + 1) / 2;  // 128 column lanes
        const int rp = tid & 1, cl = tid >> 1;
        const int r = PSD_ZH2_ROWS * t + 2 * rp;
        const bool ok0 = r < R0, ok1 = r + 1 < R0;
        psd_z a0[NKT], a1[NKT];
#pragma unroll
        for (int k = 0; k < NKT; ++k) {
            const int cc = cl + 128 * k;
            a0[k] = a1[k] = zmk(0.0, 0.0);
            if (cc < mR) psd_zh2_ld2(M, n, r, rR + cc, ok0, ok1, a0[k], a1[k]);
        }
        for (int k = tid; k < mR; k += PSD_ZH2_NT) vs[k] = SR.v[k];
        __syncthreads();
        psd_z acc0 = zmk(0.0, 0.0), acc1 = zmk(0.0, 0.0);
#pragma unroll
        for (int k = 0; k < NKT; ++k) {
            const int cc = cl + 128 * k;
            if (cc < mR) {
                const psd_z vv = vs[cc];
                acc0 = psd_zh2_fma(acc0, a0[k], vv);
                acc1 = psd_zh2_fma(acc1, a1[k], vv);
            }
        }
#pragma unroll
        for (int s = 2; s < 64; s <<= 1) {
            acc0 = zadd(acc0, psd_zh2_shfl_xor(acc0, s));
            acc1 = zadd(acc1, psd_zh2_shfl_xor(acc1, s));
        }
        if (lane < 2) {
            red[(wave * 2 + lane) * 2] = acc0;
            red[(wave * 2 + lane) * 2 + 1] = acc1;
        }
        __syncthreads();
        const int k0 = rp * 2;
        const psd_z w0 = zmul(tauR, zadd(zadd(red[k0], red[4 + k0]), zadd(red[8 + k0], red[12 + k0])));
        const psd_z w1 = zmul(tauR, zadd(zadd(red[k0 + 1], red[4 + k0 + 1]), zadd(red[8 + k0 + 1], red[12 + k0 + 1])));
#pragma unroll
        for (int k = 0; k < NKT; ++k) {
            const int cc = cl + 128 * k;
            if (cc < mR) {
                const psd_z vc = zconj(vs[cc]);
                psd_zh2_st2(M, n, r, rR + cc, ok0, ok1, zsub(a0[k], zmul(w0, vc)), zsub(a1[k], zmul(w1, vc)));
            }
        }
        return;
    }
    // rows R0..n-1: one wavefront per column, the column in registers
    if (!Lb.valid) return;
    if (ziszero(tauR) && ziszero(tauL)) return;
    const int c = Lb.i + 4 * (bb - nT) + wave;
    if (c >= n) return;
    const int mL = n - R0;
    constexpr int NKC = (NK + 1) / 2;
    psd_z a0[NKC], a1[NKC];
    const bool rightc = !ziszero(tauR) && c >= rR;
    const psd_z vc = rightc ? zmul(tauR, zconj(SR.v[c - rR])) : zmk(0.0, 0.0);
    psd_z z = zmk(0.0, 0.0);
    const bool left = !ziszero(tauL);
#pragma unroll
    for (int k = 0; k < NKC; ++k) {
        const int rr = 2 * lane + 128 * k;
        a0[k] = a1[k] = zmk(0.0, 0.0);
        if (rr < mL) {
            const bool ok1 = rr + 1 < mL;
            psd_zh2_ld2(M, n, R0 + rr, c, true, ok1, a0[k], a1[k]);
            if (rightc) {
                a0[k] = zsub(a0[k], zmul(vc, SR.w[R0 + rr]));
                if (ok1) a1[k] = zsub(a1[k], zmul(vc, SR.w[R0 + rr + 1]));
            }
            if (left) {
                z = psd_zh2_fma(z, zconj(SL.v[rr]), a0[k]);
                if (ok1) z = psd_zh2_fma(z, zconj(SL.v[rr + 1]), a1[k]);
            }
        }
    }
    z = zmul(zconj(tauL), psd_zh2_wave_sum(z));
#pragma unroll
    for (int k = 0; k < NKC; ++k) {
        const int rr = 2 * lane + 128 * k;
        if (rr < mL) {
            const bool ok1 = rr + 1 < mL;
            psd_z x0 = a0[k], x1 = a1[k];
            if (left) {
                x0 = zsub(x0, zmul(z, SL.v[rr]));
                if (ok1) x1 = zsub(x1, zmul(z, SL.v[rr + 1]));
            }
            psd_zh2_st2(M, n, R0 + rr, c, true, ok1, x0, x1);
        }
    }
}

// One launch = chain#q (+ bulk B(q-1) in the one-stream form: grid = nC + nT + nB).  See psd_hess2_link.
// LDS: (n + 8 + 2 * NT + 64) complex.
template <int NK, int CR>
__global__ void __launch_bounds__(PSD_ZH2_NT, (NK > 8 ? 2 : 4)) psd_zhess2_link(const psd_zhess2_args Gv, int n, int qi, int qj, int nC, int nT) {
    extern __shared__ __attribute__((aligned(16))) char psd_lds[];
    psd_z* vs = (psd_z*)psd_lds;   // n + 8
    psd_z* red = vs + (n + 8);     // 2 * NT + 64
    double* redd = (double*)red;
    const psd_zhess2_args* G = &Gv;
    const int p = G->p;
    const psd_h2_link L = psd_h2_linkat(qi, qj, 0, n, p);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int b = blockIdx.x;
    if (b < nC) {
        const psd_h2_link Ln = psd_h2_linkat(qi, qj, 1, n, p);
        if (!L.valid && !Ln.valid) return;
        if (!L.valid && qi > 1) return;
        const int q = L.valid ? 0 : -1;
        const int slot = qi * p - qj;
        const psd_zh2_slot S = psd_zh2_get(G, n, slot), Sn = psd_zh2_get(G, n, slot + 1);
        const int r0 = L.r0 - 1;
        const int m = (q >= 0) ? (n - r0) : 0;
        constexpr int RP = CR / 2;
        constexpr int CL = PSD_ZH2_NT / RP;
        constexpr int NKS = (64 * NK + CL - 1) / CL;
        const int ntileC = (n + CR - 1) / CR;
        const psd_z* M = Ln.valid ? (G->H + (size_t)(Ln.j - 1) * n * n) : G->H;
        const int r0n = Ln.r0 - 1;
        const int cfirst = (q >= 0) ? r0 : (Ln.i - 1);
        int t = r0n / CR + (b - 1);
        if (G->xcd) {
            constexpr int GS = (CR < 8) ? 8 / CR : 1;  // (strips per 128-byte line: 16-byte elements)
            const int idx = b - 1, grp0 = (r0n / CR) / GS;
            t = GS * (grp0 + (idx % 8) + 8 * (idx / (8 * GS))) + (idx / 8) % GS;
        }
        const bool strip = Ln.valid && b > 0 && t < ntileC && t >= r0n / CR;
        const int rp = tid % RP, cl = tid / RP;
        const int r = CR * t + 2 * rp;
        const bool ok0 = strip && r < n && r >= r0n, ok1 = strip && r + 1 < n && r + 1 >= r0n;
        psd_z a0[NKS], a1[NKS];
        const int rfin = CR * t + tid;
        const bool fin = strip && tid < CR && rfin < n && rfin >= r0n;
        constexpr int NV = (64 * NK + PSD_ZH2_NT - 1) / PSD_ZH2_NT;
        psd_z xcol[NV];
        psd_z tau = zmk(0.0, 0.0), mult = zmk(0.0, 0.0), alpha = zmk(0.0, 0.0);
        double beta = 0.0, am = 0.0, sq = 0.0;
        const bool pipe = G->pipe != 0;
        if (q >= 0 && !pipe) {
            const int np_ = ntileC - r0 / CR;
            constexpr int NPP = (64 * NK / CR + PSD_ZH2_NT - 1) / PSD_ZH2_NT;  // (amax, ssq) pairs per thread
#pragma unroll
            for (int u = 0; u < NPP; ++u) {
                const int k = tid + PSD_ZH2_NT * u;
                if (k < np_) {
                    const double am2 = S.part[2 * k], sq2 = S.part[2 * k + 1];
                    const double mx = fmax(am, am2);
                    if (mx > 0.0) {
                        const double f1 = am / mx, f2 = am2 / mx;
                        sq = sq * (f1 * f1) + sq2 * (f2 * f2);
                    }
                    am = mx;
                }
            }
            alpha = S.col[r0];
#pragma unroll
            for (int u = 0; u < NV; ++u) {
                const int k = tid + PSD_ZH2_NT * u;
                xcol[u] = (k < m) ? S.col[r0 + k] : zmk(0.0, 0.0);
            }
        }
        const psd_z mfirst = fin ? M[(size_t)cfirst * n + rfin] : zmk(0.0, 0.0);
        if (q >= 0) {
#pragma unroll
            for (int k = 0; k < NKS; ++k) {
                const int cc = cl + CL * k;
                a0[k] = a1[k] = zmk(0.0, 0.0);
                if (cc < m) psd_zh2_ld2(M, n, r, r0 + cc, ok0, ok1, a0[k], a1[k]);
            }
        }
        if (q >= 0 && pipe) {
            // the strip is on its way; now the column of the previous launch, which may still be running (psd_hess2.h:
            // every double as a record (bits, bits ^ tag), polled with 16-byte coherent loads until all of them validate)
            const unsigned long long tag = psd_h2_tag(slot);
            const unsigned long long* rec = S.rec + 4 * (size_t)r0;
            const __amdgpu_buffer_rsrc_t rsr = __builtin_amdgcn_make_buffer_rsrc((void*)rec, 0, m * 32, 0x00020000);
            unsigned long long xr[NV], xi[NV];
            int spins = 0;
            long long wait_t0 = 0;
            for (;;) {
                bool okr = true;
#pragma unroll
                for (int u = 0; u < NV; ++u) {
                    const int k = tid + PSD_ZH2_NT * u;
                    xr[u] = xi[u] = 0;
                    if (PSD_ZH2_NT * u < m) {
                        const unsigned off = (k < m) ? (unsigned)k * 32u : 0xffffffe0u;
                        const psd_h2_u4 t4 = __builtin_amdgcn_raw_buffer_load_b128(rsr, off, 0, 16);
                        const psd_h2_u4 s4 = __builtin_amdgcn_raw_buffer_load_b128(rsr, off, 16, 16);
                        const unsigned long long x = ((unsigned long long)t4.y << 32) | t4.x, cx = ((unsigned long long)t4.w << 32) | t4.z;
                        const unsigned long long y = ((unsigned long long)s4.y << 32) | s4.x, cy = ((unsigned long long)s4.w << 32) | s4.z;
                        okr = okr && (k >= m || ((x ^ cx) == tag && (y ^ cy) == tag));
                        xr[u] = x;
                        xi[u] = y;
                    }
                }
                asm volatile("" ::: "memory");
                const bool giveup = __hip_atomic_load(G->err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0;
                if (__syncthreads_and((okr || giveup) ? 1 : 0)) break;
                if (psd_h2_wait_expired(spins, wait_t0)) {  // (every wave reaches an exit; the results are then void and the host says so)
                    if (tid == 0) __hip_atomic_store(G->err, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                }
                __builtin_amdgcn_s_sleep(2);
            }
            // this thread's share of the tail (entries k >= 1): largest |re|, |im| and PLAIN sum of squares; the raw column
            // goes to LDS with its first entry as zero (w = M v = M[:, 0] + mult (M[:, 1:] x[1:]))
            am = 0.0;
            sq = 0.0;
#pragma unroll
            for (int u = 0; u < NV; ++u) {
                const int k = tid + PSD_ZH2_NT * u;
                xcol[u] = (k < m) ? zmk(__longlong_as_double((long long)xr[u]), __longlong_as_double((long long)xi[u])) : zmk(0.0, 0.0);
                if (k >= 1 && k < m) {
                    am = fmax(am, zabs1(xcol[u]));
                    sq = __builtin_fma(xcol[u].re, xcol[u].re, sq);
                    sq = __builtin_fma(xcol[u].im, xcol[u].im, sq);
                }
                if (k < m) vs[k] = (k == 0) ? zmk(0.0, 0.0) : xcol[u];
            }
        }
        double xnorm = 0.0;
        if (q >= 0) {
            if (pipe) {
                const double amw = psd_h2_wave_max(am);
                const double s2w = psd_h2_wave_sum(sq);
                if (lane == 0) {
                    redd[2 * wave] = amw;
                    redd[2 * wave + 1] = s2w;
                }
                if (tid == 0) {
                    redd[8] = xcol[0].re;
                    redd[9] = xcol[0].im;
                }
                __syncthreads();
                alpha = zmk(redd[8], redd[9]);
                const double amax = fmax(fmax(redd[0], redd[2]), fmax(redd[4], redd[6]));
                const double s2 = (redd[1] + redd[3]) + (redd[5] + redd[7]);
                if (m > 1 && amax > 0.0) {
                    if (amax < 1e140 && amax > 1e-140) {
                        double g, rg;
                        psd_sqrt_pair_fast(s2, g, rg);
                        xnorm = g;
                    } else {
                        __syncthreads();
                        double ss = 0.0;
#pragma unroll
                        for (int u = 0; u < NV; ++u) {
                            const int k = tid + PSD_ZH2_NT * u;
                            if (k >= 1 && k < m) {
                                const double zr = xcol[u].re / amax, zi = xcol[u].im / amax;
                                ss += zr * zr + zi * zi;
                            }
                        }
                        ss = psd_h2_wave_sum(ss);
                        if (lane == 0) redd[16 + wave] = ss;
                        __syncthreads();
                        xnorm = amax * sqrt((redd[16] + redd[17]) + (redd[18] + redd[19]));
                    }
                }
            } else {
                const double amw = psd_h2_wave_max(am);
                double ssw = 0.0;
                if (amw > 0.0) {
                    const double f = am / amw;
                    ssw = sq * (f * f);
                }
                ssw = psd_h2_wave_sum(ssw);
                if (lane == 0) {
                    redd[2 * wave] = amw;
                    redd[2 * wave + 1] = ssw;
                }
                __syncthreads();
                const double amax = fmax(fmax(redd[0], redd[2]), fmax(redd[4], redd[6]));
                double tot = 0.0;
                if (amax > 0.0) {
#pragma unroll
                    for (int wv = 0; wv < 4; ++wv) {
                        const double f = redd[2 * wv] / amax;
                        tot += redd[2 * wv + 1] * (f * f);
                    }
                }
                xnorm = (m > 1) ? amax * sqrt(tot) : 0.0;
            }
            if (!pipe || b == 0) psd_zh2_larfg(alpha, xnorm, tau, beta, mult);  // (pipe: the strip workgroups form the scalars behind their GEMV)
            if (!pipe) {
#pragma unroll
                for (int u = 0; u < NV; ++u) {
                    const int k = tid + PSD_ZH2_NT * u;
                    if (k < m) vs[k] = (k == 0) ? zmk(1.0, 0.0) : zmul(xcol[u], mult);
                }
                __syncthreads();
            }
            if (b == 0) {  // publish v_q, store it LAPACK-style (PSD.jl:232-236,241-244)
                psd_z* Mq = G->H + (size_t)(L.j - 1) * n * n;
                const int c = L.i - 1;
                const bool nz = !ziszero(tau);
#pragma unroll
                for (int u = 0; u < NV; ++u) {
                    const int k = tid + PSD_ZH2_NT * u;
                    if (k < m) {
                        const psd_z vk = (k == 0) ? zmk(1.0, 0.0) : zmul(xcol[u], mult);
                        S.v[k] = vk;
                        Mq[(size_t)c * n + r0 + k] = nz ? ((k == 0) ? zmk(beta, 0.0) : vk) : xcol[u];
                    }
                }
                if (tid == 0) {
                    S.hdr[0] = tau.re;
                    S.hdr[1] = tau.im;
                    S.hdr[2] = beta;
                    G->tau[(size_t)(L.j - 1) * n + (L.i - 1)] = tau;
                }
            }
        }
        if (!strip) return;
        psd_z acc0 = zmk(0.0, 0.0), acc1 = zmk(0.0, 0.0);
        if (q >= 0 && (pipe || !ziszero(tau))) {
#pragma unroll
            for (int k = 0; k < NKS; ++k) {
                const int cc = cl + CL * k;
                if (cc < m) {
                    const psd_z vv = vs[cc];
                    acc0 = psd_zh2_fma(acc0, a0[k], vv);
                    acc1 = psd_zh2_fma(acc1, a1[k], vv);
                }
            }
        }
        psd_z* const redg = pipe ? (red + 64) : red;  // (pipe: workgroup mates may still be reading the norm's exchange area)
#pragma unroll
        for (int sft = RP; sft < 64; sft <<= 1) {
            acc0 = zadd(acc0, psd_zh2_shfl_xor(acc0, sft));
            acc1 = zadd(acc1, psd_zh2_shfl_xor(acc1, sft));
        }
        if (lane < RP) {
            redg[(wave * RP + lane) * 2] = acc0;
            redg[(wave * RP + lane) * 2 + 1] = acc1;
        }
        if (pipe && q >= 0) psd_zh2_larfg(alpha, xnorm, tau, beta, mult);
        __syncthreads();
        double amt = 0.0;
        psd_z y = zmk(0.0, 0.0);
        if (fin) {
            psd_z w = zadd(zadd(redg[tid], redg[CR + tid]), zadd(redg[2 * CR + tid], redg[3 * CR + tid]));
            if (pipe && q >= 0) w = ziszero(tau) ? zmk(0.0, 0.0) : zadd(mfirst, zmul(mult, w));  // (the unit entry of v times column r0)
            y = zsub(mfirst, zmul(tau, w));
            if (pipe) {
                const unsigned long long tg = psd_h2_tag(slot + 1);
                const unsigned long long br = (unsigned long long)__double_as_longlong(y.re), bi = (unsigned long long)__double_as_longlong(y.im);
                unsigned long long* q4 = Sn.rec + 4 * (size_t)rfin;
                __hip_atomic_store(q4, br, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                __hip_atomic_store(q4 + 1, br ^ tg, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                __hip_atomic_store(q4 + 2, bi, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                __hip_atomic_store(q4 + 3, bi ^ tg, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            } else {
                Sn.col[rfin] = y;
            }
            S.w[rfin] = w;
            if (rfin > r0n) amt = zabs1(y);
            else y = zmk(0.0, 0.0);
        }
        if (wave == 0 && !pipe) {
            const double amax = psd_h2_wave_max(amt);
            double s2 = 0.0;
            if (amax > 0.0 && amt > 0.0) {
                const double zr = y.re / amax, zi = y.im / amax;
                s2 = zr * zr + zi * zi;
            }
            s2 = psd_h2_wave_sum(s2);
            if (lane == 0) {
                Sn.part[2 * (t - r0n / CR)] = amax;
                Sn.part[2 * (t - r0n / CR) + 1] = s2;
            }
        }
        return;
    }
    psd_zh2_bulk_body<NK>(G, n, psd_h2_linkat(qi, qj, -1, n, p), psd_h2_linkat(qi, qj, -2, n, p), qi * p - qj - 1, b - nC, nT, vs, red);
}

template <int NK>
__global__ void __launch_bounds__(PSD_ZH2_NT, (NK > 8 ? 2 : 4)) psd_zhess2_bulk(const psd_zhess2_args Gv, int n, int idx0, int nT) {
    extern __shared__ __attribute__((aligned(16))) char psd_lds[];
    psd_z* vs = (psd_z*)psd_lds;
    psd_z* red = vs + (n + 8);
    const int p = Gv.p;
    const int idx = idx0 + (int)blockIdx.y;
    const int Q = (n - 1) * p;
    if (idx > Q) return;
    psd_h2_link Lb, La;
    Lb.i = idx / p + 1;
    Lb.j = p - idx % p;
    Lb.valid = (idx < Q) ? 1 : 0;
    Lb.r0 = (Lb.j == 1) ? (Lb.i + 1) : Lb.i;
    La = psd_h2_linkat(Lb.i, Lb.j, -1, n, p);
    psd_zh2_bulk_body<NK>(&Gv, n, Lb, La, idx, (int)blockIdx.x, nT, vs, red);
}

#endif
