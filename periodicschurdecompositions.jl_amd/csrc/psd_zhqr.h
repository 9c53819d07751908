// Eigenvalues of a small complex upper Hessenberg matrix (N <= PSD_ZHQR_MAX) by the explicitly shifted QR iteration with
// Wilkinson shifts and Givens rotations; eigenvalues only, h (row-major, leading dimension LD) is destroyed.  One lane
// turns the trailing block of the periodic product into the shifts of a multishift train with it.
#pragma once
#include "psd_complex.h"

#define PSD_ZHQR_MAX 16

// principal square root of a complex number
PSD_HD psd_z psd_zsqrt(psd_z a) {
    const double m = zabs(a);
    if (m == 0.0) return zmk(0.0, 0.0);
    const double re = sqrt(0.5 * (m + fabs(a.re)));
    const double im = 0.5 * a.im / re;
    return (a.re >= 0.0) ? zmk(re, im) : zmk(fabs(im), copysign(re, a.im));
}

PSD_HD bool psd_zhqr(psd_z* h, int N, int LD, psd_z* w) {
#define PSD_ZH(r, c) h[(r) * LD + (c)]
    const double eps = PSD_DBL_EPS;
    double cs[PSD_ZHQR_MAX];
    psd_z sn[PSD_ZHQR_MAX];
    int en = N - 1;
    while (en >= 0) {
        int its = 0;
        for (;;) {
            int l;
            for (l = en; l >= 1; --l) {
                const double s = zabs1(PSD_ZH(l - 1, l - 1)) + zabs1(PSD_ZH(l, l));
                if (zabs1(PSD_ZH(l, l - 1)) <= eps * s) break;
            }
            if (l == en) {
                w[en] = PSD_ZH(en, en);
                en -= 1;
                break;
            }
            if (its == 60) return false;
            // Wilkinson shift: the eigenvalue of the trailing 2x2 block closer to its last diagonal entry
            const psd_z a = PSD_ZH(en - 1, en - 1), b = PSD_ZH(en - 1, en), c = PSD_ZH(en, en - 1), d = PSD_ZH(en, en);
            psd_z mu;
            if (its == 10 || its == 20) {
                mu = zmk(zabs1(c) + ((en >= 2) ? zabs1(PSD_ZH(en - 1, en - 2)) : 0.0) + d.re, d.im);
            } else {
                const psd_z hd = zscal(0.5, zsub(a, d));
                const psd_z disc = psd_zsqrt(zadd(zmul(hd, hd), zmul(b, c)));
                const psd_z mid = zscal(0.5, zadd(a, d));
                const psd_z m1 = zadd(mid, disc), m2 = zsub(mid, disc);
                mu = (zabs1(zsub(m1, d)) <= zabs1(zsub(m2, d))) ? m1 : m2;
            }
            ++its;
            for (int k = l; k <= en; ++k) PSD_ZH(k, k) = zsub(PSD_ZH(k, k), mu);
            for (int k = l; k < en; ++k) {  // H - mu I = Q R
                psd_z r;
                psd_zgivens(PSD_ZH(k, k), PSD_ZH(k + 1, k), cs[k], sn[k], r);
                PSD_ZH(k, k) = r;
                PSD_ZH(k + 1, k) = zmk(0.0, 0.0);
                for (int j = k + 1; j <= en; ++j) psd_zrot_left(cs[k], sn[k], PSD_ZH(k, j), PSD_ZH(k + 1, j));
            }
            for (int k = l; k < en; ++k) {  // R Q
                const int top = (k + 1 < en) ? (k + 1) : en;
                for (int i = l; i <= top; ++i) psd_zrot_right_adj(cs[k], sn[k], PSD_ZH(i, k), PSD_ZH(i, k + 1));
            }
            for (int k = l; k <= en; ++k) PSD_ZH(k, k) = zadd(PSD_ZH(k, k), mu);
        }
    }
    return true;
#undef PSD_ZH
}

#ifndef PSD_HOSTSIM
// The same iteration run by a whole wavefront on a matrix in LDS (as psd_hqr_wave of the real engine): every lane follows
// the uniform control flow and computes the scalars redundantly from broadcast reads; the row rotations of the QR step
// are one lane per column, the column rotations one lane per row, the shift of the diagonal one lane per entry.
#define PSD_ZHQW_SYNC() asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory")
__device__ __forceinline__ bool psd_zhqr_wave(psd_z* h, int N, int LD, psd_z* w, int lane) {
#define PSD_ZH(r, c) h[(r) * LD + (c)]
    const double eps = PSD_DBL_EPS;
    int en = N - 1;
    PSD_ZHQW_SYNC();
    while (en >= 0) {
        int its = 0;
        for (;;) {
            PSD_ZHQW_SYNC();
            int l;
            for (l = en; l >= 1; --l) {
                const double s = zabs1(PSD_ZH(l - 1, l - 1)) + zabs1(PSD_ZH(l, l));
                if (zabs1(PSD_ZH(l, l - 1)) <= eps * s) break;
            }
            if (l == en) {
                if (lane == 0) w[en] = PSD_ZH(en, en);
                en -= 1;
                break;
            }
            if (its == 60) return false;
            const psd_z a = PSD_ZH(en - 1, en - 1), b = PSD_ZH(en - 1, en), c = PSD_ZH(en, en - 1), d = PSD_ZH(en, en);
            psd_z mu;
            if (its == 10 || its == 20) {
                mu = zmk(zabs1(c) + ((en >= 2) ? zabs1(PSD_ZH(en - 1, en - 2)) : 0.0) + d.re, d.im);
            } else {
                const psd_z hd = zscal(0.5, zsub(a, d));
                const psd_z disc = psd_zsqrt(zadd(zmul(hd, hd), zmul(b, c)));
                const psd_z mid = zscal(0.5, zadd(a, d));
                const psd_z m1 = zadd(mid, disc), m2 = zsub(mid, disc);
                mu = (zabs1(zsub(m1, d)) <= zabs1(zsub(m2, d))) ? m1 : m2;
            }
            ++its;
            PSD_ZHQW_SYNC();
            {
                const int k = l + lane;
                if (k <= en) PSD_ZH(k, k) = zsub(PSD_ZH(k, k), mu);
            }
            PSD_ZHQW_SYNC();
            double cs[PSD_ZHQR_MAX];
            psd_z sn[PSD_ZHQR_MAX];
#pragma unroll 1
            for (int k = l; k < en; ++k) {  // H - mu I = Q R
                psd_z r;
                double ck;
                psd_z sk;
                psd_zgivens(PSD_ZH(k, k), PSD_ZH(k + 1, k), ck, sk, r);
                cs[k] = ck;
                sn[k] = sk;
                PSD_ZHQW_SYNC();  // (every lane has read the two entries rewritten below)
                if (lane == 0) {
                    PSD_ZH(k, k) = r;
                    PSD_ZH(k + 1, k) = zmk(0.0, 0.0);
                }
                {
                    const int j = k + 1 + lane;
                    if (j <= en) psd_zrot_left(ck, sk, PSD_ZH(k, j), PSD_ZH(k + 1, j));
                }
                PSD_ZHQW_SYNC();
            }
#pragma unroll 1
            for (int k = l; k < en; ++k) {  // R Q
                const int top = (k + 1 < en) ? (k + 1) : en;
                const int i = l + lane;
                if (i <= top) psd_zrot_right_adj(cs[k], sn[k], PSD_ZH(i, k), PSD_ZH(i, k + 1));
                PSD_ZHQW_SYNC();
            }
            {
                const int k = l + lane;
                if (k <= en) PSD_ZH(k, k) = zadd(PSD_ZH(k, k), mu);
            }
        }
    }
    PSD_ZHQW_SYNC();
    return true;
#undef PSD_ZH
}
#endif

