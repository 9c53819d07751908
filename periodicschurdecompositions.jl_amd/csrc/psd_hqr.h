// Eigenvalues of a small real upper Hessenberg matrix (N <= PSD_HQR_MAX) by the Francis double-shift QR iteration
// (the classical EISPACK `hqr` organisation: deflation test, two-consecutive-small-subdiagonals start, exceptional shifts
// at iterations 10 and 20), eigenvalues only.  Used on the device by ONE lane to turn the trailing block of the
// periodic product into the shifts of a multishift train; h is destroyed.
#pragma once
#include "psd_scalar.h"

#define PSD_HQR_MAX 16

PSD_HD bool psd_hqr(double* h, int N, int LD, double* wr, double* wi) {
#define PSD_HQ(r, c) h[(r) * LD + (c)]
    const double eps = PSD_DBL_EPS;
    double norm = 0.0;
    for (int i = 0; i < N; ++i)
        for (int j = (i > 0 ? i - 1 : 0); j < N; ++j) norm += fabs(PSD_HQ(i, j));
    int en = N - 1;
    double t = 0.0;
    while (en >= 0) {
        int its = 0;
        for (;;) {
            int l;
            for (l = en; l >= 1; --l) {
                double s = fabs(PSD_HQ(l - 1, l - 1)) + fabs(PSD_HQ(l, l));
                if (s == 0.0) s = norm;
                if (fabs(PSD_HQ(l, l - 1)) <= eps * s) break;
            }
            double x = PSD_HQ(en, en);
            if (l == en) {  // one root
                wr[en] = x + t;
                wi[en] = 0.0;
                en -= 1;
                break;
            }
            double y = PSD_HQ(en - 1, en - 1);
            double w = PSD_HQ(en, en - 1) * PSD_HQ(en - 1, en);
            if (l == en - 1) {  // two roots
                const double pp = 0.5 * (y - x), qq = pp * pp + w;
                double z = sqrt(fabs(qq));
                x += t;
                if (qq >= 0.0) {
                    z = pp + copysign(z, pp);
                    wr[en - 1] = wr[en] = x + z;
                    if (z != 0.0) wr[en] = x - w / z;
                    wi[en - 1] = wi[en] = 0.0;
                } else {
                    wr[en - 1] = wr[en] = x + pp;
                    wi[en - 1] = z;
                    wi[en] = -z;
                }
                en -= 2;
                break;
            }
            if (its == 60) return false;
            if (its == 10 || its == 20) {  // exceptional shift
                t += x;
                for (int i = 0; i <= en; ++i) PSD_HQ(i, i) -= x;
                const double s = fabs(PSD_HQ(en, en - 1)) + fabs(PSD_HQ(en - 1, en - 2));
                x = y = 0.75 * s;
                w = -0.4375 * s * s;
            }
            ++its;
            int m;
            double p = 0.0, q = 0.0, r = 0.0, z = 0.0;
            for (m = en - 2; m >= l; --m) {
                z = PSD_HQ(m, m);
                r = x - z;
                double s = y - z;
                p = (r * s - w) / PSD_HQ(m + 1, m) + PSD_HQ(m, m + 1);
                q = PSD_HQ(m + 1, m + 1) - z - r - s;
                r = PSD_HQ(m + 2, m + 1);
                s = fabs(p) + fabs(q) + fabs(r);
                p /= s;
                q /= s;
                r /= s;
                if (m == l) break;
                const double u = fabs(PSD_HQ(m, m - 1)) * (fabs(q) + fabs(r));
                const double v = fabs(p) * (fabs(PSD_HQ(m - 1, m - 1)) + fabs(z) + fabs(PSD_HQ(m + 1, m + 1)));
                if (u <= eps * v) break;
            }
            for (int i = m + 2; i <= en; ++i) {
                PSD_HQ(i, i - 2) = 0.0;
                if (i != m + 2) PSD_HQ(i, i - 3) = 0.0;
            }
            for (int k = m; k <= en - 1; ++k) {  // double QR step on rows l..en and columns m..en
                const bool notlast = (k != en - 1);
                if (k != m) {
                    p = PSD_HQ(k, k - 1);
                    q = PSD_HQ(k + 1, k - 1);
                    r = notlast ? PSD_HQ(k + 2, k - 1) : 0.0;
                    x = fabs(p) + fabs(q) + fabs(r);
                    if (x == 0.0) continue;
                    p /= x;
                    q /= x;
                    r /= x;
                }
                const double s = copysign(sqrt(p * p + q * q + r * r), p);
                if (k != m) {
                    PSD_HQ(k, k - 1) = -s * x;
                } else if (l != m) {
                    PSD_HQ(k, k - 1) = -PSD_HQ(k, k - 1);
                }
                p += s;
                x = p / s;
                y = q / s;
                z = r / s;
                q /= p;
                r /= p;
                for (int j = k; j <= en; ++j) {  // row modification
                    p = PSD_HQ(k, j) + q * PSD_HQ(k + 1, j);
                    if (notlast) {
                        p += r * PSD_HQ(k + 2, j);
                        PSD_HQ(k + 2, j) -= p * z;
                    }
                    PSD_HQ(k + 1, j) -= p * y;
                    PSD_HQ(k, j) -= p * x;
                }
                const int mmin = (en < k + 3) ? en : (k + 3);
                for (int i = l; i <= mmin; ++i) {  // column modification
                    p = x * PSD_HQ(i, k) + y * PSD_HQ(i, k + 1);
                    if (notlast) {
                        p += z * PSD_HQ(i, k + 2);
                        PSD_HQ(i, k + 2) -= p * r;
                    }
                    PSD_HQ(i, k + 1) -= p * q;
                    PSD_HQ(i, k) -= p;
                }
            }
        }
    }
    return true;
#undef PSD_HQ
}

#ifndef PSD_HOSTSIM
// The same iteration run by a whole wavefront on a matrix in LDS: every lane follows the (uniform) control flow and
// computes the scalars redundantly from broadcast reads; the two O(N) inner loops of a double QR step — the row and the
// column modification — are one lane per column / per row.  One lane alone pays an LDS round trip per operand: 1-3 ms for
// N = 16, longer than three windows of the chase it feeds.  Same arithmetic per entry as psd_hqr.
#define PSD_HQW_SYNC() asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory")
__device__ __forceinline__ bool psd_hqr_wave(double* h, int N, int LD, double* wr, double* wi, int lane) {
#define PSD_HQ(r, c) h[(r) * LD + (c)]
    const double eps = PSD_DBL_EPS;
    PSD_HQW_SYNC();
    double norm = 0.0;
    for (int i = 0; i < N; ++i)
        for (int j = (i > 0 ? i - 1 : 0); j < N; ++j) norm += fabs(PSD_HQ(i, j));
    int en = N - 1;
    double t = 0.0;
    while (en >= 0) {
        int its = 0;
        for (;;) {
            PSD_HQW_SYNC();
            int l;
            for (l = en; l >= 1; --l) {
                double s = fabs(PSD_HQ(l - 1, l - 1)) + fabs(PSD_HQ(l, l));
                if (s == 0.0) s = norm;
                if (fabs(PSD_HQ(l, l - 1)) <= eps * s) break;
            }
            double x = PSD_HQ(en, en);
            if (l == en) {  // one root
                wr[en] = x + t;
                wi[en] = 0.0;
                en -= 1;
                break;
            }
            double y = PSD_HQ(en - 1, en - 1);
            double w = PSD_HQ(en, en - 1) * PSD_HQ(en - 1, en);
            if (l == en - 1) {  // two roots
                const double pp = 0.5 * (y - x), qq = pp * pp + w;
                double z = sqrt(fabs(qq));
                x += t;
                if (qq >= 0.0) {
                    z = pp + copysign(z, pp);
                    double w0 = x + z, w1 = x + z;
                    if (z != 0.0) w1 = x - w / z;
                    wr[en - 1] = w0;
                    wr[en] = w1;
                    wi[en - 1] = wi[en] = 0.0;
                } else {
                    wr[en - 1] = wr[en] = x + pp;
                    wi[en - 1] = z;
                    wi[en] = -z;
                }
                en -= 2;
                break;
            }
            if (its == 60) return false;
            if (its == 10 || its == 20) {  // exceptional shift
                t += x;
                const double s = fabs(PSD_HQ(en, en - 1)) + fabs(PSD_HQ(en - 1, en - 2));
                PSD_HQW_SYNC();
                if (lane <= en) PSD_HQ(lane, lane) -= x;
                PSD_HQW_SYNC();
                x = y = 0.75 * s;
                w = -0.4375 * s * s;
            }
            ++its;
            int m;
            double p = 0.0, q = 0.0, r = 0.0, z = 0.0;
            for (m = en - 2; m >= l; --m) {
                z = PSD_HQ(m, m);
                r = x - z;
                double s = y - z;
                p = (r * s - w) / PSD_HQ(m + 1, m) + PSD_HQ(m, m + 1);
                q = PSD_HQ(m + 1, m + 1) - z - r - s;
                r = PSD_HQ(m + 2, m + 1);
                s = fabs(p) + fabs(q) + fabs(r);
                p /= s;
                q /= s;
                r /= s;
                if (m == l) break;
                const double u = fabs(PSD_HQ(m, m - 1)) * (fabs(q) + fabs(r));
                const double v = fabs(p) * (fabs(PSD_HQ(m - 1, m - 1)) + fabs(z) + fabs(PSD_HQ(m + 1, m + 1)));
                if (u <= eps * v) break;
            }
            PSD_HQW_SYNC();
            {
                const int i = m + 2 + lane;
                if (i <= en) {
                    PSD_HQ(i, i - 2) = 0.0;
                    if (i != m + 2) PSD_HQ(i, i - 3) = 0.0;
                }
            }
            PSD_HQW_SYNC();
            for (int k = m; k <= en - 1; ++k) {  // double QR step on rows l..en and columns m..en
                const bool notlast = (k != en - 1);
                if (k != m) {
                    p = PSD_HQ(k, k - 1);
                    q = PSD_HQ(k + 1, k - 1);
                    r = notlast ? PSD_HQ(k + 2, k - 1) : 0.0;
                    x = fabs(p) + fabs(q) + fabs(r);
                    if (x == 0.0) continue;
                    p /= x;
                    q /= x;
                    r /= x;
                }
                const double s = copysign(sqrt(p * p + q * q + r * r), p);
                PSD_HQW_SYNC();  // (every lane has read the entry rewritten below)
                if (lane == 0) {
                    if (k != m) {
                        PSD_HQ(k, k - 1) = -s * x;
                    } else if (l != m) {
                        PSD_HQ(k, k - 1) = -PSD_HQ(k, k - 1);
                    }
                }
                p += s;
                x = p / s;
                y = q / s;
                z = r / s;
                q /= p;
                r /= p;
                {  // row modification: lane = column
                    const int j = k + lane;
                    if (j <= en) {
                        double pj = PSD_HQ(k, j) + q * PSD_HQ(k + 1, j);
                        if (notlast) {
                            pj += r * PSD_HQ(k + 2, j);
                            PSD_HQ(k + 2, j) -= pj * z;
                        }
                        PSD_HQ(k + 1, j) -= pj * y;
                        PSD_HQ(k, j) -= pj * x;
                    }
                }
                PSD_HQW_SYNC();
                {  // column modification: lane = row
                    const int mmin = (en < k + 3) ? en : (k + 3);
                    const int i = l + lane;
                    if (i <= mmin) {
                        double pi = x * PSD_HQ(i, k) + y * PSD_HQ(i, k + 1);
                        if (notlast) {
                            pi += z * PSD_HQ(i, k + 2);
                            PSD_HQ(i, k + 2) -= pi * r;
                        }
                        PSD_HQ(i, k + 1) -= pi * q;
                        PSD_HQ(i, k) -= pi;
                    }
                }
                PSD_HQW_SYNC();
            }
        }
    }
    PSD_HQW_SYNC();
    return true;
#undef PSD_HQ
}
#endif
