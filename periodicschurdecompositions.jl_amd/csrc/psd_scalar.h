// Scalar building blocks of the engine (device code; every lane of a wavefront evaluates them
// redundantly on wave-uniform inputs, so control flow stays uniform).
// Each function names the reference routine whose contract it implements
// (paths under /root/reference/src).
#pragma once
#include "psd_platform.h"

#define PSD_DBL_MIN 2.2250738585072014e-308
#define PSD_DBL_EPS 2.220446049250313e-16

// householder.jl:5-24 (_norm2) for the 1- and 2-entry tails that occur in the QR/QZ sweeps
PSD_HD double psd_norm2_small(const double* x, int n) {
    if (n < 1) return 0.0;
    if (n == 1) return fabs(x[0]);
    double scale = 0.0, ssq = 0.0;
    for (int k = 0; k < n; ++k) {
        const double a = fabs(x[k]);
        if (a != 0.0) {
            if (scale < a) {
                const double q = scale / a;
                ssq = 1.0 + ssq * q * q;
                scale = a;
            } else {
                const double q = a / scale;
                ssq += q * q;
            }
        }
    }
    return scale * sqrt(ssq);
}

// householder.jl:66-108 (_xreflector!, LAPACK dlarfg) for vectors of length n <= 3 held in
// registers: x <- (beta, v2, v3), returns tau.  H = I - tau [1;v][1;v]'.
PSD_HD double psd_reflector_small(double* x, int n) {
    if (n <= 1) return 0.0;
    const double sfmin = 2.0 * PSD_DBL_MIN / PSD_DBL_EPS;
    double alpha = x[0];
    double xnorm = psd_norm2_small(x + 1, n - 1);
    if (xnorm == 0.0) return 0.0;
    double beta = -copysign(hypot(alpha, xnorm), alpha);
    int kount = 0;
    if (fabs(beta) < sfmin) {
        const double rsfmin = 1.0 / sfmin;
        bool smallb = true;
        while (smallb) {
            kount += 1;
            for (int j = 1; j < n; ++j) x[j] *= rsfmin;
            beta *= rsfmin;
            alpha *= rsfmin;
            smallb = (fabs(beta) < sfmin) && (kount < 20);
        }
        xnorm = psd_norm2_small(x + 1, n - 1);
        beta = -copysign(hypot(alpha, xnorm), alpha);
    }
    const double tau = (beta - alpha) / beta;
    const double t = 1.0 / (alpha - beta);
    for (int j = 1; j < n; ++j) x[j] *= t;
    for (int j = 0; j < kount; ++j) beta *= sfmin;
    x[0] = beta;
    return tau;
}

// Register-only forms of the same reflector for the sweep's 3- and 2-vectors.  When no square can
// overflow or underflow (the common case) the norm is formed directly — one sqrt and two
// independent divisions on the dependency chain instead of the scaled-sum-of-squares recurrence;
// otherwise the dlarfg path above runs.  (x0, x1, x2) <- (beta, v2, v3); returns tau.
PSD_HD double psd_refl3(double& x0, double& x1, double& x2) {
    const double tmax = fmax(fabs(x1), fabs(x2));
    if (tmax == 0.0) return 0.0;
    const double big = fmax(tmax, fabs(x0));
    if (big < 1e140 && tmax > 1e-140) {
        // beta = -sign(x0) nrm;  tau = (beta - x0)/beta = 1 + |x0|/nrm;  v = x / (x0 - beta)
        double nrm, rn;
        psd_sqrt_pair_fast(x0 * x0 + (x1 * x1 + x2 * x2), nrm, rn);
        const double ax0 = fabs(x0);
        const double tau = 1.0 + ax0 * rn;
        const double t = psd_rcp_fast(copysign(ax0 + nrm, x0));
        x1 *= t;
        x2 *= t;
        x0 = -copysign(nrm, x0);
        return tau;
    }
    double x[3] = {x0, x1, x2};
    const double tau = psd_reflector_small(x, 3);
    x0 = x[0];
    x1 = x[1];
    x2 = x[2];
    return tau;
}
PSD_HD double psd_refl2(double& x0, double& x1) {
    const double tmax = fabs(x1);
    if (tmax == 0.0) return 0.0;
    const double big = fmax(tmax, fabs(x0));
    if (big < 1e140 && tmax > 1e-140) {
        double nrm, rn;
        psd_sqrt_pair_fast(x0 * x0 + x1 * x1, nrm, rn);
        const double ax0 = fabs(x0);
        const double tau = 1.0 + ax0 * rn;
        const double t = psd_rcp_fast(copysign(ax0 + nrm, x0));
        x1 *= t;
        x0 = -copysign(nrm, x0);
        return tau;
    }
    double x[2] = {x0, x1};
    const double tau = psd_reflector_small(x, 2);
    x0 = x[0];
    x1 = x[1];
    return tau;
}

// psd_refl3 on (x0, x1, x2) and psd_refl2 on (y0, y1) at once: the two fast paths are evaluated in ONE straight-line
// block (no control flow between them), so that the instruction scheduler interleaves the two independent dependency
// chains; the range checks come afterwards and send the rare cases to the individual routines.
PSD_HD void psd_refl32_pair(double& x0, double& x1, double& x2, double& tau3, double& y0, double& y1, double& tau2) {
    // Both fast paths are evaluated unconditionally (on out-of-range data they produce values nobody uses) and ONE
    // combined range test follows: with a test per reflector the compiler branches between the two chains and they
    // run one after the other.  The test is on the sums of squares: finite and < 1e280 means no square overflowed,
    // tail sum > 1e-280 means the tail is neither zero (H = I, tau = 0: dlarfg path) nor lost to underflow.
    const double tx2 = x1 * x1 + x2 * x2, nx2 = x0 * x0 + tx2;
    const double ty2 = y1 * y1, ny2 = y0 * y0 + ty2;
    double nx, rnx, ny, rny;
    psd_sqrt_pair_fast(nx2, nx, rnx);
    psd_sqrt_pair_fast(ny2, ny, rny);
    const double ax = fabs(x0), ay = fabs(y0);
    double t3 = 1.0 + ax * rnx, t2 = 1.0 + ay * rny;
    const double sx = psd_rcp_fast(copysign(ax + nx, x0)), sy = psd_rcp_fast(copysign(ay + ny, y0));
    double fx1 = x1 * sx, fx2 = x2 * sx, fx0 = -copysign(nx, x0);
    double fy1 = y1 * sy, fy0 = -copysign(ny, y0);
#if defined(__HIP_DEVICE_COMPILE__)
    PSD_KEEP(fx1);
    PSD_KEEP(fx2);
    PSD_KEEP(t3);
    PSD_KEEP(fy1);
    PSD_KEEP(t2);
#endif
    const bool ok = (nx2 < 1e280) & (tx2 > 1e-280) & (ny2 < 1e280) & (ty2 > 1e-280);
    if (ok) {
        x0 = fx0;
        x1 = fx1;
        x2 = fx2;
        tau3 = t3;
        y0 = fy0;
        y1 = fy1;
        tau2 = t2;
    } else {
        tau3 = psd_refl3(x0, x1, x2);
        tau2 = psd_refl2(y0, y1);
    }
}

// The same reflectors for a chain that must stay short in code as well as in time (psd_c2_run: one reflector per link
// and wavefront): the fast path straight-line, ONE range test on the sums of squares behind it (as psd_refl32_pair), the
// dlarfg path out of line.
// (by value both ways: an argument passed by reference to a function that is not inlined lives in scratch memory, and
//  its loads and stores would sit on the chain of every link)
struct psd_refl_out {
    double x0, x1, x2, tau;
};
PSD_D_NOINLINE psd_refl_out psd_refl3_slow(double x0, double x1, double x2) {
    psd_refl_out o;
    o.tau = psd_refl3(x0, x1, x2);
    o.x0 = x0;
    o.x1 = x1;
    o.x2 = x2;
    return o;
}
PSD_D_NOINLINE psd_refl_out psd_refl2_slow(double x0, double x1) {
    psd_refl_out o;
    o.tau = psd_refl2(x0, x1);
    o.x0 = x0;
    o.x1 = x1;
    o.x2 = 0.0;
    return o;
}
PSD_D double psd_refl3_lean(double& x0, double& x1, double& x2) {
    const double tx2 = x1 * x1 + x2 * x2, nx2 = x0 * x0 + tx2;
    double nx, rnx;
    psd_sqrt_pair_fast(nx2, nx, rnx);
    const double ax = fabs(x0);
    double t3 = 1.0 + ax * rnx;
    const double sx = psd_rcp_fast(copysign(ax + nx, x0));
    double f0 = -copysign(nx, x0), f1 = x1 * sx, f2 = x2 * sx;
    const bool ok = (nx2 < 1e280) & (tx2 > 1e-280);
    if (!ok) {
        const psd_refl_out o = psd_refl3_slow(x0, x1, x2);
        f0 = o.x0;
        f1 = o.x1;
        f2 = o.x2;
        t3 = o.tau;
    }
    x0 = f0;
    x1 = f1;
    x2 = f2;
    return t3;
}
PSD_D double psd_refl2_lean(double& y0, double& y1) {
    const double ty2 = y1 * y1, ny2 = y0 * y0 + ty2;
    double ny, rny;
    psd_sqrt_pair_fast(ny2, ny, rny);
    const double ay = fabs(y0);
    double t2 = 1.0 + ay * rny;
    const double sy = psd_rcp_fast(copysign(ay + ny, y0));
    double f0 = -copysign(ny, y0), f1 = y1 * sy;
    const bool ok = (ny2 < 1e280) & (ty2 > 1e-280);
    if (!ok) {
        const psd_refl_out o = psd_refl2_slow(y0, y1);
        f0 = o.x0;
        f1 = o.x1;
        t2 = o.tau;
    }
    y0 = f0;
    y1 = f1;
    return t2;
}

// stdlib LinearAlgebra.givensAlgorithm(f::Float64, g::Float64) (imported by the reference at
// PSD.jl:9): (c, s, r) with [c s; -s c][f; g] = [r; 0].
PSD_HD void psd_givens(double f, double g, double& cs, double& sn, double& r) {
    // floatmin2(Float64) = 2^trunc(log2(floatmin/eps)/2) = 2^-485
    const double safmn2 = 1.0010415475915505e-146;
    const double safmx2 = 9.989595361011175e+145;
    if (g == 0.0) {
        cs = 1.0; sn = 0.0; r = f;
    } else if (f == 0.0) {
        cs = 0.0; sn = 1.0; r = g;
    } else {
        double f1 = f, g1 = g;
        double scale = fmax(fabs(f1), fabs(g1));
        if (scale >= safmx2) {
            int count = 0;
            do {
                count += 1;
                f1 *= safmn2; g1 *= safmn2;
                scale = fmax(fabs(f1), fabs(g1));
            } while (scale >= safmx2 && count < 20);
            r = sqrt(f1 * f1 + g1 * g1);
            cs = f1 / r; sn = g1 / r;
            for (int q = 0; q < count; ++q) r *= safmx2;
        } else if (scale <= safmn2) {
            int count = 0;
            do {
                count += 1;
                f1 *= safmx2; g1 *= safmx2;
                scale = fmax(fabs(f1), fabs(g1));
            } while (scale <= safmn2 && count < 40);
            r = sqrt(f1 * f1 + g1 * g1);
            cs = f1 / r; sn = g1 / r;
            for (int q = 0; q < count; ++q) r *= safmn2;
        } else {
            // common case: rsqrt + Newton pair instead of the IEEE sqrt and two divisions (the argument is in
            // (2^-970, 2^970), far from the range limits)
            double rinv;
            psd_sqrt_pair_fast(f1 * f1 + g1 * g1, r, rinv);
            cs = f1 * rinv; sn = g1 * rinv;
        }
        if (fabs(f) > fabs(g) && cs < 0.0) {
            cs = -cs; sn = -sn; r = -r;
        }
    }
}

// rschur2x2.jl:9-96 (_gs2x2!, LAPACK dlanv2): standard form of a real 2x2 block; rotation
// (cs, sn) and eigenvalues (w1r + i w1i, w2r + i w2i).
PSD_HD void psd_gs2x2(double& a, double& b, double& c, double& d, double& cs, double& sn, double& w1r,
                      double& w1i, double& w2r, double& w2i) {
    const double half = 0.5;
    const double small = 4.0 * PSD_DBL_EPS;
#define PSD_SGN(x) (((x) < 0) ? -1.0 : 1.0)
    if (c == 0) {
        cs = 1.0; sn = 0.0;
    } else if (b == 0) {
        cs = 0.0; sn = 1.0;
        const double a0 = a, c0 = c, d0 = d;
        a = d0; b = -c0; c = 0.0; d = a0;
    } else if (((a - d) == 0) && (b * c < 0)) {
        cs = 1.0; sn = 0.0;
    } else {
        const double asubd = a - d;
        double p = half * asubd;
        const double bcmax = fmax(fabs(b), fabs(c));
        const double bcmis = fmin(fabs(b), fabs(c)) * PSD_SGN(b) * PSD_SGN(c);
        const double scale = fmax(fabs(p), bcmax);
        double z = (p / scale) * p + (bcmax / scale) * bcmis;
        if (z >= small) {
            z = p + sqrt(scale) * sqrt(z) * PSD_SGN(p);
            a = d + z;
            d -= (bcmax / z) * bcmis;
            const double tau = hypot(c, z);
            cs = z / tau;
            sn = c / tau;
            b -= c;
            c = 0.0;
        } else {
            const double sigma = b + c;
            double tau = hypot(sigma, asubd);
            cs = sqrt(half * (1.0 + fabs(sigma) / tau));
            sn = -(p / (tau * cs)) * PSD_SGN(sigma);
            const double aa = a * cs + b * sn;
            const double bb = -a * sn + b * cs;
            const double cc = c * cs + d * sn;
            const double dd = -c * sn + d * cs;
            a = aa * cs + cc * sn;
            b = bb * cs + dd * sn;
            c = -aa * sn + cc * cs;
            d = -bb * sn + dd * cs;
            const double midad = half * (a + d);
            a = midad;
            d = a;
            if (c != 0) {
                if (b != 0) {
                    if (b * c >= 0) {
                        const double sab = sqrt(fabs(b));
                        const double sac = sqrt(fabs(c));
                        p = sab * sac * PSD_SGN(c);
                        tau = 1.0 / sqrt(fabs(b + c));
                        a = midad + p;
                        d = midad - p;
                        b -= c;
                        c = 0;
                        const double cs1 = sab * tau;
                        const double sn1 = sac * tau;
                        const double cs2 = cs * cs1 - sn * sn1;
                        const double sn2 = cs * sn1 + sn * cs1;
                        cs = cs2; sn = sn2;
                    }
                } else {
                    b = -c; c = 0.0;
                    const double cs2 = -sn, sn2 = cs;
                    cs = cs2; sn = sn2;
                }
            }
        }
    }
#undef PSD_SGN
    if (c == 0) {
        w1r = a; w1i = 0.0; w2r = d; w2i = 0.0;
    } else {
        const double rti = sqrt(fabs(b)) * sqrt(fabs(c));
        w1r = a; w1i = rti; w2r = d; w2i = -rti;
    }
}
