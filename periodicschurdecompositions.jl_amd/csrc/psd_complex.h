// Complex (Float64 pair) scalar layer of the engine: arithmetic, the complex Givens generator the
// reference takes from stdlib (LinearAlgebra.givensAlgorithm(::ComplexF64, ::ComplexF64), imported at
// PeriodicSchurDecompositions.jl:9; a port of LAPACK zlartg), and `_safeprod`
// (generalized.jl:939-976).  Device code; evaluated redundantly by every lane on uniform inputs.
#pragma once
#include "psd_scalar.h"

struct psd_z {
    double re, im;
};
PSD_HD psd_z zmk(double re, double im) {
    psd_z z;
    z.re = re;
    z.im = im;
    return z;
}
PSD_HD psd_z zadd(psd_z a, psd_z b) { return zmk(a.re + b.re, a.im + b.im); }
PSD_HD psd_z zsub(psd_z a, psd_z b) { return zmk(a.re - b.re, a.im - b.im); }
PSD_HD psd_z zneg(psd_z a) { return zmk(-a.re, -a.im); }
PSD_HD psd_z zconj(psd_z a) { return zmk(a.re, -a.im); }
PSD_HD psd_z zmul(psd_z a, psd_z b) { return zmk(a.re * b.re - a.im * b.im, a.re * b.im + a.im * b.re); }
PSD_HD psd_z zscal(double s, psd_z a) { return zmk(s * a.re, s * a.im); }
PSD_HD double zabs2(psd_z a) { return a.re * a.re + a.im * a.im; }
PSD_HD double zabs(psd_z a) { return hypot(a.re, a.im); }
PSD_HD double zabs1(psd_z a) { return fmax(fabs(a.re), fabs(a.im)); }
PSD_HD bool ziszero(psd_z a) { return a.re == 0.0 && a.im == 0.0; }
// Smith's algorithm (what Julia's complex division reduces to in the normal range)
PSD_HD psd_z zdiv(psd_z a, psd_z b) {
    if (fabs(b.re) >= fabs(b.im)) {
        const double r = b.im / b.re, d = b.re + b.im * r;
        return zmk((a.re + a.im * r) / d, (a.im - a.re * r) / d);
    }
    const double r = b.re / b.im, d = b.re * r + b.im;
    return zmk((a.re * r + a.im) / d, (a.im * r - a.re) / d);
}

// (c, s, r): c real, [c s; -conj(s) c][f; g] = [r; 0]
PSD_HD void psd_zgivens(psd_z f, psd_z g, double& cs, psd_z& sn, psd_z& r) {
    const double safmin = PSD_DBL_MIN;
    const double safmn2 = 1.0010415475915505e-146;
    const double safmx2 = 9.989595361011175e+145;
    double scale = fmax(zabs1(f), zabs1(g));
    if (scale < safmx2 && scale > safmn2) {
        // common case of zlartg (no rescaling, f not negligible against g), with the two square roots
        // taken as independent rsqrt pairs: cs = |f|/N, r = f N/|f|, sn = f conj(g)/(|f| N), N^2 = |f|^2+|g|^2
        const double f2 = zabs2(f), g2 = zabs2(g);
        if (f2 > fmax(g2, 1.0) * safmin) {
            double sd, rsd, sf, rsf;
            psd_sqrt_pair_fast(f2 + g2, sd, rsd);
            psd_sqrt_pair_fast(f2, sf, rsf);
            cs = sf * rsd;
            r = zscal(sd * rsf, f);
            sn = zmul(zscal(rsf * rsd, f), zconj(g));
            return;
        }
    }
    psd_z fs = f, gs = g;
    int count = 0;
    if (scale >= safmx2) {
        do {
            count += 1;
            fs = zscal(safmn2, fs);
            gs = zscal(safmn2, gs);
            scale *= safmn2;
        } while (scale >= safmx2 && count < 20);
    } else if (scale <= safmn2) {
        if (ziszero(g)) {
            cs = 1.0;
            sn = zmk(0.0, 0.0);
            r = f;
            return;
        }
        do {
            count -= 1;
            fs = zscal(safmx2, fs);
            gs = zscal(safmx2, gs);
            scale *= safmx2;
        } while (scale <= safmn2 && count > -40);
    }
    const double f2 = zabs2(fs), g2 = zabs2(gs);
    if (f2 <= fmax(g2, 1.0) * safmin) {
        if (ziszero(f)) {
            cs = 0.0;
            r = zmk(hypot(g.re, g.im), 0.0);
            const double d = hypot(gs.re, gs.im);
            sn = zmk(gs.re / d, -gs.im / d);
            return;
        }
        const double f2s = hypot(fs.re, fs.im);
        const double g2s = sqrt(g2);
        cs = f2s / g2s;
        psd_z ff;
        if (zabs1(f) > 1) {
            const double d = hypot(f.re, f.im);
            ff = zmk(f.re / d, f.im / d);
        } else {
            const double dr = safmx2 * f.re, di = safmx2 * f.im;
            const double d = hypot(dr, di);
            ff = zmk(dr / d, di / d);
        }
        sn = zmul(ff, zmk(gs.re / g2s, -gs.im / g2s));
        r = zadd(zscal(cs, f), zmul(sn, g));
    } else {
        const double f2s = sqrt(1.0 + g2 / f2);
        r = zmk(f2s * fs.re, f2s * fs.im);
        cs = 1.0 / f2s;
        const double d = f2 + g2;
        sn = zmul(zmk(r.re / d, r.im / d), zconj(gs));
        if (count > 0)
            for (int i = 0; i < count; ++i) r = zscal(safmx2, r);
        else if (count < 0)
            for (int i = 0; i < -count; ++i) r = zscal(safmn2, r);
    }
}

// psd_zgivens for the chase chains: the common case as straight-line code (one range test; the squares neither overflow
// nor vanish, f not negligible against g), everything else out of line and by value (an argument passed by reference to a
// function that is not inlined lives in scratch memory).  Same quantities as the fast path of psd_zgivens, from reciprocal roots alone.
struct psd_zgiv_out {
    double cs;
    psd_z sn, r;
};
PSD_D_NOINLINE psd_zgiv_out psd_zgivens_slow(psd_z f, psd_z g) {
    psd_zgiv_out o;
    psd_zgivens(f, g, o.cs, o.sn, o.r);
    return o;
}
PSD_D void psd_zgivens_lean(psd_z f, psd_z g, double& cs, psd_z& sn, psd_z& r) {
    // cs = |f| / N, r = f N / |f|, sn = f conj(g) / (|f| N), N^2 = |f|^2 + |g|^2: with rf = 1/|f|, rn = 1/N and k = rf rn
    // that is cs = |f|^2 k, r = f (N^2 k), sn = (f conj(g)) k — only the two reciprocal roots are on the chain, f conj(g)
    // and the range test run beside them
    const double f2 = zabs2(f), g2 = zabs2(g), n2 = f2 + g2;
    double rf, rn;
    psd_rsqrt2_fast(f2, n2, rf, rn);
    const psd_z fg = zmul(f, zconj(g));
    const double k = rf * rn;
    double c0 = f2 * k;
    psd_z r0 = zscal(n2 * k, f);
    psd_z s0 = zscal(k, fg);
    // (magnitudes between 1e-146 and 1e145 as in psd_zgivens: then n2 < 1e291, and f2 is normal)
    const double scale = fmax(zabs1(f), zabs1(g));
    const bool ok = (scale < 9.989595361011175e+145) & (scale > 1.0010415475915505e-146) & (f2 > fmax(g2, 1.0) * PSD_DBL_MIN);
    if (!ok) {
        const psd_zgiv_out o = psd_zgivens_slow(f, g);
        c0 = o.cs;
        s0 = o.sn;
        r0 = o.r;
    }
    cs = c0;
    sn = s0;
    r = r0;
}

// stdlib lmul!(G, .): (a1, a2) <- (c a1 + s a2, -conj(s) a1 + c a2)
PSD_HD void psd_zrot_left(double c, psd_z s, psd_z& a1, psd_z& a2) {
    const psd_z b1 = zadd(zscal(c, a1), zmul(s, a2));
    const psd_z b2 = zsub(zscal(c, a2), zmul(zconj(s), a1));
    a1 = b1;
    a2 = b2;
}
// stdlib rmul!(., G') with G' = Givens(c, -s): (a1, a2) <- (a1 c + a2 conj(s), -a1 s + a2 c)
PSD_HD void psd_zrot_right_adj(double c, psd_z s, psd_z& a1, psd_z& a2) {
    const psd_z b1 = zadd(zscal(c, a1), zmul(a2, zconj(s)));
    const psd_z b2 = zsub(zscal(c, a2), zmul(a1, s));
    a1 = b1;
    a2 = b2;
}
