// ORACLE — TEST INFRASTRUCTURE ONLY (rules in psd_oracle_real.hpp).
//
// CPU restatement of the real generalized (signed) periodic QZ iteration
//   pschur!(H1, Hs, S; wantZ, wantT, Q, maxitfac)          rgeneralized.jl:49-1083   (after SLICOT MB03BD)
// with its helpers
//   _check_deflate_hess / _check_deflate_tr                 rgeneralized.jl:1086-1136
//   _qzrots   (MB03AF 'Double': implicit double shift)      rgeneralized.jl:1140-1359
//   _qzrot2x2 (MB03AF 'Single', n = 2)                      rgeneralized.jl:1364-1396
//   _rp2x2ssr! (MB03BF: real single-shift PQZ on 2x2s)      rpschur2x2.jl:280-318
//   _rpeigvals2x2 with a signature                          rpschur2x2.jl:9-235
//
// NOT reproduced (reference defects, listed in DESIGN.md section 5):
//   * the explicit-shift branch rgeneralized.jl:804-887 (every 11th sweep): it reads the undefined `hnorm`
//     (:826), `_shift2rot` returns the undefined `c, s` for a single shift (:1442), indexes the whole H1 where the
//     active view is meant (:1457) and has "missing logic for recip".  Every sweep here uses the implicit double
//     shift of _qzrots, which is what MB03BD itself does while its own explicit-shift counter is idle; the
//     decomposition contract (invariants) is unaffected.
//   * `aggressive = true` (throws NotImplemented in the zero-shift block :245): only the default is restated.
#pragma once
#include "psd_oracle_rord.hpp"
#include "psd_oracle_sghess.hpp"

namespace psdo {

struct RGLog {
    long niter = 0, nsweeps = 0, nzero = 0, ncase2 = 0, ncase3 = 0, n2x2real = 0, n2x2cplx = 0, ndefl1 = 0;
    int iwarn = 0;
};

typedef GivT<double> GivR;
typedef MatT<double> MatR;

inline void rgiv(double f, double g, double& c, double& s, double& r) { givens_algorithm(f, g, c, s, r); }

inline double opnorm1r(const MatR& M, int r0, int r1, int c0, int c1, bool upper) {
    double best = 0;
    for (int c = c0; c <= c1; ++c) {
        double s = 0;
        for (int r = r0; r <= r1; ++r) {
            if (upper && (r - r0) > (c - c0)) break;
            s += std::fabs(M(r, c));
        }
        best = std::max(best, s);
    }
    return best;
}

// rgeneralized.jl:1086-1112
inline bool rg_check_deflate_hess(const MatR& H1, int ilo, int ilast, double ulp, double smlnum, int& jlo) {
    jlo = ilo;
    for (int j = ilast; j >= ilo + 1; --j) {
        double tol = std::fabs(H1(j - 1, j - 1)) + std::fabs(H1(j, j));
        if (tol == 0) tol = opnorm1r(H1, ilo, j, ilo, j, false);
        tol = std::max(ulp * tol, smlnum);
        if (std::fabs(H1(j, j - 1)) <= tol) {
            H1(j, j - 1) = 0.0;
            jlo = j;
            return j == ilast;
        }
    }
    return false;
}
// rgeneralized.jl:1114-1136
inline bool rg_check_deflate_tr(const MatR& Hl, int jlo, int ilast, double ulp, double smlnum, int& jx) {
    for (int j = ilast; j >= jlo; --j) {
        double tol;
        if (j == ilast) tol = std::fabs(Hl(j - 1, j));
        else if (j == jlo) tol = std::fabs(Hl(j, j + 1));
        else tol = std::fabs(Hl(j - 1, j)) + std::fabs(Hl(j, j + 1));
        if (tol == 0) tol = opnorm1r(Hl, jlo, j, jlo, j, true);
        tol = std::max(ulp * tol, smlnum);
        if (std::fabs(Hl(j, j)) <= tol) {
            Hl(j, j) = 0.0;
            jx = j;
            return true;
        }
    }
    jx = 0;
    return false;
}

// rgeneralized.jl:1140-1359.  H[1] Hessenberg, H[2..p] triangular; active block starts at i1, has order nb.
inline void rg_qzrots(int p, const std::vector<MatR>& H, const std::vector<char>& S, int i1, int nb, double& c1o,
                      double& s1o, double& c2o, double& s2o) {
    const MatR& H1 = H[1];
    double c1, s1, c2, s2, r, al, be, ga, de;
    rgiv(H1(i1, i1), H1(i1 + 1, i1), c1, s1, r);
    rgiv(r, 1.0, c2, s2, r);
    const int i2 = i1 + nb - 1;
    for (int l = p; l >= 2; --l) {
        const MatR& Hl = H[l];
        if (S[l]) {
            al = c2 * (c1 * Hl(i1, i1) + s1 * Hl(i1, i1 + 1));
            be = s1 * c2 * Hl(i1 + 1, i1 + 1);
            ga = s2 * Hl(i2, i2);
            rgiv(al, be, c1, s1, r);
            double v;
            rgiv(r, ga, c2, s2, v);
        } else {
            al = c1 * s2 * Hl(i1, i1);
            ga = s1 * Hl(i1, i1);
            be = s2 * (c1 * Hl(i1, i1 + 1) + s1 * Hl(i1 + 1, i1 + 1));
            de = c1 * Hl(i1 + 1, i1 + 1) - s1 * Hl(i1, i1 + 1);
            rgiv(de, ga, c1, s1, r);
            al = c1 * al + s1 * be;
            be = c2 * Hl(i2, i2);
            rgiv(be, al, c2, s2, r);
        }
    }
    al = s2 * H1(i2, i2) - c1 * c2;
    be = -s1 * c2;
    const int n = nb, m = nb - 1;
    auto V1 = [&](int a, int b) { return H1(i1 - 1 + a, i1 - 1 + b); };
    ga = -s2 * V1(n, m);
    rgiv(al, ga, c2, s2, r);
    rgiv(r, be, c1, s1, r);
    double cx = c1 * c2, sx = c1 * s2;
    be = s1 * V1(n, m);
    al = cx * V1(n, m) + sx * V1(n, n);
    ga = s1 * V1(m, m);
    de = cx * V1(m, m) + sx * V1(m, n);
    double val1 = s1 * V1(3, 2), val2 = cx * V1(2, 1) + s1 * V1(2, 2), val3 = cx * V1(1, 1) + s1 * V1(1, 2);
    double c3, s3, c4, s4, c5, s5, c6, s6;
    rgiv(al, be, c1, s1, r);
    rgiv(ga, r, c2, s2, r);
    rgiv(de, r, c3, s3, r);
    rgiv(val1, r, c4, s4, r);
    rgiv(val2, r, c5, s5, r);
    rgiv(val3, r, c6, s6, r);
    for (int i = p; i >= 2; --i) {
        const MatR& Hi = H[i];
        auto V = [&](int a, int b) { return Hi(i1 - 1 + a, i1 - 1 + b); };
        if (S[i]) {
            double ss = s3 * s4, sss = s2 * ss, ssss = s1 * sss;
            val1 = c4 * V(1, 3);
            val2 = c4 * V(2, 3);
            val3 = c4 * V(3, 3);
            al = s4 * c3 * V(m, m) + sss * c1 * V(m, n);
            be = ss * c2 * V(m, m) + ssss * V(m, n);
            ga = sss * c1 * V(n, n);
            de = ssss * V(n, n);
            ss = s5 * s6;
            const double cs = c5 * s6;
            val1 = ss * val1 + cs * V(1, 2) + c6 * V(1, 1);
            val2 = ss * val2 + cs * V(2, 2);
            val3 = ss * val3;
            al = ss * al;
            be = ss * be;
            ga = ss * ga;
            de = ss * de;
            rgiv(ga, de, c1, s1, r);
            rgiv(be, r, c2, s2, r);
            rgiv(al, r, c3, s3, r);
            rgiv(val3, r, c4, s4, r);
            rgiv(val2, r, c5, s5, r);
            rgiv(val1, r, c6, s6, r);
        } else {
            double ep, ze, et, th, val4, val5;
            double c2R, s2R, c3R, s3R, c4R, s4R, c5R, s5R, c6R, s6R;
            de = c1 * V(n, n);
            ep = s1 * V(n, n);
            al = c2 * V(m, m);
            be = s2 * de;
            ga = -s2 * V(m, m);
            ze = c2 * V(m, n) + s2 * ep;
            et = -s2 * V(m, n) + c2 * ep;
            de = c1 * c2 * de + s1 * et;
            rgiv(de, -ga, c2R, s2R, r);
            de = c3 * V(m, m);
            ep = s3 * al;
            et = c3 * V(m, n) + s3 * be;
            th = s3 * ze;
            ga = -s3 * V(m, m);
            be = -s3 * V(m, n) + c3 * be;
            al = c2R * c3 * al + s2R * (c1 * be + s1 * c3 * ze);
            rgiv(al, -ga, c3R, s3R, r);
            val1 = c4 * V(3, 3);
            val2 = s4 * de;
            val3 = s4 * ep;
            val4 = s4 * et;
            val5 = s4 * th;
            be = -s4 * V(3, 3);
            de = c4 * de;
            ep = c4 * ep;
            ze = c4 * et;
            et = c4 * th;
            al = c3R * de + s3R * (c2R * ep + s2R * (c1 * ze + s1 * et));
            rgiv(al, -be, c4R, s4R, r);
            be = c5 * V(2, 2);
            de = c5 * V(2, 3) + s5 * val1;
            ep = s5 * val2;
            ze = s5 * val3;
            et = s5 * val4;
            th = s5 * val5;
            ga = -s5 * V(2, 2);
            val1 = c5 * val1 - s5 * V(2, 3);
            val2 = c5 * val2;
            val3 = c5 * val3;
            val4 = c5 * val4;
            val5 = c5 * val5;
            al = c4R * val1 + s4R * (c3R * val2 + s3R * (c2R * val3 + s2R * (c1 * val4 + s1 * val5)));
            rgiv(al, -ga, c5R, s5R, r);
            ga = -s6 * V(1, 1);
            be = c6 * be - s6 * V(1, 2);
            de = c6 * de - s6 * V(1, 3);
            ep = c6 * ep;
            ze = c6 * ze;
            et = c6 * et;
            th = c6 * th;
            al = c5R * be + s5R * (c4R * de + s4R * (c3R * ep + s3R * (c2R * ze + s2R * (c1 * et + s1 * th))));
            rgiv(al, -ga, c6R, s6R, r);
            c2 = c2R; s2 = s2R; c3 = c3R; s3 = s3R; c4 = c4R; s4 = s4R; c5 = c5R; s5 = s5R; c6 = c6R; s6 = s6R;
        }
    }
    val1 = s5 * s6;
    val2 = s4 * val1;
    val3 = s3 * val2;
    al = c3 * val2 - c6;
    be = c2 * val3 - c5 * s6;
    ga = -c4 * val1;
    rgiv(be, ga, c2, s2, r);
    rgiv(al, r, c1, s1, r);
    c1o = c1; s1o = s1; c2o = c2; s2o = s2;
}

struct R2 {
    double a, b, c, d;  // [a b; c d]
};

// rgeneralized.jl:1364-1396.  X[0..p-1], Hessenberg (full) block LAST; S2[l] its signature
inline void rg_qzrot2x2(int p, const std::vector<R2>& X, const std::vector<char>& S2, double& c1, double& s1) {
    double c2, s2, r, al, be, ga, de;
    const R2& Hp = X[p - 1];
    rgiv(Hp.a, Hp.c, c1, s1, r);
    rgiv(r, 1.0, c2, s2, r);
    for (int l = p - 2; l >= 0; --l) {
        const R2& Hl = X[l];
        if (S2[l]) {
            al = c2 * (c1 * Hl.a + s1 * Hl.b);
            be = s1 * c2 * Hl.d;
            ga = s2 * Hl.d;
            rgiv(al, be, c1, s1, r);
            double v;
            rgiv(r, ga, c2, s2, v);
        } else {
            al = c1 * s2 * Hl.a;
            ga = s1 * Hl.a;
            be = s2 * (c1 * Hl.b + s1 * Hl.d);
            de = c1 * Hl.d - s1 * Hl.b;
            rgiv(de, ga, c1, s1, r);
            al = c1 * al + s1 * be;
            be = c2 * Hl.d;
            rgiv(be, al, c2, s2, r);
        }
    }
    al = s2 * Hp.d - c1 * c2;
    be = -s1 * c2;
    rgiv(al, be, c1, s1, r);
}

// rpschur2x2.jl:280-318
inline bool rg_rp2x2ssr(int p, std::vector<R2>& X, const std::vector<char>& S2, int maxit = 20) {
    const double ulp = std::numeric_limits<double>::epsilon();
    bool done = false;
    for (int iter = 1; iter <= maxit; ++iter) {
        double c, s, r;
        rg_qzrot2x2(p, X, S2, c, s);
        {  // rmul!(H2s[p], G')
            R2& Y = X[p - 1];
            double a1 = Y.a, a2 = Y.b;
            Y.a = a1 * c + a2 * s;
            Y.b = -a1 * s + a2 * c;
            a1 = Y.c; a2 = Y.d;
            Y.c = a1 * c + a2 * s;
            Y.d = -a1 * s + a2 * c;
        }
        for (int l = 0; l < p - 1; ++l) {
            R2& Y = X[l];
            if (S2[l]) {
                double a1 = Y.a, a2 = Y.c;  // lmul!(G, Hl)
                Y.a = c * a1 + s * a2;
                Y.c = -s * a1 + c * a2;
                a1 = Y.b; a2 = Y.d;
                Y.b = c * a1 + s * a2;
                Y.d = -s * a1 + c * a2;
                rgiv(Y.d, -Y.c, c, s, r);
                Y.d = r;
                Y.c = 0.0;
                const double t1 = c * Y.a + s * Y.b, t2 = c * Y.b - s * Y.a;
                Y.a = t1;
                Y.b = t2;
            } else {
                double a1 = Y.a, a2 = Y.b;  // rmul!(Hl, G')
                Y.a = a1 * c + a2 * s;
                Y.b = -a1 * s + a2 * c;
                a1 = Y.c; a2 = Y.d;
                Y.c = a1 * c + a2 * s;
                Y.d = -a1 * s + a2 * c;
                rgiv(Y.a, Y.c, c, s, r);
                Y.a = r;
                Y.c = 0.0;
                const double t1 = c * Y.b + s * Y.d, t2 = c * Y.d - s * Y.b;
                Y.b = t1;
                Y.d = t2;
            }
        }
        R2& Y = X[p - 1];
        double a1 = Y.a, a2 = Y.c;
        Y.a = c * a1 + s * a2;
        Y.c = -s * a1 + c * a2;
        a1 = Y.b; a2 = Y.d;
        Y.b = c * a1 + s * a2;
        Y.d = -s * a1 + c * a2;
        done = std::fabs(Y.c) < ulp * std::max(std::fabs(Y.a), std::max(std::fabs(Y.b), std::fabs(Y.d)));
        if (done) break;
    }
    return done;
}

// rgeneralized.jl:49-1083.  H[1..p] (H[1] Hessenberg), Z[1..p] accumulated from the right when wantZ.
// Returns 0, or PSD-style 1000000 + ilast on non-convergence (the reference throws at :1058).
inline int rg_pschur_hess(int n, int p, std::vector<MatR>& H, const std::vector<char>& S, std::vector<MatR>& Z,
                          bool wantT, bool wantZ, int maxitfac, cplx* alpha, double* beta, int* ascale, RGLog* log) {
    if (!S[1]) return -5;
    const double unfl = std::numeric_limits<double>::min();
    const double ulp = std::numeric_limits<double>::epsilon();
    const double smlnum = unfl * (n / ulp);
    MatR& H1 = H[1];
    for (int j = 0; j < n; ++j) {
        alpha[j] = 0.0;
        beta[j] = 0.0;
        ascale[j] = 0;
    }
    int ziter = (p >= std::log2(unfl) / std::log2(ulp)) ? -1 : 0;  // :107
    for (int c = 1; c <= n; ++c)                                     // :112 triu!(Hs[j-1], -1)
        for (int l = 2; l <= p; ++l)
            for (int r = c + 2; r <= n; ++r) H[l](r, c) = 0.0;
    std::vector<GivR> Gt(n + 2);
    std::vector<cplx> v4ev(std::max(p - 1, 1));
    std::vector<char> S2(p);  // circshift(S, -1): S2[l-1] = S[l+1], last = S[1]
    for (int l = 1; l <= p; ++l) S2[l - 1] = S[(l % p) + 1];
    int ilast = n, ifirst = 1, ifirstm = 1, ilastm = n;
    const long maxit = (long)maxitfac * n;
    bool done = (n == 0);
    auto Zr = [&](int l, const GivR& G) {  // rmul!(Z[l], G')
        if (wantZ) rmulGT<double>(Z[l], 1, n, G.adj());
    };
    for (long jiter = 1; jiter <= maxit && !done; ++jiter) {
        if (log) log->niter = jiter;
        bool split1block = false, deflate_pos = false, deflate_neg = false, doqziter = true;
        int ldeflate = -1, jdeflate = -1, jlo = 1;
        do {
            if (ilast == 1) {
                split1block = true;
                break;
            }
            split1block = rg_check_deflate_hess(H1, 1, ilast, ulp, smlnum, jlo);
            if (split1block) break;
            for (int l = 2; l <= p; ++l)  // Test 2
                if (S[l]) {
                    int jx;
                    if (rg_check_deflate_tr(H[l], jlo, ilast, ulp, smlnum, jx)) {
                        deflate_pos = true;
                        ldeflate = l;
                        jdeflate = jx;
                        break;
                    }
                }
            for (int l = 2; l <= p; ++l)  // Test 3 (as the reference, this may overwrite ldeflate/jdeflate of Test 2;
                if (!S[l]) {               // Case II is dispatched first at :329, with the overwritten indices — see below)
                    int jx;
                    if (rg_check_deflate_tr(H[l], jlo, ilast, ulp, smlnum, jx)) {
                        deflate_neg = true;
                        if (!deflate_pos) {
                            ldeflate = l;
                            jdeflate = jx;
                        }
                        break;
                    }
                }
            if (ziter >= 7 || ziter < 0) {  // Test 4: controlled zero shift :229-324
                if (log) log->nzero++;
                for (int j = jlo; j <= ilast - 1; ++j) {
                    double c, s, r;
                    rgiv(H1(j, j), H1(j + 1, j), c, s, r);
                    H1(j, j) = r;
                    H1(j + 1, j) = 0.0;
                    GivR G{j, j + 1, c, s};
                    lmulGT<double>(G, H1, j + 1, ilastm);
                    Gt[j] = G;
                }
                for (int j = jlo; j <= ilast - 1; ++j) Zr(1, Gt[j]);
                for (int l = p; l >= 2; --l) {
                    MatR& Hl = H[l];
                    for (int j = jlo; j <= ilast - 1; ++j) {
                        GivR G = Gt[j];
                        if (G.s == 0) continue;
                        if (S[l]) rmulGT<double>(Hl, ifirstm, j + 1, G.adj());
                        else lmulGT<double>(G, Hl, j, ilastm);
                        double tol = std::fabs(Hl(j, j)) + std::fabs(Hl(j + 1, j + 1));
                        if (tol == 0) tol = opnorm1r(Hl, jlo, j + 1, jlo, j + 1, false);
                        tol = std::max(ulp * tol, smlnum);
                        if (std::fabs(Hl(j + 1, j)) <= tol) {
                            Hl(j + 1, j) = 0.0;
                            Gt[j] = GivR{j, j + 1, 1.0, 0.0};
                        } else if (S[l]) {
                            double c, s, r;
                            rgiv(Hl(j, j), Hl(j + 1, j), c, s, r);
                            Hl(j, j) = r;
                            Hl(j + 1, j) = 0.0;
                            GivR G2{j, j + 1, c, s};
                            lmulGT<double>(G2, Hl, j + 1, ilastm);
                            Gt[j] = G2;
                        } else {
                            double c, s, r;
                            rgiv(Hl(j + 1, j + 1), Hl(j + 1, j), c, s, r);
                            Hl(j + 1, j + 1) = r;
                            Hl(j + 1, j) = 0.0;
                            GivR G2{j + 1, j, c, s};
                            rmulGT<double>(Hl, ifirstm, j, G2.adj());
                            Gt[j] = GivR{j, j + 1, c, -s};
                        }
                    }
                    for (int j = jlo; j <= ilast - 1; ++j) Zr(l, Gt[j]);
                }
                ziter = 0;
                for (int j = jlo; j <= ilast - 1; ++j) {
                    rmulGT<double>(H1, ifirstm, j + 1, Gt[j].adj());
                    if (Gt[j].s == 0) ziter = 1;
                }
                doqziter = false;
            }
        } while (false);

        // As in the complex oracle: when the zero-shift block ran this iteration the matrices Test 2/3 looked at have
        // changed, and the reference's fall-through into Case II/III with the stale (ldeflate, jdeflate) destroys the
        // decomposition; the deflation is taken up by the tests of the next iteration instead.
        if (!doqziter && !split1block) {
            deflate_pos = false;
            deflate_neg = false;
        }

        if (deflate_pos) {  // Case II :329-442
            if (log) log->ncase2++;
            for (int j = jlo; j <= jdeflate - 1; ++j) {
                double c, s, r;
                rgiv(H1(j, j), H1(j + 1, j), c, s, r);
                H1(j, j) = r;
                H1(j + 1, j) = 0.0;
                GivR G{j, j + 1, c, s};
                lmulGT<double>(G, H1, j + 1, ilastm);
                Gt[j] = G;
            }
            for (int j = jlo; j <= jdeflate - 1; ++j) Zr(1, Gt[j]);
            for (int l = p; l >= 2; --l) {
                const int ntra = (l < ldeflate) ? (jdeflate - 2) : (jdeflate - 1);
                MatR& Hl = H[l];
                for (int j = jlo; j <= ntra; ++j) {
                    double c, s, r;
                    if (S[l]) {
                        rmulGT<double>(Hl, ifirstm, j + 1, Gt[j].adj());
                        rgiv(Hl(j, j), Hl(j + 1, j), c, s, r);
                        Hl(j, j) = r;
                        Hl(j + 1, j) = 0.0;
                        GivR G{j, j + 1, c, s};
                        lmulGT<double>(G, Hl, j + 1, ilastm);
                        Gt[j] = G;
                    } else {
                        lmulGT<double>(Gt[j], Hl, j, ilastm);
                        rgiv(Hl(j + 1, j + 1), Hl(j + 1, j), c, s, r);
                        Hl(j + 1, j + 1) = r;
                        Hl(j + 1, j) = 0.0;
                        GivR G{j + 1, j, c, s};
                        rmulGT<double>(Hl, ifirstm, j, G.adj());
                        Gt[j] = GivR{j, j + 1, c, -s};
                    }
                }
                for (int j = jlo; j <= ntra; ++j) Zr(l, Gt[j]);
            }
            for (int j = jlo; j <= jdeflate - 2; ++j) rmulGT<double>(H1, ifirstm, j + 1, Gt[j].adj());
            // second unshifted step, from the bottom
            for (int j = ilast; j >= jdeflate + 1; --j) {
                double c, s, r;
                rgiv(H1(j, j), H1(j, j - 1), c, s, r);
                H1(j, j) = r;
                H1(j, j - 1) = 0.0;
                GivR G{j, j - 1, c, s};
                rmulGT<double>(H1, ifirstm, j - 1, G.adj());
                Gt[j] = GivR{j - 1, j, c, -s};
            }
            for (int j = ilast; j >= jdeflate + 1; --j) Zr(p >= 2 ? 2 : 1, Gt[j]);
            for (int l = 2; l <= p; ++l) {
                const int ntra = (l > ldeflate) ? (jdeflate + 2) : (jdeflate + 1);
                MatR& Hl = H[l];
                for (int j = ilast; j >= ntra; --j) {
                    double c, s, r;
                    if (!S[l]) {
                        rmulGT<double>(Hl, ifirstm, j, Gt[j].adj());
                        rgiv(Hl(j - 1, j - 1), Hl(j, j - 1), c, s, r);
                        Hl(j - 1, j - 1) = r;
                        Hl(j, j - 1) = 0.0;
                        GivR G{j - 1, j, c, s};
                        lmulGT<double>(G, Hl, j, ilastm);
                        Gt[j] = G;
                    } else {
                        lmulGT<double>(Gt[j], Hl, j - 1, ilastm);
                        rgiv(Hl(j, j), Hl(j, j - 1), c, s, r);
                        Hl(j, j) = r;
                        Hl(j, j - 1) = 0.0;
                        GivR G{j, j - 1, c, s};
                        rmulGT<double>(Hl, ifirstm, j - 1, G.adj());
                        Gt[j] = GivR{j - 1, j, c, -s};
                    }
                }
                const int ln = (l % p) + 1;
                for (int j = ilast; j >= ntra; --j) Zr(ln, Gt[j]);
            }
            for (int j = ilast; j >= jdeflate + 2; --j) lmulGT<double>(Gt[j], H1, j - 1, ilastm);
            doqziter = false;
        } else if (deflate_neg) {  // Case III :444-616
            if (log) log->ncase3++;
            MatR& Hd = H[ldeflate];
            if (jdeflate > (ilast - jlo + 1) / 2.0) {
                for (int j1 = jdeflate; j1 <= ilast - 1; ++j1) {
                    int j = j1;
                    double c, s, r;
                    rgiv(Hd(j, j + 1), Hd(j + 1, j + 1), c, s, r);
                    Hd(j, j + 1) = r;
                    Hd(j + 1, j + 1) = 0.0;
                    GivR G{j, j + 1, c, s};
                    lmulGT<double>(G, Hd, j + 2, ilastm);
                    int ln = (ldeflate % p) + 1;
                    Zr(ln, G);
                    for (int l = 1; l <= p - 1; ++l) {
                        if (ln == 1) {
                            lmulGT<double>(G, H1, j - 1, ilastm);
                            rgiv(H1(j + 1, j), H1(j + 1, j - 1), c, s, r);
                            H1(j + 1, j) = r;
                            H1(j + 1, j - 1) = 0.0;
                            GivR Gb{j, j - 1, c, s};
                            rmulGT<double>(H1, ifirstm, j, Gb.adj());
                            G = GivR{j - 1, j, c, -s};
                            j -= 1;
                        } else if (S[ln]) {
                            MatR& Hn = H[ln];
                            lmulGT<double>(G, Hn, j, ilastm);
                            rgiv(Hn(j + 1, j + 1), Hn(j + 1, j), c, s, r);
                            Hn(j + 1, j + 1) = r;
                            Hn(j + 1, j) = 0.0;
                            GivR Gb{j + 1, j, c, s};
                            rmulGT<double>(Hn, ifirstm, j, Gb.adj());
                            G = GivR{j, j + 1, c, -s};
                        } else {
                            MatR& Hn = H[ln];
                            rmulGT<double>(Hn, ifirstm, j + 1, G.adj());
                            rgiv(Hn(j, j), Hn(j + 1, j), c, s, r);
                            Hn(j, j) = r;
                            Hn(j + 1, j) = 0.0;
                            G = GivR{j, j + 1, c, s};
                            lmulGT<double>(G, Hn, j + 1, ilastm);
                        }
                        ln = (ln % p) + 1;
                        Zr(ln, G);
                    }
                    rmulGT<double>(Hd, ifirstm, j, G.adj());
                }
                int j = ilast;
                double c, s, r;
                rgiv(H1(j, j), H1(j, j - 1), c, s, r);
                H1(j, j) = r;
                H1(j, j - 1) = 0.0;
                GivR Gb{j, j - 1, c, s};
                rmulGT<double>(H1, ifirstm, j - 1, Gb.adj());
                GivR G{j - 1, j, c, -s};
                Zr(2, G);
                for (int l = 2; l <= ldeflate - 1; ++l) {
                    MatR& Hl = H[l];
                    if (!S[l]) {
                        rmulGT<double>(Hl, ifirstm, j, G.adj());
                        rgiv(Hl(j - 1, j - 1), Hl(j, j - 1), c, s, r);
                        Hl(j - 1, j - 1) = r;
                        Hl(j, j - 1) = 0.0;
                        G = GivR{j - 1, j, c, s};
                        lmulGT<double>(G, Hl, j, ilastm);
                    } else {
                        lmulGT<double>(G, Hl, j - 1, ilastm);
                        rgiv(Hl(j, j), Hl(j, j - 1), c, s, r);
                        Hl(j, j) = r;
                        Hl(j, j - 1) = 0.0;
                        GivR Gc{j, j - 1, c, s};
                        rmulGT<double>(Hl, ifirstm, j - 1, Gc.adj());
                        G = GivR{j - 1, j, c, -s};
                    }
                    Zr((l % p) + 1, G);
                }
                rmulGT<double>(Hd, ifirstm, j, G.adj());
            } else {
                for (int j1 = jdeflate; j1 >= jlo + 1; --j1) {
                    int j = j1;
                    double c, s, r;
                    rgiv(Hd(j - 1, j), Hd(j - 1, j - 1), c, s, r);
                    Hd(j - 1, j) = r;
                    Hd(j - 1, j - 1) = 0.0;
                    GivR Gb{j, j - 1, c, s};
                    rmulGT<double>(Hd, ifirstm, j - 2, Gb.adj());
                    GivR G{j - 1, j, c, -s};
                    Zr(ldeflate, G);
                    int ln = ldeflate - 1;
                    for (int l = 1; l <= p - 1; ++l) {
                        MatR& Hn = H[ln];
                        if (ln == 1) {
                            rmulGT<double>(Hn, ifirstm, j + 1, G.adj());
                            rgiv(Hn(j, j - 1), Hn(j + 1, j - 1), c, s, r);
                            Hn(j, j - 1) = r;
                            Hn(j + 1, j - 1) = 0.0;
                            G = GivR{j, j + 1, c, s};
                            lmulGT<double>(G, Hn, j, ilastm);
                            j += 1;
                        } else if (!S[ln]) {
                            lmulGT<double>(G, Hn, j - 1, ilastm);
                            rgiv(Hn(j, j), Hn(j, j - 1), c, s, r);
                            Hn(j, j) = r;
                            Hn(j, j - 1) = 0.0;
                            GivR Gc{j, j - 1, c, s};
                            rmulGT<double>(Hn, ifirstm, j - 1, Gc.adj());
                            G = GivR{j - 1, j, c, -s};
                        } else {
                            rmulGT<double>(Hn, ifirstm, j, G.adj());
                            rgiv(Hn(j - 1, j - 1), Hn(j, j - 1), c, s, r);
                            Hn(j - 1, j - 1) = r;
                            Hn(j, j - 1) = 0.0;
                            G = GivR{j - 1, j, c, s};
                            lmulGT<double>(G, Hn, j, ilastm);
                        }
                        Zr(ln, G);
                        ln = (ln == 1) ? p : (ln - 1);
                    }
                    lmulGT<double>(G, Hd, j, ilastm);
                }
                int j = jlo;
                double c, s, r;
                rgiv(H1(j, j), H1(j + 1, j), c, s, r);
                H1(j, j) = r;
                H1(j + 1, j) = 0.0;
                GivR G{j, j + 1, c, s};
                lmulGT<double>(G, H1, j + 1, ilastm);
                Zr(1, G);
                for (int l = p; l >= ldeflate + 1; --l) {
                    MatR& Hl = H[l];
                    if (S[l]) {
                        rmulGT<double>(Hl, ifirstm, j + 1, G.adj());
                        rgiv(Hl(j, j), Hl(j + 1, j), c, s, r);
                        Hl(j, j) = r;
                        Hl(j + 1, j) = 0.0;
                        G = GivR{j, j + 1, c, s};
                        lmulGT<double>(G, Hl, j + 1, ilastm);
                    } else {
                        lmulGT<double>(G, Hl, j, ilastm);
                        rgiv(Hl(j + 1, j + 1), Hl(j + 1, j), c, s, r);
                        Hl(j + 1, j + 1) = r;
                        Hl(j + 1, j) = 0.0;
                        GivR Gc{j + 1, j, c, s};
                        rmulGT<double>(Hl, ifirstm, j, Gc.adj());
                        G = GivR{j, j + 1, c, -s};
                    }
                    Zr(l, G);
                }
                lmulGT<double>(G, Hd, j + 1, ilastm);
            }
            doqziter = false;
        } else if (split1block) {  // :617-642
            if (log) log->ndefl1++;
            for (int l = 1; l <= p - 1; ++l) v4ev[l - 1] = H[l + 1](ilast, ilast);
            cplx a;
            double b;
            int sc;
            safeprod(S, p, cplx(H1(ilast, ilast)), v4ev.data(), a, b, sc);
            alpha[ilast - 1] = a;
            beta[ilast - 1] = b;
            ascale[ilast - 1] = sc;
            ilast -= 1;
            if (ilast < 1) {
                done = true;
                break;
            }
            if (ziter != -1) ziter = 0;
            if (!wantT) {
                ilastm = ilast;
                if (ifirstm > ilast) ifirstm = 1;
            }
            doqziter = false;
        } else if (doqziter) {
            ifirst = jlo;
        }

        if (!doqziter) continue;
        ziter += 1;
        if (!wantT) ifirstm = ifirst;
        if (ifirst + 1 == ilast) {  // 2x2 block :661-790
            int j = ilast - 1;
            std::vector<R2> X(p);
            auto blk = [&](const MatR& M) { return R2{M(j, j), M(j, j + 1), M(j + 1, j), M(j + 1, j + 1)}; };
            for (int l = 1; l <= p; ++l) X[l - 1] = blk(l == p ? H1 : H[l + 1]);  // order 2,3,...,p,1
            bool done2x2 = false;
            int titer = 0;
            while (!done2x2 && titer < 2) {
                titer += 1;
                rg_rp2x2ssr(p, X, S2);
                const R2& Xp = X[p - 1];
                if (std::fabs(Xp.c) < ulp * std::max(std::fabs(Xp.a), std::max(std::fabs(Xp.b), std::fabs(Xp.d)))) {
                    done2x2 = true;
                    if (log) log->n2x2real++;
                    double c1 = 1.0, s1 = 1.0, r;
                    for (int l = p; l >= 2; --l) {
                        const double rr = X[l - 2].d;
                        if (S[l]) rgiv(c1 * H[l](j, j), s1 * rr, c1, s1, r);
                        else rgiv(c1 * rr, s1 * H[l](j, j), c1, s1, r);
                    }
                    const double rr = X[p - 1].d;
                    rgiv(c1 * H1(j, j) - rr * s1, c1 * H1(j + 1, j), c1, s1, r);
                    GivR G2{j, j + 1, c1, s1};
                    lmulGT<double>(G2, H1, j, ilastm);
                    Zr(1, G2);
                    for (int l = p; l >= 2; --l) {
                        MatR& Hl = H[l];
                        double r1;
                        if (S[l]) {
                            rmulGT<double>(Hl, ifirstm, j + 1, G2.adj());
                            rgiv(Hl(j, j), Hl(j + 1, j), c1, s1, r1);
                            Hl(j, j) = r1;
                            Hl(j + 1, j) = 0.0;
                            G2 = GivR{j, j + 1, c1, s1};
                            lmulGT<double>(G2, Hl, j + 1, ilastm);
                        } else {
                            lmulGT<double>(G2, Hl, j, ilastm);
                            rgiv(Hl(j + 1, j + 1), -Hl(j + 1, j), c1, s1, r1);
                            Hl(j + 1, j + 1) = r1;
                            Hl(j + 1, j) = 0.0;
                            G2 = GivR{j, j + 1, c1, s1};
                            rmulGT<double>(Hl, ifirstm, j, G2.adj());
                        }
                        Zr(l, G2);
                    }
                    rmulGT<double>(H1, ifirstm, ilastm, G2.adj());
                }
            }
            if (!done2x2) {  // complex pair :748-790
                if (log) log->n2x2cplx++;
                std::vector<SM> Xs(p, SM(2, 2));
                std::vector<char> Sx(p);
                for (int l = 1; l <= p; ++l) {
                    for (int a = 0; a < 2; ++a)
                        for (int c = 0; c < 2; ++c) Xs[l - 1](a, c) = H[l](j + a, j + c);
                    Sx[l - 1] = S[l];
                }
                cplx a2[2];
                double sc2[2], b2[2];
                bool cvg, good;
                rpeigvals2x2(p, Xs, a2, sc2, cvg, good, Sx.data(), b2);
                for (int q = 0; q < 2; ++q) {
                    alpha[j - 1 + q] = a2[q];
                    beta[j - 1 + q] = b2[q];
                    ascale[j - 1 + q] = (int)sc2[q];
                }
                if (log) {
                    if (!cvg) log->iwarn = std::max(log->iwarn, j);
                    else if (!good && log->iwarn == 0) log->iwarn = n;
                }
                ilast = ifirst - 1;
                if (ilast < 1) done = true;
                if (ziter != -1) ziter = 0;
                if (!wantT) {
                    ilastm = ilast;
                    if (ifirstm > ilast) ifirstm = 1;
                }
            }
            continue;
        }
        // implicit double-shift sweep :792-1054
        if (log) log->nsweeps++;
        double c1, s1, c2, s2, r;
        rg_qzrots(p, H, S, ifirst, ilast - ifirst + 1, c1, s1, c2, s2);
        int i1, i2, j;
        GivR G1, G2;
        if (p > 1) {
            i1 = ifirst + 1;
            i2 = ilast - 2;
            j = ifirst;
            G1 = GivR{j + 1, j + 2, c2, s2};
            G2 = GivR{j, j + 1, c1, s1};
            rmulGT<double>(H1, ifirstm, ilast, G1.adj());
            rmulGT<double>(H1, ifirstm, ilast, G2.adj());
            Zr(2, G1);
            Zr(2, G2);
            for (int l = 2; l <= p; ++l) {
                MatR& Hl = H[l];
                if (S[l]) {
                    lmulGT<double>(G1, Hl, j, ilastm);
                    rgiv(Hl(j + 2, j + 2), -Hl(j + 2, j + 1), c2, s2, r);
                    Hl(j + 2, j + 2) = r;
                    Hl(j + 2, j + 1) = 0.0;
                    G1 = GivR{j + 1, j + 2, c2, s2};
                    rmulGT<double>(Hl, ifirstm, j + 1, G1.adj());
                    lmulGT<double>(G2, Hl, j, ilastm);
                    rgiv(Hl(j + 1, j + 1), -Hl(j + 1, j), c1, s1, r);
                    Hl(j + 1, j + 1) = r;
                    Hl(j + 1, j) = 0.0;
                    G2 = GivR{j, j + 1, c1, s1};
                    rmulGT<double>(Hl, ifirstm, j, G2.adj());
                } else {
                    rmulGT<double>(Hl, ifirstm, j + 2, G1.adj());
                    rgiv(Hl(j + 1, j + 1), Hl(j + 2, j + 1), c2, s2, r);
                    Hl(j + 1, j + 1) = r;
                    Hl(j + 2, j + 1) = 0.0;
                    G1 = GivR{j + 1, j + 2, c2, s2};
                    lmulGT<double>(G1, Hl, j + 2, ilastm);
                    rmulGT<double>(Hl, ifirstm, j + 1, G2.adj());
                    rgiv(Hl(j, j), Hl(j + 1, j), c1, s1, r);
                    Hl(j, j) = r;
                    Hl(j + 1, j) = 0.0;
                    G2 = GivR{j, j + 1, c1, s1};
                    lmulGT<double>(G2, Hl, j + 1, ilastm);
                }
                const int ln = (l % p) + 1;
                Zr(ln, G1);
                Zr(ln, G2);
            }
            lmulGT<double>(G1, H1, ifirst, ilastm);
            lmulGT<double>(G2, H1, ifirst, ilastm);
        } else {
            i1 = ifirst - 1;
            i2 = ilast - 3;
            j = ifirst;  // the reference leaves j undefined here and uses jt; p == 1 keeps j running below
            G1 = GivR{j + 1, j + 2, c2, s2};
            G2 = GivR{j, j + 1, c1, s1};
            j = ifirst - 1;
        }
        for (int j1 = i1; j1 <= i2; ++j1) {
            if (j1 < ifirst) {
                j = j1 + 1;
                lmulGT<double>(G1, H1, j, ilastm);
                lmulGT<double>(G2, H1, j, ilastm);
            } else {
                j = (p == 1) ? (j + 1) : j1;
                double r2, r1;
                rgiv(H1(j + 1, j - 1), H1(j + 2, j - 1), c2, s2, r2);
                rgiv(H1(j, j - 1), r2, c1, s1, r1);
                H1(j, j - 1) = r1;
                H1(j + 1, j - 1) = 0.0;
                H1(j + 2, j - 1) = 0.0;
                G1 = GivR{j + 1, j + 2, c2, s2};
                G2 = GivR{j, j + 1, c1, s1};
                lmulGT<double>(G1, H1, j, ilastm);
                lmulGT<double>(G2, H1, j, ilastm);
            }
            Zr(1, G1);
            Zr(1, G2);
            for (int l = p; l >= 2; --l) {
                MatR& Hl = H[l];
                double r2, r1;
                if (S[l]) {
                    rmulGT<double>(Hl, ifirstm, j + 2, G1.adj());
                    rgiv(Hl(j + 1, j + 1), Hl(j + 2, j + 1), c2, s2, r2);
                    Hl(j + 1, j + 1) = r2;
                    Hl(j + 2, j + 1) = 0.0;
                    G1 = GivR{j + 1, j + 2, c2, s2};
                    lmulGT<double>(G1, Hl, j + 2, ilastm);
                    rmulGT<double>(Hl, ifirstm, j + 1, G2.adj());
                    rgiv(Hl(j, j), Hl(j + 1, j), c1, s1, r1);
                    Hl(j, j) = r1;
                    Hl(j + 1, j) = 0.0;
                    G2 = GivR{j, j + 1, c1, s1};
                    lmulGT<double>(G2, Hl, j + 1, ilastm);
                } else {
                    lmulGT<double>(G1, Hl, j, ilastm);
                    rgiv(Hl(j + 2, j + 2), -Hl(j + 2, j + 1), c2, s2, r2);
                    Hl(j + 2, j + 2) = r2;
                    Hl(j + 2, j + 1) = 0.0;
                    G1 = GivR{j + 1, j + 2, c2, s2};
                    rmulGT<double>(Hl, ifirstm, j + 1, G1.adj());
                    lmulGT<double>(G2, Hl, j, ilastm);
                    rgiv(Hl(j + 1, j + 1), -Hl(j + 1, j), c1, s1, r1);
                    Hl(j + 1, j + 1) = r1;
                    Hl(j + 1, j) = 0.0;
                    G2 = GivR{j, j + 1, c1, s1};
                    rmulGT<double>(Hl, ifirstm, j, G2.adj());
                }
                Zr(l, G1);
                Zr(l, G2);
            }
            const int lm = std::min(j + 3, ilastm);
            rmulGT<double>(H1, ifirstm, lm, G1.adj());
            rmulGT<double>(H1, ifirstm, lm, G2.adj());
        }
        j = ilast - 1;
        {
            double r1;
            rgiv(H1(j, j - 1), H1(j + 1, j - 1), c1, s1, r1);
            H1(j, j - 1) = r1;
            H1(j + 1, j - 1) = 0.0;
            G2 = GivR{j, j + 1, c1, s1};
            lmulGT<double>(G2, H1, j, ilastm);
            Zr(1, G2);
            for (int l = p; l >= 2; --l) {
                MatR& Hl = H[l];
                if (S[l]) {
                    rmulGT<double>(Hl, ifirstm, j + 1, G2.adj());
                    rgiv(Hl(j, j), Hl(j + 1, j), c1, s1, r1);
                    Hl(j, j) = r1;
                    Hl(j + 1, j) = 0.0;
                    G2 = GivR{j, j + 1, c1, s1};
                    lmulGT<double>(G2, Hl, j + 1, ilastm);
                } else {
                    lmulGT<double>(G2, Hl, j, ilastm);
                    rgiv(Hl(j + 1, j + 1), -Hl(j + 1, j), c1, s1, r1);
                    Hl(j + 1, j + 1) = r1;
                    Hl(j + 1, j) = 0.0;
                    G2 = GivR{j, j + 1, c1, s1};
                    rmulGT<double>(Hl, ifirstm, j, G2.adj());
                }
                Zr(l, G2);
            }
            rmulGT<double>(H1, ifirstm, ilastm, G2.adj());
        }
    }
    if (!done) return 1000000 + ilast;
    return 0;
}

}  // namespace psdo
