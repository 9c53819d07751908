// ORACLE — TEST INFRASTRUCTURE ONLY.
//
// CPU restatement (C++17, serial, no BLAS) of the *real standard* periodic Schur path of
// RalphAS/PeriodicSchurDecompositions.jl v0.1.6.  It follows the reference's algorithm and loop
// order function by function (citations are `file:line` under /root/reference/src).
//
// Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load the library built
// from this directory.  The product (periodicschurdecompositions.jl_amd/) never includes, links
// or calls anything here.
//
// Parity pinning: the reference is pure Julia and no Julia toolchain exists in the build or GPU
// containers, so the reference itself cannot be executed.  This restatement is pinned by the
// reference's own known-answer fixtures and property suites (tests/test_oracle_*.py,
// tests/golden/) — see DESIGN.md "Oracle".
//
// Deliberate deviations from the reference text (documented reference defects):
//   * PSD.jl:970 writes xi[2] (the scaled reflector entry) where beta = xi[1] is meant; we write beta.
//   * PSD.jl:1048 compares against lambda[1], lambda[2]; lambda[i-1], lambda[i] is meant; we use those.
#pragma once
#include <algorithm>
#include <cmath>
#include <complex>
#include <cstdint>
#include <cstring>
#include <limits>
#include <vector>

namespace psdo {

// Column-major n x n matrix view with 1-based element access, mirroring Julia indexing so that
// the restatement can be checked line by line against the reference.
struct MatD {
    double* a;
    int ld;
    inline double& operator()(int r, int c) const { return a[(size_t)(c - 1) * ld + (r - 1)]; }
};

// ---------------------------------------------------------------------------------------------
// householder.jl:5-24  _norm2 (real): scaled sum of squares
#ifndef PSDO_OMP_MIN_WORK
#define PSDO_OMP_MIN_WORK (1LL << 17)  // elements of a panel below which a Householder application stays on one thread
#endif
// bounded samples for the CPU baseline of bench.py: stop pschur_hess after this many QR sweeps (< 0: no cap)
static long long g_sweep_cap = -1;
static long long g_sweeps_done = 0;

inline double norm2(const double* x, int n, int inc) {
    if (n < 1) return 0.0;
    if (n == 1) return std::fabs(x[0]);
    double scale = 0.0, ssq = 0.0;
    for (int k = 0; k < n; ++k) {
        double xi = x[(size_t)k * inc];
        if (xi != 0.0) {
            double a = std::fabs(xi);
            if (scale < a) {
                double q = scale / a;
                ssq = 1.0 + ssq * q * q;
                scale = a;
            } else {
                double q = a / scale;
                ssq += q * q;
            }
        }
    }
    return scale * std::sqrt(ssq);
}

// householder.jl:66-108  _xreflector! (real, xLARFG): x <- (beta, v2..vn); returns tau
inline double xreflector(double* x, int n, int inc) {
    if (n <= 1) return 0.0;
    const double sfmin = 2.0 * std::numeric_limits<double>::min() / std::numeric_limits<double>::epsilon();
    double alpha = x[0];
    double xnorm = norm2(x + inc, n - 1, inc);
    if (xnorm == 0.0) return 0.0;
    double beta = -std::copysign(std::hypot(alpha, xnorm), alpha);
    int kount = 0;
    bool smallb = std::fabs(beta) < sfmin;
    if (smallb) {
        const double rsfmin = 1.0 / sfmin;
        while (smallb) {
            kount += 1;
            for (int j = 1; j < n; ++j) x[(size_t)j * inc] *= rsfmin;
            beta *= rsfmin;
            alpha *= rsfmin;
            smallb = (std::fabs(beta) < sfmin) && (kount < 20);
        }
        xnorm = norm2(x + inc, n - 1, inc);
        beta = -std::copysign(std::hypot(alpha, xnorm), alpha);
    }
    double tau = (beta - alpha) / beta;
    double t = 1.0 / (alpha - beta);
    for (int j = 1; j < n; ++j) x[(size_t)j * inc] *= t;
    for (int j = 0; j < kount; ++j) beta *= sfmin;
    x[0] = beta;
    return tau;
}

// householder.jl:222-237  lmul!(H', A): A is m x ncols starting at (r0,c0); H = I - tau [1;v][1;v]'
// v has m-1 entries with stride vinc.
inline void lmul_Hadj(const double* v, int vinc, double tau, const MatD& A, int r0, int c0, int m,
                      int ncols) {
    // columns are independent: the all-core CPU baseline (bench.py cpu_baseline.allcore) threads them, the way a
    // threaded BLAS would thread the reference's gemv / ger! (householder.jl:215,252); one thread by default
#pragma omp parallel for schedule(static) if ((long long)m * ncols >= PSDO_OMP_MIN_WORK)
    for (int j = 0; j < ncols; ++j) {
        double va = A(r0, c0 + j);
        for (int r = 1; r < m; ++r) va += v[(size_t)(r - 1) * vinc] * A(r0 + r, c0 + j);
        va = tau * va;
        A(r0, c0 + j) -= va;
        for (int r = 1; r < m; ++r) A(r0 + r, c0 + j) -= va * v[(size_t)(r - 1) * vinc];
    }
}

// householder.jl:190-205  lmul!(H, A) — identical to lmul!(H',A) for real tau
inline void lmul_H(const double* v, int vinc, double tau, const MatD& A, int r0, int c0, int m,
                   int ncols) {
    lmul_Hadj(v, vinc, tau, A, r0, c0, m, ncols);
}

// householder.jl:207-220  rmul!(A, H): A is nrows x m starting at (r0,c0)
inline void rmul_H(const MatD& A, int r0, int c0, int nrows, int m, const double* v, int vinc,
                   double tau) {
    // x = a1 + A1 v column by column (the reference's gemv, householder.jl:215), then the rank-one update column by
    // column (ger!, :218,252).  Per row the floating-point operations and their order are those of the row-by-row
    // form x = A[r,1] + sum_c A[r,c] v[c]; A[r,c] -= (tau x) v[c], so results are bit-identical to it; the column
    // order only makes the pass over a large matrix stream through memory.  Row chunks are independent (threaded for
    // the all-core baseline, see lmul_Hadj).
    if (nrows <= 4) {
        for (int r = 0; r < nrows; ++r) {
            double x = A(r0 + r, c0);
            for (int c = 1; c < m; ++c) x += A(r0 + r, c0 + c) * v[(size_t)(c - 1) * vinc];
            A(r0 + r, c0) -= tau * x;
            for (int c = 1; c < m; ++c) A(r0 + r, c0 + c) -= tau * x * v[(size_t)(c - 1) * vinc];
        }
        return;
    }
    const int CH = 512;  // rows per chunk: x stays in L1
#pragma omp parallel for schedule(static) if ((long long)m * nrows >= PSDO_OMP_MIN_WORK)
    for (int rb = 0; rb < nrows; rb += CH) {
        const int nr = (nrows - rb < CH) ? (nrows - rb) : CH;
        double x[CH];
        double* a1 = &A(r0 + rb, c0);
        for (int r = 0; r < nr; ++r) x[r] = a1[r];
        for (int c = 1; c < m; ++c) {
            const double vc = v[(size_t)(c - 1) * vinc];
            const double* ac = &A(r0 + rb, c0 + c);
            for (int r = 0; r < nr; ++r) x[r] += ac[r] * vc;
        }
        for (int r = 0; r < nr; ++r) {
            a1[r] -= tau * x[r];
            x[r] = tau * x[r];
        }
        for (int c = 1; c < m; ++c) {
            const double vc = v[(size_t)(c - 1) * vinc];
            double* ac = &A(r0 + rb, c0 + c);
            for (int r = 0; r < nr; ++r) ac[r] -= x[r] * vc;
        }
    }
}

// householder.jl:269-304  HH2: 2-vector reflector with explicit (v1, v2, tau)
struct HH2 {
    double v1, v2, tau;
};
inline void rmul_HH2(const MatD& A, int r0, int c0, int nrows, const HH2& h) {
    const double t1 = h.v1 * h.tau, t2 = h.v2 * h.tau;
    for (int r = 0; r < nrows; ++r) {
        double s = A(r0 + r, c0) * h.v1 + A(r0 + r, c0 + 1) * h.v2;
        A(r0 + r, c0) -= s * t1;
        A(r0 + r, c0 + 1) -= s * t2;
    }
}
inline void lmul_HH2adj(const HH2& h, const MatD& A, int r0, int c0, int ncols) {
    const double t1 = h.tau * h.v1, t2 = h.tau * h.v2;
    for (int j = 0; j < ncols; ++j) {
        double s = h.v1 * A(r0, c0 + j) + h.v2 * A(r0 + 1, c0 + j);
        A(r0, c0 + j) -= s * t1;
        A(r0 + 1, c0 + j) -= s * t2;
    }
}

// stdlib LinearAlgebra.givensAlgorithm(f::Float64, g::Float64) (port of LAPACK dlartg, as the
// reference imports it at PSD.jl:9): returns (c, s, r) with [c s; -s c][f; g] = [r; 0].
inline void givens_algorithm(double f, double g, double& cs, double& sn, double& r) {
    const double eps = std::numeric_limits<double>::epsilon();
    const double safmin = std::numeric_limits<double>::min();
    const double safmn2 = std::pow(2.0, std::trunc(std::log(safmin / eps) / std::log(2.0) / 2.0));
    const double safmx2 = 1.0 / safmn2;
    if (g == 0.0) {
        cs = 1.0; sn = 0.0; r = f;
    } else if (f == 0.0) {
        cs = 0.0; sn = 1.0; r = g;
    } else {
        double f1 = f, g1 = g;
        double scale = std::max(std::fabs(f1), std::fabs(g1));
        if (scale >= safmx2) {
            int count = 0;
            while (true) {
                count += 1;
                f1 *= safmn2; g1 *= safmn2;
                scale = std::max(std::fabs(f1), std::fabs(g1));
                if (scale < safmx2 || count >= 20) break;
            }
            r = std::sqrt(f1 * f1 + g1 * g1);
            cs = f1 / r; sn = g1 / r;
            for (int i = 0; i < count; ++i) r *= safmx2;
        } else if (scale <= safmn2) {
            int count = 0;
            while (true) {
                count += 1;
                f1 *= safmx2; g1 *= safmx2;
                scale = std::max(std::fabs(f1), std::fabs(g1));
                if (scale > safmn2) break;
            }
            r = std::sqrt(f1 * f1 + g1 * g1);
            cs = f1 / r; sn = g1 / r;
            for (int i = 0; i < count; ++i) r *= safmn2;
        } else {
            r = std::sqrt(f1 * f1 + g1 * g1);
            cs = f1 / r; sn = g1 / r;
        }
        if (std::fabs(f) > std::fabs(g) && cs < 0.0) {
            cs = -cs; sn = -sn; r = -r;
        }
    }
}

// stdlib lmul!(G::Givens, A) on rows (r1, r1+1): a1 <- c a1 + s a2 ; a2 <- -s a1 + c a2
inline void lmul_G(double c, double s, const MatD& A, int r1, int c0, int ncols) {
    for (int j = 0; j < ncols; ++j) {
        double a1 = A(r1, c0 + j), a2 = A(r1 + 1, c0 + j);
        A(r1, c0 + j) = c * a1 + s * a2;
        A(r1 + 1, c0 + j) = -s * a1 + c * a2;
    }
}
// stdlib rmul!(A, G') on columns (c1, c1+1): G' = Givens(c, -s): a1 <- a1 c + a2 s ; a2 <- -a1 s + a2 c
inline void rmul_Gadj(const MatD& A, int r0, int nrows, int c1, double c, double s) {
    for (int r = 0; r < nrows; ++r) {
        double a1 = A(r0 + r, c1), a2 = A(r0 + r, c1 + 1);
        A(r0 + r, c1) = a1 * c + a2 * s;
        A(r0 + r, c1 + 1) = -a1 * s + a2 * c;
    }
}

// rschur2x2.jl:9-96  _gs2x2! (LAPACK dlanv2): standardise a real 2x2; returns rotation (cs,sn)
// and the eigenvalue pair (w1, w2).
inline void gs2x2(double& a, double& b, double& c, double& d, double& cs, double& sn,
                  std::complex<double>& w1, std::complex<double>& w2) {
    auto sgn = [](double x) { return (x < 0) ? -1.0 : 1.0; };
    const double half = 0.5;
    const double small = 4 * std::numeric_limits<double>::epsilon();
    if (c == 0) {
        cs = 1.0; sn = 0.0;
    } else if (b == 0) {
        cs = 0.0; sn = 1.0;
        double a0 = a, c0 = c, d0 = d;
        a = d0; b = -c0; c = 0.0; d = a0;
    } else if (((a - d) == 0) && (b * c < 0)) {
        cs = 1.0; sn = 0.0;
    } else {
        double asubd = a - d;
        double p = half * asubd;
        double bcmax = std::max(std::fabs(b), std::fabs(c));
        double bcmis = std::min(std::fabs(b), std::fabs(c)) * sgn(b) * sgn(c);
        double scale = std::max(std::fabs(p), bcmax);
        double z = (p / scale) * p + (bcmax / scale) * bcmis;
        if (z >= small) {
            z = p + std::sqrt(scale) * std::sqrt(z) * sgn(p);
            a = d + z;
            d -= (bcmax / z) * bcmis;
            double tau = std::hypot(c, z);
            cs = z / tau;
            sn = c / tau;
            b -= c;
            c = 0.0;
        } else {
            double sigma = b + c;
            double tau = std::hypot(sigma, asubd);
            cs = std::sqrt(half * (1.0 + std::fabs(sigma) / tau));
            sn = -(p / (tau * cs)) * sgn(sigma);
            double aa = a * cs + b * sn;
            double bb = -a * sn + b * cs;
            double cc = c * cs + d * sn;
            double dd = -c * sn + d * cs;
            a = aa * cs + cc * sn;
            b = bb * cs + dd * sn;
            c = -aa * sn + cc * cs;
            d = -bb * sn + dd * cs;
            double midad = half * (a + d);
            a = midad;
            d = a;
            if (c != 0) {
                if (b != 0) {
                    if (b * c >= 0) {
                        double sab = std::sqrt(std::fabs(b));
                        double sac = std::sqrt(std::fabs(c));
                        p = sab * sac * sgn(c);
                        tau = 1.0 / std::sqrt(std::fabs(b + c));
                        a = midad + p;
                        d = midad - p;
                        b -= c;
                        c = 0;
                        double cs1 = sab * tau;
                        double sn1 = sac * tau;
                        double cs2 = cs * cs1 - sn * sn1;
                        double sn2 = cs * sn1 + sn * cs1;
                        cs = cs2; sn = sn2;
                    }
                } else {
                    b = -c; c = 0.0;
                    double cs2 = -sn, sn2 = cs;
                    cs = cs2; sn = sn2;
                }
            }
        }
    }
    if (c == 0) {
        w1 = {a, 0.0};
        w2 = {d, 0.0};
    } else {
        double rti = std::sqrt(std::fabs(b)) * std::sqrt(std::fabs(c));
        w1 = {a, rti};
        w2 = {d, -rti};
    }
}

// opnorm(view(M, r0:r1, c0:c1), 1): max column sum
inline double opnorm1(const MatD& M, int r0, int r1, int c0, int c1) {
    double best = 0.0;
    for (int c = c0; c <= c1; ++c) {
        double s = 0.0;
        for (int r = r0; r <= r1; ++r) s += std::fabs(M(r, c));
        if (s > best || std::isnan(s)) best = s;
    }
    return best;
}

// ---------------------------------------------------------------------------------------------
// PSD.jl:213-259  phessenberg!: A[j] (j = 1..p, 1-based) overwritten LAPACK-style (H above, reflectors
// below), tau[j] of length n (tau[1][n] unused, tau[j>=2][n] = 0).
inline void phessenberg(int n, int p, std::vector<MatD>& A, std::vector<std::vector<double>>& tau, int ncols = -1) {
    tau.assign(p + 1, std::vector<double>(n + 1, 0.0));
    const int ilast = (ncols >= 0 && ncols < n - 1) ? ncols : (n - 1);  // (ncols: bounded sample of the first columns)
    for (int i = 1; i <= ilast; ++i) {
        const int i1 = i + 1;
        for (int j = p; j >= 2; --j) {
            double* xi = &A[j](i, i);
            double t = xreflector(xi, n - i + 1, 1);
            tau[j][i] = t;
            const double* v = xi + 1;
            lmul_Hadj(v, 1, t, A[j], i, i1, n - i + 1, n - i);
            rmul_H(A[j - 1], 1, i, n, n - i + 1, v, 1, t);
        }
        double* xi = &A[1](i1, i);
        double t = xreflector(xi, n - i, 1);
        tau[1][i] = t;
        const double* v = xi + 1;
        lmul_Hadj(v, 1, t, A[1], i1, i1, n - i, n - i);
        rmul_H(A[p], 1, i1, n, n - i, v, 1, t);
    }
}

// PSD.jl:136-143,180-197: explicit Q factors.  Q_1 = prod_i H_{1,i} (reflector i acts on rows
// i+1..n), Q_j = prod_i H_{j,i} (rows i..n) — what Matrix(H.Q) / Matrix(QR.Q) return.
inline void materializeQ(int n, int j, const MatD& Aj, const std::vector<double>& tauj, const MatD& Q) {
    for (int c = 1; c <= n; ++c)
        for (int r = 1; r <= n; ++r) Q(r, c) = (r == c) ? 1.0 : 0.0;
    const int off = (j == 1) ? 1 : 0;  // Hessenberg reflectors start one row lower
    for (int i = n - 1; i >= 1; --i) {
        const int r0 = i + off;
        const int m = n - r0 + 1;
        if (m < 1) continue;
        const double* v = &Aj.a[(size_t)(i - 1) * Aj.ld + (r0 - 1)] + 1;  // below A(r0, i)
        lmul_H(v, 1, tauj[i], Q, r0, r0, m, n - r0 + 1);
    }
}

static double dbg_maxdust = 0; static int dbg_dustpos = 0;
struct SweepLog {
    // per inner iteration: kind (0 = QR sweep, 1 = RQ clean-up pass), l, i
    std::vector<int32_t> rec;
    void add(int kind, int l, int i) {
        rec.push_back(kind); rec.push_back(l); rec.push_back(i);
    }
};

// PSD.jl:322-1096  real periodic QR on Hessenberg x triangular...  H[1] Hessenberg, H[2..p] upper
// triangular, Z[1..p] (ignored when !wantZ; must be preset to Q or I by the caller).
// Returns 0 or the level i at which convergence failed (PSD.jl:892).
inline int pschur_hess(int n, int p, std::vector<MatD>& H, std::vector<MatD>& Z, bool wantT, bool wantZ,
                       int maxitfac, std::complex<double>* lam, int64_t* niter_out, SweepLog* log) {
    if (niter_out) *niter_out = 0;
    if (n == 1) {  // PSD.jl:333-352
        double l1 = H[1](1, 1);
        for (int j = 2; j <= p; ++j) l1 *= H[j](1, 1);
        lam[0] = {l1, 0.0};
        return 0;
    }
    const double dat1 = 0.75, dat2 = -0.4375;
    std::vector<double> wr(n + 1, 0.0), wi(n + 1, 0.0), hsup(n + 1, 0.0);
    double v[3] = {0, 0, 0};
    const double unfl = std::numeric_limits<double>::min();
    const double ulp = std::numeric_limits<double>::epsilon();
    // PSD.jl:366-375 with _AT_pwr16[] = 4: ulpx = ulp^(1+4/16)
    double ulpx = ulp;
    {
        const int AT = 4, AT_hi = AT / 16, AT_lo = AT % 16;
        for (int q = 0; q < AT_hi; ++q) ulpx *= ulp;
        double s = ulp;
        for (int iu : {8, 4, 2, 1}) {
            s = std::sqrt(s);
            if (AT_lo & iu) ulpx *= s;
        }
    }
    const double smlnum = unfl * (n / ulp);
    const double sn = ulp * n;
    MatD& H1 = H[1];
    if (n > 2)
        for (int r = 3; r <= n; ++r) H1(r, 1) = 0.0;
    std::vector<double> hnorms(p + 1, 0.0);
    for (int j = 2; j <= p; ++j) {
        for (int r = 2; r <= n; ++r) H[j](r, 1) = 0.0;
        hnorms[j] = sn * opnorm1(H[j], 1, n, 1, n);
    }
    int i1 = 1, i2 = n;
    double* hdiag = wr.data();
    double* hsub = wi.data();
    // _gethess! PSD.jl:262,406
    for (int c = 1; c <= n; ++c)
        for (int r = c + 2; r <= n; ++r) H1(r, c) = 0.0;
    MatD& Hp = H[p];

    const int maxit = maxitfac * n;
    int maxitleft = maxit;
    int i = n;
    int64_t niter = 0;
    double tst1 = 0.0;
    double hh10 = 0, hh11 = 0, hh12 = 0, hh21 = 0, hh22 = 0;
    double hp00, hp01, hp02, hp11 = 0, hp12 = 0, hp22;
    double h33 = 0, h44 = 0, h43h34 = 0, h43, h34;
    double rt1r = 0, rt2r = 0, rt1i = 0, rt2i = 0;

    while (i >= 1) {
        int l = 1;
        int its = 1;
        bool splitting = false;
        while (its < maxitleft) {
            splitting = false;
            // PSD.jl:475-495
            hp22 = 1.0;
            if (i > l) {
                hp12 = 0.0;
                hp11 = 1.0;
                for (int j = 2; j <= p; ++j) {
                    MatD& Hj = H[j];
                    hp22 *= Hj(i, i);
                    hp12 = hp11 * Hj(i - 1, i) + hp12 * Hj(i, i);
                    hp11 *= Hj(i - 1, i - 1);
                }
                hh21 = H1(i, i - 1) * hp11;
                hh22 = H1(i, i - 1) * hp12 + H1(i, i) * hp22;
                hdiag[i] = hh22;
                hsub[i] = hh21;
            } else {
                hp22 *= H1(i, i);
                for (int j = 2; j <= p; ++j) hp22 *= H[j](i, i);
                hdiag[i] = hp22;
            }
            // PSD.jl:501-576
            int klast = i;
            bool found = false;
            double xmin = std::numeric_limits<double>::infinity();
            for (int k = i; k >= l + 1; --k) {
                klast = k;
                hp00 = 1.0;
                hp01 = 0.0;
                if (k > l + 1) {
                    hp02 = 0.0;
                    for (int j = 2; j <= p; ++j) {
                        MatD& Hj = H[j];
                        hp02 = hp00 * Hj(k - 2, k) + hp01 * Hj(k - 1, k) + hp02 * Hj(k, k);
                        hp01 = hp00 * Hj(k - 2, k - 1) + hp01 * Hj(k - 1, k - 1);
                        hp00 *= Hj(k - 2, k - 2);
                    }
                    hh10 = H1(k - 1, k - 2) * hp00;
                    hh11 = H1(k - 1, k - 2) * hp01 + H1(k - 1, k - 1) * hp11;
                    hh12 = H1(k - 1, k - 2) * hp02 + H1(k - 1, k - 1) * hp12 + H1(k - 1, k) * hp22;
                    hsub[k - 1] = hh10;
                } else {
                    hh10 = 0.0;
                    hh11 = H1(k - 1, k - 1) * hp11;
                    hh12 = H1(k - 1, k - 1) * hp12 + H1(k - 1, k) * hp22;
                }
                hdiag[k - 1] = hh11;
                hsup[n - i + k - 1] = hh12;
                if (std::fabs(hh21) < std::fabs(xmin)) xmin = hh21;
                tst1 = std::fabs(hh11) + std::fabs(hh22);
                if (tst1 == 0) tst1 = opnorm1(H1, l, i, l, i);
                // PSD.jl:546-565 (LAPACK + Ahues-Tisseur with shrunk threshold)
                if (std::fabs(hh21) <= smlnum) {
                    found = true;
                } else if (std::fabs(hh21) <= ulp * tst1) {
                    double ab = std::max(std::fabs(hh21), std::fabs(hh12));
                    double ba = std::min(std::fabs(hh21), std::fabs(hh12));
                    double aa = std::max(std::fabs(hh22), std::fabs(hh11 - hh22));
                    double bb = std::min(std::fabs(hh22), std::fabs(hh11 - hh22));
                    double stmp = aa + ab;
                    found = ba * (ab / stmp) <= std::max(smlnum, ulpx * (bb * (aa / stmp)));
                }
                if (found) break;
                hp22 = hp11;
                hp11 = hp00;
                hp12 = hp01;
                hh22 = hh11;
                hh21 = hh10;
            }
            // PSD.jl:585
            l = (i > l) ? (found ? klast : l) : i;

            if (l > 1) {
                if (wantT) {  // PSD.jl:590-665
                    tst1 = std::fabs(H1(l - 1, l - 1)) + std::fabs(H1(l, l));
                    if (tst1 == 0) tst1 = opnorm1(H1, l, i, l, i);
                    if (std::fabs(H1(l, l - 1)) > std::max(ulp * tst1, smlnum)) {
                        if (log) log->add(1, l, i);
                        for (int k = i; k >= l; --k) {
                            for (int j = 1; j <= p - 1; ++j) {
                                MatD& Hj = H[j];
                                double xi[2] = {Hj(k, k), Hj(k, k - 1)};
                                double t = xreflector(xi, 2, 1);
                                Hj(k, k - 1) = 0.0;
                                Hj(k, k) = xi[0];
                                HH2 hr{xi[1], 1.0, t};
                                rmul_HH2(Hj, i1, k - 1, (k - 1) - i1 + 1, hr);
                                lmul_HH2adj(hr, H[j + 1], k - 1, k - 1, i2 - (k - 1) + 1);
                                if (wantZ) rmul_HH2(Z[j + 1], 1, k - 1, n, hr);
                            }
                            if (k < i) {
                                double xi[2] = {Hp(k + 1, k + 1), Hp(k + 1, k)};
                                double t = xreflector(xi, 2, 1);
                                Hp(k + 1, k) = 0.0;
                                Hp(k + 1, k + 1) = xi[0];
                                HH2 hr{xi[1], 1.0, t};
                                rmul_HH2(Hp, i1, k, k - i1 + 1, hr);
                                lmul_HH2adj(hr, H1, k, k, i2 - k + 1);
                                if (wantZ) rmul_HH2(Z[1], 1, k, n, hr);
                            }
                        }
                        // _extra_rq[] == false: PSD.jl:653-659
                        Hp(l, l - 1) = 0.0;
                    }
                    H1(l, l - 1) = 0.0;
                }
            }
            if (l >= i - 1) {  // PSD.jl:668
                splitting = true;
                break;
            }
            if (!wantT) {
                i1 = l;
                i2 = i;
            }
            if (g_sweep_cap >= 0 && g_sweeps_done >= g_sweep_cap) return -77;  // bounded sample (bench.py)
            ++g_sweeps_done;
            if (log) log->add(0, l, i);
            bool exc_shift = false;
            if (its == 10) {  // PSD.jl:680-689
                exc_shift = true;
                double s = std::fabs(hsub[l + 1]) + std::fabs(hsub[l + 2]);
                h44 = dat1 * s + hdiag[l];
                h33 = h44;
                h43h34 = dat2 * s * s;
            } else if (its % 10 == 0) {  // PSD.jl:690-699
                exc_shift = true;
                double s = std::fabs(hsub[i]) + std::fabs(hsub[i - 1]);
                h44 = dat1 * s + hdiag[i];
                h33 = h44;
                h43h34 = dat2 * s * s;
            } else {  // PSD.jl:700-763 with _slicot_shifts[] == false (dlahqr shifts)
                h44 = hdiag[i];
                h33 = hdiag[i - 1];
                h43h34 = hsub[i] * hsup[n - 1];
                h43 = hsub[i];
                h34 = hsup[n - 1];
                double s = std::fabs(h33) + std::fabs(h34) + std::fabs(h43) + std::fabs(h44);
                if (s == 0) {
                    rt1r = rt2r = rt1i = rt2i = 0.0;
                } else {
                    h33 /= s; h44 /= s; h34 /= s; h43 /= s;
                    double trc = (h33 + h44) * 0.5;
                    double disc = (h33 - trc) * (h44 - trc) - h34 * h43;
                    double rtdisc = std::sqrt(std::fabs(disc));
                    if (disc >= 0) {
                        rt1r = trc * s;
                        rt2r = rt1r;
                        rt1i = rtdisc * s;
                        rt2i = -rt1i;
                    } else {
                        rt1r = trc + rtdisc;
                        rt2r = trc - rtdisc;
                        rt1r = (std::fabs(rt1r - h44) <= std::fabs(rt2r - h44)) ? (rt1r * s) : (rt2r * s);
                        rt2r = rt1r;
                        rt1i = rt2i = 0.0;
                    }
                }
            }
            // PSD.jl:766-803 with _allow_early_QR[] == false: mmax = l, single pass m = l
            const int mlast = l;
            {
                const int m = l;
                double h11 = hdiag[m];
                double h12 = hsup[n - i + m];
                double h21 = hsub[m + 1];
                double h22 = hdiag[m + 1];
                double v1, v2, v3;
                if (exc_shift) {
                    double h44s = h44 - h11;
                    double h33s = h33 - h11;
                    v1 = (h33s * h44s - h43h34) / h21 + h12;
                    v2 = h22 - h11 - h33s - h44s;
                    v3 = hsub[m + 2];
                } else {
                    double s = std::fabs(h11 - rt2r) + std::fabs(rt2i) + std::fabs(h21);
                    double h21s = h21 / s;
                    v1 = h21s * h12 + (h11 - rt1r) * ((h11 - rt2r) / s) - rt1i * (rt2i / s);
                    v2 = h21s * (h11 + h22 - rt1r - rt2r);
                    v3 = h21s * hsub[m + 2];
                }
                double s = std::fabs(v1) + std::fabs(v2) + std::fabs(v3);
                v[0] = v1 / s; v[1] = v2 / s; v[2] = v3 / s;
            }
            // PSD.jl:806-886 double-shift periodic QR sweep
            for (int k = mlast; k <= i - 1; ++k) {
                const int nr = std::min(3, i - k + 1);
                const int nrow = std::min(k + nr, i) - i1 + 1;
                if (k > mlast)
                    for (int q = 0; q < nr; ++q) v[q] = H1(k + q, k - 1);
                double tau1 = xreflector(v, nr, 1);
                if (k > mlast) {
                    H1(k, k - 1) = v[0];
                    H1(k + 1, k - 1) = 0.0;
                    if (k < i - 1) H1(k + 2, k - 1) = 0.0;
                }
                lmul_Hadj(v + 1, 1, tau1, H1, k, k, nr, i2 - k + 1);
                rmul_H(Hp, i1, k, nrow, nr, v + 1, 1, tau1);
                if (wantZ) rmul_H(Z[1], 1, k, n, nr, v + 1, 1, tau1);
                for (int j = p; j >= 2; --j) {
                    MatD& Hj = H[j];
                    for (int q = 0; q < nr; ++q) v[q] = Hj(k + q, k);
                    double t = xreflector(v, nr, 1);
                    Hj(k, k) = v[0];
                    Hj(k + 1, k) = 0.0;
                    if (nr == 3) Hj(k + 2, k) = 0.0;
                    lmul_Hadj(v + 1, 1, t, Hj, k, k + 1, nr, i2 - (k + 1) + 1);
                    rmul_H(H[j - 1], i1, k, nrow, nr, v + 1, 1, t);
                    if (wantZ) rmul_H(Z[j], 1, k, n, nr, v + 1, 1, t);
                    if (nr == 3) {
                        v[0] = Hj(k + 1, k + 1);
                        v[1] = Hj(k + 2, k + 1);
                        t = xreflector(v, 2, 1);
                        Hj(k + 1, k + 1) = v[0];
                        Hj(k + 2, k + 1) = 0.0;
                        lmul_Hadj(v + 1, 1, t, Hj, k + 1, k + 2, 2, i2 - (k + 2) + 1);
                        rmul_H(H[j - 1], i1, k + 1, nrow, 2, v + 1, 1, t);
                        if (wantZ) rmul_H(Z[j], 1, k + 1, n, 2, v + 1, 1, t);
                    }
                }
            }
            its += 1;
        }
        if (!splitting) {
            if (niter_out) *niter_out = niter + its;
            return i;  // PSD.jl:892 "convergence failed at level i"
        }
        // PSD.jl:896-1054 deflation
        if (l == i) {
            lam[i - 1] = {hdiag[i], 0.0};
        } else if (l == i - 1) {
            if (wantT) {
                hp22 = 1.0; hp12 = 0.0; hp11 = 1.0;
                for (int j = 2; j <= p; ++j) {
                    MatD& Hj = H[j];
                    hp22 *= Hj(i, i);
                    hp12 = hp11 * Hj(i - 1, i) + hp12 * Hj(i, i);
                    hp11 *= Hj(i - 1, i - 1);
                }
                hh21 = H1(i, i - 1) * hp11;
                hh22 = H1(i, i - 1) * hp12 + H1(i, i) * hp22;
                hh11 = H1(i - 1, i - 1) * hp11;
                hh12 = H1(i - 1, i - 1) * hp12 + H1(i - 1, i) * hp22;
            } else {
                hh11 = hdiag[i - 1];
                hh12 = hsup[n - 1];
                hh21 = hsub[i];
                hh22 = hdiag[i];
            }
            double a = hh11, b = hh12, c = hh21, d = hh22, gcs, gsn;
            std::complex<double> w1, w2;
            gs2x2(a, b, c, d, gcs, gsn, w1, w2);
            lam[i - 2] = w1;
            lam[i - 1] = w2;
            hdiag[i - 1] = w1.real(); hdiag[i] = w2.real();
            hsub[i - 1] = w1.imag(); hsub[i] = w2.imag();
            if (wantT) {
                int jmin = 0, jmax = 0;
                for (int j = 2; j <= p; ++j) {
                    MatD& Hj = H[j];
                    if (jmin == 0 && std::fabs(Hj(i - 1, i - 1)) <= hnorms[j]) jmin = j;
                    if (std::fabs(Hj(i, i)) <= hnorms[j]) jmax = j;
                }
                if (jmin != 0 && jmax != 0) {
                    if (jmin - 1 <= p - jmax + 1) jmax = 0;
                    else jmin = 0;
                }
                if (jmin != 0) {  // PSD.jl:959-977
                    for (int j = 1; j <= jmin - 1; ++j) {
                        MatD& Hj = H[j];
                        double xi[2] = {Hj(i, i), Hj(i, i - 1)};
                        double t = xreflector(xi, 2, 1);
                        HH2 hr{xi[1], 1.0, t};
                        Hj(i, i - 1) = 0.0;
                        Hj(i, i) = xi[0];  // reference writes xi[2] (defect, see header)
                        rmul_HH2(Hj, i1, i - 1, (i - 1) - i1 + 1, hr);
                        lmul_HH2adj(hr, H[j + 1], i - 1, i - 1, i2 - (i - 1) + 1);
                        if (wantZ) rmul_HH2(Z[j + 1], 1, i - 1, n, hr);
                    }
                } else {  // PSD.jl:978-1052
                    bool replaceG = (jmax > 0) && (hsub[i - 1] == 0);
                    const double a1 = std::abs(lam[i - 2]), a2 = std::abs(lam[i - 1]);
                    if (lam[i - 1] * lam[i - 2] == std::complex<double>(0.0, 0.0)) {
                        replaceG = true;
                    } else if (hsub[i - 1] == 0) {
                        if (std::min(a1, a2) / std::max(a1, a2) < ulp) replaceG = true;
                    }
                    for (int its2 = 1; its2 <= 20; ++its2) {
                        if (replaceG) {
                            double rr;
                            givens_algorithm(H1(i - 1, i - 1), H1(i, i - 1), gcs, gsn, rr);
                        }
                        lmul_G(gcs, gsn, H1, i - 1, i - 1, i2 - (i - 1) + 1);
                        rmul_Gadj(Hp, i1, i - i1 + 1, i - 1, gcs, gsn);
                        if (wantZ) rmul_Gadj(Z[1], 1, n, i - 1, gcs, gsn);
                        for (int j = p; j >= std::max(2, jmax + 1); --j) {
                            MatD& Hj = H[j];
                            v[0] = Hj(i - 1, i - 1);
                            v[1] = Hj(i, i - 1);
                            double t = xreflector(v, 2, 1);
                            Hj(i - 1, i - 1) = v[0];
                            Hj(i, i - 1) = 0.0;
                            lmul_Hadj(v + 1, 1, t, Hj, i - 1, i, 2, i2 - i + 1);
                            rmul_H(H[j - 1], i1, i - 1, i - i1 + 1, 2, v + 1, 1, t);
                            if (wantZ) rmul_H(Z[j], 1, i - 1, n, 2, v + 1, 1, t);
                        }
                        if (!replaceG || (std::fabs(H1(i, i - 1)) < std::max(smlnum, ulp * std::max(a1, a2))))
                            break;
                        replaceG = true;
                    }
                    if (jmax > 0) {
                        H1(i, i - 1) = 0.0;
                        if (jmax > 1) H[jmax](i, i - 1) = 0.0;
                    } else if (hh21 == 0) {
                        H1(i, i - 1) = 0.0;
                    }
                    if (replaceG) {
                        double l1 = H1(i - 1, i - 1), l2 = H1(i, i);
                        for (int j = 2; j <= p; ++j) {
                            l1 *= H[j](i - 1, i - 1);
                            l2 *= H[j](i, i);
                        }
                        // reference indexes lambda[1], lambda[2] here (defect, see header)
                        if (std::abs(l1 - lam[i - 2]) > std::abs(l1 - lam[i - 1])) std::swap(lam[i - 2], lam[i - 1]);
                    }
                }
            }
        }
        maxitleft -= its;
        i = l - 1;
        niter += its;
    }
    // PSD.jl:1066-1073 clear dust
    for (int q = 1; q <= n - 1; ++q)
        if (lam[q - 1].imag() == 0.0) {
            if (std::fabs(H1(q + 1, q)) > dbg_maxdust) { dbg_maxdust = std::fabs(H1(q + 1, q)); dbg_dustpos = q; }
            H1(q + 1, q) = 0.0;
        }
    if (niter_out) *niter_out = niter;
    return 0;
}

}  // namespace psdo
