// ORACLE — TEST INFRASTRUCTURE ONLY (rules in psd_oracle_real.hpp).
//
// CPU restatement of the reference's REAL reordering path with 1x1 and 2x2 blocks:
//   ordschur!(P, select) real, _moveblock!, _swapschur!   rordschur.jl:3-132,141-251,253-260
//   _swapadjqr! (standard), _filled2hess!                 sylswap.jl:14-157,159-191
//   _psylsolve / Kronecker form                           sylvester.jl:11-33,170-193 (the reference solves the
//       same cyclic block-bidiagonal system by the structured QR of babd.jl:17-96; here the dense Kronecker
//       form of sylvester.jl:11-33 is solved by a dense Householder QR — same unique solution)
//   _phess2x2!                                            rpschur2x2.jl:326-359
//   _updateλ! (real), _rpeigvals2x2, _sanitize_reigpair!  ordschur.jl:122-204, rpschur2x2.jl:9-275
#pragma once
#include "psd_oracle_ord.hpp"

namespace psdo {

// small dense column-major matrix helper
struct SM {
    int r, c;
    std::vector<double> a;
    SM() : r(0), c(0) {}
    SM(int r_, int c_) : r(r_), c(c_), a((size_t)r_ * c_, 0.0) {}
    double& operator()(int i, int j) { return a[(size_t)j * r + i]; }
    double operator()(int i, int j) const { return a[(size_t)j * r + i]; }
    static SM eye(int n) {
        SM m(n, n);
        for (int i = 0; i < n; ++i) m(i, i) = 1.0;
        return m;
    }
};
inline SM smul(const SM& A, const SM& B) {
    SM C(A.r, B.c);
    for (int j = 0; j < B.c; ++j)
        for (int k = 0; k < A.c; ++k)
            for (int i = 0; i < A.r; ++i) C(i, j) += A(i, k) * B(k, j);
    return C;
}
inline SM strans(const SM& A) {
    SM T(A.c, A.r);
    for (int i = 0; i < A.r; ++i)
        for (int j = 0; j < A.c; ++j) T(j, i) = A(i, j);
    return T;
}
inline double sfnorm(const SM& A) {
    double s = 0.0;
    for (double v : A.a) s = std::hypot(s, v);
    return s;
}

// full Q (m x m) of the Householder QR of an m x ncol matrix (what `qr(Xi)` + rmul!/lmul! with q use)
inline SM full_q(SM X) {
    const int m = X.r, nc = X.c;
    SM Q = SM::eye(m);
    for (int k = 0; k < std::min(m - 1, nc); ++k) {
        std::vector<double> x(m - k);
        for (int i = k; i < m; ++i) x[i - k] = X(i, k);
        double tau = xreflector(x.data(), m - k, 1);  // x <- (beta, v2..)
        if (tau == 0.0) continue;
        std::vector<double> v(m - k, 1.0);
        for (int i = 1; i < m - k; ++i) v[i] = x[i];
        // X <- H X (columns k..), Q <- Q H
        for (int j = k; j < nc; ++j) {
            double d = 0.0;
            for (int i = k; i < m; ++i) d += v[i - k] * X(i, j);
            d *= tau;
            for (int i = k; i < m; ++i) X(i, j) -= d * v[i - k];
        }
        for (int i = 0; i < m; ++i) {
            double d = 0.0;
            for (int j = k; j < m; ++j) d += Q(i, j) * v[j - k];
            d *= tau;
            for (int j = k; j < m; ++j) Q(i, j) -= d * v[j - k];
        }
    }
    return Q;
}

// sylvester.jl:170-193 (dense form :11-33): A_k X_k - X_{k+1} B_k = -C_k, k = 1..K (X_{K+1} = X_1);
// A_k p1 x p1, B_k p2 x p2, C_k p1 x p2.  Returns false on a singular system.
inline bool psylsolve(int K, int p1, int p2, const std::vector<SM>& A, const std::vector<SM>& B, const std::vector<SM>& C,
                      std::vector<SM>& X) {
    const int pp = p1 * p2, N = K * pp;
    std::vector<double> M((size_t)N * N, 0.0), y(N, 0.0);
    auto at = [&](int r, int c) -> double& { return M[(size_t)c * N + r]; };
    // unknown ordering: x = [vec X_1; ...; vec X_K] (column-major vec); block row 1 <- equation K, row k+1 <- equation k
    auto put = [&](int rowblk, int eq) {
        // equation eq (1-based): kron(I_p2, A_eq) vec(X_eq) + kron(B_eq^T, -I_p1) vec(X_{eq+1}) = -vec(C_eq)
        const int cx = (eq - 1) * pp, cy = (eq % K) * pp, r0 = rowblk * pp;
        for (int j = 0; j < p2; ++j)
            for (int i = 0; i < p1; ++i) {
                const int r = r0 + j * p1 + i;
                for (int k = 0; k < p1; ++k) at(r, cx + j * p1 + k) += A[eq - 1](i, k);
                for (int k = 0; k < p2; ++k) at(r, cy + k * p1 + i) += -B[eq - 1](k, j);
                y[r] = -C[eq - 1](i, j);
            }
    };
    put(0, K);
    for (int k = 1; k <= K - 1; ++k) put(k, k);
    if (!qr_solve<double>(N, M, y)) return false;
    X.assign(K, SM(p1, p2));
    for (int k = 0; k < K; ++k)
        for (int j = 0; j < p2; ++j)
            for (int i = 0; i < p1; ++i) X[k](i, j) = y[k * pp + j * p1 + i];
    return true;
}

// rpschur2x2.jl:326-359 _phess2x2!(As, 1) for 2x2 copies (S = nothing): As[0] untouched structure-wise,
// As[1..] made upper triangular; Qs[l] are the 2x2 transformations (Qs[lp] = hr').
inline void phess2x2(int k, std::vector<SM>& As, std::vector<SM>& Qs) {
    Qs.assign(k, SM::eye(2));
    for (int l = 2; l <= k; ++l) {
        SM& Al = As[l - 1];
        const int lp = (l % k) + 1;
        SM& Ap = As[lp - 1];
        double xi[2] = {Al(0, 0), Al(1, 0)};
        const double tau = xreflector(xi, 2, 1);
        Al(0, 0) = xi[0];
        Al(1, 0) = 0.0;
        const HH2 hr{1.0, xi[1], tau};
        {  // lmul!(hr', view(Al, 1:2, 2:2))
            const double t1 = hr.tau * hr.v1, t2 = hr.tau * hr.v2;
            const double s = hr.v1 * Al(0, 1) + hr.v2 * Al(1, 1);
            Al(0, 1) -= s * t1;
            Al(1, 1) -= s * t2;
        }
        {  // lmul!(hr', Qs[lp])
            SM& Q = Qs[lp - 1];
            const double t1 = hr.tau * hr.v1, t2 = hr.tau * hr.v2;
            for (int j = 0; j < 2; ++j) {
                const double s = hr.v1 * Q(0, j) + hr.v2 * Q(1, j);
                Q(0, j) -= s * t1;
                Q(1, j) -= s * t2;
            }
        }
        {  // rmul!(view(Ap, 1:2, 1:2), hr)
            const double t1 = hr.v1 * hr.tau, t2 = hr.v2 * hr.tau;
            for (int r = 0; r < 2; ++r) {
                const double s = Ap(r, 0) * hr.v1 + Ap(r, 1) * hr.v2;
                Ap(r, 0) -= s * t1;
                Ap(r, 1) -= s * t2;
            }
        }
    }
}

// sylswap.jl:159-191 _filled2hess! (S = nothing); j0 0-based
inline void filled2hess(int k, std::vector<SM>& Txx, std::vector<SM>& Ws, int j0) {
    const int m = Txx[0].r, j1 = j0 + 1;
    std::vector<SM> Th(k, SM(2, 2)), H1qs;
    for (int l = 0; l < k; ++l)
        for (int a = 0; a < 2; ++a)
            for (int b = 0; b < 2; ++b) Th[l](a, b) = Txx[l](j0 + a, j0 + b);
    phess2x2(k, Th, H1qs);
    for (int l = 1; l <= k; ++l) {
        const int lp = (l % k) + 1;
        const SM& q = H1qs[l - 1];
        const SM& qp = H1qs[lp - 1];
        SM& Tl = Txx[l - 1];
        for (int r = 0; r < m; ++r) {  // Tl[:, j0:j1] <- Tl[:, j0:j1] q
            const double a = Tl(r, j0), b = Tl(r, j1);
            Tl(r, j0) = a * q(0, 0) + b * q(1, 0);
            Tl(r, j1) = a * q(0, 1) + b * q(1, 1);
        }
        for (int c = 0; c < m; ++c) {  // Tl[j0:j1, :] <- qp' Tl[j0:j1, :]
            const double a = Tl(j0, c), b = Tl(j1, c);
            Tl(j0, c) = qp(0, 0) * a + qp(1, 0) * b;
            Tl(j1, c) = qp(0, 1) * a + qp(1, 1) * b;
        }
        SM& W = Ws[l - 1];
        for (int r = 0; r < m; ++r) {
            const double a = W(r, j0), b = W(r, j1);
            W(r, j0) = a * q(0, 0) + b * q(1, 0);
            W(r, j1) = a * q(0, 1) + b * q(1, 1);
        }
    }
}

// sylswap.jl:14-157 _swapadjqr!: swap adjacent blocks of sizes (p1, p2) starting at i1 (1-based).
// Returns 1 ok, 0 failed strong test (transformation applied anyway, as in the reference), -1 singular.
inline int swapadjqr(int n, int k, std::vector<MatT<double>>& X, std::vector<MatT<double>>& Zs, bool haveZ, int i1,
                     int p1, int p2) {
    const double tol = 100.0;
    const int i2 = i1 + p1, i2new = i1 + p2, i3 = i2 + p2 - 1, m = p1 + p2;
    double tnrm = 0.0;
    std::vector<SM> T11(k, SM(p1, p1)), T12(k, SM(p1, p2)), T22(k, SM(p2, p2));
    for (int l = 1; l <= k; ++l) {
        for (int c = i1; c <= i3; ++c)
            for (int r = i1; r <= i3; ++r) tnrm = std::hypot(tnrm, X[l](r, c));
        for (int a = 0; a < p1; ++a)
            for (int b = 0; b < p1; ++b) T11[l - 1](a, b) = X[l](i1 + a, i1 + b);
        for (int a = 0; a < p1; ++a)
            for (int b = 0; b < p2; ++b) T12[l - 1](a, b) = X[l](i1 + a, i2 + b);
        for (int a = 0; a < p2; ++a)
            for (int b = 0; b < p2; ++b) T22[l - 1](a, b) = X[l](i2 + a, i2 + b);
    }
    std::vector<SM> Xs;
    // k == 1: sylvester(T11, -T22, T12) solves T11 X - X T22 = -T12, the K = 1 case of the same system
    if (!psylsolve(k, p1, p2, T11, T22, T12, Xs)) return -1;
    const double thresh = std::max(std::numeric_limits<double>::min(), tol * std::numeric_limits<double>::epsilon() * tnrm);
    std::vector<SM> Txx(k, SM(m, m));
    for (int l = 0; l < k; ++l) {
        for (int a = 0; a < p1; ++a) {
            for (int b = 0; b < p1; ++b) Txx[l](a, b) = T11[l](a, b);
            for (int b = 0; b < p2; ++b) Txx[l](a, p1 + b) = T12[l](a, b);
        }
        for (int a = 0; a < p2; ++a)
            for (int b = 0; b < p2; ++b) Txx[l](p1 + a, p1 + b) = T22[l](a, b);
    }
    std::vector<SM> Qs(k);
    for (int l = 1; l <= k; ++l) {
        SM Xi(m, p2);
        for (int a = 0; a < p1; ++a)
            for (int b = 0; b < p2; ++b) Xi(a, b) = Xs[l - 1](a, b);
        for (int b = 0; b < p2; ++b) Xi(p1 + b, b) = 1.0;
        Qs[l - 1] = full_q(Xi);
    }
    for (int l = 1; l <= k; ++l) {  // rmul!(Txx[l], q_l); lmul!(q_l', Txx[l-1])
        const SM& q = Qs[l - 1];
        Txx[l - 1] = smul(Txx[l - 1], q);
        const int lp = (l == 1) ? k : l - 1;
        Txx[lp - 1] = smul(strans(q), Txx[lp - 1]);
    }
    bool fillin1 = false, fillin2 = false;
    if (p2 > 1)
        for (int l = 0; l < k; ++l) fillin1 |= std::fabs(Txx[l](1, 0)) > thresh;
    if (p1 > 1)
        for (int l = 0; l < k; ++l) fillin2 |= std::fabs(Txx[l](p2 + 1, p2)) > thresh;
    const bool fillin = fillin1 || fillin2;
    std::vector<SM> Ws;
    if (fillin) {
        Ws.assign(k, SM::eye(m));
        if (fillin1) filled2hess(k, Txx, Ws, 0);
        if (fillin2) filled2hess(k, Txx, Ws, p2);
    }
    bool ok = true;
    for (int l = 1; l <= k; ++l) {  // strong stability test (:116-129)
        const int l1 = (l % k) + 1;
        SM Tt = Txx[l - 1];
        if (fillin) Tt = smul(smul(Ws[l1 - 1], Tt), strans(Ws[l - 1]));
        Tt = smul(smul(Qs[l1 - 1], Tt), strans(Qs[l - 1]));
        double d = 0.0;
        for (int a = 0; a < m; ++a)
            for (int b = 0; b < m; ++b) d = std::hypot(d, Tt(a, b) - X[l](i1 + a, i1 + b));
        if (d > thresh) ok = false;
    }
    for (int l = 1; l <= k; ++l) {  // (:131-148)
        SM q = Qs[l - 1];
        if (fillin) q = smul(q, Ws[l - 1]);  // (Tl q) W == Tl (q W); W' (q' Tp) == (q W)' Tp
        MatT<double>& Tl = X[l];
        MatT<double>& Tp = X[(l == 1) ? k : l - 1];
        std::vector<double> tmp(m);
        for (int r = 1; r <= n; ++r) {
            for (int b = 0; b < m; ++b) {
                double s = 0.0;
                for (int a = 0; a < m; ++a) s += Tl(r, i1 + a) * q(a, b);
                tmp[b] = s;
            }
            for (int b = 0; b < m; ++b) Tl(r, i1 + b) = tmp[b];
        }
        if (haveZ) {
            MatT<double>& Zl = Zs[l];
            for (int r = 1; r <= n; ++r) {
                for (int b = 0; b < m; ++b) {
                    double s = 0.0;
                    for (int a = 0; a < m; ++a) s += Zl(r, i1 + a) * q(a, b);
                    tmp[b] = s;
                }
                for (int b = 0; b < m; ++b) Zl(r, i1 + b) = tmp[b];
            }
        }
        for (int c = 1; c <= n; ++c) {
            for (int b = 0; b < m; ++b) {
                double s = 0.0;
                for (int a = 0; a < m; ++a) s += q(a, b) * Tp(i1 + a, c);
                tmp[b] = s;
            }
            for (int b = 0; b < m; ++b) Tp(i1 + b, c) = tmp[b];
        }
    }
    // sweep up the dust (:150-154)
    for (int r = i2new; r <= i3; ++r)
        for (int c = i1; c <= i2new - 1; ++c) X[1](r, c) = 0.0;
    for (int l = 2; l <= k; ++l)
        for (int c = i1; c <= i3; ++c)
            for (int r = c + 1; r <= i3; ++r) X[l](r, c) = 0.0;
    return ok ? 1 : 0;
}

// rordschur.jl:253-260 _swapschur!
inline int swapschur(int n, int k, std::vector<MatT<double>>& X, std::vector<MatT<double>>& Zs, bool haveZ, int i1,
                     int nb1, int nb2) {
    if (nb1 == 1 && nb2 == 1) return swapadj1x1g<double>(n, k, X, Zs, haveZ, i1);
    return swapadjqr(n, k, X, Zs, haveZ, i1, nb1, nb2);
}

// rordschur.jl:141-251 _moveblock!: returns ok (1/0/-1); jsrc/jdest updated as in the reference
inline int moveblock(int n, int k, std::vector<MatT<double>>& X, std::vector<MatT<double>>& Zs, bool haveZ, int& jsrc,
                     int& jdest, int64_t* nswaps) {
    MatT<double>& A1 = X[1];
    if (jsrc > 1 && A1(jsrc, jsrc - 1) != 0) jsrc -= 1;
    int nbsrc = 1;
    if (jsrc < n && A1(jsrc + 1, jsrc) != 0) nbsrc = 2;
    if (jdest > 1 && A1(jdest, jdest - 1) != 0) jdest -= 1;
    if (jsrc == jdest) return 1;
    if (!(jdest < jsrc)) return -2;
    int here = jsrc;
    bool splitsrc = false;
    auto sw = [&](int i1, int nb1, int nb2) {
        if (nswaps) *nswaps += 1;
        return swapschur(n, k, X, Zs, haveZ, i1, nb1, nb2);
    };
    while (here > jdest) {
        int nbnext = 1;
        if (here >= 3 && A1(here - 1, here - 2) != 0) nbnext = 2;
        if (!splitsrc) {
            int ok = sw(here - nbnext, nbnext, nbsrc);
            if (ok != 1) {
                jdest = here;
                return ok;
            }
            here -= nbnext;
            if (nbsrc == 2 && A1(here + 1, here) == 0) splitsrc = true;
        } else {
            int ok = sw(here - nbnext, nbnext, 1);
            if (ok != 1) {
                jdest = here;
                return ok;
            }
            if (nbnext == 1) {
                ok = sw(here, nbnext, 1);
                if (ok != 1) {
                    jdest = here;
                    return ok;
                }
                // NOTE: the reference does not decrement `here` in this branch (rordschur.jl:207-215) and would loop
                // forever on a split pair moving through 1x1 blocks; one position is consumed here.
                here -= 1;
            } else {
                if (A1(here, here - 1) == 0) nbnext = 1;
                if (nbnext == 2) {
                    ok = sw(here - 1, 2, 1);
                    if (ok != 1) {
                        jdest = here;
                        return ok;
                    }
                    here -= 2;
                } else {
                    ok = sw(here, 1, 1);
                    if (ok != 1) {
                        jdest = here;
                        return ok;
                    }
                    ok = sw(here - 1, 1, 1);
                    if (ok != 1) {
                        jdest = here;
                        return ok;
                    }
                    here -= 2;
                }
            }
        }
    }
    jdest = here;
    return 1;
}

// rpschur2x2.jl:238-275
inline bool sanitize_reigpair(cplx* alpha, double* scal) {
    bool good = true;
    const double ulp = std::numeric_limits<double>::epsilon();
    if (alpha[0].imag() != 0 || alpha[1].imag() != 0) {
        const double sl = scal[0] - scal[1];
        cplx zt1, zt2;
        double cst;
        if (sl >= 0) {
            zt1 = alpha[1] * std::exp2(-sl);
            zt2 = alpha[0] - std::conj(zt1);
            cst = alpha[0].imag();
        } else {
            zt1 = alpha[0] * std::exp2(sl);
            zt2 = alpha[1] - std::conj(zt1);
            cst = alpha[1].imag();
        }
        const double misr = std::hypot(cst, zt1.imag());
        const double misc = std::abs(zt2) / 2;
        const double cs = std::max(std::abs(alpha[0]), std::max((double)1.0, std::abs(alpha[1])));
        good = std::min(misr, misc) <= cs * std::sqrt(ulp);
        if (misr > misc) {
            const int j = (scal[0] >= scal[1]) ? 0 : 1;
            const cplx at = (alpha[j] + std::conj(zt1)) / (double)2.0;
            const double ai = std::fabs(at.imag());
            alpha[0] = cplx(at.real(), ai);
            alpha[1] = std::conj(alpha[0]);
        } else {
            for (int j = 0; j < 2; ++j) alpha[j] = cplx(alpha[j].real(), 0.0);
        }
    }
    return good;
}

// rpschur2x2.jl:9-235 _rpeigvals2x2 with Aord = 1:k, schurindex = 1, recip = false:
// eigenvalues (alpha / beta * 2^scal) of X_1 X_2^{s_2} ... X_k^{s_k} for real 2x2 blocks, X_1 full, others upper
// triangular.  S == nullptr means all true (the ordschur callers); S[l-1] is the signature of X_l.
inline void rpeigvals2x2(int k, const std::vector<SM>& Xin, cplx* alpha, double* scal, bool& converged, bool& good,
                         const char* S = nullptr, double* beta = nullptr) {
    auto sg = [&](int l) { return S == nullptr || S[l - 1]; };
    struct Z2 { cplx a, b, c, d; };  // [a b; c d]
    std::vector<Z2> Xs(k);
    for (int l = 0; l < k; ++l) Xs[l] = Z2{cplx(Xin[l](0, 0)), cplx(Xin[l](0, 1)), cplx(Xin[l](1, 0)), cplx(Xin[l](1, 1))};
    const double ulp = std::numeric_limits<double>::epsilon();
    converged = false;
    auto giv = [&](cplx f, cplx g, double& c, cplx& s, cplx& r) { givens_algorithm_z(f, g, c, s, r); };
    for (int iter = 1; iter <= 80; ++iter) {
        Z2& X1 = Xs[0];
        const double lhs = std::abs(X1.c);
        double rhs = std::max(std::abs(X1.a), std::abs(X1.d));
        if (rhs == 0) rhs = std::abs(X1.b);
        if (lhs <= ulp * rhs) {
            converged = true;
            break;
        }
        double c;
        cplx s, r;
        if (iter == 1) {
            giv(cplx(1.0, -2.0), cplx(2.0, 2.0), c, s, r);
        } else if (iter % 40 == 0) {
            giv(cplx((double)k, 1.0), cplx(1.0, -2.0), c, s, r);
        } else {
            c = 1.0;
            s = cplx(0.0);
            double ct;
            cplx st;
            giv(cplx(1.0), cplx(1.0), ct, st, r);
            for (int l = k; l >= 2; --l) {
                const Z2& Xl = Xs[l - 1];
                cplx Z[3][3] = {{Xl.a, 0.0, 0.0}, {0.0, Xl.a, Xl.b}, {0.0, Xl.c, Xl.d}};
                if (!sg(l)) {  // rpschur2x2.jl:116-130
                    for (int q = 0; q < 3; ++q) {  // lmul!(Givens(1,3,ct,st), Z)
                        cplx a1 = Z[0][q], a2 = Z[2][q];
                        Z[0][q] = ct * a1 + st * a2;
                        Z[2][q] = -std::conj(st) * a1 + ct * a2;
                    }
                    for (int q = 0; q < 3; ++q) {  // lmul!(Givens(1,2,c,s), Z)
                        cplx a1 = Z[0][q], a2 = Z[1][q];
                        Z[0][q] = c * a1 + s * a2;
                        Z[1][q] = -std::conj(s) * a1 + c * a2;
                    }
                    giv(Z[2][2], Z[2][0], ct, st, r);
                    Z[2][2] = r;
                    st = -st;
                    for (int q = 0; q < 2; ++q) {  // rmul!(view(Z, 1:2, :), Givens(1,3,ct,st)')
                        cplx a1 = Z[q][0], a2 = Z[q][2];
                        Z[q][0] = a1 * ct + a2 * std::conj(st);
                        Z[q][2] = -a1 * st + a2 * ct;
                    }
                    giv(Z[1][1], Z[1][0], c, s, r);
                    s = -s;
                    continue;
                }
                // S true: rmul!(Z, G1') with G1 = Givens(1,3,ct,st); rmul!(Z, G2') with G2 = Givens(1,2,c,s)
                for (int q = 0; q < 3; ++q) {
                    cplx a1 = Z[q][0], a2 = Z[q][2];
                    Z[q][0] = a1 * ct + a2 * std::conj(st);
                    Z[q][2] = -a1 * st + a2 * ct;
                }
                for (int q = 0; q < 3; ++q) {
                    cplx a1 = Z[q][0], a2 = Z[q][1];
                    Z[q][0] = a1 * c + a2 * std::conj(s);
                    Z[q][1] = -a1 * s + a2 * c;
                }
                giv(Z[0][0], Z[2][0], ct, st, r);
                giv(Xl.a, Z[1][0], c, s, r);
            }
            const Z2& Xl = Xs[0];
            cplx Z[2][3] = {{Xl.a, -Xl.c, -Xl.d}, {Xl.c, 0.0, 0.0}};
            for (int q = 0; q < 2; ++q) {
                cplx a1 = Z[q][0], a2 = Z[q][2];
                Z[q][0] = a1 * ct + a2 * std::conj(st);
                Z[q][2] = -a1 * st + a2 * ct;
            }
            for (int q = 0; q < 2; ++q) {
                cplx a1 = Z[q][0], a2 = Z[q][1];
                Z[q][0] = a1 * c + a2 * std::conj(s);
                Z[q][1] = -a1 * s + a2 * c;
            }
            giv(Z[0][0], Z[1][0], c, s, r);
        }
        const double ct0 = c;
        const cplx st0 = s;
        for (int l = k; l >= 2; --l) {
            Z2 Y = Xs[l - 1];
            if (!sg(l)) {  // rpschur2x2.jl:162-176
                cplx a1 = Y.a, a2 = Y.c;
                Y.a = c * a1 + s * a2;
                Y.c = -std::conj(s) * a1 + c * a2;
                a1 = Y.b; a2 = Y.d;
                Y.b = c * a1 + s * a2;
                Y.d = -std::conj(s) * a1 + c * a2;
                giv(Y.d, Y.c, c, s, r);
                Y.d = r;
                Y.c = cplx(0.0);
                s = -s;
                a1 = Y.a; a2 = Y.b;
                Y.a = a1 * c + a2 * std::conj(s);
                Y.b = -a1 * s + a2 * c;
                Xs[l - 1] = Y;
                continue;
            }
            {  // rmul!(Y, G')
                cplx a1 = Y.a, a2 = Y.b;
                Y.a = a1 * c + a2 * std::conj(s);
                Y.b = -a1 * s + a2 * c;
                a1 = Y.c; a2 = Y.d;
                Y.c = a1 * c + a2 * std::conj(s);
                Y.d = -a1 * s + a2 * c;
            }
            giv(Y.a, Y.c, c, s, r);
            Y.a = r;
            Y.c = cplx(0.0);
            {  // lmul!(G, view(Y, :, 2:2))
                cplx a1 = Y.b, a2 = Y.d;
                Y.b = c * a1 + s * a2;
                Y.d = -std::conj(s) * a1 + c * a2;
            }
            Xs[l - 1] = Y;
        }
        Z2 Y = Xs[0];
        {  // lmul!(Givens(ct, st), Y)
            cplx a1 = Y.a, a2 = Y.c;
            Y.a = ct0 * a1 + st0 * a2;
            Y.c = -std::conj(st0) * a1 + ct0 * a2;
            a1 = Y.b; a2 = Y.d;
            Y.b = ct0 * a1 + st0 * a2;
            Y.d = -std::conj(st0) * a1 + ct0 * a2;
        }
        {  // rmul!(Y, G')
            cplx a1 = Y.a, a2 = Y.b;
            Y.a = a1 * c + a2 * std::conj(s);
            Y.b = -a1 * s + a2 * c;
            a1 = Y.c; a2 = Y.d;
            Y.c = a1 * c + a2 * std::conj(s);
            Y.d = -a1 * s + a2 * c;
        }
        Xs[0] = Y;
    }
    double bet[2] = {1.0, 1.0};
    for (int j = 0; j < 2; ++j) {
        cplx aj(1.0);
        scal[j] = 0.0;
        for (int l = 1; l <= k; ++l) {
            cplx z = (j == 0) ? Xs[l - 1].a : Xs[l - 1].d;
            double rhs = std::abs(z);
            int sl = 0;
            if (rhs != 0) {
                sl = (int)std::floor(std::log2(rhs));
                z *= std::exp2(-(double)sl);
            }
            if (sg(l)) {
                aj *= z;
                scal[j] += sl;
            } else if (rhs == 0) {
                bet[j] = 0.0;
            } else {
                aj /= z;
                scal[j] -= sl;
            }
            if ((l % 10 == 0) || (l == k)) {
                rhs = std::abs(aj);
                if (rhs == 0) {
                    scal[j] = 0;
                } else {
                    sl = (int)std::floor(std::log2(rhs));
                    aj *= std::exp2(-(double)sl);
                    scal[j] += sl;
                }
            }
        }
        alpha[j] = aj;
    }
    if (alpha[1].imag() > 0) {
        std::swap(alpha[0], alpha[1]);
        std::swap(scal[0], scal[1]);
        std::swap(bet[0], bet[1]);
    }
    if (beta) {
        beta[0] = bet[0];
        beta[1] = bet[1];
    }
    good = sanitize_reigpair(alpha, scal);
}

// ordschur.jl:122-204 _updateλ! (real, strict): Tu user-order full list, T1 at `schurindex`
inline int update_lambda_real(int n, int p, std::vector<MatT<double>>& Tu, char orient, int schurindex, cplx* lam) {
    MatT<double>& A1 = Tu[schurindex];
    std::vector<MatT<double>> As;  // P.T in order
    for (int l = 1; l <= p; ++l)
        if (l != schurindex) As.push_back(Tu[l]);
    int j = 1;
    while (j <= n) {
        bool pair = false;
        for (int l = 1; l <= p; ++l)
            if (j < n && std::fabs(Tu[l](j + 1, j)) > 0) {
                if (l == schurindex) pair = true;
                else return -88;  // "unexpected subdiag in triang factor"
            }
        if (pair) {
            std::vector<SM> Xs(p, SM(2, 2));
            auto blk = [&](const MatT<double>& M) {
                SM b(2, 2);
                for (int a = 0; a < 2; ++a)
                    for (int c = 0; c < 2; ++c) b(a, c) = M(j + a, j + c);
                return b;
            };
            Xs[0] = blk(A1);
            // ordschur.jl:150-176 with si in (1, p): 'L' -> reversed As, 'R' -> As in order
            for (int l = 1; l <= p - 1; ++l) Xs[l] = blk((orient == 'L') ? As[p - 1 - l] : As[l - 1]);
            cplx alpha[2];
            double scal[2];
            bool cvg, good;
            rpeigvals2x2(p, Xs, alpha, scal, cvg, good);
            lam[j - 1] = alpha[0] * std::exp2(scal[0]);
            lam[j] = alpha[1] * std::exp2(scal[1]);
            j += 2;
        } else {
            double v = 1.0;  // _safeprod with all true, reduced to the plain scaled product
            int sc = 0;
            v = A1(j, j);
            for (auto& M : As) {
                v *= M(j, j);
                if (v != 0) {
                    int e;
                    v = std::frexp(v, &e);
                    sc += e;
                }
            }
            lam[j - 1] = cplx(std::ldexp(v, sc), 0.0);
            j += 1;
        }
    }
    return 0;
}

// rordschur.jl:3-132 ordschur!(P, select) real.  Tu/Zu user-order full lists.
// Returns 0, 2000+jsrc (IllConditionedException), 3000 singular, -6 bad schurindex.
inline int rordschur(int n, int p, std::vector<MatT<double>>& Tu, std::vector<MatT<double>>& Zu, bool wantZ, char orient,
                     int schurindex, const uint8_t* select, cplx* lam, int64_t* nswaps) {
    std::vector<MatT<double>> F(p + 1), Zl(p + 1);
    int ks = schurindex;
    if (orient == 'R') {
        for (int l = 1; l <= p; ++l) F[l] = Tu[p + 1 - l];
        if (wantZ) {
            Zl[1] = Zu[1];
            for (int l = 2; l <= p; ++l) Zl[l] = Zu[p + 2 - l];
        }
        ks = p + 1 - ks;
    } else {
        for (int l = 1; l <= p; ++l) F[l] = Tu[l];
        if (wantZ)
            for (int l = 1; l <= p; ++l) Zl[l] = Zu[l];
    }
    std::vector<MatT<double>> X(p + 1), Zx(p + 1);
    if (ks == 1) {
        X = F;
        Zx = Zl;
    } else if (ks == p) {
        X[1] = F[p];
        for (int l = 2; l <= p; ++l) X[l] = F[l - 1];
        if (wantZ) {
            Zx[1] = Zl[p];
            for (int l = 2; l <= p; ++l) Zx[l] = Zl[l - 1];
        }
    } else {
        return -6;
    }
    MatT<double>& A1 = X[1];
    if (nswaps) *nswaps = 0;
    int jdest = 0;
    bool pair = false;
    for (int j = 1; j <= n; ++j) {  // rordschur.jl:77-110
        if (pair) {
            pair = false;
            continue;
        }
        bool swap = select[j - 1] != 0;
        if (j < n && A1(j + 1, j) != 0) {
            pair = true;
            swap = swap || (select[j] != 0);
        }
        if (swap) {
            jdest += 1;
            int jsrc = j;
            if (j != jdest) {
                int jd = jdest;
                int ok = moveblock(n, p, X, Zx, wantZ, jsrc, jd, nswaps);
                if (ok == -1) return 3000;
                if (ok != 1) return 2000 + jsrc;
                jdest = jd;
            }
            if (pair) jdest += 1;
        }
    }
    int rc = update_lambda_real(n, p, Tu, orient, schurindex, lam);
    if (rc != 0) return rc;
    pair = false;
    for (int j = 1; j <= n; ++j) {  // rordschur.jl:117-130 (A1 here is Px.T1 == P.T1)
        if (pair) {
            pair = false;
            continue;
        }
        pair = lam[j - 1].imag() != 0;
        const int j0 = pair ? j + 2 : j + 1;
        for (int r = j0; r <= n; ++r) A1(r, j) = 0.0;
        if (pair)
            for (int r = j0; r <= n; ++r) A1(r, j + 1) = 0.0;
    }
    return 0;
}

}  // namespace psdo
