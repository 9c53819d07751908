// ORACLE — TEST INFRASTRUCTURE ONLY (rules in psd_oracle_real.hpp).
//
// CPU restatement of ordschur!(P::GeneralizedPeriodicSchur{Float64}, select) with 1x1 and 2x2 blocks:
//   rordschur.jl:3-132, 141-251 (driver, _moveblock!; generic over the decomposition type), :261-268 (_swapschur!)
//   _swapadjqr!(T1, Ts, Zs, S, i1, p1, p2)     sylswap.jl:197-538   (signed block swap: QR of [X; I] or RQ of [I -X])
//   _filled2hess!(Txx, Ws, j0, ..., S)         sylswap.jl:159-191
//   _phess2x2!(As, i1, S)                      rpschur2x2.jl:326-359
//   _pgsyl2kron / _pgsylsolve                  sylvester.jl:53-87,207-232 (dense here; the reference uses babd.jl)
//   _updateλ!(P::GeneralizedPeriodicSchur{<:Real})   ordschur.jl:206-314
// `rq!` is MatrixFactorizations.jl's (not vendored under /root/reference): A = [0 R] Q; any orthogonal Q with that
// property gives the same swap up to an orthogonal change of basis inside the two blocks, which the reference's tests
// (invariants) do not see.
#pragma once
#include "psd_oracle_rord.hpp"
#include "psd_oracle_sghess.hpp"

namespace psdo {

// generalized block periodic Sylvester system, sylvester.jl:53-87: equation k (1-based)
//   S[k]:  A_k X_k - X_{k+1} B_k = -C_k        !S[k]:  A_k X_{k+1} - X_k B_k = -C_k        (X_{K+1} = X_1)
inline bool pgsylsolve(int K, int p1, int p2, const std::vector<SM>& A, const std::vector<SM>& B,
                       const std::vector<SM>& C, const std::vector<char>& S /*1-based*/, std::vector<SM>& X) {
    const int pp = p1 * p2, N = K * pp;
    std::vector<double> M((size_t)N * N, 0.0), y(N, 0.0);
    auto at = [&](int r, int c) -> double& { return M[(size_t)c * N + r]; };
    auto put = [&](int rowblk, int eq) {
        const int cself = (eq - 1) * pp, cnext = (eq % K) * pp, r0 = rowblk * pp;
        const int cA = S[eq] ? cself : cnext;  // kron(I, A_eq) multiplies X_eq (S) or X_{eq+1} (!S)
        const int cB = S[eq] ? cnext : cself;  // kron(B_eq', -I) multiplies the other one
        for (int j = 0; j < p2; ++j)
            for (int i = 0; i < p1; ++i) {
                const int r = r0 + j * p1 + i;
                for (int k = 0; k < p1; ++k) at(r, cA + j * p1 + k) += A[eq - 1](i, k);
                for (int k = 0; k < p2; ++k) at(r, cB + k * p1 + i) += -B[eq - 1](k, j);
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

// orthogonal q (m x m) with Xi = [0 R] q for the p1 x m matrix Xi (what `r, q = rq!(Xi)` provides)
inline SM full_q_rq(const SM& Xi) {
    const int p1 = Xi.r, m = Xi.c;
    SM B(m, p1);  // B = J_m Xi' J_p1
    for (int r = 0; r < m; ++r)
        for (int c = 0; c < p1; ++c) B(r, c) = Xi(p1 - 1 - c, m - 1 - r);
    const SM Qb = full_q(B);
    SM q(m, m);  // q = J Qb' J
    for (int r = 0; r < m; ++r)
        for (int c = 0; c < m; ++c) q(r, c) = Qb(m - 1 - c, m - 1 - r);
    return q;
}

// rpschur2x2.jl:326-359 on 2x2 copies with a signature (0-based S over the k copies)
inline void phess2x2_s(int k, std::vector<SM>& As, std::vector<SM>& Qs, const std::vector<char>& S0) {
    Qs.assign(k, SM::eye(2));
    auto lmulH = [](const HH2& h, double& x1, double& x2) {  // (x1, x2)' <- H' (x1, x2)'
        const double s = h.v1 * x1 + h.v2 * x2;
        x1 -= s * h.tau * h.v1;
        x2 -= s * h.tau * h.v2;
    };
    for (int l = 2; l <= k; ++l) {
        SM& Al = As[l - 1];
        const int lp = (l % k) + 1;
        SM& Ap = As[lp - 1];
        HH2 hr;
        if (S0[l - 1]) {
            double xi[2] = {Al(0, 0), Al(1, 0)};
            const double tau = xreflector(xi, 2, 1);
            Al(0, 0) = xi[0];
            Al(1, 0) = 0.0;
            hr = HH2{1.0, xi[1], tau};
            lmulH(hr, Al(0, 1), Al(1, 1));  // lmul!(hr', view(Al, 1:2, 2:2))
        } else {
            double xi[2] = {Al(1, 1), Al(1, 0)};
            const double tau = xreflector(xi, 2, 1);
            Al(1, 0) = 0.0;
            Al(1, 1) = xi[0];
            hr = HH2{xi[1], 1.0, tau};
            lmulH(hr, Al(0, 0), Al(0, 1));  // rmul!(view(Al, 1:1, 1:2), hr)  (same formula on a row)
        }
        SM& Q = Qs[lp - 1];
        for (int j = 0; j < 2; ++j) lmulH(hr, Q(0, j), Q(1, j));  // lmul!(hr', Qs[lp])
        if (S0[lp - 1]) {
            for (int r = 0; r < 2; ++r) lmulH(hr, Ap(r, 0), Ap(r, 1));  // rmul!(view(Ap, 1:2, 1:2), hr)
        } else {
            for (int c = 0; c < 2; ++c) lmulH(hr, Ap(0, c), Ap(1, c));  // lmul!(hr', view(Ap, 1:2, 1:2))
        }
    }
}

// sylswap.jl:159-191 with S; j0 0-based
inline void filled2hess_s(int k, std::vector<SM>& Txx, std::vector<SM>& Ws, int j0, const std::vector<char>& S0) {
    const int m = Txx[0].r, j1 = j0 + 1;
    std::vector<SM> Th(k, SM(2, 2)), H1qs;
    for (int l = 0; l < k; ++l)
        for (int a = 0; a < 2; ++a)
            for (int b = 0; b < 2; ++b) Th[l](a, b) = Txx[l](j0 + a, j0 + b);
    phess2x2_s(k, Th, H1qs, S0);
    auto colsmul = [&](SM& T, const SM& q) {  // T[:, j0:j1] <- T[:, j0:j1] q
        for (int r = 0; r < T.r; ++r) {
            const double a = T(r, j0), b = T(r, j1);
            T(r, j0) = a * q(0, 0) + b * q(1, 0);
            T(r, j1) = a * q(0, 1) + b * q(1, 1);
        }
    };
    auto rowsmulT = [&](SM& T, const SM& q) {  // T[j0:j1, :] <- q' T[j0:j1, :]
        for (int c = 0; c < T.c; ++c) {
            const double a = T(j0, c), b = T(j1, c);
            T(j0, c) = q(0, 0) * a + q(1, 0) * b;
            T(j1, c) = q(0, 1) * a + q(1, 1) * b;
        }
    };
    (void)m;
    for (int l = 1; l <= k; ++l) {
        const int lp = (l % k) + 1;
        const SM& q = H1qs[l - 1];
        const SM& qp = H1qs[lp - 1];
        SM& Tl = Txx[l - 1];
        if (S0[l - 1]) {
            colsmul(Tl, q);
            rowsmulT(Tl, qp);
        } else {
            rowsmulT(Tl, q);
            colsmul(Tl, qp);
        }
        colsmul(Ws[l - 1], q);
    }
}

// sylswap.jl:197-538.  X[1..k] left-oriented sequence (X[1] = T1), S[1..k] its signature.
// Returns 1 ok, 0 failed a stability test (transformation applied anyway, as in the reference), -1 singular.
inline int swapadjqr_s(int n, int k, std::vector<MatT<double>>& X, std::vector<MatT<double>>& Zs, bool haveZ,
                       const std::vector<char>& S, int i1, int p1, int p2) {
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
    if (!pgsylsolve(k, p1, p2, T11, T22, T12, S, Xs)) return -1;
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
    std::vector<char> S0(k);
    for (int l = 1; l <= k; ++l) S0[l - 1] = S[l];
    // Q_l (:243-301): from the QR of [X_l; I] if S[l-1], from the RQ of [I -X_l] (Q_l = q') otherwise; then
    //   X_l:     S[l]   ? Txx[l] Q_l    : Q_l' Txx[l]
    //   X_{l-1}: S[l-1] ? Q_l' Txx[l-1] : Txx[l-1] Q_l
    std::vector<SM> Qs(k);
    for (int l = 1; l <= k; ++l) {
        const int lp = (l == 1) ? k : l - 1;
        if (S[lp]) {
            SM Xi(m, p2);
            for (int a = 0; a < p1; ++a)
                for (int b = 0; b < p2; ++b) Xi(a, b) = Xs[l - 1](a, b);
            for (int b = 0; b < p2; ++b) Xi(p1 + b, b) = 1.0;
            Qs[l - 1] = full_q(Xi);
        } else {
            SM Xi(p1, m);
            for (int a = 0; a < p1; ++a) {
                Xi(a, a) = 1.0;
                for (int b = 0; b < p2; ++b) Xi(a, p1 + b) = -Xs[l - 1](a, b);
            }
            Qs[l - 1] = strans(full_q_rq(Xi));
        }
    }
    for (int l = 1; l <= k; ++l) {
        const SM& Q = Qs[l - 1];
        const int lp = (l == 1) ? k : l - 1;
        Txx[l - 1] = S[l] ? smul(Txx[l - 1], Q) : smul(strans(Q), Txx[l - 1]);
        Txx[lp - 1] = S[lp] ? smul(strans(Q), Txx[lp - 1]) : smul(Txx[lp - 1], Q);
    }
    bool ok = true;
    {  // weak test (:303-313)
        double ws = 0.0;
        for (int l = 0; l < k; ++l) {
            double s = 0.0;
            for (int a = p2; a < m; ++a)
                for (int b = 0; b < p2; ++b) s = std::hypot(s, Txx[l](a, b));
            ws = std::max(ws, s);
        }
        if (ws > thresh) ok = false;
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
        if (fillin1) filled2hess_s(k, Txx, Ws, 0, S0);
        if (fillin2) filled2hess_s(k, Txx, Ws, p2, S0);
    }
    for (int l = 1; l <= k; ++l) {  // strong test (:349-377)
        const int l1 = (l % k) + 1;
        SM Tt = Txx[l - 1];
        if (S[l]) {
            if (fillin) Tt = smul(smul(Ws[l1 - 1], Tt), strans(Ws[l - 1]));
            Tt = smul(smul(Qs[l1 - 1], Tt), strans(Qs[l - 1]));
        } else {
            if (fillin) Tt = smul(smul(Ws[l - 1], Tt), strans(Ws[l1 - 1]));
            Tt = smul(smul(Qs[l - 1], Tt), strans(Qs[l1 - 1]));
        }
        double d = 0.0;
        for (int a = 0; a < m; ++a)
            for (int b = 0; b < m; ++b) d = std::hypot(d, Tt(a, b) - X[l](i1 + a, i1 + b));
        if (d > thresh) ok = false;
    }
    auto colsmul = [&](MatT<double>& T, const SM& q) {  // T[:, i1:i3] <- T[:, i1:i3] q
        std::vector<double> tmp(m);
        for (int r = 1; r <= n; ++r) {
            for (int b = 0; b < m; ++b) {
                double s = 0.0;
                for (int a = 0; a < m; ++a) s += T(r, i1 + a) * q(a, b);
                tmp[b] = s;
            }
            for (int b = 0; b < m; ++b) T(r, i1 + b) = tmp[b];
        }
    };
    auto rowsmulT = [&](MatT<double>& T, const SM& q) {  // T[i1:i3, :] <- q' T[i1:i3, :]
        std::vector<double> tmp(m);
        for (int c = 1; c <= n; ++c) {
            for (int b = 0; b < m; ++b) {
                double s = 0.0;
                for (int a = 0; a < m; ++a) s += q(a, b) * T(i1 + a, c);
                tmp[b] = s;
            }
            for (int b = 0; b < m; ++b) T(i1 + b, c) = tmp[b];
        }
    };
    for (int l = 1; l <= k; ++l) {  // (:396-445)
        SM q = Qs[l - 1];
        if (fillin) q = smul(q, Ws[l - 1]);
        const int lp = (l == 1) ? k : l - 1;
        if (S[l]) colsmul(X[l], q);
        else rowsmulT(X[l], q);
        if (haveZ) colsmul(Zs[l], q);
        if (S[lp]) rowsmulT(X[lp], q);
        else colsmul(X[lp], q);
    }
    for (int r = i2new; r <= i3; ++r)  // sweep up the dust (:484-489)
        for (int c = i1; c <= i2new - 1; ++c) X[1](r, c) = 0.0;
    for (int l = 2; l <= k; ++l)
        for (int c = i1; c <= i3; ++c)
            for (int r = c + 1; r <= i3; ++r) X[l](r, c) = 0.0;
    return ok ? 1 : 0;
}

// rordschur.jl:141-251 with the signed swaps (same control flow as `moveblock`)
inline int moveblock_s(int n, int k, std::vector<MatT<double>>& X, std::vector<MatT<double>>& Zs, bool haveZ,
                       const std::vector<char>& S, int& jsrc, int& jdest, int64_t* nswaps) {
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
        if (nb1 == 1 && nb2 == 1) return swapadj1x1g_signed<double>(n, k, X, Zs, haveZ, S, i1);
        return swapadjqr_s(n, k, X, Zs, haveZ, S, i1, nb1, nb2);
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
                here -= 1;  // (see psd_oracle_rord.hpp: missing in the reference)
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

// ordschur!(P::GeneralizedPeriodicSchur{Float64}, select).  Tu/Zu/Su user order; eigenvalues in scaled form
// (ordschur.jl:206-314 for the supported alignments ('R', 1) and ('L', p)).
inline int grordschur(int n, int p, std::vector<MatT<double>>& Tu, std::vector<MatT<double>>& Zu,
                      const std::vector<char>& Su, bool wantZ, char orient, int schurindex, const uint8_t* select,
                      cplx* alpha, double* beta, int32_t* ascale, int64_t* nswaps) {
    std::vector<MatT<double>> F(p + 1), Zl(p + 1);
    std::vector<char> Sl(p + 1, 1);
    int ks = schurindex;
    if (orient == 'R') {
        for (int l = 1; l <= p; ++l) {
            F[l] = Tu[p + 1 - l];
            Sl[l] = Su[p + 1 - l];
        }
        if (wantZ) {
            Zl[1] = Zu[1];
            for (int l = 2; l <= p; ++l) Zl[l] = Zu[p + 2 - l];
        }
        ks = p + 1 - ks;
    } else {
        for (int l = 1; l <= p; ++l) {
            F[l] = Tu[l];
            Sl[l] = Su[l];
        }
        if (wantZ)
            for (int l = 1; l <= p; ++l) Zl[l] = Zu[l];
    }
    std::vector<MatT<double>> X(p + 1), Zx(p + 1);
    std::vector<char> Sx(p + 1, 1);
    if (ks == 1) {
        X = F;
        Zx = Zl;
        Sx = Sl;
    } else if (ks == p) {
        X[1] = F[p];
        Sx[1] = Sl[p];
        for (int l = 2; l <= p; ++l) {
            X[l] = F[l - 1];
            Sx[l] = Sl[l - 1];
        }
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
                int ok = moveblock_s(n, p, X, Zx, wantZ, Sx, jsrc, jd, nswaps);
                if (ok == -1) return 3000;
                if (ok != 1) return 2000 + jsrc;
                jdest = jd;
            }
            if (pair) jdest += 1;
        }
    }
    // ordschur.jl:206-314: sequence [T1, others...] as the product is taken for the orientation
    const bool left = orient == 'L';
    std::vector<int> seq(p);  // user indices in product order, T1 first
    std::vector<char> xS(p);
    for (int l = 0; l < p; ++l) {
        seq[l] = left ? (p - l) : (l + 1);
        xS[l] = Su[seq[l]];
    }
    int j = 1;
    while (j <= n) {
        const bool pr = (j < n) && Tu[schurindex](j + 1, j) != 0;
        if (pr) {
            std::vector<SM> Xs(p, SM(2, 2));
            for (int l = 0; l < p; ++l)
                for (int a = 0; a < 2; ++a)
                    for (int c = 0; c < 2; ++c) Xs[l](a, c) = Tu[seq[l]](j + a, j + c);
            cplx a2[2];
            double sc2[2], b2[2];
            bool cvg, good;
            rpeigvals2x2(p, Xs, a2, sc2, cvg, good, xS.data(), b2);
            for (int q = 0; q < 2; ++q) {
                alpha[j - 1 + q] = a2[q];
                beta[j - 1 + q] = b2[q];
                ascale[j - 1 + q] = (int32_t)sc2[q];
            }
            j += 2;
        } else {
            std::vector<cplx> v(p);
            for (int l = 2; l <= p; ++l) v[l - 2] = cplx(Tu[l](j, j));
            int sc;
            safeprod(Su, p, cplx(Tu[1](j, j)), v.data(), alpha[j - 1], beta[j - 1], sc);
            ascale[j - 1] = sc;
            j += 1;
        }
    }
    // rordschur.jl:117-130
    pair = false;
    for (int jj = 1; jj <= n; ++jj) {
        if (pair) {
            pair = false;
            continue;
        }
        pair = alpha[jj - 1].imag() != 0;
        const int j0 = pair ? jj + 2 : jj + 1;
        for (int r = j0; r <= n; ++r) A1(r, jj) = 0.0;
        if (pair)
            for (int r = j0; r <= n; ++r) A1(r, jj + 1) = 0.0;
    }
    return 0;
}

}  // namespace psdo
