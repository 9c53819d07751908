// ORACLE — TEST INFRASTRUCTURE ONLY (rules in psd_oracle_real.hpp).
//
// CPU restatement of the signed periodic Hessenberg-triangular reduction _phessenberg!(A, S; wantQ)
// (generalized.jl:988-1082; generic twin :1085-1179), for Float64 and ComplexF64:
//   stage 1 (:1009-1028)  for l = p..2: QR (S[l]) or RQ (!S[l]) of A_l, the unitary factor applied to A_{l-1}
//                         from the side S[l-1] dictates and accumulated in Qs[l]; triu!(A_l);
//   stage 2 (:1034-1079)  Givens Hessenberg reduction of A_1, every rotation propagated through all factors
//                         with re-triangularisation.
// The QR/RQ factors come from LAPACK geqrf/gerqf in the reference; any Householder QR/RQ gives the same
// decomposition up to the signs/phases of the rows of R, which the reference's tests (invariants) do not see.
#pragma once
#include "psd_oracle_ord.hpp"

namespace psdo {

template <class T> struct Refl;
template <> struct Refl<double> {
    static double make(double* x, int n) { return xreflector(x, n, 1); }
};
template <> struct Refl<cplx> {
    static cplx make(cplx* x, int n) { return xreflectorz(x, n, 1); }
};

// Householder QR of the n x n matrix A (in place: R in the upper triangle), explicit unitary Q (n x n)
template <class T> void qr_full(int n, MatT<T>& A, std::vector<T>& Q) {
    Q.assign((size_t)n * n, T(0.0));
    for (int i = 0; i < n; ++i) Q[(size_t)i * n + i] = T(1.0);
    for (int k = 1; k <= n; ++k) {
        const int m = n - k + 1;
        std::vector<T> x(m);
        for (int r = 0; r < m; ++r) x[r] = A(k + r, k);
        T tau = Refl<T>::make(x.data(), m);
        A(k, k) = x[0];
        for (int r = 1; r < m; ++r) A(k + r, k) = T(0.0);
        if (tau == T(0.0)) continue;
        // A[k:n, k+1:n] <- H' A ;  Q <- Q H   (H = I - tau v v^H, v = (1, x[1..]))
        for (int c = k + 1; c <= n; ++c) {
            T d = A(k, c);
            for (int r = 1; r < m; ++r) d += Sc<T>::conj(x[r]) * A(k + r, c);
            d *= Sc<T>::conj(tau);
            A(k, c) -= d;
            for (int r = 1; r < m; ++r) A(k + r, c) -= d * x[r];
        }
        for (int r = 1; r <= n; ++r) {
            T d = Q[(size_t)(k - 1) * n + (r - 1)];
            for (int q = 1; q < m; ++q) d += Q[(size_t)(k - 1 + q) * n + (r - 1)] * x[q];
            d *= tau;
            Q[(size_t)(k - 1) * n + (r - 1)] -= d;
            for (int q = 1; q < m; ++q) Q[(size_t)(k - 1 + q) * n + (r - 1)] -= d * Sc<T>::conj(x[q]);
        }
    }
}

// RQ: A = R Q (R upper triangular, in place), explicit Q.  Via QR of J A^H J.
template <class T> void rq_full(int n, MatT<T>& A, std::vector<T>& Q) {
    std::vector<T> B((size_t)n * n), Qb;
    MatT<T> Bm{B.data(), n};
    for (int r = 1; r <= n; ++r)
        for (int c = 1; c <= n; ++c) Bm(r, c) = Sc<T>::conj(A(n + 1 - c, n + 1 - r));
    qr_full<T>(n, Bm, Qb);
    Q.assign((size_t)n * n, T(0.0));
    for (int r = 1; r <= n; ++r)
        for (int c = 1; c <= n; ++c) {
            A(r, c) = Sc<T>::conj(Bm(n + 1 - c, n + 1 - r));                                   // R = J Rb^H J
            Q[(size_t)(c - 1) * n + (r - 1)] = Sc<T>::conj(Qb[(size_t)(n - r) * n + (n - c)]);  // Q = J Qb^H J
        }
}

template <class T> void givT(T f, T g, double& c, T& s, T& r);
template <> inline void givT<double>(double f, double g, double& c, double& s, double& r) { givens_algorithm(f, g, c, s, r); }
template <> inline void givT<cplx>(cplx f, cplx g, double& c, cplx& s, cplx& r) { givens_algorithm_z(f, g, c, s, r); }

template <class T> struct GivT {
    int i1, i2;
    double c;
    T s;
    GivT adj() const { return GivT{i1, i2, c, -s}; }
};
template <class T> void lmulGT(const GivT<T>& G, const MatT<T>& A, int c0, int c1) {
    for (int c = c0; c <= c1; ++c) {
        T a1 = A(G.i1, c), a2 = A(G.i2, c);
        A(G.i1, c) = G.c * a1 + G.s * a2;
        A(G.i2, c) = -Sc<T>::conj(G.s) * a1 + G.c * a2;
    }
}
template <class T> void rmulGT(const MatT<T>& A, int r0, int r1, const GivT<T>& G) {
    for (int r = r0; r <= r1; ++r) {
        T a1 = A(r, G.i1), a2 = A(r, G.i2);
        A(r, G.i1) = a1 * G.c - a2 * Sc<T>::conj(G.s);
        A(r, G.i2) = a1 * G.s + a2 * G.c;
    }
}

// generalized.jl:988-1082.  A[1..p] overwritten by H_1 (Hessenberg), H_l (upper triangular); Qs[1..p] explicit.
template <class T>
void sg_phessenberg(int n, int p, std::vector<MatT<T>>& A, const std::vector<char>& S, std::vector<MatT<T>>& Qs,
                    bool wantQ) {
    if (wantQ)
        for (int l = 1; l <= p; ++l)
            for (int c = 1; c <= n; ++c)
                for (int r = 1; r <= n; ++r) Qs[l](r, c) = (r == c) ? T(1.0) : T(0.0);
    std::vector<T> Q, tmp((size_t)n * n);
    auto QM = [&](int r, int c) -> T { return Q[(size_t)(c - 1) * n + (r - 1)]; };
    for (int l = p; l >= 2; --l) {
        MatT<T>& Am = A[l - 1];
        if (S[l]) qr_full<T>(n, A[l], Q);
        else rq_full<T>(n, A[l], Q);
        // U = Q (QR case) or Q^H (RQ case) is the new Z_l: A_{l-1} <- A_{l-1} U (S[l-1]) or U^H A_{l-1} (!S[l-1])
        auto U = [&](int r, int c) -> T { return S[l] ? QM(r, c) : Sc<T>::conj(QM(c, r)); };
        if (S[l - 1]) {
            for (int r = 1; r <= n; ++r) {
                for (int c = 1; c <= n; ++c) {
                    T s = T(0.0);
                    for (int k = 1; k <= n; ++k) s += Am(r, k) * U(k, c);
                    tmp[c - 1] = s;
                }
                for (int c = 1; c <= n; ++c) Am(r, c) = tmp[c - 1];
            }
        } else {
            for (int c = 1; c <= n; ++c) {
                for (int r = 1; r <= n; ++r) {
                    T s = T(0.0);
                    for (int k = 1; k <= n; ++k) s += Sc<T>::conj(U(k, r)) * Am(k, c);
                    tmp[r - 1] = s;
                }
                for (int r = 1; r <= n; ++r) Am(r, c) = tmp[r - 1];
            }
        }
        if (wantQ) {
            MatT<T>& Ql = Qs[l];
            for (int r = 1; r <= n; ++r) {
                for (int c = 1; c <= n; ++c) {
                    T s = T(0.0);
                    for (int k = 1; k <= n; ++k) s += Ql(r, k) * U(k, c);
                    tmp[c - 1] = s;
                }
                for (int c = 1; c <= n; ++c) Ql(r, c) = tmp[c - 1];
            }
        }
        for (int c = 1; c <= n; ++c)
            for (int r = c + 1; r <= n; ++r) A[l](r, c) = T(0.0);  // triu!
    }
    // stage 2
    std::vector<GivT<T>> Gt(n + 2);
    MatT<T>& A1 = A[1];
    for (int j = 1; j <= n - 2; ++j) {
        for (int i = n; i >= j + 2; --i) {
            double c;
            T s, r;
            givT<T>(A1(i - 1, j), A1(i, j), c, s, r);
            A1(i - 1, j) = r;
            A1(i, j) = T(0.0);
            GivT<T> G{i - 1, i, c, s};
            lmulGT<T>(G, A1, j + 1, n);
            if (wantQ) rmulGT<T>(Qs[1], 1, n, G.adj());
            Gt[i] = G;
        }
        for (int l = p; l >= 2; --l) {
            MatT<T>& Al = A[l];
            if (S[l]) {
                for (int i = n; i >= j + 2; --i) {
                    rmulGT<T>(Al, 1, i, Gt[i].adj());
                    double c;
                    T s, r;
                    givT<T>(Al(i - 1, i - 1), Al(i, i - 1), c, s, r);
                    Al(i - 1, i - 1) = r;
                    Al(i, i - 1) = T(0.0);
                    GivT<T> G{i - 1, i, c, s};
                    lmulGT<T>(G, Al, i, n);
                    Gt[i] = G;
                }
            } else {
                for (int i = n; i >= j + 2; --i) {
                    lmulGT<T>(Gt[i], Al, i - 1, n);
                    double c;
                    T s, r;
                    givT<T>(Al(i, i), Al(i, i - 1), c, s, r);
                    Al(i, i) = r;
                    Al(i, i - 1) = T(0.0);
                    GivT<T> G{i, i - 1, c, Sc<T>::conj(s)};
                    rmulGT<T>(Al, 1, i - 1, G.adj());
                    Gt[i] = GivT<T>{i - 1, i, c, -s};
                }
            }
            if (wantQ)
                for (int i = n; i >= j + 2; --i) rmulGT<T>(Qs[l], 1, n, Gt[i].adj());
        }
        for (int i = n; i >= j + 2; --i) rmulGT<T>(A1, 1, n, Gt[i].adj());
    }
}


// ---------------------------------------------------------------------------------------------------------------------
// ordschur!(P::GeneralizedPeriodicSchur, select) by adjacent 1x1 swaps (ComplexF64; Float64 with a real spectrum):
//   _swapadj1x1g!(T1, Ts, Zs, S, i1)   sylswap.jl:638-764     _pgsyl1rep / _pgsylsolve1   sylvester.jl:141-168,235-245
//   _rev_alias / _circshift for GeneralizedPeriodicSchur       utils.jl:6-21,49-72
//   _updateλ!(P::GeneralizedPeriodicSchur)                     ordschur.jl:75-96
// X[1..k] is the left-oriented sequence with X[1] = T1, S[1..k] its signature.  Returns 1 ok, 0 rejected, -1 singular.
template <class T>
int swapadj1x1g_signed(int n, int k, std::vector<MatT<T>>& X, std::vector<MatT<T>>& Zs, bool haveZ,
                       const std::vector<char>& S, int i1) {
    const int i2 = i1 + 1;
    std::vector<T> T11(k + 1), T12(k + 1), T22(k + 1);
    for (int l = 1; l <= k; ++l) {
        T11[l] = X[l](i1, i1);
        T12[l] = X[l](i1, i2);
        T22[l] = X[l](i2, i2);
    }
    auto vnorm = [&](const std::vector<T>& v) {
        double s = 0.0;
        for (int l = 1; l <= k; ++l) s = std::hypot(s, std::abs(v[l]));
        return s;
    };
    const double eps = std::numeric_limits<double>::epsilon();
    const double thresh = std::max(20.0 * hypot3(vnorm(T11), vnorm(T12), vnorm(T22)) * eps,
                                   std::numeric_limits<double>::min());
    // sylvester.jl:141-168 (dense, as the reference) and :235-245
    std::vector<T> M((size_t)k * k, T(0.0)), rhs(k);
    auto at = [&](int r, int c) -> T& { return M[(size_t)(c - 1) * k + (r - 1)]; };
    if (k == 1) {
        // the reference's two assignments hit the same cell for K = 1 and the second one wins (sylvester.jl:146-149),
        // which is not the equation; (A - B) x = -C is used here (and on the device)
        at(1, 1) = T11[1] - T22[1];
    } else if (S[k]) {
        at(1, 1) = -T22[k];
        at(1, k) = T11[k];
    } else {
        at(1, 1) = T11[k];
        at(1, k) = -T22[k];
    }
    for (int q = 1; q <= k - 1; ++q) {
        if (S[q]) {
            at(q + 1, q + 1) = -T22[q];
            at(q + 1, q) = T11[q];
        } else {
            at(q + 1, q + 1) = T11[q];
            at(q + 1, q) = -T22[q];
        }
    }
    rhs[0] = -T12[k];
    for (int q = 1; q <= k - 1; ++q) rhs[q] = -T12[q];
    if (!qr_solve<T>(k, M, rhs)) return -1;
    // 2x2 working copies as MatT over local storage
    std::vector<T> store((size_t)4 * (k + 1));
    std::vector<MatT<T>> Txx(k + 1);
    for (int l = 1; l <= k; ++l) {
        Txx[l] = MatT<T>{store.data() + 4 * l, 2};
        Txx[l](1, 1) = T11[l];
        Txx[l](1, 2) = T12[l];
        Txx[l](2, 1) = T(0.0);
        Txx[l](2, 2) = T22[l];
    }
    std::vector<GivT<T>> Gs(k + 1);
    {
        double c;
        T s, r;
        givT<T>(rhs[0], T(1.0), c, s, r);
        Gs[1] = GivT<T>{1, 2, c, s};
        rmulGT<T>(Txx[1], 1, 2, Gs[1].adj());
        if (S[k]) lmulGT<T>(Gs[1], Txx[k], 1, 2);
        else rmulGT<T>(Txx[k], 1, 2, Gs[1].adj());
    }
    for (int l = 2; l <= k; ++l) {
        double c;
        T s, r;
        if (S[l]) {
            givT<T>(rhs[l - 1], T(1.0), c, s, r);
            Gs[l] = GivT<T>{1, 2, c, s};
            rmulGT<T>(Txx[l], 1, 2, Gs[l].adj());
        } else {
            givT<T>(-rhs[l - 1], T(1.0), c, s, r);
            Gs[l] = GivT<T>{2, 1, c, Sc<T>::conj(s)};
            lmulGT<T>(Gs[l], Txx[l], 1, 2);
        }
        const int lp = l - 1;
        if (S[lp]) lmulGT<T>(Gs[l], Txx[lp], 1, 2);
        else rmulGT<T>(Txx[lp], 1, 2, Gs[l].adj());
    }
    bool ok = true;
    double ws = 0.0;
    for (int l = 1; l <= k; ++l) ws += std::abs(Txx[l](2, 1));
    if (ws > thresh) ok = false;
    {  // strong test (:700-730): Ws[l] = I * G_l'
        std::vector<T> wst((size_t)4 * (k + 1));
        std::vector<MatT<T>> Ws(k + 1);
        for (int l = 1; l <= k; ++l) {
            Ws[l] = MatT<T>{wst.data() + 4 * l, 2};
            Ws[l](1, 1) = T(1.0); Ws[l](1, 2) = T(0.0); Ws[l](2, 1) = T(0.0); Ws[l](2, 2) = T(1.0);
            rmulGT<T>(Ws[l], 1, 2, Gs[l].adj());
        }
        auto mul3 = [&](const MatT<T>& A, const MatT<T>& B, const MatT<T>& Cm, T* out) {  // A * B * Cm'
            T P[4];
            for (int r = 1; r <= 2; ++r)
                for (int c = 1; c <= 2; ++c) P[(c - 1) * 2 + (r - 1)] = A(r, 1) * B(1, c) + A(r, 2) * B(2, c);
            for (int r = 1; r <= 2; ++r)
                for (int c = 1; c <= 2; ++c)
                    out[(c - 1) * 2 + (r - 1)] = P[0 * 2 + (r - 1)] * Sc<T>::conj(Cm(c, 1)) + P[1 * 2 + (r - 1)] * Sc<T>::conj(Cm(c, 2));
        };
        double ss = 0.0;
        for (int l = 1; l <= k; ++l) {
            const int l1 = (l == k) ? 1 : l + 1;
            T R[4];
            if (S[l]) mul3(Ws[l1], Txx[l], Ws[l], R);
            else mul3(Ws[l], Txx[l], Ws[l1], R);
            double d = 0.0;
            for (int r = 1; r <= 2; ++r)
                for (int c = 1; c <= 2; ++c) d = std::hypot(d, std::abs(R[(c - 1) * 2 + (r - 1)] - X[l](i1 + r - 1, i1 + c - 1)));
            ss = std::hypot(ss, d);
        }
        if (ss > thresh) ok = false;
    }
    for (int l = 1; l <= k; ++l) {  // :731-756
        const int lp = (l == 1) ? k : l - 1;
        GivT<T> G{i1 - 1 + Gs[l].i1, i1 - 1 + Gs[l].i2, Gs[l].c, Gs[l].s};
        if (S[l]) rmulGT<T>(X[l], 1, n, G.adj());
        else lmulGT<T>(G, X[l], 1, n);
        if (S[lp]) lmulGT<T>(G, X[lp], 1, n);
        else rmulGT<T>(X[lp], 1, n, G.adj());
        if (haveZ) rmulGT<T>(Zs[l], 1, n, G.adj());
    }
    for (int l = 1; l <= k; ++l) X[l](i2, i1) = T(0.0);
    return ok ? 1 : 0;
}

// ordschur!(P::GeneralizedPeriodicSchur, select): Tu/Zu/Su are the user-order full lists (T1 at `schurindex`).
// Returns 0, 2000+j (IllConditionedException(j)), 3000 (SingularException), -77 (2x2 block met), -6 bad schurindex.
template <class T>
int gordschur1x1(int n, int p, std::vector<MatT<T>>& Tu, std::vector<MatT<T>>& Zu, const std::vector<char>& Su,
                 bool wantZ, char orient, int schurindex, const uint8_t* select, int64_t* nswaps) {
    std::vector<MatT<T>> F(p + 1), Zl(p + 1);
    std::vector<char> Sl(p + 1, 1);
    int ks = schurindex;
    if (orient == 'R') {  // utils.jl:49-72
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
    std::vector<MatT<T>> X(p + 1), Zx(p + 1);
    std::vector<char> Sx(p + 1, 1);
    if (ks == 1) {
        X = F;
        Zx = Zl;
        Sx = Sl;
    } else if (ks == p) {  // utils.jl:6-21 _circshift(P, 1)
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
    for (int j = 1; j < n; ++j)
        if (X[1](j + 1, j) != T(0.0)) return -77;
    if (nswaps) *nswaps = 0;
    int js = 0;
    for (int j = 1; j <= n; ++j) {
        if (!select[j - 1]) continue;
        js += 1;
        if (j != js) {
            for (int i = j - 1; i >= js; --i) {
                int rc = swapadj1x1g_signed<T>(n, p, X, Zx, wantZ, Sx, i);
                if (nswaps) *nswaps += 1;
                if (rc < 0) return 3000;
                if (rc == 0) return 2000 + j;
            }
        }
    }
    return 0;
}

}  // namespace psdo
