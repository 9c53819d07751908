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

}  // namespace psdo
