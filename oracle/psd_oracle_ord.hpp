// ORACLE — TEST INFRASTRUCTURE ONLY (rules in psd_oracle_real.hpp).
//
// CPU restatement of the reference's reordering by adjacent 1x1 swaps:
//   ordschur!(P, select) complex          ordschur.jl:11-73
//   _swapschur1! / _swapadj1x1g!          ordschur.jl:317-322, sylswap.jl:542-635
//   _psylsolve1 / _psyl1rep / _checkqr    sylvester.jl:124-138,196-205, utils.jl:123-131
//   _rev_alias / _circshift               utils.jl:25-85
//   _updateλ! / _safeprod(P, ·)           ordschur.jl:97-120, utils.jl:90-120
// The real driver (rordschur.jl:3-132, _moveblock! :141-251) reduces to the same swap sequence when
// every block it meets is 1x1; that sub-case is restated too (2x2 blocks: not yet, returns -77).
#pragma once
#include "psd_oracle_complex.hpp"

namespace psdo {

template <class T> struct Sc;
template <> struct Sc<double> {
    static double conj(double x) { return x; }
    static void givens(double f, double g, double& c, double& s, double& r) { givens_algorithm(f, g, c, s, r); }
};
template <> struct Sc<cplx> {
    static cplx conj(cplx x) { return std::conj(x); }
    static void givens(cplx f, cplx g, double& c, cplx& s, cplx& r) { givens_algorithm_z(f, g, c, s, r); }
};

template <class T> struct MatT {
    T* a;
    int ld;
    inline T& operator()(int r, int c) const { return a[(size_t)(c - 1) * ld + (r - 1)]; }
};

// dense Householder QR solve of the K x K system (what `qr!(Zpse)`; `F \ Cr` does, sylvester.jl:196-205);
// returns false on an exactly zero pivot (utils.jl:123-131 SingularException)
template <class T> bool qr_solve(int K, std::vector<T>& M /*col-major*/, std::vector<T>& b) {
    auto at = [&](int r, int c) -> T& { return M[(size_t)c * K + r]; };
    for (int k = 0; k < K; ++k) {
        double nrm = 0.0;
        for (int r = k; r < K; ++r) nrm = std::hypot(nrm, std::abs(at(r, k)));
        if (nrm == 0.0) return false;
        T x0 = at(k, k);
        T phase = (std::abs(x0) == 0.0) ? T(1.0) : x0 / std::abs(x0);
        T alpha = -phase * nrm;
        std::vector<T> v(K - k);
        for (int r = k; r < K; ++r) v[r - k] = at(r, k);
        v[0] -= alpha;
        double vn = 0.0;
        for (auto& e : v) vn = std::hypot(vn, std::abs(e));
        if (vn > 0.0) {
            for (auto& e : v) e /= vn;
            for (int c = k; c < K; ++c) {
                T d = T(0.0);
                for (int r = k; r < K; ++r) d += Sc<T>::conj(v[r - k]) * at(r, c);
                for (int r = k; r < K; ++r) at(r, c) -= (double)2.0 * v[r - k] * d;
            }
            T d = T(0.0);
            for (int r = k; r < K; ++r) d += Sc<T>::conj(v[r - k]) * b[r];
            for (int r = k; r < K; ++r) b[r] -= (double)2.0 * v[r - k] * d;
        }
    }
    for (int k = 0; k < K; ++k)
        if (at(k, k) == T(0.0)) return false;
    for (int k = K - 1; k >= 0; --k) {
        T s = b[k];
        for (int c = k + 1; c < K; ++c) s -= at(k, c) * b[c];
        b[k] = s / at(k, k);
    }
    return true;
}

// sylswap.jl:542-635 _swapadj1x1g!: X[1..k] (X[1] = T1), Zs[1..k] or empty; returns 1 ok, 0 rejected, -1 singular
template <class T>
int swapadj1x1g(int n, int k, std::vector<MatT<T>>& X, std::vector<MatT<T>>& Zs, bool haveZ, int i1) {
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
    const double n11 = vnorm(T11), n12 = vnorm(T12), n22 = vnorm(T22);
    const double thresh = std::max(20.0 * hypot3(n11, n12, n22) * eps, std::numeric_limits<double>::min());
    struct M2 { T a, b, c, d; };  // [a b; c d]
    std::vector<M2> Txx(k + 1);
    for (int l = 1; l <= k; ++l) Txx[l] = M2{T11[l], T12[l], T(0.0), T22[l]};
    std::vector<T> Xv(k + 1);
    std::vector<double> Gc(k + 1);
    std::vector<T> Gs(k + 1);
    if (k > 1) {
        // sylvester.jl:124-138 _psyl1rep(A = T11, B = T22); rhs = -circshift(T12, 1)
        std::vector<T> M((size_t)k * k, T(0.0)), rhs(k);
        auto at = [&](int r, int c) -> T& { return M[(size_t)(c - 1) * k + (r - 1)]; };
        at(1, 1) = -T22[k];
        at(1, k) = (k == 1) ? at(1, k) + T11[k] : T11[k];
        for (int q = 1; q <= k - 1; ++q) {
            at(q + 1, q + 1) = -T22[q];
            at(q + 1, q) = T11[q];
        }
        rhs[0] = -T12[k];
        for (int q = 1; q <= k - 1; ++q) rhs[q] = -T12[q];
        if (!qr_solve<T>(k, M, rhs)) return -1;
        for (int l = 1; l <= k; ++l) Xv[l] = rhs[l - 1];
        T r;
        Sc<T>::givens(Xv[1], T(1.0), Gc[1], Gs[1], r);
    } else {
        T r;
        Sc<T>::givens(T12[1], T22[1] - T11[1], Gc[1], Gs[1], r);
    }
    auto rmulGadj = [&](M2& m, double c, T s) {  // rows of the 2x2: cols (1,2) <- rmul!(., G')
        T a1 = m.a, a2 = m.b;
        m.a = a1 * c + a2 * Sc<T>::conj(s);
        m.b = -a1 * s + a2 * c;
        a1 = m.c; a2 = m.d;
        m.c = a1 * c + a2 * Sc<T>::conj(s);
        m.d = -a1 * s + a2 * c;
    };
    auto lmulG = [&](M2& m, double c, T s) {
        T a1 = m.a, a2 = m.c;
        m.a = c * a1 + s * a2;
        m.c = -Sc<T>::conj(s) * a1 + c * a2;
        a1 = m.b; a2 = m.d;
        m.b = c * a1 + s * a2;
        m.d = -Sc<T>::conj(s) * a1 + c * a2;
    };
    rmulGadj(Txx[1], Gc[1], Gs[1]);
    lmulG(Txx[k], Gc[1], Gs[1]);
    for (int l = 2; l <= k; ++l) {
        T r;
        Sc<T>::givens(Xv[l], T(1.0), Gc[l], Gs[l], r);
        rmulGadj(Txx[l], Gc[l], Gs[l]);
        lmulG(Txx[l - 1], Gc[l], Gs[l]);
    }
    bool ok = true;
    double ws = 0.0;
    for (int l = 1; l <= k; ++l) ws += std::abs(Txx[l].c);
    if (ws > thresh) ok = false;
    {  // strong test (:589-617): W_l = G_l' (as a matrix), Txx[l] <- W_{l+1} Txx[l] W_l', compare with the originals
        auto Wm = [&](int l) { return M2{T(Gc[l]), -Gs[l] * T(1.0), Sc<T>::conj(Gs[l]), T(Gc[l])}; };
        // rmul!(I, G') gives [c -s; conj(s) c]
        double ss = 0.0;
        for (int l = 1; l <= k; ++l) {
            const int l1 = (l == k) ? 1 : l + 1;
            M2 A = Wm(l1), B = Txx[l], Cw = Wm(l);
            // P = A * B
            M2 Pm{A.a * B.a + A.b * B.c, A.a * B.b + A.b * B.d, A.c * B.a + A.d * B.c, A.c * B.b + A.d * B.d};
            // R = P * Cw'  (Cw' = conj transpose)
            M2 Ct{Sc<T>::conj(Cw.a), Sc<T>::conj(Cw.c), Sc<T>::conj(Cw.b), Sc<T>::conj(Cw.d)};
            M2 R{Pm.a * Ct.a + Pm.b * Ct.c, Pm.a * Ct.b + Pm.b * Ct.d, Pm.c * Ct.a + Pm.d * Ct.c, Pm.c * Ct.b + Pm.d * Ct.d};
            double d = 0.0;
            d = std::hypot(d, std::abs(R.a - X[l](i1, i1)));
            d = std::hypot(d, std::abs(R.b - X[l](i1, i2)));
            d = std::hypot(d, std::abs(R.c - X[l](i2, i1)));
            d = std::hypot(d, std::abs(R.d - X[l](i2, i2)));
            ss = std::hypot(ss, d);
        }
        if (ss > thresh) ok = false;
    }
    for (int l = 1; l <= k; ++l) {  // :618-628
        MatT<T>& Tl = X[l];
        MatT<T>& Tp = X[(l == 1) ? k : l - 1];
        const double c = Gc[l];
        const T s = Gs[l];
        for (int r = 1; r <= n; ++r) {
            T a1 = Tl(r, i1), a2 = Tl(r, i2);
            Tl(r, i1) = a1 * c + a2 * Sc<T>::conj(s);
            Tl(r, i2) = -a1 * s + a2 * c;
        }
        for (int cc = 1; cc <= n; ++cc) {
            T a1 = Tp(i1, cc), a2 = Tp(i2, cc);
            Tp(i1, cc) = c * a1 + s * a2;
            Tp(i2, cc) = -Sc<T>::conj(s) * a1 + c * a2;
        }
        if (haveZ) {
            MatT<T>& Zl = Zs[l];
            for (int r = 1; r <= n; ++r) {
                T a1 = Zl(r, i1), a2 = Zl(r, i2);
                Zl(r, i1) = a1 * c + a2 * Sc<T>::conj(s);
                Zl(r, i2) = -a1 * s + a2 * c;
            }
        }
    }
    for (int l = 1; l <= k; ++l) X[l](i2, i1) = T(0.0);
    return ok ? 1 : 0;
}

// ordschur!(P, select): Tu/Zu are the user-order full lists (T1 at `schurindex`).  Returns 0, or
// 2000+j (IllConditionedException(j)), 3000 (SingularException), -77 (2x2 block met, real), -6 bad schurindex.
template <class T>
int ordschur1x1(int n, int p, std::vector<MatT<T>>& Tu, std::vector<MatT<T>>& Zu, bool wantZ, char orient,
                int schurindex, const uint8_t* select, int64_t* nswaps) {
    // utils.jl:49-85 _rev_alias (orientation 'R' -> left view): full reversal of T, Z_1 kept, Z_2..p reversed
    std::vector<MatT<T>> F(p + 1), Zl(p + 1);
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
    std::vector<MatT<T>> X(p + 1), Zx(p + 1);
    if (ks == 1) {
        X = F;
        Zx = Zl;
    } else if (ks == p) {  // utils.jl:6-40 _circshift(P, 1)
        X[1] = F[p];
        for (int l = 2; l <= p; ++l) X[l] = F[l - 1];
        if (wantZ) {
            Zx[1] = Zl[p];
            for (int l = 2; l <= p; ++l) Zx[l] = Zl[l - 1];
        }
    } else {
        return -6;
    }
    if (nswaps) *nswaps = 0;
    int js = 0;
    for (int j = 1; j <= n; ++j) {
        if (!select[j - 1]) continue;
        js += 1;
        if (j != js) {
            for (int i = j - 1; i >= js; --i) {
                int rc = swapadj1x1g<T>(n, p, X, Zx, wantZ, i);
                if (nswaps) *nswaps += 1;
                if (rc < 0) return 3000;
                if (rc == 0) return 2000 + j;
            }
        }
    }
    return 0;
}

}  // namespace psdo
