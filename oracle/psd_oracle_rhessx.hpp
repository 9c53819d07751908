// TEST INFRASTRUCTURE ONLY (see psd_oracle.cpp).  Row-wise periodic Hessenberg reduction for the left orientation:
// restatement of /root/reference/src/rhessx.jl:7-109 (`RHouseholder`, its lmul!/rmul!', `_rphessenberg!`), generic in
// the element type like the reference.  The only caller in the reference is the Krylov driver (krylov.jl), which is
// out of scope; the routine itself is row a21 of SURVEY.md section 8.
#pragma once
#include <complex>
#include <vector>

#include "psd_oracle_complex.hpp"
#include "psd_oracle_ord.hpp"
#include "psd_oracle_real.hpp"

namespace psdo {

inline double rh_conj(double x) { return x; }
inline cplx rh_conj(cplx x) { return std::conj(x); }
inline double rh_reflector(double* x, int n) { return xreflector(x, n, 1); }
inline cplx rh_reflector(cplx* x, int n) { return xreflectorz(x, n, 1); }

// rhessx.jl:22-36  lmul!(H::RHouseholder, A): A has `rows` rows (the pivot is the LAST row), all `ncols` columns
template <class T> void rh_lmul(const std::vector<T>& v, T tau, const MatT<T>& A, int rows, int ncols) {
    const T tc = rh_conj(tau);
    for (int j = 1; j <= ncols; ++j) {
        T va = A(rows, j);
        for (int r = 1; r <= rows - 1; ++r) va += rh_conj(v[r - 1]) * A(r, j);  // dot(v, Aj) conjugates v
        va = tc * va;
        A(rows, j) -= va;
        for (int r = 1; r <= rows - 1; ++r) A(r, j) -= va * v[r - 1];
    }
}
// rhessx.jl:38-52  rmul!(A, H'): A has `cols` columns (the pivot is the LAST column), all `nrows` rows
template <class T> void rh_rmul_adj(const MatT<T>& A, int nrows, int cols, const std::vector<T>& v, T tau) {
    for (int r = 1; r <= nrows; ++r) {
        T x = A(r, cols);
        for (int c = 1; c <= cols - 1; ++c) x += A(r, c) * v[c - 1];
        A(r, cols) -= tau * x;
        for (int c = 1; c <= cols - 1; ++c) A(r, c) += x * (-tau) * rh_conj(v[c - 1]);  // rankUpdate!(-tau, x, v, A1)
    }
}

// rhessx.jl:55-109.  Ap: m x n (m = n or n + 1); A[1..p-1]: n x n; Q[1..p]: nq rows, at least n columns (or empty)
template <class T>
int rphessenberg(int m, int n, int p, const MatT<T>& Ap, std::vector<MatT<T>>& A, std::vector<MatT<T>>& Q, int nq) {
    if (!(m == n || m == n + 1)) return -1;  // :62
    const bool wantQ = !Q.empty();
    std::vector<T> xi, xr;
    auto ap_step = [&](int i) {  // rows i of Ap against its columns i-1..1 (:67-77, :93-103)
        const int i1 = i - 1;
        xi.assign(i1, T(0));
        for (int k = 1; k <= i1; ++k) xi[k - 1] = rh_conj(Ap(i, i1 + 1 - k));
        const T t = rh_reflector(xi.data(), i1);
        xr.assign(i1 > 1 ? i1 - 1 : 0, T(0));
        for (int q = 1; q <= i1 - 1; ++q) xr[q - 1] = xi[i1 - q];  // xi[i1:-1:2]
        const MatT<T>& Ax = (p == 1) ? Ap : A[p - 1];
        rh_lmul(xr, t, Ax, i1, n);
        rh_rmul_adj(Ap, m, i1, xr, t);
        if (wantQ) rh_rmul_adj(Q[p], nq, i1, xr, t);
    };
    if (m == n + 1) ap_step(n + 1);
    for (int i = n; i >= 2; --i) {
        for (int l = p - 1; l >= 1; --l) {  // :82-92
            xi.assign(i, T(0));
            for (int k = 1; k <= i; ++k) xi[k - 1] = rh_conj(A[l](i, i + 1 - k));
            const T t = rh_reflector(xi.data(), i);
            xr.assign(i - 1, T(0));
            for (int q = 1; q <= i - 1; ++q) xr[q - 1] = xi[i - q];  // xi[i:-1:2]
            const MatT<T>& Al1 = (l == 1) ? Ap : A[l - 1];
            rh_lmul(xr, t, Al1, i, n);
            rh_rmul_adj(A[l], n, i, xr, t);
            if (wantQ) rh_rmul_adj(Q[l], nq, i, xr, t);
        }
        ap_step(i);
    }
    for (int c = 1; c <= n; ++c)  // triu!(Ap, -1) (:104)
        for (int r = c + 2; r <= m; ++r) Ap(r, c) = T(0);
    for (int l = 1; l <= p - 1; ++l)  // :105-107
        for (int c = 1; c <= n; ++c)
            for (int r = c + 1; r <= n; ++r) A[l](r, c) = T(0);
    return 0;
}

}  // namespace psdo
