// Row-wise periodic Hessenberg reduction for the left orientation — SURVEY.md section 8, row a21.
//
// Replaces _rphessenberg!(Ap, A, Q) of /root/reference/src/rhessx.jl:55-109 with its row reflectors
// (`RHouseholder`, lmul! :19-34, rmul!(., H') :36-52; reflector generation _xreflector! householder.jl:66-156).
// The reference calls it from the Krylov driver only (krylov.jl:809) on the small projected system
// ((k+1) x k and k x k blocks, k of the order of the Krylov dimension), so ONE workgroup walks the whole chain
// (i = n..2, l = p-1..1, then the Ap step) on matrices that stay in L2; the row/column updates of a step run one
// thread per column resp. row.
#pragma once
#include "psd_complex.h"
#include "psd_hess.h"

#define PSD_RH_NT 256

struct psd_rh_real {
    typedef double T;
    static PSD_HD T zero() { return 0.0; }
    static PSD_HD T cj(T a) { return a; }
    static PSD_HD T mul(T a, T b) { return a * b; }
    static PSD_HD T add(T a, T b) { return a + b; }
    static PSD_HD T sub(T a, T b) { return a - b; }
    static PSD_HD T neg(T a) { return -a; }
    static PSD_HD double abs1(T a) { return fabs(a); }
    static PSD_HD double nrm2s(T a, double sc) {
        const double y = a / sc;
        return y * y;
    }
};
struct psd_rh_cplx {
    typedef psd_z T;
    static PSD_HD T zero() { return zmk(0.0, 0.0); }
    static PSD_HD T cj(T a) { return zconj(a); }
    static PSD_HD T mul(T a, T b) { return zmul(a, b); }
    static PSD_HD T add(T a, T b) { return zadd(a, b); }
    static PSD_HD T sub(T a, T b) { return zsub(a, b); }
    static PSD_HD T neg(T a) { return zneg(a); }
    static PSD_HD double abs1(T a) { return zabs1(a); }
    static PSD_HD double nrm2s(T a, double sc) {
        const double yr = a.re / sc, yi = a.im / sc;
        return yr * yr + yi * yi;
    }
};

// scaled 2-norm of x[1..L-1] (householder.jl:5-56), block-wide
template <class O>
PSD_D double psd_rh_tailnorm(const typename O::T* x, int L, double* red) {
    const int NT = PSD_NTHREADS;
    PSD_PAR_FOR(t, NT) {
        double a = 0.0;
        for (int q = 1 + t; q < L; q += NT) a = fmax(a, O::abs1(x[q]));
        red[t] = a;
    }
    PSD_SYNC();
    const double amax = psd_block_max(red, NT);
    if (!(amax > 0.0)) return 0.0;
    PSD_PAR_FOR(t, NT) {
        double s = 0.0;
        for (int q = 1 + t; q < L; q += NT) s += O::nrm2s(x[q], amax);
        red[t] = s;
    }
    PSD_SYNC();
    return amax * sqrt(psd_block_sum(red, NT));
}

// _xreflector! on the LDS vector x[0..L-1]: x <- (beta, v); returns tau.  householder.jl:66-108
PSD_D double psd_rh_reflector(double* x, int L, double* red) {
    if (L <= 1) return 0.0;
    double xnorm = psd_rh_tailnorm<psd_rh_real>(x, L, red);
    if (xnorm == 0.0) return 0.0;  // :74-76: H = I
    const double sfmin = 2.0 * PSD_DBL_MIN / PSD_DBL_EPS;
    double alpha = x[0];
    double beta = -copysign(hypot(alpha, xnorm), alpha);
    int kount = 0;
    double acc = 1.0;
    if (fabs(beta) < sfmin) {
        const double rsfmin = 1.0 / sfmin;
        bool smallb = true;
        while (smallb) {
            kount += 1;
            acc *= rsfmin;
            beta *= rsfmin;
            alpha *= rsfmin;
            smallb = (fabs(beta) < sfmin) && (kount < 20);
        }
        xnorm *= acc;
        beta = -copysign(hypot(alpha, xnorm), alpha);
    }
    const double tau = (beta - alpha) / beta;
    const double mult = acc * (1.0 / (alpha - beta));
    for (int q = 0; q < kount; ++q) beta *= sfmin;
    PSD_SYNC();
    PSD_PAR_FOR(q, L - 1) { x[1 + q] *= mult; }
    PSD_ONE { x[0] = beta; }
    PSD_SYNC();
    return tau;
}
// complex: householder.jl:110-156 (zlarfg: beta real, tau complex, also for L = 1)
PSD_D psd_z psd_rh_reflector(psd_z* x, int L, double* red) {
    if (L < 1) return zmk(0.0, 0.0);
    double xnorm = psd_rh_tailnorm<psd_rh_cplx>(x, L, red);
    double ar = x[0].re, ai = x[0].im;
    if (xnorm == 0.0 && ai == 0.0) return zmk(0.0, 0.0);  // :121-123
    const double sfmin = PSD_DBL_MIN / PSD_DBL_EPS;
    double w = fmax(fabs(ar), fmax(fabs(ai), xnorm));
    double beta = -copysign(w * sqrt((ar / w) * (ar / w) + (ai / w) * (ai / w) + (xnorm / w) * (xnorm / w)), ar);
    int kount = 0;
    double acc = 1.0;
    if (fabs(beta) < sfmin) {
        const double rsfmin = 1.0 / sfmin;
        bool smallb = true;
        while (smallb) {
            kount += 1;
            acc *= rsfmin;
            beta *= rsfmin;
            ar *= rsfmin;
            ai *= rsfmin;
            smallb = (fabs(beta) < sfmin) && (kount < 20);
        }
        xnorm *= acc;
        w = fmax(fabs(ar), fmax(fabs(ai), xnorm));
        beta = -copysign(w * sqrt((ar / w) * (ar / w) + (ai / w) * (ai / w) + (xnorm / w) * (xnorm / w)), ar);
    }
    const psd_z tau = zmk((beta - ar) / beta, -ai / beta);
    const psd_z mult = zscal(acc, zdiv(zmk(1.0, 0.0), zmk(ar - beta, ai)));
    for (int q = 0; q < kount; ++q) beta *= sfmin;
    PSD_SYNC();
    PSD_PAR_FOR(q, L - 1) { x[1 + q] = zmul(x[1 + q], mult); }
    PSD_ONE { x[0] = zmk(beta, 0.0); }
    PSD_SYNC();
    return tau;
}

template <class T> struct psd_rh_mat {
    T* a;
    int ld;
    PSD_HD T& operator()(int r, int c) const { return a[(size_t)(c - 1) * ld + (r - 1)]; }
};

// One step: reflector from row `i` of R against its columns L..1 (pivot at column L); H on the rows 1..L of ML (all
// ncl columns), H' on the columns 1..L of R (all nrr rows) and of MQ (nq rows, if any).  x: LDS, L entries.
template <class O>
PSD_D void psd_rh_step(const psd_rh_mat<typename O::T>& R, int i, int L, int nrr, const psd_rh_mat<typename O::T>& ML,
                       int ncl, const psd_rh_mat<typename O::T>* MQ, int nq, typename O::T* x, double* red) {
    typedef typename O::T T;
    PSD_SYNC();
    PSD_PAR_FOR(k, L) { x[k] = O::cj(R(i, L - k)); }  // conj.(A[i, L:-1:1])
    PSD_SYNC();
    const T tau = psd_rh_reflector(x, L, red);
    const T tc = O::cj(tau);
    // v for column/row c (1 <= c < L) is x[L - c]  (xi[L:-1:2])
    // lmul!(H, view(ML, 1:L, :))  rhessx.jl:19-34
    PSD_PAR_FOR(jj, ncl) {
        const int j = jj + 1;
        T va = ML(L, j);
        for (int r = 1; r < L; ++r) va = O::add(va, O::mul(O::cj(x[L - r]), ML(r, j)));
        va = O::mul(tc, va);
        ML(L, j) = O::sub(ML(L, j), va);
        for (int r = 1; r < L; ++r) ML(r, j) = O::sub(ML(r, j), O::mul(va, x[L - r]));
    }
    PSD_SYNC();
    // rmul!(view(R, :, 1:L), H')  rhessx.jl:36-52
    PSD_PAR_FOR(rr, nrr) {
        const int r = rr + 1;
        T xx = R(r, L);
        for (int c = 1; c < L; ++c) xx = O::add(xx, O::mul(R(r, c), x[L - c]));
        R(r, L) = O::sub(R(r, L), O::mul(tau, xx));
        const T xt = O::mul(xx, O::neg(tau));
        for (int c = 1; c < L; ++c) R(r, c) = O::add(R(r, c), O::mul(xt, O::cj(x[L - c])));
    }
    if (MQ) {
        const psd_rh_mat<T>& Q = *MQ;
        PSD_PAR_FOR(rr, nq) {
            const int r = rr + 1;
            T xx = Q(r, L);
            for (int c = 1; c < L; ++c) xx = O::add(xx, O::mul(Q(r, c), x[L - c]));
            Q(r, L) = O::sub(Q(r, L), O::mul(tau, xx));
            const T xt = O::mul(xx, O::neg(tau));
            for (int c = 1; c < L; ++c) Q(r, c) = O::add(Q(r, c), O::mul(xt, O::cj(x[L - c])));
        }
    }
    PSD_SYNC();
}

// Ap: m x n (ld m); A: [p-1][n][n]; Q: [p][nq][nqc] or null.  One workgroup.
template <class O>
PSD_KERNEL psd_rphess_kernel(typename O::T* Ap, typename O::T* A, typename O::T* Q, int m, int n, int p, int nq, int nqc) {
    typedef typename O::T T;
    PSD_LDS_DECL;
    double* red = (double*)psd_lds;                       // PSD_RH_NT doubles
    T* x = (T*)(psd_lds + sizeof(double) * PSD_RH_NT);    // n + 1 entries
    const psd_rh_mat<T> Apm{Ap, m};
    auto fac = [&](int l) { return psd_rh_mat<T>{A + (size_t)(l - 1) * n * n, n}; };
    auto qm = [&](int l) { return psd_rh_mat<T>{Q + (size_t)(l - 1) * nq * nqc, nq}; };
    auto ap_step = [&](int i) {  // rhessx.jl:67-77, 93-103
        const int i1 = i - 1;
        if (i1 < 1) return;
        const psd_rh_mat<T> Ax = (p == 1) ? Apm : fac(p - 1);
        psd_rh_mat<T> qp{nullptr, 0};
        if (Q) qp = qm(p);
        psd_rh_step<O>(Apm, i, i1, m, Ax, n, Q ? &qp : nullptr, nq, x, red);
    };
    if (m == n + 1) ap_step(n + 1);
    for (int i = n; i >= 2; --i) {
        for (int l = p - 1; l >= 1; --l) {  // :82-92
            const psd_rh_mat<T> Al = fac(l);
            const psd_rh_mat<T> Al1 = (l == 1) ? Apm : fac(l - 1);
            psd_rh_mat<T> ql{nullptr, 0};
            if (Q) ql = qm(l);
            psd_rh_step<O>(Al, i, i, n, Al1, n, Q ? &ql : nullptr, nq, x, red);
        }
        ap_step(i);
    }
    PSD_SYNC();
    PSD_PAR_FOR(c, n) {  // triu!(Ap, -1)  :104
        for (int r = c + 3; r <= m; ++r) Apm(r, c + 1) = O::zero();
    }
    for (int l = 1; l <= p - 1; ++l) {  // :105-107
        const psd_rh_mat<T> Al = fac(l);
        PSD_PAR_FOR(c, n) {
            for (int r = c + 2; r <= n; ++r) Al(r, c + 1) = O::zero();
        }
    }
}
