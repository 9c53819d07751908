// ORACLE — TEST INFRASTRUCTURE ONLY (see psd_oracle_real.hpp for the rules).
// C ABI over the CPU restatement so that tests/ and bench.py's cpu_baseline leg can drive it with
// ctypes.  All matrix arguments are packed [p][n][n], each n x n block column-major (ld = n),
// factor j (1-based, user order) at offset (j-1)*n*n.
#include "psd_oracle_complex.hpp"
#include "psd_oracle_ord.hpp"
#include "psd_oracle_rord.hpp"
#include "psd_oracle_sghess.hpp"
#include "psd_oracle_rgen.hpp"
#include "psd_oracle_grord.hpp"
#include "psd_oracle_rhessx.hpp"

#include <chrono>
#ifdef _OPENMP
#include <omp.h>
#endif

using namespace psdo;

#ifdef _OPENMP
// the restatement is serial unless bench.py's all-core baseline asks for threads (psdo_set_threads)
namespace {
struct OneThreadByDefault {
    OneThreadByDefault() { omp_set_num_threads(1); }
} one_thread_by_default;
}  // namespace
#endif

extern "C" {
double psdo_dbg_maxdust(int reset) { double v = dbg_maxdust; if (reset) dbg_maxdust = 0; return v; }
int psdo_dbg_dustpos() { return dbg_dustpos; }

// PSD.jl:213-259 phessenberg!(A): A overwritten LAPACK-style, tau is [p][n].
int psdo_d_phessenberg(int n, int p, double* A, double* tau) {
    std::vector<MatD> Av(p + 1);
    for (int j = 1; j <= p; ++j) Av[j] = MatD{A + (size_t)(j - 1) * n * n, n};
    std::vector<std::vector<double>> t;
    phessenberg(n, p, Av, t);
    for (int j = 1; j <= p; ++j)
        for (int i = 1; i <= n; ++i) tau[(size_t)(j - 1) * n + (i - 1)] = t[j][i];
    return 0;
}

// bounded sample: only the first ncols columns of the reduction (bench.py cpu_baseline)
int psdo_d_phessenberg_cols(int n, int p, double* A, double* tau, int ncols) {
    std::vector<MatD> Av(p + 1);
    for (int j = 1; j <= p; ++j) Av[j] = MatD{A + (size_t)(j - 1) * n * n, n};
    std::vector<std::vector<double>> t;
    phessenberg(n, p, Av, t, ncols);
    for (int j = 1; j <= p; ++j)
        for (int i = 1; i <= n; ++i) tau[(size_t)(j - 1) * n + (i - 1)] = t[j][i];
    return 0;
}
// cap < 0: no cap.  With a cap, psdo_d_pschur_hess returns -77 after that many QR sweeps (H, Z left mid-iteration).
void psdo_set_sweep_cap(long long cap) {
    g_sweep_cap = cap;
    g_sweeps_done = 0;
}
// threads of the Householder applications (1 = the serial restatement; the reference itself is serial apart from BLAS)
void psdo_set_threads(int nthreads) {
#ifdef _OPENMP
    omp_set_num_threads(nthreads < 1 ? 1 : nthreads);
#else
    (void)nthreads;
#endif
}
int psdo_get_max_threads() {
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}

// Explicit Q factors of a phessenberg! result (what PSD.jl:136-143 builds).  Q is [p][n][n].
int psdo_d_hessenberg_q(int n, int p, const double* A, const double* tau, double* Q) {
    for (int j = 1; j <= p; ++j) {
        std::vector<double> tj(n + 1, 0.0);
        for (int i = 1; i <= n; ++i) tj[i] = tau[(size_t)(j - 1) * n + (i - 1)];
        MatD Aj{const_cast<double*>(A) + (size_t)(j - 1) * n * n, n};
        MatD Qj{Q + (size_t)(j - 1) * n * n, n};
        materializeQ(n, j, Aj, tj, Qj);
    }
    return 0;
}

// PSD.jl:322-1096 pschur!(H1, Hs; wantT, wantZ, Q): H is [p][n][n] with H[0] Hessenberg, the rest
// upper triangular (internal order, no orientation handling); Z in/out (Q on entry or identity).
// sweeplog: optional [3*maxlog] int32 (kind,l,i per inner iteration); nlog receives the count.
// Returns info: 0 ok, >0 level at which convergence failed.
int psdo_d_pschur_hess(int n, int p, double* H, double* Z, int wantT, int wantZ, int maxitfac,
                       double* wr, double* wi, int64_t* niter, int32_t* sweeplog, int64_t maxlog,
                       int64_t* nlog) {
    std::vector<MatD> Hv(p + 1), Zv(p + 1);
    for (int j = 1; j <= p; ++j) {
        Hv[j] = MatD{H + (size_t)(j - 1) * n * n, n};
        Zv[j] = MatD{Z ? Z + (size_t)(j - 1) * n * n : nullptr, n};
    }
    std::vector<std::complex<double>> lam(n);
    SweepLog log;
    int info = pschur_hess(n, p, Hv, Zv, wantT != 0, wantZ != 0 && Z, maxitfac, lam.data(), niter,
                           &log);
    for (int q = 0; q < n; ++q) {
        wr[q] = lam[q].real();
        wi[q] = lam[q].imag();
    }
    int64_t cnt = (int64_t)log.rec.size() / 3;
    if (nlog) *nlog = cnt;
    if (sweeplog)
        for (int64_t q = 0; q < std::min(cnt, maxlog) * 3; ++q) sweeplog[q] = log.rec[q];
    return info;
}

// PSD.jl:120-152 pschur!(A, lr; wantZ, wantT, maxitfac), real.
// A [p][n][n] in user order: on exit slot s holds the user-order factor T_s; the quasi-triangular
// factor sits in slot *schurindex (1 for 'R', p for 'L' — PSD.jl:1092-1094).  Z [p][n][n] in user
// order (ignored if !wantZ).  phase_ms: optional [3] = {hessenberg, q formation, iteration} wall ms.
int psdo_d_pschur(int n, int p, double* A, char orient, int wantT, int wantZ, int maxitfac, double* Z,
                  double* wr, double* wi, int* schurindex, int64_t* niter, int32_t* sweeplog,
                  int64_t maxlog, int64_t* nlog, double* phase_ms) {
    if (orient != 'R' && orient != 'L') return -4;  // PSD.jl:175-177 ArgumentError
    const bool left = orient == 'L';
    auto slotA = [&](int j) { return left ? (p + 1 - j) : j; };           // PSD.jl:127-131
    auto slotZ = [&](int j) { return (!left || j == 1) ? j : (p + 2 - j); };  // PSD.jl:1079-1084
    std::vector<MatD> Av(p + 1), Zv(p + 1);
    for (int j = 1; j <= p; ++j) {
        Av[j] = MatD{A + (size_t)(slotA(j) - 1) * n * n, n};
        Zv[j] = MatD{(wantZ && Z) ? Z + (size_t)(slotZ(j) - 1) * n * n : nullptr, n};
    }
    auto t0 = std::chrono::steady_clock::now();
    std::vector<std::vector<double>> tau;
    phessenberg(n, p, Av, tau);
    auto t1 = std::chrono::steady_clock::now();
    if (wantZ && Z)
        for (int j = 1; j <= p; ++j) materializeQ(n, j, Av[j], tau[j], Zv[j]);
    // PSD.jl:147,149: Hs = R factors (triu), H1 = triu(H1,-1)
    for (int j = 1; j <= p; ++j) {
        const int sub = (j == 1) ? 1 : 0;
        for (int c = 1; c <= n; ++c)
            for (int r = c + sub + 1; r <= n; ++r) Av[j](r, c) = 0.0;
    }
    auto t2 = std::chrono::steady_clock::now();
    std::vector<std::complex<double>> lam(n);
    SweepLog log;
    int info = pschur_hess(n, p, Av, Zv, wantT != 0, wantZ != 0 && Z, maxitfac, lam.data(), niter, &log);
    auto t3 = std::chrono::steady_clock::now();
    for (int q = 0; q < n; ++q) {
        wr[q] = lam[q].real();
        wi[q] = lam[q].imag();
    }
    if (schurindex) *schurindex = left ? p : 1;
    int64_t cnt = (int64_t)log.rec.size() / 3;
    if (nlog) *nlog = cnt;
    if (sweeplog)
        for (int64_t q = 0; q < std::min(cnt, maxlog) * 3; ++q) sweeplog[q] = log.rec[q];
    if (phase_ms) {
        auto ms = [](auto a, auto b) { return std::chrono::duration<double, std::milli>(b - a).count(); };
        phase_ms[0] = ms(t0, t1);
        phase_ms[1] = ms(t1, t2);
        phase_ms[2] = ms(t2, t3);
    }
    return info;
}

// ---------------------------------------------------------------------------------------------
// complex path.  Matrices are interleaved (re, im) doubles, [p][n][n] column-major blocks.

// generalized.jl:166-931 pschur!(H1, Hs, S; wantT, wantZ, Q): internal order, general signatures.
// alpha [n] interleaved, beta [n], ascale [n]: values = alpha/beta*2^ascale (generalized.jl:74-76).
int psdo_z_pschur_hess(int n, int p, double* H, const uint8_t* S, double* Z, int wantT, int wantZ, int maxitfac,
                       double* alpha, double* beta, int32_t* ascale, int64_t* niter, int32_t* zlog, int64_t maxlog,
                       int64_t* nlog) {
    std::vector<MatZ> Hv(p + 1), Zv(p + 1);
    std::vector<char> Sv(p + 1, 1);
    for (int j = 1; j <= p; ++j) {
        Hv[j] = MatZ{reinterpret_cast<cplx*>(H) + (size_t)(j - 1) * n * n, n};
        Zv[j] = MatZ{Z ? reinterpret_cast<cplx*>(Z) + (size_t)(j - 1) * n * n : nullptr, n};
        if (S) Sv[j] = S[j - 1] ? 1 : 0;
    }
    if (!Sv[1]) return -5;  // generalized.jl:182
    std::vector<int> sc(n, 0);
    ZLog log;
    int info = pschur_hess_z(n, p, Hv, Sv, Zv, wantT != 0, wantZ != 0 && Z, maxitfac, reinterpret_cast<cplx*>(alpha),
                             beta, sc.data(), niter, &log);
    for (int q = 0; q < n; ++q) ascale[q] = sc[q];
    int64_t cnt = (int64_t)log.rec.size() / 3;
    if (nlog) *nlog = cnt;
    if (zlog)
        for (int64_t q = 0; q < std::min(cnt, maxlog) * 3; ++q) zlog[q] = log.rec[q];
    return info;
}

// PSD.jl:213-259 for ComplexF64: packed reflectors + tau [p][n] (interleaved)
int psdo_z_phessenberg(int n, int p, double* A, double* tau, double* Q) {
    std::vector<MatZ> Av(p + 1);
    for (int j = 1; j <= p; ++j) Av[j] = MatZ{reinterpret_cast<cplx*>(A) + (size_t)(j - 1) * n * n, n};
    std::vector<std::vector<cplx>> t;
    phessenbergz(n, p, Av, t);
    cplx* tz = reinterpret_cast<cplx*>(tau);
    for (int j = 1; j <= p; ++j)
        for (int i = 1; i <= n; ++i) tz[(size_t)(j - 1) * n + (i - 1)] = t[j][i];
    if (Q)
        for (int j = 1; j <= p; ++j)
            materializeQz(n, j, Av[j], t[j], MatZ{reinterpret_cast<cplx*>(Q) + (size_t)(j - 1) * n * n, n});
    return 0;
}

// PSD.jl:1106-1111 -> generalized.jl:108-148 with all(S): pschur!(A::Vector{Matrix{ComplexF64}}, lr).
// User-order in/out as in psdo_d_pschur.
int psdo_z_pschur(int n, int p, double* A, char orient, int wantT, int wantZ, int maxitfac, double* Z, double* alpha,
                  double* beta, int32_t* ascale, int* schurindex, int64_t* niter, int32_t* zlog, int64_t maxlog,
                  int64_t* nlog, double* phase_ms) {
    if (orient != 'R' && orient != 'L') return -4;
    const bool left = orient == 'L';
    auto slotA = [&](int j) { return left ? (p + 1 - j) : j; };
    auto slotZ = [&](int j) { return (!left || j == 1) ? j : (p + 2 - j); };
    std::vector<MatZ> Av(p + 1), Zv(p + 1);
    std::vector<char> Sv(p + 1, 1);
    for (int j = 1; j <= p; ++j) {
        Av[j] = MatZ{reinterpret_cast<cplx*>(A) + (size_t)(slotA(j) - 1) * n * n, n};
        Zv[j] = MatZ{(wantZ && Z) ? reinterpret_cast<cplx*>(Z) + (size_t)(slotZ(j) - 1) * n * n : nullptr, n};
    }
    auto t0 = std::chrono::steady_clock::now();
    std::vector<std::vector<cplx>> tau;
    phessenbergz(n, p, Av, tau);
    auto t1 = std::chrono::steady_clock::now();
    if (wantZ && Z)
        for (int j = 1; j <= p; ++j) materializeQz(n, j, Av[j], tau[j], Zv[j]);
    for (int j = 1; j <= p; ++j) {
        const int sub = (j == 1) ? 1 : 0;
        for (int c = 1; c <= n; ++c)
            for (int r = c + sub + 1; r <= n; ++r) Av[j](r, c) = cplx(0.0);
    }
    auto t2 = std::chrono::steady_clock::now();
    std::vector<int> sc(n, 0);
    ZLog log;
    int info = pschur_hess_z(n, p, Av, Sv, Zv, wantT != 0, wantZ != 0 && Z, maxitfac, reinterpret_cast<cplx*>(alpha),
                             beta, sc.data(), niter, &log);
    auto t3 = std::chrono::steady_clock::now();
    for (int q = 0; q < n; ++q) ascale[q] = sc[q];
    if (schurindex) *schurindex = left ? p : 1;
    int64_t cnt = (int64_t)log.rec.size() / 3;
    if (nlog) *nlog = cnt;
    if (zlog)
        for (int64_t q = 0; q < std::min(cnt, maxlog) * 3; ++q) zlog[q] = log.rec[q];
    if (phase_ms) {
        auto ms = [](auto a, auto b) { return std::chrono::duration<double, std::milli>(b - a).count(); };
        phase_ms[0] = ms(t0, t1);
        phase_ms[1] = ms(t1, t2);
        phase_ms[2] = ms(t2, t3);
    }
    return info;
}

// ---------------------------------------------------------------------------------------------
// ordschur!(P, select) for ComplexF64 (ordschur.jl:11-73).  T/Z [p][n][n] in user order (T1 at `schurindex`),
// mutated in place; values recomputed from the diagonals (ordschur.jl:97-120) as alpha/beta*2^scale.
int psdo_z_ordschur(int n, int p, double* Tz, double* Zz, char orient, int schurindex, const uint8_t* select, int wantZ,
                    double* alpha, double* beta, int32_t* ascale, int64_t* nswaps) {
    std::vector<MatT<cplx>> Tu(p + 1), Zu(p + 1);
    for (int l = 1; l <= p; ++l) {
        Tu[l] = MatT<cplx>{reinterpret_cast<cplx*>(Tz) + (size_t)(l - 1) * n * n, n};
        Zu[l] = MatT<cplx>{Zz ? reinterpret_cast<cplx*>(Zz) + (size_t)(l - 1) * n * n : nullptr, n};
    }
    int info = ordschur1x1<cplx>(n, p, Tu, Zu, wantZ != 0 && Zz, orient, schurindex, select, nswaps);
    if (info != 0) return info;
    std::vector<char> Sv(p + 1, 1);
    std::vector<cplx> v(p);
    cplx* al = reinterpret_cast<cplx*>(alpha);
    for (int j = 1; j <= n; ++j) {  // T1 first, then the others in order (ordschur.jl:104-110)
        int q = 0;
        for (int l = 1; l <= p; ++l)
            if (l != schurindex) v[q++] = Tu[l](j, j);
        int sc;
        safeprod(Sv, p, Tu[schurindex](j, j), v.data(), al[j - 1], beta[j - 1], sc);
        ascale[j - 1] = sc;
    }
    return 0;
}

// real ordschur! restricted to decompositions whose T1 is triangular (all eigenvalues real; rordschur.jl:3-132
// then reduces to 1x1 swaps): returns -77 if a 2x2 block is present.
int psdo_d_ordschur_real1x1(int n, int p, double* Td, double* Zd, char orient, int schurindex, const uint8_t* select,
                            int wantZ, double* wr, int64_t* nswaps) {
    std::vector<MatT<double>> Tu(p + 1), Zu(p + 1);
    for (int l = 1; l <= p; ++l) {
        Tu[l] = MatT<double>{Td + (size_t)(l - 1) * n * n, n};
        Zu[l] = MatT<double>{Zd ? Zd + (size_t)(l - 1) * n * n : nullptr, n};
    }
    for (int j = 1; j < n; ++j)
        if (Tu[schurindex](j + 1, j) != 0.0) return -77;
    int info = ordschur1x1<double>(n, p, Tu, Zu, wantZ != 0 && Zd, orient, schurindex, select, nswaps);
    if (info != 0) return info;
    for (int j = 1; j <= n; ++j) {
        double v = 1.0;
        for (int l = 1; l <= p; ++l) v *= Tu[l](j, j);
        wr[j - 1] = v;
    }
    return 0;
}

// ordschur!(P, select) for Float64 (rordschur.jl:3-132) with 1x1 and 2x2 blocks.
int psdo_d_ordschur(int n, int p, double* Td, double* Zd, char orient, int schurindex, const uint8_t* select, int wantZ,
                    double* wr, double* wi, int64_t* nswaps) {
    std::vector<MatT<double>> Tu(p + 1), Zu(p + 1);
    for (int l = 1; l <= p; ++l) {
        Tu[l] = MatT<double>{Td + (size_t)(l - 1) * n * n, n};
        Zu[l] = MatT<double>{Zd ? Zd + (size_t)(l - 1) * n * n : nullptr, n};
    }
    std::vector<cplx> lam(n);
    int info = rordschur(n, p, Tu, Zu, wantZ != 0 && Zd, orient, schurindex, select, lam.data(), nswaps);
    for (int q = 0; q < n; ++q) {
        wr[q] = lam[q].real();
        wi[q] = lam[q].imag();
    }
    return info;
}

// _phessenberg!(A, S) — generalized.jl:988-1082.  is_complex selects the element type; A, Q are [p][n][n].
int psdo_sg_phessenberg(int n, int p, int is_complex, double* A, const uint8_t* S, double* Q) {
    std::vector<char> Sv(p + 1, 1);
    for (int l = 1; l <= p; ++l) Sv[l] = S[l - 1] ? 1 : 0;
    if (!Sv[1]) return -5;  // generalized.jl:990
    if (is_complex) {
        std::vector<MatT<cplx>> Av(p + 1), Qv(p + 1);
        for (int l = 1; l <= p; ++l) {
            Av[l] = MatT<cplx>{reinterpret_cast<cplx*>(A) + (size_t)(l - 1) * n * n, n};
            Qv[l] = MatT<cplx>{reinterpret_cast<cplx*>(Q) + (size_t)(l - 1) * n * n, n};
        }
        sg_phessenberg<cplx>(n, p, Av, Sv, Qv, true);
    } else {
        std::vector<MatT<double>> Av(p + 1), Qv(p + 1);
        for (int l = 1; l <= p; ++l) {
            Av[l] = MatT<double>{A + (size_t)(l - 1) * n * n, n};
            Qv[l] = MatT<double>{Q + (size_t)(l - 1) * n * n, n};
        }
        sg_phessenberg<double>(n, p, Av, Sv, Qv, true);
    }
    return 0;
}

// ordschur!(P::GeneralizedPeriodicSchur, select) by 1x1 swaps (ordschur.jl:11-96, sylswap.jl:638-764).  T/Z [p][n][n]
// in user order (T1 at `schurindex`), S user-order signature; eigenvalues recomputed in scaled form.
int psdo_gordschur(int n, int p, int is_complex, double* Td, double* Zd, const uint8_t* S, char orient, int schurindex,
                   const uint8_t* select, int wantZ, double* alpha, double* beta, int32_t* ascale, int64_t* nswaps) {
    std::vector<char> Su(p + 1, 1);
    for (int l = 1; l <= p; ++l) Su[l] = S[l - 1] ? 1 : 0;
    cplx* al = reinterpret_cast<cplx*>(alpha);
    std::vector<cplx> v(p);
    int info;
    if (is_complex) {
        std::vector<MatT<cplx>> Tu(p + 1), Zu(p + 1);
        for (int l = 1; l <= p; ++l) {
            Tu[l] = MatT<cplx>{reinterpret_cast<cplx*>(Td) + (size_t)(l - 1) * n * n, n};
            Zu[l] = MatT<cplx>{Zd ? reinterpret_cast<cplx*>(Zd) + (size_t)(l - 1) * n * n : nullptr, n};
        }
        info = gordschur1x1<cplx>(n, p, Tu, Zu, Su, wantZ != 0 && Zd, orient, schurindex, select, nswaps);
        if (info != 0) return info;
        for (int j = 1; j <= n; ++j) {  // ordschur.jl:75-96: user order, _safeprod(P, v4ev) with P.S
            for (int l = 2; l <= p; ++l) v[l - 2] = Tu[l](j, j);
            int sc;
            safeprod(Su, p, Tu[1](j, j), v.data(), al[j - 1], beta[j - 1], sc);
            ascale[j - 1] = sc;
        }
    } else {
        std::vector<MatT<double>> Tu(p + 1), Zu(p + 1);
        for (int l = 1; l <= p; ++l) {
            Tu[l] = MatT<double>{Td + (size_t)(l - 1) * n * n, n};
            Zu[l] = MatT<double>{Zd ? Zd + (size_t)(l - 1) * n * n : nullptr, n};
        }
        // 1x1 and 2x2 blocks, signed swaps (rordschur.jl + sylswap.jl:197-538,638-764)
        info = grordschur(n, p, Tu, Zu, Su, wantZ != 0 && Zd, orient, schurindex, select, al, beta, ascale, nswaps);
        if (info != 0) return info;
    }
    return 0;
}

// pschur!(H1, Hs, S; wantT, wantZ, Q, maxitfac) for Float64 — rgeneralized.jl:49-1083.  H, Z [p][n][n]; Z holds Q on
// entry (identity for Q = nothing).  alpha is complex (2n doubles).  counters[8]: niter, sweeps, zero-shift passes,
// Case II, Case III, 2x2 real deflations, 2x2 complex blocks, iwarn.
int psdo_d_gpschur_hess(int n, int p, double* H, const uint8_t* S, double* Z, int wantT, int wantZ, int maxitfac,
                        double* alpha, double* beta, int32_t* ascale, int64_t* counters) {
    std::vector<MatT<double>> Hv(p + 1), Zv(p + 1);
    std::vector<char> Sv(p + 1, 1);
    for (int l = 1; l <= p; ++l) {
        Hv[l] = MatT<double>{H + (size_t)(l - 1) * n * n, n};
        Zv[l] = MatT<double>{(wantZ && Z) ? Z + (size_t)(l - 1) * n * n : nullptr, n};
        Sv[l] = S ? (S[l - 1] ? 1 : 0) : 1;
    }
    std::vector<int> sc(n, 0);
    RGLog log;
    int info = rg_pschur_hess(n, p, Hv, Sv, Zv, wantT != 0, wantZ != 0 && Z, maxitfac, reinterpret_cast<cplx*>(alpha),
                              beta, sc.data(), &log);
    for (int q = 0; q < n; ++q) ascale[q] = sc[q];
    if (counters) {
        counters[0] = log.niter; counters[1] = log.nsweeps; counters[2] = log.nzero; counters[3] = log.ncase2;
        counters[4] = log.ncase3; counters[5] = log.n2x2real; counters[6] = log.n2x2cplx; counters[7] = log.iwarn;
    }
    return info;
}

// pschur!(A, S, lr; wantZ, wantT) — real: rgeneralized.jl:3-45; complex: generalized.jl:108-148.  User-order in/out
// (A[l] -> T_l, Z[l]); *schurindex = 1 ('R') or p ('L').  The Hessenberg-triangular stage is _phessenberg!(A, S) for
// every signature (the reference takes the Householder phessenberg! when all(S); same contract).
int psdo_gpschur(int n, int p, int is_complex, double* A, const uint8_t* S, char orient, int wantT, int wantZ,
                 int maxitfac, double* Z, double* alpha, double* beta, int32_t* ascale, int* schurindex,
                 int64_t* counters) {
    if (orient != 'R' && orient != 'L') return -4;
    const bool left = orient == 'L';
    auto slotA = [&](int j) { return left ? (p + 1 - j) : j; };
    auto slotZ = [&](int j) { return (!left || j == 1) ? j : (p + 2 - j); };
    std::vector<char> Sv(p + 1, 1);
    for (int j = 1; j <= p; ++j) Sv[j] = S[slotA(j) - 1] ? 1 : 0;
    if (!Sv[1]) return -5;
    std::vector<int> sc(n, 0);
    int info;
    std::vector<double> Zloc;
    if (!Z) {
        Zloc.assign((size_t)p * n * n * (is_complex ? 2 : 1), 0.0);
        Z = Zloc.data();
    }
    if (is_complex) {
        std::vector<MatT<cplx>> Av(p + 1), Qv(p + 1);
        std::vector<MatZ> Hz(p + 1), Zz(p + 1);
        for (int j = 1; j <= p; ++j) {
            Av[j] = MatT<cplx>{reinterpret_cast<cplx*>(A) + (size_t)(slotA(j) - 1) * n * n, n};
            Qv[j] = MatT<cplx>{reinterpret_cast<cplx*>(Z) + (size_t)(slotZ(j) - 1) * n * n, n};
            Hz[j] = MatZ{Av[j].a, n};
            Zz[j] = MatZ{Qv[j].a, n};
        }
        sg_phessenberg<cplx>(n, p, Av, Sv, Qv, true);
        int64_t niter = 0;
        info = pschur_hess_z(n, p, Hz, Sv, Zz, wantT != 0, wantZ != 0, maxitfac, reinterpret_cast<cplx*>(alpha), beta,
                             sc.data(), &niter, nullptr);
        if (counters) counters[0] = niter;
    } else {
        std::vector<MatT<double>> Av(p + 1), Qv(p + 1);
        for (int j = 1; j <= p; ++j) {
            Av[j] = MatT<double>{A + (size_t)(slotA(j) - 1) * n * n, n};
            Qv[j] = MatT<double>{Z + (size_t)(slotZ(j) - 1) * n * n, n};
        }
        sg_phessenberg<double>(n, p, Av, Sv, Qv, true);
        RGLog log;
        info = rg_pschur_hess(n, p, Av, Sv, Qv, wantT != 0, wantZ != 0, maxitfac, reinterpret_cast<cplx*>(alpha), beta,
                              sc.data(), &log);
        if (counters) {
            counters[0] = log.niter; counters[1] = log.nsweeps; counters[2] = log.nzero; counters[3] = log.ncase2;
            counters[4] = log.ncase3; counters[5] = log.n2x2real; counters[6] = log.n2x2cplx; counters[7] = log.iwarn;
        }
    }
    for (int q = 0; q < n; ++q) ascale[q] = sc[q];
    if (schurindex) *schurindex = left ? p : 1;
    return info;
}

// _rphessenberg!(Ap, A, Q) — rhessx.jl:55-109.  Ap: m x n (ld m); A: [p-1][n][n]; Q: [p][nq][nqc] (nqc >= n) or NULL.
int psdo_rphessenberg(int m, int n, int p, int is_complex, double* Ap, double* A, double* Q, int nq, int nqc) {
    if (is_complex) {
        MatT<cplx> Apm{reinterpret_cast<cplx*>(Ap), m};
        std::vector<MatT<cplx>> Av(p), Qv;
        for (int l = 1; l <= p - 1; ++l) Av[l] = MatT<cplx>{reinterpret_cast<cplx*>(A) + (size_t)(l - 1) * n * n, n};
        if (Q) {
            Qv.resize(p + 1);
            for (int l = 1; l <= p; ++l) Qv[l] = MatT<cplx>{reinterpret_cast<cplx*>(Q) + (size_t)(l - 1) * nq * nqc, nq};
        }
        return rphessenberg<cplx>(m, n, p, Apm, Av, Qv, nq);
    }
    MatT<double> Apm{Ap, m};
    std::vector<MatT<double>> Av(p), Qv;
    for (int l = 1; l <= p - 1; ++l) Av[l] = MatT<double>{A + (size_t)(l - 1) * n * n, n};
    if (Q) {
        Qv.resize(p + 1);
        for (int l = 1; l <= p; ++l) Qv[l] = MatT<double>{Q + (size_t)(l - 1) * nq * nqc, nq};
    }
    return rphessenberg<double>(m, n, p, Apm, Av, Qv, nq);
}

}  // extern "C"
