"""CPU tier: the extended-precision build of the oracle (oracle/psd_oracle_ld.cpp: the SAME restatement compiled with
`long double`, eps = 2^-63) against the reference's generic-element-type test sets — the ones it runs for BigFloat:
test/runtests.jl:89-100 (full random, real), test/generalized.jl:68-76 (generalized Hess+UT, real and complex),
test/generalized.jl:201-214 (full complex, all-true and generalized).  SURVEY.md section 8 row f4: CPU restatement
only, never the GPU.  What is pinned: the restated algorithm is generic in the scalar type, and its accuracy follows
the working precision (invariants at a few hundred eps_ld, eigenvalues against 50-digit mpmath values)."""
import ctypes as C

import numpy as np
import pytest

import psdtest as pt

LD = np.longdouble
CLD = np.clongdouble
EPS_LD = float(np.finfo(LD).eps)  # 1.08e-19


def _vp(a):
    return a.ctypes.data_as(C.c_void_p)


def _pack(As, dt):
    n = As[0].shape[0]
    out = np.zeros((len(As), n, n), dtype=dt)
    for j, a in enumerate(As):
        out[j] = np.asarray(a, dtype=dt).T  # column-major blocks
    return out


def _unpack(P):
    return [np.array(P[j].T) for j in range(P.shape[0])]


def ld_pschur(As, lr="R"):
    """psdo_ld_d_pschur: pschur!(A, lr) in long double (PSD.jl:120-152)."""
    lib = pt.oracle_lib()
    p, n = len(As), As[0].shape[0]
    A, Z = _pack(As, LD), np.zeros((p, n, n), dtype=LD)
    wr, wi, ph = np.zeros(n, dtype=LD), np.zeros(n, dtype=LD), np.zeros(3, dtype=LD)
    si, niter, nlog = C.c_int(0), C.c_int64(0), C.c_int64(0)
    f = lib.psdo_ld_d_pschur
    f.argtypes = [C.c_int, C.c_int, C.c_void_p, C.c_char, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p,
                  C.POINTER(C.c_int), C.POINTER(C.c_int64), C.c_void_p, C.c_int64, C.POINTER(C.c_int64), C.c_void_p]
    info = f(n, p, _vp(A), lr.encode()[0:1], 1, 1, 30, _vp(Z), _vp(wr), _vp(wi), C.byref(si), C.byref(niter), None, 0,
             C.byref(nlog), _vp(ph))
    assert info == 0
    return _unpack(A), _unpack(Z), wr + 1j * wi.astype(CLD), si.value


def ld_gpschur(As, S, lr="R"):
    """psdo_ld_gpschur: pschur!(A, S, lr) in long double (real: rgeneralized.jl:3-45, complex: generalized.jl:108-148)."""
    lib = pt.oracle_lib()
    p, n = len(As), As[0].shape[0]
    cplx = any(np.iscomplexobj(a) for a in As)
    dt = CLD if cplx else LD
    A, Z = _pack(As, dt), np.zeros((p, n, n), dtype=dt)
    alpha, beta, sc = np.zeros(n, dtype=CLD), np.zeros(n, dtype=LD), np.zeros(n, dtype=np.int32)
    cnt = np.zeros(8, dtype=np.int64)
    si = C.c_int(0)
    Sarr = (C.c_uint8 * p)(*[1 if x else 0 for x in S])
    f = lib.psdo_ld_gpschur
    f.argtypes = [C.c_int, C.c_int, C.c_int, C.c_void_p, C.POINTER(C.c_uint8), C.c_char, C.c_int, C.c_int, C.c_int,
                  C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.POINTER(C.c_int), C.c_void_p]
    info = f(n, p, int(cplx), _vp(A), Sarr, lr.encode()[0:1], 1, 1, 30 if cplx else 120, _vp(Z), _vp(alpha), _vp(beta),
             _vp(sc), C.byref(si), _vp(cnt))
    assert info == 0
    lam = alpha / beta * np.exp2(sc.astype(LD))
    return _unpack(A), _unpack(Z), lam, si.value


def check_ld(As, S, Ts, Zs, lr, js, tol=100, qtol=10):
    """The invariants of test/testfuncs.jl:56-145,238-382 evaluated in long double at eps = 2^-63."""
    p, n = len(As), As[0].shape[0]
    left = lr == "L"
    real = not np.iscomplexobj(Ts[0])
    worst = 0.0
    for l in range(p):
        ln = (l + 1) % p
        Zl, Zn = Zs[l], Zs[ln]
        if bool(S[l]) != left:
            Ax = Zl @ Ts[l] @ Zn.conj().T
        else:
            Ax = Zn @ Ts[l] @ Zl.conj().T
        assert np.all(np.tril(Ts[l], -2 if (real and l == js - 1) else -1) == 0)
        orth = float(np.sqrt(np.sum(np.abs(Zl @ Zl.conj().T - np.eye(n, dtype=Zl.dtype)) ** 2)))
        assert orth < qtol * EPS_LD * n, (l, orth / EPS_LD)
        A = np.asarray(As[l], dtype=Ts[l].dtype)
        res = float(np.sqrt(np.sum(np.abs(A - Ax) ** 2)))
        a1 = float(np.max(np.sum(np.abs(A), axis=0)))
        assert res < tol * EPS_LD * max(a1, 1.0), (l, res / EPS_LD / a1)
        worst = max(worst, res / EPS_LD / a1)
    return worst


def mp_eigs(As, S, left):
    """eigenvalues of the product at 50 digits (mpmath), factors with S[l] false inverted"""
    import mpmath as mp

    mp.mp.dps = 50
    n = As[0].shape[0]

    def tomp(a):
        return mp.matrix([[mp.mpc(complex(x).real, complex(x).imag) if np.iscomplexobj(a) else mp.mpf(float(x))
                           for x in row] for row in np.asarray(a)])

    P = mp.eye(n)
    for l, a in enumerate(As):
        F = tomp(a)
        if not S[l]:
            F = F ** -1
        P = F * P if left else P * F
    ev, _ = mp.eig(P)
    return [complex(e) for e in ev], [e for e in ev]


def match_mp(lam, ev_mp, rtol):
    """greedy match of long-double eigenvalues against mpmath values, relative to the largest modulus"""
    import mpmath as mp

    scale = max(abs(e) for e in ev_mp)
    left = list(ev_mp)
    worst = mp.mpf(0)
    for z in lam:
        zm = mp.mpc(mp.mpf(float(np.real(z))) + mp.mpf(float(np.real(z) - LD(float(np.real(z))))),
                    mp.mpf(float(np.imag(z))) + mp.mpf(float(np.imag(z) - LD(float(np.imag(z))))))
        d = [abs(zm - e) for e in left]
        k = min(range(len(d)), key=lambda i: d[i])
        worst = max(worst, d[k])
        left.pop(k)
    assert worst <= rtol * scale, (float(worst / scale), rtol)
    return float(worst / scale)


@pytest.mark.parametrize("p", [1, 2, 3, 5])
def test_ld_full_real(p):
    """test/runtests.jl:89-100 with the extended element type: rand(T, 5, 5) factors, pschur_test invariants"""
    A = pt.rand_uniform_factors(5, p, seed=500 + p)
    Ts, Zs, lam, js = ld_pschur(A, "R")
    w = check_ld(A, [True] * p, Ts, Zs, "R", js, tol=32)
    assert w < 32
    _, ev = mp_eigs(A, [True] * p, left=False)
    err = match_mp(lam, ev, 1000 * EPS_LD)  # the reference's ltol = 1000 eps(T) (test/testfuncs.jl:140)
    # ... and the double build of the same code is where double precision puts it: three to four digits worse
    po = pt.oracle_pschur(A, "R")
    errd = match_mp(po.values.astype(CLD), ev, 1000 * pt.EPS)
    assert err < 1e-2 * max(errd, 1e-17)


@pytest.mark.parametrize("cplx", [False, True])
@pytest.mark.parametrize("p", [2, 3, 5])
def test_ld_generalized_hess_ut(cplx, p):
    """test/generalized.jl:68-76 for BigFloat / Complex{BigFloat}: S = [true, false, trues...], Hessenberg + triangular"""
    n = 5
    rs = np.random.RandomState(40 + p + 7 * cplx)
    S = [True, False] + [True] * (p - 2)
    mk = (lambda: rs.rand(n, n) + 1j * rs.rand(n, n)) if cplx else (lambda: rs.rand(n, n))
    A = [np.triu(mk()) for _ in range(p)]
    A[0] = np.triu(mk(), -1)
    Ts, Zs, lam, js = ld_gpschur(A, S, "R")
    check_ld(A, S, Ts, Zs, "R", js, tol=200)
    _, ev = mp_eigs(A, S, left=False)
    match_mp(lam, ev, 1e5 * EPS_LD)  # (an inverted triangular factor: the eigenvalue condition enters)


def test_ld_full_complex_and_generalized():
    """test/generalized.jl:201-214 for Complex{BigFloat}: full matrices, all-true signature p = 5 and mixed p = 4"""
    n = 5
    rs = np.random.RandomState(77)
    for S in ([True] * 5, [True, False, True, False]):
        A = [rs.rand(n, n) + 1j * rs.rand(n, n) for _ in S]
        Ts, Zs, lam, js = ld_gpschur(A, S, "R")
        check_ld(A, S, Ts, Zs, "R", js, tol=200)
        _, ev = mp_eigs(A, S, left=False)
        match_mp(lam, ev, 1e6 * EPS_LD)
