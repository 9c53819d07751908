"""CPU tier: pins the complex oracle (generalized.jl:166-976 restatement) against the reference's own complex
tests (test/generalized.jl, test/runtests.jl) — incl. the literal zero-diagonal "hole" cases that force deflation
Cases II/III and infinite eigenvalues."""
import numpy as np
import pytest

import psdtest as pt


def hess_ut(n, p, seed):
    A = [np.asfortranarray(np.triu(a)) for a in pt.rand_uniform_zfactors(n, p, seed)]
    A[0] = np.asfortranarray(np.triu(pt.rand_uniform_zfactors(n, 1, seed + 7)[0], -1))
    return A


def run_hess(A, S):
    ps = pt.oracle_zpschur_hess(A[0].copy(), [a.copy() for a in A[1:]], S)
    assert ps.info == 0
    return ps


# test/runtests.jl:14-50 for ComplexF64
@pytest.mark.parametrize("p", [1, 2, 5])
def test_zphessenberg(built, p):
    n, tol, qtol = 5, 20, 10
    A = pt.rand_uniform_zfactors(n, p, seed=60 + p)
    Hs, Qs, _, _ = pt.oracle_zphessenberg(A)
    assert np.all(np.tril(Hs[0], -2) == 0)
    for j in range(p):
        assert np.linalg.norm(Qs[j] @ Qs[j].conj().T - np.eye(n)) < qtol * pt.EPS * n
        Ax = Qs[j] @ Hs[j] @ Qs[(j + 1) % p].conj().T
        assert np.linalg.norm(A[j] - Ax) < tol * pt.EPS * n


# test/generalized.jl:224-245 "Periodic Schur Hess+UT ComplexF64" (+ hole)
@pytest.mark.parametrize("p", [1, 2, 3, 5])
def test_hess_ut_all_true(built, p):
    A = hess_ut(5, p, 80 + p)
    pt.gpschur_check(A, [True] * p, run_hess(A, [True] * p))
    if p > 1:
        A[1][2, 2] = 0  # A[2][3,3] = 0
        ps = run_hess(A, [True] * p)
        pt.gpschur_check(A, [True] * p, ps)
        assert (ps.sweeplog[:, 0] == 2).any()  # Case II exercised


# test/generalized.jl:68-78 generalized Hess+UT (small)
@pytest.mark.parametrize("p", [2, 3, 5])
def test_generalized_hess_ut(built, p):
    S = [True, False] + [True] * (p - 2)
    A = hess_ut(5, p, 90 + p)
    pt.gpschur_check(A, S, run_hess(A, S))


# test/generalized.jl:79-151: literal hole positions and signatures
@pytest.mark.parametrize("S,fac,idx,case", [
    ([True, True, False, True, False], 2, 3, 2),   # early +hole
    ([True, True, False, True, False], 4, 3, 2),   # late +hole
    ([True, False, True, False, True], 4, 2, 3),   # late upper -hole
    ([True, False, True, False, True], 4, 4, 3),   # late lower -hole
    ([True, False, True, False, True], 2, 2, 3),   # upper -hole
    ([True, False, True, False, True], 2, 4, 3),   # lower -hole
])
def test_holes(built, S, fac, idx, case):
    A = hess_ut(5, 5, 100 + fac * 10 + idx)
    A[fac - 1][idx - 1, idx - 1] = 0
    ps = run_hess(A, S)
    pt.gpschur_check(A, S, ps)
    assert (ps.sweeplog[:, 0] == case).any()
    if case == 3:
        assert (~np.isfinite(ps.values)).sum() >= 1  # singular inverted factor -> infinite eigenvalue


# test/generalized.jl:154-173 moderate N
def test_moderate_n(built):
    n, p = 32, 4
    A = hess_ut(n, p, 333)
    A[1][2, 2] = 0
    pt.gpschur_check(A, [True] * p, run_hess(A, [True] * p))
    S = [True, False, True, False]
    pt.gpschur_check(A, S, run_hess(A, S))


# test/generalized.jl:175-185,201-211 full complex, both orientations
@pytest.mark.parametrize("lr", ["R", "L"])
def test_full_complex(built, lr):
    A = pt.rand_uniform_zfactors(5, 5, seed=71)
    ps = pt.oracle_zpschur(A, lr)
    assert ps.info == 0
    pt.gpschur_check(A, [True] * 5, ps)
    lam = np.linalg.eigvals(pt.product(A, lr == "L"))
    assert pt.match_eigs(lam, ps.values) < 1000 * pt.EPS * abs(lam).max()


# test/generalized.jl:268-303 fast paths (complex)
@pytest.mark.parametrize("p", [1, 5])
def test_fast_paths_complex(built, p):
    n, tol = 5, 20
    A = pt.rand_uniform_zfactors(n, p, seed=120 + p)
    p2 = pt.oracle_zpschur(A, wantZ=True)
    p0 = pt.oracle_zpschur(A, wantT=False, wantZ=False)
    assert np.allclose(p2.values, p0.values, rtol=1e-8)
    p1 = pt.oracle_zpschur(A, wantT=True, wantZ=False)
    assert np.linalg.norm(p1.T1 - p2.T1) < tol * pt.EPS * n * max(1, np.abs(p2.T1).max())
    assert np.allclose(p2.values, p1.values, rtol=1e-8)


# test/runtests.jl:68-87 exp. split for ComplexF64
@pytest.mark.parametrize("p", [5, 20])
def test_expsplit_complex(built, golden, p):
    A, lam = pt.expsplit(p)
    A = [a.astype(np.complex128) for a in A]
    ps = pt.oracle_zpschur(A, "R")
    assert ps.info == 0
    pt.pschur_check(A, ps, check_lam=False, tol=128, real=False)
    for lj in lam:
        d = np.abs(ps.values - lj)
        k = int(np.argmin(d))
        assert d[k] < 1e-3 * abs(lj) or max(abs(lj), abs(ps.values[k])) < pt.EPS ** 2


def test_medium_complex_bench_ensemble(built):
    n, p = 64, 8
    As = pt.bench_factors(n, p, seed=9, dtype=np.complex128)
    ps = pt.oracle_zpschur(As, "R")
    assert ps.info == 0
    ok, err = pt.checkpsd(ps, As, thresh=100 * np.sqrt(n / 32))
    assert ok, err
    P = pt.product(As)
    assert pt.match_eigs(np.linalg.eigvals(P), ps.values) < 1e-10 * np.linalg.norm(P, 2)
    # diagonals of T_2..T_p real and non-negative (generalized.jl:860-908)
    for T in ps.Ts[1:]:
        assert np.all(np.diag(T).imag == 0) and np.all(np.diag(T).real >= 0)
