"""CPU tier: pins the oracle's ordschur! restatement (adjacent 1x1 swaps) with the reference's constructed-spectrum
tests, test/ordschur.jl:1-55 (lambda_j = 4^j; select the 2 smallest / 2 largest), for Float64 and ComplexF64,
p in {5, 1}, both orientations."""
import numpy as np
import pytest

import psdtest as pt


def run_case(dtype, p, lr, which):
    n, nsel = 7, 2
    A = pt.ord_test_factors(n, p, seed=4000 + p, dtype=dtype)
    if lr == "R":  # the construction hides the spectrum in the LEFT product A_p...A_1
        A = A[::-1]
    cplx = np.issubdtype(dtype, np.complexfloating)
    ps0 = pt.oracle_zpschur(A, lr) if cplx else pt.oracle_pschur(A, lr)
    assert ps0.info == 0
    lam0 = ps0.values
    expected = np.array([4.0 ** (j + 1) for j in range(n)])
    assert np.allclose(np.sort(np.abs(lam0)), expected, rtol=1e-8)
    idx = np.argsort(np.abs(lam0))
    if which == "largest":
        idx = idx[::-1]
    select = np.zeros(n, dtype=bool)
    select[idx[:nsel]] = True
    ps1 = pt.oracle_ordschur(ps0, select)
    assert ps1.info == 0
    pt.pschur_check(A, ps1, check_lam=False, real=not cplx)
    for j in range(nsel):
        l0 = lam0[idx[j]]
        assert np.any(np.isclose(ps1.values[:nsel], l0, rtol=1e-8))
    # the multiset of eigenvalues is preserved
    assert pt.match_eigs(lam0, ps1.values) < 1e-8 * abs(lam0).max()
    return ps1


@pytest.mark.parametrize("dtype", [np.float64, np.complex128])
@pytest.mark.parametrize("p", [5, 1])
@pytest.mark.parametrize("lr", ["L", "R"])
@pytest.mark.parametrize("which", ["smallest", "largest"])
def test_ordschur_distinct_real(built, dtype, p, lr, which):
    run_case(dtype, p, lr, which)


def test_ordschur_complex_random_select(built):
    n, p = 24, 4
    A = pt.bench_factors(n, p, seed=77, dtype=np.complex128)
    ps0 = pt.oracle_zpschur(A, "R")
    select = np.abs(ps0.values) > np.median(np.abs(ps0.values))
    ps1 = pt.oracle_ordschur(ps0, select)
    assert ps1.info == 0 and ps1.nswaps > 0
    pt.pschur_check(A, ps1, check_lam=False, real=False, tol=64)
    m = select.sum()
    assert pt.match_eigs(ps0.values[select], ps1.values[:m]) < 1e-9 * abs(ps0.values).max()
    assert pt.match_eigs(ps0.values[~select], ps1.values[m:]) < 1e-9 * abs(ps0.values).max()


# test/ordschur.jl:127-164 "conjugate pair(s)": constructed real PSD with 2x2 blocks, literal selections
@pytest.mark.parametrize("p", [5, 1])
@pytest.mark.parametrize("selset", [[1, 2, 5], [1, 3, 4], [1, 2, 6, 7]])
def test_ordschur_conjugate_pairs(built, p, selset):
    n = 7
    ps0, A = pt.mkrps(n, p, [3, 6], seed=900 + p)
    pt.pschur_check(A, ps0, check_lam=False)
    lam0 = ps0.values
    select = np.zeros(n, dtype=bool)
    select[[j - 1 for j in selset]] = True
    ps1 = pt.oracle_ordschur(ps0, select)
    assert ps1.info == 0
    pt.pschur_check(A, ps1, check_lam=False, tol=20000 if p == 1 else 32)
    nsel = len(selset)
    for j in selset:
        assert np.any(np.isclose(ps1.values[:nsel], lam0[j - 1], rtol=1e-8)), (lam0[j - 1], ps1.values[:nsel])
    assert pt.match_eigs(lam0, ps1.values) < 1e-8 * abs(lam0).max()


def test_ordschur_real_random(built):
    """pschur! result with a generic mix of real eigenvalues and conjugate pairs; select by modulus (conjugates closed),
    as BASELINE config 5 does."""
    for (n, p, lr) in [(12, 3, "R"), (16, 4, "L"), (30, 2, "R")]:
        A = pt.bench_factors(n, p, seed=n + p)
        ps0 = pt.oracle_pschur(A, lr)
        lam0 = ps0.values
        thr = np.sort(np.abs(lam0))[n // 2]
        select = np.abs(lam0) <= thr  # the smaller half sits at the bottom after pschur!
        ps1 = pt.oracle_ordschur(ps0, select)
        assert ps1.info == 0 and ps1.nswaps > 0
        pt.pschur_check(A, ps1, check_lam=False, tol=64)
        m = int(select.sum())
        sc = abs(lam0).max()
        assert pt.match_eigs(lam0[select], ps1.values[:m]) < 1e-8 * sc
        assert pt.match_eigs(lam0[~select], ps1.values[m:]) < 1e-8 * sc
