"""CPU tier: pins the oracle's signed (generalized) path against the reference's tests (test/generalized.jl)."""
import numpy as np
import pytest

import psdtest as pt


# test/generalized.jl:1-40 "Generalized Periodic Hessenberg": n = 5, p in {2,5}, alternating signature
@pytest.mark.parametrize("cplx", [False, True])
@pytest.mark.parametrize("p", [2, 5])
def test_generalized_phessenberg(built, cplx, p):
    n = 5
    S = [True]
    for l in range(1, p):
        S.append(not S[-1])
    A = pt.rand_uniform_zfactors(n, p, seed=700 + p) if cplx else pt.rand_uniform_factors(n, p, seed=700 + p)
    Hs, Qs = pt.oracle_sg_phessenberg(A, S)
    pt.sg_hess_check(A, S, Hs, Qs)


@pytest.mark.parametrize("cplx", [False, True])
def test_generalized_phessenberg_patterns(built, cplx):
    n, p = 9, 6
    for S in ([True] * 6, [True, True, False, False, True, False], [True, False, False, False, False, False]):
        A = pt.bench_factors(n, p, seed=3, dtype=np.complex128 if cplx else np.float64)
        Hs, Qs = pt.oracle_sg_phessenberg(A, S)
        pt.sg_hess_check(A, S, Hs, Qs)


def _hess_ut(n, p, seed):
    A = [np.triu(a) for a in pt.rand_uniform_factors(n, p, seed)]
    A[0] = np.triu(pt.rand_uniform_factors(n, 1, seed + 77)[0], -1)
    return [np.asfortranarray(a) for a in A]


def _run_hess(A, S, **kw):
    ps = pt.oracle_gpschur_hess(A[0].copy(), [a.copy() for a in A[1:]], S, **kw)
    assert ps.info == 0, ps.info
    pt.rgpschur_check(A, S, ps)
    return ps


# test/generalized.jl:67-76 "Generalized Periodic Schur Hess+UT Float64 (small)"
@pytest.mark.parametrize("p", [2, 3, 5])
def test_rgen_hess_ut_small(built, p):
    for seed in range(6):
        S = [True, False] + [True] * (p - 2)
        _run_hess(_hess_ut(5, p, 800 + 10 * p + seed), S)


# test/generalized.jl:77-152: holes in +/- factors (both signature variants of SINGLE_MINUS_SIG)
HOLES = [
    ("early +hole", [True, True, False, True, True], [True, True, False, True, False], 2, 3),
    ("late +hole", [True, True, False, True, True], [True, True, False, True, False], 4, 3),
    ("late upper -hole", [True, True, True, False, True], [True, False, True, False, True], 4, 2),
    ("late lower -hole", [True, True, True, False, True], [True, False, True, False, True], 4, 4),
    ("upper -hole", [True, False, True, True, True], [True, False, True, False, True], 2, 2),
    ("lower -hole", [True, False, True, True, True], [True, False, True, False, True], 2, 4),
]


@pytest.mark.parametrize("name,S1,S2,l,j", HOLES, ids=[h[0] for h in HOLES])
def test_rgen_holes(built, name, S1, S2, l, j):
    for S in (S1, S2):
        for seed in range(4):
            A = _hess_ut(5, 5, 900 + seed)
            A[l - 1][j - 1, j - 1] = 0.0
            ps = _run_hess(A, S)
            if not S[l - 1]:
                assert np.sum(~np.isfinite(ps.values)) >= 1  # an infinite eigenvalue


# test/generalized.jl:42-65 full matrices, both orientations
@pytest.mark.parametrize("lr", ["R", "L"])
def test_rgen_full(built, lr):
    n, p = 5, 4
    S = [True, False, True, False] if lr == "R" else [False, True, False, True]
    for seed in range(6):
        A = pt.rand_uniform_factors(n, p, 950 + seed)
        ps = pt.oracle_gpschur(A, S, lr)
        assert ps.info == 0 and ps.schurindex == (1 if lr == "R" else p)
        pt.rgpschur_check(A, S, ps)


@pytest.mark.parametrize("n,p", [(12, 3), (24, 6), (40, 4)])
def test_rgen_moderate(built, n, p):
    S = [True] + [bool((l * 7 + n) % 3) for l in range(1, p)]
    A = pt.bench_factors(n, p, seed=n + p)
    ps = pt.oracle_gpschur(A, S, "R")
    assert ps.info == 0
    pt.rgpschur_check(A, S, ps, tol=100 * max(1.0, np.sqrt(n / 32)))
    assert ps.counters["sweeps"] > 0


# all(S): the generalized driver must agree with the standard periodic QR on the spectrum
def test_rgen_all_true_matches_standard(built):
    n, p = 16, 5
    A = pt.bench_factors(n, p, seed=5)
    ps = pt.oracle_gpschur(A, [True] * p, "R")
    assert ps.info == 0
    pt.rgpschur_check(A, [True] * p, ps)
    ref = pt.oracle_pschur(A, "R")
    pt.compare_reigvals(ps.values, ref.values, 1e-10 * max(1.0, np.max(np.abs(ref.values))))


# complex signed full driver (generalized.jl:108-148), test/generalized.jl:188-232
@pytest.mark.parametrize("lr", ["R", "L"])
def test_zgen_full(built, lr):
    n, p = 5, 4
    S = [True, False, True, False] if lr == "R" else [False, True, False, True]
    for seed in range(4):
        A = pt.rand_uniform_zfactors(n, p, 970 + seed)
        ps = pt.oracle_gpschur(A, S, lr)
        assert ps.info == 0
        pt.gpschur_check(A, S, ps)


# test/ordschur.jl:165-222 "gen. ordschur: distinct real" for Float64 and ComplexF64
@pytest.mark.parametrize("cplx", [False, True])
@pytest.mark.parametrize("lr", ["L", "R"])
def test_gordschur_distinct_real(built, cplx, lr):
    p, n, nsel = 5, 7, 2
    S = [True] * p
    for l in range(0, p - 1, 2):
        S[l] = False
    A = pt.gord_test_factors(n, p, S, seed=31, cplx=cplx)
    if lr == "R":  # the same problem seen from the right: reversed sequence and signature
        A, S = A[::-1], S[::-1]
    ps0 = pt.oracle_gpschur(A, S, lr)
    assert ps0.info == 0
    check = pt.gpschur_check if cplx else pt.rgpschur_check
    check(A, S, ps0)
    lam0 = ps0.values
    for rev in (False, True):
        idx = np.argsort(-np.abs(lam0) if rev else np.abs(lam0))
        select = np.zeros(n, dtype=bool)
        select[idx[:nsel]] = True
        ps1 = pt.oracle_gordschur(ps0, select)
        assert ps1.info == 0
        check(A, S, ps1)
        for j in range(nsel):
            assert np.any(np.isclose(ps1.values[:nsel], lam0[idx[j]], rtol=1e-8))
    # all-true signature: the signed swap must agree with the plain one
    St = [True] * p
    At = pt.gord_test_factors(n, p, St, seed=5, cplx=cplx)
    if lr == "R":
        At = At[::-1]
    pg = pt.oracle_gpschur(At, St, lr)
    select = np.zeros(n, dtype=bool)
    select[[2, 5]] = True
    g1 = pt.oracle_gordschur(pg, select)
    assert g1.info == 0 and g1.nswaps > 0
    check(At, St, g1)


# test/ordschur.jl:226-275 "gen. ordschur Float64: conjugate pair(s)" (mkrps with alt = true) + random decompositions
def test_gordschur_pairs_reference(built):
    n, p = 7, 5
    ps0, A = pt.mkrps(n, p, [3, 6], seed=905, alt=True)
    pt.rgpschur_check(A, ps0.S, ps0)
    lam0 = ps0.values
    for selset in ([6, 7], [3, 4], [1, 2, 5], [1, 3, 4], [1, 2, 6, 7], [5]):
        select = np.zeros(n, dtype=bool)
        select[[j - 1 for j in selset]] = True
        ps1 = pt.oracle_gordschur(ps0, select)
        assert ps1.info == 0, ps1.info
        pt.rgpschur_check(A, ps0.S, ps1, tol=400)
        nsel = len(selset)
        for j in selset:
            assert np.any(np.isclose(ps1.values[:nsel], lam0[j - 1], rtol=1e-7)), (selset, ps1.values, lam0)


@pytest.mark.parametrize("lr", ["L", "R"])
def test_gordschur_pairs_random(built, lr):
    for (n, p, seed) in [(10, 3, 1), (14, 4, 2), (16, 5, 3)]:
        S = [bool((q * 7 + n) % 3) for q in range(p)]
        if all(S):
            S[1] = False
        S[p - 1 if lr == "L" else 0] = True
        A = pt.bench_factors(n, p, seed=seed + 40)
        ps0 = pt.oracle_gpschur(A, S, lr)
        assert ps0.info == 0
        lam0 = ps0.values.copy()
        assert np.any(lam0.imag != 0)
        thr = np.sort(np.abs(lam0))[n // 2]
        select = np.abs(lam0) <= thr
        ps1 = pt.oracle_gordschur(ps0, select)
        assert ps1.info == 0 and ps1.nswaps > 0
        pt.rgpschur_check(A, S, ps1, tol=400)
        m = int(select.sum())
        sc = abs(lam0).max()
        assert pt.match_eigs(lam0[select], ps1.values[:m]) < 1e-8 * sc
        assert pt.match_eigs(lam0[~select], ps1.values[m:]) < 1e-8 * sc
