"""CPU tier: pins the oracle (C++ restatement of the reference's real path) against the reference's own
fixtures and property suites and against independent mpmath goldens."""
import numpy as np
import pytest

import psdtest as pt


# test/runtests.jl:14-50 "Periodic Hessenberg"
@pytest.mark.parametrize("p", [1, 2, 5])
def test_phessenberg_reference_suite(built, p):
    n, tol, qtol = 5, 20, 10
    A = pt.rand_uniform_factors(n, p, seed=40 + p)
    Hs, Qs, _, _ = pt.oracle_phessenberg(A)
    assert np.all(np.tril(Hs[0], -2) == 0)
    for j in range(p):
        if j > 0:
            assert np.all(np.tril(Hs[j], -1) == 0)
        assert np.linalg.norm(Qs[j] @ Qs[j].T - np.eye(n)) < qtol * pt.EPS * n
        Ax = Qs[j] @ Hs[j] @ Qs[(j + 1) % p].T
        assert np.linalg.norm(A[j] - Ax) < tol * pt.EPS * n


# test/runtests.jl:53-66 "Periodic Schur Hess+UT"
@pytest.mark.parametrize("p", [1, 2, 3, 5])
def test_hess_ut_inputs(built, p):
    n = 5
    A = [np.asfortranarray(np.triu(a)) for a in pt.rand_uniform_factors(n, p, seed=7 + p)]
    A[0] = np.asfortranarray(np.triu(pt.rand_uniform_factors(n, 1, seed=77 + p)[0], -1))
    pt.pschur_check(A, pt.oracle_pschur(A, "R"))
    if p > 1:
        A[0], A[p - 1] = A[p - 1], A[0]
    pt.pschur_check(A, pt.oracle_pschur(A, "L"))


# test/runtests.jl:68-87 "exp. split" known-answer case (Kressner 2001), literal inputs + asymptotic values
@pytest.mark.parametrize("p", [5, 20])
def test_expsplit_known_answer(built, golden, p):
    A, lam = pt.expsplit(p)
    ps = pt.oracle_pschur(A, "R")
    pt.pschur_check(A, ps, check_lam=False, tol=128)
    for lj in lam:
        d = np.abs(ps.values - lj)
        k = int(np.argmin(d))
        assert d[k] < 1e-3 * abs(lj) or max(abs(lj), abs(ps.values[k])) < pt.EPS ** 2
    # tighter: mpmath eigenvalues of the same literal inputs, relative accuracy per eigenvalue
    for lj in golden[f"expsplit_p{p}_lam"]:
        d = np.abs(ps.values - lj)
        k = int(np.argmin(d))
        assert d[k] < 1e-9 * abs(lj) or max(abs(lj), abs(ps.values[k])) < pt.EPS ** 2
    A[0], A[p - 1] = A[p - 1], A[0]
    pt.pschur_check(A, pt.oracle_pschur(A, "L"), check_lam=False, tol=128)


# test/runtests.jl:89-100 "Periodic Schur full" + goldens
@pytest.mark.parametrize("p", [1, 2, 3, 5])
def test_full_random_small(built, golden, p):
    A = pt.rand_uniform_factors(5, p, seed=500 + p)
    ps = pt.oracle_pschur(A, "R")
    pt.pschur_check(A, ps, lam=golden[f"rand5_p{p}_lam"])
    ps = pt.oracle_pschur(A, "L")
    pt.pschur_check(A, ps, lam=golden[f"rand5_p{p}_lamL"])


# test/runtests.jl:103-132 fast paths
@pytest.mark.parametrize("p", [1, 5])
def test_fast_paths(built, p):
    n, tol = 5, 20
    A = pt.rand_uniform_factors(n, p, seed=9 + p)
    p2 = pt.oracle_pschur(A, wantZ=True)
    p0 = pt.oracle_pschur(A, wantT=False, wantZ=False)
    pt.compare_reigvals(p2.values, p0.values, 1000 * pt.EPS)
    p1 = pt.oracle_pschur(A, wantT=True, wantZ=False)
    assert np.linalg.norm(p1.T1 - p2.T1) < tol * pt.EPS * n
    pt.compare_reigvals(p2.values, p1.values, 1000 * pt.EPS)


# BASELINE config 1 (n=32, p=4, :R, wantZ) and the largest shape in the reference's tests
def test_config1_golden(built, golden):
    As = pt.bench_factors(32, 4, seed=int(golden["cfg1_seed"][0]))
    ps = pt.oracle_pschur(As, "R")
    pt.pschur_check(As, ps, lam=golden["cfg1_lam"])
    ok, err = pt.checkpsd(ps, As)
    assert ok, err
    As = pt.rand_uniform_factors(32, 4, seed=532)
    pt.pschur_check(As, pt.oracle_pschur(As, "R"), lam=golden["rand32_p4_lam"])


def test_rq_cleanup_path(built):
    """A tiny diagonal entry in a triangular factor makes the product sub-diagonal negligible while H1's is
    not: exercises the RQ clean-up pass (PeriodicSchurDecompositions.jl:589-666)."""
    n, p = 8, 3
    A = [np.asfortranarray(np.triu(a) + np.eye(n)) for a in pt.rand_uniform_factors(n, p, seed=321)]
    A[0] = np.asfortranarray(np.triu(pt.rand_uniform_factors(n, 1, seed=322)[0], -1) + np.eye(n))
    A[1][3, 3] = 1e-19
    ps = pt.oracle_pschur(A, "R")
    assert (ps.sweeplog[:, 0] == 1).sum() >= 1
    pt.pschur_check(A, ps, check_lam=False)
    lam = np.linalg.eigvals(pt.product(A))
    assert pt.match_eigs(lam, ps.values) < 1e-12 * abs(lam).max()


def test_n1_and_orientation_argument(built):
    A = [np.asfortranarray(np.array([[2.0]])), np.asfortranarray(np.array([[-3.0]]))]
    ps = pt.oracle_pschur(A, "R")
    assert ps.values[0] == -6.0
    assert pt.oracle_pschur(A, "X").info == -4


def test_medium_bench_ensemble(built):
    n, p = 96, 12
    As = pt.bench_factors(n, p, seed=5)
    ps = pt.oracle_pschur(As, "R")
    ok, err = pt.checkpsd(ps, As, thresh=100 * np.sqrt(n / 32))
    assert ok, err
    lam = np.linalg.eigvals(pt.product(As))
    assert pt.match_eigs(lam, ps.values) < 1e-10 * np.linalg.norm(pt.product(As), 2)


def test_input_ensemble_claim_plain_gaussian_breaks_reference_checkpsd():
    """DESIGN.md section 6 / psdtest.bench_factors: SURVEY.md 8(d) proposed plain N(0, 1/n) factors, but on their
    period-32 products the REFERENCE algorithm (the oracle restates it) fails its own checkpsd — the 2x2
    standardisation of PSD.jl:900-1054 on real pairs of widely split modulus leaves a sub-diagonal that :1066-1073
    zeroes.  This pins the claim: most plain-Gaussian seeds fail by orders of magnitude, the shifted ensemble
    I + 0.5 G / sqrt(n) used by the bench and the parity tests passes with the reference's own threshold scaling."""
    n, p = 128, 32
    thresh = 100 * np.sqrt(n / 32)
    bad = 0
    for seed in (2, 3, 4, 5):  # (measured: seeds 2..5 fail with residuals of 5e5..1e9 eps ||A||, seed 1 passes)
        A = pt.synth_factors(n, p, seed)
        po = pt.oracle_pschur(A, "R")
        ok, err = pt.checkpsd(po, A, thresh=thresh)
        bad += (not ok) and err.max() > 100 * thresh
    assert bad >= 3, bad
    for seed in (1, 2):
        A = pt.bench_factors(n, p, seed)
        po = pt.oracle_pschur(A, "R")
        ok, err = pt.checkpsd(po, A, thresh=thresh)
        assert ok and err.max() < thresh, (seed, err.max())
