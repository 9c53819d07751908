"""Parity cases shared by the simulated-device tier (CPU, tests/hostsim) and the GPU tier: every case drives the
engine through the C ABI (via the Python mirror) and judges it with the reference's own checkers against the
oracle / goldens."""
import pytest
import os

import numpy as np

import psdtest as pt


def case_phessenberg(eng, p):
    # test/runtests.jl:14-50
    n, tol, qtol = 5, 20, 10
    A = pt.rand_uniform_factors(n, p, seed=40 + p)
    W = [a.copy(order="F") for a in A]
    Hs, tau, _ = eng.phessenberg_(W)
    Ho, Qo, packed, tauo = pt.oracle_phessenberg(A)
    assert np.all(np.tril(Hs[0], -2) == 0)
    for j in range(p):
        if j > 0:
            assert np.all(np.tril(Hs[j], -1) == 0)
        # same reflector convention as the oracle => same packed storage up to rounding
        assert np.allclose(W[j], packed[j], rtol=0, atol=200 * pt.EPS * n * np.abs(packed[j]).max())
        assert np.allclose(tau[j], tauo[j], rtol=0, atol=100 * pt.EPS)
        Ax = Qo[j] @ Hs[j] @ Qo[(j + 1) % p].T
        assert np.linalg.norm(A[j] - Ax) < tol * pt.EPS * n * max(1.0, np.linalg.norm(A[j], 1))


def case_hess_ut(eng, p):
    n = 5
    A = [np.asfortranarray(np.triu(a)) for a in pt.rand_uniform_factors(n, p, seed=7 + p)]
    A[0] = np.asfortranarray(np.triu(pt.rand_uniform_factors(n, 1, seed=77 + p)[0], -1))
    pt.pschur_check(A, eng.pschur(A, "R"))
    if p > 1:
        A[0], A[p - 1] = A[p - 1], A[0]
    pt.pschur_check(A, eng.pschur(A, "L"))


def case_expsplit(eng, golden, p):
    A, lam = pt.expsplit(p)
    ps = eng.pschur(A, "R")
    pt.pschur_check(A, ps, check_lam=False, tol=128)
    for lj in lam:
        d = np.abs(ps.values - lj)
        k = int(np.argmin(d))
        assert d[k] < 1e-3 * abs(lj) or max(abs(lj), abs(ps.values[k])) < pt.EPS ** 2
    for lj in golden[f"expsplit_p{p}_lam"]:
        d = np.abs(ps.values - lj)
        k = int(np.argmin(d))
        assert d[k] < 1e-9 * abs(lj) or max(abs(lj), abs(ps.values[k])) < pt.EPS ** 2
    A[0], A[p - 1] = A[p - 1], A[0]
    pt.pschur_check(A, eng.pschur(A, "L"), check_lam=False, tol=128)


def case_full_small(eng, golden, p):
    A = pt.rand_uniform_factors(5, p, seed=500 + p)
    pt.pschur_check(A, eng.pschur(A, "R"), lam=golden[f"rand5_p{p}_lam"])
    pt.pschur_check(A, eng.pschur(A, "L"), lam=golden[f"rand5_p{p}_lamL"])


def case_fast_paths(eng, p):
    n, tol = 5, 20
    A = pt.rand_uniform_factors(n, p, seed=9 + p)
    p2 = eng.pschur(A, wantZ=True)
    p0 = eng.pschur(A, wantT=False, wantZ=False)
    assert len(p0.Z) == 0
    pt.compare_reigvals(p2.values, p0.values, 1000 * pt.EPS)
    p1 = eng.pschur(A, wantT=True, wantZ=False)
    assert len(p1.Z) == 0
    assert np.linalg.norm(p1.T1 - p2.T1) < tol * pt.EPS * n
    for j in range(1, p - 1):
        assert np.linalg.norm(p1.T[j] - p2.T[j]) < tol * pt.EPS * n
    pt.compare_reigvals(p2.values, p1.values, 1000 * pt.EPS)


def case_config1(eng, golden):
    As = pt.bench_factors(32, 4, seed=int(golden["cfg1_seed"][0]))
    ps = eng.pschur(As, "R")
    pt.pschur_check(As, ps, lam=golden["cfg1_lam"])
    ok, err = pt.checkpsd(ps, As)
    assert ok, err
    po = pt.oracle_pschur(As, "R")
    assert pt.match_eigs(po.values, ps.values) < 1000 * pt.EPS * abs(po.values).max()
    As = pt.rand_uniform_factors(32, 4, seed=532)
    pt.pschur_check(As, eng.pschur(As, "R"), lam=golden["rand32_p4_lam"])
    pt.pschur_check(As, eng.pschur(As, "L"))


def case_rq_cleanup(eng):
    n, p = 8, 3
    A = [np.asfortranarray(np.triu(a) + np.eye(n)) for a in pt.rand_uniform_factors(n, p, seed=321)]
    A[0] = np.asfortranarray(np.triu(pt.rand_uniform_factors(n, 1, seed=322)[0], -1) + np.eye(n))
    A[1][3, 3] = 1e-19
    ps = eng.pschur(A, "R")
    assert ps.stats.nrqpass >= 1
    pt.pschur_check(A, ps, check_lam=False)
    lam = np.linalg.eigvals(pt.product(A))
    assert pt.match_eigs(lam, ps.values) < 1e-12 * abs(lam).max()
    po = pt.oracle_pschur(A, "R")
    assert pt.match_eigs(po.values, ps.values) < 1e-12 * abs(lam).max()


def case_rq_cleanup_windows(eng):
    """RQ clean-up pass long enough to span several diagonal windows, for every window width."""
    for (n, p) in [(80, 3), (60, 20), (50, 40), (40, 80)]:
        A = [np.asfortranarray(np.triu(a) + 2 * np.eye(n)) for a in pt.bench_factors(n, p, seed=n + p)]
        A[0] = np.asfortranarray(np.triu(pt.bench_factors(n, 1, seed=n + p + 1)[0], -1))
        A[p - 1][4, 4] = 1e-19
        ps = eng.pschur(A, "R")
        assert ps.stats.nrqpass >= 1
        ok, err = pt.checkpsd(ps, A, thresh=100 * np.sqrt(max(n / 32, 1)))
        assert ok, (n, p, err)
        po = pt.oracle_pschur(A, "R")
        scale = abs(po.values).max()
        assert pt.match_eigs(po.values, ps.values) < 1e-10 * scale


def case_edge(eng):
    import psd_amd

    A = [np.asfortranarray(np.array([[2.0]])), np.asfortranarray(np.array([[-3.0]]))]
    ps = eng.pschur(A, "R")
    assert ps.values[0] == -6.0 and ps.Z[0][0, 0] == 1.0
    for n, p in [(2, 1), (2, 3), (3, 2), (4, 1)]:
        A = pt.rand_uniform_factors(n, p, seed=n * 10 + p)
        pt.pschur_check(A, eng.pschur(A, "R"))
        pt.pschur_check(A, eng.pschur(A, "L"))
    try:
        eng.pschur(A, "X")
        raise AssertionError("orientation must be rejected")
    except ValueError:
        pass
    try:
        eng.pschur([np.zeros((3, 3)), np.zeros((4, 4))])
        raise AssertionError("ragged input must be rejected")
    except psd_amd.DimensionMismatch:
        pass
    # the signed case dispatches to the generalized driver (rgeneralized.jl:3-45)
    gs = eng.pschur([np.eye(3) * 2.0, np.eye(3) * 4.0], S=[True, False])
    assert isinstance(gs, psd_amd.GeneralizedPeriodicSchur) and np.allclose(gs.values, 0.5)
    # non-convergence is reported, not hidden: maxitfac=1 cannot converge a 12x12 problem
    A = pt.rand_uniform_factors(12, 3, seed=99)
    try:
        eng.pschur(A, maxitfac=1)
        raise AssertionError("expected ConvergenceError")
    except psd_amd.ConvergenceError as e:
        assert 1 <= e.level <= 12


def expected_window(p, esize=8):
    """The engine's rule (psd_engine.cpp choose_window): the largest W <= 32 whose p window blocks plus the chase
    scratch fit the 160 KiB LDS of one CU; a block is W x (W+1) elements, or W x (W+2) where that makes the column
    pitch even (real standard engine: the LDS-DMA window load wants it); PSD_WINDOW (test hook) lowers the cap."""
    cap = 32
    e = os.environ.get("PSD_WINDOW")
    if e and 6 <= int(e) < cap:
        cap = int(e)
    for W in range(cap, 5, -1):
        pitch = W + 1 if (esize != 8 or (W + 1) % 2 == 0) else W + 2
        need = (p * W * pitch * esize + 64 * 8 + (2 * 64 + p) * 4 + 15) // 16 * 16
        if p <= 64:  # the reflector / rotation table of the scan chase (psd_chase3.h: 8 doubles per factor, psd_zchase3.h: 4)
            need += p * (8 if esize == 8 else 4) * 8
        if (need + 15) // 16 * 16 <= 160 * 1024:
            return W
    return 0


def case_window_widths(eng, sizes):
    """Multi-window sweeps for a range of LDS window widths W (chosen from p; the third entry is the lower bound the
    width had with the coarser rule of the first version)."""
    for (n, p, Wmin) in sizes:
        W = expected_window(p, 8)
        assert W >= Wmin or "PSD_WINDOW" in os.environ
        for lr in "RL":
            As = pt.bench_factors(n, p, seed=n + p)
            ps = eng.pschur(As, lr)
            assert ps.stats.window == W, (p, ps.stats.window, W)
            ok, err = pt.checkpsd(ps, As, thresh=100 * np.sqrt(max(n / 32, 1)))
            assert ok, (n, p, lr, err)
            po = pt.oracle_pschur(As, lr)
            Pn = np.linalg.norm(pt.product(As, lr == "L"), 2)
            assert pt.match_eigs(po.values, ps.values) < 1e-10 * Pn


def case_pschur_hess(eng):
    """pre-reduced entry pschur!(H1, Hs; Q) (PeriodicSchurDecompositions.jl:322-330) incl. rev=true."""
    n, p = 24, 4
    A = pt.bench_factors(n, p, seed=77)
    Hs, Qs, _, _ = pt.oracle_phessenberg(A)
    H = [np.asfortranarray(h.copy()) for h in Hs]
    Q = [np.asfortranarray(q.copy()) for q in Qs]
    ps = eng.pschur_hess_(H[0], H[1:], Q=Q)
    pt.pschur_check(A, ps)
    H = [np.asfortranarray(h.copy()) for h in Hs]
    ps = eng.pschur_hess_(H[0], H[1:], Q=None)  # Z = accumulated transformations only
    pt.pschur_check(Hs, ps)


# ------------------------------------------------------------------------------------------------
# complex path (PeriodicSchurDecompositions.jl:1106-1111 -> generalized.jl:108-137,166-931)
def zhess_ut(n, p, seed):
    A = [np.asfortranarray(np.triu(a)) for a in pt.rand_uniform_zfactors(n, p, seed)]
    A[0] = np.asfortranarray(np.triu(pt.rand_uniform_zfactors(n, 1, seed + 7)[0], -1))
    return A


def case_zphessenberg(eng, p):
    # test/runtests.jl:14-50 for ComplexF64
    n, tol = 5, 20
    A = pt.rand_uniform_zfactors(n, p, seed=60 + p)
    W = [a.copy(order="F") for a in A]
    Hs, tau, _ = eng.phessenberg_(W)
    Ho, Qo, packed, tauo = pt.oracle_zphessenberg(A)
    assert np.all(np.tril(Hs[0], -2) == 0)
    for j in range(p):
        assert np.allclose(W[j], packed[j], rtol=0, atol=200 * pt.EPS * n * np.abs(packed[j]).max())
        assert np.allclose(tau[j], tauo[j], rtol=0, atol=100 * pt.EPS)
        Ax = Qo[j] @ Hs[j] @ Qo[(j + 1) % p].conj().T
        assert np.linalg.norm(A[j] - Ax) < tol * pt.EPS * n * max(1.0, np.linalg.norm(A[j], 1))


def case_zfull(eng, lr):
    # test/generalized.jl:175-185,201-211
    import psd_amd

    A = pt.rand_uniform_zfactors(5, 5, seed=71)
    ps = eng.pschur(A, lr)
    assert isinstance(ps, psd_amd.PeriodicSchur) and not isinstance(ps, psd_amd.GeneralizedPeriodicSchur)
    pt.pschur_check(A, ps, real=False, check_lam=False)
    lam = np.linalg.eigvals(pt.product(A, lr == "L"))
    assert pt.match_eigs(lam, ps.values) < 1000 * pt.EPS * abs(lam).max()
    g = eng.pschur(A, lr, S=[True] * 5)
    assert isinstance(g, psd_amd.GeneralizedPeriodicSchur)
    pt.gpschur_check(A, [True] * 5, g)


def case_zhess_ut(eng, p):
    # test/generalized.jl:224-234 (all-true signature, pre-reduced entry)
    A = zhess_ut(5, p, 80 + p)
    ps = eng.zpschur_hess_(A[0].copy(order="F"), [a.copy(order="F") for a in A[1:]], [True] * p)
    pt.gpschur_check(A, [True] * p, ps)
    po = pt.oracle_zpschur_hess(A[0], A[1:], [True] * p)
    assert pt.match_eigs(po.values, ps.values) < 1000 * pt.EPS * abs(po.values).max()
    # rev=true (generalized.jl:910-927)
    pr = eng.zpschur_hess_(A[0].copy(order="F"), [a.copy(order="F") for a in A[1:]], [True] * p, rev=True)
    assert pr.orientation == "L" and pr.schurindex == p
    Arev = A[1:][::-1] + [A[0]]
    pt.gpschur_check(Arev, [True] * p, pr)


def case_zholes(eng):
    """test/generalized.jl:235-244,154-163: exact zero on the diagonal of a triangular factor (deflation Case II),
    literal positions from the reference's tests plus multi-window variants incl. p >= 20 (zero shift first,
    generalized.jl:199)."""
    cases = [(5, 2, 2, 3), (5, 3, 2, 3), (5, 5, 2, 3), (5, 5, 4, 3), (5, 5, 5, 1), (5, 5, 3, 5), (32, 4, 2, 3),
             (60, 20, 7, 30), (60, 20, 20, 2), (60, 20, 2, 59), (70, 3, 3, 35)]
    for (n, p, fac, idx) in cases:
        A = zhess_ut(n, p, 80 + p + n)
        if n > 32:
            A = [np.asfortranarray(a + 2 * np.eye(n)) if k > 0 else a for k, a in enumerate(A)]
        A[fac - 1][idx - 1, idx - 1] = 0
        ps = eng.zpschur_hess_(A[0].copy(order="F"), [a.copy(order="F") for a in A[1:]], [True] * p)
        pt.gpschur_check(A, [True] * p, ps, tol=100 * max(1, n / 32))
        po = pt.oracle_zpschur_hess(A[0], A[1:], [True] * p)
        assert (po.sweeplog[:, 0] == 2).sum() == ps.stats.ndefl2
        if p < 20:
            assert ps.stats.ndefl2 >= 1
        fin = np.isfinite(po.values)
        assert pt.match_eigs(po.values[fin], ps.values[fin]) < 1e-10 * abs(po.values[fin]).max()


def case_zfast_paths(eng, p):
    # test/generalized.jl:268-303
    n, tol = 5, 20
    A = pt.rand_uniform_zfactors(n, p, seed=120 + p)
    p2 = eng.pschur(A, wantZ=True)
    p0 = eng.pschur(A, wantT=False, wantZ=False)
    assert len(p0.Z) == 0
    assert np.allclose(p2.values, p0.values, rtol=1e-8)
    p1 = eng.pschur(A, wantT=True, wantZ=False)
    assert np.linalg.norm(p1.T1 - p2.T1) < tol * pt.EPS * n * max(1, np.abs(p2.T1).max())
    assert np.allclose(p2.values, p1.values, rtol=1e-8)


def case_zexpsplit(eng, p):
    A, lam = pt.expsplit(p)
    A = [a.astype(np.complex128) for a in A]
    ps = eng.pschur(A, "R")
    pt.pschur_check(A, ps, check_lam=False, tol=128, real=False)
    for lj in lam:
        d = np.abs(ps.values - lj)
        k = int(np.argmin(d))
        assert d[k] < 1e-3 * abs(lj) or max(abs(lj), abs(ps.values[k])) < pt.EPS ** 2


def case_zedge(eng):
    import psd_amd

    # negative signatures dispatch to the signed engine (generalized.jl:138-146)
    A = zhess_ut(5, 3, 91)
    pt.gpschur_check(A, [True, False, True],
                     eng.zpschur_hess_(A[0].copy(order="F"), [a.copy(order="F") for a in A[1:]], [True, False, True]))
    A2 = pt.rand_uniform_zfactors(4, 2, 5)
    pt.gpschur_check(A2, [True, False], eng.pschur(A2, S=[True, False]))
    try:
        eng.pschur(pt.rand_uniform_zfactors(4, 2, 5), S=[False, True])
        raise AssertionError("leftmost S must be true")
    except ValueError:
        pass
    for n, p in [(1, 3), (2, 2), (3, 1)]:
        A = pt.rand_uniform_zfactors(n, p, seed=n * 7 + p)
        ps = eng.pschur(A, "R")
        lam = np.linalg.eigvals(pt.product(A))
        assert pt.match_eigs(lam, ps.values) < 1000 * pt.EPS * abs(lam).max()
        pt.pschur_check(A, ps, real=False, check_lam=False)
    A = pt.rand_uniform_zfactors(12, 3, seed=99)
    try:
        eng.pschur(A, maxitfac=1)
        raise AssertionError("expected ConvergenceError")
    except psd_amd.ConvergenceError:
        pass


def case_zwindow_widths(eng, sizes):
    for (n, p, Wmin) in sizes:
        W = expected_window(p, 16)
        assert W >= Wmin or "PSD_WINDOW" in os.environ
        for lr in "RL":
            As = pt.bench_factors(n, p, seed=n + p, dtype=np.complex128)
            ps = eng.pschur(As, lr)
            assert ps.stats.window == W, (p, ps.stats.window)
            ok, err = pt.checkpsd(ps, As, thresh=100 * np.sqrt(max(n / 32, 1)))
            assert ok, (n, p, lr, err)
            po = pt.oracle_zpschur(As, lr)
            Pn = np.linalg.norm(pt.product(As, lr == "L"), 2)
            assert pt.match_eigs(po.values, ps.values) < 1e-10 * Pn
            if p >= 20:
                assert ps.stats.nrqpass >= 1  # controlled zero shift (generalized.jl:199,356)
            for T in ps.Ts[:ps.schurindex - 1] + ps.Ts[ps.schurindex:]:
                assert np.all(np.diag(T).imag == 0) and np.all(np.diag(T).real >= 0)


# ------------------------------------------------------------------------------------------------
# ordschur! (complex): ordschur.jl:11-73, sylswap.jl:542-635
def _clone(ps):
    import psd_amd

    return psd_amd.PeriodicSchur([t.copy(order="F") for t in ps.Ts], [z.copy(order="F") for z in ps.Z],
                                 ps.values.copy(), ps.orientation, ps.schurindex)


def case_zordschur_reference(eng, p, lr):
    """test/ordschur.jl:1-55 for ComplexF64: constructed spectrum 4^j, select the 2 smallest / 2 largest."""
    n, nsel = 7, 2
    A = pt.ord_test_factors(n, p, seed=4000 + p, dtype=np.complex128)
    if lr == "R":
        A = A[::-1]
    ps0 = eng.pschur(A, lr)
    lam0 = ps0.values.copy()
    assert np.allclose(np.sort(np.abs(lam0)), [4.0 ** (j + 1) for j in range(n)], rtol=1e-8)
    for which in ("smallest", "largest"):
        idx = np.argsort(np.abs(lam0))
        if which == "largest":
            idx = idx[::-1]
        select = np.zeros(n, dtype=bool)
        select[idx[:nsel]] = True
        ps1 = eng.ordschur_(_clone(ps0), select)
        pt.pschur_check(A, ps1, check_lam=False, real=False)
        for j in range(nsel):
            assert np.any(np.isclose(ps1.values[:nsel], lam0[idx[j]], rtol=1e-8))
        po = pt.oracle_ordschur(pt.PSD(ps0.Ts, ps0.Z, lam0, ps0.orientation, ps0.schurindex), select)
        assert ps1.stats.nsweeps == po.nswaps
        assert np.allclose(ps1.values, po.values, rtol=1e-10, atol=0)


def case_zordschur_windows(eng, sizes):
    """Random selections that need many swaps spanning several windows, every window width, both orientations,
    with and without Z."""
    for (n, p) in sizes:
        for lr in "RL":
            A = pt.bench_factors(n, p, seed=70 + n + p, dtype=np.complex128)
            ps0 = eng.pschur(A, lr)
            lam0 = ps0.values.copy()
            rng = pt.randn_counter(n + p, 5, n)
            select = rng > 0.3
            select[-1] = True
            select[0] = False
            ps1 = eng.ordschur_(_clone(ps0), select)
            assert ps1.stats.nsweeps > 0
            ok, err = pt.checkpsd(ps1, A, thresh=100 * np.sqrt(max(n / 32, 1)))
            assert ok, (n, p, lr, err)
            m = int(select.sum())
            sc = abs(lam0).max()
            assert pt.match_eigs(lam0[select], ps1.values[:m]) < 1e-9 * sc
            assert pt.match_eigs(lam0[~select], ps1.values[m:]) < 1e-9 * sc
            # order inside the selected / unselected groups is preserved (stable bubbling)
            assert np.allclose(ps1.values[:m], lam0[select], rtol=1e-7, atol=1e-9 * sc)
            assert np.allclose(ps1.values[m:], lam0[~select], rtol=1e-7, atol=1e-9 * sc)
            po = pt.oracle_ordschur(pt.PSD(ps0.Ts, ps0.Z, lam0, ps0.orientation, ps0.schurindex), select)
            assert ps1.stats.nsweeps == po.nswaps
            # wantZ = false leaves Z untouched and gives the same T
            ps2 = eng.ordschur_(_clone(ps0), select, wantZ=False)
            assert all(np.array_equal(a, b) for a, b in zip(ps2.Z, ps0.Z))
            assert max(np.abs(a - b).max() for a, b in zip(ps2.Ts, ps1.Ts)) < 1e-12 * sc


def case_zordschur_edge(eng):
    import psd_amd

    A = pt.bench_factors(6, 3, seed=3, dtype=np.complex128)
    ps0 = eng.pschur(A, "R")
    same = eng.ordschur_(_clone(ps0), np.zeros(6, dtype=bool))  # nothing selected: no-op
    assert same.stats.nsweeps == 0 and all(np.array_equal(a, b) for a, b in zip(same.Ts, ps0.Ts))
    allsel = eng.ordschur_(_clone(ps0), np.ones(6, dtype=bool))
    assert allsel.stats.nsweeps == 0
    try:
        eng.ordschur_(_clone(ps0), np.ones(5, dtype=bool))
        raise AssertionError("wrong select length must be rejected")
    except psd_amd.DimensionMismatch:
        pass
    # two equal eigenvalues cannot be swapped: the periodic Sylvester system is singular / the swap is rejected
    n, p = 4, 2
    T = [np.asfortranarray(np.triu(np.ones((n, n))).astype(np.complex128)) for _ in range(p)]
    Z = [np.asfortranarray(np.eye(n, dtype=np.complex128)) for _ in range(p)]
    P = psd_amd.PeriodicSchur(T, Z, np.ones(n, dtype=complex), "R", 1)
    try:
        eng.ordschur_(P, np.array([False, True, False, False]))
        raise AssertionError("expected SingularException / IllConditionedException")
    except (psd_amd.SingularException, psd_amd.IllConditionedException):
        pass


# ------------------------------------------------------------------------------------------------
# ordschur! (real, 1x1 and 2x2 blocks): rordschur.jl:3-132, sylswap.jl:14-191
def case_rordschur_reference_real(eng, p, lr):
    """test/ordschur.jl:1-55 for Float64 (distinct real eigenvalues 4^j)."""
    n, nsel = 7, 2
    A = pt.ord_test_factors(n, p, seed=4000 + p)
    if lr == "R":
        A = A[::-1]
    ps0 = eng.pschur(A, lr)
    lam0 = ps0.values.copy()
    for which in ("smallest", "largest"):
        idx = np.argsort(np.abs(lam0))
        if which == "largest":
            idx = idx[::-1]
        select = np.zeros(n, dtype=bool)
        select[idx[:nsel]] = True
        ps1 = eng.ordschur_(_clone(ps0), select)
        pt.pschur_check(A, ps1, check_lam=False)
        for j in range(nsel):
            assert np.any(np.isclose(ps1.values[:nsel], lam0[idx[j]], rtol=1e-8))


def case_rordschur_pairs(eng, p, selset):
    """test/ordschur.jl:127-164: constructed real PSD (mkrps) with conjugate pairs at rows 3:4 and 6:7."""
    import psd_amd

    n = 7
    ps0, A = pt.mkrps(n, p, [3, 6], seed=900 + p)
    P0 = psd_amd.PeriodicSchur([t.copy(order="F") for t in ps0.Ts], [z.copy(order="F") for z in ps0.Z],
                               ps0.values.copy(), "L", p)
    select = np.zeros(n, dtype=bool)
    select[[j - 1 for j in selset]] = True
    ps1 = eng.ordschur_(P0, select)
    pt.pschur_check(A, ps1, check_lam=False, tol=20000 if p == 1 else 32)
    nsel = len(selset)
    for j in selset:
        assert np.any(np.isclose(ps1.values[:nsel], ps0.values[j - 1], rtol=1e-8))
    po = pt.oracle_ordschur(ps0, select)
    assert ps1.stats.nsweeps == po.nswaps
    assert pt.match_eigs(po.values, ps1.values) < 1e-10 * abs(po.values).max()


def case_rordschur_windows(eng, sizes):
    """pschur! results with a generic mix of real eigenvalues and conjugate pairs; the smaller half is selected
    (conjugates closed, as BASELINE config 5 selects by modulus): many swaps of all block-size combinations over
    several windows, every window width, both orientations, with and without Z."""
    for (n, p) in sizes:
        for lr in "RL":
            A = pt.bench_factors(n, p, seed=90 + n + p)
            ps0 = eng.pschur(A, lr)
            lam0 = ps0.values.copy()
            thr = np.sort(np.abs(lam0))[n // 2]
            select = np.abs(lam0) <= thr
            ps1 = eng.ordschur_(_clone(ps0), select)
            assert ps1.stats.nsweeps > 0
            ok, err = pt.checkpsd(ps1, A, thresh=100 * np.sqrt(max(n / 32, 1)))
            assert ok, (n, p, lr, err)
            m = int(select.sum())
            sc = abs(lam0).max()
            assert pt.match_eigs(lam0[select], ps1.values[:m]) < 1e-8 * sc
            assert pt.match_eigs(lam0[~select], ps1.values[m:]) < 1e-8 * sc
            for i in range(n - 1):  # structure: sub-diagonal entries only inside conjugate pairs
                if ps1.values[i].imag == 0 or ps1.values[i].imag < 0:
                    assert ps1.T1[i + 1, i] == 0
            po = pt.oracle_ordschur(pt.PSD(ps0.Ts, ps0.Z, lam0, ps0.orientation, ps0.schurindex), select)
            assert po.info == 0 and ps1.stats.nsweeps == po.nswaps
            ps2 = eng.ordschur_(_clone(ps0), select, wantZ=False)
            assert all(np.array_equal(a, b) for a, b in zip(ps2.Z, ps0.Z))
            assert max(np.abs(a - b).max() for a, b in zip(ps2.Ts, ps1.Ts)) < 1e-12 * max(1.0, sc)


def case_rordschur_pipelined(make_engine, sizes):
    """The pipelined driver (psd_rord_plan / psd_rord_step_mb: several selected blocks under way at once, a window apart)
    against the serial one (PSD_ORD_PIPE=0: rordschur.jl:77-110 block by block) and the oracle: every block makes the
    serial order's swaps, so the swap counts agree, the eigenvalues keep their order inside both groups, and the
    invariants hold.  `make_engine(env)` builds an engine under the given environment."""
    e_pipe = make_engine({"PSD_ORD_PIPE": "1"})
    e_ser = make_engine({"PSD_ORD_PIPE": "0"})
    for (n, p, lr, frac) in sizes:
        A = pt.bench_factors(n, p, seed=300 + n + p)
        ps0 = e_pipe.pschur(A, lr)
        lam0 = ps0.values.copy()
        thr = np.sort(np.abs(lam0))[int(n * frac)]
        select = np.abs(lam0) <= thr
        ps1 = e_pipe.ordschur_(_clone(ps0), select)
        ps2 = e_ser.ordschur_(_clone(ps0), select)
        assert ps1.stats.nsweeps == ps2.stats.nsweeps > 0, (n, p, ps1.stats.nsweeps, ps2.stats.nsweeps)
        assert ps1.stats.nlaunch_step < ps2.stats.nlaunch_step or ps2.stats.nlaunch_step <= 32
        ok, err = pt.checkpsd(ps1, A, thresh=100 * np.sqrt(max(n / 32, 1)))
        assert ok, (n, p, lr, err)
        m = int(select.sum())
        sc = abs(lam0).max()
        assert pt.match_eigs(lam0[select], ps1.values[:m]) < 1e-8 * sc
        assert pt.match_eigs(lam0[~select], ps1.values[m:]) < 1e-8 * sc
        assert pt.match_eigs(ps2.values, ps1.values) < 1e-9 * sc
        assert np.allclose(ps1.values, ps2.values, rtol=1e-6, atol=1e-8 * sc)  # same order, block by block
        for i in range(n - 1):
            if ps1.values[i].imag == 0 or ps1.values[i].imag < 0:
                assert ps1.T1[i + 1, i] == 0
        po = pt.oracle_ordschur(pt.PSD(ps0.Ts, ps0.Z, lam0, ps0.orientation, ps0.schurindex), select)
        assert po.info == 0 and ps1.stats.nsweeps == po.nswaps


def case_zordschur_pipelined(make_engine, sizes, gsizes):
    """Pipelined ordschur! drivers of the ComplexF64 engines (psd_ord1_plan + psd_zord_step_mb / psd_zgord_step_mb:
    selected eigenvalues under way together, a window apart) against the serial ones (PSD_ORD_PIPE=0: ordschur.jl:53-65
    eigenvalue by eigenvalue): same swap counts (and the oracle's), same order, invariants.  gsizes: signed problems."""
    e_pipe = make_engine({"PSD_ORD_PIPE": "1"})
    e_ser = make_engine({"PSD_ORD_PIPE": "0"})
    for (n, p, lr, frac) in sizes:
        A = pt.bench_factors(n, p, seed=500 + n + p, dtype=np.complex128)
        ps0 = e_pipe.pschur(A, lr)
        lam0 = ps0.values.copy()
        thr = np.sort(np.abs(lam0))[int(n * frac)]
        select = np.abs(lam0) <= thr
        ps1 = e_pipe.ordschur_(_clone(ps0), select)
        ps2 = e_ser.ordschur_(_clone(ps0), select)
        assert ps1.stats.nsweeps == ps2.stats.nsweeps > 0, (n, p, ps1.stats.nsweeps, ps2.stats.nsweeps)
        assert ps1.stats.nlaunch_step < ps2.stats.nlaunch_step or ps2.stats.nlaunch_step <= 32
        ok, err = pt.checkpsd(ps1, A, thresh=100 * np.sqrt(max(n / 32, 1)))
        assert ok, (n, p, lr, err)
        m = int(select.sum())
        sc = abs(lam0).max()
        assert np.allclose(ps1.values[:m], lam0[select], rtol=1e-7, atol=1e-9 * sc)
        assert np.allclose(ps1.values[m:], lam0[~select], rtol=1e-7, atol=1e-9 * sc)
        assert np.allclose(ps1.values, ps2.values, rtol=1e-7, atol=1e-9 * sc)
        po = pt.oracle_ordschur(pt.PSD(ps0.Ts, ps0.Z, lam0, ps0.orientation, ps0.schurindex), select)
        assert ps1.stats.nsweeps == po.nswaps
    for (n, p, lr) in gsizes:
        S = [bool((q * 7 + n) % 3) for q in range(p)]
        if all(S):
            S[p // 2] = False
        S[p - 1 if lr == "R" else 0] = True
        A = pt.gord_test_factors(n, p, S, seed=n + p, cplx=True)
        if lr == "R":
            A, S = A[::-1], S[::-1]
        res = []
        for e in (e_pipe, e_ser):
            ps0 = e.pschur_([a.copy(order="F") for a in A], lr, S=S)
            lam0 = ps0.values.copy()
            order = np.argsort(np.abs(lam0))
            select = np.zeros(n, dtype=bool)
            select[order[: n // 2]] = True
            ps1 = e.ordschur_(ps0, select)
            pt.gpschur_check(A, S, ps1, tol=100 * max(1.0, np.sqrt(n / 32)), qtol=10 * max(1.0, np.sqrt(n / 32)))
            m = int(select.sum())
            assert pt.match_eigs(lam0[select], ps1.values[:m]) < 1e-7 * abs(lam0).max()
            res.append(ps1)
        assert res[0].stats.nsweeps == res[1].stats.nsweeps > 0
        assert res[0].stats.nlaunch_step < res[1].stats.nlaunch_step or res[1].stats.nlaunch_step <= 32
        fin = np.isfinite(res[1].values)
        assert np.allclose(res[0].values[fin], res[1].values[fin], rtol=1e-6, atol=1e-8 * abs(res[1].values[fin]).max())


def case_ordschur_pipelined_failure(make_engine):
    """A swap that cannot be done (two equal eigenvalues, rows 40 and 45 of 60) while earlier selected eigenvalues are
    still travelling: the pipelined driver stops the later blocks, lets the earlier ones arrive and raises what the
    serial driver raises (ComplexF64: SingularException from the cyclic Sylvester system)."""
    import psd_amd

    n, p = 60, 2
    rng = np.random.default_rng(5)
    T = [np.asfortranarray(np.triu(rng.standard_normal((n, n)))).astype(np.complex128) for _ in range(p)]
    for j in range(p):
        T[j][np.diag_indices(n)] = 1.0 + np.arange(n) * 0.1 * (j + 1)
        T[j][44, 44] = T[j][39, 39]
    Z = [np.asfortranarray(np.eye(n, dtype=np.complex128)) for _ in range(p)]
    lam = np.prod([np.diag(t) for t in T], axis=0).astype(complex)
    sel = np.zeros(n, dtype=bool)
    sel[[9, 19, 44, 50]] = True
    raised = []
    for pipe in ("1", "0"):
        eng = make_engine({"PSD_ORD_PIPE": pipe})
        P = psd_amd.PeriodicSchur([np.asfortranarray(t.copy()) for t in T], [np.asfortranarray(z.copy()) for z in Z], lam.copy(), "R", 1)
        try:
            eng.ordschur_(P, sel)
            raised.append(None)
        except (psd_amd.SingularException, psd_amd.IllConditionedException) as ex:
            raised.append(type(ex).__name__)
    # (the serial simulation keeps the two eigenvalues exactly equal until they meet and raises; on the device the earlier
    #  swaps leave them a rounding error apart and the swap goes through — with either driver)
    assert raised[0] == raised[1], raised
    return raised[0]


def case_rordschur_edge(eng):
    import psd_amd

    A = pt.bench_factors(8, 3, seed=5)
    ps0 = eng.pschur(A, "R")
    n = 8
    assert eng.ordschur_(_clone(ps0), np.zeros(n, dtype=bool)).stats.nsweeps == 0
    assert eng.ordschur_(_clone(ps0), np.ones(n, dtype=bool)).stats.nsweeps == 0
    cp = np.where(ps0.values.imag > 0)[0]
    if len(cp):  # selecting one member of a conjugate pair takes the partner along (rordschur.jl:46-75)
        j = int(cp[-1])
        sel = np.zeros(n, dtype=bool)
        sel[j + 1] = True
        ps1 = eng.ordschur_(_clone(ps0), sel)
        pt.pschur_check(A, ps1, check_lam=False)
        assert np.isclose(ps1.values[0], ps0.values[j]) and np.isclose(ps1.values[1], ps0.values[j + 1])
    n, p = 4, 2  # equal eigenvalues: singular periodic Sylvester system / rejected swap
    T = [np.asfortranarray(np.triu(np.ones((n, n)))) for _ in range(p)]
    Z = [np.asfortranarray(np.eye(n)) for _ in range(p)]
    P = psd_amd.PeriodicSchur(T, Z, np.ones(n, dtype=complex), "R", 1)
    try:
        eng.ordschur_(P, np.array([False, True, False, False]))
        raise AssertionError("expected SingularException / IllConditionedException")
    except (psd_amd.SingularException, psd_amd.IllConditionedException):
        pass


# ---- real generalized (signed) periodic QZ: rgeneralized.jl / test/generalized.jl Float64 parts ----------------
def rg_hess_ut(n, p, seed, shift=0.0):
    A = [np.triu(a) for a in pt.rand_uniform_factors(n, p, seed)]
    A[0] = np.triu(pt.rand_uniform_factors(n, 1, seed + 77)[0], -1)
    if shift:
        A = [a if k == 0 else a + shift * np.eye(n) for k, a in enumerate(A)]
    return [np.asfortranarray(a) for a in A]


def _rg_run(eng, A, S, tol=100, **kw):
    ps = eng.gpschur_hess_(A[0].copy(order="F"), [a.copy(order="F") for a in A[1:]], S, **kw)
    pt.rgpschur_check(A, S, ps, tol=tol)
    return ps


def _rg_match_oracle(A, S, ps, rtol=1e-9):
    po = pt.oracle_gpschur_hess(A[0], A[1:], S)
    assert po.info == 0
    fin = np.isfinite(po.values)
    assert fin.sum() == np.isfinite(ps.values).sum()
    if fin.any():
        scale = abs(po.values[fin]).max()
        assert pt.match_eigs(po.values[fin], ps.values[np.isfinite(ps.values)]) < rtol * max(scale, 1.0)
    return po


def case_rg_hess_ut(eng, p):
    # test/generalized.jl:67-76
    S = [True, False] + [True] * (p - 2)
    for seed in range(3):
        A = rg_hess_ut(5, p, 800 + 10 * p + seed)
        ps = _rg_run(eng, A, S)
        _rg_match_oracle(A, S, ps)
    # rev = true (rgeneralized.jl:1062-1079)
    A = rg_hess_ut(5, p, 831 + p)
    pr = eng.gpschur_hess_(A[0].copy(order="F"), [a.copy(order="F") for a in A[1:]], S, rev=True)
    assert pr.orientation == "L" and pr.schurindex == p and pr.S == S[::-1]
    Arev = A[1:][::-1] + [A[0]]
    pt.rgpschur_check(Arev, S[::-1], pr)


RG_HOLES = [
    ([True, True, False, True, True], 2, 3), ([True, True, False, True, False], 2, 3),
    ([True, True, False, True, True], 4, 3), ([True, True, False, True, False], 4, 3),
    ([True, True, True, False, True], 4, 2), ([True, False, True, False, True], 4, 2),
    ([True, True, True, False, True], 4, 4), ([True, False, True, False, True], 4, 4),
    ([True, False, True, True, True], 2, 2), ([True, False, True, False, True], 2, 2),
    ([True, False, True, True, True], 2, 4), ([True, False, True, False, True], 2, 4),
]


def case_rg_holes(eng):
    """test/generalized.jl:77-152: exact zeros on the diagonal of positive (Case II) and negative (Case III) factors."""
    for (S, l, j) in RG_HOLES:
        for seed in range(2):
            A = rg_hess_ut(5, 5, 900 + seed)
            A[l - 1][j - 1, j - 1] = 0.0
            ps = _rg_run(eng, A, S)
            po = _rg_match_oracle(A, S, ps)
            c2, c3 = ps.stats.reserved % 1000, ps.stats.reserved // 1000
            assert (c2, c3) == (po.counters["case2"], po.counters["case3"])
            assert c2 + c3 >= 1
    # larger: holes inside multi-window problems, both halves of Case III, p >= 20 (zero shift first)
    for (n, p, l, j) in [(40, 4, 2, 30), (40, 4, 3, 8), (40, 4, 4, 20), (30, 21, 5, 11), (30, 21, 12, 25)]:
        S = [True] + [bool((q * 5 + 1) % 3) for q in range(1, p)]
        A = rg_hess_ut(n, p, 40 + n + p, shift=2.0)
        A[l - 1][j - 1, j - 1] = 0.0
        ps = _rg_run(eng, A, S, tol=100 * max(1, n / 32))
        _rg_match_oracle(A, S, ps, rtol=1e-8)


def case_rg_windows(eng, sizes):
    """multi-window sweeps, zero-shift passes, 2x2 blocks of both kinds, every signature pattern class"""
    for (n, p, pat) in sizes:
        if pat == "alt":
            S = [True] + [bool(q % 2 == 0) for q in range(1, p)]
        elif pat == "true":
            S = [True] * p
        elif pat == "neg":
            S = [True] + [False] * (p - 1)
        else:
            S = [True] + [bool((q * 7 + n) % 3) for q in range(1, p)]
        A = rg_hess_ut(n, p, 300 + n + p, shift=1.0)
        ps = _rg_run(eng, A, S, tol=100 * max(1, np.sqrt(n / 32)))
        _rg_match_oracle(A, S, ps, rtol=1e-8)
        assert ps.stats.nsweeps > 0


def case_rg_fast_paths(eng):
    # wantT / wantZ variants (rgeneralized.jl:636-641, 658-660, 784-789)
    n, p = 24, 4
    S = [True, False, True, False]
    A = rg_hess_ut(n, p, 77, shift=1.0)
    full = _rg_run(eng, A, S)
    noz = eng.gpschur_hess_(A[0].copy(order="F"), [a.copy(order="F") for a in A[1:]], S, wantZ=False)
    assert noz.Z == []
    assert pt.match_eigs(full.values, noz.values) < 1e-10 * abs(full.values).max()
    for l in range(p):
        assert np.linalg.norm(noz.Ts[l] - full.Ts[l]) < 1e-10 * np.linalg.norm(full.Ts[l])
    fast = eng.gpschur_hess_(A[0].copy(order="F"), [a.copy(order="F") for a in A[1:]], S, wantZ=False, wantT=False)
    assert pt.match_eigs(full.values, fast.values) < 1e-9 * abs(full.values).max()


def case_rg_edge(eng):
    # n = 1, n = 2 (real pair / complex pair), n = 3, p = 1
    for (n, p) in [(1, 3), (2, 3), (2, 4), (3, 2), (4, 1), (7, 1)]:
        for seed in range(3):
            S = [True] + [bool((q + seed) % 2) for q in range(1, p)]
            A = rg_hess_ut(n, p, 500 + 10 * n + p + seed)
            ps = _rg_run(eng, A, S)
            _rg_match_oracle(A, S, ps)
    with pytest.raises(ValueError):
        A = rg_hess_ut(4, 2, 1)
        eng.gpschur_hess_(A[0], A[1:], [False, True])


def case_rg_phessenberg(eng, p):
    # test/generalized.jl:1-40 "Generalized Periodic Hessenberg" (Float64): n = 5, alternating signature
    n = 5
    S = [True]
    for _ in range(1, p):
        S.append(not S[-1])
    A = pt.rand_uniform_factors(n, p, seed=700 + p)
    Hs, Qs = eng.gphessenberg_([a.copy(order="F") for a in A], S)
    pt.sg_hess_check(A, S, Hs, Qs)
    # every signature class, multi-window stage 2
    for (n2, p2, S2) in [(9, 6, [True] * 6), (9, 6, [True, True, False, False, True, False]),
                         (33, 4, [True, False, False, False]), (70, 3, [True, False, True]), (45, 22, None)]:
        if S2 is None:
            S2 = [True] + [bool((q * 5) % 3) for q in range(1, p2)]
        A = pt.bench_factors(n2, p2, seed=3 + n2)
        Hs, Qs = eng.gphessenberg_([a.copy(order="F") for a in A], S2)
        pt.sg_hess_check(A, S2, Hs, Qs, tol=20 * max(1, n2 / 8), qtol=10 * max(1, n2 / 16))


def case_rg_full(eng, lr):
    # test/generalized.jl:42-65
    n, p = 5, 4
    S = [True, False, True, False] if lr == "R" else [False, True, False, True]
    for seed in range(4):
        A = pt.rand_uniform_factors(n, p, 950 + seed)
        ps = eng.pschur_([a.copy(order="F") for a in A], lr, S=S)
        assert ps.schurindex == (1 if lr == "R" else p) and ps.orientation == lr
        pt.rgpschur_check(A, S, ps)
        po = pt.oracle_gpschur(A, S, lr)
        assert pt.match_eigs(po.values, ps.values) < 1e-9 * max(1.0, abs(po.values).max())
    with pytest.raises(ValueError):
        eng.pschur_([a.copy(order="F") for a in A], lr, S=S[::-1])


def case_rg_full_sizes(eng, sizes):
    for (n, p, lr, pat) in sizes:
        if pat == "true":
            S = [True] * p
        else:
            S = [bool((q * 7 + n) % 3) for q in range(p)]
            S[p - 1 if lr == "L" else 0] = True
        A = pt.bench_factors(n, p, seed=n + p)
        ps = eng.pschur_([a.copy(order="F") for a in A], lr, S=S)
        pt.rgpschur_check(A, S, ps, tol=100 * max(1.0, np.sqrt(n / 32)), qtol=10 * max(1.0, np.sqrt(n / 32)))
        po = pt.oracle_gpschur(A, S, lr)
        assert pt.match_eigs(po.values, ps.values) < 1e-8 * max(1.0, abs(po.values).max())


# real signed path: multishift trains with explicit shifts (psd_set_train_g) against the reference's iteration
def case_rg_trains(eng, sizes):
    m0 = eng.get_train_g()
    try:
        for (n, p, lr, pat) in sizes:
            if pat == "true":
                S = [True] * p
            else:
                S = [bool((q * 7 + n) % 3) for q in range(p)]
                S[p - 1 if lr == "L" else 0] = True
            A = pt.bench_factors(n, p, seed=500 + n + p)
            eng.set_train_g(0)
            ref = eng.pschur_([a.copy(order="F") for a in A], lr, S=S)
            assert ref.stats.maxits == 0
            fin = np.isfinite(ref.values)
            scale = max(1.0, abs(ref.values[fin]).max())
            for m in (-2, 2, 6):  # -2: explicit-shift start without a train
                eng.set_train_g(m)
                ps = eng.pschur_([a.copy(order="F") for a in A], lr, S=S)
                if m >= 2:
                    assert ps.stats.maxits > 0, (n, p, lr, m, "no train ran")
                pt.rgpschur_check(A, S, ps, tol=100 * max(1.0, np.sqrt(n / 32)), qtol=10 * max(1.0, np.sqrt(n / 32)))
                assert pt.match_eigs(ref.values[fin], ps.values[np.isfinite(ps.values)]) < 1e-8 * scale, (n, p, lr, m)
                p0 = eng.pschur_([a.copy(order="F") for a in A], lr, S=S, wantT=False, wantZ=False)
                assert pt.match_eigs(ref.values[fin], p0.values[np.isfinite(p0.values)]) < 1e-8 * scale, (n, p, lr, m)
    finally:
        eng.set_train_g(m0)


# complex signed path: multishift trains (width shared with the real signed path, psd_set_train_g)
def case_zg_trains(eng, sizes):
    m0 = eng.get_train_g()
    try:
        for (n, p, lr) in sizes:
            S = [bool((q * 7 + n) % 3) for q in range(p)]
            S[p - 1 if lr == "L" else 0] = True
            A = pt.bench_factors(n, p, seed=600 + n + p, dtype=np.complex128)
            eng.set_train_g(0)
            ref = eng.pschur_([a.copy(order="F") for a in A], lr, S=S)
            assert ref.stats.maxits == 0
            fin = np.isfinite(ref.values)
            scale = max(1.0, abs(ref.values[fin]).max())
            for m in (-2, 2, 6):  # -2: explicit-shift start without a train
                eng.set_train_g(m)
                ps = eng.pschur_([a.copy(order="F") for a in A], lr, S=S)
                if m >= 2:
                    assert ps.stats.maxits > 0, (n, p, lr, m, "no train ran")
                pt.gpschur_check(A, S, ps, tol=100 * max(1.0, n / 32))
                assert pt.match_eigs(ref.values[fin], ps.values[np.isfinite(ps.values)]) < 1e-8 * scale, (n, p, lr, m)
                p0 = eng.pschur_([a.copy(order="F") for a in A], lr, S=S, wantT=False, wantZ=False)
                assert pt.match_eigs(ref.values[fin], p0.values[np.isfinite(p0.values)]) < 1e-8 * scale, (n, p, lr, m)
    finally:
        eng.set_train_g(m0)


# ---- complex generalized (signed) path: generalized.jl with S containing false; test/generalized.jl complex parts ----
def _zg_run(eng, A, S, tol=100, **kw):
    ps = eng.zpschur_hess_(A[0].copy(order="F"), [a.copy(order="F") for a in A[1:]], S, **kw)
    pt.gpschur_check(A, S, ps, tol=tol)
    return ps


def _zg_match_oracle(A, S, ps, rtol=1e-9):
    po = pt.oracle_zpschur_hess(A[0], A[1:], S)
    assert po.info == 0
    fin = np.isfinite(po.values)
    assert fin.sum() == np.isfinite(ps.values).sum()
    if fin.any():
        assert pt.match_eigs(po.values[fin], ps.values[np.isfinite(ps.values)]) < rtol * max(1.0, abs(po.values[fin]).max())
    return po


def case_zg_hess_ut(eng, p):
    # test/generalized.jl:67-76 (ComplexF64)
    S = [True, False] + [True] * (p - 2)
    for seed in range(3):
        A = zhess_ut(5, p, 1800 + 10 * p + seed)
        ps = _zg_run(eng, A, S)
        _zg_match_oracle(A, S, ps)
    A = zhess_ut(5, p, 1831 + p)
    pr = eng.zpschur_hess_(A[0].copy(order="F"), [a.copy(order="F") for a in A[1:]], S, rev=True)
    assert pr.orientation == "L" and pr.schurindex == p and pr.S == S[::-1]
    pt.gpschur_check(A[1:][::-1] + [A[0]], S[::-1], pr)


def case_zg_holes(eng):
    """test/generalized.jl:77-152 (ComplexF64) and :165-174 (n = 32, p = 4, S = [T,F,T,F], A[2][3,3] = 0)."""
    for (S, l, j) in RG_HOLES:
        A = zhess_ut(5, 5, 1900)
        A[l - 1][j - 1, j - 1] = 0.0
        ps = _zg_run(eng, A, S)
        po = _zg_match_oracle(A, S, ps)
        c2, c3 = ps.stats.reserved % 1000, ps.stats.reserved // 1000
        assert (c2, c3) == ((po.sweeplog[:, 0] == 2).sum(), (po.sweeplog[:, 0] == 3).sum())
        assert c2 + c3 >= 1
    for (n, p, S, l, j) in [(32, 4, [True, False, True, False], 2, 3), (40, 4, [True, True, False, True], 3, 30),
                            (40, 4, [True, False, True, True], 2, 8), (30, 21, None, 5, 11)]:
        if S is None:
            S = [True] + [bool((q * 5 + 1) % 3) for q in range(1, p)]
        A = zhess_ut(n, p, 140 + n + p)
        A = [np.asfortranarray(a + 2 * np.eye(n)) if k > 0 else a for k, a in enumerate(A)]
        A[l - 1][j - 1, j - 1] = 0.0
        ps = _zg_run(eng, A, S, tol=100 * max(1, n / 32))
        _zg_match_oracle(A, S, ps, rtol=1e-8)


def case_zg_windows(eng, sizes):
    for (n, p, pat) in sizes:
        if pat == "alt":
            S = [True] + [bool(q % 2 == 0) for q in range(1, p)]
        elif pat == "neg":
            S = [True] + [False] * (p - 1)
        else:
            S = [True] + [bool((q * 7 + n) % 3) for q in range(1, p)]
            if all(S):
                S[-1] = False
        A = zhess_ut(n, p, 1300 + n + p)
        A = [np.asfortranarray(a + 1.0 * np.eye(n)) if k > 0 else a for k, a in enumerate(A)]
        ps = _zg_run(eng, A, S, tol=100 * max(1, np.sqrt(n / 32)))
        _zg_match_oracle(A, S, ps, rtol=1e-8)
        assert ps.stats.nsweeps > 0
        fast = eng.zpschur_hess_(A[0].copy(order="F"), [a.copy(order="F") for a in A[1:]], S, wantZ=False, wantT=False)
        assert pt.match_eigs(ps.values, fast.values) < 1e-8 * abs(ps.values).max()


def case_zg_phessenberg(eng, p):
    # test/generalized.jl:1-40 (ComplexF64)
    n = 5
    S = [True]
    for _ in range(1, p):
        S.append(not S[-1])
    A = pt.rand_uniform_zfactors(n, p, seed=1700 + p)
    Hs, Qs = eng.gphessenberg_([a.copy(order="F") for a in A], S)
    pt.sg_hess_check(A, S, Hs, Qs)
    for (n2, p2, S2) in [(9, 6, [True, True, False, False, True, False]), (33, 4, [True, False, False, False]),
                         (40, 3, [True, False, True]), (30, 22, None)]:
        if S2 is None:
            S2 = [True] + [bool((q * 5) % 3) for q in range(1, p2)]
        A = pt.bench_factors(n2, p2, seed=3 + n2, dtype=np.complex128)
        Hs, Qs = eng.gphessenberg_([a.copy(order="F") for a in A], S2)
        pt.sg_hess_check(A, S2, Hs, Qs, tol=20 * max(1, n2 / 8), qtol=10 * max(1, n2 / 16))


def case_zg_full(eng, lr):
    # test/generalized.jl:188-232
    n, p = 5, 4
    S = [True, False, True, False] if lr == "R" else [False, True, False, True]
    for seed in range(3):
        A = pt.rand_uniform_zfactors(n, p, 1970 + seed)
        ps = eng.pschur_([a.copy(order="F") for a in A], lr, S=S)
        assert ps.schurindex == (1 if lr == "R" else p) and ps.orientation == lr
        pt.gpschur_check(A, S, ps)
        po = pt.oracle_gpschur(A, S, lr)
        assert pt.match_eigs(po.values, ps.values) < 1e-9 * max(1.0, abs(po.values).max())
    for (n2, p2) in [(24, 3), (40, 6)]:
        S2 = [bool((q * 7 + n2) % 3) for q in range(p2)]
        S2[p2 - 1 if lr == "L" else 0] = True
        if all(S2):
            S2[1] = False
        A = pt.bench_factors(n2, p2, seed=n2 + p2, dtype=np.complex128)
        ps = eng.pschur_([a.copy(order="F") for a in A], lr, S=S2)
        pt.gpschur_check(A, S2, ps, tol=100 * max(1.0, np.sqrt(n2 / 32)), qtol=10 * max(1.0, np.sqrt(n2 / 32)))
        po = pt.oracle_gpschur(A, S2, lr)
        assert pt.match_eigs(po.values, ps.values) < 1e-8 * max(1.0, abs(po.values).max())


# ---- ordschur! for GeneralizedPeriodicSchur (signed 1x1 swaps): test/ordschur.jl:165-222 ----
def case_gordschur_reference(eng, cplx, lr):
    import psd_amd

    p, n, nsel = 5, 7, 2
    S = [True] * p
    for l in range(0, p - 1, 2):
        S[l] = False
    A = pt.gord_test_factors(n, p, S, seed=31, cplx=cplx)
    if lr == "R":
        A, S = A[::-1], S[::-1]
    ps0 = eng.pschur_([a.copy(order="F") for a in A], lr, S=S)
    check = pt.gpschur_check if cplx else pt.rgpschur_check
    check(A, S, ps0)
    lam0 = ps0.values.copy()
    po0 = pt.GPSD(S, [t.copy() for t in ps0.Ts], [z.copy() for z in ps0.Z], ps0.alpha, ps0.beta, ps0.alphascale, lr,
                  ps0.schurindex)
    for rev in (False, True):
        idx = np.argsort(-np.abs(lam0) if rev else np.abs(lam0))
        select = np.zeros(n, dtype=bool)
        select[idx[:nsel]] = True
        P1 = psd_amd.GeneralizedPeriodicSchur(S, [t.copy(order="F") for t in ps0.Ts], [z.copy(order="F") for z in ps0.Z],
                                              ps0.alpha.copy(), ps0.beta.copy(), ps0.alphascale.copy(), lr, ps0.schurindex)
        ps1 = eng.ordschur_(P1, select)
        check(A, S, ps1)
        for j in range(nsel):
            assert np.any(np.isclose(ps1.values[:nsel], lam0[idx[j]], rtol=1e-8))
        po = pt.oracle_gordschur(po0, select)
        assert po.info == 0 and po.nswaps == ps1.stats.nsweeps
        assert pt.match_eigs(po.values, ps1.values) < 1e-9 * abs(po.values).max()


def case_gordschur_windows(eng, sizes):
    """larger signed problems: several windows, every signature class, with and without Z"""
    import psd_amd

    for (n, p, cplx, lr) in sizes:
        S = [bool((q * 7 + n) % 3) for q in range(p)]
        if all(S):
            S[p // 2] = False
        S[p - 1 if lr == "R" else 0] = True  # (after the reversal below the leftmost working entry is true)
        A = pt.gord_test_factors(n, p, S if lr == "L" else S, seed=n + p, cplx=cplx)
        if lr == "R":
            A, S = A[::-1], S[::-1]
        ps0 = eng.pschur_([a.copy(order="F") for a in A], lr, S=S)
        lam0 = ps0.values.copy()
        order = np.argsort(np.abs(lam0))
        select = np.zeros(n, dtype=bool)
        select[order[: n // 2]] = True
        check = pt.gpschur_check if cplx else pt.rgpschur_check
        ps1 = eng.ordschur_(ps0, select)
        assert ps1.stats.nsweeps > 0
        check(A, S, ps1, tol=100 * max(1.0, np.sqrt(n / 32)), qtol=10 * max(1.0, np.sqrt(n / 32)))
        m = int(select.sum())
        assert pt.match_eigs(lam0[select], ps1.values[:m]) < 1e-7 * abs(lam0).max()


def case_gordschur_pairs_reference(eng):
    """test/ordschur.jl:226-275 "gen. ordschur Float64: conjugate pair(s)" (mkrps, alt = true)"""
    import psd_amd

    n, p = 7, 5
    ps0, A = pt.mkrps(n, p, [3, 6], seed=905, alt=True)
    lam0 = ps0.values
    for selset in ([6, 7], [3, 4], [1, 2, 5], [1, 3, 4], [1, 2, 6, 7], [5]):
        select = np.zeros(n, dtype=bool)
        select[[j - 1 for j in selset]] = True
        P1 = psd_amd.GeneralizedPeriodicSchur(ps0.S, [t.copy(order="F") for t in ps0.Ts], [z.copy(order="F") for z in ps0.Z],
                                              ps0.alpha.copy(), ps0.beta.copy(), ps0.ascale.copy(), "L", p)
        ps1 = eng.ordschur_(P1, select)
        pt.rgpschur_check(A, ps0.S, ps1, tol=400)
        nsel = len(selset)
        for j in selset:
            assert np.any(np.isclose(ps1.values[:nsel], lam0[j - 1], rtol=1e-7)), (selset, ps1.values)
        po = pt.oracle_gordschur(ps0, select)
        assert po.info == 0 and po.nswaps == ps1.stats.nsweeps
        assert pt.match_eigs(po.values, ps1.values) < 1e-9 * abs(po.values).max()


def case_gordschur_pairs_random(eng, sizes):
    """real signed decompositions with conjugate pairs from pschur!(A, S, lr); smaller half selected"""
    for (n, p, lr, seed) in sizes:
        S = [bool((q * 7 + n) % 3) for q in range(p)]
        if all(S):
            S[1] = False
        S[p - 1 if lr == "L" else 0] = True
        A = pt.bench_factors(n, p, seed=seed + 40)
        ps0 = eng.pschur_([a.copy(order="F") for a in A], lr, S=S)
        lam0 = ps0.values.copy()
        assert np.any(lam0.imag != 0)
        thr = np.sort(np.abs(lam0))[n // 2]
        select = np.abs(lam0) <= thr
        ps1 = eng.ordschur_(ps0, select)
        assert ps1.stats.nsweeps > 0
        pt.rgpschur_check(A, S, ps1, tol=400 * max(1.0, np.sqrt(n / 32)), qtol=10 * max(1.0, np.sqrt(n / 32)))
        m = int(select.sum())
        sc = abs(lam0).max()
        assert pt.match_eigs(lam0[select], ps1.values[:m]) < 1e-8 * sc
        assert pt.match_eigs(lam0[~select], ps1.values[m:]) < 1e-8 * sc


def case_gpschur_pairs(eng):
    """gpschur(As, Bs) (generalized.jl:1191-1211): eigenvalues of B_p^-1 A_p ... B_1^-1 A_1"""
    for (n, ph, seed) in [(5, 1, 1), (6, 2, 2), (7, 3, 3), (20, 4, 4)]:
        As = pt.bench_factors(n, ph, seed=seed + 70)
        Bs = pt.bench_factors(n, ph, seed=seed + 170)
        ps = eng.gpschur(As, Bs)
        assert len(ps.Ts) == 2 * ph and ps.S == [True, False] * ph
        Cs = [np.asarray(As[ph - 1], dtype=complex), np.asarray(Bs[0 if ph == 1 else ph - 2], dtype=complex)]
        for j in range(ph - 1, 0, -1):
            Cs += [np.asarray(As[j - 1], dtype=complex), np.asarray(Bs[(ph if j == 1 else j - 1) - 1], dtype=complex)]
        pt.gpschur_check(Cs, ps.S, ps, tol=100 * max(1.0, n / 8))
        P = np.eye(n, dtype=complex)
        for j in range(ph):  # left operator order: factor j+1 applied after factor j
            P = np.linalg.solve(Bs[j], As[j] @ P)
        lam = np.linalg.eigvals(P)
        assert pt.match_eigs(lam, ps.values) < 1e-8 * abs(lam).max()


# ------------------------------------------------------------------------------------------------
# ordschur! alignments: the reference accepts schurindex 1 or p in either orientation (ordschur.jl:20-33,
# rordschur.jl:15-27: _rev_alias for 'R', _circshift(P, 1) for schurindex p).  A decomposition with the quasi-triangular
# factor at the other end of the list is the circular shift by one of all lists (utils.jl:6-40); it maps to the same
# working arrays, so the reordered factors must agree bit for bit with those of the natural alignment.
def _shifted(lst, lr):
    lst = list(lst)
    return (lst[1:] + lst[:1]) if lr == "R" else (lst[-1:] + lst[:-1])


def case_ordschur_alignments(eng):
    import psd_amd

    n, p = 9, 4
    for cplx in (True, False):
        for signed in (False, True):
            for lr in "RL":
                S = [True] * p
                if signed:
                    S[1] = S[2] = False
                    if lr == "L":
                        S = S[::-1]
                A = pt.bench_factors(n, p, seed=77 + p, dtype=np.complex128 if cplx else np.float64)
                A = [np.asfortranarray(a) for a in A]
                ps0 = eng.pschur_([a.copy(order="F") for a in A], lr, S=S) if signed else eng.pschur(A, lr)
                k0 = ps0.schurindex
                assert k0 == (1 if lr == "R" else p)
                lam0 = ps0.values.copy()
                thr = np.sort(np.abs(lam0))[n // 2]
                select = np.abs(lam0) <= thr
                k1 = p if lr == "R" else 1

                def build(Ts, Zs, k, Sx):
                    Ts = [t.copy(order="F") for t in Ts]
                    Zs = [z.copy(order="F") for z in Zs]
                    if signed:
                        return psd_amd.GeneralizedPeriodicSchur(Sx, Ts, Zs, ps0.alpha.copy(), ps0.beta.copy(),
                                                                ps0.alphascale.copy(), lr, k)
                    return psd_amd.PeriodicSchur(Ts, Zs, lam0.copy(), lr, k)

                P_nat = eng.ordschur_(build(ps0.Ts, ps0.Z, k0, S), select)
                As, Ss = _shifted(A, lr), _shifted(S, lr)
                P_sh0 = build(_shifted(ps0.Ts, lr), _shifted(ps0.Z, lr), k1, Ss)
                ok, err = pt.checkpsd(P_sh0, As, S=Ss, strict=False)
                assert ok, ("shifted input is not a decomposition", cplx, signed, lr, err)
                P_sh = eng.ordschur_(P_sh0, select)
                assert P_sh.stats.nsweeps == P_nat.stats.nsweeps > 0
                for a, b in zip(_shifted(P_nat.Ts, lr), P_sh.Ts):
                    assert np.array_equal(a, b), (cplx, signed, lr)
                for a, b in zip(_shifted(P_nat.Z, lr), P_sh.Z):
                    assert np.array_equal(a, b), (cplx, signed, lr)
                assert np.array_equal(P_nat.values, P_sh.values)
                ok, err = pt.checkpsd(P_sh, As, S=Ss, strict=False, thresh=400)
                assert ok, (cplx, signed, lr, err)
                m = int(select.sum())
                assert pt.match_eigs(lam0[select], P_sh.values[:m]) < 1e-8 * abs(lam0).max()
    # a schurindex strictly inside the period is an ArgumentError (ordschur.jl:32)
    ps0 = eng.pschur(pt.bench_factors(5, 3, seed=5), "R")
    bad = psd_amd.PeriodicSchur(ps0.Ts, ps0.Z, ps0.values, "R", 2)
    try:
        eng.ordschur_(bad, np.array([False, True, False, False, False]))
        raise AssertionError("schurindex 2 of 3 must be rejected")
    except ValueError:
        pass


# ------------------------------------------------------------------------------------------------
# Stage 2 of the signed Hessenberg reduction: the pipelined multi-wave kernel against the single-wave chase
# (PSD_HESS_SERIAL=1).  Same rotations up to the order in which A_1 takes its row and column updates, so the factors
# agree to rounding and both satisfy the invariants of test/generalized.jl:1-40.
def case_hess_pipeline_vs_serial(eng):
    for (n, p, cplx) in [(40, 7, False), (36, 19, False), (28, 6, True), (24, 18, True), (12, 1, False), (10, 2, True)]:
        S = [True] + [bool((q * 5 + n) % 3) for q in range(1, p)]
        A = pt.bench_factors(n, p, seed=11 + n, dtype=np.complex128 if cplx else np.float64)
        old = os.environ.pop("PSD_HESS_SERIAL", None)
        try:
            Hp, Qp = eng.gphessenberg_([a.copy(order="F") for a in A], S)
            os.environ["PSD_HESS_SERIAL"] = "1"
            Hs, Qs = eng.gphessenberg_([a.copy(order="F") for a in A], S)
        finally:
            os.environ.pop("PSD_HESS_SERIAL", None)
            if old is not None:
                os.environ["PSD_HESS_SERIAL"] = old
        pt.sg_hess_check(A, S, Hp, Qp, tol=20 * max(1, n / 8), qtol=10 * max(1, n / 16))
        pt.sg_hess_check(A, S, Hs, Qs, tol=20 * max(1, n / 8), qtol=10 * max(1, n / 16))
        for a, b in zip(Hp, Hs):
            assert np.abs(a - b).max() <= 1e-10 * max(1.0, np.abs(b).max()), (n, p, cplx, np.abs(a - b).max())


# ------------------------------------------------------------------------------------------------
# _rphessenberg! (rhessx.jl:55-109): invariants of the reduction and agreement with the
# CPU restatement.  Shapes as the Krylov driver uses them (krylov.jl:801-809): Ap with one extra row, Q = I.
def case_rphessenberg(eng):
    import psd_amd

    rs = np.random.RandomState(17)
    for cplx in (False, True):
        dt = np.complex128 if cplx else np.float64
        for (m, n, p) in [(6, 6, 1), (7, 6, 1), (9, 8, 3), (8, 8, 3), (21, 20, 5), (3, 2, 2), (2, 2, 4), (41, 40, 2), (1, 1, 3)]:
            def rnd(a, b):
                x = rs.randn(a, b)
                return np.asfortranarray((x + 1j * rs.randn(a, b)) if cplx else x).astype(dt, order="F")

            Ap0 = rnd(m, n)
            As0 = [rnd(n, n) for _ in range(p - 1)]
            Ap = Ap0.copy(order="F")
            As = [a.copy(order="F") for a in As0]
            Qs = [np.asfortranarray(np.eye(n, dtype=dt)) for _ in range(p)]
            eng.rphessenberg_(Ap, As, Qs)
            pt.rphess_check(Ap0, As0, Ap, As, Qs)
            Apo, Aso, Qso = pt.oracle_rphessenberg(Ap0, As0, [np.eye(n, dtype=dt) for _ in range(p)])
            sc = max(1.0, np.abs(Apo).max())
            assert np.abs(Ap - Apo).max() < 1e-11 * sc, (m, n, p, cplx, np.abs(Ap - Apo).max())
            for a, b in zip(As, Aso):
                assert np.abs(a - b).max() < 1e-11 * max(1.0, np.abs(b).max())
            for a, b in zip(Qs, Qso):
                assert np.abs(a - b).max() < 1e-11
            # without Q: same factors
            Ap2 = Ap0.copy(order="F")
            As2 = [a.copy(order="F") for a in As0]
            eng.rphessenberg_(Ap2, As2, None)
            assert np.abs(Ap2 - Ap).max() < 1e-13 * sc
    try:
        eng.rphessenberg_(np.asfortranarray(np.zeros((5, 3))), [], None)
        raise AssertionError("m = n + 2 must be rejected")
    except ValueError:
        pass


# ordschur!(P, select; Z = supplementary matrices) — ordschur.jl:17,34-42: the swaps act on Z instead of P.Z
def case_ordschur_supplementary_z(eng):
    import psd_amd

    n, p = 8, 3
    for cplx in (True, False):
        A = pt.bench_factors(n, p, seed=91, dtype=np.complex128 if cplx else np.float64)
        ps0 = eng.pschur(A, "L")
        lam0 = ps0.values.copy()
        select = np.abs(lam0) <= np.sort(np.abs(lam0))[n // 2]
        ref = eng.ordschur_(_clone(ps0), select)
        P = _clone(ps0)
        Zkeep = [z.copy() for z in P.Z]
        Zsup = [z.copy(order="F") for z in P.Z]
        out = eng.ordschur_(P, select, Z=Zsup)
        assert all(np.array_equal(a, b) for a, b in zip(out.Z, Zkeep))  # P.Z untouched
        assert all(np.array_equal(a, b) for a, b in zip(Zsup, ref.Z))   # the supplementary set took the swaps
        assert all(np.array_equal(a, b) for a, b in zip(out.Ts, ref.Ts))
        try:
            eng.ordschur_(_clone(eng.pschur(A, "R")), select, Z=Zsup)
            raise AssertionError("supplementary Z with the right orientation must be rejected")
        except psd_amd.NotImplementedPSD:
            pass


# ------------------------------------------------------------------------------------------------
# Multishift trains of the real pschur! path against the reference's one-shift-one-sweep
# iteration (psd_set_train(0)): same eigenvalues to the north-star tolerance, the reference's invariants in both modes,
# both orientations, the wantT / wantZ fast paths, p = 1.
def case_trains(eng, sizes):
    m0 = eng.get_train()
    try:
        for (n, p, lr) in sizes:
            As = pt.bench_factors(n, p, seed=300 + n + p)
            P = pt.product(As, lr == "L")
            Pn = np.linalg.norm(P, 2)
            eng.set_train(0)
            ref = eng.pschur(As, lr)
            assert ref.stats.reserved == 0
            for m in (2, 6):
                eng.set_train(m)
                ps = eng.pschur(As, lr)
                assert ps.stats.reserved > 0, (n, p, lr, m, "no train ran")
                # (a train may run narrower windows than a single sweep: more, cheaper ticks; only a runaway count fails)
                assert ps.stats.nlaunch_step < 2 * ref.stats.nlaunch_step
                ok, err = pt.checkpsd(ps, As, thresh=100 * np.sqrt(max(n / 32, 1)))
                assert ok, (n, p, lr, m, err)
                assert pt.match_eigs(ref.values, ps.values) <= 1e-10 * Pn
                assert pt.match_eigs(np.linalg.eigvals(P), ps.values) <= 1e-10 * Pn
                # eigenvalues-only and T-without-Z fast paths (PSD.jl:120-124, :675-678)
                p0 = eng.pschur(As, lr, wantT=False, wantZ=False)
                assert len(p0.Z) == 0 and pt.match_eigs(ref.values, p0.values) <= 1e-10 * Pn
                p1 = eng.pschur(As, lr, wantT=True, wantZ=False)
                assert len(p1.Z) == 0 and pt.match_eigs(ref.values, p1.values) <= 1e-10 * Pn
    finally:
        eng.set_train(m0)


# complex single-shift path: trains (shifts = eigenvalues of the trailing block of the product) against the reference's
# one-shift iteration
def case_ztrains(eng, sizes):
    m0 = eng.get_train_z()
    try:
        for (n, p, lr) in sizes:
            As = pt.bench_factors(n, p, seed=400 + n + p, dtype=np.complex128)
            P = pt.product(As, lr == "L")
            Pn = np.linalg.norm(P, 2)
            eng.set_train_z(0)
            ref = eng.pschur(As, lr)
            for m in (2, 6):
                eng.set_train_z(m)
                ps = eng.pschur(As, lr)
                assert ps.stats.nlaunch_step < 2 * ref.stats.nlaunch_step, (n, p, lr, m)  # (narrower windows: more, cheaper ticks)
                ok, err = pt.checkpsd(ps, As, thresh=100 * np.sqrt(max(n / 32, 1)))
                assert ok, (n, p, lr, m, err)
                assert pt.match_eigs(ref.values, ps.values) <= 1e-10 * Pn
                assert pt.match_eigs(np.linalg.eigvals(P), ps.values) <= 1e-10 * Pn
                p0 = eng.pschur(As, lr, wantT=False, wantZ=False)
                assert pt.match_eigs(ref.values, p0.values) <= 1e-10 * Pn
    finally:
        eng.set_train_z(m0)


# ---- device-side checkpsd (src/diagnostics.jl:190-263, csrc/psd_check.h) -----------------------------------------
def _checkpsd_numpy_details(ps, As, S=None):
    """orthogonality / triangularity norms the way pt.checkpsd (the numpy restatement of the reference) forms them."""
    real = not np.iscomplexobj(ps.Ts[0])
    n = As[0].shape[0]
    orth = np.array([np.linalg.norm(z @ z.conj().T - np.eye(n)) for z in ps.Z])
    tri = np.array([np.linalg.norm(np.tril(t, -2 if (real and l == ps.schurindex - 1) else -1))
                    for l, t in enumerate(ps.Ts)])
    return orth, tri


def case_checkpsd(eng, sizes):
    """The device verifier against the numpy restatement of the reference's checkpsd on decompositions produced by the
    engine (real + complex, both orientations, signed), and on deliberately damaged ones (it has to say no)."""
    for (n, p, lr, kind) in sizes:
        cplx = kind.startswith("z")
        S = None
        if kind.endswith("g"):  # signed
            S = [True] + [bool(q % 2 == 0) for q in range(1, p)]
        A = pt.bench_factors(n, p, seed=900 + n + p, dtype=np.complex128 if cplx else np.float64)
        if S is None:
            ps = eng.pschur(A, lr)
        else:
            ps = eng.pschur_([a.copy(order="F") for a in A], lr, S=S)
        thresh = 100 * np.sqrt(max(n / 32, 1))
        ok_ref, err_ref = pt.checkpsd(ps, A, thresh=thresh, S=S)
        ok, err, orth, tri = eng.checkpsd(ps, A, thresh=thresh, S=S, details=True)
        orth_ref, tri_ref = _checkpsd_numpy_details(ps, A, S)
        assert ok == ok_ref and ok, (n, p, lr, kind, err, err_ref)
        # the residual is a difference at rounding level: two summation orders agree to a few eps-units, not to digits
        assert np.all(np.abs(err - err_ref) <= 0.25 * err_ref + 2.0), (err, err_ref)
        assert np.all(np.abs(orth - orth_ref) <= 0.5 * orth_ref + 4 * pt.EPS * np.sqrt(n)), (orth, orth_ref)
        assert np.all(tri == tri_ref)
        # damage: a non-zero below the diagonal, a perturbed Z, a perturbed T
        j = p // 2
        keep = ps.Ts[j][n - 1, 0]
        ps.Ts[j][n - 1, 0] = 1e-3
        ok1, _, _, tri1 = eng.checkpsd(ps, A, thresh=thresh, S=S, details=True)
        assert not ok1 and abs(tri1[j] - 1e-3) < 1e-12
        ok1s, _ = eng.checkpsd(ps, A, thresh=1e9, strict=False, S=S)
        assert not ok1s  # 1e-3 is far above 10 eps n as well
        ps.Ts[j][n - 1, 0] = keep
        Zk = ps.Z[0].copy()
        ps.Z[0] = np.asfortranarray(ps.Z[0] * (1 + 1e-9))
        ok2, err2, orth2, _ = eng.checkpsd(ps, A, thresh=thresh, S=S, details=True)
        ok2_ref, err2_ref = pt.checkpsd(ps, A, thresh=thresh, S=S)
        assert not ok2 and not ok2_ref and orth2[0] > 10 * pt.EPS * n
        assert np.allclose(err2, err2_ref, rtol=1e-3, atol=2.0)
        ps.Z[0] = Zk
        keep = ps.Ts[0][0, n - 1]
        ps.Ts[0][0, n - 1] += 1e-6
        ok3, err3 = eng.checkpsd(ps, A, thresh=thresh, S=S)
        ok3_ref, err3_ref = pt.checkpsd(ps, A, thresh=thresh, S=S)
        assert not ok3 and not ok3_ref
        assert np.allclose(err3, err3_ref, rtol=1e-3, atol=2.0)
        ps.Ts[0][0, n - 1] = keep
    # argument errors (diagnostics.jl:194-202)
    with pytest.raises(Exception):
        eng.checkpsd(ps, A[:-1] if p > 1 else A + A)


# ---- eigvecs (src/vectors.jl:25-138; test/vectors.jl, checker test/testfuncs.jl:424-436) ----------------------------
def ev_check(As, Vs, lams, left=True, tol=np.sqrt(pt.EPS)):
    """test/testfuncs.jl:424-436 (left orientation): A_l v_l = mu v_{l+1}, mu = lambda^(1/p); for the right orientation
    the sequence runs the other way (A_l v_{l+1} = mu v_l)."""
    p = len(As)
    for k in range(Vs[0].shape[1]):
        mu = complex(lams[k]) ** (1.0 / p)
        for l in range(p):
            l1 = (l + 1) % p
            if left:
                ref = abs(mu) * np.linalg.norm(Vs[l1][:, k])
                err = np.linalg.norm(As[l] @ Vs[l][:, k] - mu * Vs[l1][:, k])
            else:
                ref = abs(mu) * np.linalg.norm(Vs[l][:, k])
                err = np.linalg.norm(As[l] @ Vs[l1][:, k] - mu * Vs[l][:, k])
            assert err < tol * ref, (k, l, err, ref)


def _distinct_real_factors(n, p, cplx, seed):
    # test/vectors.jl:4-21: diagonal 2^(2j/p) in every factor, small strict upper part, unitary similarity chain
    rs = np.random.RandomState(seed)
    dt = np.complex128 if cplx else np.float64
    A = []
    for _ in range(p):
        u = rs.rand(n, n) + (1j * rs.rand(n, n) if cplx else 0)
        A.append(0.01 * np.triu(u).astype(dt))
    for j in range(n):
        mu = 2.0 ** (2 * (j + 1) / p)
        for l in range(p):
            A[l][j, j] = mu
    for l in range(p):
        g = rs.randn(n, n) + (1j * rs.randn(n, n) if cplx else 0)
        q, _ = np.linalg.qr(g)
        A[l] = q @ A[l]
        l1 = (l + 1) % p
        A[l1] = A[l1] @ q.conj().T
    return [np.asfortranarray(a) for a in A]


def case_eigvecs(eng):
    for cplx in (False, True):
        for p in (5, 1):
            n, nsel = 7, 2
            A = _distinct_real_factors(n, p, cplx, seed=60 + p + 10 * cplx)
            ps0 = eng.pschur(A, "L")
            lam0 = np.array(ps0.values)
            T0 = [t.copy() for t in ps0.Ts]
            for rev in (False, True):  # smallest, largest (test/vectors.jl:25-42)
                idx = np.argsort(-np.abs(lam0) if rev else np.abs(lam0), kind="stable")
                select = np.zeros(n, dtype=bool)
                select[idx[:nsel]] = True
                Vs = eng.eigvecs(ps0, select)
                assert len(Vs) == p and Vs[0].shape == (n, nsel)
                ev_check(A, Vs, lam0[select])
                V1 = eng.eigvecs(ps0, select, shifted=False)
                assert len(V1) == 1 and np.allclose(V1[0], Vs[0])
            assert all(np.array_equal(a, b) for a, b in zip(T0, ps0.Ts))  # ps0 is not modified
    # conjugate pairs of a real decomposition (the 2x2 cyclic problem, vectors.jl:73-112) and a right-oriented one
    for lr in ("L", "R"):
        A = pt.bench_factors(12, 3, seed=91)
        ps0 = eng.pschur(A, lr)
        lam0 = np.array(ps0.values)
        cidx = [j for j in range(12) if lam0[j].imag > 0]
        assert cidx, "test matrix has no complex pair"
        select = np.zeros(12, dtype=bool)
        select[cidx[0]] = True  # one member: eigvecs completes the pair
        ridx = [j for j in range(12) if lam0[j].imag == 0]
        if ridx:
            select[ridx[-1]] = True
        Vs = eng.eigvecs(ps0, select)
        full = select.copy()
        full[cidx[0] + 1] = True
        # eigenvalues in the order the vectors come: selected ones top to bottom after the first ordschur!
        ps1 = eng.ordschur_(type(ps0)([t.copy(order="F") for t in ps0.Ts], [z.copy(order="F") for z in ps0.Z],
                                      lam0.copy(), ps0.orientation, ps0.schurindex), full)
        lams = ps1.values[: int(full.sum())]
        ev_check(A, Vs, lams, left=(lr == "L"))
    # argument errors (vectors.jl:30-36)
    with pytest.raises(ValueError):
        eng.eigvecs(ps0, [True] * 5)


# ---- batch of small Hessenberg-triangular problems (psd_d_pschur_hess_batch; krylov.jl:575-592,800-829) ------------
def _hess_ut_problem(n, p, seed):
    A = [np.asfortranarray(np.triu(a)) for a in pt.bench_factors(n, p, seed=seed)]
    A[0] = np.asfortranarray(np.triu(pt.bench_factors(n, 1, seed=seed + 977)[0], -1))
    return A


def case_pschur_hess_batch(eng, shapes):
    for (nb, n, p) in shapes:
        probs = [_hess_ut_problem(n, p, 300 + 7 * q + n + p) for q in range(nb)]
        # one by one (the reference's order of work) ...
        single = []
        for A in probs:
            W = [a.copy(order="F") for a in A]
            single.append(eng.pschur_hess_(W[0], W[1:]))
        # ... and all in one call
        Ws = [[a.copy(order="F") for a in A] for A in probs]
        batch = eng.pschur_hess_batch_([(W[0], W[1:]) for W in Ws])
        assert len(batch) == nb
        for q in range(nb):
            A = probs[q]
            pt.pschur_check(A, batch[q], tol=100 * max(1.0, np.sqrt(n / 32)), check_lam=False)
            P = pt.product(A)
            sc = np.linalg.norm(P, 2)
            assert pt.match_eigs(np.linalg.eigvals(P), batch[q].values) <= 1e-10 * sc * max(1.0, np.linalg.cond(P) * 1e-6)
            # same iteration as the single call when no trains / splits reorder it; always the same spectrum
            assert pt.match_eigs(single[q].values, batch[q].values) <= 1e-10 * sc
    # wantZ = False, and a batch of one equals the plain call
    A = _hess_ut_problem(12, 3, 5)
    W = [a.copy(order="F") for a in A]
    b1 = eng.pschur_hess_batch_([(W[0], W[1:])], wantZ=False)
    W2 = [a.copy(order="F") for a in A]
    s1 = eng.pschur_hess_(W2[0], W2[1:], wantZ=False)
    assert pt.match_eigs(s1.values, b1[0].values) <= 1e-10 * np.linalg.norm(pt.product(A), 2)
    with pytest.raises(Exception):
        eng.pschur_hess_batch_([(W[0], W[1:]), (W[0][:5, :5].copy(order="F"), [w[:5, :5].copy(order="F") for w in W[1:]])])


def case_pschur_hess_batch_one_fails(eng, n=24, p=3):
    """A batch in which ONE problem exhausts its sweep budget (maxitfac = 2: 2 n sweeps, PSD.jl:471,891-893; every deflation
    step takes at least one off it, PSD.jl:1057-1060): its info says
    so, the other problems of the call are complete and correct (the failure used to stop the whole batch
    with info = 0 for the unfinished problems)."""
    def easy(seed):  # one 3 x 3 block in an otherwise triangular H1: a handful of sweeps
        A = _hess_ut_problem(n, p, seed)
        sub = np.diag(A[0], -1).copy()
        keep = np.zeros_like(sub)
        for k in (9, 10):
            keep[k] = sub[k]
        A[0] = np.asfortranarray(np.triu(A[0]) + np.diag(keep, -1))
        return A
    probs = [easy(41), _hess_ut_problem(n, p, 42), easy(43), easy(44)]
    Ws = [[a.copy(order="F") for a in A] for A in probs]
    infos = []
    out = eng.pschur_hess_batch_([(W[0], W[1:]) for W in Ws], maxitfac=2, infos_out=infos)
    assert len(infos) == 4 and infos[0] == 0 and infos[2] == 0 and infos[3] == 0, infos
    assert infos[1] != 0, infos  # (a full random Hessenberg problem of order 24 needs more than 48 sweeps)
    for q in (0, 2, 3):
        pt.pschur_check(probs[q], out[q], tol=100, check_lam=False)
        P = pt.product(probs[q])
        assert pt.match_eigs(np.linalg.eigvals(P), out[q].values) <= 1e-10 * np.linalg.norm(P, 2)
    with pytest.raises(Exception):  # the default form raises for the failed problem after all have run
        Ws = [[a.copy(order="F") for a in A] for A in probs]
        eng.pschur_hess_batch_([(W[0], W[1:]) for W in Ws], maxitfac=2)


def case_zformq_blocked(make_engine, sizes):
    """ComplexF64: B reflectors per pass over the Q_j (csrc/psd_zhess.h, psd_zformq_blk) against one launch per reflector:
    same T and eigenvalues, Z equal to rounding, unitary."""
    blocked = make_engine({"PSD_FORMQ_BLOCKED": "1"})
    plain = make_engine({"PSD_FORMQ_BLOCKED": "0"})
    for (n, p, lr) in sizes:
        A = pt.bench_factors(n, p, seed=300 + n + p, dtype=np.complex128)
        pb = blocked.pschur(A, lr)
        pp = plain.pschur(A, lr)
        assert np.array_equal(pb.values, pp.values)
        for j in range(p):
            assert np.array_equal(pb.Ts[j], pp.Ts[j])
            d = np.abs(pb.Z[j] - pp.Z[j]).max()
            assert d < 50 * pt.EPS * np.sqrt(n), (n, p, j, d)
            orth = np.linalg.norm(pb.Z[j].conj().T @ pb.Z[j] - np.eye(n))
            assert orth < 10 * pt.EPS * n, (n, p, j, orth)


def case_formq_blocked(make_engine, sizes):
    """Blocked (compact-WY, csrc/psd_formq2.h) against reflector-by-reflector (csrc/psd_hess.h) materialisation of the
    Q_j (PSD.jl:136-143): both engines reduce and iterate the same H, so their Z_j differ by the rounding of Q_j only.
    make_engine(env) -> Engine created with that environment."""
    blocked = make_engine({"PSD_FORMQ_BLOCKED": "1"})
    plain = make_engine({"PSD_FORMQ_BLOCKED": "0"})
    for (n, p, lr) in sizes:
        A = pt.bench_factors(n, p, seed=300 + n + p)
        pb = blocked.pschur(A, lr)
        pp = plain.pschur(A, lr)
        assert np.array_equal(pb.values, pp.values)  # (the iteration never sees Q)
        for j in range(p):
            assert np.array_equal(pb.Ts[j], pp.Ts[j])
            d = np.abs(pb.Z[j] - pp.Z[j]).max()
            assert d < 50 * pt.EPS * np.sqrt(n), (n, p, j, d)
            orth = np.linalg.norm(pb.Z[j].T @ pb.Z[j] - np.eye(n))
            assert orth < 10 * pt.EPS * n, (n, p, j, orth)
