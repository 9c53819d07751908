"""GPU tier: factor-sliced sweep windows (csrc/psd_slice3.h, psd_set_slices) — the north_star partition (SURVEY.md section
8e: factors sharded by period, the chain handed over at slice boundaries) as device code, with the G workgroups of a window
slot as the ranks and the inboxes as plain device pointers."""
import numpy as np
import pytest

import psdtest as pt

pytestmark = pytest.mark.gpu


def test_sliced_windows_real(gpu_engine):
    """pschur!(A, lr) with G = 2 and G = 4 workgroups per sweep window, each holding the window blocks of its slice of the
    period only, against the unsliced engine and the CPU oracle (eigenvalues at 1e-10 ||prod A||) and checkpsd; periods the
    slices split evenly and raggedly, both orientations, the longest period the scan chase serves.  What crosses a slice
    boundary is a chain vector, and a reflector is a function of it, so the result does not depend on the NUMBER of
    slices: G = 2 and G = 4 agree to the bit (the unsliced kernel is a different instruction stream — the compiler
    contracts its multiply-adds in its own way — and agrees to rounding)."""
    import torch

    torch.cuda.init()
    import psd_amd

    engs = {}
    for G in (2, 4):
        engs[G] = psd_amd.Engine(device=0)
        engs[G].set_slices(G)
        assert engs[G].get_slices() == G
    for (n, p, lr, oracle) in [(96, 8, "R", True), (130, 10, "L", True), (200, 12, "R", False), (150, 64, "L", False), (260, 16, "R", False)]:
        A = pt.bench_factors(n, p, seed=900 + n + p)
        ref = gpu_engine.pschur(A, lr)
        P = pt.product(A, left=(lr == "L"))
        nP = np.linalg.norm(P, 2)
        lam = np.linalg.eigvals(P)
        res = {}
        for G in (2, 4):
            ps = engs[G].pschur(A, lr)
            res[G] = ps
            ok, err = engs[G].checkpsd(ps, A, thresh=100 * np.sqrt(max(n / 32, 1)))
            assert ok, (G, n, p, lr, float(err.max()))
            assert pt.match_eigs(lam, ps.values) <= 1e-10 * nP
            assert pt.match_eigs(ref.values, ps.values) <= 1e-10 * nP
            if oracle:
                po = pt.oracle_pschur(A, lr)
                assert pt.match_eigs(po.values, ps.values) <= 1e-10 * nP
        assert res[2].stats.nsweeps == res[4].stats.nsweeps
        assert np.array_equal(res[2].values, res[4].values)
        for j in range(p):
            assert np.array_equal(res[2].Ts[j], res[4].Ts[j]), (n, p, lr, j)
            assert np.array_equal(res[2].Z[j], res[4].Z[j]), (n, p, lr, j)


def test_eight_slices_are_clamped_to_what_fits(gpu_engine):
    """G = 8 on one GPU: 64 x 8 workgroups that each take a compute unit's LDS and wait for one another cannot all be
    resident on 256 CUs, so the engine runs 4 slices (clamped, not an error and not a hang) — and, the result not depending
    on the number of slices, gives exactly what G = 4 gives."""
    import torch

    torch.cuda.init()
    import psd_amd

    e8 = psd_amd.Engine(device=0)
    e8.set_slices(8)
    e4 = psd_amd.Engine(device=0)
    e4.set_slices(4)
    for (n, p, lr) in [(200, 16, "R"), (150, 64, "L")]:
        A = pt.bench_factors(n, p, seed=950 + n + p)
        p8 = e8.pschur(A, lr)
        p4 = e4.pschur(A, lr)
        ok, err = e8.checkpsd(p8, A, thresh=100 * np.sqrt(max(n / 32, 1)))
        assert ok, (n, p, float(err.max()))
        assert np.array_equal(p8.values, p4.values)
        for j in range(p):
            assert np.array_equal(p8.Ts[j], p4.Ts[j]) and np.array_equal(p8.Z[j], p4.Z[j])


def test_slices_argument_checks(gpu_engine):
    import psd_amd

    eng = psd_amd.Engine(device=0)
    with pytest.raises(ValueError):
        eng.set_slices(0)
    with pytest.raises(ValueError):
        eng.set_slices(9)
    # fewer than two factors per slice: the call runs unsliced
    eng.set_slices(4)
    A = pt.bench_factors(60, 5, seed=3)
    ps = eng.pschur(A, "R")
    ok, err = eng.checkpsd(ps, A, thresh=100 * np.sqrt(60 / 32))
    assert ok


def test_sliced_windows_complex(gpu_engine):
    """The same partition for the ComplexF64 engine (csrc/psd_zslice3.h; BASELINE configs[2] names this element type):
    G = 2 and G = 4 workgroups per sweep window against the unsliced engine and the CPU oracle at 1e-10 ||prod A||, and
    checkpsd.  (Runs are reproducible; where the slices coincide with the unsliced chase's groups of four links the result
    is the unsliced one to the bit, otherwise it agrees to rounding: tools/r04/zslice_dbg.py.)"""
    import torch

    torch.cuda.init()
    import psd_amd

    engs = {}
    for G in (2, 4):
        engs[G] = psd_amd.Engine(device=0)
        engs[G].set_slices(G)
    for (n, p, lr, oracle) in [(80, 8, "R", True), (110, 10, "L", True), (160, 12, "R", False), (140, 64, "L", False), (220, 16, "R", False)]:
        A = pt.bench_factors(n, p, seed=700 + n + p, dtype=np.complex128)
        ref = gpu_engine.pschur(A, lr)
        P = pt.product(A, left=(lr == "L"))
        nP = np.linalg.norm(P, 2)
        lam = np.linalg.eigvals(P)
        res = {}
        for G in (2, 4):
            ps = engs[G].pschur(A, lr)
            res[G] = ps
            ok, err = engs[G].checkpsd(ps, A, thresh=100 * np.sqrt(max(n / 32, 1)))
            assert ok, (G, n, p, lr, float(err.max()))
            assert pt.match_eigs(lam, ps.values) <= 1e-10 * nP
            assert pt.match_eigs(ref.values, ps.values) <= 1e-10 * nP
            if oracle:
                po = pt.oracle_zpschur(A, lr)
                assert pt.match_eigs(po.values, ps.values) <= 1e-10 * nP
        again = engs[4].pschur(A, lr)  # reproducible
        assert all(np.array_equal(a, b) for a, b in zip(again.Ts, res[4].Ts))
