"""CPU tier: the device state machines (window chase, bulk apply, Hessenberg kernels) executed by the TEST-ONLY
serial simulation build of the same sources (tests/hostsim).  This is not the product path: the product is the
HIP build, covered by tests/test_gpu_real.py with the same cases."""
import pytest

import engine_cases as ec


@pytest.mark.parametrize("p", [1, 2, 5])
def test_phessenberg(sim_engine, p):
    ec.case_phessenberg(sim_engine, p)


@pytest.mark.parametrize("p", [1, 2, 3, 5])
def test_hess_ut(sim_engine, p):
    ec.case_hess_ut(sim_engine, p)


@pytest.mark.parametrize("p", [5, 20])
def test_expsplit(sim_engine, golden, p):
    ec.case_expsplit(sim_engine, golden, p)


@pytest.mark.parametrize("p", [1, 2, 3, 5])
def test_full_small(sim_engine, golden, p):
    ec.case_full_small(sim_engine, golden, p)


@pytest.mark.parametrize("p", [1, 5])
def test_fast_paths(sim_engine, p):
    ec.case_fast_paths(sim_engine, p)


def test_config1(sim_engine, golden):
    ec.case_config1(sim_engine, golden)


def test_rq_cleanup(sim_engine):
    ec.case_rq_cleanup(sim_engine)
    ec.case_rq_cleanup_windows(sim_engine)


def test_edge(sim_engine):
    ec.case_edge(sim_engine)


def test_window_widths(sim_engine):
    ec.case_window_widths(sim_engine, [(72, 6, 32), (60, 20, 24), (48, 40, 20), (44, 60, 16), (40, 80, 12)])


def test_pschur_hess(sim_engine):
    ec.case_pschur_hess(sim_engine)


@pytest.mark.parametrize("p", [5, 1])
@pytest.mark.parametrize("lr", ["L", "R"])
def test_rordschur_reference_real(sim_engine, p, lr):
    ec.case_rordschur_reference_real(sim_engine, p, lr)


@pytest.mark.parametrize("p", [5, 1])
@pytest.mark.parametrize("selset", [[1, 2, 5], [1, 3, 4], [1, 2, 6, 7]])
def test_rordschur_pairs(sim_engine, p, selset):
    ec.case_rordschur_pairs(sim_engine, p, selset)


def test_rordschur_windows(sim_engine):
    ec.case_rordschur_windows(sim_engine, [(48, 3), (44, 12), (40, 22), (36, 34), (30, 70)])


def test_rordschur_edge(sim_engine):
    ec.case_rordschur_edge(sim_engine)


def test_rordschur_pipelined(built, monkeypatch):
    import os

    import psd_amd

    lib = os.path.join(os.path.dirname(os.path.abspath(__file__)), "hostsim", "_build", "libpsd_hostsim.so")

    def make(env):
        for k, v in env.items():
            monkeypatch.setenv(k, v)
        return psd_amd.Engine(libpath=lib)

    ec.case_rordschur_pipelined(make, [(140, 2, "R", 0.5), (160, 5, "L", 0.25), (90, 3, "R", 0.6)])


def test_zordschur_pipelined(built, monkeypatch):
    import os

    import psd_amd

    lib = os.path.join(os.path.dirname(os.path.abspath(__file__)), "hostsim", "_build", "libpsd_hostsim.so")

    def make(env):
        for k, v in env.items():
            monkeypatch.setenv(k, v)
        return psd_amd.Engine(libpath=lib)

    ec.case_zordschur_pipelined(make, [(130, 2, "R", 0.5), (150, 4, "L", 0.3)], [(130, 3, "R"), (140, 4, "L")])
    assert ec.case_ordschur_pipelined_failure(make) is not None  # (exact arithmetic order of the simulation: the swap is refused)


def test_rphessenberg(sim_engine):
    ec.case_rphessenberg(sim_engine)


def test_trains(sim_engine):
    ec.case_trains(sim_engine, [(150, 3, "R"), (130, 1, "L"), (120, 5, "L")])


def test_checkpsd(sim_engine):
    ec.case_checkpsd(sim_engine, [(12, 3, "R", "d"), (17, 4, "L", "d"), (10, 2, "R", "z"), (14, 3, "L", "z"),
                                  (12, 4, "R", "dg")])


def test_eigvecs(sim_engine):
    ec.case_eigvecs(sim_engine)


def test_pschur_hess_batch(sim_engine):
    ec.case_pschur_hess_batch(sim_engine, [(3, 10, 2), (5, 20, 3), (4, 33, 1)])


def test_pschur_hess_batch_one_fails(sim_engine):
    ec.case_pschur_hess_batch_one_fails(sim_engine)


def test_column_roles_deferred(built, monkeypatch):
    """PSD_CDEFER=2 (what the HIP build does by itself for n >= 1024): the rows of a window's column role that lie more
    than a window's width above it (psd_cdefer_edge) are their own launch (psd_rq_apply_wl mode 6) — on the GPU on the
    second stream beside the next tick's chases; the serial simulation runs it at the latest point the streams allow,
    BEHIND the next tick's chases and decisions, which shows that none of those reads what it writes.  Every element
    still sees the same sequence of operations, so T, Z and the eigenvalues are those of the one-stream order to the
    last bit, with and without the Schur vectors on the second stream as well (PSD_OVERLAP=2).  PSD_RDEFER=2 (the HIP build: for n < 1536)
    does the same with the far columns of the rows roles (psd_rdefer_edge, modes 7 / 8): the serial simulation runs them
    behind the next tick's chases too, in front of the far column roles."""
    import os

    import numpy as np
    import psd_amd
    import psdtest as pt

    lib = os.path.join(os.path.dirname(os.path.abspath(__file__)), "hostsim", "_build", "libpsd_hostsim.so")
    monkeypatch.setenv("PSD_OVERLAP", "0")
    monkeypatch.setenv("PSD_CDEFER", "0")
    ref = psd_amd.Engine(libpath=lib)
    engs = []
    for ovl, rdef in (("0", "2"), ("2", "2"), ("2", "0")):
        monkeypatch.setenv("PSD_OVERLAP", ovl)
        monkeypatch.setenv("PSD_CDEFER", "2")
        monkeypatch.setenv("PSD_RDEFER", rdef)
        engs.append(psd_amd.Engine(libpath=lib))
    for (n, p, lr, win) in [(100, 1, "R", None), (150, 7, "L", None), (200, 2, "R", None), (180, 3, "R", "10"), (260, 5, "R", None)]:
        A = pt.bench_factors(n, p, seed=7)
        pr = ref.pschur(A, lr)
        for eng in engs:
            ps = eng.pschur(A, lr)
            ok, err = pt.checkpsd(ps, A, thresh=100 * np.sqrt(n / 32))
            assert ok, (n, p, err.max())
            assert ps.stats.nsweeps == pr.stats.nsweeps
            for j in range(p):
                assert np.array_equal(ps.Z[j], pr.Z[j])
                assert np.array_equal(ps.Ts[j], pr.Ts[j])
            assert np.array_equal(ps.values, pr.values)


def test_schur_vector_updates_deferred(built, monkeypatch):
    """PSD_OVERLAP=2 (what the HIP build does by itself for n >= 1024): the Z role of a tick's bulk update is its own
    launch (psd_rq_apply_wl mode 4) behind the two H passes (mode 3) — on the GPU it runs on a second stream beside the
    next tick's chases.  Same invariants, and the same T and Z as the one-stream order to the last bit: the Z role
    commutes with everything else because only owner m's lists touch Z_m."""
    import os

    import numpy as np
    import psd_amd
    import psdtest as pt

    lib = os.path.join(os.path.dirname(os.path.abspath(__file__)), "hostsim", "_build", "libpsd_hostsim.so")
    monkeypatch.setenv("PSD_OVERLAP", "0")
    ref = psd_amd.Engine(libpath=lib)
    monkeypatch.setenv("PSD_OVERLAP", "2")
    eng = psd_amd.Engine(libpath=lib)
    for (n, p, lr) in [(100, 1, "R"), (150, 7, "L"), (200, 2, "R")]:
        A = pt.bench_factors(n, p, seed=11)
        ps = eng.pschur(A, lr)
        ok, err = pt.checkpsd(ps, A, thresh=100 * np.sqrt(n / 32))
        assert ok, (n, p, err.max())
        pr = ref.pschur(A, lr)
        for j in range(p):
            assert np.array_equal(ps.Z[j], pr.Z[j])
        assert np.array_equal(ps.values, pr.values)


def test_formq_blocked(built, monkeypatch):
    """compact-WY Q formation (T factors by the device code, the matrix-core step by its plain-loop stand-in) against
    the reflector-by-reflector form: n not a multiple of the block, p = 1 (H_1 alone: reflectors one row lower)"""
    import os

    import psd_amd

    lib = os.path.join(os.path.dirname(os.path.abspath(__file__)), "hostsim", "_build", "libpsd_hostsim.so")

    def make(env):
        for k, v in env.items():
            monkeypatch.setenv(k, v)
        return psd_amd.Engine(libpath=lib)

    ec.case_formq_blocked(make, [(64, 3, "R"), (97, 2, "L"), (130, 1, "R")])


def test_band_helper_same_decisions(built, monkeypatch):
    """psd_rq_band (the product bands of wide pending decisions computed in front of the chase launch, found by
    psd_rq_decide through bandinfo) against leaders that compute their bands themselves (PSD_BAND_HELPER=0): the same
    recurrence in the same order per row, so the same decisions: the same number of sweeps and the same eigenvalues to
    rounding in the serial simulation.  n >= 128 so that the helper has wide ranges to serve."""
    import os

    import numpy as np
    import psd_amd
    import psdtest as pt

    lib = os.path.join(os.path.dirname(os.path.abspath(__file__)), "hostsim", "_build", "libpsd_hostsim.so")
    monkeypatch.setenv("PSD_BAND_HELPER", "0")
    plain = psd_amd.Engine(libpath=lib)
    monkeypatch.setenv("PSD_BAND_HELPER", "1")
    helped = psd_amd.Engine(libpath=lib)
    for (n, p, lr) in [(200, 3, "R"), (260, 2, "L")]:
        A = pt.bench_factors(n, p, seed=31)
        ph = helped.pschur(A, lr)
        pp = plain.pschur(A, lr)
        # (the same sweeps; a leader that ends its launch for the helper shifts its range's ticks against the other
        #  ranges', so updates of different ranges may reach a shared off-diagonal element in the other order: equal to
        #  rounding, not to the bit)
        assert ph.stats.nsweeps == pp.stats.nsweeps
        assert pt.match_eigs(pp.values, ph.values) <= 1e-13 * np.abs(pp.values).max()
        for res in (ph, pp):
            ok, err = pt.checkpsd(res, A, thresh=100 * np.sqrt(n / 32))
            assert ok, (n, p, err.max())
