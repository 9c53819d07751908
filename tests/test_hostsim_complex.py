"""CPU tier: complex device path executed by the TEST-ONLY simulation build (see test_hostsim_real.py)."""
import pytest

import engine_cases as ec


@pytest.mark.parametrize("p", [1, 2, 5])
def test_zphessenberg(sim_engine, p):
    ec.case_zphessenberg(sim_engine, p)


@pytest.mark.parametrize("lr", ["R", "L"])
def test_zfull(sim_engine, lr):
    ec.case_zfull(sim_engine, lr)


@pytest.mark.parametrize("p", [1, 2, 3, 5])
def test_zhess_ut(sim_engine, p):
    ec.case_zhess_ut(sim_engine, p)


@pytest.mark.parametrize("p", [1, 5])
def test_zfast_paths(sim_engine, p):
    ec.case_zfast_paths(sim_engine, p)


@pytest.mark.parametrize("p", [5, 20])
def test_zexpsplit(sim_engine, p):
    ec.case_zexpsplit(sim_engine, p)


def test_zholes(sim_engine):
    ec.case_zholes(sim_engine)


def test_zedge(sim_engine):
    ec.case_zedge(sim_engine)


def test_zwindow_widths(sim_engine):
    ec.case_zwindow_widths(sim_engine, [(40, 4, 32), (36, 12, 24), (30, 22, 20), (28, 34, 16), (26, 70, 10)])


@pytest.mark.parametrize("p", [5, 1, 2])
@pytest.mark.parametrize("lr", ["L", "R"])
def test_zordschur_reference(sim_engine, p, lr):
    ec.case_zordschur_reference(sim_engine, p, lr)


def test_zordschur_windows(sim_engine):
    ec.case_zordschur_windows(sim_engine, [(48, 3), (44, 12), (40, 22), (36, 34), (30, 70)])


def test_zordschur_edge(sim_engine):
    ec.case_zordschur_edge(sim_engine)


def test_ordschur_alignments(sim_engine):
    ec.case_ordschur_alignments(sim_engine)


def test_ordschur_supplementary_z(sim_engine):
    ec.case_ordschur_supplementary_z(sim_engine)


def test_ztrains(sim_engine):
    ec.case_ztrains(sim_engine, [(150, 3, "R"), (120, 12, "L")])


def test_zcolumn_roles_deferred(built, monkeypatch):
    """ComplexF64 engine, PSD_OVERLAP=2 with PSD_ZCDEFER=1 (a selector; off by default — it measured slower): the rows of a sweep
    window's column role that lie more than psd_cdefer_edge above it are their own launch (psd_zq_apply_wl pass 5) — on the
    GPU on the second stream beside the next tick's chases; the serial simulation runs it at the latest point the streams
    allow, BEHIND the next tick's chases (the last window of a sweep keeps its column role whole: the check behind it
    runs in the next launch).  Every element still sees
    the same sequence of operations: T, Z and the eigenvalues are those of the undeferred order to the last bit.  Cases with
    the reference's hole fixtures (Case II, zero-shift passes: those windows keep their column roles whole) and the
    exponentially split example run through the deferring engine as well."""
    import os

    import numpy as np
    import psd_amd
    import psdtest as pt

    lib = os.path.join(os.path.dirname(os.path.abspath(__file__)), "hostsim", "_build", "libpsd_hostsim.so")
    monkeypatch.setenv("PSD_OVERLAP", "2")
    monkeypatch.setenv("PSD_ZCDEFER", "0")
    ref = psd_amd.Engine(libpath=lib)
    monkeypatch.setenv("PSD_ZCDEFER", "1")
    eng = psd_amd.Engine(libpath=lib)
    for (n, p, lr) in [(100, 1, "R"), (150, 4, "L"), (200, 2, "R"), (130, 22, "R")]:
        A = pt.bench_factors(n, p, seed=17 + n, dtype=np.complex128)
        pr = ref.pschur(A, lr)
        ps = eng.pschur(A, lr)
        ok, err = pt.checkpsd(ps, A, thresh=100 * np.sqrt(n / 32))
        assert ok, (n, p, err.max())
        assert ps.stats.nsweeps == pr.stats.nsweeps
        for j in range(p):
            assert np.array_equal(ps.Z[j], pr.Z[j]), (n, p, j)
            assert np.array_equal(ps.Ts[j], pr.Ts[j]), (n, p, j)
        assert np.array_equal(ps.values, pr.values)
    ec.case_zholes(eng)
    ec.case_zexpsplit(eng, 20)
    ec.case_ztrains(eng, [(150, 3, "R")])
