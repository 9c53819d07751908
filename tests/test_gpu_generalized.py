"""GPU tier (MI355X): parity of the real signed periodic QZ (csrc/psd_rgz.h) through the C ABI: the same cases as
the simulated tier plus larger sizes (the full-size BASELINE configs[3], n = 512, p = 32, is
tests/test_gpu_baseline_configs.py::test_config4_full_size)."""
import numpy as np
import pytest

import engine_cases as ec
import psdtest as pt

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("p", [2, 3, 5])
def test_rg_hess_ut(gpu_engine, p):
    ec.case_rg_hess_ut(gpu_engine, p)


def test_rg_holes(gpu_engine):
    ec.case_rg_holes(gpu_engine)


def test_rg_windows(gpu_engine):
    ec.case_rg_windows(gpu_engine, [(40, 3, "alt"), (50, 6, "mix"), (36, 4, "true"), (33, 5, "neg"), (30, 22, "mix"),
                                    (70, 2, "alt"), (130, 8, "alt"), (100, 17, "mix")])


def test_rg_fast_paths(gpu_engine):
    ec.case_rg_fast_paths(gpu_engine)


def test_rg_edge(gpu_engine):
    ec.case_rg_edge(gpu_engine)


def test_rg_alternating_256x8_hess(gpu_engine):
    """n = 256, p = 8, alternating signature (a reduced shape of BASELINE configs[3]) from a Hessenberg-triangular start."""
    n, p = 256, 8
    S = [True] + [bool(q % 2 == 0) for q in range(1, p)]
    A = ec.rg_hess_ut(n, p, 4242, shift=4.0)
    ps = gpu_engine.gpschur_hess_(A[0].copy(order="F"), [a.copy(order="F") for a in A[1:]], S)
    pt.rgpschur_check(A, S, ps, tol=100 * np.sqrt(n / 32), lam_check=False)
    po = pt.oracle_gpschur_hess(A[0], A[1:], S)
    assert po.info == 0
    assert pt.match_eigs(po.values, ps.values) < 1e-10 * abs(po.values).max()


@pytest.mark.parametrize("p", [2, 5])
def test_rg_phessenberg(gpu_engine, p):
    ec.case_rg_phessenberg(gpu_engine, p)


@pytest.mark.parametrize("lr", ["R", "L"])
def test_rg_full(gpu_engine, lr):
    ec.case_rg_full(gpu_engine, lr)


def test_rg_full_sizes(gpu_engine):
    ec.case_rg_full_sizes(gpu_engine, [(24, 3, "R", "mix"), (40, 6, "L", "mix"), (30, 4, "R", "true"),
                                       (33, 21, "L", "mix"), (150, 5, "R", "mix"), (128, 8, "L", "mix")])


def test_rg_trains(gpu_engine):
    ec.case_rg_trains(gpu_engine, [(120, 3, "R", "mix"), (150, 6, "L", "mix"), (128, 4, "R", "true"),
                                   (256, 8, "R", "mix"), (200, 40, "L", "mix")])


def test_rg_alternating_256x8_full(gpu_engine):
    """pschur!(A, S, :R), n = 256, p = 8, Float64, alternating signature, full matrices (reduced shape of BASELINE
    configs[3]; oracle eigenvalues at 1e-10 max|lambda|)."""
    n, p = 256, 8
    S = [True] + [bool(q % 2 == 0) for q in range(1, p)]
    A = pt.bench_factors(n, p, seed=4)
    ps = gpu_engine.pschur_([a.copy(order="F") for a in A], "R", S=S)
    pt.rgpschur_check(A, S, ps, tol=100 * np.sqrt(n / 32), qtol=10 * np.sqrt(n / 32), lam_check=False)
    po = pt.oracle_gpschur(A, S, "R")
    assert po.info == 0
    assert pt.match_eigs(po.values, ps.values) < 1e-10 * abs(po.values).max()


# ---- complex signed path (csrc/psd_zgz.h) ----
@pytest.mark.parametrize("p", [2, 3, 5])
def test_zg_hess_ut(gpu_engine, p):
    ec.case_zg_hess_ut(gpu_engine, p)


def test_zg_holes(gpu_engine):
    ec.case_zg_holes(gpu_engine)


def test_zg_windows(gpu_engine):
    ec.case_zg_windows(gpu_engine, [(40, 3, "alt"), (45, 6, "mix"), (33, 5, "neg"), (30, 22, "mix"), (130, 8, "alt")])


@pytest.mark.parametrize("p", [2, 5])
def test_zg_phessenberg(gpu_engine, p):
    ec.case_zg_phessenberg(gpu_engine, p)


@pytest.mark.parametrize("lr", ["R", "L"])
def test_zg_full(gpu_engine, lr):
    ec.case_zg_full(gpu_engine, lr)


# ---- ordschur! of a GeneralizedPeriodicSchur (csrc/psd_zgord.h) ----
@pytest.mark.parametrize("cplx", [False, True])
@pytest.mark.parametrize("lr", ["L", "R"])
def test_gordschur_reference(gpu_engine, cplx, lr):
    ec.case_gordschur_reference(gpu_engine, cplx, lr)


def test_gordschur_windows(gpu_engine):
    ec.case_gordschur_windows(gpu_engine, [(12, 3, True, "L"), (12, 4, False, "R"), (40, 3, True, "R"),
                                           (36, 21, True, "L"), (90, 6, True, "L"), (64, 5, False, "R")])


def test_gordschur_pairs_reference(gpu_engine):
    ec.case_gordschur_pairs_reference(gpu_engine)


def test_gordschur_pairs_random(gpu_engine):
    ec.case_gordschur_pairs_random(gpu_engine, [(10, 3, "L", 1), (14, 4, "R", 2), (16, 5, "L", 3), (40, 3, "R", 4),
                                                (30, 21, "L", 5), (96, 6, "R", 6), (128, 8, "L", 7)])


def test_gpschur_pairs(gpu_engine):
    ec.case_gpschur_pairs(gpu_engine)


def test_hess_pipeline_vs_serial(gpu_engine):
    ec.case_hess_pipeline_vs_serial(gpu_engine)


def test_zg_trains(gpu_engine):
    ec.case_zg_trains(gpu_engine, [(100, 3, "R"), (120, 6, "L"), (256, 8, "R"), (200, 40, "L")])
