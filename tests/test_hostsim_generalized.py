"""CPU tier: the device state machine of the real signed periodic QZ (csrc/psd_rgz.h), run through the TEST-ONLY
serial simulation (tests/hostsim) and checked with the reference's invariants and against the oracle."""
import pytest

import engine_cases as ec


@pytest.mark.parametrize("p", [2, 3, 5])
def test_rg_hess_ut(sim_engine, p):
    ec.case_rg_hess_ut(sim_engine, p)


def test_rg_holes(sim_engine):
    ec.case_rg_holes(sim_engine)


def test_rg_windows(sim_engine):
    ec.case_rg_windows(sim_engine, [(40, 3, "alt"), (50, 6, "mix"), (36, 4, "true"), (33, 5, "neg"), (30, 22, "mix"),
                                    (70, 2, "alt")])


def test_rg_fast_paths(sim_engine):
    ec.case_rg_fast_paths(sim_engine)


def test_rg_edge(sim_engine):
    ec.case_rg_edge(sim_engine)


@pytest.mark.parametrize("p", [2, 5])
def test_rg_phessenberg(sim_engine, p):
    ec.case_rg_phessenberg(sim_engine, p)


@pytest.mark.parametrize("lr", ["R", "L"])
def test_rg_full(sim_engine, lr):
    ec.case_rg_full(sim_engine, lr)


def test_rg_full_sizes(sim_engine):
    ec.case_rg_full_sizes(sim_engine, [(24, 3, "R", "mix"), (40, 6, "L", "mix"), (30, 4, "R", "true"), (33, 21, "L", "mix")])


def test_rg_trains(sim_engine):
    ec.case_rg_trains(sim_engine, [(120, 3, "R", "mix"), (150, 6, "L", "mix"), (128, 4, "R", "true")])


# ---- complex signed path (csrc/psd_zgz.h) ----
@pytest.mark.parametrize("p", [2, 3, 5])
def test_zg_hess_ut(sim_engine, p):
    ec.case_zg_hess_ut(sim_engine, p)


def test_zg_holes(sim_engine):
    ec.case_zg_holes(sim_engine)


def test_zg_windows(sim_engine):
    ec.case_zg_windows(sim_engine, [(40, 3, "alt"), (45, 6, "mix"), (33, 5, "neg"), (30, 22, "mix")])


@pytest.mark.parametrize("p", [2, 5])
def test_zg_phessenberg(sim_engine, p):
    ec.case_zg_phessenberg(sim_engine, p)


@pytest.mark.parametrize("lr", ["R", "L"])
def test_zg_full(sim_engine, lr):
    ec.case_zg_full(sim_engine, lr)


# ---- ordschur! of a GeneralizedPeriodicSchur (csrc/psd_zgord.h) ----
@pytest.mark.parametrize("cplx", [False, True])
@pytest.mark.parametrize("lr", ["L", "R"])
def test_gordschur_reference(sim_engine, cplx, lr):
    ec.case_gordschur_reference(sim_engine, cplx, lr)


def test_gordschur_windows(sim_engine):
    ec.case_gordschur_windows(sim_engine, [(12, 3, True, "L"), (12, 4, False, "R"), (40, 3, True, "R"), (36, 21, True, "L")])


def test_gordschur_pairs_reference(sim_engine):
    ec.case_gordschur_pairs_reference(sim_engine)


def test_gordschur_pairs_random(sim_engine):
    ec.case_gordschur_pairs_random(sim_engine, [(10, 3, "L", 1), (14, 4, "R", 2), (16, 5, "L", 3), (40, 3, "R", 4), (30, 21, "L", 5)])


def test_gpschur_pairs(sim_engine):
    ec.case_gpschur_pairs(sim_engine)


def test_hess_pipeline_vs_serial(sim_engine):
    ec.case_hess_pipeline_vs_serial(sim_engine)


def test_zg_trains(sim_engine):
    ec.case_zg_trains(sim_engine, [(100, 3, "R"), (120, 6, "L")])
