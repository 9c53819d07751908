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
