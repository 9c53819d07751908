"""GPU tier (MI355X): the size BASELINE.json quotes its target on — pschur!(A,:R), n = 1024, p = 64, Float64 — and the
paths only the bench used to touch.

  * accuracy at the headline size: device checkpsd (pinned to the numpy restatement of diagnostics.jl:190-263 by
    test_checkpsd, here also at n = 520) and eigenvalues within 1e-10 * ||prod A||_2 of LAPACK on the explicit product;
  * reproducibility: the same input twice gives bit-identical T, Z, eigenvalues and the same sweep count — the tick
    schedule of the slot scheduler does not depend on the order in which the workgroups of a launch happen to run
    (psd_rq_plan);
  * n = 1536, p = 8: wider than any long-train pipeline that fits the 63 cursor slots;
  * trains off at n = 512: the reference's one-shift-pair iteration, sweep for sweep against the CPU oracle
    (PSD.jl:471-888).
"""
import numpy as np
import pytest

import engine_cases as ec
import psdtest as pt

pytestmark = pytest.mark.gpu


def _accuracy(eng, ps, As, left=False):
    n = As[0].shape[0]
    ok, err = eng.checkpsd(ps, As, thresh=100 * np.sqrt(n / 32))
    assert ok, err
    P = pt.product(As, left)
    assert pt.match_eigs(np.linalg.eigvals(P), ps.values) <= 1e-10 * np.linalg.norm(P, 2)
    return err


def test_headline_1024x64_accuracy_and_reproducibility(gpu_engine):
    n, p = 1024, 64
    As = pt.bench_factors(n, p, seed=1234 + 2)  # (bench.py's input)
    ps = gpu_engine.pschur(As, "R")
    assert ps.stats.nsweeps > 0 and ps.stats.reserved > 0  # multishift trains ran
    _accuracy(gpu_engine, ps, As)
    ps2 = gpu_engine.pschur(As, "R")
    assert ps2.stats.nsweeps == ps.stats.nsweeps and ps2.stats.nlaunch_step == ps.stats.nlaunch_step
    assert np.array_equal(ps.values, ps2.values)
    for j in range(p):
        assert np.array_equal(ps.Ts[j], ps2.Ts[j]), j
        assert np.array_equal(ps.Z[j], ps2.Z[j]), j


def test_cfg2_reproducible(gpu_engine):
    n, p = 512, 16
    As = pt.bench_factors(n, p, seed=1234 + 2)
    a = gpu_engine.pschur(As, "L")
    b = gpu_engine.pschur(As, "L")
    assert a.stats.nsweeps == b.stats.nsweeps
    assert all(np.array_equal(x, y) for x, y in zip(a.Ts, b.Ts)) and all(np.array_equal(x, y) for x, y in zip(a.Z, b.Z))


def test_wider_than_the_slot_pipeline_1536x8(gpu_engine):
    n, p = 1536, 8
    As = pt.bench_factors(n, p, seed=77)
    ps = gpu_engine.pschur(As, "R")
    _accuracy(gpu_engine, ps, As)


def test_trains_off_sweep_parity_512(gpu_engine):
    n, p = 512, 6
    As = pt.bench_factors(n, p, seed=1234 + 2)
    po = pt.oracle_pschur(As, "R")
    m = gpu_engine.get_train()
    gpu_engine.set_train(0)
    try:
        pr = gpu_engine.pschur(As, "R")
    finally:
        gpu_engine.set_train(m)
    assert pr.stats.reserved == 0
    nref = int((po.sweeplog[:, 0] == 0).sum())
    # (the device contracts multiply-adds and refines reciprocals by Newton steps: the iteration drifts from the
    #  oracle's in the last bits, a few percent of the sweeps over a long run)
    assert abs(pr.stats.nsweeps - nref) <= 0.05 * nref + 5, (pr.stats.nsweeps, nref)
    Pn = np.linalg.norm(pt.product(As), 2)
    assert pt.match_eigs(po.values, pr.values) <= 1e-10 * Pn
    _accuracy(gpu_engine, pr, As)


def test_checkpsd_pinned_near_full_size(gpu_engine):
    """the verifier the full-size tests rely on, against the numpy restatement at an order off the tile grid"""
    ec.case_checkpsd(gpu_engine, [(520, 3, "R", "d")])


def test_pipe_form_above_1024(gpu_engine):
    """n > 1024 takes the 32-column-step instantiation of the chain kernel (`psd_hess2_link<32, 8>`), p >= 32 the
    overlapping launches: accuracy of the whole call at n = 1100, p = 32"""
    n, p = 1100, 32
    As = pt.bench_factors(n, p, seed=91)
    ps = gpu_engine.pschur(As, "R")
    _accuracy(gpu_engine, ps, As)
