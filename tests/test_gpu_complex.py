"""GPU tier (MI355X): parity of the complex HIP path through the C ABI (same cases as the simulated tier, plus
larger sizes and the device-resident entry)."""
import os

import numpy as np
import pytest

import engine_cases as ec
import psdtest as pt

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("p", [1, 2, 5])
def test_zphessenberg(gpu_engine, p):
    ec.case_zphessenberg(gpu_engine, p)


@pytest.mark.parametrize("lr", ["R", "L"])
def test_zfull(gpu_engine, lr):
    ec.case_zfull(gpu_engine, lr)


@pytest.mark.parametrize("p", [1, 2, 3, 5])
def test_zhess_ut(gpu_engine, p):
    ec.case_zhess_ut(gpu_engine, p)


@pytest.mark.parametrize("p", [1, 5])
def test_zfast_paths(gpu_engine, p):
    ec.case_zfast_paths(gpu_engine, p)


@pytest.mark.parametrize("p", [5, 20])
def test_zexpsplit(gpu_engine, p):
    ec.case_zexpsplit(gpu_engine, p)


def test_zholes(gpu_engine):
    ec.case_zholes(gpu_engine)


def test_zedge(gpu_engine):
    ec.case_zedge(gpu_engine)


def test_zwindow_widths(gpu_engine):
    ec.case_zwindow_widths(gpu_engine, [(40, 4, 32), (36, 12, 24), (30, 22, 20), (28, 34, 16), (26, 70, 10)])


def test_config3_shape_reduced(gpu_engine):
    """The code path and window width (p = 64 -> W = 12) of BASELINE configs[2] at a reduced order n = 128 (the full
    n = 1024 case is tests/test_gpu_baseline_configs.py::test_config3_full_size): eigenvalues vs numpy's eigvals of the
    explicit product + invariants."""
    n, p = 128, 64
    As = pt.bench_factors(n, p, seed=1234 + 3, dtype=np.complex128)
    ps = gpu_engine.pschur(As, "R")
    assert ps.stats.window == ec.expected_window(p, 16)
    assert ps.stats.window == 12 or "PSD_WINDOW" in os.environ
    ok, err = pt.checkpsd(ps, As, thresh=100 * np.sqrt(n / 32))
    assert ok, err
    P = pt.product(As)
    assert pt.match_eigs(np.linalg.eigvals(P), ps.values) <= 1e-10 * np.linalg.norm(P, 2)


def test_zdevice_resident_entry(gpu_engine):
    import ctypes as C

    import torch

    import psd_amd

    n, p = 48, 6
    As = pt.bench_factors(n, p, seed=31, dtype=np.complex128)
    dA = torch.from_numpy(pt.pack(As, np.complex128)).to("cuda:0")
    dZ = torch.zeros_like(dA)
    torch.cuda.synchronize()
    alpha = np.zeros(n, dtype=np.complex128)
    beta = np.zeros(n)
    sc = np.zeros(n, dtype=np.int32)
    si, info, st = C.c_int(0), C.c_int(0), psd_amd.Stats()
    dp = C.POINTER(C.c_double)
    gpu_engine.lib.psd_z_pschur_dev(gpu_engine.ctx, n, p, C.c_void_p(dA.data_ptr()), b"R", 1, 1, 30,
                                    C.c_void_p(dZ.data_ptr()), alpha.view(np.float64).ctypes.data_as(dp),
                                    beta.ctypes.data_as(dp), sc.ctypes.data_as(C.POINTER(C.c_int32)), C.byref(si),
                                    C.byref(st), None, 0, C.byref(info))
    assert info.value == 0
    Ts = pt.unpack(dA.cpu().numpy())
    Zs = pt.unpack(dZ.cpu().numpy())
    ps = pt.GPSD([True] * p, Ts, Zs, alpha, beta, sc, "R", si.value)
    pt.gpschur_check(As, [True] * p, ps)


@pytest.mark.parametrize("p", [5, 1, 2])
@pytest.mark.parametrize("lr", ["L", "R"])
def test_zordschur_reference(gpu_engine, p, lr):
    ec.case_zordschur_reference(gpu_engine, p, lr)


def test_zordschur_windows(gpu_engine):
    ec.case_zordschur_windows(gpu_engine, [(48, 3), (44, 12), (40, 22), (36, 34), (30, 70)])


def test_zordschur_edge(gpu_engine):
    ec.case_zordschur_edge(gpu_engine)


def test_ordschur_alignments(gpu_engine):
    ec.case_ordschur_alignments(gpu_engine)


def test_ordschur_supplementary_z(gpu_engine):
    ec.case_ordschur_supplementary_z(gpu_engine)


def test_ztrains(gpu_engine):
    ec.case_ztrains(gpu_engine, [(150, 3, "R"), (120, 12, "L"), (256, 40, "R")])
