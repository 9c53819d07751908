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


def _zhess_vs_oracle(eng, n, p, seed):
    A = pt.bench_factors(n, p, seed=seed, dtype=np.complex128)
    W = [a.copy(order="F") for a in A]
    Hs, tau, _ = eng.phessenberg_(W)
    Ho, Qo, packed, tauo = pt.oracle_zphessenberg(A)
    assert np.all(np.tril(Hs[0], -2) == 0)
    for j in range(p):
        assert np.linalg.norm(W[j] - packed[j]) < 1e-11 * max(np.linalg.norm(packed[j]), 1.0), (j,)
        assert np.allclose(tau[j], tauo[j], rtol=0, atol=1e-11)
        Ax = Qo[j] @ Hs[j] @ Qo[(j + 1) % p].conj().T
        assert np.linalg.norm(A[j] - Ax) < 1e-10 * max(np.linalg.norm(A[j]), 1.0)
    # the subdiagonal of H_1 and the diagonals of the triangular factors are real (householder.jl:113-114)
    assert np.abs(np.diag(Hs[0], -1).imag).max() == 0.0
    for j in range(1, p):
        assert np.abs(np.diag(Hs[j]).imag)[:-1].max(initial=0.0) == 0.0


@pytest.mark.parametrize("n,p", [(2, 3), (3, 4), (7, 3), (9, 5), (33, 3), (64, 7), (130, 4), (257, 5), (300, 16), (520, 3)])
def test_zphessenberg_lookahead_vs_oracle(gpu_engine, n, p):
    """The look-ahead reduction for ComplexF64 (csrc/psd_zhess2.h: one launch per chain link) against the oracle's
    packed Householder storage and tau (PSD.jl:213-259 with householder.jl:110-156): odd orders, orders off the 4-row
    strips and off the 128-column steps, the shortest period it serves (p = 3), the one-row last reflector of H_1 that
    only turns the subdiagonal real."""
    _zhess_vs_oracle(gpu_engine, n, p, 270 + n + p)


@pytest.mark.parametrize("pipe", ["2", "0"])
@pytest.mark.parametrize("n,p,K", [(33, 8, 4), (130, 9, 4), (257, 16, 8), (300, 40, 16), (64, 70, 16), (96, 150, 16), (520, 8, 4)])
def test_zphessenberg_two_stream_vs_oracle(monkeypatch, n, p, K, pipe):
    """multi-stream forms (see test_phessenberg_two_stream_vs_oracle) of the complex reduction, forced on small problems:
    pipe "2" = overlapping chain launches with the column as self-validating records, "0" = back to back"""
    import torch

    torch.cuda.init()
    import psd_amd

    monkeypatch.setenv("PSD_HESS_ASYNC", str(K))
    monkeypatch.setenv("PSD_H2_PIPE", pipe)
    _zhess_vs_oracle(psd_amd.Engine(device=0), n, p, 370 + n + p)


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


def test_zordschur_pipelined(monkeypatch):
    """pipelined ordschur! drivers of the ComplexF64 engines (standard and signed) against the serial ones and the
    oracle, through the C ABI"""
    import psd_amd

    def make(env):
        for k, v in env.items():
            monkeypatch.setenv(k, v)
        return psd_amd.Engine()

    ec.case_zordschur_pipelined(make, [(130, 2, "R", 0.5), (150, 4, "L", 0.3), (300, 16, "R", 0.4), (256, 40, "L", 0.5)],
                                [(130, 3, "R"), (140, 4, "L"), (200, 12, "R")])
    ec.case_ordschur_pipelined_failure(make)


def test_ordschur_alignments(gpu_engine):
    ec.case_ordschur_alignments(gpu_engine)


def test_ordschur_supplementary_z(gpu_engine):
    ec.case_ordschur_supplementary_z(gpu_engine)


def test_ztrains(gpu_engine):
    ec.case_ztrains(gpu_engine, [(150, 3, "R"), (120, 12, "L"), (256, 40, "R")])


def test_zformq_blocked(monkeypatch):
    """eight reflectors per pass over the Q_j against one launch per reflector; orders off the 64-row lanes and the
    8-reflector blocks, both factor-1 and factor-j row offsets, both kernel instantiations (n <= 512, n <= 1024)"""
    import torch

    torch.cuda.init()
    import psd_amd

    def make(env):
        for k, v in env.items():
            monkeypatch.setenv(k, v)
        return psd_amd.Engine(device=0)

    ec.case_zformq_blocked(make, [(64, 3, "R"), (97, 2, "L"), (130, 1, "R"), (257, 5, "R"), (520, 3, "L")])
