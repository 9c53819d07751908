"""GPU tier (MI355X): parity of the HIP path, called through the C ABI, against the CPU oracle, the committed
goldens and the reference's own invariants.  Tolerances: eigenvalues |dlam| <= 1e-10 * ||prod A||_2 (north_star)
and the reference's 1000*eps*|lam|max at its own test sizes; factors by `checkpsd` / `pschur_check`."""
import numpy as np
import pytest

import engine_cases as ec
import psdtest as pt

pytestmark = pytest.mark.gpu


def test_native_library_is_hip(gpu_engine):
    assert "hip-gfx950" in gpu_engine.version()


@pytest.mark.parametrize("p", [1, 2, 5])
def test_phessenberg(gpu_engine, p):
    ec.case_phessenberg(gpu_engine, p)


@pytest.mark.parametrize("p", [1, 2, 3, 5])
def test_hess_ut(gpu_engine, p):
    ec.case_hess_ut(gpu_engine, p)


@pytest.mark.parametrize("p", [5, 20])
def test_expsplit(gpu_engine, golden, p):
    ec.case_expsplit(gpu_engine, golden, p)


@pytest.mark.parametrize("p", [1, 2, 3, 5])
def test_full_small(gpu_engine, golden, p):
    ec.case_full_small(gpu_engine, golden, p)


@pytest.mark.parametrize("p", [1, 5])
def test_fast_paths(gpu_engine, p):
    ec.case_fast_paths(gpu_engine, p)


def test_config1(gpu_engine, golden):
    ec.case_config1(gpu_engine, golden)


def test_rq_cleanup(gpu_engine):
    ec.case_rq_cleanup(gpu_engine)
    ec.case_rq_cleanup_windows(gpu_engine)


def test_edge(gpu_engine):
    ec.case_edge(gpu_engine)


def test_window_widths(gpu_engine):
    ec.case_window_widths(gpu_engine, [(72, 6, 32), (60, 20, 24), (48, 40, 20), (44, 60, 16), (40, 80, 12)])


def test_pschur_hess(gpu_engine):
    ec.case_pschur_hess(gpu_engine)


def test_phessenberg_medium_vs_oracle(gpu_engine):
    n, p = 96, 6
    A = pt.bench_factors(n, p, seed=3)
    W = [a.copy(order="F") for a in A]
    Hs, tau, _ = gpu_engine.phessenberg_(W)
    Ho, Qo, packed, tauo = pt.oracle_phessenberg(A)
    for j in range(p):
        assert np.linalg.norm(W[j] - packed[j]) < 1e-11 * np.linalg.norm(packed[j])
        Ax = Qo[j] @ Hs[j] @ Qo[(j + 1) % p].T
        assert np.linalg.norm(A[j] - Ax) < 100 * pt.EPS * n


def test_config2_full_size(gpu_engine):
    """BASELINE config 2: pschur!(A,:R) n=512 p=16 Float64, eigenvalues + Schur vectors.  The oracle needs ~20 s
    here, so eigenvalues are compared with numpy's eigvals of the explicit product (what the reference's tests
    use) and the factors through the size-independent invariants."""
    n, p = 512, 16
    As = pt.bench_factors(n, p, seed=1234 + 2)
    ps = gpu_engine.pschur(As, "R")
    ok, err = pt.checkpsd(ps, As, thresh=100 * np.sqrt(n / 32))
    assert ok, err
    P = pt.product(As)
    lam = np.linalg.eigvals(P)
    assert pt.match_eigs(lam, ps.values) <= 1e-10 * np.linalg.norm(P, 2)
    # eigenvalues are consistent with the diagonal blocks of the returned factors
    d = np.ones(n)
    for T in ps.Ts:
        d = d * np.diag(T)
    real = ps.values.imag == 0
    assert np.allclose(d[real], ps.values.real[real], rtol=1e-9, atol=1e-12 * abs(lam).max())
    assert ps.stats.nsweeps > 0 and ps.stats.bytes_sweeps > 0


def test_medium_vs_oracle_eigenvalues(gpu_engine):
    n, p = 192, 16
    As = pt.bench_factors(n, p, seed=11)
    ps = gpu_engine.pschur(As, "L")
    po = pt.oracle_pschur(As, "L")
    Pn = np.linalg.norm(pt.product(As, True), 2)
    assert pt.match_eigs(po.values, ps.values) <= 1e-10 * Pn
    ok, err = pt.checkpsd(ps, As, thresh=100 * np.sqrt(n / 32))
    assert ok, err
    # in the reference's one-shift-one-sweep mode the sweep counts track the oracle's (same algorithm, same shifts) to
    # within rounding-induced drift; the default mode (multishift trains) takes another iteration path
    m = gpu_engine.get_train()
    gpu_engine.set_train(0)
    try:
        pr = gpu_engine.pschur(As, "L")
    finally:
        gpu_engine.set_train(m)
    assert pr.stats.reserved == 0
    assert abs(pr.stats.nsweeps - (po.sweeplog[:, 0] == 0).sum()) <= 0.1 * pr.stats.nsweeps + 5
    assert pt.match_eigs(po.values, pr.values) <= 1e-10 * Pn
    assert ps.stats.reserved > 0 or m < 2  # trains ran in the default mode


def test_device_resident_entry(gpu_engine):
    """psd_d_pschur_dev: operands already in HBM (torch is only the allocator here)."""
    import torch

    n, p = 64, 8
    As = pt.bench_factors(n, p, seed=21)
    dA = torch.from_numpy(pt.pack(As)).to("cuda:0")
    dZ = torch.zeros_like(dA)
    torch.cuda.synchronize()
    lam, si, st, log = gpu_engine.pschur_dev(dA.data_ptr(), n, p, "R", dZ_ptr=dZ.data_ptr())
    Ts = pt.unpack(dA.cpu().numpy())
    Zs = pt.unpack(dZ.cpu().numpy())
    ps = pt.PSD(Ts, Zs, lam, "R", si)
    pt.pschur_check(As, ps, tol=64)
    assert st.ms_iter > 0 and st.nwindows > 0


@pytest.mark.parametrize("p", [5, 1])
@pytest.mark.parametrize("lr", ["L", "R"])
def test_rordschur_reference_real(gpu_engine, p, lr):
    ec.case_rordschur_reference_real(gpu_engine, p, lr)


@pytest.mark.parametrize("p", [5, 1])
@pytest.mark.parametrize("selset", [[1, 2, 5], [1, 3, 4], [1, 2, 6, 7]])
def test_rordschur_pairs(gpu_engine, p, selset):
    ec.case_rordschur_pairs(gpu_engine, p, selset)


def test_rordschur_windows(gpu_engine):
    ec.case_rordschur_windows(gpu_engine, [(48, 3), (44, 12), (40, 22), (36, 34), (30, 70)])


def test_rordschur_edge(gpu_engine):
    ec.case_rordschur_edge(gpu_engine)


def test_rordschur_pipelined(monkeypatch):
    """pipelined ordschur! driver against the serial one and the oracle, through the C ABI (PSD_ORD_PIPE is a path
    selector of the product library: both drivers give valid results)"""
    import psd_amd

    def make(env):
        for k, v in env.items():
            monkeypatch.setenv(k, v)
        return psd_amd.Engine()

    ec.case_rordschur_pipelined(make, [(140, 2, "R", 0.5), (160, 5, "L", 0.25), (300, 8, "R", 0.3), (256, 40, "L", 0.5)])


def test_rphessenberg(gpu_engine):
    ec.case_rphessenberg(gpu_engine)


def test_trains(gpu_engine):
    ec.case_trains(gpu_engine, [(150, 3, "R"), (130, 1, "L"), (120, 5, "L"), (384, 8, "R"), (200, 40, "R")])


def test_checkpsd(gpu_engine):
    """device checkpsd (matrix cores) against the numpy restatement of diagnostics.jl:190-263; sizes off the 64-tile
    grid and off the K-step of 16 on purpose"""
    ec.case_checkpsd(gpu_engine, [(12, 3, "R", "d"), (70, 4, "L", "d"), (129, 2, "R", "z"), (64, 3, "L", "z"),
                                  (100, 4, "R", "dg"), (200, 5, "R", "d")])


@pytest.mark.parametrize("n,p", [(2, 3), (3, 4), (7, 3), (9, 5), (33, 3), (64, 7), (130, 4), (257, 5), (300, 16), (520, 3)])
def test_phessenberg_lookahead_vs_oracle(gpu_engine, n, p):
    """The look-ahead reduction (csrc/psd_hess2.h: one launch per chain link, panel updates one launch behind) against
    the oracle's packed Householder storage and tau (PSD.jl:213-259): odd orders, orders off the 8-row strips and off
    the 64-column steps, the shortest period it serves (p = 3)."""
    A = pt.bench_factors(n, p, seed=70 + n + p)
    W = [a.copy(order="F") for a in A]
    Hs, tau, _ = gpu_engine.phessenberg_(W)
    Ho, Qo, packed, tauo = pt.oracle_phessenberg(A)
    for j in range(p):
        assert np.linalg.norm(W[j] - packed[j]) < 1e-11 * max(np.linalg.norm(packed[j]), 1.0), (j,)
        assert np.allclose(tau[j], tauo[j], rtol=0, atol=1e-11)
        Ax = Qo[j] @ Hs[j] @ Qo[(j + 1) % p].T
        assert np.linalg.norm(A[j] - Ax) < 1e-10 * max(np.linalg.norm(A[j]), 1.0)


def test_eigvecs(gpu_engine):
    """eigvecs(ps, select; shifted) (src/vectors.jl:25-138) on top of the device ordschur!: the reference's own test
    (test/vectors.jl) plus conjugate pairs and the right orientation"""
    ec.case_eigvecs(gpu_engine)


@pytest.mark.parametrize("n,p", [(256, 8), (512, 16)])
def test_residual_growth_oracle_vs_device(gpu_engine, n, p):
    """The residual gate of the full-size tests is 100 sqrt(n/32) eps ||A||_1 (the reference's tol 100 belongs to its
    n = 32 tests).  This prints the REFERENCE algorithm's own residual (CPU oracle, IEEE division, no FMA contraction)
    beside the device's (contracted FMAs, Newton-refined rcp/rsq, multishift trains) on the same input, and bounds the
    device by twice the oracle's figure: the growth with n belongs to the algorithm, not to the fast arithmetic."""
    A = pt.bench_factors(n, p, seed=1234 + 2)
    ps = gpu_engine.pschur(A, "R")
    ok, err = gpu_engine.checkpsd(ps, A, thresh=100 * np.sqrt(n / 32))
    po = pt.oracle_pschur(A, "R")
    oko, erro = pt.checkpsd(po, A, thresh=100 * np.sqrt(n / 32))
    print(f"\\nresidual / (eps ||A||_1), n={n} p={p}: device max {err.max():.1f} (mean {err.mean():.1f}), "
          f"oracle max {erro.max():.1f} (mean {erro.mean():.1f}), reference tol at n=32: 100, gate here: "
          f"{100 * np.sqrt(n / 32):.0f}")
    assert ok and oko
    assert erro.max() > 100 * 0.5 or n < 256  # the reference algorithm itself is beyond half its n = 32 tolerance here
    assert err.max() <= 2.0 * erro.max() + 20.0


def test_pschur_hess_batch(gpu_engine):
    """many small Hessenberg-triangular problems in one call on the slot scheduler"""
    ec.case_pschur_hess_batch(gpu_engine, [(3, 10, 2), (16, 40, 4), (32, 24, 3), (6, 64, 8)])


def test_pschur_hess_batch_one_fails(gpu_engine):
    """a problem of a batch that exhausts its sweep budget ends alone; the others are complete"""
    ec.case_pschur_hess_batch_one_fails(gpu_engine)


def test_formq_blocked(monkeypatch):
    """compact-WY Q formation on the matrix cores (csrc/psd_formq2.h) against the reflector-by-reflector kernel:
    identical T and eigenvalues, Z equal to rounding, orthogonal to 10 eps n; sizes off the 32 / 64 tiling"""
    import torch

    torch.cuda.init()
    import psd_amd

    def make(env):
        for k, v in env.items():
            monkeypatch.setenv(k, v)
        return psd_amd.Engine(device=0)

    ec.case_formq_blocked(make, [(64, 3, "R"), (97, 2, "L"), (130, 1, "R"), (257, 5, "R"), (512, 4, "L")])


@pytest.mark.parametrize("pipe", ["2", "0"])
@pytest.mark.parametrize("n,p,K", [(33, 8, 4), (130, 9, 4), (257, 16, 8), (300, 40, 16), (64, 70, 16), (96, 150, 16)])
def test_phessenberg_two_stream_vs_oracle(monkeypatch, n, p, K, pipe):
    """The multi-stream forms of the look-ahead reduction — pipe "2": consecutive chain launches overlap on two streams
    and hand the staged column over as self-validating records; "0": back to back on one stream.
    The two-stream form of the look-ahead reduction (chain launches alone on the main stream, the panel updates of K
    consecutive links as one launch on the second stream, awaited by event p - 1 links later; by itself only for
    p >= 32, n >= 512) forced on small problems: periods that are not multiples of K, a last batch that is only the
    drain position, p >= 9 K (K is raised), against the oracle's packed storage and tau."""
    import torch

    torch.cuda.init()
    import psd_amd

    monkeypatch.setenv("PSD_HESS_ASYNC", str(K))
    monkeypatch.setenv("PSD_H2_PIPE", pipe)  # ("2": overlapping chain launches although the session's engine is alive too)
    eng = psd_amd.Engine(device=0)
    A = pt.bench_factors(n, p, seed=170 + n + p)
    W = [a.copy(order="F") for a in A]
    Hs, tau, _ = eng.phessenberg_(W)
    Ho, Qo, packed, tauo = pt.oracle_phessenberg(A)
    for j in range(p):
        assert np.linalg.norm(W[j] - packed[j]) < 1e-11 * max(np.linalg.norm(packed[j]), 1.0), (j,)
        assert np.allclose(tau[j], tauo[j], rtol=0, atol=1e-11)


@pytest.mark.parametrize("env", [
    {"PSD_BAND_HELPER": "0"},
    {"PSD_OVERLAP": "0"},
    {"PSD_OVERLAP": "2"},
    {"PSD_OVERLAP": "2", "PSD_CDEFER": "2"},
    {"PSD_CDEFER": "0", "PSD_OVERLAP": "2"},
    {"PSD_C3": "0"},
    {"PSD_C3": "0", "PSD_C2": "0"},
    {"PSD_HESS_ASYNC": "4"},
    {"PSD_HESS_LOOKAHEAD": "0", "PSD_FORMQ_BLOCKED": "0"},
    {"PSD_MB": "0"},
    {"PSD_OVERLAP_CUS": "0", "PSD_HESS_CUS": "0", "PSD_OVERLAP": "2", "PSD_HESS_ASYNC": "4", "_diag": "1"},
], ids=lambda e: ",".join(f"{k}={v}" for k, v in e.items()))
def test_pschur_switches(monkeypatch, env):
    """Every path selector of the real engine still gives a decomposition: the default-on pieces turned off one at a time
    (band helper, second-stream Schur vectors and column roles, scan chase, two-wave chase, two-stream Hessenberg,
    look-ahead, blocked Q, multi-block scheduler), the size-gated ones forced on a small problem (PSD_OVERLAP=2,
    PSD_CDEFER=2, PSD_HESS_ASYNC=4) and — in the diagnostic build, where such knobs exist — the unmasked second streams.
    Invariants by checkpsd, eigenvalues against LAPACK on the explicit product."""
    import torch

    torch.cuda.init()
    import psd_amd

    env = dict(env)
    diag = env.pop("_diag", None)
    for k, v in env.items():
        monkeypatch.setenv(k, v)
    eng = psd_amd.Engine(device=0, libpath=psd_amd.DIAG_LIB_PATH if diag else None)
    for (n, p, lr) in [(260, 9, "R"), (150, 3, "L")]:
        A = pt.bench_factors(n, p, seed=400 + n)
        ps = eng.pschur(A, lr)
        ok, err = eng.checkpsd(ps, A, thresh=100 * np.sqrt(n / 32))
        assert ok, (env, n, p, err.max())
        P = pt.product(A, left=(lr == "L"))
        assert pt.match_eigs(np.linalg.eigvals(P), ps.values) <= 1e-10 * np.linalg.norm(P, 2)


def test_pipe_form_gives_up_loudly(monkeypatch):
    """A hand-over that never validates (test hook: the records of one link carry a wrong tag) must end in a runtime error
    within the bounded wait, not in a hang and not in silently wrong factors; the engine works again afterwards."""
    import time

    import torch

    torch.cuda.init()
    import psd_amd

    monkeypatch.setenv("PSD_HESS_ASYNC", "4")
    monkeypatch.setenv("PSD_H2_PIPE", "2")
    monkeypatch.setenv("PSD_H2_FAULT", "200")
    bad = psd_amd.Engine(device=0, libpath=psd_amd.DIAG_LIB_PATH)  # (the hook exists in the diagnostic build only)
    A = pt.bench_factors(96, 12, seed=5)
    t0 = time.time()
    with pytest.raises(RuntimeError):
        bad.phessenberg_([a.copy(order="F") for a in A])
    assert time.time() - t0 < 20.0
    monkeypatch.delenv("PSD_H2_FAULT")
    good = psd_amd.Engine(device=0)
    W = [a.copy(order="F") for a in A]
    Hs, tau, _ = good.phessenberg_(W)
    Ho, Qo, packed, tauo = pt.oracle_phessenberg(A)
    for j in range(len(A)):
        assert np.linalg.norm(W[j] - packed[j]) < 1e-11 * max(np.linalg.norm(packed[j]), 1.0)


def test_2x2_standardisation_guard(gpu_engine):
    """Real eigenvalue pairs many orders of magnitude apart (eps = 0.8, long periods, tiny orders): the rotation that the
    reference takes from the explicitly formed 2 x 2 product (PSD.jl:900-1054) can leave a subdiagonal entry in T1 far
    above rounding level, which :1066-1073 then zeroes.  Round 3's records hold such inputs — the seed-777 fuzz case
    (n = 7, p = 39: residual 5.4e5 eps on the device while the CPU oracle passed) and 14 of the 800 problems of
    tools/micro/repro_scan.py, 10 of which the CPU oracle fails as well.  With the guard of psd_rq_deflate (rotations
    from H_1's own column until the entry is negligible in H_1) every one of them passes on the device, with the scan
    chase, the two-wave chase and the one-wave chase alike (tools/r04/scan2x2.py is the full scan)."""
    cases = [(7, 39, "L", 0.8, 658499203)]
    for n in (4, 5, 6, 7, 8, 9, 10, 12):
        for p in (39, 64):
            for lr in "RL":
                for seed in (1, 2, 3, 658499203):
                    cases.append((n, p, lr, 0.8, seed))
    worst = 0.0
    for (n, p, lr, eps_, seed) in cases:
        A = pt.bench_factors(n, p, seed=seed, eps=eps_)
        ps = gpu_engine.pschur(A, lr)
        ok, err = gpu_engine.checkpsd(ps, A, thresh=100 * np.sqrt(max(n / 32, 1)))
        worst = max(worst, float(np.max(err)))
        assert ok, (n, p, lr, eps_, seed, float(np.max(err)))
    assert worst < 100.0


def test_hess_pipe_is_a_property_of_the_context(monkeypatch):
    """The form of the multi-stream Hessenberg reduction is fixed when the context is created (psd_get_hess_pipe): the
    session's engine came first and has the pipe form; one created beside it runs back to back — whatever is alive when
    a call is made —, PSD_H2_PIPE=2 / 0 force / forbid it, and a period-sharded context always has it."""
    import torch

    torch.cuda.init()
    import psd_amd

    first = psd_amd.Engine(device=0)
    second = psd_amd.Engine(device=0)
    # (whether `first` has it depends on what the test session created before; `second` was created beside `first`)
    assert second.hess_pipe() == 0
    second.set_shard(0, 2)
    assert second.hess_pipe() == 1
    second.set_shard(0, 1)
    assert second.hess_pipe() == 0
    monkeypatch.setenv("PSD_H2_PIPE", "2")
    forced = psd_amd.Engine(device=0)
    assert forced.hess_pipe() == 1
    monkeypatch.setenv("PSD_H2_PIPE", "0")
    forbidden = psd_amd.Engine(device=0)
    forbidden.set_shard(1, 2)
    assert forbidden.hess_pipe() == 0
    del first
