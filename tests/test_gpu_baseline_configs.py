"""GPU tier (MI355X): BASELINE.json configs[2], [3], [4] at their FULL size through the C ABI.

configs[2]  pschur!(A,:R)        n = 1024, p = 64, ComplexF64    (src/generalized.jl:108-137,166-931)
configs[3]  pschur!(A,S,:R)      n = 512,  p = 32, Float64, alternating signature (src/rgeneralized.jl:3-1083)
configs[4]  pschur!(A,:L) + ordschur!(P, select), n = 1024, p = 16, Float64, largest and smallest n/4
            (src/rordschur.jl:3-132, src/sylswap.jl)

Tolerances: eigenvalues |dlam| <= 1e-10 * ||prod A||_2 (BASELINE.json north_star; for config 4 against the CPU
oracle with the tighter scale max|lam| <= ||prod||), checkpsd residual <= 100*sqrt(n/32) eps ||A_l||_1 (the reference's
tol 100 at its n = 32 test size, scaled with sqrt(n): residuals of n-by-n orthogonal similarity chains grow like
sqrt(n); tests/test_gpu_real.py::test_residual_growth_oracle_vs_device prints the oracle's own figure next to it),
||Z Z' - I|| <= 10 eps n, exact structural zeros.
"""
import numpy as np
import pytest

import psdtest as pt

pytestmark = pytest.mark.gpu


def test_config3_full_size(gpu_engine):
    """BASELINE configs[2]: pschur!(A,:R), n = 1024, p = 64, ComplexF64 (2 GiB of factors + Schur vectors)."""
    n, p = 1024, 64
    As = pt.bench_factors(n, p, seed=1234 + 3, dtype=np.complex128)
    ps = gpu_engine.pschur(As, "R")
    assert ps.stats.nsweeps > 0
    ok, err = gpu_engine.checkpsd(ps, As, thresh=100 * np.sqrt(n / 32))
    assert ok, err
    P = pt.product(As)
    scale = np.linalg.norm(P, 2)
    assert pt.match_eigs(np.linalg.eigvals(P), ps.values) <= 1e-10 * scale


def test_config4_full_size(gpu_engine):
    """BASELINE configs[3]: pschur!(A,S,:R), n = 512, p = 32, Float64, S = [T,F,T,F,...] against the CPU oracle."""
    n, p = 512, 32
    S = [True] + [bool(q % 2 == 0) for q in range(1, p)]
    A = pt.bench_factors(n, p, seed=4)
    ps = gpu_engine.pschur_([a.copy(order="F") for a in A], "R", S=S)
    pt.rgpschur_check(A, S, ps, tol=100 * np.sqrt(n / 32), qtol=10 * np.sqrt(n / 32), lam_check=False)
    ok, err = gpu_engine.checkpsd(ps, A, thresh=100 * np.sqrt(n / 32), S=S)
    assert ok, err
    po = pt.oracle_gpschur(A, S, "R")
    assert po.info == 0
    assert pt.match_eigs(po.values, ps.values) <= 1e-10 * abs(po.values).max()


@pytest.mark.parametrize("mode", ["largest", "smallest"])
def test_config5_full_size(gpu_engine, mode):
    """BASELINE configs[4]: pschur!(A,:L) n = 1024, p = 16, then ordschur! moving the n/4 eigenvalues of largest
    (smallest) modulus, conjugates closed, to the top: invariants + selected / rest multisets."""
    n, p = 1024, 16
    As = pt.bench_factors(n, p, seed=1234 + 5)
    ps = gpu_engine.pschur(As, "L")
    lam0 = ps.values.copy()
    order = np.argsort(-np.abs(lam0) if mode == "largest" else np.abs(lam0), kind="stable")
    select = np.zeros(n, dtype=bool)
    select[order[: n // 4]] = True
    for i in np.where(lam0.imag != 0)[0]:  # close conjugate pairs (rordschur.jl:44-58 does the same)
        if select[i]:
            select[i + 1 if lam0[i].imag > 0 else i - 1] = True
    m = int(select.sum())
    ps1 = gpu_engine.ordschur_(ps, select)
    ok, err = gpu_engine.checkpsd(ps1, As, thresh=100 * np.sqrt(n / 32))
    assert ok, err
    P = pt.product(As, left=True)
    scale = np.linalg.norm(P, 2)
    assert pt.match_eigs(lam0[select], ps1.values[:m]) <= 1e-10 * scale
    assert pt.match_eigs(lam0[~select], ps1.values[m:]) <= 1e-10 * scale
    assert pt.match_eigs(np.linalg.eigvals(P), ps1.values) <= 1e-10 * scale
