"""GPU tier (MI355X): the period shard (psd_set_shard) on the HIP build, all four engines.

One GPU is all the box has, so rank 0 of 2 and rank 1 of 2 run one after the other on it: each must reproduce the
unsharded call's T factors and eigenvalues bit for bit (the chains are replicated, and reproducible run to run:
tests/test_gpu_headline.py), and the Schur vectors Z_j of its own slice of the period bit for bit — the slices of the two
ranks together are the whole result.  BASELINE configs[2] names this shard for ComplexF64 (src/generalized.jl:108-137,
808-852); the real signed and complex signed engines (src/rgeneralized.jl:953-1014) honour it in their iteration.
"""
import numpy as np
import pytest

import psdtest as pt

pytestmark = pytest.mark.gpu


def _run(eng, A, S, lr):
    W = [np.array(a, order="F", copy=True) for a in A]
    return eng.pschur_(W, lr) if S is None else eng.pschur_(W, lr, S=S)


@pytest.mark.parametrize("kind,n,p", [("d", 260, 12), ("d", 1024, 8), ("z", 150, 10), ("z", 512, 16), ("dg", 120, 6), ("zg", 90, 5)])
def test_two_ranks_one_after_the_other(gpu_engine, kind, n, p):
    eng = gpu_engine
    cplx = kind.startswith("z")
    S = ([True] + [bool(q % 2 == 0) for q in range(1, p)]) if kind.endswith("g") else None
    A = pt.bench_factors(n, p, seed=700 + n + p, dtype=np.complex128 if cplx else np.float64)
    full = _run(eng, A, S, "R")
    covered = np.zeros(p, dtype=bool)
    try:
        for rank in range(2):
            eng.set_shard(rank, 2)
            part = _run(eng, A, S, "R")
            owned = eng.owned_slots(p, "R")
            assert owned.sum() in (p // 2, p - p // 2)
            assert np.array_equal(np.asarray(part.values), np.asarray(full.values)), (kind, rank)
            for j in range(p):
                assert np.array_equal(part.Ts[j], full.Ts[j]), (kind, rank, j)
                if owned[j]:
                    assert np.array_equal(part.Z[j], full.Z[j]), (kind, rank, j)
            covered |= owned
    finally:
        eng.set_shard(0, 1)
    assert covered.all()
