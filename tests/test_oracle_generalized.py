"""CPU tier: pins the oracle's signed (generalized) path against the reference's tests (test/generalized.jl)."""
import numpy as np
import pytest

import psdtest as pt


# test/generalized.jl:1-40 "Generalized Periodic Hessenberg": n = 5, p in {2,5}, alternating signature
@pytest.mark.parametrize("cplx", [False, True])
@pytest.mark.parametrize("p", [2, 5])
def test_generalized_phessenberg(built, cplx, p):
    n = 5
    S = [True]
    for l in range(1, p):
        S.append(not S[-1])
    A = pt.rand_uniform_zfactors(n, p, seed=700 + p) if cplx else pt.rand_uniform_factors(n, p, seed=700 + p)
    Hs, Qs = pt.oracle_sg_phessenberg(A, S)
    pt.sg_hess_check(A, S, Hs, Qs)


@pytest.mark.parametrize("cplx", [False, True])
def test_generalized_phessenberg_patterns(built, cplx):
    n, p = 9, 6
    for S in ([True] * 6, [True, True, False, False, True, False], [True, False, False, False, False, False]):
        A = pt.bench_factors(n, p, seed=3, dtype=np.complex128 if cplx else np.float64)
        Hs, Qs = pt.oracle_sg_phessenberg(A, S)
        pt.sg_hess_check(A, S, Hs, Qs)
