"""Generates the committed golden vectors (run once, here; the GPU box only reads the .npz files).

The reference is pure Julia and cannot be executed in this image, so goldens are *independent* high-precision
answers for the reference's own fixtures and for BASELINE config 1:
  * eigenvalues of the explicit product computed with mpmath at 80 digits (the reference's tests use
    `eigvals(prod(A))` in Float64 as their oracle, test/testfuncs.jl:123-134; mpmath removes the Float64
    conditioning of that product from the comparison);
  * Kressner's exponentially split example (test/testfuncs.jl:412-421) for p = 5, 20: literal inputs and
    mpmath eigenvalues (the reference checks against 1e-3-accurate asymptotic values).
"""
import os
import sys

import mpmath as mp
import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
import psdtest as pt  # noqa: E402

mp.mp.dps = 80


def mp_product_eigs(As, left=False):
    n = As[0].shape[0]
    P = mp.eye(n)
    for a in As:
        M = mp.matrix(a.tolist())
        P = M * P if left else P * M
    ev = mp.eig(P, left=False, right=False)
    out = []
    for e in ev:
        z = complex(e)
        if abs(z.imag) < 1e-40 * max(abs(z), 1e-300):  # mpmath leaves 1e-80-size imaginary dust on real roots
            z = complex(z.real, 0.0)
        out.append(z)
    return np.array(out)


def main():
    out = {}
    # BASELINE config 1: pschur!(A,:R) n=32 p=4 Float64 — bench ensemble, seed 1234+1
    As = pt.bench_factors(32, 4, seed=1235)
    out["cfg1_lam"] = mp_product_eigs(As)
    out["cfg1_seed"] = np.array([1235])
    # the reference's own random-test shape (rand(T,n,n), n=5, p in 1,2,3,5; test/runtests.jl:89-100)
    for p in (1, 2, 3, 5):
        As = pt.rand_uniform_factors(5, p, seed=500 + p)
        out[f"rand5_p{p}_lam"] = mp_product_eigs(As)
        out[f"rand5_p{p}_lamL"] = mp_product_eigs(As, left=True)
    # n=32, p=4 U(0,1) (largest shape in the reference's tests, test/generalized.jl:154-173)
    As = pt.rand_uniform_factors(32, 4, seed=532)
    out["rand32_p4_lam"] = mp_product_eigs(As)
    for p in (5, 20):
        A, _ = pt.expsplit(p)
        out[f"expsplit_p{p}_lam"] = mp_product_eigs(A)
    np.savez(os.path.join(HERE, "golden_real.npz"), **out)
    for k, v in out.items():
        print(k, v.shape)


if __name__ == "__main__":
    main()
