"""Signed periodic Hessenberg reduction (generalized.jl:988-1082) on one GPU: wall time of psd_{d,z}_gphessenberg with
the pipelined stage 2 and, in a second process (PSD_HESS_SERIAL=1), with the single-wave chase.
usage: gpu_hess_timing.py n p d|z"""
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

torch.cuda.init()
import psd_amd  # noqa: E402
import psdtest as pt  # noqa: E402

n, p, kind = int(sys.argv[1]), int(sys.argv[2]), sys.argv[3]
dt = np.complex128 if kind == "z" else np.float64
S = [True] + [bool(q % 2 == 0) for q in range(1, p)]
A = pt.bench_factors(n, p, seed=9, dtype=dt)
eng = psd_amd.Engine()
eng.gphessenberg_([a.copy(order="F") for a in pt.bench_factors(24, p, seed=1, dtype=dt)], S)  # warm-up
t0 = time.time()
Hs, Qs = eng.gphessenberg_([a.copy(order="F") for a in A], S)
wall = time.time() - t0
ok = True
try:
    pt.sg_hess_check(A, S, Hs, Qs, tol=20 * max(1, n / 8), qtol=10 * max(1, n / 16))
except AssertionError as e:
    ok = str(e)[:200]
print(json.dumps({"config": f"_phessenberg!(A, S) n={n} p={p} {kind}", "serial": os.environ.get("PSD_HESS_SERIAL", "0"),
                  "wall_s": wall, "invariants_ok": ok}))
