"""Residual (checkpsd, in eps) and sweep counts of pschur! on given fuzz cases under the current environment.
usage: resid_probe.py n p lr eps seed [n p lr eps seed ...]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import torch

torch.cuda.init()
import psd_amd
import psdtest as pt

eng = psd_amd.Engine()
a = sys.argv[1:]
for k in range(0, len(a), 5):
    n, p, lr, eps_, seed = int(a[k]), int(a[k + 1]), a[k + 2], float(a[k + 3]), int(a[k + 4])
    A = pt.bench_factors(n, p, seed=seed, eps=eps_)
    ps = eng.pschur(A, lr)
    thresh = 100 * max(1.0, np.sqrt(n / 32))
    ok, err = eng.checkpsd(ps, A, thresh=thresh)
    print(f"n={n} p={p} {lr}: resid {float(err.max()):7.1f} eps (thresh {thresh:.1f}) sweeps {ps.stats.nsweeps} "
          f"in trains {ps.stats.reserved} ticks {ps.stats.nlaunch_step}", flush=True)
