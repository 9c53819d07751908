#!/usr/bin/env python3
"""Diagnostic: nb small Hessenberg-triangular problems one by one vs in one batch call (psd_d_pschur_hess_batch)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import torch
torch.cuda.init()
import psd_amd, psdtest as pt, engine_cases as ec
eng = psd_amd.Engine(0)
for (nb, n, p) in [(16, 40, 4), (32, 40, 4), (32, 24, 8), (8, 40, 16)]:
    probs = [ec._hess_ut_problem(n, p, 900 + q) for q in range(nb)]
    for rep in range(2):
        Ws = [[a.copy(order="F") for a in A] for A in probs]
        t0 = time.perf_counter()
        for W in Ws:
            eng.pschur_hess_(W[0], W[1:])
        t1 = time.perf_counter()
        Ws = [[a.copy(order="F") for a in A] for A in probs]
        t2 = time.perf_counter()
        out = eng.pschur_hess_batch_([(W[0], W[1:]) for W in Ws])
        t3 = time.perf_counter()
    print(f"nb={nb} n={n} p={p}: one by one {1e3*(t1-t0):.1f} ms, batch {1e3*(t3-t2):.1f} ms ({(t1-t0)/(t3-t2):.1f}x), "
          f"batch ticks {out[0].stats.nlaunch_step}", flush=True)
