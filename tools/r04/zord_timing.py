"""ordschur! of a ComplexF64 PeriodicSchur (n = 1024, p = 16 unless given): smallest quarter, pipelined driver against the
serial one (PSD_ORD_PIPE).  usage: python tools/r04/zord_timing.py [n p]"""
import os
import sys
import time

sys.path.insert(0, "tests")
sys.path.insert(0, ".")
import numpy as np
import torch

torch.cuda.init()
import psd_amd
import psdtest as pt

n, p = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (1024, 16)
A = pt.bench_factors(n, p, seed=1240, dtype=np.complex128)
os.environ["PSD_ORD_PIPE"] = "1"
e1 = psd_amd.Engine()
os.environ["PSD_ORD_PIPE"] = "0"
e0 = psd_amd.Engine()
ps0 = e1.pschur(A, "R")
lam0 = ps0.values.copy()
order = np.argsort(np.abs(lam0))
sel = np.zeros(n, dtype=bool)
sel[order[: n // 4]] = True
for name, e in (("pipelined", e1), ("serial", e0)):
    P = psd_amd.PeriodicSchur([t.copy(order="F") for t in ps0.Ts], [z.copy(order="F") for z in ps0.Z], lam0.copy(),
                              ps0.orientation, ps0.schurindex)
    t0 = time.time()
    ps1 = e.ordschur_(P, sel)
    wall = time.time() - t0
    ok, err = e.checkpsd(ps1, A, thresh=100 * np.sqrt(n / 32))
    s = ps1.stats
    m = int(sel.sum())
    print("c128 n %d p %d smallest quarter %s: %.1f ms (wall %.2f s), %d swaps, %d windows, %d ticks, W %d, %.0f swaps/s, checkpsd %s %.0f eps, selected match %.1e"
          % (n, p, name, s.ms_total, wall, s.nsweeps, s.nwindows, s.nlaunch_step, s.window,
             s.nsweeps / (s.ms_total * 1e-3) if s.ms_total else 0, ok, float(err.max()),
             pt.match_eigs(lam0[sel], ps1.values[:m]) / np.abs(lam0).max()), flush=True)
