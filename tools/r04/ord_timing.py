"""ordschur! of BASELINE configs[4] (n = 1024, p = 16 after pschur!(A, :L)): smallest and largest quarter, pipelined driver
against the serial one (PSD_ORD_PIPE).  usage: python tools/r04/ord_timing.py [n p]"""
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
A = pt.bench_factors(n, p, seed=1238)
os.environ["PSD_ORD_PIPE"] = "1"
e1 = psd_amd.Engine()
os.environ["PSD_ORD_PIPE"] = "0"
e0 = psd_amd.Engine()
ps0 = e1.pschur(A, "L")
lam0 = ps0.values.copy()
order = np.argsort(np.abs(lam0))
for which in ("smallest", "largest"):
    sel = np.zeros(n, dtype=bool)
    idx = order[: n // 4] if which == "smallest" else order[-(n // 4):]
    sel[idx] = True
    for name, e in (("pipelined", e1), ("serial", e0)):
        P = psd_amd.PeriodicSchur([t.copy(order="F") for t in ps0.Ts], [z.copy(order="F") for z in ps0.Z], lam0.copy(),
                                  ps0.orientation, ps0.schurindex)
        t0 = time.time()
        ps1 = e.ordschur_(P, sel)
        wall = time.time() - t0
        ok, err = e.checkpsd(ps1, A, thresh=100 * np.sqrt(n / 32))
        m = int((sel | np.isin(np.arange(n), [])).sum())
        m = int(np.sum(np.abs(ps1.values[: n]) > -1))  # (all)
        msel = int(sel.sum())
        sc = np.abs(lam0).max()
        # conjugates are taken along: compare as multisets over the leading block the engine reports
        k = msel
        while k < n and ps1.values[k - 1].imag > 0:
            k += 1
        s = ps1.stats
        print("n %d p %d %s %s: %.1f ms (wall %.2f s), %d swaps, %d windows, %d ticks, W %d, %.0f swaps/s, checkpsd %s %.0f eps"
              % (n, p, which, name, s.ms_total, wall, s.nsweeps, s.nwindows, s.nlaunch_step, s.window,
                 s.nsweeps / (s.ms_total * 1e-3) if s.ms_total else 0, ok, float(err.max())), flush=True)
