"""per-window cycle shares of the chase launch (stats.step_cycles: decide, window load, chase, window store, total, wall)"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, ROOT)
import numpy as np, torch
torch.cuda.init()
import psd_amd, psdtest as pt
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
p = int(sys.argv[2]) if len(sys.argv) > 2 else 64
eng = psd_amd.Engine()
A = pt.bench_factors(n, p, seed=1234)
for rep in range(2):
    ps = eng.pschur(A, "R")
st = ps.stats
c = list(st.step_cycles)
nw = st.nwindows
mhz = c[4] / max(c[5], 1) * 100.0
print("C3=%s n=%d p=%d ms_iter=%.1f launches=%d windows=%d sweeps=%d shader MHz=%.0f" % (os.environ.get("PSD_C3", "default"), n, p, st.ms_iter, st.nlaunch_step, nw, st.nsweeps, mhz))
print(" per window (cycles): load %.0f chase %.0f store %.0f | decide total %.0f Mcyc, all leaders' kernels %.0f Mcyc" % (c[1] / nw, c[2] / nw, c[3] / nw, c[0] / 1e6, c[4] / 1e6))
print(" per window (us at that clock): load %.1f chase %.1f store %.1f" % (c[1] / nw / mhz, c[2] / nw / mhz, c[3] / nw / mhz))
