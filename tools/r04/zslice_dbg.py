import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, ROOT)
import numpy as np, torch
torch.cuda.init()
import psd_amd, psdtest as pt
ref = psd_amd.Engine()
e2 = psd_amd.Engine(); e2.set_slices(2)
e4 = psd_amd.Engine(); e4.set_slices(4)
def dT(a, b): return max(np.abs(x - y).max() for x, y in zip(a.Ts, b.Ts))
for (n, p) in [(12, 8), (40, 8), (80, 8), (60, 16)]:
    A = pt.bench_factors(n, p, seed=700 + n + p, dtype=np.complex128)
    r1 = ref.pschur(A, "R"); a2 = e2.pschur(A, "R"); b2 = e2.pschur(A, "R"); a4 = e4.pschur(A, "R"); b4 = e4.pschur(A, "R")
    print("n", n, "p", p, "sweeps ref/G2/G2/G4/G4", r1.stats.nsweeps, a2.stats.nsweeps, b2.stats.nsweeps, a4.stats.nsweeps, b4.stats.nsweeps,
          "| G2 twice %.1e G4 twice %.1e G2 vs G4 %.1e G2 vs ref %.1e" % (dT(a2, b2), dT(a4, b4), dT(a2, a4), dT(a2, r1)))
