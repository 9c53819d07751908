import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, ROOT)
import numpy as np, torch
torch.cuda.init()
import psd_amd, psdtest as pt
ref = psd_amd.Engine()
ref2 = psd_amd.Engine()
e2 = psd_amd.Engine(); e2.set_slices(2)
e4 = psd_amd.Engine(); e4.set_slices(4)
def dT(a, b): return max(np.abs(x - y).max() for x, y in zip(a.Ts, b.Ts))
for (n, p) in [(12, 8), (24, 8)]:
    A = pt.bench_factors(n, p, seed=900 + n + p)
    r1 = ref.pschur(A, "R"); r2 = ref2.pschur(A, "R"); s2 = e2.pschur(A, "R"); s4 = e4.pschur(A, "R")
    print("n", n, "p", p, "ref vs ref2 %.2e | G2 vs G4 %.2e | G2 vs ref %.2e" % (dT(r1, r2), dT(s2, s4), dT(s2, r1)), "sweeps", r1.stats.nsweeps, s2.stats.nsweeps)
    print(" log ref:", r1.sweeplog[:12].tolist())
    print(" log G2 :", s2.sweeplog[:12].tolist())
    print(" eig order same:", np.allclose(r1.values, s2.values, rtol=1e-10), " |values| diff %.2e" % np.abs(np.sort_complex(r1.values) - np.sort_complex(s2.values)).max())
    j = int(np.argmax([np.abs(x - y).max() for x, y in zip(s2.Ts, r1.Ts)]))
    D = np.abs(s2.Ts[j] - r1.Ts[j])
    print(" worst factor", j, "diag ref", np.round(np.diag(r1.Ts[j])[:8], 4).tolist(), "diag G2", np.round(np.diag(s2.Ts[j])[:8], 4).tolist())
