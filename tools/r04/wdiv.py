"""residual / time of the real engine on the fuzz inputs that sit at the residual gate, under train settings (diagnostic build)"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, ROOT)
import numpy as np, torch
torch.cuda.init()
import psd_amd, psdtest as pt
cases = [(376, 16, 'R', 0.3, 795111496), (382, 4, 'L', 0.3, 285552735), (671, 21, 'L', 0.3, 606762519), (675, 5, 'L', 0.3, 899849684),
         (633, 6, 'L', 0.3, 1041486697), (616, 3, 'R', 0.3, 854509799), (512, 16, "R", 0.5, 1236), (1024, 16, "L", 0.5, 1238), (1024, 64, "R", 0.5, 1236), (371, 8, "R", 0.3, 7), (700, 8, "R", 0.5, 11), (650, 11, "L", 0.8, 13)]
settings = [{}, {"PSD_TRAIN_WDIV": "8"}]
engs = []
for s in settings:
    for k in ("PSD_C3", "PSD_TRAIN_WDIV", "PSD_TRAIN"):
        os.environ.pop(k, None)
    os.environ.update(s)
    engs.append(psd_amd.Engine(libpath=psd_amd.DIAG_LIB_PATH))
for (n, p, lr, eps_, seed) in cases:
    A = pt.bench_factors(n, p, seed=seed, eps=eps_)
    gate = 100 * np.sqrt(max(n / 32, 1))
    row = []
    for s, e in zip(settings, engs):
        ps = e.pschur(A, lr)
        ok, err = e.checkpsd(ps, A, thresh=gate)
        row.append("%s: %.0f/%.2f %dsw %.0fms" % (",".join(f"{k[4:]}={v}" for k, v in s.items()) or "default", float(np.max(err)), float(np.max(err)) / gate, ps.stats.nsweeps, ps.stats.ms_iter))
    print((n, p, lr, eps_), " | ".join(row), flush=True)
