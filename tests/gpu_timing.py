import sys, time
sys.path.insert(0,'/root/repo'); sys.path.insert(0,'/root/repo/tests')
import numpy as np
import torch; torch.cuda.init()
import psd_amd, psdtest as pt
eng = psd_amd.Engine()
args = sys.argv[1:] or ["256x16", "512x16"]
for a in args:
    cplx = a.endswith("z")
    n, p = map(int, a.rstrip("z").split("x"))
    As = pt.bench_factors(n, p, seed=1236, dtype=np.complex128 if cplx else np.float64)
    t = time.time(); ps = eng.pschur(As, "R"); dt = time.time() - t
    ok, err = pt.checkpsd(ps, As, thresh=100*np.sqrt(max(n/32,1)))
    s = ps.stats
    ms = s.nwindows * (s.window - (3 if cplx else 4)) * p
    print(a, "W", s.window, "sweeps", s.nsweeps, "zshift/rq", s.nrqpass, "win", s.nwindows, "ok", ok, round(err.max(),1),
          "ms hess/formq/iter/total", round(s.ms_hess,1), round(s.ms_formq,1), round(s.ms_iter,1), round(s.ms_total,1),
          "us/window", round(1e3*s.ms_iter/max(s.nwindows,1),1),
          "Mcyc decide/load/chase/store/total", [round(c/1e6,1) for c in s.step_cycles[:5]],
          "cyc/microstep~", round(s.step_cycles[2]/max(ms,1)), "wall", round(dt,2), flush=True)
