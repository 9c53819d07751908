import sys, time
sys.path.insert(0,'/root/repo'); sys.path.insert(0,'/root/repo/tests')
import numpy as np
import torch; torch.cuda.init()
import psd_amd, psdtest as pt
eng = psd_amd.Engine()
sizes = [(256,16),(512,16)] if len(sys.argv) < 2 else [tuple(map(int,a.split('x'))) for a in sys.argv[1:]]
for (n,p) in sizes:
    As = pt.bench_factors(n,p,seed=1236)
    eng.pschur(As,"R")
    t=time.time(); ps = eng.pschur(As, "R"); dt=time.time()-t
    ok, err = pt.checkpsd(ps, As, thresh=100*np.sqrt(max(n/32,1)))
    s=ps.stats
    print(n,p,"W",s.window,"sweeps",s.nsweeps,"win",s.nwindows,"ok",ok,round(err.max(),1),"ms hess/formq/iter/total",round(s.ms_hess,1),round(s.ms_formq,1),round(s.ms_iter,1),round(s.ms_total,1),"us/window",round(1e3*s.ms_iter/s.nwindows,1), "cyc decide/load/chase/store/total",[round(c/1e6,1) for c in s.step_cycles[:5]],"MHz",round(100*s.step_cycles[4]/max(s.step_cycles[5],1)), flush=True)
