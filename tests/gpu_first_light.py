import sys, time
sys.path.insert(0,'/root/repo'); sys.path.insert(0,'/root/repo/tests')
import numpy as np
import psd_amd, psdtest as pt
eng = psd_amd.Engine()
print(eng.version(), flush=True)
for (n,p) in [(5,1),(5,2),(6,4),(32,4),(40,5)]:
  for lr in "RL":
    As = pt.rand_uniform_factors(n,p,seed=100+n+p)
    ps = eng.pschur(As, lr)
    out = pt.pschur_check(As, ps)
    print(n,p,lr,"sweeps",ps.stats.nsweeps,"resid",max(out['resid']),"lam",out['lam_err'], flush=True)
for (n,p) in [(100,16),(70,20),(64,40),(48,100),(256,16),(512,16)]:
    As = pt.bench_factors(n,p,seed=n+p)
    t=time.time(); ps = eng.pschur(As, "R"); dt=time.time()-t
    ok, err = pt.checkpsd(ps, As, thresh=100*np.sqrt(max(n/32,1)))
    lam = np.linalg.eigvals(pt.product(As))
    s=ps.stats
    print(n,p,"W",s.window,"sweeps",s.nsweeps,"win",s.nwindows,"launch",s.nlaunch_step,"ok",ok,err.max(),"lam",pt.match_eigs(lam,ps.values)/abs(lam).max(), "wall",dt, "ms hess/formq/iter/total/copy",s.ms_hess,s.ms_formq,s.ms_iter,s.ms_total,s.ms_copy, flush=True)
