import os, sys
sys.path.insert(0, "/root/repo/tests"); sys.path.insert(0, "/root/repo")
import numpy as np, torch
torch.cuda.init()
import psd_amd, psdtest as pt
n, p, lr, eps_, seed = 7, 39, "L", 0.8, 658499203
eng = psd_amd.Engine()
A = pt.bench_factors(n, p, seed=seed, eps=eps_)
ps = eng.pschur(A, lr)
ok, err = eng.checkpsd(ps, A, thresh=100*np.sqrt(max(n/32,1)))
np.set_printoptions(linewidth=220, precision=3)
print(os.environ.get("TAG",""), "ok", ok, "sweeps", ps.stats.nsweeps, "defl", ps.stats.ndefl1, ps.stats.ndefl2)
print("err per factor", np.asarray(err).ravel()[:80])
print("eigs", ps.values)
po = pt.oracle_pschur(A, lr)
print("oracle eigs", po.values)
k = int(np.argmax(np.asarray(err).ravel()))
j = k % p
print("worst factor index", k, j)
print("T_sched (gpu) subdiag pattern:", [float(abs(ps.Ts[-1 if lr=='L' else 0][i+1,i])) for i in range(n-1)])
print("T_sched (oracle) subdiag:", [float(abs(po.Ts[-1 if lr=='L' else 0][i+1,i])) for i in range(n-1)])
print("cond numbers of factors (first 8):", [float(np.linalg.cond(a)) for a in A[:8]])
