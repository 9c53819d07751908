import os, sys
sys.path.insert(0, "/root/repo/tests"); sys.path.insert(0, "/root/repo")
import numpy as np, torch
torch.cuda.init()
import psd_amd, psdtest as pt
eng = psd_amd.Engine()
bad = 0; tot = 0
for n in [4, 5, 6, 7, 8, 9, 10, 12, 16, 20]:
    for p in [8, 9, 16, 39, 64]:
        for lr in "RL":
            for eps_ in [0.3, 0.8]:
                for seed in [1, 2, 3, 658499203]:
                    A = pt.bench_factors(n, p, seed=seed, eps=eps_)
                    ps = eng.pschur(A, lr)
                    ok, err = eng.checkpsd(ps, A, thresh=100*np.sqrt(max(n/32,1)))
                    tot += 1
                    if not ok:
                        bad += 1
                        po = pt.oracle_pschur(A, lr)
                        oko, erro = pt.checkpsd(po, A, thresh=100*np.sqrt(max(n/32,1)))
                        print("FAIL", n, p, lr, eps_, seed, "%.3g" % float(np.max(err)), "oracle", bool(oko), "%.3g" % float(np.max(erro)), flush=True)
print("total", tot, "bad", bad)
