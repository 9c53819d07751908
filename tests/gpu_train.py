"""Experimental multishift trains (psd_set_train) against the default iteration at cfg2 on one GPU: wall time, ticks,
sweeps, invariants.  usage: gpu_train.py [n] [p] [bulges ...]   (PSD_ELTYPE=c128: ComplexF64; GPU_MAX_HW_QUEUES is raised before HIP starts: every
cursor needs its own hardware queue to run beside the others)"""
import json
import os
import sys
import time

os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")
import numpy as np

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

torch.cuda.init()
import psd_amd  # noqa: E402
import psdtest as pt  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 512
p = int(sys.argv[2]) if len(sys.argv) > 2 else 16
ms = [int(x) for x in sys.argv[3:]] or [1, 2, 4, 6]
DT = np.complex128 if os.environ.get("PSD_ELTYPE") == "c128" else np.float64
As = pt.bench_factors(n, p, seed=1236, dtype=DT)
P = pt.product(As)
lam = np.linalg.eigvals(P)
eng = psd_amd.Engine()
for m in ms:
    eng.set_train(m)
    eng.pschur(pt.bench_factors(64, p, seed=1, dtype=DT), "R")  # warm-up
    t0 = time.time()
    ps = eng.pschur(As, "R")
    wall = time.time() - t0
    ok, err = pt.checkpsd(ps, As, thresh=100 * np.sqrt(n / 32))
    st = ps.stats
    print(json.dumps({"bulges": m, "wall_s": wall, "ms_iter": st.ms_iter, "ticks": st.nlaunch_step, "sweeps": st.nsweeps,
                      "sweeps_in_trains": st.reserved, "windows": st.nwindows, "sweeps_per_s": st.nsweeps / wall,
                      "checkpsd_ok": bool(ok), "max_err_eps": float(err.max()),
                      "eig_rel_err": float(pt.match_eigs(lam, ps.values) / np.linalg.norm(P, 2))}), flush=True)
