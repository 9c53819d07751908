"""BASELINE config 4 at full size on one GPU: pschur!(A, S, :R), Float64; usage: gpu_config4.py n p [alternating|random25].
Prints one JSON line with phase timings, the oracle's (CPU, 1 thread) time for the same input and the invariants."""
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402  (torch's bundled HIP runtime has to initialise first)

torch.cuda.init()
import psd_amd  # noqa: E402
import psdtest as pt  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 256
p = int(sys.argv[2]) if len(sys.argv) > 2 else 8
kind = sys.argv[3] if len(sys.argv) > 3 else "alternating"
if kind == "random25":  # second run with ~25 % of the factors inverted, at random (seeded)
    rs = np.random.RandomState(1234 + 4)
    S = [True] + [bool(x) for x in (rs.rand(p - 1) >= 0.25)]
else:
    S = [True] + [bool(q % 2 == 0) for q in range(1, p)]
A = pt.bench_factors(n, p, seed=4)
eng = psd_amd.Engine()
eng.pschur_([a.copy(order="F") for a in pt.bench_factors(32, p, seed=1)], "R", S=S)  # warm-up
t0 = time.time()
ps = eng.pschur_([a.copy(order="F") for a in A], "R", S=S)
wall = time.time() - t0
st = ps.stats
t0 = time.time()
po = pt.oracle_gpschur(A, S, "R")
cpu = time.time() - t0
ok = True
try:
    pt.rgpschur_check(A, S, ps, tol=100 * np.sqrt(n / 32), qtol=10 * np.sqrt(n / 32), lam_check=False)
except AssertionError as e:
    ok = str(e)
err = pt.match_eigs(po.values, ps.values) / abs(po.values).max()
print(json.dumps({
    "config": f"pschur!(A, S, :R) n={n} p={p} Float64 {kind} signature ({p - sum(S)} of {p} inverted)", "wall_s": wall,
    "ms": {"hessenberg_total": st.ms_hess, "hessenberg_stage1": st.ms_formq, "iteration": st.ms_iter, "total": st.ms_total},
    "sweeps": st.nsweeps, "zero_shift_passes": st.nrqpass, "blocks2x2": st.ndefl2, "windows": st.nwindows,
    "step_launches": st.nlaunch_step, "window": st.window, "oracle_cpu_s": cpu, "oracle_counters": po.counters,
    "invariants_ok": ok, "eig_rel_err_vs_oracle": err}))
