"""Multishift trains in the signed engines (psd_set_train_g) on one GPU: usage gpu_gtrain.py [n,p[,c] ...] (c: ComplexF64).
One JSON line per problem and train width: timings, counters, invariants, eigenvalue distance to the run without trains."""
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

cases = [tuple(a.split(",")) for a in sys.argv[1:]] or [("256", "8"), ("512", "32")]  # n,p[,c] (c: ComplexF64)
eng = psd_amd.Engine()
for cs in cases:
    n, p, cplx = int(cs[0]), int(cs[1]), len(cs) > 2 and cs[2] == "c"
    S = [True] + [bool(q % 2 == 0) for q in range(1, p)]
    dt = np.complex128 if cplx else np.float64
    A = pt.bench_factors(n, p, seed=4, dtype=dt)
    eng.pschur_([a.copy(order="F") for a in pt.bench_factors(32, p, seed=1, dtype=dt)], "R", S=S)  # warm-up
    ref = None
    for tw in [int(x) for x in os.environ.get("GTRAIN_SEQ", "0,2,4,8").split(",")]:
        eng.set_train_g(tw)
        t0 = time.time()
        ps = eng.pschur_([a.copy(order="F") for a in A], "R", S=S)
        wall = time.time() - t0
        st = ps.stats
        ok = True
        try:
            if cplx:
                pt.gpschur_check(A, S, ps, tol=100 * max(1.0, n / 32), qtol=10 * np.sqrt(n / 32))
            else:
                pt.rgpschur_check(A, S, ps, tol=100 * np.sqrt(n / 32), qtol=10 * np.sqrt(n / 32), lam_check=False)
        except AssertionError as e:
            ok = str(e)
        if ref is None:
            ref = ps.values
        fin = np.isfinite(ref)
        err = pt.match_eigs(ref[fin], ps.values[np.isfinite(ps.values)]) / max(abs(ref[fin]).max(), 1e-300)
        print(json.dumps({"n": n, "p": p, "eltype": "c128" if cplx else "f64", "train_g": tw, "wall_s": wall, "ms_iter": st.ms_iter, "ms_hess": st.ms_hess,
                          "sweeps": st.nsweeps, "sweeps_in_trains": st.maxits, "zero_shift_passes": st.nrqpass,
                          "windows": st.nwindows, "step_launches": st.nlaunch_step, "window": st.window,
                          "invariants_ok": ok, "eig_rel_dist_vs_no_trains": float(err)}), flush=True)
eng.set_train_g(0)
