"""BASELINE config 5 at full size: pschur!(A,:L) n=1024 p=16 Float64, then ordschur! moving the n/4 eigenvalues of
largest modulus (conjugates closed) to the top.  Timing + invariants."""
import json, sys, time
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/tests')
import numpy as np
import torch; torch.cuda.init()
import psd_amd, psdtest as pt
n, p = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (1024, 16)
mode = sys.argv[3] if len(sys.argv) > 3 else "largest"
eng = psd_amd.Engine()
As = pt.bench_factors(n, p, seed=1234 + 5)
t = time.time(); ps = eng.pschur(As, "L"); t_ps = time.time() - t
s0 = ps.stats
print("pschur done", t_ps, "sweeps", s0.nsweeps, flush=True)
lam0 = ps.values.copy()
order = np.argsort(-np.abs(lam0) if mode == "largest" else np.abs(lam0), kind="stable")
select = np.zeros(n, dtype=bool)
select[order[: n // 4]] = True
for i in np.where(lam0.imag != 0)[0]:   # close conjugate pairs
    if select[i]:
        j = i + 1 if lam0[i].imag > 0 else i - 1
        select[j] = True
m = int(select.sum())
t = time.time(); ps1 = eng.ordschur_(ps, select); t_ord = time.time() - t
s1 = ps1.stats
print("ordschur done", t_ord, "swaps", s1.nsweeps, "windows", s1.nwindows, flush=True)
ok, err = pt.checkpsd(ps1, As, thresh=100 * np.sqrt(n / 32))
sc = abs(lam0).max()
e_sel = pt.match_eigs(lam0[select], ps1.values[:m]) / sc
e_rest = pt.match_eigs(lam0[~select], ps1.values[m:]) / sc
out = {"config": f"pschur!(A,:L) then ordschur!(P, select) n={n} p={p} Float64, select = {mode} n/4 by modulus, |select| = {m}", "pschur_wall_s": t_ps,
       "pschur_ms": {"hessenberg": s0.ms_hess, "formq": s0.ms_formq, "iteration": s0.ms_iter}, "pschur_sweeps": s0.nsweeps,
       "ordschur_wall_s": t_ord, "ordschur_device_ms": s1.ms_total, "swaps": s1.nsweeps, "windows": s1.nwindows,
       "swaps_per_s": s1.nsweeps / (s1.ms_total * 1e-3), "window": s1.window, "step_cycles": list(s1.step_cycles),
       "alg_GBps_ordschur": s1.nsweeps * 2 * 8 * p * 3 * n * 2.5 / (s1.ms_total * 1e-3) / 1e9,
       "checkpsd_ok": bool(ok), "checkpsd_max_err_eps": float(err.max()), "selected_match": e_sel, "rest_match": e_rest}
print(json.dumps(out), flush=True)
