"""One full-size run of BASELINE config 3 (pschur! n=1024 p=64 ComplexF64) on one GPU: timing + accuracy gate."""
import json, sys, time
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/tests')
import numpy as np
import torch; torch.cuda.init()
import psd_amd, psdtest as pt
n, p = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (1024, 64)
eng = psd_amd.Engine()
As = pt.bench_factors(n, p, seed=1234 + 3, dtype=np.complex128)
print("inputs ready", flush=True)
t = time.time(); ps = eng.pschur(As, "R"); dt = time.time() - t
s = ps.stats
print("gpu done", dt, flush=True)
ok, err = pt.checkpsd(ps, As, thresh=100 * np.sqrt(n / 32))
P = pt.product(As)
lam_err = pt.match_eigs(np.linalg.eigvals(P), ps.values) / np.linalg.norm(P, 2)
out = {"config": f"pschur!(A,:R) n={n} p={p} ComplexF64 wantT wantZ, 1 GPU", "wall_s": dt, "window": s.window,
       "sweeps": s.nsweeps, "zero_shift_passes": s.nrqpass, "windows": s.nwindows,
       "ms": {"hessenberg": s.ms_hess, "formq": s.ms_formq, "iteration": s.ms_iter, "total": s.ms_total, "copy": s.ms_copy},
       "sweeps_per_s": s.nsweeps / (s.ms_total * 1e-3),
       "alg_GBps": {"sweeps": s.bytes_sweeps / (s.ms_iter * 1e-3) / 1e9, "hessenberg": s.bytes_hess / (s.ms_hess * 1e-3) / 1e9},
       "checkpsd_ok": bool(ok), "checkpsd_max_err_eps": float(err.max()), "eig_rel_err_vs_numpy_prod": float(lam_err)}
print(json.dumps(out), flush=True)
