#!/usr/bin/env python3
"""The reference iteration's own sweep count at a bench size (not part of the product): pschur! with the multishift
trains and the multi-block scheduler off (psd_set_train(0)) runs the reference's one-shift-pair iteration sweep for
sweep (PSD.jl:471-888; tests/test_gpu_headline.py::test_trains_off_sweep_parity_512 pins that against the CPU oracle).
Its algorithmic sweep bytes are what `roofline.reference_equivalent` in bench.py divides by the benched time: a
figure of merit that does not grow when the default mode spends more sweeps per eigenvalue.
usage: python tools/ref_equiv.py N P out.json"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import torch

torch.cuda.init()
import psd_amd
import psdtest as pt

n, p = int(sys.argv[1]), int(sys.argv[2])
seed = 1234 + 2  # bench.py's input
eng = psd_amd.Engine(0)
As = pt.bench_factors(n, p, seed)
buf = torch.from_numpy(pt.pack(As)).cuda()
z = torch.zeros_like(buf)
eng.set_train(0)
lam, si, st, log = eng.pschur_dev(buf.data_ptr(), n, p, "R", dZ_ptr=z.data_ptr())
sw = log[log[:, 0] == 0]
w = (sw[:, 2] - sw[:, 1] + 1).astype(np.int64)
P = pt.product(As)
err = pt.match_eigs(np.linalg.eigvals(P), lam) / np.linalg.norm(P, 2)
out = {"n": n, "p": p, "seed": seed, "mode": "psd_set_train(0): one shift pair per sweep, one active range at a time (the reference's iteration)",
       "sweeps": int(st.nsweeps), "sweeps_logged": int(len(sw)), "rq_passes": int(st.nrqpass),
       "sweep_positions": int(w.sum()), "bytes_sweeps": float(st.bytes_sweeps),
       "sweeps_per_eigenvalue": float(st.nsweeps) / n,
       "ms_iter": float(st.ms_iter), "ms_hess": float(st.ms_hess), "ms_formq": float(st.ms_formq), "ticks": int(st.nlaunch_step),
       "bytes_hess": float(st.bytes_hess), "bytes_formq": float(st.bytes_formq),
       "eig_rel_err_vs_numpy_prod": float(err),
       "command": "python tools/ref_equiv.py %d %d <out>" % (n, p)}
print(json.dumps(out))
with open(sys.argv[3], "w") as fh:
    json.dump(out, fh, indent=1)
