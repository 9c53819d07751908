#!/usr/bin/env python3
"""Diagnostic: phases of pschur! on the GPU for a list of NxP sizes (not part of the product)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import torch
torch.cuda.init()
import psd_amd, psdtest as pt
eng = psd_amd.Engine(0)
for a in sys.argv[1:]:
    n, p = map(int, a.split("x"))
    As = pt.bench_factors(n, p, 1236)
    for rep in range(2):
        buf = torch.from_numpy(pt.pack(As)).cuda()
        z = torch.zeros_like(buf)
        lam, si, st, log = eng.pschur_dev(buf.data_ptr(), n, p, "R", dZ_ptr=z.data_ptr())
    a0 = torch.from_numpy(pt.pack(As)).cuda()
    ok, err, orth, tri = eng.checkpsd_dev(buf.data_ptr(), z.data_ptr(), a0.data_ptr(), n, p, "R", 1, thresh=100 * np.sqrt(n / 32))
    print(a, "hess %.1f ms (%.0f GB/s alg) formq %.1f iter %.1f total %.1f sweeps %d ticks %d checkpsd %s %.1f" % (
        st.ms_hess, st.bytes_hess / st.ms_hess / 1e6, st.ms_formq, st.ms_iter, st.ms_total, st.nsweeps, st.nlaunch_step, ok, err.max()), flush=True)
