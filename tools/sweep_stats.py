#!/usr/bin/env python3
"""Diagnostic: distribution of sweep widths / train sizes of one pschur! on the GPU (not part of the product).
usage: python tools/sweep_stats.py N P [out.npz]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import torch
torch.cuda.init()
import psd_amd, psdtest as pt

n, p = int(sys.argv[1]), int(sys.argv[2])
eng = psd_amd.Engine(0)
As = pt.bench_factors(n, p, 1236)
buf = torch.from_numpy(pt.pack(As)).cuda()
z = torch.zeros_like(buf)
lam, si, st, log = eng.pschur_dev(buf.data_ptr(), n, p, "R", dZ_ptr=z.data_ptr())
sw = log[log[:, 0] == 0]
w = sw[:, 2] - sw[:, 1] + 1
print("sweeps", len(sw), "ticks", st.nlaunch_step, "windows", st.nwindows, "ms_iter", st.ms_iter, "trainsweeps", st.reserved)
bins = [0, 8, 16, 24, 32, 48, 64, 96, 128, 256, 512, 1025]
h, _ = np.histogram(w, bins)
for lo, hi, c in zip(bins[:-1], bins[1:], h):
    sel = (w >= lo) & (w < hi)
    print(f"w in [{lo},{hi}): sweeps {c}  positions {int(w[sel].sum())}")
# consecutive identical (l,i) entries = one train
runs = []
k = 0
while k < len(sw):
    j = k
    while j + 1 < len(sw) and (sw[j + 1, 1:] == sw[k, 1:]).all():
        j += 1
    runs.append((j - k + 1, int(w[k])))
    k = j + 1
runs = np.array(runs)
print("runs (consecutive equal (l,i)):", len(runs), "size>1:", int((runs[:, 0] > 1).sum()))
for lo, hi in zip(bins[:-1], bins[1:]):
    sel = (runs[:, 1] >= lo) & (runs[:, 1] < hi)
    if sel.any():
        print(f"  w in [{lo},{hi}): runs {int(sel.sum())} mean size {runs[sel,0].mean():.1f}")
if len(sys.argv) > 3:
    np.savez(sys.argv[3], log=log)
