"""A/B timing of two builds of the library on the same box: pschur!(A,:R) with device-resident operands, phase times
per repetition.  usage: ab_bench.py --lib A.so --lib B.so [--n 1024 --p 64 --reps 3]"""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

ap = argparse.ArgumentParser()
ap.add_argument("--lib", action="append", required=True)
ap.add_argument("--n", type=int, default=1024)
ap.add_argument("--p", type=int, default=64)
ap.add_argument("--reps", type=int, default=3)
args = ap.parse_args()

import torch

torch.cuda.init()
import psd_amd
import psdtest as pt

n, p = args.n, args.p
As = pt.bench_factors(n, p, 1236)
host = torch.from_numpy(pt.pack(As))
dev = torch.device("cuda", 0)
engines = [(lib, psd_amd.Engine(device=0, libpath=lib)) for lib in args.lib]
for rep in range(args.reps + 1):
    for lib, eng in engines:
        buf = host.to(dev)
        z = torch.zeros_like(buf)
        torch.cuda.synchronize()
        lam, si, st, _ = eng.pschur_dev(buf.data_ptr(), n, p, "R", dZ_ptr=z.data_ptr())
        if rep:
            print(f"{os.path.basename(lib):28s} rep {rep}: hess {st.ms_hess:7.1f} formq {st.ms_formq:5.1f} iter {st.ms_iter:7.1f} ms  "
                  f"ticks {st.nlaunch_step} sweeps {st.nsweeps}", flush=True)
