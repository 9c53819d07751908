#!/usr/bin/env python3
"""Diagnostic (not part of the product): the bulk-apply kernel of the periodic QR iteration alone, on synthetic windows.
usage: python tools/apply_bench.py [n p W]      (environment: PSD_APPLY_WL2=0|1, PSD_APPLY_WL2_GRID, PSD_APPLY_WL2_WPE)"""
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch

torch.cuda.init()
import psd_amd

n, p, W = (int(x) for x in sys.argv[1:4]) if len(sys.argv) >= 4 else (1024, 64, 17)
eng = psd_amd.Engine(0, libpath=psd_amd.DIAG_LIB_PATH)  # (tuning knobs exist in the diagnostic build only)
f = eng.lib.psd_dbg_apply_bench
f.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_double), C.POINTER(C.c_double)]
f.restype = C.c_int
tag = " ".join(f"{k}={v}" for k, v in os.environ.items() if k.startswith("PSD_APPLY"))
for nwin in (1, 4, 16, 40, 60):
    ms = (C.c_double * 3)()
    by = (C.c_double * 3)()
    rc = f(eng.ctx, n, p, nwin, W, 7, 20, ms, by)
    if rc != 0:
        print("nwin", nwin, "rc", rc)
        continue
    out = []
    for q, name in enumerate(("rows", "cols", "Z")):
        out.append(f"{name} {ms[q]*1e3:7.1f} us {by[q]/1e6:7.1f} MB {by[q]/ms[q]/1e9 if ms[q] else 0:5.2f} TB/s")
    print(f"[{tag}] n={n} p={p} W={W} nwin={nwin:2d}: " + " | ".join(out), flush=True)
