#!/bin/bash
set -o pipefail
O=gpurun_out/r04l; mkdir -p $O
timeout -k 10 600 python -m pytest tests/test_gpu_slices.py -m gpu -x -q > $O/pytest_slices.log 2>&1; tail -12 $O/pytest_slices.log
timeout -k 10 600 python -m pytest tests/test_gpu_complex.py -m gpu -x -q > $O/pytest_complex.log 2>&1; tail -3 $O/pytest_complex.log
python - <<'PY' 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r04l/zslices_timing.log
import os, sys
sys.path.insert(0, "tests"); sys.path.insert(0, ".")
import numpy as np, torch
torch.cuda.init()
import psd_amd, psdtest as pt
for (n, p) in [(1024, 64)]:
    A = pt.bench_factors(n, p, seed=1237, dtype=np.complex128)
    for G in (1, 2, 4):
        e = psd_amd.Engine(); e.set_slices(G)
        ps = e.pschur(A, "R")
        ok, err = e.checkpsd(ps, A, thresh=100 * np.sqrt(n / 32))
        print("c128 n %d p %d slices %d: iteration %.1f ms, %d ticks, %d sweeps, checkpsd %s %.0f eps" % (n, p, G, ps.stats.ms_iter, ps.stats.nlaunch_step, ps.stats.nsweeps, ok, float(err.max())), flush=True)
        del e
PY
