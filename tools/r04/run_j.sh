#!/bin/bash
set -o pipefail
O=gpurun_out/r04j; mkdir -p $O
timeout -k 10 600 python -m pytest tests/test_gpu_slices.py -m gpu -x -q > $O/pytest_slices.log 2>&1; tail -15 $O/pytest_slices.log
timeout -k 10 900 python -m pytest tests/test_gpu_real.py tests/test_gpu_headline.py -m gpu -x -q > $O/pytest_real.log 2>&1; tail -3 $O/pytest_real.log
python - <<'PY' 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r04j/resid2.log
import os, sys
sys.path.insert(0, "tests"); sys.path.insert(0, ".")
import numpy as np, torch
torch.cuda.init()
import psd_amd, psdtest as pt
cases = [(376, 16, 'R', 0.3, 795111496), (382, 4, 'L', 0.3, 285552735), (671, 21, 'L', 0.3, 606762519), (675, 5, 'L', 0.3, 899849684),
         (633, 6, 'L', 0.3, 1041486697), (616, 3, 'R', 0.3, 854509799), (512, 16, 'R', 0.5, 1236), (1024, 64, 'R', 0.5, 1236)]
engs = []
for s in ({}, {"PSD_C3": "0"}):
    os.environ.pop("PSD_C3", None); os.environ.update(s); engs.append(psd_amd.Engine())
for (n, p, lr, eps_, seed) in cases:
    A = pt.bench_factors(n, p, seed=seed, eps=eps_)
    gate = 100 * np.sqrt(max(n / 32, 1)); row = []
    for e in engs:
        ps = e.pschur(A, lr); ok, err = e.checkpsd(ps, A, thresh=gate)
        row.append("%.0f/%.2f %dsw %.0fms" % (float(np.max(err)), float(np.max(err)) / gate, ps.stats.nsweeps, ps.stats.ms_iter))
    print((n, p, lr, eps_), "c3:", row[0], "| c2:", row[1], flush=True)
PY
tools/psd_profile 1024 64 3 | tail -1
