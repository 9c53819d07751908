"""One BASELINE config alone (for profiler runs): python tools/cfg_run.py cfg3|cfg4 [repeat]
cfg3: pschur!(A,:R) n=1024 p=64 ComplexF64; cfg4: pschur!(A,S,:R) n=512 p=32 Float64 alternating signature."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import psdtest as pt  # noqa: E402
import psd_amd  # noqa: E402

which = sys.argv[1] if len(sys.argv) > 1 else "cfg4"
rep = int(sys.argv[2]) if len(sys.argv) > 2 else 1
# PSD_DIAG_LIB=1: the diagnostic build (tick logs, PSD_GDBG stage cycles)
eng = psd_amd.Engine(device=0, libpath=psd_amd.DIAG_LIB_PATH) if os.environ.get("PSD_DIAG_LIB") else psd_amd.Engine(device=0)
for _ in range(rep):
    if which == "cfg3":
        n, p = 1024, 64
        A = pt.bench_factors(n, p, 1234 + 3, dtype=np.complex128)
        ps = eng.pschur(A, "R")
    else:
        n, p = 512, 32
        S = [True] + [bool(q % 2 == 0) for q in range(1, p)]
        A = pt.bench_factors(n, p, seed=4)
        ps = eng.pschur_([a.copy(order="F") for a in A], "R", S=S)
    st = ps.stats
    print(which, "ms", st.ms_total - st.ms_copy, "hess", st.ms_hess, "formq/stage1", st.ms_formq, "iter", st.ms_iter, "sweeps", st.nsweeps,
          "launches", st.nlaunch_step, "windows", st.nwindows, "cycles(decide,load,chase,store,total,wall)", list(st.step_cycles), flush=True)
