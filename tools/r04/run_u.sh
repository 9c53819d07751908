#!/bin/bash
set -o pipefail
O=gpurun_out/r04u; mkdir -p $O
PSD_DIAG_LIB=1 PSD_GDBG=1 python tools/cfg_run.py cfg4 2 2>&1 | grep -v amdgpu | tail -6 | tee $O/cfg4_gdbg.log
