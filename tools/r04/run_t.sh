#!/bin/bash
# round 4, GPU session T: stage-2 scan windows with the LDS-only barrier; where the signed leader's decisions spend their time
set -o pipefail
O=gpurun_out/r04t; mkdir -p $O
timeout -k 10 900 python -m pytest tests/test_gpu_generalized.py -m gpu -x -q > $O/pytest_g.log 2>&1; tail -4 $O/pytest_g.log
python tools/cfg_run.py cfg4 2 2>&1 | grep -v amdgpu | tail -3 | tee $O/cfg4.log
PSD_DIAG_LIB=1 PSD_GDBG=1 python tools/cfg_run.py cfg4 1 2>&1 | grep -v amdgpu | tail -3 | tee $O/cfg4_gdbg.log
