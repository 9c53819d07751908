#!/bin/bash
# round 4, GPU session S: scan form of the signed double-shift sweep (tests, configs[3] timing A/B)
set -o pipefail
O=gpurun_out/r04s; mkdir -p $O
timeout -k 10 900 python -m pytest tests/test_gpu_generalized.py -m gpu -x -q > $O/pytest_g.log 2>&1; tail -4 $O/pytest_g.log
python tools/cfg_run.py cfg4 2 2>&1 | grep -v amdgpu | tail -3 | tee $O/cfg4.log
PSD_GSCAN=0 python tools/cfg_run.py cfg4 1 2>&1 | grep -v amdgpu | tail -2 | tee -a $O/cfg4.log
