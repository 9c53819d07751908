#!/bin/bash
# round 4, GPU session Q: pipelined real ordschur! (tests, timing against the serial driver)
set -o pipefail
O=gpurun_out/r04q; mkdir -p $O
timeout -k 10 900 python -m pytest tests/test_gpu_real.py tests/test_gpu_generalized.py -m gpu -x -q -k "ord or eigvecs" > $O/pytest_ord.log 2>&1; tail -3 $O/pytest_ord.log
timeout -k 10 600 python tools/r04/ord_timing.py > $O/ord_timing.log 2>&1; grep -v amdgpu.ids $O/ord_timing.log | tail -6
timeout -k 10 600 python tools/r04/ord_timing.py 512 64 > $O/ord_timing_512x64.log 2>&1; grep -v amdgpu.ids $O/ord_timing_512x64.log | tail -6
timeout -k 10 900 python -m pytest tests/test_gpu_baseline_configs.py -m gpu -x -q > $O/pytest_cfg.log 2>&1; tail -3 $O/pytest_cfg.log
