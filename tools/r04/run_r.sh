#!/bin/bash
# round 4, GPU session R: pipelined complex ordschur! drivers (tests, timing)
set -o pipefail
O=gpurun_out/r04r; mkdir -p $O
timeout -k 10 900 python -m pytest tests/test_gpu_complex.py tests/test_gpu_generalized.py tests/test_gpu_real.py -m gpu -x -q -k "ord or eigvecs" > $O/pytest_ord.log 2>&1; tail -3 $O/pytest_ord.log
timeout -k 10 600 python tools/r04/zord_timing.py > $O/zord_timing.log 2>&1; grep -v amdgpu.ids $O/zord_timing.log | tail -3
timeout -k 10 600 python tools/r04/zord_timing.py 512 64 > $O/zord_timing_512x64.log 2>&1; grep -v amdgpu.ids $O/zord_timing_512x64.log | tail -3
