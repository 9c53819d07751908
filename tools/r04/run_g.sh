#!/bin/bash
set -o pipefail
O=gpurun_out/r04g; mkdir -p $O
timeout -k 10 900 python -m pytest tests/test_gpu_complex.py tests/test_gpu_shard.py -m gpu -x -q > $O/pytest_z.log 2>&1; tail -4 $O/pytest_z.log
PSD_C3=1 timeout -k 10 300 tools/psd_profile 1024 64 2 z > $O/prof_z_c3.log 2>&1; tail -1 $O/prof_z_c3.log
PSD_C3=0 timeout -k 10 300 tools/psd_profile 1024 64 1 z > $O/prof_z_c1.log 2>&1; tail -1 $O/prof_z_c1.log
PSD_C3=1 timeout -k 10 300 tools/psd_profile 512 16 2 z > $O/prof_z512_c3.log 2>&1; tail -1 $O/prof_z512_c3.log
PSD_C3=0 timeout -k 10 300 tools/psd_profile 512 16 2 z > $O/prof_z512_c1.log 2>&1; tail -1 $O/prof_z512_c1.log
