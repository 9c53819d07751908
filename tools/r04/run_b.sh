#!/bin/bash
# round 4, GPU session B: scan chase (psd_chase3.h) first light
set -o pipefail
O=gpurun_out/r04b; mkdir -p $O
tools/r04/dpp_rol > $O/dpp_rol.log 2>&1; head -c 400 $O/dpp_rol.log; echo
timeout -k 10 300 python tools/micro/repro_case.py > $O/repro.log 2>&1; head -3 $O/repro.log
timeout -k 10 600 python -m pytest tests/test_gpu_real.py -m gpu -x -q > $O/pytest_real.log 2>&1; tail -5 $O/pytest_real.log
PSD_C3=1 timeout -k 10 120 tools/psd_profile 1024 64 3 > $O/prof_c3.log 2>&1; tail -2 $O/prof_c3.log
PSD_C3=0 timeout -k 10 120 tools/psd_profile 1024 64 3 > $O/prof_c2.log 2>&1; tail -1 $O/prof_c2.log
PSD_C3=1 timeout -k 10 120 tools/psd_profile 512 16 3 > $O/prof512_c3.log 2>&1; tail -1 $O/prof512_c3.log
PSD_C3=0 timeout -k 10 120 tools/psd_profile 512 16 3 > $O/prof512_c2.log 2>&1; tail -1 $O/prof512_c2.log
