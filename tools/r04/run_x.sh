#!/bin/bash
# round 4, GPU session X: three chain launches in flight in the Hessenberg pipe form (PSD_H2_DEPTH=3)
set -o pipefail
O=gpurun_out/r04x; mkdir -p $O
PSD_H2_DEPTH=3 timeout -k 10 600 python -m pytest tests/test_gpu_real.py -m gpu -x -q -k "hess" > $O/pytest_hess.log 2>&1; tail -3 $O/pytest_hess.log
for i in 1 2; do
PSD_H2_DEPTH=2 tools/psd_profile 1024 64 3 > $O/prof_d2_$i.log 2>&1; tail -1 $O/prof_d2_$i.log | cut -c1-200
PSD_H2_DEPTH=3 tools/psd_profile 1024 64 3 > $O/prof_d3_$i.log 2>&1; tail -1 $O/prof_d3_$i.log | cut -c1-200
done
PSD_H2_DEPTH=3 PSD_HESS_ASYNC=4 PSD_H2_PIPE=2 timeout -k 10 300 python tests/gpu_fuzz_real.py --seconds 60 --nmax 300 > $O/fuzz_d3.log 2>&1; tail -1 $O/fuzz_d3.log | cut -c1-300
