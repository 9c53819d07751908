#!/bin/bash
set -o pipefail
O=gpurun_out/r04uu; mkdir -p $O
PSD_HESS_ASYNC=4 PSD_H2_PIPE=2 python tests/gpu_fuzz.py --seconds 300 --nmax 600 --seed 3 > $O/fuzz_all_nmax600.log 2>&1; tail -1 $O/fuzz_all_nmax600.log | cut -c1-600
