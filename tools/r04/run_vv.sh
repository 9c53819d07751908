#!/bin/bash
set -o pipefail
O=gpurun_out/r04vv; mkdir -p $O
PSD_HESS_ASYNC=4 PSD_H2_PIPE=2 python tests/gpu_fuzz.py --seconds 500 --nmax 250 --seed 11 > $O/fuzz_all_seed11.log 2>&1; tail -1 $O/fuzz_all_seed11.log | cut -c1-400
python tests/gpu_fuzz_real.py --seconds 400 --seed 2024 > $O/fuzz_real_seed2024.log 2>&1; tail -1 $O/fuzz_real_seed2024.log | cut -c1-400
