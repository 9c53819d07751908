#!/bin/bash
# round 4, GPU session DD: long random sweeps on the final build
set -o pipefail
O=gpurun_out/r04dd; mkdir -p $O
PSD_HESS_ASYNC=4 PSD_H2_PIPE=2 python tests/gpu_fuzz.py --seconds 400 --nmax 300 --seed 4242 > $O/fuzz_all_seed4242.log 2>&1; tail -1 $O/fuzz_all_seed4242.log | cut -c1-300
python tests/gpu_fuzz_real.py --seconds 300 --seed 777 > $O/fuzz_real_seed777.log 2>&1; tail -1 $O/fuzz_real_seed777.log | cut -c1-300
python tests/gpu_fuzz_real.py --seconds 250 --nmax 700 --seed 99 > $O/fuzz_real_nmax700.log 2>&1; tail -1 $O/fuzz_real_nmax700.log | cut -c1-300
