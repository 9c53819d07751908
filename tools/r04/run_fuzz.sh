#!/bin/bash
set -o pipefail
O=gpurun_out/r04z; mkdir -p $O
python tests/gpu_fuzz_real.py --seconds 120 > $O/fuzz_real.log 2>&1; tail -1 $O/fuzz_real.log | cut -c1-250
python tests/gpu_fuzz_real.py --seconds 150 --seed 777 > $O/fuzz_real_seed777.log 2>&1; tail -1 $O/fuzz_real_seed777.log | cut -c1-250
python tests/gpu_fuzz_real.py --seconds 200 --nmax 700 --seed 99 > $O/fuzz_real_nmax700.log 2>&1; tail -1 $O/fuzz_real_nmax700.log | cut -c1-250
PSD_HESS_ASYNC=4 PSD_H2_PIPE=2 python tests/gpu_fuzz.py --seconds 150 --nmax 200 > $O/fuzz_all.log 2>&1; tail -1 $O/fuzz_all.log | cut -c1-250
