#!/bin/bash
# round 4, GPU session V: the whole GPU test tier and the random sweeps on the code with the pipelined ordschur! drivers
# and the scan form of the signed sweep
set -o pipefail
O=gpurun_out/r04v; mkdir -p $O
timeout -k 10 1100 python -m pytest tests -m gpu -x -q > $O/pytest_gpu.log 2>&1; tail -4 $O/pytest_gpu.log
PSD_HESS_ASYNC=4 PSD_H2_PIPE=2 python tests/gpu_fuzz.py --seconds 200 --nmax 200 > $O/fuzz_all.log 2>&1; tail -1 $O/fuzz_all.log | cut -c1-300
python tests/gpu_fuzz.py --seconds 120 --nmax 300 --seed 31 > $O/fuzz_all_seed31.log 2>&1; tail -1 $O/fuzz_all_seed31.log | cut -c1-300
python tests/gpu_fuzz_real.py --seconds 100 > $O/fuzz_real.log 2>&1; tail -1 $O/fuzz_real.log | cut -c1-300
