#!/bin/bash
# round 4, GPU session II: schedule defaults by period (width rule, bulges per train): probes, tests, random sweeps
set -o pipefail
O=gpurun_out/r04ii; mkdir -p $O
python tools/r04/wdiv.py 2>&1 | grep -v amdgpu.ids | tee $O/wdiv_final.log | cut -c1-200
timeout -k 10 900 python -m pytest tests/test_gpu_real.py tests/test_gpu_headline.py tests/test_gpu_baseline_configs.py tests/test_gpu_slices.py tests/test_gpu_shard.py -m gpu -x -q > $O/pytest_real.log 2>&1; tail -2 $O/pytest_real.log
python tests/gpu_fuzz_real.py --seconds 200 > $O/fuzz_real.log 2>&1; tail -1 $O/fuzz_real.log | cut -c1-300
python tests/gpu_fuzz_real.py --seconds 200 --nmax 700 --seed 99 > $O/fuzz_real_nmax700.log 2>&1; tail -1 $O/fuzz_real_nmax700.log | cut -c1-300
python tests/gpu_fuzz_real.py --seconds 150 --seed 777 > $O/fuzz_real_seed777.log 2>&1; tail -1 $O/fuzz_real_seed777.log | cut -c1-300
