#!/bin/bash
# round 4, GPU session GG: width rule with the overhead ratio growing with the active width: sizes, tests, fuzz
set -o pipefail
O=gpurun_out/r04gg; mkdir -p $O
for np in "256 8" "512 16" "1024 16" "768 32" "1024 64" "1536 16" "2048 8"; do
set -- $np
tools/psd_profile $1 $2 2 > $O/prof_$1x$2.log 2>&1
echo "n $1 p $2 $(tail -1 $O/prof_$1x$2.log | cut -c30-140)"
done
timeout -k 10 700 python -m pytest tests/test_gpu_real.py tests/test_gpu_headline.py tests/test_gpu_baseline_configs.py -m gpu -x -q > $O/pytest_real.log 2>&1; tail -2 $O/pytest_real.log
python tests/gpu_fuzz_real.py --seconds 120 > $O/fuzz_real.log 2>&1; tail -1 $O/fuzz_real.log | cut -c1-300
python tests/gpu_fuzz_real.py --seconds 150 --nmax 700 --seed 99 > $O/fuzz_real_nmax700.log 2>&1; tail -1 $O/fuzz_real_nmax700.log | cut -c1-300
