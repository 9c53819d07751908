#!/bin/bash
# round 4, GPU session SS: batches of 8 links for periods 24 .. 47: sizes, tests, random sweep with periods up to 40
set -o pipefail
O=gpurun_out/r04ss; mkdir -p $O
for np in "768 24" "512 32" "1024 40" "600 47"; do set -- $np
echo "n $1 p $2 $(tools/psd_profile $1 $2 2 2>&1 | tail -1 | cut -c60-160)"
done
timeout -k 10 700 python -m pytest tests/test_gpu_real.py tests/test_gpu_headline.py tests/test_gpu_shard.py -m gpu -x -q > $O/pytest_real.log 2>&1; tail -2 $O/pytest_real.log
python tests/gpu_fuzz_real.py --seconds 200 --nmax 700 --seed 5 > $O/fuzz_real_nmax700_seed5.log 2>&1; tail -1 $O/fuzz_real_nmax700_seed5.log | cut -c1-300
