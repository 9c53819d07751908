#!/bin/bash
# round 4, GPU session TT: fewer bulges per train for short periods at large orders: residual probes and the two large-order sweeps
set -o pipefail
O=gpurun_out/r04tt; mkdir -p $O
python tools/r04/wdiv.py 2>&1 | grep -v amdgpu.ids | tee $O/wdiv_p_lt_12.log | cut -c1-200
python tests/gpu_fuzz_real.py --seconds 220 --nmax 700 --seed 5 > $O/fuzz_real_nmax700_seed5.log 2>&1; tail -1 $O/fuzz_real_nmax700_seed5.log | cut -c1-400
python tests/gpu_fuzz_real.py --seconds 220 --nmax 700 --seed 99 > $O/fuzz_real_nmax700_seed99.log 2>&1; tail -1 $O/fuzz_real_nmax700_seed99.log | cut -c1-400
