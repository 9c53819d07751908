#!/bin/bash
# round 4, GPU session MM: column-role deferral of the ComplexF64 engine (configs[2] A/B, tests, random sweeps)
set -o pipefail
O=gpurun_out/r04mm; mkdir -p $O
for i in 1 2; do
echo "cdefer 1: $(python tools/cfg_run.py cfg3 1 2>&1 | grep -v amdgpu | tail -1 | cut -c1-140)"
echo "cdefer 0: $(PSD_CDEFER=0 python tools/cfg_run.py cfg3 1 2>&1 | grep -v amdgpu | tail -1 | cut -c1-140)"
done | tee $O/cfg3_cdefer.log
timeout -k 10 900 python -m pytest tests/test_gpu_complex.py tests/test_gpu_baseline_configs.py tests/test_gpu_shard.py tests/test_gpu_slices.py -m gpu -x -q > $O/pytest_z.log 2>&1; tail -3 $O/pytest_z.log
PSD_OVERLAP=2 PSD_HESS_ASYNC=4 PSD_H2_PIPE=2 python tests/gpu_fuzz.py --seconds 200 --nmax 300 --seed 77 > $O/fuzz_all_overlap2.log 2>&1; tail -1 $O/fuzz_all_overlap2.log | cut -c1-300
