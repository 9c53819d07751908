#!/bin/bash
# round 4, GPU session LL: bulges per train of the ComplexF64 engine at configs[2]
set -o pipefail
O=gpurun_out/r04ll; mkdir -p $O
for m in 16 24 32 48 64; do
echo "cfg3 ztrain $m $(PSD_TRAIN_Z=$m python tools/cfg_run.py cfg3 1 2>&1 | grep -v amdgpu | tail -1 | cut -c1-170)"
done | tee $O/cfg3_ztrain.log
