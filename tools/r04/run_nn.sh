#!/bin/bash
set -o pipefail
O=gpurun_out/r04mm; mkdir -p $O
for i in 1 2; do
echo "cdefer 1: $(python tools/cfg_run.py cfg3 1 2>&1 | grep -v amdgpu | tail -1 | cut -c1-140)"
echo "cdefer 0: $(PSD_CDEFER=0 python tools/cfg_run.py cfg3 1 2>&1 | grep -v amdgpu | tail -1 | cut -c1-140)"
done | tee $O/cfg3_cdefer.log
