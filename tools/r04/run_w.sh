#!/bin/bash
set -o pipefail
O=gpurun_out/r04w; mkdir -p $O
for i in 1 2; do
PSD_RDEFER=0 tools/psd_profile 1024 64 3 > $O/prof_rdefer0_$i.log 2>&1; tail -1 $O/prof_rdefer0_$i.log | cut -c1-220
PSD_RDEFER=1 tools/psd_profile 1024 64 3 > $O/prof_rdefer1_$i.log 2>&1; tail -1 $O/prof_rdefer1_$i.log | cut -c1-220
done
python tools/cfg_run.py cfg3 1 2>&1 | grep -v amdgpu | tail -1 | cut -c1-300
