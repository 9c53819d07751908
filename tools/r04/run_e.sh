#!/bin/bash
set -o pipefail
O=gpurun_out/r04e; mkdir -p $O
for w in 8 4 6 12 16; do
echo "WDIV=$w"; PSD_TRAIN_WDIV=$w tools/psd_profile 1024 64 2 2>&1 | tail -1 | cut -c1-200 | tee -a $O/wdiv.log
done
echo "512x16"; tools/psd_profile 512 16 2 2>&1 | tail -1 | cut -c1-200 | tee -a $O/wdiv.log
PSD_TRAIN_WDIV=4 tools/psd_profile 512 16 2 2>&1 | tail -1 | cut -c1-200 | tee -a $O/wdiv.log
PSD_C3=1 python tools/r04/cycles.py 1024 64 2>&1 | grep -v amdgpu.ids | tee $O/cyc_c3.log
PSD_C3=1 python tools/r04/cycles.py 512 16 2>&1 | grep -v amdgpu.ids | tee -a $O/cyc_c3.log
PSD_C3=0 python tools/r04/cycles.py 512 16 2>&1 | grep -v amdgpu.ids | tee -a $O/cyc_c3.log
