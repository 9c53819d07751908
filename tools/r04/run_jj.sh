#!/bin/bash
# round 4, GPU session JJ: the signed engine's width-rule ratio under the scan sweep (configs[3]); the complex engine's at p = 16
set -o pipefail
O=gpurun_out/r04jj; mkdir -p $O
for oc in 25 50 100 200 400 800; do
echo "cfg4 oc $oc $(PSD_DIAG_LIB=1 PSD_TRAIN_OC=$oc python tools/cfg_run.py cfg4 1 2>&1 | grep -v amdgpu | tail -1 | cut -c1-150)"
done | tee $O/cfg4_oc.log
for oc in 50 100 200 400 800; do
echo "z 512x16 oc $oc $(PSD_TRAIN_OC=$oc tools/psd_profile_diag 512 16 2 z 2>&1 | tail -1 | cut -c30-150)"
done | tee $O/z_oc.log
