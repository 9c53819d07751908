#!/bin/bash
# round 4, GPU session EE: the window-width rule's overhead / position ratio (PSD_TRAIN_OC, diagnostic build) under the scan chase
set -o pipefail
O=gpurun_out/r04ee; mkdir -p $O
for np in "512 16" "1024 16" "1024 64"; do
for oc in 30 60 104 200 400 800; do
set -- $np
PSD_TRAIN_OC=$oc tools/psd_profile_diag $1 $2 2 > $O/prof_$1x$2_oc$oc.log 2>&1
echo "n $1 p $2 oc $oc $(tail -1 $O/prof_$1x$2_oc$oc.log | cut -c30-160)"
done; done
