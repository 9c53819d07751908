#!/bin/bash
set -o pipefail
O=gpurun_out/r04ee; mkdir -p $O
for np in "512 16" "1024 16" "256 8" "768 32"; do
for oc in 104 300 400 500 600 800 1200 1600 3200; do
set -- $np
PSD_TRAIN_OC=$oc tools/psd_profile_diag $1 $2 2 > $O/prof_$1x$2_oc$oc.log 2>&1
echo "n $1 p $2 oc $oc $(tail -1 $O/prof_$1x$2_oc$oc.log | cut -c30-140)"
done; done
