#!/bin/bash
# round 4, GPU session BB: near / far boundaries of the deferred roles (does stream 1 stop waiting for stream 2?)
set -o pipefail
O=gpurun_out/r04bb; mkdir -p $O
run() { tag=$1; shift; env "$@" tools/psd_profile 1024 64 3 > $O/prof_$tag.log 2>&1; echo "$tag $(tail -1 $O/prof_$tag.log | cut -c88-150)"; }
run base PSD_X=0
run r96 PSD_RDEFER_EDGE=96
run r160 PSD_RDEFER_EDGE=160
run r256 PSD_RDEFER_EDGE=256
run r400 PSD_RDEFER_EDGE=400
run c64 PSD_CDEFER_EDGE=64
run c128 PSD_CDEFER_EDGE=128
run r160c64 PSD_RDEFER_EDGE=160 PSD_CDEFER_EDGE=64
run r256c128 PSD_RDEFER_EDGE=256 PSD_CDEFER_EDGE=128
run r400c200 PSD_RDEFER_EDGE=400 PSD_CDEFER_EDGE=200
run base2 PSD_X=0
run rdefer0 PSD_RDEFER=0
