#!/bin/bash
set -o pipefail
O=gpurun_out/r04oo; mkdir -p $O
run() { tag=$1; shift; echo "$tag $(env "$@" tools/psd_profile 1024 64 3 2>&1 | tail -1 | cut -c88-150)"; }
run z3 X=0
run z2 PSD_ZSTREAM=2
run z3b X=0
run z2b PSD_ZSTREAM=2
run z2_rdef0 PSD_ZSTREAM=2 PSD_RDEFER=0
