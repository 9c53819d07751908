#!/bin/bash
# round 4, GPU session KK: links per panel batch and the panel stream's CU share with three chain launches in flight
set -o pipefail
O=gpurun_out/r04kk; mkdir -p $O
run() { tag=$1; shift; echo "$tag $(env "$@" tools/psd_profile_diag 1024 64 2 2>&1 | tail -1 | cut -c95-150)"; }
run base X=0
run K16 PSD_HESS_ASYNC=16
run K32 PSD_HESS_ASYNC=32
run K40 PSD_HESS_ASYNC=40
run cus32 PSD_HESS_CUS=32
run cus96 PSD_HESS_CUS=96
run cus128 PSD_HESS_CUS=128
run base2 X=0
