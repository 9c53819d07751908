#!/bin/bash
# round 4, GPU session RR: the two-stream / pipe forms of the Hessenberg reduction at p = 16 (links per panel batch)
set -o pipefail
O=gpurun_out/r04rr; mkdir -p $O
run() { tag=$1; n=$2; p=$3; shift 3; echo "$tag $(env "$@" tools/psd_profile $n $p 2 2>&1 | tail -1 | cut -c60-160)"; }
run base_1024x16 1024 16 X=0
run K4_1024x16 1024 16 PSD_HESS_ASYNC=4
run K8_1024x16 1024 16 PSD_HESS_ASYNC=8
run base_512x16 512 16 X=0
run K4_512x16 512 16 PSD_HESS_ASYNC=4
run K8_512x16 512 16 PSD_HESS_ASYNC=8
run base_512x32 512 32 X=0
run K8_512x32 512 32 PSD_HESS_ASYNC=8
run base_768x24 768 24 X=0
run K8_768x24 768 24 PSD_HESS_ASYNC=8
run K12_768x24 768 24 PSD_HESS_ASYNC=12
