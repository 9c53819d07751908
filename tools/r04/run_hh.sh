#!/bin/bash
set -o pipefail
O=gpurun_out/r04hh; mkdir -p $O
python tools/r04/wdiv.py 2>&1 | grep -v amdgpu.ids | tee $O/wdiv.log
