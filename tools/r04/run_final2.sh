#!/bin/bash
# round 4: the GPU test tier, then the records job, on the final code
set -o pipefail
O=gpurun_out/r04y; mkdir -p $O
timeout -k 10 1000 python -m pytest tests -m gpu -x -q > $O/pytest_gpu.log 2>&1; tail -3 $O/pytest_gpu.log
bash tools/r04/run_final.sh
