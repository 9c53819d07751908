#!/bin/bash
# round 4, GPU session AA: slices clamp test; what the leaders' launches consist of now (tick log of the diagnostic build)
set -o pipefail
O=gpurun_out/r04aa; mkdir -p $O
timeout -k 10 600 python -m pytest tests/test_gpu_slices.py -m gpu -x -q > $O/pytest_slices.log 2>&1; tail -3 $O/pytest_slices.log
PSD_TICKLOG=$O/ticklog.txt tools/psd_profile_diag 1024 64 1 > $O/tl.log 2>&1; tail -4 $O/tl.log | cut -c1-600
python tools/ticklog_summary.py $O/ticklog.txt | head -12 | tee $O/ticklog_summary.txt
