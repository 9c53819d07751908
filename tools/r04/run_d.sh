#!/bin/bash
set -o pipefail
O=gpurun_out/r04d; mkdir -p $O
R=$GRAFT_REPO_ROOT
PSD_C3=1 python tools/r04/cycles.py 1024 64 2>&1 | grep -v amdgpu.ids | tee $O/cyc_c3.log
PSD_TICKLOG=$O/ticklog_c3.txt PSD_C3=1 tools/psd_profile_diag 1024 64 1 > $O/tl.log 2>&1; tail -3 $O/tl.log
python tools/ticklog_summary.py $O/ticklog_c3.txt | head -8 | tee $O/ticklog_c3_summary.txt
timeout -k 10 600 python -m pytest tests/test_gpu_real.py -m gpu -x -q > $O/pytest_real.log 2>&1; tail -3 $O/pytest_real.log
rm -rf $O/prof
(cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/prof -o c3 -- $R/tools/psd_profile 1024 64 2 > $R/$O/rocprof.log 2>&1 < /dev/null)
f=$(find $O/prof -name "*kernel_stats.csv" | head -1)
if [ -n "$f" ]; then head -9 "$f" | cut -c1-160; fi
find $O/prof -name "*kernel_trace.csv" -delete
