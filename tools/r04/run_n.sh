#!/bin/bash
set -o pipefail
O=gpurun_out/r04n; mkdir -p $O
R=$GRAFT_REPO_ROOT
(cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/prof_cfg4 -o k -- python3 $R/tools/cfg_run.py cfg4 1 > $R/$O/cfg4_run.log 2>&1 < /dev/null)
cp $(find $O/prof_cfg4 -name "*kernel_stats.csv" | head -1) $O/cfg4_signed_kernel_stats.csv; head -12 $O/cfg4_signed_kernel_stats.csv | cut -c1-170
find $O -name "*kernel_trace.csv" -delete
