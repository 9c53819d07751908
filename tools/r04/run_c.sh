#!/bin/bash
set -o pipefail
O=gpurun_out/r04c; mkdir -p $O
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/prof -o c3 -- $R/tools/psd_profile 1024 64 2 > $R/$O/rocprof.log 2>&1 < /dev/null
cd $R
ls $O/prof | head
f=$(find $O/prof -name "*kernel_stats.csv" | head -1)
if [ -n "$f" ]; then head -25 "$f" | cut -c1-220; fi
# keep the merge small: drop the raw trace
find $O/prof -name "*kernel_trace.csv" -size +20M -delete
