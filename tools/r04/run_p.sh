#!/bin/bash
# round 4, GPU session P: rows-role deferral (PSD_RDEFER) A/B at the headline and at 512x16, kernel trace, real tests
set -o pipefail
O=gpurun_out/r04p; mkdir -p $O
for i in 1 2; do
PSD_RDEFER=0 tools/psd_profile 1024 64 3 > $O/prof_rdefer0_$i.log 2>&1; tail -1 $O/prof_rdefer0_$i.log
PSD_RDEFER=1 tools/psd_profile 1024 64 3 > $O/prof_rdefer1_$i.log 2>&1; tail -1 $O/prof_rdefer1_$i.log
done
PSD_RDEFER=0 tools/psd_profile 2048 64 2 > $O/prof2048_rdefer0.log 2>&1; tail -1 $O/prof2048_rdefer0.log
PSD_RDEFER=1 tools/psd_profile 2048 64 2 > $O/prof2048_rdefer1.log 2>&1; tail -1 $O/prof2048_rdefer1.log
cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -o k -- tools/psd_profile 1024 64 2 > $O/rocprof.log 2>&1 < /dev/null
f=$(ls $O/prof/*/k_kernel_stats.csv $O/prof/k_kernel_stats.csv 2>/dev/null | head -1)
[ -n "$f" ] && cp "$f" $O/kernel_stats.csv && head -8 $O/kernel_stats.csv | cut -c1-150
t=$(ls $O/prof/*/k_kernel_trace.csv $O/prof/k_kernel_trace.csv 2>/dev/null | head -1)
[ -n "$t" ] && python tools/trace_ticks.py "$t" > $O/trace_ticks_summary.txt 2>&1 < /dev/null; tail -14 $O/trace_ticks_summary.txt
rm -rf $O/prof
timeout -k 10 900 python -m pytest tests/test_gpu_real.py tests/test_gpu_headline.py -m gpu -x -q > $O/pytest_real.log 2>&1; tail -3 $O/pytest_real.log
