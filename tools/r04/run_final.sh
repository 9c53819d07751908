#!/bin/bash
# round 4: records for profiles/r04 on the final code
set -o pipefail
O=gpurun_out/r04y; mkdir -p $O
R=$GRAFT_REPO_ROOT
python bench.py > $O/bench_default.json 2> $O/bench_default.err; tail -c 300 $O/bench_default.err
python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_driver_command.json 2> $O/bench_driver.err; tail -c 300 $O/bench_driver.err
python - <<'PY'
import json
for f in ("bench_default", "bench_driver_command"):
    d = json.load(open("gpurun_out/r04y/%s.json" % f))
    print(f, "ms_per_step %.1f" % d["ms_per_step"], d["phase_ms_per_step"], "tick frac %.3f" % d["roofline"]["frac"], "whole-call GB/s %.0f" % d["algorithmic_GBps"]["whole_call"], "gate", d["accuracy"]["gate_ok"], "%.0f eps" % d["accuracy"]["checkpsd_max_err_eps"])
    if "configs" in d:
        for k, v in d["configs"].items(): print("  ", k, "%.1f ms" % v["ms"], v.get("phase_ms"), v["gate_ok"], "%.0f eps" % v["checkpsd_max_err_eps"], v.get("swaps"), v.get("residual_over_oracle"))
PY
python bench.py --dtype c128 --steps 1 --warmup 1 --no-cpu-baseline --no-configs > $O/bench_c128_1024x64.json 2>> $O/bench_default.err
for n in 512 1536 2048; do python bench.py --order $n --period 64 --steps 1 --warmup 1 --no-cpu-baseline --no-configs > $O/bench_${n}x64.json 2>> $O/bench_default.err; done
python - <<'PY'
import json
for f in ("bench_c128_1024x64", "bench_512x64", "bench_1536x64", "bench_2048x64"):
    d = json.load(open("gpurun_out/r04y/%s.json" % f))
    print(f, "ms_per_step %.1f" % d["ms_per_step"], d["phase_ms_per_step"], "tick frac %.3f" % d["roofline"]["frac"], "hess link frac %.3f" % d["roofline"]["hessenberg_link"]["frac"], "whole %.0f GB/s" % d["algorithmic_GBps"]["whole_call"], "gate", d["accuracy"]["gate_ok"], "%.0f eps of %.0f" % (d["accuracy"]["checkpsd_max_err_eps"], d["accuracy"]["checkpsd_thresh_eps"]))
PY
(cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/prof_bench -o k -- python3 $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-configs > $R/$O/bench_under_rocprof_1024x64.json 2> $R/$O/rocprof_bench.log < /dev/null)
cp $(find $O/prof_bench -name "*kernel_stats.csv" | head -1) $O/kernel_stats_1024x64.csv; python tools/trace_ticks.py $(find $O/prof_bench -name "*kernel_trace.csv" | head -1) > $O/trace_ticks_summary.txt 2>&1; head -12 $O/kernel_stats_1024x64.csv | cut -c1-150
(cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/prof_cfg3 -o k -- python3 $R/tools/cfg_run.py cfg3 1 > $R/$O/cfg3_run.log 2>&1 < /dev/null)
cp $(find $O/prof_cfg3 -name "*kernel_stats.csv" | head -1) $O/cfg3_complex_kernel_stats.csv; head -8 $O/cfg3_complex_kernel_stats.csv | cut -c1-150
find $O -name "*kernel_trace.csv" -delete
(cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/prof_cfg4 -o k -- python3 $R/tools/cfg_run.py cfg4 1 > $R/$O/cfg4_run.log 2>&1 < /dev/null)
cp $(find $O/prof_cfg4 -name "*kernel_stats.csv" | head -1) $O/cfg4_signed_kernel_stats.csv; head -8 $O/cfg4_signed_kernel_stats.csv | cut -c1-150
find $O -name "*kernel_trace.csv" -delete
