#!/bin/bash
set -o pipefail
O=gpurun_out/r04y; mkdir -p $O
python bench.py > $O/bench_default.json 2> $O/bench_default.err; tail -c 200 $O/bench_default.err
python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_driver_command.json 2> $O/bench_driver.err
python - <<'PY'
import json
for f in ("bench_default", "bench_driver_command"):
    d = json.load(open("gpurun_out/r04y/%s.json" % f))
    print(f, "ms_per_step %.1f" % d["ms_per_step"], d["phase_ms_per_step"], "frac %.3f" % d["roofline"]["frac"], "whole %.0f" % d["algorithmic_GBps"]["whole_call"], d["accuracy"]["gate_ok"])
    for k, v in d["configs"].items(): print("  ", k, "%.1f ms" % v["ms"], v["gate_ok"])
PY
