#!/bin/bash
set -o pipefail
O=gpurun_out/r04f; mkdir -p $O
timeout -k 10 1000 python -m pytest tests -m gpu -x -q > $O/pytest_gpu.log 2>&1; tail -4 $O/pytest_gpu.log
python bench.py --steps 3 --warmup 1 > $O/bench.json 2> $O/bench.err; tail -c 600 $O/bench.err; python - <<'PY'
import json
d=json.load(open("gpurun_out/r04f/bench.json"))
print("ms_per_step", d["ms_per_step"], d["phase_ms_per_step"], "frac", d["roofline"]["frac"], "gate", d["accuracy"]["gate_ok"], d["accuracy"]["checkpsd_max_err_eps"])
for k,v in d["configs"].items(): print(k, round(v["ms"],1), v.get("phase_ms"), v["gate_ok"], round(v["checkpsd_max_err_eps"],1), v.get("swaps"), v.get("residual_over_oracle"))
PY
