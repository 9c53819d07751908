#!/bin/bash
set -o pipefail
O=gpurun_out/r04y; mkdir -p $O
for n in 1536 2048; do python bench.py --order $n --period 64 --steps 1 --warmup 1 --no-cpu-baseline --no-configs > $O/bench_${n}x64.json 2>> $O/bench_default.err; done
python - <<'PY'
import json
for f in ("bench_1536x64", "bench_2048x64"):
    d = json.load(open("gpurun_out/r04y/%s.json" % f))
    print(f, "ms_per_step %.1f" % d["ms_per_step"], d["phase_ms_per_step"], "gate", d["accuracy"]["gate_ok"], "%.0f eps of %.0f" % (d["accuracy"]["checkpsd_max_err_eps"], d["accuracy"]["checkpsd_thresh_eps"]))
PY
PSD_HESS_ASYNC=4 PSD_H2_PIPE=2 timeout -k 10 300 python tests/gpu_fuzz_real.py --seconds 100 --nmax 300 > $O/fuzz_pipe_forced.log 2>&1; tail -1 $O/fuzz_pipe_forced.log | cut -c1-300
timeout -k 10 600 python -m pytest tests/test_gpu_real.py tests/test_gpu_headline.py -m gpu -x -q > $O/pytest_real.log 2>&1; tail -2 $O/pytest_real.log
