#!/bin/bash
# round 4, GPU session CC: whole GPU tier + smoke on the final build
set -o pipefail
O=gpurun_out/r04cc; mkdir -p $O
timeout -k 10 1100 python -m pytest tests -m gpu -x -q > $O/pytest_gpu.log 2>&1; tail -3 $O/pytest_gpu.log
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | grep -v amdgpu.ids | tail -2
