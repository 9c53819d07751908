#!/bin/bash
# round 4, GPU session A: 2x2 guard (repro + scan, both chases), column-role deferral A/B at the headline
set -o pipefail
O=gpurun_out/r04a; mkdir -p $O
python tools/micro/repro_case.py > $O/repro_c2.log 2>&1; tail -3 $O/repro_c2.log | head -1
PSD_C2=0 python tools/micro/repro_case.py > $O/repro_c1.log 2>&1
head -2 $O/repro_c2.log; head -2 $O/repro_c1.log
python tools/r04/scan2x2.py $O/scan_c2.json > $O/scan_c2.log 2>&1; tail -1 $O/scan_c2.log
PSD_C2=0 python tools/r04/scan2x2.py $O/scan_c1.json > $O/scan_c1.log 2>&1; tail -1 $O/scan_c1.log
for i in 1 2; do
PSD_CDEFER=0 tools/psd_profile 1024 64 3 > $O/prof_cdefer0_$i.log 2>&1; tail -1 $O/prof_cdefer0_$i.log
PSD_CDEFER=1 tools/psd_profile 1024 64 3 > $O/prof_cdefer1_$i.log 2>&1; tail -1 $O/prof_cdefer1_$i.log
done
PSD_CDEFER=0 tools/psd_profile 512 16 3 > $O/prof512_cdefer0.log 2>&1; tail -1 $O/prof512_cdefer0.log
PSD_CDEFER=2 tools/psd_profile 512 16 3 > $O/prof512_cdefer2.log 2>&1; tail -1 $O/prof512_cdefer2.log
PSD_CDEFER=2 PSD_OVERLAP=2 tools/psd_profile 512 16 3 > $O/prof512_cdefer2_ovl2.log 2>&1; tail -1 $O/prof512_cdefer2_ovl2.log
timeout -k 10 900 python -m pytest tests/test_gpu_real.py tests/test_gpu_headline.py -m gpu -x -q > $O/pytest_real.log 2>&1; tail -3 $O/pytest_real.log
