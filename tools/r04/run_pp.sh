#!/bin/bash
set -o pipefail
O=gpurun_out/r04pp; mkdir -p $O
for np in "512 16" "1024 16" "512 64"; do set -- $np
PSD_TICKLOG=$O/ticklog_$1x$2.txt tools/psd_profile_diag $1 $2 1 > $O/tl_$1x$2.log 2>&1; echo "== $1 x $2"; tail -3 $O/tl_$1x$2.log | cut -c1-420
python tools/ticklog_summary.py $O/ticklog_$1x$2.txt | head -6
done
