#!/bin/bash
set -o pipefail
O=gpurun_out/r04i; mkdir -p $O
R=$GRAFT_REPO_ROOT
python tools/r04/resid.py 2>&1 | grep -v amdgpu.ids | tee $O/resid.log
tools/psd_profile 1024 64 1 H /tmp/hess_1024x64.bin | tee $O/hessdump.log
tools/psd_profile 1024 64 1 I /tmp/hess_1024x64.bin | tee -a $O/hessdump.log
cd /tmp && export TMPDIR=/tmp
pmc() {
  local name=$1 ctr=$2 rx=$3; shift 3
  rm -rf $R/$O/pmc_$name
  rocprofv3 --pmc $ctr --kernel-include-regex "$rx" --output-format csv -d $R/$O/pmc_$name -o p -- "$@" > $R/$O/pmc_$name.log 2>&1 < /dev/null
  local rc=$?
  python3 $R/tools/pmc_summary.py $R/$O/pmc_$name > $R/$O/pmc_$name.json 2>/dev/null
  find $R/$O/pmc_$name -name "*.csv" -size +1M -delete
  echo "pmc $name rc=$rc"; tail -1 $R/$O/pmc_$name.log | cut -c1-200
  return $rc
}
pmc iter_fetch FETCH_SIZE 'psd_rq_(step|apply|band)' $R/tools/psd_profile 1024 64 1 I /tmp/hess_1024x64.bin &&
pmc iter_write WRITE_SIZE 'psd_rq_(step|apply|band)' $R/tools/psd_profile 1024 64 1 I /tmp/hess_1024x64.bin &&
PSD_HESS_ASYNC=0 pmc link1s_fetch FETCH_SIZE 'psd_hess2' $R/tools/psd_profile 1024 16 1 H /tmp/h16.bin &&
PSD_HESS_ASYNC=0 pmc link1s_write WRITE_SIZE 'psd_hess2' $R/tools/psd_profile 1024 16 1 H /tmp/h16.bin
echo "pmc chain rc=$?"
