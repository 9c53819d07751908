#!/bin/bash
# round 4, GPU session H: fuzz runs on the final chase code, counter passes (iteration kernels; Hessenberg panel kernel and chain, once)
set -o pipefail
O=gpurun_out/r04h; mkdir -p $O
R=$GRAFT_REPO_ROOT
python tests/gpu_fuzz_real.py --seconds 120 > $O/fuzz_real.log 2>&1; tail -1 $O/fuzz_real.log | cut -c1-300
python tests/gpu_fuzz_real.py --seconds 150 --seed 777 > $O/fuzz_real_seed777.log 2>&1; tail -1 $O/fuzz_real_seed777.log | cut -c1-300
python tests/gpu_fuzz_real.py --seconds 200 --nmax 700 --seed 99 > $O/fuzz_real_nmax700.log 2>&1; tail -1 $O/fuzz_real_nmax700.log | cut -c1-300
PSD_HESS_ASYNC=4 PSD_H2_PIPE=2 python tests/gpu_fuzz.py --seconds 150 --nmax 200 > $O/fuzz_all.log 2>&1; tail -1 $O/fuzz_all.log | cut -c1-300
cd /tmp && export TMPDIR=/tmp
pmc() {  # name counter regex -- command...
  local name=$1 ctr=$2 rx=$3; shift 3
  rm -rf $R/$O/pmc_$name
  "$@" > /dev/null 2>&1 < /dev/null  # (warm the box's page cache with one plain run)
  rocprofv3 --pmc $ctr --kernel-include-regex "$rx" --output-format csv -d $R/$O/pmc_$name -o p -- "$@" > $R/$O/pmc_$name.log 2>&1 < /dev/null
  local rc=$?
  python3 $R/tools/pmc_summary.py $R/$O/pmc_$name > $R/$O/pmc_$name.json 2>/dev/null
  find $R/$O/pmc_$name -name "*.csv" -size +1M -delete
  echo "pmc $name rc=$rc"; tail -1 $R/$O/pmc_$name.log | cut -c1-200
  return $rc
}
pmc iter_fetch FETCH_SIZE 'psd_rq_(step|apply|band)' $R/tools/psd_profile 1024 64 1 &&
pmc iter_write WRITE_SIZE 'psd_rq_(step|apply|band)' $R/tools/psd_profile 1024 64 1 &&
pmc bulk_fetch FETCH_SIZE 'psd_hess2_bulk' $R/tools/psd_profile 1024 64 1 &&
pmc bulk_write WRITE_SIZE 'psd_hess2_bulk' $R/tools/psd_profile 1024 64 1 &&
PSD_HESS_ASYNC=2 PSD_H2_PIPE=2 pmc link_fetch FETCH_SIZE 'psd_hess2_link' $R/tools/psd_profile 1024 4 1 &&
PSD_HESS_ASYNC=2 PSD_H2_PIPE=2 pmc link_write WRITE_SIZE 'psd_hess2_link' $R/tools/psd_profile 1024 4 1
echo "pmc chain rc=$?"
