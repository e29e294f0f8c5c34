#!/bin/bash
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out
cd $ROOT
for L in "" tools/ab/libs/libmsd_amd_pfwave2.so tools/ab/libs/libmsd_amd_pfwave3.so; do
  echo "=== $L"
  MSD_AMD_LIB=$L AMD_LOG_LEVEL=1 timeout 120 python tools/diag/pfwave_probe.py 2>&1 | grep -v "^$" | tail -12
done > $OUT/r03e_pfwave_probe.log 2>&1
cat $OUT/r03e_pfwave_probe.log
