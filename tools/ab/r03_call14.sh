#!/bin/bash
# final session on the reproducibly built binary + the phase-timestamp reading of the debug build
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
cd $ROOT
bash tools/ab/r03_final.sh r03o
cd $ROOT
MSD_AMD_LIB=$ROOT/tools/ab/libs/libmsd_amd_ts.so timeout 300 python tools/diag/phase_times.py > gpurun_out/r03o_phase_times.txt 2>&1
tail -60 gpurun_out/r03o_phase_times.txt
