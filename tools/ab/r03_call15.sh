#!/bin/bash
# phase timestamps (debug build): one song and 8 songs per handle
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
cd $ROOT
mkdir -p gpurun_out
MSD_AMD_LIB=$ROOT/tools/ab/libs/libmsd_amd_ts.so timeout 300 python tools/diag/phase_times.py > gpurun_out/r03p_phase_times_b1.txt 2>&1
BATCH=8 MSD_AMD_LIB=$ROOT/tools/ab/libs/libmsd_amd_ts.so timeout 300 python tools/diag/phase_times.py > gpurun_out/r03p_phase_times_b8.txt 2>&1
tail -5 gpurun_out/r03p_phase_times_b1.txt gpurun_out/r03p_phase_times_b8.txt
