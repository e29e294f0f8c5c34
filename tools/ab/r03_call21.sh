#!/bin/bash
# GPU call 21: phase stamps of the loader-wave tile at 8 songs per handle (debug build)
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
cd $ROOT
mkdir -p gpurun_out
BATCH=8 MSD_BIG_LS=1 MSD_AMD_LIB=$ROOT/tools/ab/libs/libmsd_amd_ts.so timeout 300 python tools/diag/phase_times.py > gpurun_out/r03v_phase_times_b8_ls.txt 2>&1
grep -A14 "loader waves" gpurun_out/r03v_phase_times_b8_ls.txt | tail -18
