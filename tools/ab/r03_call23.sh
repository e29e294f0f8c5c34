#!/bin/bash
# GPU call 23: ring prologue without its last tile (issued from the first half step) + scan index read by an early
# scalar load: bitwise comparison and same-box A/B against the r03w binary, then the final session on the new binary.
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out
mkdir -p $OUT
cd $ROOT
OLD=$ROOT/tools/ab/libs/libmsd_amd_r03w.so
{ timeout 200 python tools/diag/lib_bitwise.py 2>/dev/null | tail -1; MSD_AMD_LIB=$OLD timeout 200 python tools/diag/lib_bitwise.py 2>/dev/null | tail -1; } | tee $OUT/r03x_bitwise.log
bash tools/ab/run_env.sh "MSD_AMD_LIB=$OLD" "X=0" 2>&1 | tee $OUT/r03x_lib_ab.log
bash tools/ab/r03_final.sh r03x
