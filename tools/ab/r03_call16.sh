#!/bin/bash
# GPU call 16: branch-free epilogues (bias row always in LDS, loop variants chosen per launch) against the r03o
# binary: bitwise comparison, gpu tests, same-box A/B; and the runtime's kernarg placement switches.
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out
mkdir -p $OUT
cd $ROOT
OLD=$ROOT/tools/ab/libs/libmsd_amd_r03o.so
{ timeout 200 python tools/diag/lib_bitwise.py 2>/dev/null | tail -1; MSD_AMD_LIB=$OLD timeout 200 python tools/diag/lib_bitwise.py 2>/dev/null | tail -1; } | tee $OUT/r03q_bitwise.log
timeout 600 python -m pytest tests -m gpu -x -q > $OUT/r03q_gpu_tests.log 2>&1; tail -2 $OUT/r03q_gpu_tests.log
bash tools/ab/run_env.sh "MSD_AMD_LIB=$OLD" "X=0" "HIP_FORCE_DEV_KERNARG=1" "HIP_FORCE_DEV_KERNARG=0" "DEBUG_CLR_GRAPH_PACKET_CAPTURE=0" 2>&1 | tee $OUT/r03q_env_ab.log
