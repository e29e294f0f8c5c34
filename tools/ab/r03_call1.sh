#!/bin/bash
# GPU call 1 of round 3: -m gpu tests on the new library (range flag, packed splits, distinct precision enums),
# then a same-box A/B of the r02 library (tools/ab/libs/libmsd_amd_r02.so) against it.
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out
mkdir -p $OUT
cd $ROOT
timeout 500 python -m pytest tests -m gpu -q -x -s > $OUT/r03a_gpu_tests.log 2>&1; tail -5 $OUT/r03a_gpu_tests.log
B="python bench.py --steps 4 --warmup 1 --no-cpu-baseline --batched-songs 0 --small-segments 0 --profile-steps 1"
for r in 1 2 3; do
  for L in tools/ab/libs/libmsd_amd_r02.so ""; do
    MSD_AMD_LIB=$L timeout 120 $B 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('[lib=${L:-new}]', d['value'], round(d['sample_ms_per_segment'],1), round(d['encode_ms_per_segment'],2))"
  done
done 2>&1 | tee $OUT/r03a_lib_ab.log
