#!/bin/bash
# GPU call 6 of round 3: the prefetch-wave build (fixed destination-register hazard) -- parity, then same-box A/B
# against the in-epilogue prefetch, with / without the hoisted query projection (QP = 3 is the default now).
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out
mkdir -p $OUT
cd $ROOT
MSD_AMD_LIB=tools/ab/libs/libmsd_amd_pfwave.so timeout 400 python -m pytest tests/test_gpu_model.py tests/test_ref_golden.py -m gpu -q > $OUT/r03f_pfwave_tests.log 2>&1; tail -3 $OUT/r03f_pfwave_tests.log
B="python bench.py --steps 4 --warmup 1 --no-cpu-baseline --batched-songs 0 --small-segments 0 --profile-steps 1"
for r in 1 2; do
  for E in "MSD_HOIST_Q=0" "MSD_HOIST_Q=0 MSD_AMD_LIB=tools/ab/libs/libmsd_amd_pfwave.so" "MSD_HOIST_Q=1" "MSD_HOIST_Q=1 MSD_AMD_LIB=tools/ab/libs/libmsd_amd_pfwave.so"; do
    env $E timeout 120 $B 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('[$E]', d['value'], round(d['sample_ms_per_segment'],1), {k: round(v*1000,1) for k,v in d['roofline']['per_class_ms_per_step'].items()})"
  done
done 2>&1 | tee $OUT/r03f_env_ab.log
