#!/bin/bash
# GPU call 10 of round 3: two-stage hoisted query projection (first half on the QKV launch's idle CUs, second half
# with the out-projection): parity, then same-box A/B against MSD_HOIST_Q=0.
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out
mkdir -p $OUT
cd $ROOT
timeout 600 python -m pytest tests/test_gpu_model.py tests/test_ref_golden.py tests/test_gpu_fused_ops.py -m gpu -q > $OUT/r03j_tests.log 2>&1; tail -3 $OUT/r03j_tests.log; grep -E "^FAILED" $OUT/r03j_tests.log | cut -c1-160
B="python bench.py --steps 4 --warmup 1 --no-cpu-baseline --batched-songs 0 --small-segments 0 --profile-steps 1"
for r in 1 2 3; do
  for E in "MSD_HOIST_Q=1" "MSD_HOIST_Q=0"; do
    env $E timeout 120 $B 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('[$E]', d['value'], round(d['sample_ms_per_segment'],1), {k: round(v*1000,1) for k,v in d['roofline']['per_class_ms_per_step'].items()})"
  done
done 2>&1 | tee $OUT/r03j_hoist_ab.log
