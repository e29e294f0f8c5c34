#!/bin/bash
# GPU call 3 of round 3: full -m gpu tests with the hoisted query projection on (default); the golden runs (small
# 1000 steps, 12-segment chain) under the query-side single-plane attention switches; same-box A/B of the switches.
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out
mkdir -p $OUT
cd $ROOT
timeout 700 python -m pytest tests -m gpu -q -s > $OUT/r03c_gpu_tests.log 2>&1; tail -4 $OUT/r03c_gpu_tests.log
for QP in 1 2 3; do
  MSD_ATT_QP_SELF=$QP MSD_ATT_QP_CROSS=$QP timeout 300 python -m pytest tests/test_golden.py -m gpu -q -s -k "f16x3 or chain or every_segment or trained" > $OUT/r03c_golden_qp$QP.log 2>&1
  echo "== QP=$QP"; grep -E "rms|passed|failed|segment" $OUT/r03c_golden_qp$QP.log | tail -30
done
B="python bench.py --steps 4 --warmup 1 --no-cpu-baseline --batched-songs 0 --small-segments 0 --profile-steps 1"
for r in 1 2; do
  for E in "MSD_HOIST_Q=1" "MSD_HOIST_Q=0" "MSD_ATT_QP_SELF=1 MSD_ATT_QP_CROSS=1" "MSD_ATT_QP_SELF=3 MSD_ATT_QP_CROSS=3" "MSD_SPLITK=1"; do
    env $E timeout 120 $B 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('[$E]', d['value'], round(d['sample_ms_per_segment'],1), {k: round(v*1000,1) for k,v in d['roofline']['per_class_ms_per_step'].items()})"
  done
done 2>&1 | tee $OUT/r03c_env_ab.log
