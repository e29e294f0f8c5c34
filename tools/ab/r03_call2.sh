#!/bin/bash
# GPU call 2 of round 3: full -m gpu tests (split-K op test, range tests), then same-box A/B of the launch-time
# switches: split-K MLP-out on/off, query-side single-plane attention (speed only; precision study decides).
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out
mkdir -p $OUT
cd $ROOT
timeout 600 python -m pytest tests -m gpu -q -s > $OUT/r03b_gpu_tests.log 2>&1; tail -5 $OUT/r03b_gpu_tests.log
B="python bench.py --steps 4 --warmup 1 --no-cpu-baseline --batched-songs 0 --small-segments 0 --profile-steps 1"
for r in 1 2; do
  for E in "MSD_SPLITK=1" "MSD_SPLITK=0" "MSD_ATT_QP_SELF=3 MSD_ATT_QP_CROSS=3" "MSD_ATT_QP_SELF=1 MSD_ATT_QP_CROSS=1" "MSD_ATT_QP_SELF=2 MSD_ATT_QP_CROSS=2"; do
    env $E timeout 120 $B 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('[$E]', d['value'], round(d['sample_ms_per_segment'],1), {k: round(v*1000,1) for k,v in d['roofline']['per_class_ms_per_step'].items()})"
  done
done 2>&1 | tee $OUT/r03b_env_ab.log
