#!/bin/bash
# GPU call 19: batched gated-MLP input projection on 256 x 128 eight-wave tiles (gemm_h16_wide.h, MSD_BIG_WIDE=1):
# parity of the batched test, then same-box A/B at 8 / 16 songs per GPU.
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out
mkdir -p $OUT
cd $ROOT
MSD_BIG_WIDE=1 timeout 300 python -m pytest tests/test_gpu_model.py -m gpu -q -s -k "batched" > $OUT/r03t_batched_test.log 2>&1; tail -3 $OUT/r03t_batched_test.log; grep -E "batched B=16" $OUT/r03t_batched_test.log | cut -c1-200
for r in 1 2; do
  for K in 1 0; do
    for nb in 8 16; do
      MSD_BIG_WIDE=$K timeout 200 python bench.py --batch $nb --steps 1 --warmup 1 --no-cpu-baseline --batched-songs 0 --small-segments 0 --profile-steps 1 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('[WIDE=$K songs=$nb]', d['value'], d['ms_per_step'], {k: round(v*1000,1) for k,v in d['roofline']['per_class_ms_per_step'].items()})"
    done
  done
done 2>&1 | tee $OUT/r03t_wide_ab.log
