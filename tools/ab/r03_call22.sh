#!/bin/bash
# GPU call 22: final session of round 3 on the shipped binary (tag r03w) + one more runtime switch
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
cd $ROOT
bash tools/ab/r03_final.sh r03w
cd $ROOT
for r in 1 2; do
  for E in "X=0" "AMD_OPT_FLUSH=0" "AMD_OPT_FLUSH=1"; do
    env $E timeout 100 python bench.py --steps 4 --warmup 1 --no-cpu-baseline --batched-songs 0 --small-segments 0 --profile-steps 1 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('[$E]', d['value'], round(d['sample_ms_per_segment'],1))"
  done
done 2>&1 | tee gpurun_out/r03w_env_ab.log
