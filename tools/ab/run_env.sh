#!/bin/bash
# same-box A/B of launch-time switches with full bench runs: bash tools/ab/run_env.sh "" "VAR=1" ...
B="python bench.py --steps 4 --warmup 1 --no-cpu-baseline --batched-songs 0 --small-segments 0 --profile-steps 1"
for r in 1 2 3; do
  for E in "$@"; do
    env $E timeout 100 $B 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('[$E]', d['value'], round(d['sample_ms_per_segment'],1))"
  done
done
