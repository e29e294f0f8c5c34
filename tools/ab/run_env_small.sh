#!/bin/bash
B="python bench.py --preset small --steps 5 --warmup 1 --no-cpu-baseline --batched-songs 0 --small-segments 0 --profile-steps 1"
for r in 1 2; do
  for E in "$@"; do
    env $E timeout 100 $B 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('[$E]', d['value'], round(d['sample_ms_per_segment'],1))"
  done
done
