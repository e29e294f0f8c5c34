#!/bin/bash
# same-box comparison of runtime switches: each argument is an env assignment list ("" = defaults)
# usage: bash tools/env_ab.sh "" "MSD_CROSS_KSPLIT=3" "MSD_XCD_ROWS=1" ...
for r in 1 2; do
  for E in "$@"; do
    env $E python bench.py --steps 2 --warmup 1 --no-cpu-baseline --batched-songs 0 --small-segments 0 --profile-steps 1 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('[$E]', d['value'], d['ms_per_step'])"
  done
done
