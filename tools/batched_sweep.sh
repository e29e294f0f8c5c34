#!/bin/bash
# songs-per-GPU sweep of the batched path (same kernels, M = 2 * songs * 256 rows) -> one JSON line per batch size
# usage: bash tools/batched_sweep.sh > gpurun_out/<tag>_batched_sweep.jsonl
for nb in ${@:-1 2 4 8 16}; do
  timeout 200 python bench.py --batch $nb --steps 1 --warmup 1 --no-cpu-baseline --batched-songs 0 --small-segments 0 --profile-steps 1 2>/dev/null | tail -1
done
