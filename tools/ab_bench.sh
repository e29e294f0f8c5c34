#!/bin/bash
# A/B two builds of libmsd_amd.so on the SAME box (box-to-box variance is ~2 %): alternates them.
# usage: bash tools/ab_bench.sh <libA.so> <libB.so> [rounds]
A=$1; B=$2; R=${3:-3}
LIB=music-spectrogram-diffusion_amd/csrc/libmsd_amd.so
cp $LIB /tmp/lib_keep.so
for r in $(seq $R); do
  for L in $A $B; do
    cp $L $LIB
    python bench.py --steps 2 --warmup 1 --no-cpu-baseline --batched-songs 0 --small-segments 0 --profile-steps 1 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$L', d['value'], d['ms_per_step'])"
  done
done
cp /tmp/lib_keep.so $LIB
