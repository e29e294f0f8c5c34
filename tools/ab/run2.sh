#!/bin/bash
LIB=music-spectrogram-diffusion_amd/csrc/libmsd_amd.so
B="python bench.py --steps 4 --warmup 1 --no-cpu-baseline --batched-songs 0 --profile-steps 1"
show() { python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$1', d['value'], round(d['sample_ms_per_segment'],1))"; }
for r in 1 2 3; do
  for v in prev new; do cp tools/ab/lib_$v.so $LIB; timeout 100 $B 2>/dev/null | show "$v"; done
done
cp tools/ab/lib_new.so $LIB

