#!/bin/bash
# GPU call 9 of round 3: prefetch-wave family A/B (both / GEMM only / attention only), then the profiling round of
# the shipped binary (kernel trace + PMC passes), the batched sweep and the 8-song kernel trace.
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out
mkdir -p $OUT
cd $ROOT
B="python bench.py --steps 4 --warmup 1 --no-cpu-baseline --batched-songs 0 --small-segments 0 --profile-steps 1"
for r in 1 2; do
  for E in "MSD_X=1" "MSD_AMD_LIB=tools/ab/libs/libmsd_amd_pfwave2.so" "MSD_AMD_LIB=tools/ab/libs/libmsd_amd_pfwave3.so"; do
    env $E timeout 120 $B 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('[$E]', d['value'], round(d['sample_ms_per_segment'],1), {k: round(v*1000,1) for k,v in d['roofline']['per_class_ms_per_step'].items()})"
  done
done 2>&1 | tee $OUT/r03i_pfwave_family_ab.log
bash tools/batched_sweep.sh 1 2 4 8 16 > $OUT/r03i_batched_sweep.jsonl 2>/dev/null
python - <<PY
import json
for line in open('$OUT/r03i_batched_sweep.jsonl'):
    d = json.loads(line); print('batched', d['config']['workload'].split(',')[3], d['value'], d['ms_per_step'])
PY
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/prof_b8
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_b8 -- \
    python $ROOT/bench.py --batch 8 --steps 1 --warmup 1 --no-cpu-baseline --batched-songs 0 --small-segments 0 --profile-steps 1 > $OUT/r03i_bench_b8_under_rocprof.json 2>/dev/null
find /tmp/prof_b8 -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $OUT/r03i_b8_kernel_stats.csv
head -14 $OUT/r03i_b8_kernel_stats.csv | cut -c1-170
cd $ROOT
bash tools/profile_round.sh r03i > $OUT/r03i_profile.log 2>&1; tail -8 $OUT/r03i_profile.log
