#!/bin/bash
# One GPU session of the round: the -m gpu tests, the batched sweep, a kernel-stats profile of the batched leg
# and the profiling round of the headline (kernel trace + PMC passes).  Outputs under gpurun_out/<tag>_*.
TAG=${1:-r02}
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out
mkdir -p $OUT
cd $ROOT
timeout 400 python -m pytest tests -m gpu -q -s > $OUT/${TAG}_gpu_tests.log 2>&1; tail -4 $OUT/${TAG}_gpu_tests.log
grep -A8 "chained song, rms" $OUT/${TAG}_gpu_tests.log
bash tools/batched_sweep.sh 1 2 4 8 16 > $OUT/${TAG}_batched_sweep.jsonl 2>/dev/null
python - <<PY
import json
for line in open('$OUT/${TAG}_batched_sweep.jsonl'):
    d = json.loads(line); print('batched', d['config']['workload'].split(',')[3], d['value'], d['ms_per_step'])
PY
MSD_CROSS_KSPLIT=4 bash tools/batched_sweep.sh 4 8 2>/dev/null | python -c "
import json,sys
for line in sys.stdin:
    d = json.loads(line); print('batched ksplit=4', d['config']['workload'].split(',')[3], d['value'], d['ms_per_step'])"
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/prof_b8
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_b8 -- \
    python $ROOT/bench.py --batch 8 --steps 1 --warmup 1 --no-cpu-baseline --batched-songs 0 --small-segments 0 --profile-steps 1 > $OUT/${TAG}_bench_b8_under_rocprof.json 2>/dev/null
find /tmp/prof_b8 -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $OUT/${TAG}_b8_kernel_stats.csv
head -16 $OUT/${TAG}_b8_kernel_stats.csv | cut -c1-160
cd $ROOT
bash tools/profile_round.sh $TAG > $OUT/${TAG}_profile.log 2>&1; tail -6 $OUT/${TAG}_profile.log
