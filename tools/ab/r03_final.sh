#!/bin/bash
# Final GPU session of round 3 on the shipped binary: full -m gpu suite + smoke (the green log of HEAD), profiling round
# (kernel trace + PMC passes, stamped with the library hash), default bench line, 8-song kernel trace.
TAG=${1:-r03w}
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out
mkdir -p $OUT
cd $ROOT
timeout 800 python -m pytest tests -m gpu -q -s > $OUT/${TAG}_gpu_tests.log 2>&1; tail -3 $OUT/${TAG}_gpu_tests.log; grep -E "^FAILED" $OUT/${TAG}_gpu_tests.log | cut -c1-160
timeout 200 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
bash tools/profile_round.sh $TAG > $OUT/${TAG}_profile.log 2>&1; tail -3 $OUT/${TAG}_profile.log
timeout 600 python bench.py > $OUT/${TAG}_bench_default.json 2> $OUT/${TAG}_bench_default.err; python -c "
import json; d=json.load(open('$OUT/${TAG}_bench_default.json')); print(d['value'], d['ms_per_step'], d['roofline']['frac'], d['roofline']['whole_step'], d.get('batched',{}).get('value'), d.get('small',{}).get('value'), d['cpu_baseline']['value'])"
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/prof_b8
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_b8 -- \
    python $ROOT/bench.py --batch 8 --steps 1 --warmup 1 --no-cpu-baseline --batched-songs 0 --small-segments 0 --profile-steps 1 > $OUT/${TAG}_bench_b8_under_rocprof.json 2>/dev/null
find /tmp/prof_b8 -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $OUT/${TAG}_b8_kernel_stats.csv
head -8 $OUT/${TAG}_b8_kernel_stats.csv | cut -c1-150
