#!/bin/bash
# GPU call 11 of round 3: the round's final default build (prefetch wave, one query-side plane, hoist / split-K /
# K-V prefetch off): full -m gpu suite, the profiling round (kernel trace + PMC passes on THIS binary), default bench.
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out
mkdir -p $OUT
cd $ROOT
timeout 800 python -m pytest tests -m gpu -q -s > $OUT/r03k_gpu_tests.log 2>&1; tail -3 $OUT/r03k_gpu_tests.log; grep -E "^FAILED" $OUT/r03k_gpu_tests.log | cut -c1-160
bash tools/profile_round.sh r03k > $OUT/r03k_profile.log 2>&1; tail -4 $OUT/r03k_profile.log
timeout 600 python bench.py > $OUT/r03k_bench_default.json 2> $OUT/r03k_bench_default.err; python -c "
import json; d=json.load(open('$OUT/r03k_bench_default.json')); print(d['value'], d['ms_per_step'], d['roofline']['frac'], d['roofline']['whole_step'], d.get('batched',{}).get('value'), d.get('small',{}).get('value'), d['cpu_baseline']['value'])"
