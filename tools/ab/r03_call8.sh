#!/bin/bash
# GPU call 8 of round 3: the full -m gpu suite on the round's default build, smoke(), and the default bench line
# (headline + batched + small legs + CPU baseline).
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out
mkdir -p $OUT
cd $ROOT
timeout 800 python -m pytest tests -m gpu -q -s > $OUT/r03h_gpu_tests.log 2>&1; tail -3 $OUT/r03h_gpu_tests.log
timeout 200 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
timeout 600 python bench.py > $OUT/r03h_bench_default.json 2> $OUT/r03h_bench_default.err; tail -c 1500 $OUT/r03h_bench_default.json
