#!/bin/bash
# Round-5 GPU sessions (gpurun -- 'bash tools/ab/r05_session.sh <what> <tag>').
#   first   new GPU tests (key splits, layer-0 de-duplication, freed staging copies) + the launch-overlap micro-benchmark
#           + in-process knob A/Bs (tools/ab/knob_ab.py) + the stripped library against round 4's shipped binary
#   tests   full -m gpu suite
#   final   tests + smoke + tools/profile_round.sh + default bench on the shipped binary
WHAT=${1:-first}; TAG=${2:-r05a}
ROOT=${GRAFT_REPO_ROOT:-/root/repo}; OUT=$ROOT/gpurun_out; mkdir -p $OUT; cd $ROOT
B="python bench.py --steps 4 --warmup 1 --no-cpu-baseline --batched-songs 0 --small-segments 0 --profile-steps 1"
one() {  # label, env assignments...
  local label=$1; shift
  env "$@" timeout 150 $B 2>$OUT/${TAG}_err.tmp | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('[$label]', d['value'], round(d['sample_ms_per_segment'],1))" || { echo "[$label] FAILED"; tail -5 $OUT/${TAG}_err.tmp; }
}
case $WHAT in
first)
  timeout 900 python -m pytest tests/test_gpu_model.py -m gpu -q -x -k "key_split or dedup or staging or single_decoder_pass or explicit_noise or batched_songs" > $OUT/${TAG}_new_tests.log 2>&1; tail -5 $OUT/${TAG}_new_tests.log
  timeout 120 tools/ubench/launch_overlap_0 > $OUT/${TAG}_launch_overlap.log 2>&1; cat $OUT/${TAG}_launch_overlap.log
  timeout 300 python tools/ab/knob_ab.py --rounds 5 --json $OUT/${TAG}_dedup_ab.json 'dedup_layer0=False' 'dedup_layer0=None' 2>&1 | grep -v Warning | tee $OUT/${TAG}_dedup_ab.log
  timeout 600 python tools/ab/knob_ab.py --steps 500 --rounds 3 --json $OUT/${TAG}_split_sweep.json --tokens 128 --tokens 400 --tokens 700 --tokens 1000 --tokens 1300 --tokens 1536 \
      'cross_key_split=4' 'cross_key_split=1' 'cross_key_split=2' 'cross_key_split=8' 2>&1 | grep -v Warning | tee $OUT/${TAG}_split_sweep.log
  for r in 1 2; do
    one "r04z (round 4's shipped binary)" MSD_AMD_LIB=$ROOT/tools/ab/libs/libmsd_amd_r04z.so
    one "new (library defaults)" X=0
  done 2>&1 | tee $OUT/${TAG}_lib_ab.log
  ;;
second)  # layer-0 de-duplication diagnostic; K / V touch-ahead: parity tests, then in-process A/B at one song and at 8 songs
  timeout 120 python tools/diag/dedup_diff.py tiny_context 1 2>&1 | grep -v Warning | tee $OUT/${TAG}_dedup_diff.log
  timeout 120 python tools/diag/dedup_diff.py base_with_context 1 2>&1 | grep -v Warning | tee -a $OUT/${TAG}_dedup_diff.log
  timeout 600 python -m pytest tests/test_gpu_model.py tests/test_golden.py -m gpu -q -x -k "key_split or base_size or base_with_context_1000 or batched_songs or sum_cross" > $OUT/${TAG}_touch_tests.log 2>&1; tail -5 $OUT/${TAG}_touch_tests.log
  timeout 400 python tools/ab/knob_ab.py --rounds 4 --json $OUT/${TAG}_touch_ab.json 'kv_touch_ahead=0' 'kv_touch_ahead=2' 'kv_touch_ahead=4' 'kv_touch_ahead=8' 2>&1 | grep -v Warning | tee $OUT/${TAG}_touch_ab.log
  timeout 400 python tools/ab/knob_ab.py --batch 8 --steps 200 --rounds 3 --json $OUT/${TAG}_touch_ab_b8.json 'kv_touch_ahead=0' 'kv_touch_ahead=2' 'kv_touch_ahead=4' 'kv_touch_ahead=8' 2>&1 | grep -v Warning | tee $OUT/${TAG}_touch_ab_b8.log
  ;;
third)   # where the de-duplicated step parts ways (debug build with stop points); key split and touch-ahead by batch size
  for stop in 2 1; do
    echo "== stop point $stop (2: after layer 0's self-attention, 1: after its attention-out)"
    MSD_AMD_LIB=$ROOT/tools/ab/libs/libmsd_amd_dbg.so MSD_DBG_STOP=$stop timeout 120 python tools/diag/dedup_diff.py tiny_context 1 2>&1 | grep -v Warning
  done | tee $OUT/${TAG}_dedup_diff.log
  timeout 400 python tools/ab/knob_ab.py --batch 8 --steps 200 --rounds 3 --json $OUT/${TAG}_split_b8.json 'kv_touch_ahead=0,cross_key_split=1' 'kv_touch_ahead=0,cross_key_split=2' 'kv_touch_ahead=0,cross_key_split=4' 2>&1 | grep -v Warning | tee $OUT/${TAG}_split_b8.log
  timeout 400 python tools/ab/knob_ab.py --batch 2 --steps 300 --rounds 3 --json $OUT/${TAG}_touch_b2.json 'kv_touch_ahead=0' 'kv_touch_ahead=2' 'kv_touch_ahead=0,cross_key_split=4' 'kv_touch_ahead=2,cross_key_split=4' 2>&1 | grep -v Warning | tee $OUT/${TAG}_touch_b2.log
  timeout 400 python tools/ab/knob_ab.py --batch 4 --steps 300 --rounds 3 --json $OUT/${TAG}_touch_b4.json 'kv_touch_ahead=0' 'kv_touch_ahead=2' 'kv_touch_ahead=0,cross_key_split=2' 'kv_touch_ahead=2,cross_key_split=2' 2>&1 | grep -v Warning | tee $OUT/${TAG}_touch_b4.log
  ;;
fourth)  # bitwise tests of the de-duplication after the contraction fix; batched path with the new tile rule / key split;
         # then the whole -m gpu suite on this binary
  timeout 600 python -m pytest tests/test_gpu_model.py -m gpu -q -x -k "dedup or staging or batched_songs or base_size" > $OUT/${TAG}_dedup_tests.log 2>&1; tail -4 $OUT/${TAG}_dedup_tests.log
  for r in 1 2; do
    for L in "MSD_AMD_LIB=$ROOT/tools/ab/libs/libmsd_amd_r04z.so" "X=0"; do
      env $L timeout 200 python bench.py --batch 8 --steps 2 --warmup 1 --no-cpu-baseline --batched-songs 0 --small-segments 0 --profile-steps 1 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('[batch 8 $L]', d['value'], d['ms_per_step'])"
    done
  done 2>&1 | tee $OUT/${TAG}_batch_ab.log
  for r in 1 2; do
    one "r04z" MSD_AMD_LIB=$ROOT/tools/ab/libs/libmsd_amd_r04z.so
    one "new" X=0
  done 2>&1 | tee $OUT/${TAG}_lib_ab.log
  timeout 1500 python -m pytest tests -m gpu -q > $OUT/${TAG}_gpu_tests.log 2>&1; tail -6 $OUT/${TAG}_gpu_tests.log
  grep -E "^FAILED|^ERROR" $OUT/${TAG}_gpu_tests.log | head -20
  ;;
fifth)   # fused projection + sampler: bitwise tests in every sampler mode; the new library with every round-5 shortcut OFF
         # against round 4's shipped binary (what did the source clean-up cost?), then with the defaults
  timeout 900 python -m pytest tests/test_gpu_model.py tests/test_gpu_fused_ops.py -m gpu -q -x -k "exact_launch or fused_projection or staging or sampler" > $OUT/${TAG}_fuse_tests.log 2>&1; tail -4 $OUT/${TAG}_fuse_tests.log
  OFF='dedup_layer0=False,fuse_final_sampler=False,kv_touch_ahead=0,cross_key_split=4'
  for r in 1 2; do
    MSD_AMD_LIB=$ROOT/tools/ab/libs/libmsd_amd_r04z.so timeout 200 python tools/ab/knob_ab.py --rounds 4 'graph_steps=8' 2>&1 | grep -E "median" | sed 's/^/[r04z] /'
    timeout 300 python tools/ab/knob_ab.py --rounds 4 "$OFF" 'fuse_final_sampler=False' 'graph_steps=8' 2>&1 | grep -E "median" | sed 's/^/[new ] /'
  done | tee $OUT/${TAG}_lib_knob_ab.log
  timeout 400 python tools/ab/knob_ab.py --batch 8 --steps 200 --rounds 3 'fuse_final_sampler=False' 'graph_steps=8' 2>&1 | grep -v Warning | tee $OUT/${TAG}_fuse_b8.log
  ;;
sixth)   # touch-ahead of the NEXT cross-attention launch's entry stages: bitwise tests, then in-process A/B
  timeout 900 python -m pytest tests/test_gpu_model.py -m gpu -q -x -k "exact_launch or staging" > $OUT/${TAG}_exact_tests.log 2>&1; tail -4 $OUT/${TAG}_exact_tests.log
  timeout 400 python tools/ab/knob_ab.py --rounds 5 --json $OUT/${TAG}_touch_next_ab.json 'kv_touch_ahead=0' 'kv_touch_ahead=102' 'kv_touch_ahead=2' 'kv_touch_ahead=4' 2>&1 | grep -v Warning | tee $OUT/${TAG}_touch_next_ab.log
  timeout 400 python tools/ab/knob_ab.py --rounds 3 --tokens 300 --tokens 900 --json $OUT/${TAG}_touch_next_ab2.json 'kv_touch_ahead=0' 'kv_touch_ahead=102' 'kv_touch_ahead=2' 2>&1 | grep -v Warning | tee $OUT/${TAG}_touch_next_ab2.log
  ;;
seventh) # 128-row cross-attention blocks at batch: parity, then this round's record binary (r05y) against the new one
  timeout 900 python -m pytest tests/test_gpu_model.py -m gpu -q -x -k "batched_songs or base_size or exact_launch" > $OUT/${TAG}_qb4_tests.log 2>&1; tail -4 $OUT/${TAG}_qb4_tests.log
  for nb in 8 4 16; do
    for r in 1 2; do
      for L in "MSD_AMD_LIB=$ROOT/tools/ab/libs/libmsd_amd_r05y.so" "X=0"; do
        env $L timeout 300 python bench.py --batch $nb --steps 2 --warmup 1 --no-cpu-baseline --batched-songs 0 --small-segments 0 --profile-steps 1 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('[batch $nb $L]', d['value'], d['ms_per_step'])"
      done
    done
  done 2>&1 | tee $OUT/${TAG}_qb4_ab.log
  ;;
eighth)  # one song: split 8 now runs on 128-row blocks (192 blocks whose whole key range fits the ring) -- against the default
  timeout 600 python tools/ab/knob_ab.py --rounds 4 --tokens 300 --tokens 900 --tokens 1300 --tokens 1536 --json $OUT/${TAG}_split8_qb4.json '' 'cross_key_split=8' 'cross_key_split=4' 2>&1 | grep -v Warning | tee $OUT/${TAG}_split8_qb4.log
  ;;
ninth)   # 128-row blocks for the batched decoder SELF-attention too: parity, then the r05x binary (one commit earlier) against the new one
  timeout 900 python -m pytest tests/test_gpu_model.py -m gpu -q -x -k "batched_songs or base_size or exact_launch" > $OUT/${TAG}_qb4s_tests.log 2>&1; tail -4 $OUT/${TAG}_qb4s_tests.log
  for nb in 8 16; do
    for r in 1 2; do
      for L in "MSD_AMD_LIB=$ROOT/tools/ab/libs/libmsd_amd_r05x.so" "X=0"; do
        env $L timeout 300 python bench.py --batch $nb --steps 2 --warmup 1 --no-cpu-baseline --batched-songs 0 --small-segments 0 --profile-steps 1 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('[batch $nb $L]', d['value'], d['ms_per_step'])"
      done
    done
  done 2>&1 | tee $OUT/${TAG}_qb4s_ab.log
  ;;
tests)
  timeout 1500 python -m pytest tests -m gpu -q -s > $OUT/${TAG}_gpu_tests.log 2>&1; tail -6 $OUT/${TAG}_gpu_tests.log
  grep -E "FAILED|ERROR" $OUT/${TAG}_gpu_tests.log | head -20
  ;;
final)
  timeout 1500 python -m pytest tests -m gpu -q -s > $OUT/${TAG}_gpu_tests.log 2>&1; tail -4 $OUT/${TAG}_gpu_tests.log
  timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
  bash tools/profile_round.sh $TAG > $OUT/${TAG}_profile.log 2>&1; tail -4 $OUT/${TAG}_profile.log
  TS=$ROOT/tools/ab/libs/libmsd_amd_ts.so   # the same sources with -DMSD_TIMESTAMPS=1 (tools/README.md)
  if [ -f $TS ]; then
    MSD_AMD_LIB=$TS timeout 200 python tools/diag/phase_times.py > $OUT/${TAG}_phase_times.txt 2>&1; tail -3 $OUT/${TAG}_phase_times.txt
    BATCH=8 MSD_AMD_LIB=$TS timeout 300 python tools/diag/phase_times.py > $OUT/${TAG}_phase_times_b8.txt 2>&1; tail -3 $OUT/${TAG}_phase_times_b8.txt
  fi
  cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/prof_b8
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_b8 -- \
      python $ROOT/bench.py --batch 8 --steps 1 --warmup 1 --no-cpu-baseline --batched-songs 0 --small-segments 0 --profile-steps 1 > $OUT/${TAG}_bench_b8_under_rocprof.json 2>/dev/null
  find /tmp/prof_b8 -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $OUT/${TAG}_b8_kernel_stats.csv
  cd $ROOT
  timeout 900 python bench.py > $OUT/${TAG}_bench_default.json 2> $OUT/${TAG}_bench_default.err; tail -c 1500 $OUT/${TAG}_bench_default.json
  ;;
esac
