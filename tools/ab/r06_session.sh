#!/bin/bash
# Round-6 GPU sessions (gpurun -- 'bash tools/ab/r06_session.sh <what> <tag>').
#   first   new GPU tests (in-kernel step noise, refused late set_weight, touch variants with the prefetch wave forced on)
#           + the default bench line with its own rocprofv3 legs (base + small) + PMC passes of the `small` preset
#   tests   full -m gpu suite
#   final   tests + smoke + tools/profile_round.sh (base) + default bench on the shipped binary
WHAT=${1:-first}; TAG=${2:-r06a}
ROOT=${GRAFT_REPO_ROOT:-/root/repo}; OUT=$ROOT/gpurun_out; mkdir -p $OUT; cd $ROOT
sha256sum $ROOT/music-spectrogram-diffusion_amd/csrc/libmsd_amd.so | cut -c1-16 > $OUT/${TAG}_library_sha.txt
case $WHAT in
first)
  timeout 900 python -m pytest tests/test_gpu_model.py -m gpu -q -x -k "draws_the_step_noise or staging or exact_launch or philox or explicit_noise" > $OUT/${TAG}_new_tests.log 2>&1; tail -5 $OUT/${TAG}_new_tests.log
  mkdir -p $OUT/${TAG}_selfprof
  timeout 600 python bench.py --self-profile-keep $OUT/${TAG}_selfprof > $OUT/${TAG}_bench_default.json 2> $OUT/${TAG}_bench_default.err; tail -c 600 $OUT/${TAG}_bench_default.json; tail -3 $OUT/${TAG}_bench_default.err
  PRESET=small SKIP_TRACE=1 bash tools/profile_round.sh ${TAG}_small
  ;;
second)  # in-launch merge of the key-split cross-attention: op-level and model-level bitwise tests, then in-process A/B
  timeout 900 python -m pytest tests/test_gpu_ops.py -m gpu -q -x -k "merged_inside_the_launch or test_attention" > $OUT/${TAG}_op_tests.log 2>&1; tail -4 $OUT/${TAG}_op_tests.log
  timeout 1200 python -m pytest tests/test_gpu_model.py -m gpu -q -x -k "draws_the_step_noise or staging or exact_launch or key_split or batched_songs or sum_cross" > $OUT/${TAG}_model_tests.log 2>&1; tail -4 $OUT/${TAG}_model_tests.log
  timeout 400 python tools/ab/knob_ab.py --rounds 5 --json $OUT/${TAG}_merge_ab.json 'cross_merge_in_launch=False' 'cross_merge_in_launch=True' 2>&1 | grep -v Warning | tee $OUT/${TAG}_merge_ab.log
  timeout 400 python tools/ab/knob_ab.py --rounds 3 --tokens 300 --tokens 1300 --json $OUT/${TAG}_merge_ab2.json 'cross_merge_in_launch=False' 'cross_merge_in_launch=True' 2>&1 | grep -v Warning | tee $OUT/${TAG}_merge_ab2.log
  timeout 300 python tools/ab/knob_ab.py --preset small --rounds 4 --json $OUT/${TAG}_merge_ab_small.json 'cross_merge_in_launch=False' 'cross_merge_in_launch=True' 2>&1 | grep -v Warning | tee $OUT/${TAG}_merge_ab_small.log
  mkdir -p $OUT/${TAG}_selfprof
  timeout 600 python bench.py --no-cpu-baseline --self-profile-keep $OUT/${TAG}_selfprof > $OUT/${TAG}_bench_default.json 2> $OUT/${TAG}_bench_default.err; tail -c 300 $OUT/${TAG}_bench_default.json; tail -3 $OUT/${TAG}_bench_default.err
  ;;
third)  # folded cross-attention query projection (S6): parity tests, in-process A/B, bench with its own kernel trace
  timeout 1500 python -m pytest tests/test_gpu_model.py -m gpu -q -x -s -k "folded_cross_query or exact_launch or sum_cross or encode_and_single or does_not_depend_on_the_batch or batched_songs" > $OUT/${TAG}_fold_tests.log 2>&1; tail -4 $OUT/${TAG}_fold_tests.log; grep "folded vs unfolded" $OUT/${TAG}_fold_tests.log | head -40
  timeout 500 python tools/ab/knob_ab.py --rounds 5 --json $OUT/${TAG}_fold_ab.json 'cross_q_fold=False' 'cross_q_fold=True' 2>&1 | grep -v Warning | tee $OUT/${TAG}_fold_ab.log
  timeout 400 python tools/ab/knob_ab.py --rounds 3 --tokens 300 --tokens 1300 --json $OUT/${TAG}_fold_ab2.json 'cross_q_fold=False' 'cross_q_fold=True' 2>&1 | grep -v Warning | tee $OUT/${TAG}_fold_ab2.log
  timeout 300 python tools/ab/knob_ab.py --preset small --rounds 4 --json $OUT/${TAG}_fold_ab_small.json 'cross_q_fold=False' 'cross_q_fold=True' 2>&1 | grep -v Warning | tee $OUT/${TAG}_fold_ab_small.log
  timeout 400 python tools/ab/knob_ab.py --rounds 3 --batch 2 --steps 500 --json $OUT/${TAG}_fold_ab_b2.json 'cross_q_fold=False' 'cross_q_fold=True' 2>&1 | grep -v Warning | tee $OUT/${TAG}_fold_ab_b2.log
  mkdir -p $OUT/${TAG}_selfprof
  timeout 600 python bench.py --no-cpu-baseline --self-profile-keep $OUT/${TAG}_selfprof > $OUT/${TAG}_bench_default.json 2> $OUT/${TAG}_bench_default.err; tail -c 300 $OUT/${TAG}_bench_default.json; tail -3 $OUT/${TAG}_bench_default.err
  ;;
fourth)  # the fold after: stage 0 peeled in the QS attention kernel, fold matrices behind Wqkv / Wo (no extra prefetch waves)
  timeout 600 python tests/diag/fold_stats.py --seeds 4 2>&1 | grep -v Warning | tee $OUT/${TAG}_fold_stats.log | tail -8
  timeout 500 python tools/ab/knob_ab.py --rounds 5 --json $OUT/${TAG}_fold_ab.json 'cross_q_fold=False' 'cross_q_fold=True' 2>&1 | grep -v Warning | tee $OUT/${TAG}_fold_ab.log
  timeout 400 python tools/ab/knob_ab.py --rounds 3 --tokens 300 --tokens 1300 --json $OUT/${TAG}_fold_ab2.json 'cross_q_fold=False' 'cross_q_fold=True' 2>&1 | grep -v Warning | tee $OUT/${TAG}_fold_ab2.log
  timeout 300 python tools/ab/knob_ab.py --preset small --rounds 4 --json $OUT/${TAG}_fold_ab_small.json 'cross_q_fold=False' 'cross_q_fold=True' 2>&1 | grep -v Warning | tee $OUT/${TAG}_fold_ab_small.log
  mkdir -p $OUT/${TAG}_selfprof
  timeout 600 python bench.py --no-cpu-baseline --batched-songs 0 --small-segments 0 --self-profile-keep $OUT/${TAG}_selfprof > $OUT/${TAG}_bench_default.json 2> $OUT/${TAG}_bench_default.err; tail -c 300 $OUT/${TAG}_bench_default.json; tail -3 $OUT/${TAG}_bench_default.err
  ;;
fifth)  # same-box kernel traces of both orders (bench self-profile with --knob)
  mkdir -p $OUT/${TAG}_selfprof_fold $OUT/${TAG}_selfprof_plain
  for v in plain fold; do
    K="cross_q_fold=$([ $v = fold ] && echo True || echo False)"
    timeout 400 python bench.py --steps 3 --no-cpu-baseline --batched-songs 0 --small-segments 0 --knob $K --self-profile-keep $OUT/${TAG}_selfprof_$v > $OUT/${TAG}_bench_$v.json 2> $OUT/${TAG}_bench_$v.err; tail -c 200 $OUT/${TAG}_bench_$v.json; tail -2 $OUT/${TAG}_bench_$v.err
  done
  ;;
sixth)  # two tools on one box: alternating processes (bench) and one process with both models (knob_ab, both orders)
  for i in 1 2; do for v in plain fold; do
    K="cross_q_fold=$([ $v = fold ] && echo True || echo False)"
    timeout 300 python bench.py --steps 3 --no-cpu-baseline --batched-songs 0 --small-segments 0 --no-self-profile --knob $K 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().split(chr(10))[-1]); print('$v', d['value'], d['ms_per_step'])" | tee -a $OUT/${TAG}_process_ab.log
  done; done
  timeout 500 python tools/ab/knob_ab.py --rounds 4 --json $OUT/${TAG}_fold_ab.json 'cross_q_fold=False' 'cross_q_fold=True' 2>&1 | grep -v Warning | tee $OUT/${TAG}_fold_ab.log
  timeout 500 python tools/ab/knob_ab.py --rounds 4 --json $OUT/${TAG}_fold_ab_rev.json 'cross_q_fold=True' 'cross_q_fold=False' 2>&1 | grep -v Warning | tee $OUT/${TAG}_fold_ab_rev.log
  timeout 500 python tools/ab/knob_ab.py --rounds 4 --tokens 900 --json $OUT/${TAG}_fold_ab_900.json 'cross_q_fold=False' 'cross_q_fold=True' 2>&1 | grep -v Warning | tee $OUT/${TAG}_fold_ab_900.log
  ;;
seventh)  # fold: full GPU suite + small / 2 / 3 songs A/B
  timeout 2400 python -m pytest tests -m gpu -q -x > $OUT/${TAG}_gpu_tests.log 2>&1; tail -4 $OUT/${TAG}_gpu_tests.log
  timeout 300 python tools/ab/knob_ab.py --preset small --rounds 4 --json $OUT/${TAG}_fold_ab_small.json 'cross_q_fold=False' 'cross_q_fold=True' 2>&1 | grep -v Warning | tee $OUT/${TAG}_fold_ab_small.log
  timeout 400 python tools/ab/knob_ab.py --rounds 3 --batch 2 --steps 500 --tokens 900 --json $OUT/${TAG}_fold_ab_b2.json 'cross_q_fold=False' 'cross_q_fold=True' 2>&1 | grep -v Warning | tee $OUT/${TAG}_fold_ab_b2.log
  timeout 400 python tools/ab/knob_ab.py --rounds 3 --batch 3 --steps 300 --tokens 900 --json $OUT/${TAG}_fold_ab_b3.json 'cross_q_fold=False' 'cross_q_fold=True' 2>&1 | grep -v Warning | tee $OUT/${TAG}_fold_ab_b3.log
  ;;
eighth)  # persistent gated-MLP-in at batch: parity test + A/B at 4 / 8 / 16 songs
  timeout 900 python -m pytest tests/test_gpu_model.py -m gpu -q -x -k "batched_songs" > $OUT/${TAG}_batched_test.log 2>&1; tail -3 $OUT/${TAG}_batched_test.log
  for nb in 8 4 16; do
    timeout 500 python tools/ab/knob_ab.py --rounds 3 --batch $nb --steps 120 --tokens 900 --json $OUT/${TAG}_persist_ab_b$nb.json 'mlp_in_persistent=False' 'mlp_in_persistent=True' 2>&1 | grep -v Warning | tee $OUT/${TAG}_persist_ab_b$nb.log
  done
  ;;
final)   # the round's record: tests + smoke + kernel trace + counter passes + stamps + default bench, ONE binary
  timeout 2400 python -m pytest tests -m gpu -q > $OUT/${TAG}_gpu_tests.log 2>&1; tail -4 $OUT/${TAG}_gpu_tests.log
  timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
  bash tools/profile_round.sh $TAG > $OUT/${TAG}_profile.log 2>&1; tail -4 $OUT/${TAG}_profile.log
  PRESET=small SKIP_TRACE=1 bash tools/profile_round.sh ${TAG}_small > $OUT/${TAG}_small_profile.log 2>&1; tail -2 $OUT/${TAG}_small_profile.log
  TS=$ROOT/tools/ubench/exp/libmsd_amd_ts.so   # the same sources with -DMSD_TIMESTAMPS=1
  if [ -f $TS ]; then
    MSD_AMD_LIB=$TS timeout 200 python tools/diag/phase_times.py > $OUT/${TAG}_phase_times.txt 2>&1; tail -3 $OUT/${TAG}_phase_times.txt
    BATCH=8 MSD_AMD_LIB=$TS timeout 300 python tools/diag/phase_times.py > $OUT/${TAG}_phase_times_b8.txt 2>&1; tail -3 $OUT/${TAG}_phase_times_b8.txt
  fi
  cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/prof_b8
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_b8 -- \
      python $ROOT/bench.py --batch 8 --steps 1 --warmup 1 --no-cpu-baseline --no-self-profile --batched-songs 0 --small-segments 0 --profile-steps 1 > $OUT/${TAG}_bench_b8_under_rocprof.json 2>/dev/null
  find /tmp/prof_b8 -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $OUT/${TAG}_b8_kernel_stats.csv
  cd $ROOT
  mkdir -p $OUT/${TAG}_selfprof
  timeout 900 python bench.py --self-profile-keep $OUT/${TAG}_selfprof > $OUT/${TAG}_bench_default.json 2> $OUT/${TAG}_bench_default.err; tail -c 1200 $OUT/${TAG}_bench_default.json
  ;;
driver)   # what the driver runs at round end, on one fresh box: the -m gpu suite (-x), smoke, the default bench
  timeout 2400 python -m pytest tests -x -q -m gpu > $OUT/${TAG}_gpu_tests.log 2>&1; tail -3 $OUT/${TAG}_gpu_tests.log
  timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $OUT/${TAG}_smoke.log 2>&1; tail -2 $OUT/${TAG}_smoke.log
  T0=$(date +%s); timeout 900 python bench.py > $OUT/${TAG}_bench_default.json 2> $OUT/${TAG}_bench_default.err; echo "bench wall $(( $(date +%s) - T0 )) s" | tee -a $OUT/${TAG}_bench_default.err; tail -c 900 $OUT/${TAG}_bench_default.json; tail -2 $OUT/${TAG}_bench_default.err
  ;;
tests)
  timeout 2400 python -m pytest tests -m gpu -q > $OUT/${TAG}_gpu_tests.log 2>&1; tail -6 $OUT/${TAG}_gpu_tests.log
  grep -E "^FAILED|^ERROR" $OUT/${TAG}_gpu_tests.log | head -20
  ;;
esac
ls -la $OUT | tail -20
