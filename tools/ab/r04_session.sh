#!/bin/bash
# Round-4 GPU sessions (gpurun -- 'bash tools/ab/r04_session.sh <what> <tag>').  One script, several sessions:
#   tests   full -m gpu suite (product libraries + the experiments library's parity tests)
#   chain   same-box A/B of the persistent XCD-resident chain on the experiments library (default launches /
#           MSD_CHAIN=1 round 2's chain / MSD_CHAIN=2 with pre-staged weights) + the product library, then the phase
#           stamps of the separate launches and of the fused launch (timestamps build)
#   final   tests + smoke + tools/profile_round.sh + default bench on the shipped binary
WHAT=${1:-tests}; TAG=${2:-r04a}
ROOT=${GRAFT_REPO_ROOT:-/root/repo}; OUT=$ROOT/gpurun_out; mkdir -p $OUT; cd $ROOT
EXP=$ROOT/tools/ubench/exp/libmsd_amd_exp.so; EXPTS=$ROOT/tools/ubench/exp/libmsd_amd_exp_ts.so
B="python bench.py --steps 4 --warmup 1 --no-cpu-baseline --batched-songs 0 --small-segments 0 --profile-steps 1"
one() {  # label, env assignments...
  local label=$1; shift
  local extra=""
  for kv in "$@"; do case $kv in B_EXTRA=1) extra="--attn-planes 1,1";; B_EXTRA=2) extra="--attn-planes 2,1";; esac; done
  [[ "$label" == r04a* ]] && extra=""   # (the r04a build's default IS one plane for Q and P)
  env "$@" timeout 150 $B $extra 2>$OUT/${TAG}_err.tmp | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('[$label]', d['value'], round(d['sample_ms_per_segment'],1))" || { echo "[$label] FAILED"; tail -5 $OUT/${TAG}_err.tmp; }
}
case $WHAT in
tests)
  timeout 1200 python -m pytest tests -m gpu -q -s > $OUT/${TAG}_gpu_tests.log 2>&1; tail -6 $OUT/${TAG}_gpu_tests.log
  grep -E "FAILED|ERROR|Error|error" $OUT/${TAG}_gpu_tests.log | head -20
  ;;
chain)
  timeout 600 python -m pytest tests/test_gpu_experiments.py -m gpu -q -s -k "chain" > $OUT/${TAG}_chain_tests.log 2>&1; tail -4 $OUT/${TAG}_chain_tests.log
  grep -E "chain vs separate" $OUT/${TAG}_chain_tests.log
  for r in 1 2 3; do
    one "exp default" MSD_AMD_LIB=$EXP
    one "exp MSD_CHAIN=1" MSD_AMD_LIB=$EXP MSD_CHAIN=1
    one "exp MSD_CHAIN=2" MSD_AMD_LIB=$EXP MSD_CHAIN=2
    one "product" X=0
  done 2>&1 | tee $OUT/${TAG}_chain_ab.log
  MSD_AMD_LIB=$EXPTS timeout 200 python tools/diag/phase_times.py > $OUT/${TAG}_phase_times_launches.txt 2>&1; tail -3 $OUT/${TAG}_phase_times_launches.txt
  MSD_AMD_LIB=$EXPTS MSD_CHAIN=1 timeout 200 python tools/diag/phase_times.py > $OUT/${TAG}_phase_times_chain1.txt 2>&1; tail -5 $OUT/${TAG}_phase_times_chain1.txt
  MSD_AMD_LIB=$EXPTS MSD_CHAIN=2 timeout 200 python tools/diag/phase_times.py > $OUT/${TAG}_phase_times_chain2.txt 2>&1; tail -5 $OUT/${TAG}_phase_times_chain2.txt
  ;;
micro)   # round-4 micro-fixes (merge / final-proj / sampler / rstd / batched residual epilogue) against the r04a build,
         # both with the SAME attention planes; then the batched leg old vs new; then the default (all planes) line
  OLD=$ROOT/tools/ab/libs/libmsd_amd_r04a.so
  for r in 1 2 3; do
    one "r04a 1,1" MSD_AMD_LIB=$OLD X=0
    one "new 1,1" X=0 B_EXTRA=1
    one "new default (2,2)" X=0
    one "new 2,1" X=0 B_EXTRA=2
  done 2>&1 | tee $OUT/${TAG}_micro_ab.log
  for r in 1 2; do
    for L in "MSD_AMD_LIB=$OLD" "X=0"; do
      env $L timeout 200 python bench.py --batch 8 --steps 2 --warmup 1 --no-cpu-baseline --batched-songs 0 --small-segments 0 --profile-steps 1 --attn-planes 1,1 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('[batch 8 $L]', d['value'], d['ms_per_step'])"
    done
  done 2>&1 | tee $OUT/${TAG}_micro_b8_ab.log
  ;;
attn)    # attention prologue (unconditional ring issue, n_keys through the vector path) against the r04b build, default mode
  OLD=$ROOT/tools/ab/libs/libmsd_amd_${BASE:-r04b}.so
  for r in 1 2 3; do
    one "${BASE:-r04b}" MSD_AMD_LIB=$OLD X=0
    one "new" X=0
  done 2>&1 | tee $OUT/${TAG}_attn_ab.log
  ;;
multi)   # LIBS="tagA tagB ..." : the in-tree library against several tools/ab/libs/libmsd_amd_<tag>.so, alternating
  for r in 1 2 3; do
    for L in $LIBS; do one "$L" MSD_AMD_LIB=$ROOT/tools/ab/libs/libmsd_amd_$L.so X=0; done
    one "new" X=0
  done 2>&1 | tee $OUT/${TAG}_multi_ab.log
  ;;
batch)   # BASE=<tag>: 8 songs per handle, tools/ab/libs/libmsd_amd_<BASE>.so against the in-tree library, alternating
  OLD=$ROOT/tools/ab/libs/libmsd_amd_${BASE}.so
  for r in 1 2 3; do
    for L in "MSD_AMD_LIB=$OLD" "X=0"; do
      env $L timeout 200 python bench.py --batch 8 --steps 2 --warmup 1 --no-cpu-baseline --batched-songs 0 --small-segments 0 --profile-steps 1 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('[batch 8 $L]', d['value'], d['ms_per_step'])"
    done
  done 2>&1 | tee $OUT/${TAG}_batch_ab.log
  ;;
final)
  timeout 1200 python -m pytest tests -m gpu -q -s > $OUT/${TAG}_gpu_tests.log 2>&1; tail -4 $OUT/${TAG}_gpu_tests.log
  timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
  bash tools/profile_round.sh $TAG > $OUT/${TAG}_profile.log 2>&1; tail -4 $OUT/${TAG}_profile.log
  timeout 900 python bench.py > $OUT/${TAG}_bench_default.json 2> $OUT/${TAG}_bench_default.err; tail -c 1500 $OUT/${TAG}_bench_default.json
  ;;
esac
