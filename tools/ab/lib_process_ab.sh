#!/bin/bash
# Alternating-process A/B of two builds of the library on one box: `bench.py` (3 timed segments, no side legs) with
# MSD_AMD_LIB=<A> / <B> in turn, ROUNDS times.   usage: bash tools/ab/lib_process_ab.sh <libA.so> <libB.so|-> [rounds] [tag]
# ("-" = the in-tree library)
A=$1; B=$2; ROUNDS=${3:-3}; TAG=${4:-lib_ab}
ROOT=${GRAFT_REPO_ROOT:-/root/repo}; OUT=$ROOT/gpurun_out; mkdir -p $OUT; cd $ROOT
run() {  # label, lib
  if [ "$2" = "-" ]; then unset MSD_AMD_LIB; else export MSD_AMD_LIB=$2; fi
  timeout 300 python bench.py --steps 3 --no-cpu-baseline --batched-songs 0 --small-segments 0 --no-self-profile 2>/dev/null \
    | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().split(chr(10))[-1]); print('$1', d['value'], d['ms_per_step'])" | tee -a $OUT/${TAG}.log
  unset MSD_AMD_LIB
}
for i in $(seq 1 $ROUNDS); do run A $A; run B $B; done
