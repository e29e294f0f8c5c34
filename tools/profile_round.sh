#!/bin/bash
# One profiling round on the GPU box: kernel-trace stats of a full bench step + separate PMC passes, all
# stamped with the hash of the library they ran on.  Outputs -> gpurun_out/<tag>_*; copy to profiles/ and run
# `python tools/make_roofline.py <tag>` afterwards (here, in the container).
# usage: [PRESET=small] [SKIP_TRACE=1] [SKIP_PMC=1] bash tools/profile_round.sh <tag>
TAG=${1:-r02}
PRESET=${PRESET:-base_with_context}   # BASELINE config 3 (default) or `small` = config 2 (round 6)
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
sha256sum $ROOT/music-spectrogram-diffusion_amd/csrc/libmsd_amd.so | cut -c1-16 > $OUT/${TAG}_library_sha.txt
# 1. kernel trace + stats over one full 1000-step segment (+1 warm-up): what the graph replays
if [ "${SKIP_TRACE:-0}" != "1" ]; then   # (bench.py's own rocprofv3 leg records the same table: --self-profile-keep)
rm -rf /tmp/prof_$TAG
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$TAG -- \
    python $ROOT/bench.py --preset $PRESET --steps 1 --warmup 1 --no-cpu-baseline --no-self-profile --batched-songs 0 --small-segments 0 --profile-steps 1 > $OUT/${TAG}_bench_under_rocprof.json 2> /tmp/prof_$TAG.err
find /tmp/prof_$TAG -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $OUT/${TAG}_bench_kernel_stats.csv
fi
# 2. PMC passes (one counter group per pass, kernel-trace only).  Counter collection serialises the dispatches at
#    ~50 ms each on this pool (100 DDPM steps did not finish in 600 s), so: 12 DDPM steps (1 332 step launches;
#    the encoder's ~400 launches of the same GEMM templates at other shapes are in the averages too -- the
#    attention / merge / tail kernels are not affected)
run_pass() {  # name, counters
  rm -rf /tmp/pmc_$1
  timeout 420 rocprofv3 --pmc $2 --kernel-trace --output-format csv -d /tmp/pmc_$1 -- \
      python $ROOT/bench.py --preset $PRESET --steps 1 --warmup 0 --num-steps 12 --no-cpu-baseline --no-self-profile --batched-songs 0 --small-segments 0 --profile-steps 1 > /tmp/pmc_$1.log 2>&1
  f=$(find /tmp/pmc_$1 -name "*counter_collection.csv" | head -1)
  if [ -n "$f" ]; then python $ROOT/tools/pmc_summary.py $f > $OUT/${TAG}_pmc_$1.csv; else echo "no counters for $1"; tail -5 /tmp/pmc_$1.log; fi
}
if [ "${SKIP_PMC:-0}" = "1" ]; then ls -la $OUT/${TAG}_*; exit 0; fi   # kernel trace only (short GPU budget)
run_pass fetch "FETCH_SIZE"
run_pass write "WRITE_SIZE"
run_pass sq "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAVES"
ls -la $OUT/${TAG}_*
