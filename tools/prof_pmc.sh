#!/bin/bash
# PMC passes over a short bench run (each counter group in its own pass, kernel-trace only:
# /opt/skills/guides/MI355X_MICROARCH.md "rocprofv3 PMC slots").  Summaries -> gpurun_out/pmc_<tag>_<group>.csv
# usage: bash tools/prof_pmc.sh <tag>
TAG=${1:-r01}
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
run_pass() {  # name, counters
  rm -rf /tmp/pmc_$1
  timeout 400 rocprofv3 --pmc $2 --kernel-trace --output-format csv -d /tmp/pmc_$1 -- \
      python $ROOT/bench.py --steps 1 --warmup 0 --num-steps 24 --no-cpu-baseline --batched-songs 0 --profile-steps 1 > /tmp/pmc_$1.log 2>&1
  f=$(find /tmp/pmc_$1 -name "*counter_collection.csv" | head -1)
  if [ -n "$f" ]; then python $ROOT/tools/pmc_summary.py $f > $ROOT/gpurun_out/pmc_${TAG}_$1.csv; else echo "no counters for $1"; tail -5 /tmp/pmc_$1.log; fi
}
run_pass fetch "FETCH_SIZE"
run_pass write "WRITE_SIZE"
run_pass sq "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAVES"
run_pass lds "SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VALU"
ls -la $ROOT/gpurun_out/pmc_${TAG}_*.csv
