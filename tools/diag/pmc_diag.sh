#!/bin/bash
# Diagnostic counter passes (GPU box): where do the waves of the step's kernels wait?  One rocprofv3 --pmc pass per
# counter group (kernel trace only), 12 DDPM steps of base_with_context, at BATCH songs per handle.
#   usage: BATCH=1 bash tools/diag/pmc_diag.sh <tag>      -> gpurun_out/<tag>_diag_b<BATCH>_<group>.csv
TAG=${1:-diag}
NB=${BATCH:-1}
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
run_pass() {  # name, counters
  rm -rf /tmp/pmcd_$1
  timeout 300 rocprofv3 --pmc $2 --kernel-trace --output-format csv -d /tmp/pmcd_$1 -- \
      python $ROOT/bench.py --batch $NB --steps 1 --warmup 0 --num-steps 12 --no-cpu-baseline --batched-songs 0 --small-segments 0 --profile-steps 1 > /tmp/pmcd_$1.log 2>&1
  f=$(find /tmp/pmcd_$1 -name "*counter_collection.csv" | head -1)
  if [ -n "$f" ]; then python $ROOT/tools/pmc_summary.py $f > $OUT/${TAG}_diag_b${NB}_$1.csv; echo "$1 ok"; else echo "no counters for $1"; grep -i "error\|fail\|invalid" /tmp/pmcd_$1.log | head -3; fi
}
run_pass sq   "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_VALU_MFMA_BUSY_CYCLES"
run_pass lds  "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_LDS_DATA_FIFO_FULL SQ_LDS_CMD_FIFO_FULL SQ_INST_CYCLES_VMEM"
run_pass tcp  "TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_GATE_EN1_sum"
run_pass ta   "TA_TA_BUSY_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_ADDR_STALLED_BY_TD_CYCLES_sum TD_TD_BUSY_sum TD_TC_STALL_sum"
run_pass tcc  "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_READ_sum"
run_pass ifetch "SQC_ICACHE_MISSES SQC_ICACHE_HITS SQC_ICACHE_REQ SQ_IFETCH SQ_INSTS_SMEM SQ_INST_CYCLES_SMEM"
run_pass tlb  "TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_TRANSLATION_HIT_sum TCP_UTCL1_REQUEST_sum TCP_UTCL1_STALL_INFLIGHT_MAX_sum"
ls -la $OUT/${TAG}_diag_b${NB}_*
