#!/bin/bash
# GPU call 7 of round 3: prefetch-wave default build (+ K/V cache prefetch from the QKV launch): full -m gpu suite with
# all attention products (QP = 0); the model / reference-fixture tests under QP = 1, 2, 3 (which single-plane choice
# passes the UNCHANGED tests?); same-box A/B of QP and of the K/V prefetch.
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out
mkdir -p $OUT
cd $ROOT
MSD_ATT_QP_SELF=0 MSD_ATT_QP_CROSS=0 timeout 700 python -m pytest tests -m gpu -q > $OUT/r03g_gpu_tests_qp0.log 2>&1; tail -3 $OUT/r03g_gpu_tests_qp0.log
for QP in 1 2 3; do
  MSD_ATT_QP_SELF=$QP MSD_ATT_QP_CROSS=$QP timeout 300 python -m pytest tests/test_gpu_model.py tests/test_ref_golden.py tests/test_gpu_chained.py -m gpu -q > $OUT/r03g_tests_qp$QP.log 2>&1
  echo "== QP=$QP: $(tail -1 $OUT/r03g_tests_qp$QP.log)"; grep -E "^FAILED" $OUT/r03g_tests_qp$QP.log | cut -c1-150
done
B="python bench.py --steps 4 --warmup 1 --no-cpu-baseline --batched-songs 0 --small-segments 0 --profile-steps 1"
for r in 1 2; do
  for E in "MSD_PF_KV=1" "MSD_PF_KV=0" "MSD_ATT_QP_SELF=0 MSD_ATT_QP_CROSS=0" "MSD_ATT_QP_SELF=2 MSD_ATT_QP_CROSS=2" "MSD_ATT_QP_SELF=1 MSD_ATT_QP_CROSS=1"; do
    env $E timeout 120 $B 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('[$E]', d['value'], round(d['sample_ms_per_segment'],1), {k: round(v*1000,1) for k,v in d['roofline']['per_class_ms_per_step'].items()})"
  done
done 2>&1 | tee $OUT/r03g_env_ab.log
