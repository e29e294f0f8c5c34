#!/bin/bash
# diagnostics of the MLP chain: (1) plain loads instead of sc1 (parity + speed), (2) the unfused kernels with the
# chain's tile ownership (one row tile per XCD), (3) defaults
LIB=music-spectrogram-diffusion_amd/csrc/libmsd_amd.so
B="python bench.py --steps 3 --warmup 1 --no-cpu-baseline --batched-songs 0 --profile-steps 1"
show() { python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$1', d['value'], d['sample_ms_per_segment'], d['roofline']['per_class_ms_per_step'])"; }
cp tools/ab/lib_cp0.so $LIB
timeout 200 python -m pytest tests/test_golden.py tests/test_gpu_model.py -m gpu -q -x --deselect tests/test_golden.py::test_base_with_context_chain_stays_inside_the_bar 2>&1 | tail -3
timeout 100 $B 2>/dev/null | show "cp0 chain"
cp tools/ab/lib_cp16.so $LIB
MSD_CHAIN=0 MSD_XCD_GEMM_MLP_IN_GEGLU=8,1 MSD_XCD_GEMM_MLP_OUT=8,1 MSD_XCD_GEMM_QKV=8,1 timeout 100 $B 2>/dev/null | show "unfused rows8"
MSD_CHAIN=0 timeout 100 $B 2>/dev/null | show "unfused default"
timeout 100 $B 2>/dev/null | show "cp16 chain"
cp tools/ab/lib_cp0.so $LIB
timeout 100 $B 2>/dev/null | show "cp0 chain"
cp tools/ab/lib_cp16.so $LIB
