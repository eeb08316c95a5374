#!/bin/bash
# round 5, GPU call 3: the tail body (bit-identity tests first), its A/B on the routed sizes, the per-tensor form after the
# VALU-free load sections, the fused op; first-generation decode after the 16-chunk merge and the adaptive bin count
cd "${GRAFT_REPO_ROOT:-.}"
mkdir -p gpurun_out
export HSA_ENABLE_IPC_MODE_LEGACY=0
( HPC_AMD_DEV=1 timeout 600 python -m pytest tests/test_fuse_moe_blockwise.py tests/test_fuse_moe_pertensor.py -x -q -m "gpu and dev" -n 4 -k "tail_body or many_groups or activation_epilogue or kernel_arithmetic" ) > gpurun_out/r5c3_tests_tail.log 2>&1
tail -4 gpurun_out/r5c3_tests_tail.log
timeout 600 python tools/tune_ggemm.py --only=routed "0=0" "21=2" "21=1" "0=0" "21=2" > gpurun_out/r5c3_ab.log 2>&1
timeout 600 python tools/tune_ggemm.py --pertensor "0=0" "21=2" "0=0" > gpurun_out/r5c3_ab_pt.log 2>&1
timeout 300 python tools/tune_moe.py "0=0" "21=2" "22=2,21=2" "0=0" > gpurun_out/r5c3_moe.log 2>&1
timeout 600 python tools/tune_fp8.py heads=1/8 cases=uniform8k,mixed,extreme,one64k,skewed_mix,uniform512 "0=0" "33=1" "34=256" "0=0" > gpurun_out/r5c3_dec1.log 2>&1
( time timeout 900 python -m pytest tests/test_fuse_moe_blockwise.py tests/test_fuse_moe_pertensor.py tests/test_sched.py tests/test_attention_decode_fp8.py tests/test_attention_decode_bf16.py tests/test_graded_shapes.py -x -q -m gpu ) > gpurun_out/r5c3_tests.log 2>&1
tail -5 gpurun_out/r5c3_tests.log
cat gpurun_out/r5c3_ab.log gpurun_out/r5c3_ab_pt.log gpurun_out/r5c3_moe.log gpurun_out/r5c3_dec1.log
