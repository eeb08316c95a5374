#!/bin/bash
# round 5, GPU call 7: the stream body (four-stage weight rings for single-tile groups): bit-identity tests, then the fused op at
# decode sizes against the tail body with (24=2) and without (24=1) non-temporal loads
cd "${GRAFT_REPO_ROOT:-.}"
mkdir -p gpurun_out
export HSA_ENABLE_IPC_MODE_LEGACY=0
( HPC_AMD_DEV=1 timeout 600 python -m pytest tests/test_fuse_moe_blockwise.py tests/test_fuse_moe_pertensor.py -x -q -m "gpu and dev" -n 4 -k "tail_body or many_groups or activation_epilogue" ) > gpurun_out/r5c7_tests_dev.log 2>&1
tail -3 gpurun_out/r5c7_tests_dev.log
( timeout 900 python -m pytest tests/test_fuse_moe_blockwise.py tests/test_fuse_moe_pertensor.py tests/test_graded_shapes.py tests/test_oracle_golden.py -x -q -m gpu ) > gpurun_out/r5c7_tests.log 2>&1
tail -3 gpurun_out/r5c7_tests.log
: > gpurun_out/r5c7_sweep.log
for T in 128 192 256 384 512; do
  timeout 200 python tools/tune_moe.py --tokens $T "0=0" "24=2" "24=1" "0=0" "24=2" 2>/dev/null >> gpurun_out/r5c7_sweep.log
done
cat gpurun_out/r5c7_sweep.log
