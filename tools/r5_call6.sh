#!/bin/bash
# round 5, GPU call 6: MoE tests after the new kernel choice (256 x 256 kernel from 16 rows per group), non-temporal weights for
# single-tile groups (key 24 = 1: off), old threshold (key 25 = 1)
cd "${GRAFT_REPO_ROOT:-.}"
mkdir -p gpurun_out
export HSA_ENABLE_IPC_MODE_LEGACY=0
( time timeout 900 python -m pytest tests/test_fuse_moe_blockwise.py tests/test_fuse_moe_pertensor.py tests/test_graded_shapes.py tests/test_replay_check.py tests/test_oracle_golden.py -x -q -m gpu ) > gpurun_out/r5c6_tests.log 2>&1
tail -5 gpurun_out/r5c6_tests.log
( HPC_AMD_DEV=1 timeout 600 python -m pytest tests/test_fuse_moe_blockwise.py tests/test_fuse_moe_pertensor.py tests/test_graded_shapes.py -x -q -m "gpu and dev" -n 4 ) > gpurun_out/r5c6_tests_dev.log 2>&1
tail -3 gpurun_out/r5c6_tests_dev.log
: > gpurun_out/r5c6_sweep.log
for T in 16 64 128 192 256 512 1024; do
  timeout 200 python tools/tune_moe.py --tokens $T "0=0" "24=1" "25=1" "0=0" "24=1" 2>/dev/null >> gpurun_out/r5c6_sweep.log
done
cat gpurun_out/r5c6_sweep.log
