#!/bin/bash
# round 5, call 11: register-streamed tail body of the 256 x 256 grouped GEMM: bit-identity tests (development build) + A/B
cd "${GRAFT_REPO_ROOT:-.}"
mkdir -p gpurun_out
( time HPC_AMD_DEV=1 timeout 900 python -m pytest tests/test_fuse_moe_blockwise.py tests/test_fuse_moe_pertensor.py -x -q -m "gpu and dev" -k "tail_body" -n 4 ) > gpurun_out/r5c11_tests_dev.log 2>&1
tail -n 5 gpurun_out/r5c11_tests_dev.log
timeout 600 python tools/tune_ggemm.py "0=0" "26=1" "0=0" "26=1" > gpurun_out/r5c11_ggemm.log 2>&1
grep -v "amdgpu.ids" gpurun_out/r5c11_ggemm.log | tail -n 20
timeout 300 python tools/tune_ggemm.py --pertensor "0=0" "26=1" "0=0" > gpurun_out/r5c11_ggemm_pt.log 2>&1
grep -v "amdgpu.ids" gpurun_out/r5c11_ggemm_pt.log | tail -n 12
