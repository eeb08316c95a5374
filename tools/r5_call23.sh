#!/bin/bash
# round 5, call 23: router GEMM at large m - the LDS-staged tile kernel: parity suite, then A/B against the old kernel
cd "${GRAFT_REPO_ROOT:-.}"
mkdir -p gpurun_out
{
timeout 600 python -m pytest tests/test_gemm_bf16xfp32.py tests/test_router.py -m gpu -x -q -n 4 2>&1 | tail -5
timeout 300 python tools/tune_router.py
} 2>&1 | tee gpurun_out/r5c23_router.log
