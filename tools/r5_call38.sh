#!/bin/bash
# round 5, call 38: the new development tests (bit-identity of the mapping / tile kernel / transposing reads) on the development build
cd "${GRAFT_REPO_ROOT:-.}"
mkdir -p gpurun_out
HPC_AMD_DEV=1 timeout 500 python -m pytest tests/test_attention_decode_fp8.py tests/test_gemm_bf16xfp32.py tests/test_attention_prefill_bf16.py -m "gpu and dev" -q -n 4 -k "only_a_mapping or equals_the_direct or equal_the_perm" 2>&1 | tail -8 | tee gpurun_out/r5c38_newtests.log
