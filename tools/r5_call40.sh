#!/bin/bash
# round 5, call 40: bf16 head-pair decode with the XCD table as the product's default (key 38 = 1: none): parity, then A B A
cd "${GRAFT_REPO_ROOT:-.}"
mkdir -p gpurun_out
{
timeout 400 python -m pytest tests/test_attention_decode_bf16.py -m gpu -x -q -n 4 2>&1 | tail -3
timeout 200 python -m pytest tests/test_graded_shapes.py -m gpu -x -q -n 4 -k "c2_bf16" 2>&1 | tail -2
timeout 200 python tools/tune_bf16.py "38=1" 2>&1 | grep "bf16\|max"
} 2>&1 | tee gpurun_out/r5c40_bf16_default.log
