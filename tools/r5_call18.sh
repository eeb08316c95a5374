#!/bin/bash
# round 5, call 18: second workgroup of every CU on head pair p ^ mask (key 36 = mask + 1; default now mask 2 for fp8 pairs):
# parity of the decode suites with the new default, then 8/64, 4/32, 16/128 heads, then bf16 masks 1..3
cd "${GRAFT_REPO_ROOT:-.}"
mkdir -p gpurun_out
{
timeout 600 python -m pytest tests/test_attention_decode_fp8.py tests/test_attention_decode_bf16.py tests/test_graded_shapes.py -m gpu -x -q -n 4 2>&1 | tail -5
timeout 300 python tools/tune_fp8.py heads=8/64 cases=mixed,uniform8k "36=1" "" "36=1" ""
timeout 300 python tools/tune_fp8.py heads=4/32 cases=mixed,uniform8k "36=1" "36=2" "36=1" "36=2"
timeout 300 python tools/tune_fp8.py heads=16/128 cases=mixed,uniform8k "36=1" "36=2" "36=3" "36=5" "36=1" "36=3"
echo "# bf16"
timeout 300 python tools/tune_bf16.py "36=2|3|4"
} 2>&1 | tee gpurun_out/r5c18_xor.log
