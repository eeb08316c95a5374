#!/bin/bash
# round 5, call 17: more workgroup -> head pair mappings (key 36), the four-head form (key 29 = 2) with / without a rotated second half
cd "${GRAFT_REPO_ROOT:-.}"
mkdir -p gpurun_out
timeout 500 python tools/tune_fp8.py cases=mixed,uniform8k "" "36=34" "36=1954" "36=930" "29=2" "29=2,36=33" "36=34" "" 2>&1 | tee gpurun_out/r5c17_map2.log
