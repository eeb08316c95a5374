#!/bin/bash
# round 5, call 15: which workgroup -> head pair mappings help (key 36 = m1 + 4 s1 + 64 (m2 + 4 s2): pair += m * ((wg >> s) & 3))
cd "${GRAFT_REPO_ROOT:-.}"
mkdir -p gpurun_out
timeout 500 python tools/tune_fp8.py cases=mixed,uniform8k "" "36=34" "36=29" "36=30" "36=13" "36=14" "36=9" "36=866" "36=34" "" 2>&1 | tee gpurun_out/r5c15_map.log
