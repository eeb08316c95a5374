#!/bin/bash
# round 5, call 14: (a) a CU's two workgroups on different slices (key 36 = rotation of the head pair for the second half of the
# grid); (b) every workgroup on the SAME 256-byte slice of the token rows (key 37 = slice + 1, timing only): is a slice slow on its own?
cd "${GRAFT_REPO_ROOT:-.}"
mkdir -p gpurun_out
timeout 500 python tools/tune_fp8.py cases=mixed,uniform8k "" "36=1" "36=2" "36=3" "" "37=1" "37=2" "37=3" "37=4" "36=1,30=112" "36=2,30=112" "" 2>&1 | tee gpurun_out/r5c14_rot.log
