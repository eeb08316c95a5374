#!/bin/bash
# round 5, call 19: which XCD (= workgroup index % 8) streams which slice: the six assignments of (slice bit 8, slice bit 9, range
# bit) to the three bits of the XCD index (key 38: eight 3-bit entries); the CU mates' bit-9 rule stays on
cd "${GRAFT_REPO_ROOT:-.}"
mkdir -p gpurun_out
timeout 500 python tools/tune_fp8.py cases=mixed,uniform8k "" "38=16434824" "38=16362248" "38=16205392" "38=15673952" "38=16096528" "38=15637664" "" 2>&1 | tee gpurun_out/r5c19_xcd.log
