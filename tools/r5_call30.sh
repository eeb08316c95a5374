#!/bin/bash
# round 5, call 30: split-request merge with three chunks in flight per wave (default) against two (key 44 = 1)
cd "${GRAFT_REPO_ROOT:-.}"
mkdir -p gpurun_out
timeout 300 python tools/tune_fp8.py cases=mixed,one64k,skewed_mix,extreme "44=1" "" "44=1" "" "44=1" "" 2>&1 | tee gpurun_out/r5c30_merge3.log
