#!/bin/bash
# round 5, call 22: is the unfairness between the slices zero-sum?  The workgroups of the even head pairs (the fast slices) sleep
# n x 64 clocks per wave-iteration (key 39)
cd "${GRAFT_REPO_ROOT:-.}"
mkdir -p gpurun_out
timeout 500 python tools/tune_fp8.py cases=mixed,uniform8k "" "39=72" "39=80" "39=88" "" "39=76" "39=84" 2>&1 | tee gpurun_out/r5c22_sleep.log
