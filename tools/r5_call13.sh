#!/bin/bash
# round 5, call 13: head-pair decode kernel on MORE workgroups than are resident (the hardware dispatcher hands the late
# ones to whichever CU frees a slot first), alone and with the per-slice shares of key 30
cd "${GRAFT_REPO_ROOT:-.}"
mkdir -p gpurun_out
timeout 500 python tools/tune_fp8.py cases=mixed,uniform8k "" "34=640,14=640" "34=768,14=768" "34=1024,14=1024" "30=112" \
  "34=768,14=768,30=112" "34=576,14=576" "" 2>&1 | tee gpurun_out/r5c13_grid.log
