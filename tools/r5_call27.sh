#!/bin/bash
# round 5, call 27: more workgroups for the slow (odd) slices on top of the CU-mate rule (keys 30 / 31 = 100 + extra per 128, key 42 = 1: keep both)
cd "${GRAFT_REPO_ROOT:-.}"
mkdir -p gpurun_out
timeout 500 python tools/tune_fp8.py cases=mixed,uniform8k "" "42=1,30=106,31=106" "42=1,30=112,31=112" "42=1,30=118,31=118" "42=1,30=112,31=106" "" 2>&1 | tee gpurun_out/r5c27_shares.log
