#!/bin/bash
# round 5, call 29: hand-over of the odd slots' ends to the even workgroups (key 43 = permille + 1): timing sweep, then the fp8 decode
# parity suites on the development build with the hand-over on
cd "${GRAFT_REPO_ROOT:-.}"
mkdir -p gpurun_out
{
timeout 300 python tools/tune_fp8.py cases=mixed,uniform8k,skewed_mix,one64k "" "43=31" "43=51" "43=71" "43=101" "43=141" ""
HPC_AMD_DEV=1 HPC_AMD_TUNING="43=61" timeout 600 python -m pytest tests/test_attention_decode_fp8.py tests/test_graded_shapes.py -m gpu -x -q -n 4 -k "fp8 or c3" 2>&1 | tail -5
} 2>&1 | tee gpurun_out/r5c29_handover.log
