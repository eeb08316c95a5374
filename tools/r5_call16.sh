#!/bin/bash
# round 5, call 16: second half of the grid on head pair p + 2 (key 36 = 34): all C3 cases, the per-wave profile with / without, bf16
cd "${GRAFT_REPO_ROOT:-.}"
mkdir -p gpurun_out
{
timeout 300 python tools/tune_fp8.py "" "36=34" "" "36=34" "36=34,32=110" "36=34,32=120"
echo "# profile, default mapping"
timeout 200 python tools/prof_decode.py dump
echo "# profile, 36=34"
HPC_AMD_TUNING="36=34" timeout 200 python tools/prof_decode.py dump
echo "# bf16"
timeout 300 python tools/tune_bf16.py "36=34"
} 2>&1 | tee gpurun_out/r5c16_rot2.log
