#!/bin/bash
# round 5, call 28: per-XCD finish times of the head-pair decode kernel: product mapping, all workgroups on slice 0 / on slice 1
# (key 37), slice bit 8 from XCD index bit 1 instead of bit 0 (key 38)
cd "${GRAFT_REPO_ROOT:-.}"
mkdir -p gpurun_out
{
for t in "" "37=1" "37=2" "38=16205392" "36=1"; do
  echo "# tuning: ${t:-product}"
  HPC_AMD_TUNING="$t" timeout 200 python tools/prof_decode.py dump only=uniform8k 2>&1 | grep -v "WARNING\|amdgpu.ids"
done
} 2>&1 | tee gpurun_out/r5c28_xcd_prof.log
