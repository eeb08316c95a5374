#!/bin/bash
# Development tool (GPU box): one A/B question per gpurun call - a tune / profile tool of tools/ under a list of development-register
# settings (csrc/hpc_dev.h), A B A on one box, everything logged to gpurun_out/<tag>.log.  Replaces the per-call one-liners of
# round 5 (tools/r5_call*.sh, in the history up to 84e7f82; profiles/round5_*_ab.txt quote them by call number).
# usage: gpurun -- 'bash tools/ab.sh <tag> "<tool and its fixed arguments>" "<k=v,k=v>" ["<k=v>" ...]'
#   e.g.  bash tools/ab.sh r6_persist "tools/tune_ggemm.py --only=routed" "0=0" "47=1" "0=0"
#   tools that take the settings as arguments (tune_ggemm.py, tune_fp8.py ...) get them appended; with ENV=1 every setting is
#   instead passed as HPC_AMD_TUNING to a separate run of the tool (prof_decode.py, prof_p8.py).
cd "${GRAFT_REPO_ROOT:-.}"
mkdir -p gpurun_out
TAG=$1; TOOL=$2; shift 2
{
if [ -n "$ENV" ]; then
  for t in "$@"; do
    echo "# tuning: ${t:-product}"
    HPC_AMD_DEV=1 HPC_AMD_TUNING="$t" timeout ${LIMIT:-300} python $TOOL 2>&1 | grep -v "WARNING\|amdgpu.ids"
  done
else
  HPC_AMD_DEV=1 timeout ${LIMIT:-600} python $TOOL "$@" 2>&1 | grep -v "amdgpu.ids"
fi
} 2>&1 | tee gpurun_out/${TAG}.log | tail -${TAIL:-60}
