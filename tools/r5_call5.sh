#!/bin/bash
# round 5, GPU call 5: where does the 256 x 256 kernel (tail / half-tile / full bodies) overtake the ring and streaming kernels?
cd "${GRAFT_REPO_ROOT:-.}"
mkdir -p gpurun_out
export HSA_ENABLE_IPC_MODE_LEGACY=0
: > gpurun_out/r5c5_sweep.log
for T in 32 64 96 128 160 192 256 384 512 768 1024 1536 2048; do
  timeout 200 python tools/tune_moe.py --tokens $T "0=0" "3=4" "0=0" "3=4" 2>/dev/null >> gpurun_out/r5c5_sweep.log
done
cat gpurun_out/r5c5_sweep.log
