#!/bin/bash
# round 5, GPU call 1: phase profile of the 256 x 256 grouped GEMM, A/B of its k-loop variants, tail-tile forms, then the suite
cd "${GRAFT_REPO_ROOT:-.}"
mkdir -p gpurun_out
export HSA_ENABLE_IPC_MODE_LEGACY=0
timeout 300 python tools/prof_p8.py 1 11 > gpurun_out/r5c1_prof.log 2>&1
timeout 300 python tools/prof_p8.py --half 1 > gpurun_out/r5c1_prof_half.log 2>&1
timeout 600 python tools/tune_ggemm.py "0=0" "22=2" "22=3" "22=4" "22=5" "22=6" "22=7" "22=8" "22=9" "22=10" "0=0" > gpurun_out/r5c1_ab.log 2>&1
timeout 300 python tools/tune_ggemm.py --rows=18 --rows=40 --only=even18 --only=even40 "3=4" "3=2" "3=1" "3=4,22=8" "3=4,22=10" > gpurun_out/r5c1_tails.log 2>&1
( time timeout 1100 python -m pytest tests -x -q -m gpu --durations=40 ) > gpurun_out/r5c1_tests.log 2>&1
tail -3 gpurun_out/r5c1_tests.log
cat gpurun_out/r5c1_prof.log gpurun_out/r5c1_prof_half.log
tail -50 gpurun_out/r5c1_ab.log
cat gpurun_out/r5c1_tails.log
