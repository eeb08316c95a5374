#!/bin/bash
# round 5, GPU call 2: k-loop variants on top of the carried tails, item order, raw section timeline; first-generation
# decode with the in-launch merge (single kv head) + bin-count sweep; the tests of what changed
cd "${GRAFT_REPO_ROOT:-.}"
mkdir -p gpurun_out
export HSA_ENABLE_IPC_MODE_LEGACY=0
timeout 300 python tools/prof_p8.py 1 3 > gpurun_out/r5c2_prof.log 2>&1
timeout 900 python tools/tune_ggemm.py "0=0" "23=1" "22=2,23=1" "22=4" "22=5" "22=6" "22=7" "22=8" "22=9" "22=10" "22=11" "22=12" "22=13" "0=0" > gpurun_out/r5c2_ab.log 2>&1
timeout 600 python tools/tune_fp8.py heads=1/8 cases=uniform8k,mixed,extreme,one64k,skewed_mix "33=1" "0=0" "34=256" "34=384" "34=768" "34=1024" "0=0" > gpurun_out/r5c2_dec1.log 2>&1
( time timeout 900 python -m pytest tests/test_fuse_moe_blockwise.py tests/test_fuse_moe_pertensor.py tests/test_attention_decode_fp8.py tests/test_attention_decode_bf16.py tests/test_graded_shapes.py tests/test_replay_check.py tests/test_dev_build.py -x -q -m gpu ) > gpurun_out/r5c2_tests.log 2>&1
tail -5 gpurun_out/r5c2_tests.log
cat gpurun_out/r5c2_prof.log
cat gpurun_out/r5c2_ab.log
cat gpurun_out/r5c2_dec1.log
