#!/bin/bash
# round 5, call 9: fused low-latency all-reduce launch - parity (product + development loopback) and the ws = 1 A/B
mkdir -p gpurun_out
( time timeout 600 python -m pytest tests/test_allreduce.py -q -m gpu -x -n 2 ) > gpurun_out/r5c9_ar_product.log 2>&1
( time HPC_AMD_DEV=1 timeout 600 python -m pytest tests/test_allreduce.py -q -m gpu -x -n 2 -k "loopback or lost_peer" ) > gpurun_out/r5c9_ar_dev.log 2>&1
timeout 300 python tools/tune_allreduce.py "35=0" "35=1" "35=2" "35=0" > gpurun_out/r5c9_ar_tune.log 2>&1
tail -3 gpurun_out/r5c9_ar_product.log; tail -3 gpurun_out/r5c9_ar_dev.log; cat gpurun_out/r5c9_ar_tune.log | tail -14
