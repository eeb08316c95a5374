#!/bin/bash
# round 5, GPU call 8: smoke, first-generation decode with task records in registers (tests + single-kv-head timings), s_setprio in the prefill kernel
cd "${GRAFT_REPO_ROOT:-.}"
mkdir -p gpurun_out
export HSA_ENABLE_IPC_MODE_LEGACY=0
python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/r5c8_smoke.log 2>&1; tail -2 gpurun_out/r5c8_smoke.log
( timeout 900 python -m pytest tests/test_attention_decode_fp8.py tests/test_attention_decode_bf16.py tests/test_oracle_golden.py tests/test_sched.py -x -q -m gpu ) > gpurun_out/r5c8_tests.log 2>&1
tail -3 gpurun_out/r5c8_tests.log
( HPC_AMD_DEV=1 timeout 600 python -m pytest tests/test_attention_decode_fp8.py tests/test_attention_decode_bf16.py -x -q -m "gpu and dev" -n 4 ) > gpurun_out/r5c8_tests_dev.log 2>&1
tail -3 gpurun_out/r5c8_tests_dev.log
timeout 600 python tools/tune_fp8.py heads=1/8 cases=uniform8k,mixed,extreme,one64k,skewed_mix,uniform512 "0=0" "0=0" > gpurun_out/r5c8_dec1.log 2>&1
timeout 600 python tools/tune_prefill.py "0=0" "36=1" "0=0" "36=1" > gpurun_out/r5c8_prefill.log 2>&1
cat gpurun_out/r5c8_dec1.log gpurun_out/r5c8_prefill.log
