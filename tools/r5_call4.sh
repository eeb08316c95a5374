#!/bin/bash
# round 5, GPU call 4: s_setprio placement in the head-pair decode kernel (headline workload), T = 256 MoE on the tail body,
# tail-only groups on the three kernels, one bench.py line
cd "${GRAFT_REPO_ROOT:-.}"
mkdir -p gpurun_out
export HSA_ENABLE_IPC_MODE_LEGACY=0
timeout 600 python tools/tune_fp8.py cases=mixed,uniform8k "0=0" "35=1" "35=2" "35=3" "0=0" "35=1" > gpurun_out/r5c4_prio.log 2>&1
timeout 300 python tools/tune_moe.py --tokens 256 "0=0" "3=4" "0=0" "3=4" > gpurun_out/r5c4_moe256.log 2>&1
timeout 300 python tools/tune_moe.py --tokens 1024 "0=0" "3=4" "0=0" > gpurun_out/r5c4_moe1024.log 2>&1
timeout 300 python tools/tune_ggemm.py --rows=32 --only=even32 "3=4" "3=2" "3=1" "3=4,21=2" > gpurun_out/r5c4_tails.log 2>&1
( time timeout 900 python bench.py --steps 20 --warmup 5 ) > gpurun_out/r5c4_bench.log 2> gpurun_out/r5c4_bench.err
cp bench_last.json gpurun_out/r5c4_bench_last.json 2>/dev/null
cat gpurun_out/r5c4_prio.log gpurun_out/r5c4_moe256.log gpurun_out/r5c4_moe1024.log gpurun_out/r5c4_tails.log
tail -5 gpurun_out/r5c4_bench.err
python - <<'PY'
import json
d=json.load(open('gpurun_out/r5c4_bench_last.json'))
ex=d.pop('extras',{})
print(json.dumps(d)[:3000])
for k,v in ex.items():
    print(k, json.dumps(v)[:400])
PY
