#!/bin/bash
# round 5, call 10: routing prep of the fused MoE (count / slot kernels with 16-byte id loads): parity + timing
cd "${GRAFT_REPO_ROOT:-.}"
mkdir -p gpurun_out
[ -n "$SKIP_TESTS" ] || ( time timeout 900 python -m pytest tests/test_fuse_moe_blockwise.py tests/test_fuse_moe_pertensor.py tests/test_graded_shapes.py -x -q -m gpu -k "routing or moe" ) > gpurun_out/r5c10_tests.log 2>&1
tail -n 5 gpurun_out/r5c10_tests.log
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/prof10 && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof10 -o moe -- python $GRAFT_REPO_ROOT/tools/ab_moe.py 256,1024,4096 > $GRAFT_REPO_ROOT/gpurun_out/r5c10_ab.log 2>&1
cd $GRAFT_REPO_ROOT
f=$(find /tmp/prof10 -name "*kernel_stats.csv" | head -1)
cp "$f" gpurun_out/r5c10_kernel_stats.csv
grep -v "^\[rocprof" gpurun_out/r5c10_ab.log | tail -n 8
cut -c1-160 gpurun_out/r5c10_kernel_stats.csv | head -12
f2=$(find /tmp/prof10 -name "*kernel_trace.csv" | head -1)
python - "$f2" <<'PY'
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
d = collections.defaultdict(list)
for r in rows:
    nm = r["Kernel_Name"]
    if "count_kernel" in nm or "slot_kernel" in nm or "reduce_kernel" in nm:
        d[nm.split("(")[0][-24:]].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
for k, v in d.items():
    v.sort(); print(k, "n", len(v), "min %.1f med %.1f max %.1f us" % (v[0], v[len(v)//2], v[-1]))
PY
