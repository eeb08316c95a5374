#!/bin/bash
# round 5, call 12: memory-side traffic (FETCH_SIZE = L2 misses) of the grouped GEMMs: routed sizes with the tail body, with the
# tails on the half-tile body, and 512 rows per group
cd "${GRAFT_REPO_ROOT:-.}"
mkdir -p gpurun_out
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
for cfg in "0=0" "21=2"; do
  rm -rf /tmp/pmc12
  HPC_NO_GRAPH=1 timeout 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d /tmp/pmc12 -- python $R/tools/tune_ggemm.py "$cfg" > /tmp/pmc12.log 2>&1
  f=$(find /tmp/pmc12 -name "*counter_collection.csv" | head -1)
  python - "$f" "$cfg" <<'PY'
import csv, sys, collections
rows = [r for r in csv.DictReader(open(sys.argv[1])) if "gemm_fp8_p8" in r["Kernel_Name"] and r["Counter_Name"] == "FETCH_SIZE"]
# dispatch order: cases gate_up routed, down routed, gate_up even512, down even512; each 1 warm + 3 timed
vals = [float(r["Counter_Value"]) for r in rows]
per = len(vals) // 4
for i, nm in enumerate(["gate_up routed", "down routed", "gate_up even512", "down even512"]):
    v = vals[i * per:(i + 1) * per]
    print(f"[{sys.argv[2]}] {nm}: FETCH_SIZE {sum(v)/len(v)/1e6:.3f} GB(KB units) x2 = {2*sum(v)/len(v)*1024/1e9:.2f} GB read per launch ({len(v)} launches)")
PY
done 2>&1 | tee $R/gpurun_out/r5c12_fetch.log
