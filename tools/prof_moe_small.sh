#!/bin/bash
# Development tool (GPU box): per-kernel rocprofv3 trace of the fused MoE at decode-size batches (T = 16 ... 1024).
# usage: gpurun -- 'bash tools/prof_moe_small.sh "16 64 256 1024"'
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r6moe_small; mkdir -p $O
for T in ${1:-16 64 256 1024}; do
  timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O/T$T -- python $R/tools/tune_moe.py --tokens $T "0=0" > $O/T$T.log 2>&1
  echo "== T=$T"; tail -1 $O/T$T.log
  f=$(find $O/T$T -name "*kernel_stats.csv" | head -1)
  python - "$f" <<'PY'
import csv, sys
for r in list(csv.DictReader(open(sys.argv[1])))[:9]:
    print(f"{r['Name'][:120]:<120} calls {r['Calls']:>4} avg {float(r['AverageNs'])/1e3:9.1f} us  {r['Percentage']}%")
PY
  find $O/T$T -name "*kernel_trace.csv" -delete
done
