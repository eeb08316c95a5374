cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r6prof_new; rm -rf $O; mkdir -p $O
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O -- python $R/tools/prof_new_kernels.py > $O/run.log 2>&1; echo rc=$?
f=$(find $O -name "*kernel_stats.csv" | head -1)
python - "$f" <<'PY'
import csv, sys
rows=[r for r in csv.DictReader(open(sys.argv[1])) if "hpc::" in r["Name"]]
for r in rows: print(f"{r['Name'][:130]:<130} calls {r['Calls']:>4} avg {float(r['AverageNs'])/1e3:9.1f} us min {float(r['MinNs'])/1e3:9.1f}")
PY
find $O -name "*kernel_trace.csv" -delete
