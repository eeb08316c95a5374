"""Development tool: condense gpurun_out/ rocprofv3 outputs into the tracked profiles/ summaries.
usage: python tools/make_profiles.py <round-tag>   (expects gpurun_out/prof_bench, pmc_fetch, pmc_write, bench_final.log)"""
import csv, json, sys
from pathlib import Path
ROOT = Path(__file__).resolve().parent.parent
tag = sys.argv[1] if len(sys.argv) > 1 else "round1"
out = ROOT / "profiles"; out.mkdir(exist_ok=True)
g = ROOT / "gpurun_out"
def ours(name): return name.startswith(("hpc::", "void hpc::"))
for src, dst in (("prof_bench", f"{tag}_bench_kernel_stats.csv"), ("prof_all", f"{tag}_hotpath_kernel_stats.csv")):
    f = next(iter((g / src).rglob("*kernel_stats.csv")), None)
    if f is None: continue
    rows = list(csv.DictReader(open(f)))
    with open(out / dst, "w") as fo:
        w = csv.DictWriter(fo, fieldnames=list(rows[0].keys())); w.writeheader()
        for r in rows:
            if ours(r["Name"]): w.writerow(r)
agg = {}
for name, d in (("FETCH_SIZE", "pmc_fetch"), ("WRITE_SIZE", "pmc_write")):
    f = next(iter((g / d).rglob("*counter_collection.csv")), None)
    if f is None: continue
    vals = [float(r["Counter_Value"]) for r in csv.DictReader(open(f)) if "decode_kernel<false" in r["Kernel_Name"]]
    if vals: agg[name] = {"launches": len(vals), "avg_KB": sum(vals) / len(vals), "min_KB": min(vals), "max_KB": max(vals)}
if len(agg) == 2:
    hbm = int((2 * agg["FETCH_SIZE"]["avg_KB"] + agg["WRITE_SIZE"]["avg_KB"]) * 1024)
    alg = 2149580800
    json.dump({"kernel": "hpc::decode::decode_kernel<false,1,1,2> (bf16, BASELINE configs[1] workload)",
               "command": "rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -- python tools/prof_run.py decode (separate --pmc WRITE_SIZE pass)",
               "raw": agg,
               "correction": "gfx950 FETCH_SIZE reports half of the bytes of a wide coalesced streaming read (MI355X_MICROARCH.md, HBM): read bytes = 2*FETCH_SIZE*1024; WRITE_SIZE as reported (uncalibrated)",
               "hbm_bytes_per_launch": hbm, "algorithmic_bytes_per_launch": alg,
               "ratio_traffic_to_algorithmic": round(hbm / alg, 4)}, open(out / "decode_bf16_pmc.json", "w"), indent=1)
b = g / "bench_final.log"
if b.exists():
    line = [l for l in b.read_text().splitlines() if l.startswith("{")][-1]
    (out / f"{tag}_bench_n1.json").write_text(json.dumps(json.loads(line), indent=1) + "\n")
print(sorted(p.name for p in out.iterdir()))
