"""Development tool: condense gpurun_out/r6prof (tools/round6_profiles.sh) into the tracked profiles/ files.
usage: python tools/round6_summarise.py [commit-tag]"""
import csv, json, subprocess, sys
from collections import defaultdict
from pathlib import Path
ROOT = Path(__file__).resolve().parent.parent
G = ROOT / "gpurun_out" / "r6prof"
OUT = ROOT / "profiles"
tag = sys.argv[1] if len(sys.argv) > 1 else subprocess.run(["git", "rev-parse", "--short=7", "HEAD"], cwd=ROOT, capture_output=True, text=True).stdout.strip()

# 1. kernel stats of the bench run (our kernels only)
f = max((G / "bench").rglob("*kernel_stats.csv"), key=lambda q: q.stat().st_mtime, default=None)  # the newest pass (gpurun_out accumulates)
if f:
    rows = [r for r in csv.DictReader(open(f)) if "hpc::" in r["Name"]]
    with open(OUT / "round6_bench_kernel_stats.csv", "w") as fo:
        w = csv.DictWriter(fo, fieldnames=list(rows[0].keys())); w.writeheader(); w.writerows(rows)
    print("kernel stats:", [(r["Name"][:60], r.get("AverageNs") or r.get("Average")) for r in rows[:4]])
b = G / "bench.log"
if b.exists():
    lines = [l for l in b.read_text().splitlines() if l.startswith("{")]
    if lines:
        (OUT / "round6_bench_under_rocprof.json").write_text(json.dumps(json.loads(lines[-1]), indent=1) + "\n")

# 2. decode PMC
acc = defaultdict(lambda: defaultdict(list))
for d in ("pmc_fetch", "pmc_write", "pmc_sq"):
    for f in (G / d).rglob("*counter_collection.csv"):
        for r in csv.DictReader(open(f)):
            if "decode" in r["Kernel_Name"]:
                name = r["Kernel_Name"].split("(")[0].replace("void ", "")
                acc[name][r["Counter_Name"]].append(float(r["Counter_Value"]))
kern = {}
for n, cs in acc.items():
    m = {c: round(sum(v) / len(v), 1) for c, v in cs.items()}
    m["launches"] = len(next(iter(cs.values())))
    if "FETCH_SIZE" in m and "WRITE_SIZE" in m:
        m["hbm_bytes_per_launch"] = int((2 * m["FETCH_SIZE"] + m["WRITE_SIZE"]) * 1024)
    if "SQ_WAVE_CYCLES" in m:
        for c in ("SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY"):
            if c in m:
                m[c + "_over_WAVE_CYCLES"] = round(m[c] / m["SQ_WAVE_CYCLES"], 4)
    kern[n] = m
main = next((n for n in kern if "decode2_kernel" in n), None)
if main and "hbm_bytes_per_launch" in kern[main]:
    alg = 370884 * 8 * 256 + 64 * 64 * (128 * 3 + 4)
    json.dump({"workload": "FP8 decode attention, BASELINE configs[2] mix (370884 KV tokens, 8 KV heads): plain launches of tools/pmc_decode.py mixed",
               "taken_at": f"round 6, commit {tag}",
               "command": "tools/round6_profiles.sh: rocprofv3 --pmc FETCH_SIZE --kernel-trace ; separate passes --pmc WRITE_SIZE and --pmc SQ_* GRBM_GUI_ACTIVE",
               "correction": "gfx950 FETCH_SIZE (KB) reports half of the bytes of a wide coalesced streaming read (MI355X_MICROARCH.md, HBM): read bytes = 2 * FETCH_SIZE * 1024; WRITE_SIZE (KB) as reported (uncalibrated)",
               "algorithmic_bytes_per_launch": alg, "kernels": kern, "kernel": main + " (the shipped FP8 NHD path)",
               "hbm_bytes_per_launch": kern[main]["hbm_bytes_per_launch"],
               "ratio_traffic_to_algorithmic": round(kern[main]["hbm_bytes_per_launch"] / alg, 4)},
              open(OUT / "decode_fp8_pmc_r6.json", "w"), indent=1)
    print("decode pmc:", kern[main].get("hbm_bytes_per_launch"), round(kern[main]["hbm_bytes_per_launch"] / alg, 4))

# 3. MoE PMC
if (G / "pmc_moe").exists():
    subprocess.run([sys.executable, str(ROOT / "tools" / "pmc_moe.py"), "--summarise", str(OUT / "moe_tiled_gemm_pmc_r6.json"), str(G / "pmc_moe")])

# 4. the clock record (tools/moe_clock.py)
c = G / "round6_moe_clock.json"
if c.exists():
    j = json.loads(c.read_text())
    j["taken_at"] = f"round 6, commit {tag}"
    (OUT / "round6_moe_clock.json").write_text(json.dumps(j, indent=1) + "\n")
    print("clock:", j.get("summary"))
