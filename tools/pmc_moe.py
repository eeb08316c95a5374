"""Development tool: a few plain (no hipGraph) launches of the two fused-MoE APIs at T=4096 so a rocprofv3
--pmc pass can attribute SQ counters to the tiled grouped-GEMM kernels.
usage: rocprofv3 --pmc <SQ counters> --kernel-trace --output-format csv -d <dir> -- python tools/pmc_moe.py
       python tools/pmc_moe.py --summarise <out.json> <dir> [<dir>...]   (condense the counter CSVs)"""
import csv, json, sys
from collections import defaultdict
from pathlib import Path
ROOT = Path(__file__).resolve().parent.parent

if len(sys.argv) > 1 and sys.argv[1] == "--summarise":
    # usage: --summarise <out.json> <dir> [<dir> ...]; the grouped GEMM runs twice per MoE call (gate-up, then down)
    acc = defaultdict(lambda: defaultdict(list))
    for d in sys.argv[3:]:
        f = next(iter(Path(d).rglob("*counter_collection.csv")))
        for r in csv.DictReader(open(f)):
            if "hpc::ggemm" in r["Kernel_Name"]:
                name = r["Kernel_Name"].split("(hpc::")[0].replace("(anonymous namespace)::", "").replace("void ", "")
                acc[name][r["Counter_Name"]].append(float(r["Counter_Value"]))
    out = {}
    for name, cs in acc.items():
        for which, par in (("gate_up", 0), ("down", 1)):
            m = {c: sum(v[par::2]) / len(v[par::2]) for c, v in cs.items()}
            # SQ_VALU_MFMA_BUSY_CYCLES counts per SIMD, SQ_BUSY_CU_CYCLES per CU (4 SIMDs)
            m["mfma_busy_frac"] = round(m["SQ_VALU_MFMA_BUSY_CYCLES"] / (4 * m["SQ_BUSY_CU_CYCLES"]), 4)
            m["valu_busy_frac"] = round(m["SQ_ACTIVE_INST_VALU"] * 4 / (4 * m["SQ_BUSY_CU_CYCLES"]), 4)  # quad-cycles
            for c in ("SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY"):
                m[c + "_over_WAVE_CYCLES"] = round(m[c] / m["SQ_WAVE_CYCLES"], 4)
            out[f"{name} [{which}]"] = m
    json.dump(out, open(sys.argv[2], "w"), indent=1)
    print(json.dumps(out, indent=1))
    sys.exit(0)

sys.path.insert(0, str(ROOT / "hpc-ops_amd")); sys.path.insert(0, str(ROOT))
import torch
import hpc

dev = torch.device("cuda", 0)
F8 = torch.float8_e4m3fn
E, k, H, I, T = 64, 8, 4096, 11008, 4096
torch.manual_seed(41)
guw = torch.randint(-80, 80, (E, 2 * I, H), dtype=torch.int8, device=dev).view(F8)
dw = torch.randint(-80, 80, (E, H, I), dtype=torch.int8, device=dev).view(F8)
ids = torch.sort(torch.multinomial(torch.ones(T, E, device=dev), k).to(torch.int32), dim=1)[0]
sc = torch.rand(T, k, device=dev)
x = (torch.randn(T, H, device=dev) / 100).to(F8)
guws = torch.rand(E, 2 * I // 128, (H // 128 + 3) // 4 * 4, device=dev) * 0.02
dws = torch.rand(E, H // 128, (I // 128 + 3) // 4 * 4, device=dev) * 0.02
xs = torch.rand(T, H // 128, device=dev)
gus, ds, ams = torch.rand(E, device=dev) * 0.01, torch.rand(E, device=dev) * 0.01, torch.ones(1, device=dev)
for _ in range(3):
    hpc.fuse_moe_blockwise_fp8(x, xs, guw, guws, dw, dws, ids, sc, 0, E)
    hpc.fuse_moe_pertensor_fp8(x, guw, dw, gus, ds, ams, ids, sc, 0, E)
torch.cuda.synchronize()
print("done")
