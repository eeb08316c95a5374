"""Development tool: a few plain launches of the FP8 paged prefill (4 x 4096 tokens, 64 / 8 heads) for rocprofv3 --pmc.
usage: rocprofv3 --pmc <SQ counters> --kernel-trace --output-format csv -d <dir> -- python tools/pmc_prefill.py
       python tools/pmc_prefill.py --summarise <dir> [<dir>...]"""
import csv, json, math, sys
from collections import defaultdict
from pathlib import Path
ROOT = Path(__file__).resolve().parent.parent
if len(sys.argv) > 1 and sys.argv[1] == "--summarise":
    acc = defaultdict(list)
    for d in sys.argv[2:]:
        for f in Path(d).rglob("*counter_collection.csv"):
            for r in csv.DictReader(open(f)):
                if "prefill_fp8_kernel" in r["Kernel_Name"]:
                    acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
    m = {c: sum(v) / len(v) for c, v in acc.items()}
    if "SQ_BUSY_CU_CYCLES" in m:
        for c in ("SQ_VALU_MFMA_BUSY_CYCLES",):
            if c in m: m["mfma_busy_frac"] = round(m[c] / (4 * m["SQ_BUSY_CU_CYCLES"]), 4)
        if "SQ_ACTIVE_INST_VALU" in m: m["valu_busy_frac"] = round(m["SQ_ACTIVE_INST_VALU"] * 4 / (4 * m["SQ_BUSY_CU_CYCLES"]), 4)
    if "SQ_WAVE_CYCLES" in m:
        for c in list(m):
            if c.startswith("SQ_WAIT") or c.startswith("SQ_ACTIVE_INST") or c.startswith("SQ_INST_CYCLES"):
                m[c + "_over_WAVE_CYCLES"] = round(m[c] / m["SQ_WAVE_CYCLES"], 4)
    print(json.dumps(m, indent=1))
    sys.exit(0)
sys.path.insert(0, str(ROOT / "hpc-ops_amd")); sys.path.insert(0, str(ROOT))
import torch, hpc
dev = torch.device("cuda", 0)
B, S, Hq, Hkv, D, P = 4, 4096, 64, 8, 128, 64
torch.manual_seed(41)
q = (torch.randn(B * S, Hq, D, device=dev) / math.sqrt(D)).to(torch.float8_e4m3fn)
nb = S // P
kc = (torch.randn(B * nb + 8, P, Hkv, D, device=dev) / math.sqrt(D)).to(torch.float8_e4m3fn)
vc = torch.randn(B * nb + 8, P, Hkv, D, device=dev).to(torch.float8_e4m3fn)
bid = torch.randperm(B * nb + 8, device=dev)[: B * nb].to(torch.int32).reshape(B, nb).contiguous()
qs = torch.rand(B, Hq, S, device=dev) * 0.1 + 0.01
ks, vs = torch.tensor([0.5], device=dev), torch.tensor([0.7], device=dev)
cu = torch.arange(0, (B + 1) * S, S, dtype=torch.int32, device=dev)
lens = torch.full((B,), S, dtype=torch.int32, device=dev)
y = torch.empty(B * S, Hq, D, dtype=torch.bfloat16, device=dev)
for _ in range(3):
    hpc.attention_with_kvcache_prefill_fp8(q, kc, vc, qs, ks, vs, cu, bid, lens, S, output=y)
torch.cuda.synchronize()
print("done")
