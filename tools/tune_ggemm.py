"""Development tool: the two grouped GEMMs of the fused MoE at BASELINE configs[3] timed on their own
(E64, 32768 routed rows; N=22016/K=4096 and N=4096/K=11008), with the routed group sizes of the bench
generator and with exactly 512 rows per group (no partial tiles).
usage: python tools/tune_ggemm.py [--pertensor] ["k=v,k=v" ...]   each argument is one configuration of tuning registers"""
import os
os.environ.setdefault("HPC_AMD_DEV", "1")  # development build of the library: tuning registers
import sys
from pathlib import Path
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT / "hpc-ops_amd")); sys.path.insert(0, str(ROOT))
import torch, bench, hpc
from hpc import _C
dev = torch.device("cuda", 0)
NO_GRAPH = bool(os.environ.get("HPC_NO_GRAPH"))  # plain launches: rocprofv3 --pmc attributes counters per dispatch
PT = "--pertensor" in sys.argv
if PT: sys.argv.remove("--pertensor")
ROWS = [int(a.split("=")[1]) for a in sys.argv if a.startswith("--rows=")]  # extra cases: N rows in every group
ONLY = [a.split("=")[1] for a in sys.argv if a.startswith("--only=")]      # substring filter on the case names
sys.argv = [a for a in sys.argv if not a.startswith("--rows=") and not a.startswith("--only=")]
w = bench.C4
m_ = bench.c4_inputs(dev, w)
E, T, topk = w["num_expert"], w["tokens"], w["topk"]
M = T * topk
routed = torch.bincount(m_["ids"].flatten().long(), minlength=E).to(torch.int32).cpu()
F8 = torch.float8_e4m3fn
def case(seqlens, wt, wsc):
    n, k = wt.shape[1], wt.shape[2]
    cu = torch.cat([torch.zeros(1, dtype=torch.int32), torch.cumsum(seqlens, 0).to(torch.int32)])
    M = int(seqlens.sum())
    avg = M // E
    tile = hpc.aligned_size(avg)
    tiles = (seqlens + tile - 1) // tile
    m_pad = int(tiles.sum()) * tile + 64
    x = (torch.randn(M, k, device=dev) / 10).to(F8)
    xs_t = torch.rand(k // 128, m_pad, device=dev) + 0.5
    sl, cud = seqlens.to(dev), cu.to(dev)
    out = torch.empty(M, n, dtype=torch.bfloat16, device=dev)
    ys = torch.rand(E, device=dev) * 0.01 + 0.01
    if PT:
        return lambda: hpc.group_gemm_pertensor_fp8(x, wt, sl, cud, ys, num_seq_per_group_avg=avg, output=out), 2.0 * M * n * k, out
    return lambda: hpc.group_gemm_blockwise_fp8(x, wt, sl, cud, xs_t, wsc, num_seq_per_group_avg=avg, output=out), 2.0 * M * n * k, out
cases, first = [], {}
for nm, sl in [("routed", routed), ("even512", torch.full((E,), M // E, dtype=torch.int32))] + \
              [(f"even{r}", torch.full((E,), r, dtype=torch.int32)) for r in ROWS]:
    cases.append((f"gate_up {nm}",) + case(sl, m_["guw"], m_["guws"]))
    cases.append((f"down    {nm}",) + case(sl, m_["dw"], m_["dws"]))
if ONLY:
    cases = [c for c in cases if any(o in c[0] for o in ONLY)]
for cfg in (sys.argv[1:] or ["3=2", "3=4"]):
    pairs = [tuple(int(x) for x in kv.split("=")) for kv in cfg.split(",") if kv]
    for k, v in pairs: _C.lib.hpc_dev_tuning_set(k, v)
    for nm, fn, fl, out in cases:
        out.zero_()
        us = bench.timed(fn, iters=3 if NO_GRAPH else 10, warm=1 if NO_GRAPH else 3, graph=not NO_GRAPH)
        torch.cuda.synchronize()
        if nm not in first:  # every configuration must reproduce the first one's output (same arithmetic, other schedule)
            first[nm] = out.clone()
            same = "reference"
        else:
            same = "identical" if torch.equal(first[nm], out) else f"DIFFERS max {float((first[nm].float() - out.float()).abs().max()):.3g}"
        print(f"[{cfg or 'default':>10}] {nm}: {us:9.1f} us  {fl / us / 1e6:8.1f} TFLOP/s  ({same})", flush=True)
    for k, v in pairs: _C.lib.hpc_dev_tuning_set(k, 0)
