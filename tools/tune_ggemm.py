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
PT = "--pertensor" in sys.argv
if PT: sys.argv.remove("--pertensor")
w = bench.C4
m_ = bench.c4_inputs(dev, w)
E, T, topk = w["num_expert"], w["tokens"], w["topk"]
M = T * topk
routed = torch.bincount(m_["ids"].flatten().long(), minlength=E).to(torch.int32).cpu()
F8 = torch.float8_e4m3fn
def case(seqlens, wt, wsc):
    n, k = wt.shape[1], wt.shape[2]
    cu = torch.cat([torch.zeros(1, dtype=torch.int32), torch.cumsum(seqlens, 0).to(torch.int32)])
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
        return lambda: hpc.group_gemm_pertensor_fp8(x, wt, sl, cud, ys, num_seq_per_group_avg=avg, output=out), 2.0 * M * n * k
    return lambda: hpc.group_gemm_blockwise_fp8(x, wt, sl, cud, xs_t, wsc, num_seq_per_group_avg=avg, output=out), 2.0 * M * n * k
cases = []
for nm, sl in (("routed", routed), ("even512", torch.full((E,), M // E, dtype=torch.int32))):
    cases.append((f"gate_up {nm}",) + case(sl, m_["guw"], m_["guws"]))
    cases.append((f"down    {nm}",) + case(sl, m_["dw"], m_["dws"]))
for cfg in (sys.argv[1:] or ["3=2", "3=4"]):
    pairs = [tuple(int(x) for x in kv.split("=")) for kv in cfg.split(",") if kv]
    for k, v in pairs: _C.lib.hpc_dev_tuning_set(k, v)
    for nm, fn, fl in cases:
        us = bench.timed(fn, iters=10, warm=3, graph=True)
        print(f"[{cfg or 'default':>10}] {nm}: {us:9.1f} us  {fl / us / 1e6:8.1f} TFLOP/s", flush=True)
    for k, v in pairs: _C.lib.hpc_dev_tuning_set(k, 0)
