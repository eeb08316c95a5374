"""Development tool: FP8 decode timing, uniform vs mixed lengths, NHD vs HND pages."""
import math, sys
from pathlib import Path
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT / "hpc-ops_amd")); sys.path.insert(0, str(ROOT))
import torch, bench, hpc
dev = torch.device("cuda", 0)
B, Hkv, Hq, D, P = 64, 8, 64, 128, 64
def run(lens_c, layout, minlen, quant=1):
    nbl = (lens_c + P - 1) // P
    nblk = int(int(nbl.sum()) * 1.2) + B + 8
    q8 = torch.randn(B, Hq, D, device=dev).to(torch.float8_e4m3fn)
    qs = torch.rand(B, Hq, device=dev) * 0.01 + 0.005
    if layout == "NHD":
        k8 = torch.randn(nblk, P, Hkv, D, device=dev).to(torch.float8_e4m3fn)
        v8 = torch.randn(nblk, P, Hkv, D, device=dev).to(torch.float8_e4m3fn)
    else:
        k8 = torch.randn(nblk, Hkv, P, D, device=dev).to(torch.float8_e4m3fn).permute(0, 2, 1, 3)
        v8 = torch.randn(nblk, Hkv, P, D, device=dev).to(torch.float8_e4m3fn).permute(0, 2, 1, 3)
    bid = torch.zeros(B, int(nbl.max()), dtype=torch.int32, device=dev)
    perm = torch.randperm(nblk, device=dev).to(torch.int32)
    off = 0
    for i, n in enumerate(nbl.tolist()):
        bid[i, :n] = perm[off:off + n]; off += n
    lens = lens_c.to(dev)
    ks = torch.tensor([0.02], device=dev); vs = torch.tensor([0.03], device=dev)
    tm = hpc.get_attention_decode_task_workspace(B, int(lens_c.max()), Hkv, minlen)
    hpc.assign_attention_decode_task(lens, tm, Hkv, 1, True, minlen)
    o = torch.empty(B, Hq, D, dtype=torch.bfloat16, device=dev)
    us = bench.timed(lambda: hpc.attention_decode_fp8(q8, k8, v8, bid, lens, qs, ks, vs, 0, True,
                     hpc.QuantType.QPERTOKEN_PERHEAD_KPERTENSOR_VPERTENSOR, True, tm, None, o))
    kvb = int(lens_c.sum()) * Hkv * 256
    return us, kvb / us / 1e3
g = torch.Generator().manual_seed(41)
mixed = torch.exp(torch.rand(B, generator=g) * (math.log(32768) - math.log(128)) + math.log(128)).to(torch.int32)
for name, lens in (("uniform8k", torch.full((B,), 8192, dtype=torch.int32)), ("mixed", mixed)):
    for layout in ("NHD", "HND"):
        for minlen in (64, 512):
            us, gb = run(lens, layout, minlen)
            print(f"fp8 {name} {layout} minlen{minlen}: {us:8.1f} us {gb:8.1f} GB/s", flush=True)
