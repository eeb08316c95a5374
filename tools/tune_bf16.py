"""Development tool: bf16 decode timing (BASELINE configs[1]: uniform 8k; plus 4k, random and mixed lengths), NHD pages,
second generation (head-pair kernel) against the first (development key 28 = 1).
usage: python tools/tune_bf16.py"""
import os
os.environ.setdefault("HPC_AMD_DEV", "1")  # development build of the library: tuning registers
import math, sys
from pathlib import Path
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT / "hpc-ops_amd")); sys.path.insert(0, str(ROOT))
import torch, bench, hpc
from hpc import _C
dev = torch.device("cuda", 0)
B, D, P, Hkv, Hq = 64, 128, 64, 8, 64
g = torch.Generator().manual_seed(41)
cases = (("uniform8k", torch.full((B,), 8192, dtype=torch.int32)), ("uniform4k", torch.full((B,), 4096, dtype=torch.int32)),
         ("randint_1_8192", torch.randint(1, 8192, (B,), generator=g).to(torch.int32)), ("c3_mix", bench.c3_lens()))
for name, lens_c in cases:
    nb = (lens_c + P - 1) // P
    total = int(nb.sum()); nblk = int(total * 1.2) + B + 8
    torch.manual_seed(41)
    q = torch.randn(B, Hq, D, dtype=torch.bfloat16, device=dev) / math.sqrt(D)
    k = torch.randn(nblk, P, Hkv, D, dtype=torch.bfloat16, device=dev) / math.sqrt(D)
    v = torch.randn(nblk, P, Hkv, D, dtype=torch.bfloat16, device=dev)
    perm = torch.randperm(nblk, device=dev)[:total].to(torch.int32)
    bid = torch.zeros(B, int(nb.max()), dtype=torch.int32, device=dev)
    off = 0
    for i, n in enumerate(nb.tolist()):
        bid[i, :n] = perm[off:off + n]; off += n
    lens = lens_c.to(dev)
    tm = hpc.get_attention_decode_task_workspace(B, int(lens_c.max()), Hkv, 64)
    hpc.assign_attention_decode_task(lens, tm, Hkv, 1, True, 64)
    o = torch.empty_like(q)
    kvb = int(lens_c.sum()) * Hkv * 512
    outs = {}
    # order matters on these boxes (the first measurement after a pause runs 4-7 % faster): A B A B, read the later pair
    plan = (("first", 1, 0), ("head_pair", 0, 0), ("first", 1, 0), ("head_pair", 0, 0))
    if len(sys.argv) > 1:  # "36=34": head-pair kernel with and without that register, A B A B
        rk = int(sys.argv[1].split("=")[0])
        plan = (("head_pair", 0, 0),) + tuple((f"{rk}={v}", 0, int(v)) for v in sys.argv[1].split("=")[1].split("|")) + (("head_pair", 0, 0),)
    for gen, key, reg in plan:
        _C.lib.hpc_dev_tuning_set(28, key)
        if len(sys.argv) > 1: _C.lib.hpc_dev_tuning_set(rk, reg)
        us = bench.timed(lambda: hpc.attention_decode_bf16(q, k, v, bid, lens, 0, True, True, tm, None, o), graph=True, iters=30, reps=10)
        outs[gen] = o.clone()
        print(f"[{gen:>9}] bf16 {name:<15} 8/64: {us:8.1f} us {kvb / us / 1e3:8.1f} GB/s {kvb / us / 1e3 / 8000:.3f}", flush=True)
    _C.lib.hpc_dev_tuning_set(28, 0)
    if len(sys.argv) > 1: _C.lib.hpc_dev_tuning_set(rk, 0)
    ks = list(outs)
    print(f"    max |{ks[0]} - {ks[-1]}| = {(outs[ks[0]].float() - outs[ks[-1]].float()).abs().max().item():.5f}", flush=True)
    del k, v
