"""Development tool: time hpc.attention_decode_bf16 variants on one GPU (not product, not a test).
usage: python tools/tune_decode.py [--layout NHD|HND] [--seq 8192] [--batch 64]"""
import os
os.environ.setdefault("HPC_AMD_DEV", "1")  # development build of the library: tuning registers
import argparse, math, sys, os
from pathlib import Path
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT / "hpc-ops_amd")); sys.path.insert(0, str(ROOT))
import torch
import hpc
from hpc import _C

def make(dev, B, S, Hkv, Hq, layout, P=64, D=128):
    torch.manual_seed(41)
    kv_lens = torch.full((B,), S, dtype=torch.int32, device=dev)
    nb = (S + P - 1) // P
    nblk = int(B * nb * 1.2) + B + 8
    q = torch.randn(B, Hq, D, dtype=torch.bfloat16, device=dev) / math.sqrt(D)
    if layout == "NHD":
        k = torch.randn(nblk, P, Hkv, D, dtype=torch.bfloat16, device=dev) / math.sqrt(D)
        v = torch.randn(nblk, P, Hkv, D, dtype=torch.bfloat16, device=dev)
    else:
        k = (torch.randn(nblk, Hkv, P, D, dtype=torch.bfloat16, device=dev) / math.sqrt(D)).permute(0, 2, 1, 3)
        v = torch.randn(nblk, Hkv, P, D, dtype=torch.bfloat16, device=dev).permute(0, 2, 1, 3)
    bid = torch.randperm(nblk, device=dev)[: B * nb].to(torch.int32).reshape(B, nb).contiguous()
    return q, k, v, bid, kv_lens

def timeit(fn, iters=30, warm=5):
    for _ in range(warm): fn()
    torch.cuda.synchronize()
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(iters + 1)]
    ev[0].record()
    for i in range(iters):
        fn(); ev[i + 1].record()
    torch.cuda.synchronize()
    ts = sorted(ev[i].elapsed_time(ev[i + 1]) for i in range(iters))
    return ts[len(ts) // 2] * 1e3

def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=64); ap.add_argument("--seq", type=int, default=8192)
    ap.add_argument("--hkv", type=int, default=8); ap.add_argument("--hq", type=int, default=64)
    ap.add_argument("--keys", type=str, default="0")  # tuning keys to sweep, "k:v1,v2;k2:..."
    a = ap.parse_args()
    dev = torch.device("cuda", 0)
    sweeps = []
    for part in a.keys.split(";"):
        if ":" in part:
            k, vs = part.split(":"); sweeps.append((int(k), [int(x) for x in vs.split(",")]))
    for layout in ("NHD", "HND"):
        q, k, v, bid, lens = make(dev, a.batch, a.seq, a.hkv, a.hq, layout)
        tm = hpc.get_attention_decode_task_workspace(a.batch, a.seq, a.hkv, 64)
        out = torch.empty_like(q)
        nbytes = a.batch * a.seq * a.hkv * 256 * 2
        def run():
            hpc.attention_decode_bf16(q, k, v, bid, lens, 0, True, True, tm, None, out)
        import itertools
        combos = list(itertools.product(*[vs for _, vs in sweeps])) if sweeps else [()]
        for combo in combos:
            for (key, _), val in zip(sweeps, combo): _C.lib.hpc_dev_tuning_set(key, val)
            hpc.assign_attention_decode_task(lens, tm, a.hkv, 1, True, 64)
            us = timeit(run)
            print(f"{layout} B{a.batch} S{a.seq} tune={dict(zip([k for k,_ in sweeps], combo))}: {us:8.1f} us  {nbytes/us/1e3:7.1f} GB/s", flush=True)
        for key, _ in sweeps: _C.lib.hpc_dev_tuning_set(key, 0)

if __name__ == "__main__":
    main()
