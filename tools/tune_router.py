"""Development tool: router GEMM (gemm_bf16xfp32) timing at large m: the LDS-staged tile kernel against the round 1-4 kernel
(development key 40 = 1).  usage: python tools/tune_router.py"""
import os
os.environ.setdefault("HPC_AMD_DEV", "1")
import sys
from pathlib import Path
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT / "hpc-ops_amd")); sys.path.insert(0, str(ROOT))
import torch, bench, hpc
from hpc import _C
dev = torch.device("cuda", 0)
torch.manual_seed(0)
for (m, n, k) in ((4096, 256, 4096), (2048, 256, 4096), (1024, 256, 4096), (512, 256, 4096), (304, 256, 4096), (4096, 128, 7168), (8192, 256, 7168), (16384, 128, 4096)):
    x = torch.randn(m, k, device=dev).bfloat16()
    w = torch.randn(n, k, device=dev)
    wh = w.bfloat16(); wl = ((w - wh.float()) * 256).bfloat16()
    flag = hpc.get_gemm_bf16xfp32_workspace(n, max(m, 8192))
    outs = {}
    for name, key, skip, cap in (("old", 1, 0, 16), ("tile", 0, 0, 0), ("cap16", 0, 0, 16), ("cap4", 0, 0, 4), ("no_wx", 0, 3, 0), ("tile", 0, 0, 0)):
        _C.lib.hpc_dev_tuning_set(40, key); _C.lib.hpc_dev_tuning_set(41, skip); _C.lib.hpc_dev_tuning_set(45, cap)
        flag.zero_()
        y = hpc.gemm_bf16xfp32(x, wh, wl, 1 / 256, True, True, flag)
        outs[name] = y
        us = bench.timed(lambda: hpc.gemm_bf16xfp32(x, wh, wl, 1 / 256, True, True, flag), graph=True, reps=20)
        print(f"[{name:>4}] m={m} n={n} k={k}: {us:8.1f} us  {4 * m * n * k / us / 1e6:8.1f} TFLOP/s  {(m * k * 2 + 2 * n * k * 2 + m * n * 4) / us / 1e3:7.1f} GB/s", flush=True)
    _C.lib.hpc_dev_tuning_set(40, 0); _C.lib.hpc_dev_tuning_set(41, 0); _C.lib.hpc_dev_tuning_set(45, 0)
    ref = x.float() @ (wh.float() + wl.float() / 256).t()
    print(f"    max |tile - old| = {(outs['tile'] - outs['old']).abs().max().item():.3e}   max |tile - fp32 matmul| = {(outs['tile'] - ref).abs().max().item():.3e}  (|ref| max {ref.abs().max().item():.1f})", flush=True)
