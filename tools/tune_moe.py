"""Development tool: fused MoE FP8 blockwise at BASELINE configs[3] (E64 / top-8 / H4096 / I11008), timing the
whole op and (per-kernel) the two grouped GEMMs under development tuning keys.
usage: python tools/tune_moe.py [--tokens 4096] ["k=v,k=v" ...]   each argument is one configuration"""
import os
os.environ.setdefault("HPC_AMD_DEV", "1")  # development build of the library: tuning registers
import sys
from pathlib import Path
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT / "hpc-ops_amd")); sys.path.insert(0, str(ROOT))
import torch, bench, hpc
from hpc import _C
dev = torch.device("cuda", 0)
args = sys.argv[1:]
T = 4096
if args and args[0] == "--tokens":
    T = int(args[1]); args = args[2:]
w = bench.C4
m = bench.c4_inputs(dev, w, tokens=T)
E = w["num_expert"]
def step():
    return hpc.fuse_moe_blockwise_fp8(m["x"], m["x_scale"], m["guw"], m["guws"], m["dw"], m["dws"], m["ids"], m["scale"], 0, E)
ref = None
for cfg in (args or ["3=2", "3=4", ""]):
    pairs = [tuple(int(x) for x in kv.split("=")) for kv in cfg.split(",") if kv]
    for k, v in pairs: _C.lib.hpc_dev_tuning_set(k, v)
    y = step(); torch.cuda.synchronize()
    if ref is None: ref = y.float()
    err = float((y.float() - ref).abs().max())
    us = bench.timed(step, iters=10, warm=3, graph=True)
    print(f"[{cfg or 'default':>10}] T={T}: {us:9.1f} us  {bench.c4_flops(T, w) / us / 1e6:8.1f} TFLOP/s  max|y - y_first_cfg| = {err:.4g}", flush=True)
    for k, v in pairs: _C.lib.hpc_dev_tuning_set(k, 0)
