"""Development tool: quick timing of both fused-MoE APIs (C4 shape) at large T for same-box A/B runs.
usage: python tools/ab_moe.py [tokens csv] [key:v1,v2,...]   (sweeps one tuning key)"""
import os
os.environ.setdefault("HPC_AMD_DEV", "1")  # development build of the library: tuning registers
import sys
from pathlib import Path
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT / "hpc-ops_amd")); sys.path.insert(0, str(ROOT))
import torch, bench, hpc
from hpc import _C
dev = torch.device("cuda", 0)
F8 = torch.float8_e4m3fn
toks = [int(t) for t in sys.argv[1].split(",")] if len(sys.argv) > 1 else [4096]
E, k, H, I = 64, 8, 4096, 11008
import os
torch.manual_seed(41)
guw = torch.randint(-80, 80, (E, 2 * I, H), dtype=torch.int8, device=dev).view(F8)
dw = torch.randint(-80, 80, (E, H, I), dtype=torch.int8, device=dev).view(F8)
guws = torch.rand(E, 2 * I // 128, (H // 128 + 3) // 4 * 4, device=dev) * 0.02
dws = torch.rand(E, H // 128, (I // 128 + 3) // 4 * 4, device=dev) * 0.02
if os.environ.get('HPC_AB_ZERO'):
    guw.view(torch.int8).zero_(); dw.view(torch.int8).zero_()
gus, ds, ams = torch.rand(E, device=dev) * 0.01, torch.rand(E, device=dev) * 0.01, torch.ones(1, device=dev)
key, vals = (int(sys.argv[2].split(':')[0]), [int(v) for v in sys.argv[2].split(':')[1].split(',')]) if len(sys.argv) > 2 else (15, [0])
for T in toks:
    ids = torch.sort(torch.multinomial(torch.ones(T, E, device=dev), k).to(torch.int32), dim=1)[0]
    sc = torch.rand(T, k, device=dev)
    x = (torch.randn(T, H, device=dev) / 100).to(F8)
    xs = torch.rand(T, H // 128, device=dev)
    for val in vals:
        _C.lib.hpc_dev_tuning_set(key, val)
        bw = bench.timed(lambda: hpc.fuse_moe_blockwise_fp8(x, xs, guw, guws, dw, dws, ids, sc, 0, E), iters=10, warm=2)
        pt = bench.timed(lambda: hpc.fuse_moe_pertensor_fp8(x, guw, dw, gus, ds, ams, ids, sc, 0, E), iters=10, warm=2)
        fl = 2.0 * T * k * 3 * I * H
        print(f"ab key{key}={val} T{T} blockwise {bw:.0f} us {fl / bw / 1e9:.3f} PF | pertensor {pt:.0f} us {fl / pt / 1e9:.3f} PF", flush=True)
