"""Development tool: time fused MoE blockwise (C4 shape) for several token counts / tuning keys.
usage: tune_moe.py <tokens csv> <key:val,val;...>"""
import sys, itertools
from pathlib import Path
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT / "hpc-ops_amd")); sys.path.insert(0, str(ROOT))
import torch, bench, hpc
from hpc import _C
dev = torch.device("cuda", 0)
toks = [int(t) for t in sys.argv[1].split(",")] if len(sys.argv) > 1 else [16, 64, 256]
sweeps = []
if len(sys.argv) > 2:
    for part in sys.argv[2].split(";"):
        k, vs = part.split(":"); sweeps.append((int(k), [int(v) for v in vs.split(",")]))
for combo in (itertools.product(*[v for _, v in sweeps]) if sweeps else [()]):
    for (k, _), v in zip(sweeps, combo): _C.lib.hpc_dev_tuning_set(k, v)
    r = bench.extra_moe(dev, hpc, tokens=toks)
    for t, d in list(r.values())[0].items():
        print("tune", dict(zip([k for k, _ in sweeps], combo)), t, d, flush=True)
