"""Development tool: time fused MoE blockwise (C4 shape) for several token counts / tuning keys."""
import sys, itertools
from pathlib import Path
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT / "hpc-ops_amd")); sys.path.insert(0, str(ROOT))
import torch, bench, hpc
from hpc import _C
dev = torch.device("cuda", 0)
toks = [int(t) for t in sys.argv[1].split(",")] if len(sys.argv) > 1 else [16, 64, 256]
forced = [int(t) for t in sys.argv[2].split(",")] if len(sys.argv) > 2 else [0]
for f in forced:
    _C.lib.hpc_tuning_set(1, f)
    r = bench.extra_moe(dev, hpc, tokens=toks)
    print("forced_mt", f, r, flush=True)
