"""Development tool: FP8 paged prefill timing (4 x 4096 tokens, 64 / 8 heads; dense, block-sparse at skip 0.5, bf16 form).
usage: python tools/tune_prefill.py ["k=v,k=v" ...]   each argument is one configuration of development registers"""
import os
os.environ.setdefault("HPC_AMD_DEV", "1")
import sys
from pathlib import Path
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT / "hpc-ops_amd")); sys.path.insert(0, str(ROOT))
import torch, bench, hpc
from hpc import _C
dev = torch.device("cuda", 0)
for cfg in (sys.argv[1:] or ["0=0", "0=0"]):
    pairs = [tuple(int(x) for x in kv.split("=")) for kv in cfg.split(",") if kv]
    for k, v in pairs: _C.lib.hpc_dev_tuning_set(k, v)
    print(f"[{cfg}]", bench.extra_prefill(dev, hpc), flush=True)
    for k, v in pairs: _C.lib.hpc_dev_tuning_set(k, 0)
