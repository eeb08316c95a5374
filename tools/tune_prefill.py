"""Development tool: FP8 paged prefill timing (4 x 4096 tokens, 64 / 8 heads; dense, block-sparse at skip 0.5, bf16 form).
usage: python tools/tune_prefill.py"""
import sys
from pathlib import Path
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT / "hpc-ops_amd")); sys.path.insert(0, str(ROOT))
import torch, bench, hpc
dev = torch.device("cuda", 0)
for _ in range(2):
    print(bench.extra_prefill(dev, hpc), flush=True)
