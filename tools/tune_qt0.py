"""Development tool: FP8 decode with per-token K scales (quant_type 0) at the C3 length mix, 8 / 64 heads - the head-pair kernel
(round 6, NHD pages) against the first-generation kernel (development key 54 = 1).  Runs bench.extra_decode_holes (in-run parity
of the timed call against the oracle, 10 calls per hipGraph replay) under each configuration.
usage: python tools/tune_qt0.py ["k=v,k=v" ...]"""
import os
os.environ.setdefault("HPC_AMD_DEV", "1")
import sys
from pathlib import Path
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT / "hpc-ops_amd")); sys.path.insert(0, str(ROOT))
import torch, bench, hpc
from hpc import _C
dev = torch.device("cuda", 0)
for cfg in (sys.argv[1:] or ["54=1", "0=0", "54=1", "0=0"]):
    pairs = [tuple(int(x) for x in kv.split("=")) for kv in cfg.split(",") if kv]
    for k, v in pairs: _C.lib.hpc_dev_tuning_set(k, v)
    out = bench.extra_decode_holes(dev, hpc)
    for name in ("decode_fp8_qt0_mixed_nhd", "decode_fp8_qt0_mixed_hnd"):
        r = out[name]
        print(f"[{cfg:>8}] {name}: {r['us']:7.1f} us  {r['GBps']:7.1f} GB/s  {r['hbm_frac_of_8TBps']:.3f}  parity max_abs_err {r['parity']['max_abs_err']}", flush=True)
    for k, v in pairs: _C.lib.hpc_dev_tuning_set(k, 0)
