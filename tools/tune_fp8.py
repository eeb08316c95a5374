"""Development tool: FP8 decode timing (uniform 8k / the C3 mix; NHD pages), sweeping development tuning keys.
usage: python tools/tune_fp8.py [heads=8/64,1/8] [cases=uniform8k,mixed] [layout=hnd] [sq=3] ["k=v,k=v" ...]   each argument is one configuration of tuning registers"""
import os
os.environ.setdefault("HPC_AMD_DEV", "1")  # development build of the library: tuning registers
import math, sys
from pathlib import Path
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT / "hpc-ops_amd")); sys.path.insert(0, str(ROOT))
import torch, bench, hpc
from hpc import _C
dev = torch.device("cuda", 0)
B, D, P = 64, 128, 64
HND = False
SQ = 1
def run(lens_c, heads=(8, 64), graph=True):
    w = dict(bench.C3, num_head_kv=heads[0], num_head_q=heads[1], num_seq_q=SQ)
    inp = bench.c3_inputs(dev, w, lens=lens_c)
    if HND:  # the same logical [pages, P, Hkv, D] view on HND-ordered memory
        for k in ("k_cache", "v_cache"):
            inp[k] = inp[k].view(torch.uint8).permute(0, 2, 1, 3).contiguous().permute(0, 2, 1, 3).view(torch.float8_e4m3fn)
    tm = hpc.get_attention_decode_task_workspace(B, int(lens_c.max()), heads[0], 64)
    hpc.assign_attention_decode_task(inp["kv_lens"], tm, heads[0], SQ, True, 64)
    o = torch.empty(B * SQ, heads[1], D, dtype=torch.bfloat16, device=dev)
    us = bench.timed(lambda: hpc.attention_decode_fp8(inp["q"], inp["k_cache"], inp["v_cache"], inp["block_ids"], inp["kv_lens"],
                     inp["q_scale"], inp["k_scale"], inp["v_scale"], SQ - 1, True,
                     hpc.QuantType.QPERTOKEN_PERHEAD_KPERTENSOR_VPERTENSOR, True, tm, None, o), graph=graph, iters=30, reps=10)
    kvb = int(lens_c.sum()) * heads[0] * 256
    return us, kvb / us / 1e3
mixed = bench.c3_lens()
cases = (("uniform8k", torch.full((B,), 8192, dtype=torch.int32)), ("mixed", mixed),
         ("skewed_mix", torch.tensor([128] * 32 + [4096] * 32, dtype=torch.int32)),
         ("uniform512", torch.full((B,), 512, dtype=torch.int32)),
         ("extreme", torch.tensor([64] * 15 + [16384] + [0] * 48, dtype=torch.int32)),
         ("one64k", torch.tensor([65536] + [4096] * 31 + [0] * 32, dtype=torch.int32)),
         ("two32k", torch.tensor([32768] * 2 + [4096] * 30 + [0] * 32, dtype=torch.int32)),
         ("one64k_7", torch.tensor([65536] + [4096] * 7 + [0] * 56, dtype=torch.int32)),
         ("one128k", torch.tensor([131072] + [4096] * 31 + [0] * 32, dtype=torch.int32)))
args = sys.argv[1:]
heads_list = [(8, 64)]
if args and args[0].startswith("heads="):
    heads_list = [tuple(int(x) for x in h.split("/")) for h in args[0][6:].split(",")]
    args = args[1:]
if args and args[0].startswith("cases="):
    keep = args[0][6:].split(",")
    cases = tuple(c for c in cases if c[0] in keep)
    args = args[1:]
if args and args[0].startswith("layout="):
    HND = args[0][7:].lower() == "hnd"
    args = args[1:]
if args and args[0].startswith("sq="):
    SQ = int(args[0][3:])
    args = args[1:]
configs = args or ["12=1", ""]
for heads in heads_list:
    for cfg in configs:
        pairs = [tuple(int(x) for x in kv.split("=")) for kv in cfg.split(",") if kv]
        for k, v in pairs: _C.lib.hpc_dev_tuning_set(k, v)
        for name, lens in cases:
            us, gb = run(lens, heads=heads)
            print(f"[{cfg or 'default':>14}] fp8 {name:<11} {heads[0]}/{heads[1]}: {us:8.1f} us {gb:8.1f} GB/s {gb/8000:.3f}", flush=True)
        for k, v in pairs: _C.lib.hpc_dev_tuning_set(k, 0)
