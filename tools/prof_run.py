"""Development tool: run each hot-path op a few times so rocprofv3 can attribute kernel time.
usage: rocprofv3 --kernel-trace --stats ... -- python tools/prof_run.py [decode|fp8|moe|all]"""
import sys
from pathlib import Path
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT / "hpc-ops_amd")); sys.path.insert(0, str(ROOT))
import torch
import bench
import hpc

what = sys.argv[1] if len(sys.argv) > 1 else "all"
dev = torch.device("cuda", 0)
if what in ("decode", "all"):
    w = dict(bench.WORKLOAD)
    q, k, v, bid, lens = bench.make_inputs(dev, w)
    tm = hpc.get_attention_decode_task_workspace(w["batch"], w["seq_kv"], w["num_head_kv"], 64)
    hpc.assign_attention_decode_task(lens, tm, w["num_head_kv"], 1, True, 64)
    o = torch.empty_like(q)
    for _ in range(20):
        hpc.attention_decode_bf16(q, k, v, bid, lens, 0, True, True, tm, None, o)
    torch.cuda.synchronize()
if what in ("fp8", "all"):
    print(bench.extra_decode(dev, hpc))
if what in ("moe", "all"):
    print(bench.extra_moe(dev, hpc, tokens=(64, 4096)))
if what in ("next", "all"):  # the widening rows: rope + KV store, router GEMM, sampler, fp8 prefill
    print(bench.extra_rope(dev, hpc))
    print(bench.extra_router_gemm(dev, hpc))
    print(bench.extra_sampler(dev, hpc))
    print(bench.extra_prefill(dev, hpc))
torch.cuda.synchronize()
