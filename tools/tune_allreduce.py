"""Development tool: the fused all-reduce + RMSNorm (both modes) at world size 1 (what one test GPU can time: launch cost,
slot cleaning, the local scatter / poll round trip and the norm), graph replay, one call per replay.
usage: python tools/tune_allreduce.py ["k=v,k=v" ...]   each argument is one configuration of tuning registers
(key 35 = 1: two launches instead of the fused one; key 11 = n: high-throughput grid floor n instead of two workgroups per CU)
--ht: the high-throughput mode only (T = 512 / 4096 / 16384), 10 calls per replay = kernel time without the replay floor"""
import math
import os
os.environ.setdefault("HPC_AMD_DEV", "1")
import sys
from pathlib import Path
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT / "hpc-ops_amd")); sys.path.insert(0, str(ROOT))
import torch, bench, hpc
from hpc import _C
dev = torch.device("cuda", 0)
torch.cuda.set_device(dev)
comm = hpc.MulticastCommunicator(0, 1, 0, f"tune_ar_{os.getpid()}")
H = 8192
w = torch.randn(H, dtype=torch.bfloat16, device=dev)
HT = "--ht" in sys.argv
if HT: sys.argv.remove("--ht")
for cfg in (sys.argv[1:] or ["35=0", "35=1"]):
    pairs = [tuple(int(x) for x in kv.split("=")) for kv in cfg.split(",") if kv]
    for k, v in pairs: _C.lib.hpc_dev_tuning_set(k, v)
    for T in ((512, 4096, 16384) if HT else ()):
        x = torch.randn(T, H, dtype=torch.bfloat16, device=dev)
        res = torch.randn(T, H, dtype=torch.bfloat16, device=dev)
        in_x, in_hdl = hpc.empty_multimem(comm, [T, H], dtype=torch.bfloat16, device=dev)
        out_x, out_hdl = hpc.empty_multimem(comm, [T, H], dtype=torch.bfloat16, device=dev)
        in_x.copy_(x)
        out_res = torch.empty_like(res)
        mi = in_hdl.get_multimem_buff(in_x.shape, dtype=in_x.dtype)
        mo = out_hdl.get_multimem_buff(out_x.shape, dtype=out_x.dtype)
        def call():
            hpc.fuse_allreduce_rmsnorm_high_throughput(in_x, mi, res, w, 1e-6, in_hdl.signal_buffer_ptrs_dev, 0, 1, 64, out_x, mo, out_res)
        call(); torch.cuda.synchronize()
        us1 = bench.timed(call, iters=30, warm=5, graph=True)
        us10 = bench.timed(call, iters=10, warm=2, graph=True, reps=10)
        r = (x.float() + res.float()).bfloat16()
        ref = (r.float() * torch.rsqrt(r.float().pow(2).mean(-1, keepdim=True) + 1e-6) * w.float()).bfloat16()
        print(f"[{cfg:>8}] HT T={T:5d}: {us1:7.1f} us/call (1 per replay)  {us10:7.1f} us/call (10 per replay) = {4 * T * H * 2 / us10 / 1e3:7.1f} GB/s  "
              f"max|res err| {float((out_res.float() - r.float()).abs().max()):.3g}  max|y err| {float((out_x.float() - ref.float()).abs().max()):.3g}  "
              f"timeouts {_C.lib.hpc_allreduce_timeouts()}", flush=True)
    for T in (() if HT else (8, 32, 128, 512)):
        x = torch.randn(T, H, dtype=torch.bfloat16, device=dev)
        res = torch.randn(T, H, dtype=torch.bfloat16, device=dev)
        M = 2 * T * 3
        ws_buf, hdl = hpc.empty_multimem(comm, [M, H], dtype=torch.bfloat16, device=dev)
        ws_buf.view(torch.int32).fill_(-(2 ** 31))
        mc = hdl.get_multimem_buff([M, H], dtype=torch.bfloat16)
        flags = torch.tensor([0, 2, (M * H * 2 // 3) // 16 * 16, 0, 0, 0, 0, 0, 0], dtype=torch.int32, device=dev)
        out, out_res = torch.empty_like(x), torch.empty_like(x)
        def call():
            hpc.fuse_allreduce_rmsnorm_low_latency(x, mc, hdl.data_buffer_ptrs_dev, ws_buf, flags, 1, 0, res, w, 1e-6, 16, out, out_res, True)
        us1 = bench.timed(call, iters=30, warm=5, graph=True)
        us10 = bench.timed(call, iters=10, warm=2, graph=True, reps=10) if "reps" in bench.timed.__code__.co_varnames else float("nan")
        ref = (x.float() + res.float())
        print(f"[{cfg:>8}] LL T={T:4d}: {us1:7.1f} us/call (1 per replay)  {us10:7.1f} us/call (10 per replay)  "
              f"max|res err| {float((out_res.float() - ref).abs().max()):.3g}  timeouts {_C.lib.hpc_allreduce_timeouts()}", flush=True)
    for k, v in pairs: _C.lib.hpc_dev_tuning_set(k, 0)
