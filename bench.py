#!/usr/bin/env python
"""bench.py - headline measurement of the decode-step hot path on MI355X.

One "step" = one paged decode-attention call (scheduler excluded, like the reference benchmark:
benchmark/attention_decode/bench_attention_decode_bf16.py).  Workload = BASELINE.json configs[1]:
bf16, batch 64, 8 KV heads (64 q heads, GQA 8), head_dim 128, 8192-token requests in 64-token
pages, inputs resident in HBM.  Contract: python bench.py --gpus N --steps K --warmup W prints ONE
JSON line on rank 0 (for N > 1 launched through torch.distributed.run: one replica per GPU, weak
scaling, barrier + synchronize on both sides of the timed region, max over ranks).
"""
import argparse
import json
import math
import os
import sys
import time
from pathlib import Path

ROOT = Path(__file__).resolve().parent
sys.path.insert(0, str(ROOT / "hpc-ops_amd"))
sys.path.insert(0, str(ROOT))
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

import torch  # noqa: E402

HBM_PEAK_GBPS = 8000.0  # MI355X spec (MI355X_MICROARCH.md); measured copy ceiling ~6290 GB/s
HBM_COPY_GBPS = 6290.0

WORKLOAD = dict(batch=64, num_head_kv=8, num_head_q=64, head_dim=128, block_size=64, seq_kv=8192,
                num_seq_q=1)


def make_inputs(dev, w):
    """reference generator: benchmark/attention_decode/bench_attention_decode_bf16.py:125-154"""
    torch.manual_seed(41)
    B, P, D = w["batch"], w["block_size"], w["head_dim"]
    kv_lens = torch.full((B,), w["seq_kv"], dtype=torch.int32, device=dev)
    nblocks = (kv_lens + P - 1) // P
    total = int(nblocks.sum())
    max_num_blocks = int(total * 1.2) + B + 8
    q = torch.randn((B * w["num_seq_q"], w["num_head_q"], D), dtype=torch.bfloat16, device=dev) / math.sqrt(D)
    k_cache = torch.randn(max_num_blocks, P, w["num_head_kv"], D, dtype=torch.bfloat16, device=dev) / math.sqrt(D)
    v_cache = torch.randn(max_num_blocks, P, w["num_head_kv"], D, dtype=torch.bfloat16, device=dev)
    packed = torch.randperm(max_num_blocks, device=dev)[:total].to(torch.int32)
    block_ids = torch.zeros(B, int(nblocks.max()), dtype=torch.int32, device=dev)
    off = 0
    for i, nb in enumerate(nblocks.tolist()):
        block_ids[i, :nb] = packed[off : off + nb]
        off += nb
    return q, k_cache, v_cache, block_ids, kv_lens


def algorithmic_bytes(w):
    # sum_b S_b * Hkv * (128 + 128) * 2 B of KV, plus q and y
    kv = w["batch"] * w["seq_kv"] * w["num_head_kv"] * 2 * w["head_dim"] * 2
    qo = 2 * w["batch"] * w["num_seq_q"] * w["num_head_q"] * w["head_dim"] * 2
    return kv + qo


def cpu_baseline(q, k_cache, v_cache, block_ids, kv_lens, w, sample_requests=8):
    """The reference's PyTorch-eager oracle (oracle/attention.py) timed on the host cores over a
    bounded sample of the same workload (first `sample_requests` requests)."""
    from oracle import attention as oattn

    cores = os.cpu_count() or 1
    torch.set_num_threads(cores)
    rows = list(range(min(sample_requests, w["batch"])))
    qc, kc, vc = q.cpu(), k_cache.cpu(), v_cache.cpu()
    bc, lc = block_ids.cpu(), kv_lens.cpu()
    oattn.ref_attn_paged_separate(qc, kc, vc, bc, lc, w["num_seq_q"], rows[:1])  # warm-up
    t0 = time.perf_counter()
    ref = oattn.ref_attn_paged_separate(qc, kc, vc, bc, lc, w["num_seq_q"], rows)
    dt = time.perf_counter() - t0
    per_req = algorithmic_bytes(w) / w["batch"]
    return ref, rows, {
        "value": round(per_req * len(rows) / dt / 1e9, 3), "unit": "GB/s", "cores": cores,
        "kind": "port",
        "sample": f"{len(rows)} of {w['batch']} requests of the same workload, PyTorch-eager "
                  f"oracle (tests/test_attention_decode_bf16.py:15-59 restated), {dt:.2f} s",
    }


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-graph", action="store_true")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    dist_on = world > 1
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if dist_on:
        import torch.distributed as dist

        dist.init_process_group(backend="nccl", device_id=dev)

    import hpc

    w = dict(WORKLOAD)
    q, k_cache, v_cache, block_ids, kv_lens = make_inputs(dev, w)
    task_map = hpc.get_attention_decode_task_workspace(w["batch"], w["seq_kv"], w["num_head_kv"], 64)
    hpc.assign_attention_decode_task(kv_lens, task_map, w["num_head_kv"], w["num_seq_q"], True, 64)
    out = torch.empty_like(q)

    def step():
        hpc.attention_decode_bf16(q, k_cache, v_cache, block_ids, kv_lens, mtp=w["num_seq_q"] - 1,
                                  new_kv_included=True, splitk=True, task_map=task_map, output=out)

    # ---- parity of exactly what is timed (sample of requests vs the oracle) + CPU baseline -----
    step()
    torch.cuda.synchronize()
    cpu = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        ref, rows, cpu = cpu_baseline(q, k_cache, v_cache, block_ids, kv_lens, w)
        got = out.reshape(w["batch"], w["num_seq_q"], w["num_head_q"], w["head_dim"])[rows].cpu()
        err = (got.float() - ref.float()).abs().max().item()
        assert err <= 0.016, f"bench output does not match the oracle: max abs err {err}"

    # ---- graph capture of one step (reference method: graph replay + events) -------------------
    graph = None
    if not args.no_graph:
        try:
            s = torch.cuda.Stream()
            s.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(s):
                step()
            torch.cuda.current_stream().wait_stream(s)
            graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(graph):
                step()
        except Exception as e:  # noqa: BLE001
            print(f"[bench] graph capture failed ({e}); timing eager launches", file=sys.stderr)
            graph = None
    run = graph.replay if graph is not None else step

    for _ in range(args.warmup):
        run()
    torch.cuda.synchronize()
    if dist_on:
        dist.barrier()
    torch.cuda.synchronize()
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(args.steps + 1)]
    t0 = time.perf_counter()
    ev[0].record()
    for i in range(args.steps):
        run()
        ev[i + 1].record()
    torch.cuda.synchronize()
    if dist_on:
        dist.barrier()
    torch.cuda.synchronize()
    wall = time.perf_counter() - t0
    per_step_ms = sorted(ev[i].elapsed_time(ev[i + 1]) for i in range(args.steps))
    kern_ms_avg = sum(per_step_ms) / len(per_step_ms)

    if dist_on:
        t = torch.tensor([wall], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        wall = float(t.item())

    if rank == 0:
        nbytes = algorithmic_bytes(w)
        ms_per_step = wall / args.steps * 1e3
        value = nbytes * world / (wall / args.steps) / 1e9
        achieved = nbytes / (kern_ms_avg * 1e-3) / 1e9
        traffic = None
        pmc = ROOT / "profiles" / "decode_bf16_pmc.json"
        if pmc.exists():
            traffic = json.loads(pmc.read_text()).get("hbm_bytes_per_launch")
        line = {
            "metric": "decode_attention_bf16_kv_throughput", "value": round(value, 1), "unit": "GB/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(ms_per_step, 5), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "bf16", "data": "synthetic",
            "config": {
                "workload": "decode attention bf16, batch 64, 8 KV heads / 64 Q heads, head_dim 128, "
                            "seqlen 8192 uniform, paged KV (64-token pages), dynamic tile scheduler "
                            "(BASELINE.json configs[1])",
                "parallelism": f"replicas x{world}", "launch": "hipGraph replay" if graph else "eager",
                "scheduler_in_timed_region": False,
            },
            "us_per_call": round(kern_ms_avg * 1e3, 2),
            "us_per_call_median": round(per_step_ms[len(per_step_ms) // 2] * 1e3, 2),
            "roofline": {
                "bound": "hbm", "achieved": round(achieved, 1), "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                "frac": round(achieved / HBM_PEAK_GBPS, 4),
                "frac_of_measured_copy_peak": round(achieved / HBM_COPY_GBPS, 4),
                "traffic": traffic, "algorithmic_bytes_per_launch": nbytes,
                "kernel": "hpc::decode::decode_bf16_kernel<1> (+ combine), HIP events per launch",
            },
            "cpu_baseline": cpu,
        }
        print(json.dumps(line))
    if dist_on:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
