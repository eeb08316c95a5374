"""Development tool: bf16 decode at num_seq_q 1 ... 4 (C3 length mix and uniform 8k, 8 / 64 heads, NHD pages): num_seq_q 1 / 2 run the
head-pair form, 3 / 4 (24 / 32 q rows per kv head) the first generation's two-block form - round 6: 0.77 / 0.75 against 0.67 / 0.66
on the mix, 0.795 / 0.80 against 0.755 on uniform 8k (gpurun_out/r6_bf16_sq.log).
usage: python tools/tune_bf16_sq.py"""
import os, sys, math
sys.path.insert(0, "hpc-ops_amd"); sys.path.insert(0, ".")
import torch, bench, hpc
dev = torch.device("cuda", 0)
B, Hkv = 64, 8
for name, lens in (("c3_mix", bench.c3_lens()), ("uniform8k", torch.full((B,), 8192, dtype=torch.int32))):
    for sq in (1, 2, 3, 4):
        wc = dict(bench.C2, num_seq_q=sq)
        inp = bench.c2_inputs(dev, lens, wc)
        tm = hpc.get_attention_decode_task_workspace(B, int(lens.max()), Hkv, 64)
        hpc.assign_attention_decode_task(inp["kv_lens"], tm, Hkv, sq, True, 64)
        o = torch.empty_like(inp["q"])
        us = bench.timed(lambda: hpc.attention_decode_bf16(inp["q"], inp["k_cache"], inp["v_cache"], inp["block_ids"], inp["kv_lens"], sq - 1, True, True, tm, None, o), graph=True, reps=10)
        kvb = int(lens.sum()) * Hkv * 512
        print(f"bf16 {name} sq{sq}: {us:7.1f} us {kvb/us/1e3:7.1f} GB/s {kvb/us/1e3/8000:.3f}", flush=True)
        del inp, o
