"""Development tool: the kernels round 6 added, once each under rocprofv3 (kernel names + durations for profiles/): fp8 / bf16 decode with
num_seq_q 3 (one kv head per workgroup: decode2_kernel<..., kSolo>), the fused MoE at T = 64 (gemm_blockwise_stream2_kernel) and the
reference benchmark's uniform_512 decode case (head-pair kernel, underloaded launch).  Plain launches, 20 calls each.
usage: rocprofv3 --kernel-trace --stats --output-format csv -d OUT -- python tools/prof_new_kernels.py"""
import sys, math
from pathlib import Path
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT / "hpc-ops_amd")); sys.path.insert(0, str(ROOT))
import torch, bench, hpc
dev = torch.device("cuda", 0)
B, Hkv = 64, 8
N = 20
# fp8, num_seq_q 3, NHD and HND pages, C3 mix
for hnd in (False, True):
    wc = dict(bench.C3, num_seq_q=3)
    inp = bench.c3_inputs(dev, wc)
    if hnd:
        for k in ("k_cache", "v_cache"):
            inp[k] = inp[k].view(torch.uint8).permute(0, 2, 1, 3).contiguous().permute(0, 2, 1, 3).view(torch.float8_e4m3fn)
    tm = hpc.get_attention_decode_task_workspace(B, int(inp["kv_lens"].max()), Hkv, 64)
    hpc.assign_attention_decode_task(inp["kv_lens"], tm, Hkv, 3, True, 64)
    o = torch.empty(B * 3, 64, 128, dtype=torch.bfloat16, device=dev)
    for _ in range(N):
        hpc.attention_decode_fp8(inp["q"], inp["k_cache"], inp["v_cache"], inp["block_ids"], inp["kv_lens"], inp["q_scale"], inp["k_scale"],
                                 inp["v_scale"], 2, True, hpc.QuantType.QPERTOKEN_PERHEAD_KPERTENSOR_VPERTENSOR, True, tm, None, o)
    torch.cuda.synchronize()
    del inp
# bf16, num_seq_q 3, uniform 8k
lens = torch.full((B,), 8192, dtype=torch.int32)
wc = dict(bench.C2, num_seq_q=3)
inp = bench.c2_inputs(dev, lens, wc)
tm = hpc.get_attention_decode_task_workspace(B, 8192, Hkv, 64)
hpc.assign_attention_decode_task(inp["kv_lens"], tm, Hkv, 3, True, 64)
o = torch.empty_like(inp["q"])
for _ in range(N):
    hpc.attention_decode_bf16(inp["q"], inp["k_cache"], inp["v_cache"], inp["block_ids"], inp["kv_lens"], 2, True, True, tm, None, o)
torch.cuda.synchronize()
del inp
# fp8 uniform_512 (underloaded launch)
wc = dict(bench.C3)
inp = bench.c3_inputs(dev, wc, lens=torch.full((B,), 512, dtype=torch.int32))
tm = hpc.get_attention_decode_task_workspace(B, 512, Hkv, 64)
hpc.assign_attention_decode_task(inp["kv_lens"], tm, Hkv, 1, True, 64)
o = torch.empty(B, 64, 128, dtype=torch.bfloat16, device=dev)
for _ in range(N):
    hpc.attention_decode_fp8(inp["q"], inp["k_cache"], inp["v_cache"], inp["block_ids"], inp["kv_lens"], inp["q_scale"], inp["k_scale"],
                             inp["v_scale"], 0, True, hpc.QuantType.QPERTOKEN_PERHEAD_KPERTENSOR_VPERTENSOR, True, tm, None, o)
torch.cuda.synchronize()
del inp
# fused MoE, T = 64
m = bench.c4_inputs(dev, bench.C4, tokens=64)
for _ in range(N):
    hpc.fuse_moe_blockwise_fp8(m["x"], m["x_scale"], m["guw"], m["guws"], m["dw"], m["dws"], m["ids"], m["scale"], 0, bench.C4["num_expert"])
torch.cuda.synchronize()
print("done")
