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
for cfg in ([] if os.environ.get("QT0_SQ") else (sys.argv[1:] or ["54=1", "0=0", "54=1", "0=0"])):
    pairs = [tuple(int(x) for x in kv.split("=")) for kv in cfg.split(",") if kv]
    for k, v in pairs: _C.lib.hpc_dev_tuning_set(k, v)
    out = bench.extra_decode_holes(dev, hpc)
    for name in ("decode_fp8_qt0_mixed_nhd", "decode_fp8_qt0_mixed_hnd"):
        r = out[name]
        print(f"[{cfg:>8}] {name}: {r['us']:7.1f} us  {r['GBps']:7.1f} GB/s  {r['hbm_frac_of_8TBps']:.3f}  parity max_abs_err {r['parity']['max_abs_err']}", flush=True)
    for k, v in pairs: _C.lib.hpc_dev_tuning_set(k, 0)


def time_sq(sq, hnd=False):
    """quant_type 0 at num_seq_q = sq (timing only; parity: tests/test_attention_decode_fp8.py): the builder of
    bench.extra_decode_holes with sq query rows per request."""
    import math
    B, P, D, Hkv, Hq = 64, 64, 128, 8, 64
    f8 = torch.float8_e4m3fn
    case = os.environ.get("QT0_CASE", "mixed")
    lens = {"mixed": bench.c3_lens(), "uniform8k": torch.full((B,), 8192, dtype=torch.int32),
            "skewed_mix": torch.tensor([128] * 32 + [4096] * 32, dtype=torch.int32),
            "uniform512": torch.full((B,), 512, dtype=torch.int32)}[case]
    torch.manual_seed(41); torch.cuda.manual_seed(41)
    nbl = (lens + P - 1) // P
    total = int(nbl.sum()); pool = int(total * 1.2) + B + 8
    q_bf16 = torch.randn((B * sq, Hq, D), dtype=torch.bfloat16, device=dev) / math.sqrt(D)
    q_scale = q_bf16.float().abs().max(-1)[0] / 10
    q8 = (q_bf16 / q_scale[:, :, None]).to(f8)
    packed = torch.randperm(pool, device=dev)[:total].to(torch.int32)
    block_ids = torch.zeros((B, int(nbl.max())), dtype=torch.int32, device=dev)
    off = 0
    for i, n in enumerate(nbl.tolist()):
        block_ids[i, :n] = packed[off: off + n]; off += n
    kf = torch.randn(pool, P, Hkv, D, dtype=torch.bfloat16, device=dev)
    ksc = kf.float().abs().max(-1)[0] / 448
    k8 = torch.empty(pool, P + 2, Hkv, D, dtype=f8, device=dev)
    k8[:, :P] = (kf / ksc[:, :, :, None]).to(f8)
    k8[:, P:] = ksc.permute(0, 2, 1).contiguous().view(f8).reshape(pool, Hkv, -1, D).permute(0, 2, 1, 3)
    del kf, ksc
    v8 = torch.randn(pool, P, Hkv, D, dtype=torch.bfloat16, device=dev).to(f8)
    v_scale = torch.rand(Hkv, device=dev) * 0.1 + 0.01
    kd, vd = k8, v8
    if hnd:
        kd = k8.view(torch.uint8).permute(0, 2, 1, 3).contiguous().permute(0, 2, 1, 3).view(f8)
        vd = v8.view(torch.uint8).permute(0, 2, 1, 3).contiguous().permute(0, 2, 1, 3).view(f8)
    lens_dev = lens.to(dev)
    tm = hpc.get_attention_decode_task_workspace(B, int(lens.max()), Hkv, 64)
    hpc.assign_attention_decode_task(lens_dev, tm, Hkv, sq, True, 64)
    o = torch.empty(B * sq, Hq, D, dtype=torch.bfloat16, device=dev)
    qt0 = hpc.QuantType.QPERTOKEN_PERHEAD_KPERTOKEN_PERHEAD_VPERHEAD
    call = lambda: hpc.attention_decode_fp8(q8, kd[:, :P], vd, block_ids, lens_dev, q_scale, kd[:, P:], v_scale, sq - 1, True, qt0, True, tm, None, o)  # noqa: E731
    us = bench.timed(call, graph=True, reps=10)
    kvb = int(lens.sum()) * Hkv * 260
    return us, kvb / us / 1e3


if os.environ.get("QT0_SQ"):
    for cfg in (os.environ.get("QT0_CFGS", "60=1|0=0|60=1|0=0").split("|")):
        pairs = [tuple(int(x) for x in kv.split("=")) for kv in cfg.split(",") if kv]
        for k, v in pairs: _C.lib.hpc_dev_tuning_set(k, v)
        for sq in (int(x) for x in os.environ["QT0_SQ"].split(",")):
            for hnd in (False, True):
                us, gb = time_sq(sq, hnd)
                print(f"[{cfg:>8}] qt0 sq{sq} {'HND' if hnd else 'NHD'}: {us:7.1f} us {gb:7.1f} GB/s {gb / 8000:.3f}", flush=True)
        for k, v in pairs: _C.lib.hpc_dev_tuning_set(k, 0)
