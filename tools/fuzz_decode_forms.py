"""Development tool: randomised cross-check of the fp8 decode forms - one kv head per workgroup (product for 17-32 q rows per kv head;
development key 60 = 2 also below that) against the first generation (key 60 = 1), per-tensor and per-token K scales, NHD / HND
pages of 32 / 64 tokens, both new_kv_included settings, random batches with empty, short and long requests.  The two kernels
split requests at different points, so outputs agree to fp rounding of the merges, not bit for bit: max |dy| <= 0.03 at |y| ~ 1.
usage: python tools/fuzz_decode_forms.py [cases=40] [seed=0] [mode=pair]
mode=pair: the head-pair form itself (product: even head counts, <= 16 q rows, NHD pages - incl. underloaded launches whose short
requests are kept whole, round 6) against the first generation (development key 12 = 1)."""
import os
os.environ.setdefault("HPC_AMD_DEV", "1")
import math, random, sys
from pathlib import Path
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT / "hpc-ops_amd")); sys.path.insert(0, str(ROOT))
import torch, hpc
from hpc import _C
dev = torch.device("cuda", 0)
kw = dict(a.split("=") for a in sys.argv[1:])
n_cases, seed = int(kw.get("cases", 40)), int(kw.get("seed", 0))
PAIR = kw.get("mode") == "pair"
rnd = random.Random(seed)
f8 = torch.float8_e4m3fn
worst = 0.0
for case in range(n_cases):
    hkv, g = rnd.choice([(8, 8), (4, 8), (1, 8), (2, 8), (16, 8), (3, 8), (2, 4), (6, 4)])
    hq = hkv * g
    sq = rnd.choice([3, 4]) if g == 8 else 4
    low = rnd.random() < 0.3          # <= 16 q rows: reachable through key 60 = 2 only
    if low: sq = rnd.choice([1, 2]) if g == 8 else rnd.choice([1, 2, 4])
    P = rnd.choice([32, 64])
    hnd = rnd.random() < 0.5
    if PAIR:
        hkv, g = rnd.choice([(8, 8), (4, 8), (2, 8), (16, 8), (2, 4), (6, 4)])
        hq = hkv * g
        sq = rnd.choice([1, 2]) if g == 8 else rnd.choice([1, 2, 4])
        P, hnd = rnd.choice([16, 32, 64]), False
    ktok = rnd.random() < 0.4 and P >= 32
    nkv = rnd.random() < 0.7
    B = rnd.choice([1, 2, 5, 17, 64, 150])
    kinds = [rnd.choice(["zero", "tiny", "short", "mid", "long"] if not (PAIR and rnd.random() < 0.5) else ["zero", "tiny", "short", "short"]) for _ in range(B)]
    if PAIR and rnd.random() < 0.5: kinds = [rnd.choice(["tiny", "short"]) for _ in range(B)]   # underloaded launches
    lens = torch.tensor([{"zero": 0, "tiny": rnd.randint(1, 70), "short": rnd.randint(60, 700), "mid": rnd.randint(700, 5000),
                          "long": rnd.randint(5000, 40000 if B <= 17 else 12000)}[k] for k in kinds], dtype=torch.int32)
    torch.manual_seed(seed * 1000 + case)
    D = 128
    nb = (lens + sq + P - 1) // P
    total = int(nb.sum()); pool = int(total * 1.2) + 4
    q = torch.randn(B * sq, hq, D, dtype=torch.bfloat16, device=dev) / math.sqrt(D)
    q_scale = q.float().abs().max(-1)[0] / 10
    q8 = (q / q_scale[:, :, None]).to(f8)
    perm = torch.randperm(pool, device=dev)[:total].to(torch.int32)
    bid = torch.zeros(B, max(int(nb.max()), 1), dtype=torch.int32, device=dev)
    off = 0
    for i, n in enumerate(nb.tolist()):
        bid[i, :n] = perm[off:off + n]; off += n
    if ktok:
        kf = torch.randn(pool, P, hkv, D, dtype=torch.bfloat16, device=dev)
        ksc = kf.float().abs().max(-1)[0] / 448
        k8 = torch.empty(pool, P + P // 32, hkv, D, dtype=f8, device=dev)
        k8[:, :P] = (kf / ksc[:, :, :, None]).to(f8)
        k8[:, P:] = ksc.permute(0, 2, 1).contiguous().view(f8).reshape(pool, hkv, -1, D).permute(0, 2, 1, 3)
        v8 = torch.randn(pool, P, hkv, D, dtype=torch.bfloat16, device=dev).to(f8)
        vs = torch.rand(hkv, device=dev) * 0.1 + 0.01
        qt = hpc.QuantType.QPERTOKEN_PERHEAD_KPERTOKEN_PERHEAD_VPERHEAD
    else:
        k8 = (torch.randn(pool, P, hkv, D, dtype=torch.bfloat16, device=dev) / math.sqrt(D)).to(f8)
        v8 = torch.randn(pool, P, hkv, D, dtype=torch.bfloat16, device=dev).to(f8)
        vs = torch.rand(1, device=dev) + 0.5
        qt = hpc.QuantType.QPERTOKEN_PERHEAD_KPERTENSOR_VPERTENSOR
    kd, vd = k8, v8
    if hnd:
        kd = k8.view(torch.uint8).permute(0, 2, 1, 3).contiguous().permute(0, 2, 1, 3).view(f8)
        vd = v8.view(torch.uint8).permute(0, 2, 1, 3).contiguous().permute(0, 2, 1, 3).view(f8)
    ks = kd[:, P:] if ktok else torch.rand(1, device=dev) + 0.5
    lens_in = (lens + (sq if nkv else 0)).to(dev)
    tm = hpc.get_attention_decode_task_workspace(B, int(lens.max()) + sq, hkv, 64)
    hpc.assign_attention_decode_task(lens_in, tm, hkv, sq, nkv, 64)
    outs = {}
    for key in (1, 2, 0):
        _C.lib.hpc_dev_tuning_set(12 if PAIR else 60, (1 if key == 1 else 0) if PAIR else key)
        for _ in range(2):  # twice: counters left zero
            outs[key] = hpc.attention_decode_fp8(q8, kd[:, :P], vd, bid, lens_in, q_scale, ks, vs, sq - 1, nkv, qt, True, tm).float()
        torch.cuda.synchronize()
    _C.lib.hpc_dev_tuning_set(60, 0); _C.lib.hpc_dev_tuning_set(12, 0)
    live = (lens + (0 if nkv else 0) > -1)  # every request has >= sq new tokens when nkv; rows of empty requests compare too
    d2 = float((outs[2] - outs[1]).abs().max()); d0 = float((outs[0] - outs[1]).abs().max())
    fin = bool(torch.isfinite(outs[2]).all() and torch.isfinite(outs[0]).all())
    worst = max(worst, d2, d0)
    print(f"case {case:3d}: heads {hkv}/{hq} sq {sq} P {P} {'HND' if hnd else 'NHD'} {'qt0' if ktok else 'qt1'} nkv {int(nkv)} B {B:3d} "
          f"tokens {int(lens.sum()):7d}  max|one-head - first gen| {d2:.4f}  max|product - first gen| {d0:.4f}  finite {fin}", flush=True)
    assert fin and d2 <= 0.03 and d0 <= 0.03, "forms disagree"
    del k8, v8, kd, vd
print(f"{n_cases} cases, worst difference {worst:.4f}")
