"""Development tool (GPU box): the ride-along rows of the 256 x 256 grouped GEMM (csrc/group_gemm_p8.hip, p8_body<kExt>)
against the dispatch of round 5 (development key 49 = 1: a tail item for every tail), bit for bit, over many shapes and
REPEATED calls - the build of this body without -fno-slp-vectorize produced wrong values in one output row of one block on
every call of the fused op (profiles/round6_moe_ext_ab.txt) while single-call tests of the grouped GEMM alone passed.
usage: python tools/stress_ext.py [repeats]"""
import os, sys
os.environ.setdefault("HPC_AMD_DEV", "1")
from pathlib import Path
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT / "hpc-ops_amd")); sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / "tests"))
import torch
import hpc
from hpc import _C
from test_fuse_moe_blockwise import _inputs
REP = int(sys.argv[1]) if len(sys.argv) > 1 else 20
F8 = torch.float8_e4m3fn
def dev_set(k, v): _C.lib.hpc_dev_tuning_set(k, v)
bad = calls = 0
dev_set(3, 4)
# the fused op: (tokens, experts, top-k, hidden, inter) - row counts per expert that end in ride-along tails
for (T, E, k, H, I) in [(257, 2, 2, 1024, 384), (530, 4, 2, 512, 256), (1050, 4, 2, 512, 256), (1100, 8, 2, 1024, 384),
                        (2100, 8, 2, 2048, 512), (4096, 64, 8, 1024, 256), (1040, 4, 2, 4096, 1408)]:
    args = _inputs(T, k, H, I, E, 1, False, seed=T)
    dev = [t.cuda() if t is not None else None for t in args]
    counts = torch.bincount(args[6].flatten().long(), minlength=E)
    rides = int((((counts % 256) > 0) & ((counts % 256) <= 16 * (counts // 256)) & (((counts + 127) // 128) % 2 == 1)).sum())
    run = lambda: hpc.fuse_moe_blockwise_fp8(*dev[:8], 0, E)
    dev_set(49, 1); ref = run(); dev_set(49, 0)
    nb = 0
    for _ in range(REP):
        nb += 0 if torch.equal(run(), ref) else 1
    calls += REP; bad += nb
    print(f"fused T={T} E={E} k={k} H={H} I={I}: {rides} experts with ride-along rows, {nb} of {REP} calls differ from the round-5 dispatch", flush=True)
# the per-tensor fused op (the kernels accumulate in the matrix pipe; the ride-along block too)
for (T, E, k, H, I, bf) in [(530, 4, 2, 512, 256, False), (1050, 4, 2, 1024, 384, True), (2100, 8, 2, 512, 256, True), (4200, 16, 2, 2048, 768, False)]:
    torch.manual_seed(T)
    ids = torch.sort(torch.multinomial(torch.ones(T, E), k, replacement=False).to(torch.int32), dim=1)[0].cuda()
    x = (torch.randn((T, H), device="cuda") / 100).to(F8)
    guw = torch.randn((E, I * 2, H), device="cuda").to(F8)
    dw = torch.randn((E, H, I), device="cuda").to(F8)
    gus, ds, ams = torch.rand(E, device="cuda") + 0.5, torch.rand(E, device="cuda") + 0.5, torch.rand(1, device="cuda") + 0.5
    sc = torch.rand((T, k), device="cuda") / k
    run = lambda: hpc.fuse_moe_pertensor_fp8(x, guw, dw, gus, ds, ams, ids, sc, 0, E, use_bf16_mul=bf)
    dev_set(49, 1); ref = run(); dev_set(49, 0)
    nb = sum(0 if torch.equal(run(), ref) else 1 for _ in range(REP))
    calls += REP; bad += nb
    print(f"per-tensor fused T={T} E={E} H={H} I={I} bf16_mul={bf}: {nb} of {REP} calls differ from the round-5 dispatch", flush=True)
# the standalone grouped GEMM, plain epilogue
for (n, kk) in [(512, 512), (768, 1408), (256, 2048), (4096, 1024)]:
    torch.manual_seed(n + kk)
    seqlens = torch.tensor([257, 272, 273, 513, 528, 544, 545, 769, 800, 816, 817, 300, 256, 20, 0, 1030], dtype=torch.int32)
    G, total = len(seqlens), int(seqlens.sum())
    x = (torch.randn((total, kk), device="cuda") / 10).to(F8)
    w = (torch.randn((G, n, kk), device="cuda") / 10).to(F8)
    kb = kk // 128
    cu = torch.cat([torch.zeros(1, dtype=torch.int32), torch.cumsum(seqlens, 0).to(torch.int32)])
    avg = total // G
    tile_m = hpc.aligned_size(avg)
    tiles = (seqlens + tile_m - 1) // tile_m
    xs_t = torch.randn((kb, int(tiles.sum()) * tile_m + 64), device="cuda")
    wscale = torch.randn((G, n // 128, (kb + 3) // 4 * 4), device="cuda")
    run = lambda: hpc.group_gemm_blockwise_fp8(x, w, seqlens.cuda(), cu.cuda(), xs_t, wscale, num_seq_per_group_avg=avg)
    dev_set(49, 1); ref = run(); dev_set(49, 0)
    nb = sum(0 if torch.equal(run(), ref) else 1 for _ in range(REP))
    calls += REP; bad += nb
    print(f"grouped GEMM n={n} k={kk}: {nb} of {REP} calls differ from the round-5 dispatch", flush=True)
dev_set(3, 0)
print(f"TOTAL: {bad} of {calls} calls differ")
sys.exit(1 if bad else 0)
