"""Parity at the shapes the BASELINE metric is quoted on (BASELINE.json configs[2] and configs[3]).

C3 - FP8 decode attention: batch 64, 8 KV / 64 Q heads, request lengths log-uniform in [128, 32768]
     (seed 41, the bench.py headline workload), NHD and HND-backed pages of 64 tokens, both quant types,
     dynamic scheduler with the reference benchmark's min_process_len (64); every request is compared with
     the pinned CPU oracle at the reference tolerances (0.2 per-tensor K/V, 0.1 per-token K).
C4 - fused MoE FP8 blockwise: 64 experts top-8, hidden 4096, ffn 11008, T in {16, 256, 4096}, against
     oracle/fuse_moe.py with the expert weights streamed one expert at a time (all rows for T <= 256, a fixed
     row sample for T = 4096), plus group_gemm_blockwise_fp8 at K=11008/N=4096 and K=4096/N=22016 on 64 ragged
     groups for every tiled-kernel mode.  Generators: reference tests/test_fuse_moe_blockwise.py:285-319,
     benchmark/attention_decode/bench_attention_decode_fp8.py:44-67,152-158; tolerance rtol = atol = 0.01.
"""
import math

import pytest
import torch

from utils import allclose, moe_allclose, dev_set

F8 = torch.float8_e4m3fn


# ======================================================================================= C3
def _c3_lens():
    g = torch.Generator().manual_seed(41)
    lo, hi = math.log(128), math.log(32768)
    return torch.exp(torch.rand(64, generator=g) * (hi - lo) + lo).to(torch.int32)


@pytest.mark.gpu
@pytest.mark.parametrize("kvcache_shape", ["NHD", "HND"])
@pytest.mark.parametrize("k_per_token", [False, True])
def test_c3_fp8_decode_graded_shape(kvcache_shape, k_per_token):
    import hpc
    from oracle import attention as oattn
    from test_attention_decode_fp8 import _case

    B, Sq, P, heads = 64, 1, 64, (8, 64)
    lens_total = _c3_lens()
    assert int(lens_total.max()) > 24000 and int(lens_total.min()) < 200  # the mix really spans 128..32k
    lens_before = lens_total - Sq
    q8, q_scale, kv, block_ids, nblocks = _case(B, Sq, lens_before, P, heads, k_per_token)
    if k_per_token:
        kc, _ = oattn.quant_paged_cache_pertoken(kv[:, 0], P)
        vc, v_scale = oattn.quant_paged_cache_perhead(kv[:, 1], P)
        kv8 = torch.empty_like(kv, dtype=F8)
        kv8[:, 0] = kc
        kv8[:, 1] = vc
        k_scale = kv8[:, 0, P:]
        qt, atol = hpc.QuantType.QPERTOKEN_PERHEAD_KPERTOKEN_PERHEAD_VPERHEAD, 0.1
    else:
        kv8 = kv.to(F8)
        k_scale = torch.rand(1, dtype=torch.float32).clamp_min(1e-6)
        v_scale = torch.rand(1, dtype=torch.float32).clamp_min(1e-6)
        qt, atol = hpc.QuantType.QPERTOKEN_PERHEAD_KPERTENSOR_VPERTENSOR, 0.2
    del kv
    torch.set_num_threads(min(torch.get_num_threads(), 32))
    gt = oattn.ref_attn_fp8(q8, kv8[:, :, :P], block_ids, nblocks, Sq, lens_before, q_scale, k_scale, v_scale,
                            k_per_token)
    kv_dev = kv8.cuda()
    if kvcache_shape == "HND":
        kv_dev = kv_dev.view(torch.uint8).permute(0, 1, 3, 2, 4).contiguous().permute(0, 1, 3, 2, 4).view(F8)
    kcache, vcache = kv_dev[:, 0, :P], kv_dev[:, 1, :P]
    ks_dev = kv_dev[:, 0, P:] if k_per_token else k_scale.cuda()
    lens_in = lens_total.cuda()
    task_map = hpc.get_attention_decode_task_workspace(B, int(lens_total.max()), heads[0], min_process_len=64)
    hpc.assign_attention_decode_task(lens_in, task_map, heads[0], Sq, True, min_process_len=64)
    my = hpc.attention_decode_fp8(q8.cuda(), kcache, vcache, block_ids.cuda(), lens_in, q_scale.cuda(), ks_dev,
                                  v_scale.cuda(), mtp=0, new_kv_included=True, quant_type=qt, splitk=True,
                                  task_map=task_map)
    torch.cuda.synchronize()
    assert allclose(gt, my.cpu(), atol=atol)
    # the static entry (no task map: scheduled on the fly) must give the same answer on the same inputs
    my2 = hpc.attention_decode_fp8(q8.cuda(), kcache, vcache, block_ids.cuda(), lens_in, q_scale.cuda(), ks_dev,
                                   v_scale.cuda(), mtp=0, new_kv_included=True, quant_type=qt, splitk=True)
    torch.cuda.synchronize()
    assert allclose(gt, my2.cpu(), atol=atol)


@pytest.mark.gpu
def test_c3_bench_generator_matches_oracle():
    """the generator and the sampled parity check bench.py itself runs before timing, over ALL requests"""
    import bench
    import hpc
    from oracle import attention as oattn

    dev = torch.device("cuda")
    w = dict(bench.C3)
    inp = bench.c3_inputs(dev, w)
    tm = hpc.get_attention_decode_task_workspace(w["batch"], int(inp["kv_lens"].max()), w["num_head_kv"], 64)
    hpc.assign_attention_decode_task(inp["kv_lens"], tm, w["num_head_kv"], 1, True, 64)
    y = hpc.attention_decode_fp8(inp["q"], inp["k_cache"], inp["v_cache"], inp["block_ids"], inp["kv_lens"],
                                 inp["q_scale"], inp["k_scale"], inp["v_scale"], 0, True,
                                 hpc.QuantType.QPERTOKEN_PERHEAD_KPERTENSOR_VPERTENSOR, True, tm)
    torch.cuda.synchronize()
    c = {k: v.cpu() for k, v in inp.items()}
    torch.set_num_threads(min(torch.get_num_threads(), 32))
    ref = oattn.ref_attn_fp8_separate(c["q"], c["k_cache"], c["v_cache"], c["block_ids"], c["kv_lens"], 1,
                                      c["q_scale"], c["k_scale"], c["v_scale"])
    assert allclose(ref.reshape(y.shape), y.cpu(), atol=0.2)
    assert bench.c3_bytes(c["kv_lens"], w) > int(c["kv_lens"].sum()) * 8 * 256


# ======================================================================================= C2
@pytest.mark.gpu
@pytest.mark.parametrize("kvcache_shape", ["NHD", "HND"])
@pytest.mark.parametrize("lengths", ["uniform8192", "randint_1_8192_seed41"])
def test_c2_bf16_decode_graded_shape(kvcache_shape, lengths):
    """BASELINE configs[1] itself (missing #3): bf16 decode, batch 64, 8 KV / 64 Q heads, head_dim 128,
    pages of 64 tokens, uniform 8192 tokens and `randint(1, 8192)` lengths (seed 41), inputs from the reference
    benchmark's generator (benchmark/attention_decode/bench_attention_decode_bf16.py:125-154 = bench.c2_inputs), NHD
    pages (head-pair kernel) and HND-backed pages (first-generation kernel), EVERY request against the pinned
    PyTorch-eager oracle at the reference tolerance (tests/test_attention_decode_bf16.py: atol 0.016); dynamic
    schedule and the static entry (no task map) give the same answer."""
    import bench
    import hpc

    dev = torch.device("cuda")
    w = dict(bench.C2)
    B, Hkv = w["batch"], w["num_head_kv"]
    if lengths == "uniform8192":
        lens = torch.full((B,), 8192, dtype=torch.int32)
    else:
        lens = torch.randint(1, 8192, (B,), dtype=torch.int32, generator=torch.Generator().manual_seed(41))
    inp = bench.c2_inputs(dev, lens, w, hnd=kvcache_shape == "HND")
    tm = hpc.get_attention_decode_task_workspace(B, int(lens.max()), Hkv, 64)
    hpc.assign_attention_decode_task(inp["kv_lens"], tm, Hkv, 1, True, 64)
    y = hpc.attention_decode_bf16(inp["q"], inp["k_cache"], inp["v_cache"], inp["block_ids"], inp["kv_lens"], 0, True, True, tm)
    torch.cuda.synchronize()
    assert y.dtype == torch.bfloat16 and tuple(y.shape) == (B, w["num_head_q"], w["head_dim"])
    # every request on NHD pages (the head-pair kernel = the timed one); every fourth on HND-backed pages, whose
    # first-generation kernel the bf16 grid of tests/test_attention_decode_bf16.py already covers (suite wall-clock)
    rows = list(range(B)) if kvcache_shape == "NHD" else list(range(0, B, 4))
    err = bench.c2_parity(inp, y, w, rows)
    assert err <= 0.016, err
    y2 = hpc.attention_decode_bf16(inp["q"], inp["k_cache"], inp["v_cache"], inp["block_ids"], inp["kv_lens"], 0, True, True)
    torch.cuda.synchronize()
    assert float((y2.float() - y.float()).abs().max()) <= 0.016
    assert bench.c2_parity(inp, y2, w, rows=[0, 31, 63]) <= 0.016


def test_c2_parity_helper_detects_a_wrong_output():
    """CPU: bench.c2_parity (the in-run check of bench.py::extra_decode and of the test above) is the pinned oracle on
    the benchmark generator's layout: zero error on the oracle's own output, and it sees a perturbed row."""
    import bench
    from oracle import attention as oattn

    w = dict(bench.C2, batch=3, num_head_kv=2, num_head_q=8)
    lens = torch.tensor([130, 64, 1], dtype=torch.int32)
    inp = bench.c2_inputs(torch.device("cpu"), lens, w)
    ref = oattn.ref_attn_paged_separate(inp["q"], inp["k_cache"], inp["v_cache"], inp["block_ids"], inp["kv_lens"], 1)
    y = ref.reshape(3, 8, 128).clone()
    assert bench.c2_parity(inp, y, w) == 0.0
    y[1, 3, 5] += 0.25
    assert bench.c2_parity(inp, y, w) >= 0.2 and bench.c2_parity(inp, y, w, rows=[0, 2]) == 0.0


def test_fp8_separate_cache_oracle_equals_pinned_oracle():
    """CPU: the separate-cache / row-subset form used at the graded shape is bit-equal to ref_attn_fp8
    (itself bit-equal to the reference's in-file oracle, tests/test_oracle_golden.py)."""
    from oracle import attention as oattn

    torch.manual_seed(0)
    B, Sq, Hkv, Hq, D, P = 3, 2, 2, 8, 128, 64
    lens_before = torch.tensor([100, 3, 700], dtype=torch.int32)
    nblocks = (lens_before + Sq + P - 1) // P
    nblk = int(nblocks.sum()) + 2
    kv = torch.randn(nblk, 2, P, Hkv, D).to(F8)
    q = torch.randn(B * Sq, Hq, D).to(F8)
    perm = torch.randperm(nblk).int()
    bid = torch.zeros(B, int(nblocks.max()), dtype=torch.int32)
    o = 0
    for i, n in enumerate(nblocks.tolist()):
        bid[i, :n] = perm[o: o + n]
        o += n
    qs, ks, vs = torch.rand(B * Sq, Hq) * 0.1, torch.tensor([0.3]), torch.tensor([0.7])
    a = oattn.ref_attn_fp8(q, kv, bid, nblocks, Sq, lens_before, qs, ks, vs, False)
    b = oattn.ref_attn_fp8_separate(q, kv[:, 0], kv[:, 1], bid, lens_before + Sq, Sq, qs, ks, vs)
    assert torch.equal(a.reshape(B, Sq, Hq, D), b)
    c = oattn.ref_attn_fp8_separate(q, kv[:, 0], kv[:, 1], bid, lens_before + Sq, Sq, qs, ks, vs, rows=[2, 0])
    assert torch.equal(c, b[[2, 0]])


# ======================================================================================= C4
E, TOPK, H, I = 64, 8, 4096, 11008


@pytest.fixture(scope="module")
def c4_weights():
    """reference generator (tests/test_fuse_moe_blockwise.py:297-311), drawn expert by expert on the device"""
    torch.manual_seed(41)
    dev = torch.device("cuda")
    guw = torch.empty(E, 2 * I, H, dtype=F8, device=dev)
    dw = torch.empty(E, H, I, dtype=F8, device=dev)
    for e in range(E):
        guw[e] = torch.randn(2 * I, H, device=dev).to(F8)
        dw[e] = torch.randn(H, I, device=dev).to(F8)
    guws = torch.randn(E, 2 * I // 128, (H // 128 + 3) // 4 * 4, device=dev)
    dws = torch.randn(E, H // 128, (I // 128 + 3) // 4 * 4, device=dev)
    yield guw, guws, dw, dws
    del guw, dw
    torch.cuda.empty_cache()


@pytest.mark.gpu
@pytest.mark.parametrize("num_tokens,shared", [(16, False), (256, True), (4096, False)])
def test_c4_fused_moe_graded_shape(c4_weights, num_tokens, shared):
    import hpc
    from oracle import fuse_moe as omoe

    guw, guws, dw, dws = c4_weights
    torch.manual_seed(41 + num_tokens)
    T = num_tokens
    ids = torch.sort(torch.multinomial(torch.ones(T, E), TOPK, replacement=False).to(torch.int32), dim=1)[0]
    sc = torch.rand(T, TOPK)
    sc = sc / sc.sum(1, keepdim=True)
    x = (torch.randn(T, H) / 100).to(F8)
    xs = torch.randn(T, H // 128)
    so = torch.randn(T, H, dtype=torch.bfloat16) if shared else None
    my = hpc.fuse_moe_blockwise_fp8(x.cuda(), xs.cuda(), guw, guws, dw, dws, ids.cuda(), sc.cuda(), 0, E,
                                    so.cuda() if shared else None)
    torch.cuda.synchronize()
    # T = 4096: 256 rows spread over the batch - 2048 routed rows, ~32 on every one of the 64 experts
    rows = list(range(T)) if T <= 256 else sorted(set(range(0, T, 16)) | {1, 777, 3333, T - 1})[:260]
    torch.set_num_threads(min(torch.get_num_threads(), 64))
    fetch = lambda e: (guw[e].cpu(), guws[e].cpu(), dw[e].cpu(), dws[e].cpu())  # noqa: E731
    gt = omoe.fuse_moe_blockwise_fp8_rows(x, xs, fetch, ids, sc, rows, 0, E, so)
    assert moe_allclose(gt, my[rows].cpu())  # reference bar restated for hidden 4096 (see utils.py)
    assert torch.isfinite(my.float()).all()


@pytest.mark.dev
@pytest.mark.gpu
@pytest.mark.parametrize("n,k", [(4096, 11008), (22016, 4096)])
@pytest.mark.parametrize("tiled_mode", [0, 1, 2, 4, 12])  # (3 = the 128 x 128 kernel and 22 = the 64-token ring ran here until
def test_c4_group_gemm_blockwise_graded(c4_weights, n, k, tiled_mode):  # round 4; both stay pinned on the small grids)
    """group_gemm_blockwise_fp8 on the two GEMMs of the graded configuration (down: K = 11008 = 86 k-blocks with
    the pad-4 scale stride 88; gate_up: N = 22016), 64 ragged groups incl. empty / 1 / 129 / 257 / 700 rows, for
    every kernel the launcher can pick (tuning key 3: 0 auto, 1 streaming, 2 tiled 256x128, 3 tiled 128x128;
    +10 = the 32-token narrow form of the 256x128 tile).  A row sample of every group is checked against the
    oracle's per-block GEMM."""
    import hpc
    from oracle import fuse_moe as omoe

    guw, guws, dw, dws = c4_weights
    w, wsc = (dw, dws) if n == 4096 else (guw, guws)
    torch.manual_seed(7)
    base = [0, 1, 129, 257, 700, 64, 128, 31]
    seqlens = torch.tensor((base * 8)[:E], dtype=torch.int32)
    cu = torch.cat([torch.zeros(1, dtype=torch.int32), torch.cumsum(seqlens, 0).to(torch.int32)])
    m = int(seqlens.sum())
    avg = m // E
    x = (torch.randn(m, k) / 10).to(F8)
    tile = hpc.aligned_size(avg)
    tiles = (seqlens + tile - 1) // tile
    cu_tiles = torch.cat([torch.zeros(1, dtype=torch.int32), torch.cumsum(tiles, 0).to(torch.int32)])
    m_pad = int(cu_tiles[-1]) * tile + 64
    xs_rows = torch.rand(m, k // 128) + 0.5                       # [m, K/128] row-major (oracle layout)
    xs_t = torch.zeros(k // 128, m_pad)                           # tile-padded transposed layout of the op
    for g in range(E):
        s, c = int(cu[g]), int(seqlens[g])
        xs_t[:, int(cu_tiles[g]) * tile: int(cu_tiles[g]) * tile + c] = xs_rows[s: s + c].t()
    dev_set(3, tiled_mode % 10)
    dev_set(6, 1 + tiled_mode // 10)  # 1x: 32-token tiles, 2x: 64-token tiles
    try:
        y = hpc.group_gemm_blockwise_fp8(x.cuda(), w, seqlens.cuda(), cu.cuda(), xs_t.cuda(), wsc,
                                         num_seq_per_group_avg=avg)
        torch.cuda.synchronize()
    finally:
        dev_set(3, 0)
        dev_set(6, 0)
    assert tuple(y.shape) == (m, n)
    torch.set_num_threads(min(torch.get_num_threads(), 64))
    yc = y.cpu()
    for g in range(E):
        c = int(seqlens[g])
        if c == 0:
            continue
        pick = sorted({0, c // 2, c - 1})
        rows = torch.tensor([int(cu[g]) + r for r in pick])
        one, zero = torch.tensor([len(pick)], dtype=torch.int32), torch.tensor([0], dtype=torch.int32)
        gt = omoe.group_gemm_blockwise(x[rows], w[g].cpu()[None], one, zero, xs_rows[rows], wsc[g].cpu()[None])
        assert allclose(gt.float(), yc[rows].float(), rtol=0.01, atol=0.01), f"group {g}"


def test_moe_rows_oracle_equals_full_oracle():
    """CPU: the streamed row-subset form used at the graded shape is bit-equal to the pinned full oracle."""
    from oracle import fuse_moe as m

    torch.manual_seed(1)
    T, k, Et, Hh, Ii = 37, 4, 16, 256, 384
    ids = torch.sort(torch.multinomial(torch.ones(T, Et), k).to(torch.int32), 1)[0]
    sc = torch.rand(T, k)
    x, xs = (torch.randn(T, Hh) / 100).to(F8), torch.randn(T, Hh // 128)
    guw, guws = torch.randn(Et // 2, 2 * Ii, Hh).to(F8), torch.randn(Et // 2, 2 * Ii // 128, 4)
    dw, dws = torch.randn(Et // 2, Hh, Ii).to(F8), torch.randn(Et // 2, Hh // 128, 4)
    so = torch.randn(T, Hh).bfloat16()
    for rank in (0, 1):
        full = m.fuse_moe_blockwise_fp8(x, xs, guw, guws, dw, dws, ids, sc, rank, Et // 2, so)
        rows = [0, 5, 36, 17]
        part = m.fuse_moe_blockwise_fp8_rows(x, xs, lambda e: (guw[e], guws[e], dw[e], dws[e]), ids, sc, rows,
                                             rank, Et // 2, so)
        assert torch.equal(full[rows], part)
