"""Parity of hpc.fuse_moe_blockwise_fp8 / reduce / group_gemm_blockwise_fp8 with the CPU oracle.
Generators and tolerance follow reference tests/test_fuse_moe_blockwise.py:265-350 (rtol=atol=0.01)
and tests/test_group_gemm_blockwise.py:50-120."""
import pytest
import torch

from utils import allclose, dev_set, moe_allclose

F8 = torch.float8_e4m3fn


def _inputs(num_tokens, num_topk, hidden, inter, num_expert, size_ep, shared, seed=41):
    torch.manual_seed(seed)
    topk_ids = torch.multinomial(torch.ones((num_tokens, num_expert)), num_topk, replacement=False).to(torch.int32)
    topk_ids, _ = torch.sort(topk_ids, dim=1)
    topk_scale = torch.rand((num_tokens, num_topk))
    topk_scale = topk_scale / topk_scale.sum(dim=1, keepdim=True)
    x = (torch.randn((num_tokens, hidden)) / 100).to(F8)
    x_scale = torch.randn((num_tokens, hidden // 128))
    el = num_expert // size_ep
    guw = torch.randn((el, inter * 2, hidden)).to(F8)
    guws = torch.randn((el, inter * 2 // 128, (hidden // 128 + 3) // 4 * 4))
    dw = torch.randn((el, hidden, inter)).to(F8)
    dws = torch.randn((el, hidden // 128, (inter // 128 + 3) // 4 * 4))
    so = torch.randn((num_tokens, hidden), dtype=torch.bfloat16) if shared else None
    return x, x_scale, guw, guws, dw, dws, topk_ids, topk_scale, so


@pytest.mark.gpu
@pytest.mark.parametrize("num_tokens", [1, 7, 128, 700])
@pytest.mark.parametrize("inter", [512, 256])
@pytest.mark.parametrize("rank_ep,size_ep", [(0, 1), (1, 4), (0, 8)])
@pytest.mark.parametrize("shared", [False, True])
def test_fuse_moe_blockwise_fp8(num_tokens, inter, rank_ep, size_ep, shared):
    import hpc
    from oracle import fuse_moe as omoe

    num_expert, num_topk, hidden = 128, 8, 512
    args = _inputs(num_tokens, num_topk, hidden, inter, num_expert, size_ep, shared)
    x, x_scale, guw, guws, dw, dws, topk_ids, topk_scale, so = args
    gt = omoe.fuse_moe_blockwise_fp8(x, x_scale, guw, guws, dw, dws, topk_ids, topk_scale, rank_ep,
                                     num_expert, so)
    dev = [t.cuda() if t is not None else None for t in args]
    my = hpc.fuse_moe_blockwise_fp8(dev[0], dev[1], dev[2], dev[3], dev[4], dev[5], dev[6], dev[7],
                                    rank_ep, num_expert, dev[8])
    torch.cuda.synchronize()
    assert allclose(gt.float(), my.cpu().float(), rtol=0.01, atol=0.01)
    out = torch.empty_like(my)
    my2 = hpc.fuse_moe_blockwise(dev[0], dev[1], dev[2], dev[3], dev[4], dev[5], dev[6], dev[7],
                                 rank_ep, num_expert, dev[8], out)
    torch.cuda.synchronize()
    assert my2.data_ptr() == out.data_ptr() and torch.equal(my2, my)  # deterministic routing


def _literal_misses(gt, my, rtol=0.01, atol=0.01):
    """(elements, token rows, largest error) outside the literal bar"""
    a, b = gt.float(), my.float()
    err = (a - b).abs()
    bad = err > atol + rtol * a.abs()
    return int(bad.sum()), int(bad.any(dim=-1).sum()), float((err * bad).max())


# (num_tokens, inter, rank_ep, size_ep, shared): a covering sample of the reference grid's 72 large cases - every value of
# every dimension, every num_tokens (2048 is back: VERDICT round 4) with both I and with / without shared output, every
# size_ep with both ranks - 14 cases instead of round 4's 48 (the suite's wall-clock: 3-8 s each); all 72 run in
# tools/moe_literal_report.py -> profiles/round4_moe_literal_fma_form.json
_LARGE_GRID = [(4096, 512, 0, 1, False), (4096, 256, 0, 1, True), (4096, 512, 1, 4, True), (4096, 256, 0, 4, False),
               (4096, 256, 1, 8, False), (4096, 512, 0, 8, True),
               (2048, 512, 0, 1, True), (2048, 256, 0, 1, False), (2048, 512, 1, 4, False), (2048, 256, 1, 8, True),
               (1024, 512, 0, 1, False), (1024, 256, 0, 1, True), (1024, 256, 1, 4, False), (1024, 512, 0, 8, True)]


@pytest.mark.gpu
@pytest.mark.parametrize("num_tokens,inter,rank_ep,size_ep,shared", _LARGE_GRID)
def test_fuse_moe_blockwise_fp8_reference_grid_large(num_tokens, inter, rank_ep, size_ep, shared):
    """The large rows of the reference's own grid (tests/test_fuse_moe_blockwise.py:265-272: num_tokens 1024 / 2048 /
    4096 x E = 128 x H = 512 x I = 512 / 256 x rank_ep 0 / 1 x size_ep 1 / 4 / 8 x shared output), the reference's
    generator (randn scales of either sign, seed 41), through the default dispatch: 64 / 128 / 256 rows per expert on
    average, i.e. the LDS-DMA ring kernel and - from ~192 rows on - the 256 x 256 kernel that carries the graded shape.

    The bar is the reference's LITERAL `allclose(rtol=0.01, atol=0.01)` (:350) - with the one exception this test
    measures instead of hiding.  Two statements of the reference exist on the CPU:
      * the reference KERNEL's arithmetic (oracle kernel_arith=True: one FMA per k block, kernels.cuh:808-834), which
        is the arithmetic of the HIP GEMMs (::test_group_gemm_blockwise_is_the_reference_kernel_arithmetic);
      * the reference TEST's eager model (three roundings per k block, :115-128).
    These two differ from EACH OTHER outside the literal bar on a few elements per million for these CPU-seeded inputs
    (4096 tokens / I = 256: 4 of 2 097 152; 2048 / 512 / rank 1 of 4: 8 of 1 048 576; most cases 0): a sum that sits
    on a bf16 tie rounds the other way, and where that bf16 value sits on an e4m3 tie of the 128-block quantisation
    ONE activation code of one routed row moves by 6 %, which shifts that token's whole output row by ~0.015.  The same
    thing happens between any two correct implementations of the activation (`exp` of the device vs torch's: the last
    fp32 bit of silu(g) * u decides an e4m3 tie) - measured here: HIP against the kernel-arithmetic oracle, 6 elements
    of one token row in one of the 72 cases.  So the assertion is: outside the literal bar there may be at most
    3 token rows more than the kernel-arithmetic oracle itself has against the eager model, none of them off by more
    than 0.06 (one activation code of one expert's contribution; a wrong scale / row / expert is O(1)); every count is
    printed."""
    import hpc
    from oracle import fuse_moe as omoe

    num_expert, num_topk, hidden = 128, 8, 512
    args = _inputs(num_tokens, num_topk, hidden, inter, num_expert, size_ep, shared)
    x, x_scale, guw, guws, dw, dws, topk_ids, topk_scale, so = args
    eager = omoe.fuse_moe_blockwise_fp8(x, x_scale, guw, guws, dw, dws, topk_ids, topk_scale, rank_ep, num_expert, so)
    karith = omoe.fuse_moe_blockwise_fp8(x, x_scale, guw, guws, dw, dws, topk_ids, topk_scale, rank_ep, num_expert, so,
                                         kernel_arith=True)
    dev = [t.cuda() if t is not None else None for t in args]
    my = hpc.fuse_moe_blockwise_fp8(dev[0], dev[1], dev[2], dev[3], dev[4], dev[5], dev[6], dev[7],
                                    rank_ep, num_expert, dev[8]).cpu()
    hip_e, ka_e, hip_k = _literal_misses(eager, my), _literal_misses(eager, karith), _literal_misses(karith, my)
    print("outside the literal (0.01, 0.01) bar, of %d elements / %d rows (elements, rows, max err): HIP vs eager model %s | "
          "kernel-arithmetic oracle vs eager model %s | HIP vs kernel-arithmetic oracle %s"
          % (eager.numel(), eager.shape[0], hip_e, ka_e, hip_k))
    assert hip_k[1] <= 3 and hip_k[2] <= 0.06, hip_k
    assert hip_e[1] <= ka_e[1] + 3 and hip_e[2] <= 0.06, (hip_e, ka_e)


@pytest.mark.dev
@pytest.mark.gpu
@pytest.mark.parametrize("tiled_mode", [0, 1, 2, 3, 4, 12])  # auto, streaming, 256x128 ring (12: 32-token form), 128x128, 256x256
@pytest.mark.parametrize("n,k", [(512, 512), (768, 2048)])
def test_group_gemm_blockwise_is_the_reference_kernel_arithmetic(tiled_mode, n, k):
    """Every grouped-GEMM kernel against the CPU restatement of the reference KERNEL's arithmetic (oracle
    group_gemm_blockwise_kernel_arith, src/group_gemm/kernels.cuh:808-834: `tot = fma(part, xs * ws, tot)` per k
    block), on inputs whose 128-block partial sums are exactly representable in fp32: x = e4m3(randn / 100) is a
    multiple of 2^-9 below 2^-4, w = e4m3(randn) a multiple of 2^-9 below 8, so a block's 128 products are multiples
    of 2^-18 that sum to less than 64 - 24 bits.  The k-block chain is then the only rounding the ALGORITHM has, and
    HIP's is the reference kernel's.  What remains is the matrix pipe: its 128-term block sums are not correctly
    rounded even when representable (measured with unit scales: 49 of 440 320 bf16 outputs differ from the rounded
    exact sum), so the bar is: fewer than 5e-4 of the outputs differ from the restatement at all, and those by one
    bf16 ulp of the value - or, where the k blocks cancel (randn scales of either sign), of the block contributions:
    rtol 2^-7 + atol 1e-3 (a handful of near-zero sums are several ulps of THEMSELVES apart).  Measured: 40 of 440 320
    differ (61 from the reference test's eager model with its three roundings per k block)."""
    import hpc
    from oracle import fuse_moe as omoe

    torch.manual_seed(2)
    seqlens = torch.tensor([300, 0, 129, 5, 128, 1, 257, 40], dtype=torch.int32)
    num_group, total = len(seqlens), int(seqlens.sum())
    x = (torch.randn((total, k)) / 100).to(F8)
    w = torch.randn((num_group, n, k)).clamp(-7.5, 7.5).to(F8)
    kb = k // 128
    xs_rows = torch.randn((total, kb))
    wscale = torch.randn((num_group, n // 128, (kb + 3) // 4 * 4))
    cu = torch.cat([torch.zeros(1, dtype=torch.int32), torch.cumsum(seqlens, 0).to(torch.int32)])
    want = omoe.group_gemm_blockwise_kernel_arith(x, w, seqlens, cu, xs_rows, wscale)
    avg = total // num_group
    tile_m = hpc.aligned_size(avg)
    tiles = (seqlens + tile_m - 1) // tile_m
    cu_tiles = torch.cat([torch.zeros(1, dtype=torch.int64), torch.cumsum(tiles, 0)])
    xs_t = torch.zeros((kb, int(cu_tiles[-1]) * tile_m + 64))
    for g in range(num_group):
        c0 = int(cu_tiles[g]) * tile_m
        xs_t[:, c0 : c0 + int(seqlens[g])] = xs_rows[int(cu[g]) : int(cu[g]) + int(seqlens[g])].t()
    dev_set(3, tiled_mode % 10)
    dev_set(6, 1 + tiled_mode // 10)
    try:
        my = hpc.group_gemm_blockwise_fp8(x.cuda(), w.cuda(), seqlens.cuda(), cu.cuda(), xs_t.cuda(), wscale.cuda(),
                                          num_seq_per_group_avg=avg)
        torch.cuda.synchronize()
    finally:
        dev_set(3, 0)
        dev_set(6, 0)
    eager = omoe.group_gemm_blockwise(x, w, seqlens, cu, xs_rows, wscale)
    got = my.cpu()

    n_ka = int((want.view(torch.int16) != got.view(torch.int16)).sum())
    n_eager = int((eager.view(torch.int16) != got.view(torch.int16)).sum())
    print("bf16 outputs that differ, of %d: from the kernel-arithmetic restatement %d, from the eager model %d"
          % (got.numel(), n_ka, n_eager))
    assert n_ka < 5e-4 * got.numel()
    assert allclose(want.float(), got.float(), rtol=2.0 ** -7, atol=1e-3)


@pytest.mark.dev
@pytest.mark.gpu
@pytest.mark.parametrize("num_tokens,num_expert,num_topk,hidden,inter", [(600, 4, 2, 512, 256), (257, 2, 2, 1024, 384),
                                                                         (1500, 8, 4, 512, 128)])
def test_fused_activation_epilogue(num_tokens, num_expert, num_topk, hidden, inter):
    """long groups (>= 192 rows per expert): the gate-up GEMM runs on the 256 x 256 tile kernel with SiLU * up +
    128-block quantisation in its epilogue.  Bit-equal to the same kernel followed by the separate activation
    kernel (development key 19 = 1), and within the reference tolerance of the oracle; ragged groups (a few rows,
    one row past a 256-row tile) come from the random routing."""
    import hpc
    from oracle import fuse_moe as omoe

    args = _inputs(num_tokens, num_topk, hidden, inter, num_expert, 1, False, seed=5)
    x, x_scale, guw, guws, dw, dws, topk_ids, topk_scale, _ = args
    assert num_tokens * num_topk // num_expert >= 192
    gt = omoe.fuse_moe_blockwise_fp8(x, x_scale, guw, guws, dw, dws, topk_ids, topk_scale, 0, num_expert, None)
    dev = [t.cuda() if t is not None else None for t in args]
    run = lambda: hpc.fuse_moe_blockwise_fp8(dev[0], dev[1], dev[2], dev[3], dev[4], dev[5], dev[6], dev[7], 0, num_expert)
    fused = run()
    dev_set(19, 1)
    try:
        apart = run()
        torch.cuda.synchronize()
    finally:
        dev_set(19, 0)
    assert torch.equal(fused, apart)
    assert allclose(gt.float(), fused.cpu().float(), rtol=0.01, atol=0.01)


@pytest.mark.gpu
@pytest.mark.parametrize("T,k,E,rank,size_ep,misalign", [(333, 8, 128, 1, 4, 0), (4096, 8, 64, 0, 1, 0), (7, 3, 8, 0, 1, 0),
                                                         (1, 1, 4, 1, 2, 0), (1021, 5, 32, 1, 2, 0), (333, 8, 128, 3, 4, 1),
                                                         (259, 7, 16, 0, 1, 3), (1023, 5, 32, 0, 1, 2), (819, 5, 32, 1, 2, 0)])
def test_moe_routing_is_bit_exact(T, k, E, rank, size_ep, misalign):
    """routing indices must be bit-exact (BASELINE north_star): compare the device prep with the
    oracle's stable slotting through the C-ABI.  Round 5 (16-byte id loads, one quarter of the id array per wave):
    entry counts that are no multiple of 4 / 16 / 1024, a single entry, configs[3]'s 32768 entries, and an id array that
    starts 4 / 8 / 12 bytes past a 16-byte boundary (the kernels fall back to 4-byte loads); from 4096 entries on the kernels
    run 16 waves per expert instead of 4 (4095 = 819 x 5 is the last count below)."""
    import hpc  # noqa: F401
    from hpc import _C
    from oracle import fuse_moe as omoe

    torch.manual_seed(3)
    el = E // size_ep
    ids = torch.sort(torch.multinomial(torch.ones((T, E)), k, replacement=False).to(torch.int32), dim=1)[0]
    x = torch.zeros((T, 128)).to(F8)
    _, _, pos_ref, cnt_ref, cu_ref = omoe.gather_expert_inputs(x, torch.zeros(T, 1), ids, el, rank)
    buf = torch.empty(T * k + 4, dtype=torch.int32, device="cuda")
    d = buf[misalign:misalign + T * k].view(T, k)
    d.copy_(ids)
    assert d.data_ptr() % 16 == 4 * misalign
    i32 = dict(dtype=torch.int32, device="cuda")
    seqlens, cu = torch.empty(el, **i32), torch.empty(el + 1, **i32)
    tiles, cut = torch.empty(el, **i32), torch.empty(el + 1, **i32)
    pos, rowidx = torch.empty(T, k, **i32), torch.full((T * k,), -7, **i32)
    rc = _C.lib.hpc_moe_count_and_slot_async(_C.ptr(d), T, k, el, rank, 16, _C.ptr(seqlens), _C.ptr(cu),
                                             _C.ptr(tiles), _C.ptr(cut), _C.ptr(pos), _C.ptr(rowidx),
                                             _C.stream_of(d))
    assert rc == 0
    torch.cuda.synchronize()
    assert torch.equal(pos.cpu(), pos_ref)
    assert torch.equal(seqlens.cpu(), cnt_ref) and torch.equal(cu.cpu(), cu_ref)
    assert torch.equal(tiles.cpu(), (cnt_ref + 15) // 16)
    total = int(cu_ref[-1])
    flat_pos = pos_ref.flatten()
    expect = torch.full((T * k,), -7, dtype=torch.int32)
    sel = flat_pos >= 0
    expect[flat_pos[sel].long()] = (torch.arange(T * k)[sel] // k).to(torch.int32)
    assert torch.equal(rowidx.cpu()[:total], expect[:total])


@pytest.mark.dev
@pytest.mark.gpu
@pytest.mark.parametrize("num_group,actual_m,n,k", [(16, 30, 1024, 4096), (8, 5, 256, 512),
                                                    (4, 70, 384, 1408)])
@pytest.mark.parametrize("forced_mt", [0, 1, 2, 3, 4, 16, 32])
def test_group_gemm_blockwise(num_group, actual_m, n, k, forced_mt):
    """forced_mt pins one streaming-kernel variant (tuning key 1: tokens per pass / waves per workgroup),
    0 = the launcher's own choice; every variant must meet the same bar."""
    import hpc
    from oracle import fuse_moe as omoe

    if forced_mt >= 16 and n % 128:
        pytest.skip("8-wave variants need n % 128 == 0")

    torch.manual_seed(0)
    seqlens = torch.full((num_group,), actual_m, dtype=torch.int32)
    seqlens[-1] = max(actual_m - 3, 0)
    total = int(seqlens.sum())
    x = (torch.randn((total, k)) / 10).to(F8)
    w = (torch.randn((num_group, n, k)) / 10).to(F8)
    kb = k // 128
    xs_rows = torch.randn((total, kb))
    wscale = torch.randn((num_group, n // 128, (kb + 3) // 4 * 4))
    cu = torch.cat([torch.zeros(1, dtype=torch.int32), torch.cumsum(seqlens, 0).to(torch.int32)])
    gt = omoe.group_gemm_blockwise(x, w, seqlens, cu, xs_rows, wscale)
    # reference layout: [K/128, m_pad_total], group g at column cu_tiles[g]*tileM
    tile_m = hpc.aligned_size(actual_m)
    tiles = (seqlens + tile_m - 1) // tile_m
    cu_tiles = torch.cat([torch.zeros(1, dtype=torch.int64), torch.cumsum(tiles, 0)])
    xs_t = torch.zeros((kb, int(cu_tiles[-1]) * tile_m + 64))
    for g in range(num_group):
        c0 = int(cu_tiles[g]) * tile_m
        xs_t[:, c0 : c0 + int(seqlens[g])] = xs_rows[int(cu[g]) : int(cu[g]) + int(seqlens[g])].t()
    dev_set(1, forced_mt)
    try:
        my = hpc.group_gemm_blockwise_fp8(x.cuda(), w.cuda(), seqlens.cuda(), cu.cuda(), xs_t.cuda(),
                                          wscale.cuda(), num_seq_per_group_avg=actual_m)
        torch.cuda.synchronize()
    finally:
        dev_set(1, 0)
    assert allclose(gt.float(), my.cpu().float(), rtol=0.01, atol=0.02)


@pytest.mark.dev
@pytest.mark.gpu
@pytest.mark.parametrize("num_tokens,hidden,inter,num_expert,topk", [(16, 512, 256, 8, 2), (64, 1024, 384, 16, 2), (50, 512, 128, 8, 2),
                                                                     (40, 4096, 1408, 8, 2)])
def test_fuse_moe_blockwise_streaming_kernel_reordered_stage_is_bit_identical(num_tokens, hidden, inter, num_expert, topk):
    """Decode-size batches (below 16 rows per expert) run the streaming grouped GEMM; round 6 re-ordered its stage loop
    (gemm_blockwise_stream2_kernel: refill as soon as the stage's registers are in LDS, early branch-free scale loads, a
    wave-uniform scale descriptor, weight loads in front of the row-index chain) and put a k-block on ONE K = 128 MFMA.
    Development key 56 = 2 (the new loop on chains of four K = 32 MFMAs) must give the bits of key 56 = 1 (the stage loop
    of rounds 1-5); the product form must give the bits of the 256 x 256 kernel (development key 3 = 4: the same MFMA,
    operand convention and FMA order); all meet the oracle's bar."""
    import hpc
    from oracle import fuse_moe as omoe

    args = _inputs(num_tokens, topk, hidden, inter, num_expert, 1, False, seed=num_tokens)
    x, x_scale, guw, guws, dw, dws, topk_ids, topk_scale, _ = args
    assert num_tokens * topk // num_expert < 16  # the streaming kernel's range
    gt = omoe.fuse_moe_blockwise_fp8(x, x_scale, guw, guws, dw, dws, topk_ids, topk_scale, 0, num_expert, None)
    dev = [t.cuda() for t in args[:8]]
    outs = {}
    try:
        for key in (1, 2, 0):
            dev_set(56, key)
            outs[key] = hpc.fuse_moe_blockwise_fp8(*dev, 0, num_expert).cpu()
        dev_set(3, 4)
        outs["p8"] = hpc.fuse_moe_blockwise_fp8(*dev, 0, num_expert).cpu()
    finally:
        dev_set(56, 0)
        dev_set(3, 0)
    assert torch.equal(outs[1], outs[2])
    assert torch.equal(outs[0], outs["p8"])
    # the reference's literal bar at its own sizes (hidden <= 1024); beyond them the bar of the graded-shape tests (utils.moe_allclose)
    close = allclose if hidden <= 1024 else moe_allclose
    assert close(gt.float(), outs[0].float(), rtol=0.01, atol=0.01)
    assert close(gt.float(), outs[1].float(), rtol=0.01, atol=0.01)


@pytest.mark.gpu
def test_reduce():
    import hpc
    from oracle import fuse_moe as omoe

    torch.manual_seed(1)
    T, k, H = 37, 8, 640
    x = torch.randn(T * k, H, dtype=torch.bfloat16)
    pos = torch.randperm(T * k).reshape(T, k).to(torch.int32)
    pos[::3, 2] = -1
    sc = torch.rand(T, k)
    so = torch.randn(T, H, dtype=torch.bfloat16)
    for shared in (None, so):
        gt = omoe.reduce(x, pos, sc, shared)
        my = hpc.reduce(x.cuda(), pos.cuda(), sc.cuda(), None if shared is None else shared.cuda())
        assert allclose(gt.float(), my.cpu().float(), rtol=0.01, atol=0.01)


@pytest.mark.dev
@pytest.mark.gpu
@pytest.mark.parametrize("tiled_mode", [2, 3, 4, 12, 22])  # 2: 256x128 ring kernel (12: its 32-token-tile form); 3: 128x128 kernel; 4: 256x256 kernel
@pytest.mark.parametrize("n,k", [(512, 1024), (768, 4096), (384, 1408)])
def test_group_gemm_blockwise_tiled_kernels(tiled_mode, n, k):
    """the MFMA-bound tiled kernels on ragged groups (empty, 1 token, > 128 tokens, > 256 tokens)."""
    import hpc
    from oracle import fuse_moe as omoe

    torch.manual_seed(1)
    seqlens = torch.tensor([300, 0, 129, 5, 128, 1, 257], dtype=torch.int32)
    num_group, total = len(seqlens), int(seqlens.sum())
    x = (torch.randn((total, k)) / 10).to(F8)
    w = (torch.randn((num_group, n, k)) / 10).to(F8)
    kb = k // 128
    xs_rows = torch.randn((total, kb))
    wscale = torch.randn((num_group, n // 128, (kb + 3) // 4 * 4))
    cu = torch.cat([torch.zeros(1, dtype=torch.int32), torch.cumsum(seqlens, 0).to(torch.int32)])
    gt = omoe.group_gemm_blockwise(x, w, seqlens, cu, xs_rows, wscale)
    avg = total // num_group
    tile_m = hpc.aligned_size(avg)
    tiles = (seqlens + tile_m - 1) // tile_m
    cu_tiles = torch.cat([torch.zeros(1, dtype=torch.int64), torch.cumsum(tiles, 0)])
    xs_t = torch.zeros((kb, int(cu_tiles[-1]) * tile_m + 64))
    for g in range(num_group):
        c0 = int(cu_tiles[g]) * tile_m
        xs_t[:, c0 : c0 + int(seqlens[g])] = xs_rows[int(cu[g]) : int(cu[g]) + int(seqlens[g])].t()
    dev_set(3, tiled_mode % 10)
    dev_set(6, 1 + tiled_mode // 10)  # 1x: 32-token tiles, 2x: 64-token tiles
    try:
        my = hpc.group_gemm_blockwise_fp8(x.cuda(), w.cuda(), seqlens.cuda(), cu.cuda(), xs_t.cuda(), wscale.cuda(),
                                          num_seq_per_group_avg=avg)
        torch.cuda.synchronize()
    finally:
        dev_set(3, 0)
        dev_set(6, 0)
    assert allclose(gt.float(), my.cpu().float(), rtol=0.01, atol=0.02)


@pytest.mark.dev
@pytest.mark.gpu
@pytest.mark.parametrize("tiled_mode", [0, 2, 4])  # auto, 256x128 ring kernel, 256x256 kernel
@pytest.mark.parametrize("num_group", [65, 128, 192, 256])
def test_group_gemm_blockwise_many_groups(tiled_mode, num_group):
    """More than 64 groups (the reference benchmark's presets have 128 / 192 / 256 experts,
    benchmark/fused_moe/benchmark_fuse_moe.py:44-50; reference test grid E = 128): the tiled kernels find their work
    item with one round of lane-parallel loads per 64 groups.  Ragged groups incl. empty ones and ones that need
    two and three 128 / 256-row tiles; every row is checked."""
    import hpc
    from oracle import fuse_moe as omoe

    torch.manual_seed(num_group)
    n, k = 512, 512
    seqlens = torch.randint(150, 330, (num_group,), dtype=torch.int32)
    seqlens[torch.randperm(num_group)[: num_group // 8]] = 0
    seqlens[torch.randperm(num_group)[: num_group // 8]] = 1
    seqlens[-1] = 700
    seqlens[64 % num_group] = 257
    total = int(seqlens.sum())
    x = (torch.randn((total, k)) / 10).to(F8)
    w = (torch.randn((num_group, n, k)) / 10).to(F8)
    kb = k // 128
    xs_rows = torch.rand((total, kb)) + 0.5
    wscale = torch.rand((num_group, n // 128, (kb + 3) // 4 * 4)) + 0.5
    cu = torch.cat([torch.zeros(1, dtype=torch.int32), torch.cumsum(seqlens, 0).to(torch.int32)])
    gt = omoe.group_gemm_blockwise(x, w, seqlens, cu, xs_rows, wscale)
    avg = total // num_group
    tile_m = hpc.aligned_size(avg)
    tiles = (seqlens + tile_m - 1) // tile_m
    cu_tiles = torch.cat([torch.zeros(1, dtype=torch.int64), torch.cumsum(tiles, 0)])
    xs_t = torch.zeros((kb, int(cu_tiles[-1]) * tile_m + 64))
    for g in range(num_group):
        c0 = int(cu_tiles[g]) * tile_m
        xs_t[:, c0 : c0 + int(seqlens[g])] = xs_rows[int(cu[g]) : int(cu[g]) + int(seqlens[g])].t()
    dev_set(3, tiled_mode)
    try:
        my = hpc.group_gemm_blockwise_fp8(x.cuda(), w.cuda(), seqlens.cuda(), cu.cuda(), xs_t.cuda(), wscale.cuda(),
                                          num_seq_per_group_avg=avg)
        torch.cuda.synchronize()
    finally:
        dev_set(3, 0)
    assert allclose(gt.float(), my.cpu().float(), rtol=0.01, atol=0.02)


@pytest.mark.dev
@pytest.mark.gpu
@pytest.mark.parametrize("n,k", [(512, 512), (768, 1408), (256, 2048),    # 4, 11 and 16 k-tiles: shorter than / not a multiple of / longer
                                 (256, 128), (256, 256), (512, 384)])    # 1, 2, 3 k-tiles: shorter than every prefetch depth (round 6: three buffers in the half body)
def test_group_gemm_tail_body_is_bit_identical(n, k):                      # than the rings' period of 6 k-tiles
    """A group's last token tile with <= 64 rows runs the TAIL body of the 256 x 256 kernel (round 5: per-wave weight
    rings, 64-token chunks of three k-slabs, one barrier per three k-tiles; a group's ONLY tile - every third group here -
    with non-temporal weight loads; development key 26 = 1: the variant that streams the weights through six register
    stages, five k-tiles ahead, with token chunks of six k-slabs).  Same operand conventions and the same
    arithmetic order as the full / half-tile bodies, so the output must be BIT-IDENTICAL to the round-4 dispatch
    (development key 21 = 2: tails on the half-tile body) - groups of every tail size 1 ... 64 next to 65, 128, 129 and
    empty groups, blockwise scales of either sign; and within the reference tolerance of the oracle."""
    import hpc
    from oracle import fuse_moe as omoe

    torch.manual_seed(n + k)
    tails = [1, 2, 15, 16, 17, 31, 32, 33, 47, 48, 49, 63, 64, 65, 128, 129, 0, 200]
    seqlens = torch.tensor([256 * (i % 3) + t for i, t in enumerate(tails)] + [512 + 1, 512 + 16, 512 + 17, 512 + 32, 512 + 33,
                                                                              768 + 40, 768 + 48, 768 + 49, 256 + 16, 256 + 17],
                           dtype=torch.int32)  # the second list: tails that ride along with 1 / 2 / 3 full tiles, and the first that do not
    num_group, total = len(seqlens), int(seqlens.sum())
    x = (torch.randn((total, k)) / 10).to(F8)
    w = (torch.randn((num_group, n, k)) / 10).to(F8)
    kb = k // 128
    xs_rows = torch.randn((total, kb))
    wscale = torch.randn((num_group, n // 128, (kb + 3) // 4 * 4))
    cu = torch.cat([torch.zeros(1, dtype=torch.int32), torch.cumsum(seqlens, 0).to(torch.int32)])
    gt = omoe.group_gemm_blockwise(x, w, seqlens, cu, xs_rows, wscale)
    avg = total // num_group
    tile_m = hpc.aligned_size(avg)
    tiles = (seqlens + tile_m - 1) // tile_m
    cu_tiles = torch.cat([torch.zeros(1, dtype=torch.int64), torch.cumsum(tiles, 0)])
    xs_t = torch.zeros((kb, int(cu_tiles[-1]) * tile_m + 64))
    for g in range(num_group):
        c0 = int(cu_tiles[g]) * tile_m
        xs_t[:, c0: c0 + int(seqlens[g])] = xs_rows[int(cu[g]): int(cu[g]) + int(seqlens[g])].t()
    outs = {}
    dev_set(3, 4)  # the 256 x 256 kernel
    try:
        # half-tile body for the tails / tail body (the product: weights through per-wave LDS rings) / full body only /
        # the register-streamed tail body (development key 26 = 1)
        # round 6: groups with f full tiles and a tail of <= 16 f rows have NO tail item - the rows ride along with the full
        # tiles (p8_body<kExt>: a 17th token block); development key 49 = 1 is the dispatch of round 5 (a tail item for
        # every tail): "r5" = that with the tail body, "r5half" = that with the half-tile body
        for key in (2, 0, 1, "regs", "r5", "r5half"):
            dev_set(21, 2 if key == "r5half" else (0 if key in ("regs", "r5") else key))
            dev_set(26, 1 if key == "regs" else 0)
            dev_set(49, 1 if key in ("r5", "r5half") else 0)
            outs[key] = hpc.group_gemm_blockwise_fp8(x.cuda(), w.cuda(), seqlens.cuda(), cu.cuda(), xs_t.cuda(), wscale.cuda(),
                                                     num_seq_per_group_avg=avg).cpu()
    finally:
        dev_set(21, 0)
        dev_set(26, 0)
        dev_set(49, 0)
        dev_set(3, 0)
    assert torch.equal(outs[0], outs[2]) and torch.equal(outs[0], outs[1]) and torch.equal(outs[0], outs["regs"])
    assert torch.equal(outs[0], outs["r5"]) and torch.equal(outs[0], outs["r5half"])
    assert allclose(gt.float(), outs[0].float(), rtol=0.01, atol=0.02)


@pytest.mark.dev
@pytest.mark.gpu
@pytest.mark.parametrize("num_tokens,hidden,inter,num_expert,topk", [(530, 512, 256, 4, 2), (1100, 1024, 384, 8, 2), (300, 512, 128, 2, 2),
                                                                     (1050, 512, 256, 4, 2)])  # the last: ~525 rows per expert
def test_fuse_moe_blockwise_tail_body_is_bit_identical(num_tokens, hidden, inter, num_expert, topk):
    """The fused op with experts whose row counts end in a short tail (~265 / ~275 / 300 rows per expert): gate-up GEMM
    with the activation + 128-block quantisation in the tail body's epilogue (up values handed to the gate wave of the
    same columns through LDS, the token's abs-max folded over the four gate waves) and the down GEMM, against the
    round-4 dispatch (development key 21 = 2) bit for bit, and against the oracle."""
    import hpc
    from oracle import fuse_moe as omoe

    args = _inputs(num_tokens, topk, hidden, inter, num_expert, 1, False, seed=num_tokens)
    x, x_scale, guw, guws, dw, dws, topk_ids, topk_scale, _ = args
    counts = torch.bincount(topk_ids.flatten().long(), minlength=num_expert)
    # the case only means something if some expert really ends in a <= 64-row tile: the tail body + its activation epilogue
    tails = counts % 256
    assert int(((tails > 0) & (tails <= 64)).sum()) > 0, counts.tolist()
    gt = omoe.fuse_moe_blockwise_fp8(x, x_scale, guw, guws, dw, dws, topk_ids, topk_scale, 0, num_expert, None)
    dev = [t.cuda() for t in args[:8]]
    outs = {}
    dev_set(3, 4)
    try:
        for key in (2, 0, "regs", "r5"):  # "r5": development key 49 = 1 - no ride-along rows, every tail its own item
            dev_set(21, 0 if key in ("regs", "r5") else key)
            dev_set(26, 1 if key == "regs" else 0)
            dev_set(49, 1 if key == "r5" else 0)
            outs[key] = hpc.fuse_moe_blockwise_fp8(*dev, 0, num_expert).cpu()
    finally:
        dev_set(21, 0)
        dev_set(26, 0)
        dev_set(49, 0)
        dev_set(3, 0)
    assert torch.equal(outs[0], outs[2]) and torch.equal(outs[0], outs["regs"]) and torch.equal(outs[0], outs["r5"])
    assert allclose(gt.float(), outs[0].float(), rtol=0.01, atol=0.01)


@pytest.mark.dev
@pytest.mark.gpu
@pytest.mark.parametrize("variant", [2, 4, "r5"])  # development key 22: 2 = the round-4 loop (tails behind the barrier), 4 = no s_setprio;
                                                   # "r5": key 49 = 1 - the product loop without ride-along rows (a tail item for every tail)
def test_p8_k_loop_is_bit_identical_to_the_round4_loop_at_the_graded_shape(variant):
    """the k-loop of the 256 x 256 kernel was rewritten in round 5 (the rescale of a section's last
    two blocks carried under the next section's first MFMAs, `CfgProduct::kCarry`) and again in round 6 (a group's short
    tail rides along with its full tiles as a 17th token block, `p8_body<kExt>`; the scale products of the pending blocks are
    no longer copied); same arithmetic in the same order, so the output has to be BIT-IDENTICAL to the round-4 loop (`CfgRound4`,
    development key 22 = 2).  At the graded GEMM shapes of configs[3] - the fused op with H = 4096, I = 11008 (gate-up
    N = 22016 / K = 4096 with the activation epilogue, down N = 4096 / K = 11008 = 86 k-tiles) on four experts whose routed
    row counts end in full, half and tail tiles - and on the standalone grouped GEMM of the down shape."""
    import hpc

    vkey = 0 if variant == "r5" else variant
    g = torch.Generator(device="cuda").manual_seed(22 + vkey)
    T, E, k, H, I = 1100, 4, 2, 4096, 11008
    torch.manual_seed(vkey)
    topk_ids, _ = torch.sort(torch.multinomial(torch.ones((T, E)), k, replacement=False).to(torch.int32), dim=1)
    counts = torch.bincount(topk_ids.flatten().long(), minlength=E)
    assert int(counts.max()) > 512 and int(((counts % 256 > 0) & (counts % 256 <= 64)).sum()) > 0, counts.tolist()
    topk_scale = torch.rand((T, k))
    x = (torch.randn((T, H), device="cuda", generator=g) / 100).to(F8)
    x_scale = torch.randn((T, H // 128), device="cuda", generator=g)
    guw = torch.randn((E, 2 * I, H), device="cuda", generator=g).to(F8)
    guws = torch.randn((E, 2 * I // 128, H // 128), device="cuda", generator=g)
    dw = torch.randn((E, H, I), device="cuda", generator=g).to(F8)
    dws = torch.randn((E, H // 128, (I // 128 + 3) // 4 * 4), device="cuda", generator=g)
    args = (x, x_scale, guw, guws, dw, dws, topk_ids.cuda(), topk_scale.cuda())
    # the standalone GEMM: the down shape, groups of 530 / 300 / 256 / 20 rows
    seqlens = torch.tensor([530, 300, 256, 20], dtype=torch.int32)
    cu = torch.cat([torch.zeros(1, dtype=torch.int32), torch.cumsum(seqlens, 0).to(torch.int32)])
    total, avg = int(seqlens.sum()), int(seqlens.sum()) // 4
    tile_m = hpc.aligned_size(avg)
    tiles = (seqlens + tile_m - 1) // tile_m
    xa = (torch.randn((total, I), device="cuda", generator=g) / 10).to(F8)
    xs_t = torch.randn((I // 128, int(tiles.sum()) * tile_m + 64), device="cuda", generator=g)
    outs = {}
    dev_set(3, 4)  # the 256 x 256 kernel
    try:
        for key in (variant, 0):
            dev_set(22, 0 if key == "r5" else key)
            dev_set(49, 1 if key == "r5" else 0)
            outs[key] = (hpc.fuse_moe_blockwise_fp8(*args, 0, E).cpu(),
                         hpc.group_gemm_blockwise_fp8(xa, dw, seqlens.cuda(), cu.cuda(), xs_t, dws, num_seq_per_group_avg=avg).cpu())
    finally:
        dev_set(22, 0)
        dev_set(49, 0)
        dev_set(3, 0)
    assert torch.isfinite(outs[0][0].float()).all() and float(outs[0][0].float().abs().max()) > 0
    assert torch.equal(outs[0][0], outs[variant][0])
    assert torch.equal(outs[0][1], outs[variant][1])


@pytest.mark.gpu
@pytest.mark.parametrize("num_expert,num_topk", [(128, 8), (256, 8)])
def test_fuse_moe_blockwise_fp8_many_experts(num_expert, num_topk):
    """The fused op with 128 / 256 experts (reference benchmark presets) at a token count that puts ~200+ rows on
    every expert, i.e. on the 256 x 256 kernel with the activation epilogue; sampled rows of every expert."""
    import hpc
    from oracle import fuse_moe as omoe

    torch.manual_seed(num_expert)
    T, H, I = 32 * num_expert, 512, 256
    ids = torch.sort(torch.multinomial(torch.ones(T, num_expert), num_topk, replacement=False).to(torch.int32), dim=1)[0]
    sc = torch.rand(T, num_topk)
    sc = sc / sc.sum(1, keepdim=True)
    x = (torch.randn(T, H) / 10).to(F8)
    xs = torch.rand(T, H // 128) + 0.5
    guw = (torch.randn(num_expert, 2 * I, H) / 10).to(F8)
    dw = (torch.randn(num_expert, H, I) / 10).to(F8)
    guws = torch.rand(num_expert, 2 * I // 128, (H // 128 + 3) // 4 * 4) + 0.5
    dws = torch.rand(num_expert, H // 128, (I // 128 + 3) // 4 * 4) + 0.5
    my = hpc.fuse_moe_blockwise_fp8(x.cuda(), xs.cuda(), guw.cuda(), guws.cuda(), dw.cuda(), dws.cuda(), ids.cuda(),
                                    sc.cuda(), 0, num_expert)
    torch.cuda.synchronize()
    rows = sorted(set(torch.randperm(T)[:512].tolist()))
    fetch = lambda e: (guw[e], guws[e], dw[e], dws[e])  # noqa: E731
    gt = omoe.fuse_moe_blockwise_fp8_rows(x, xs, fetch, ids, sc, rows, 0, num_expert)
    assert allclose(gt.float(), my[rows].cpu().float(), rtol=0.01, atol=0.01)


@pytest.mark.gpu
@pytest.mark.parametrize("num_group,actual_m,m", [(8, 30, 1280), (8, 7, 64), (4, 48, 96), (4, 100, 128)])
def test_reformat_x_scale_and_deepep_group_gemm(num_group, actual_m, m):
    """reformat_x_scale (reference tests/test_group_gemm_blockwise.py:86-148) and the DeepEP-format call it
    serves: every group owns m padded rows of x, seqlens[g] of them valid (:162-223)."""
    import hpc
    from oracle import fuse_moe as omoe

    k, n = 4096, 256
    g = torch.Generator().manual_seed(3)
    total_pad = m * num_group
    xscale = torch.rand((total_pad, k // 128), generator=g)
    seqlens = torch.full((num_group,), actual_m, dtype=torch.int32)
    seqlens[-1] = actual_m - 3
    cu = torch.arange(0, (num_group + 1) * m, m, dtype=torch.int32)
    mean = int(seqlens.sum()) // num_group
    tilem = 8 if mean <= 8 else 16 if mean <= 16 else 32 if mean <= 32 else 48 if mean <= 48 else 64
    ref = torch.zeros((k // 128, total_pad))
    col = 0
    for i in range(num_group):
        c = int(seqlens[i])
        ref[:, col : col + c] = xscale[int(cu[i]) : int(cu[i]) + c].t()
        col += (c + tilem - 1) // tilem * tilem
    out = torch.zeros((k // 128, total_pad), device="cuda")
    got = hpc.reformat_x_scale(xscale.cuda(), seqlens.cuda(), cu.cuda(), mean, out)
    assert got.data_ptr() == out.data_ptr()
    assert torch.equal(ref, got.cpu())  # pure data movement: bit-exact, padding columns untouched (zeros)
    # the grouped GEMM on the reformatted scales == the oracle on the valid rows
    x = (torch.randn((total_pad, k), generator=g) / 10).to(F8)
    w = (torch.randn((num_group, n, k), generator=g) / 10).to(F8)
    wscale = torch.randn((num_group, n // 128, k // 128), generator=g)
    gt = omoe.group_gemm_blockwise(x, w, seqlens, cu, xscale, wscale)
    my = hpc.group_gemm_blockwise_fp8(x.cuda(), w.cuda(), seqlens.cuda(), cu.cuda(), got, wscale.cuda(),
                                      num_seq_per_group_avg=mean)
    for i in range(num_group):
        a, c = int(cu[i]), int(seqlens[i])
        assert allclose(gt[a : a + c].float(), my.cpu()[a : a + c].float(), rtol=0.01, atol=0.05)
