"""Fused softmax + top-k router (hpc.topk_router): bit-exact indices against the stable PyTorch formulation
(oracle/router.py), weights at fp32 tolerance, and the chain router GEMM -> router -> fused MoE against the oracle
chain.  The reference has no router kernel or test (hpc/gemm.py:16-61); BASELINE north_star: "bit-exact for routing
indices and top-k"."""
import pytest
import torch

from utils import allclose

F8 = torch.float8_e4m3fn


def test_router_oracle_agrees_with_torch_topk_and_breaks_ties_low():
    from oracle import router as orouter

    g = torch.Generator().manual_seed(5)
    lg = torch.randn(64, 256, generator=g)
    ids, w = orouter.ref_topk_router(lg, 8, renormalize=False)
    tv, ti = torch.topk(torch.softmax(lg, -1), 8)
    assert torch.equal(ids.long(), ti) and torch.allclose(w, tv)  # tie-free rows: identical to torch.topk
    ids2, w2 = orouter.ref_topk_router(lg, 8, renormalize=True)
    assert torch.equal(ids2, ids) and torch.allclose(w2.sum(-1), torch.ones(64))
    tie = torch.zeros(2, 8)
    tie[0, [5, 2, 7]] = 1.0
    ids3, _ = orouter.ref_topk_router(tie, 4)
    assert ids3[0].tolist() == [2, 5, 7, 0] and ids3[1].tolist() == [0, 1, 2, 3]


@pytest.mark.gpu
@pytest.mark.parametrize("num_tokens", [1, 5, 64, 1000, 4096])
@pytest.mark.parametrize("num_expert,topk", [(256, 8), (64, 8), (128, 6), (384, 4), (1024, 16), (8, 8), (16, 1)])
@pytest.mark.parametrize("renormalize", [True, False])
def test_topk_router_matches_oracle(num_tokens, num_expert, topk, renormalize):
    import hpc
    from oracle import router as orouter

    g = torch.Generator().manual_seed(num_tokens * 31 + num_expert + topk)
    lg = torch.randn(num_tokens, num_expert, generator=g) * 3
    ids, w = hpc.topk_router(lg.cuda(), topk, renormalize)
    rid, rw = orouter.ref_topk_router(lg, topk, renormalize)
    assert ids.dtype == torch.int32 and w.dtype == torch.float32
    assert torch.equal(ids.cpu(), rid)
    assert allclose(rw, w.cpu(), rtol=1e-5, atol=1e-7)


@pytest.mark.gpu
def test_topk_router_ties_go_to_the_smaller_expert_id_and_padded_rows():
    import hpc
    from oracle import router as orouter

    g = torch.Generator().manual_seed(9)
    # bf16-valued logits: many exact ties inside every row
    lg = (torch.randn(300, 256, generator=g)).bfloat16().float().round()
    big = torch.zeros(300, 320)
    big[:, :256] = lg
    big[:, 256:] = 1e9  # padding columns beyond num_expert must never be read as experts
    view = big.cuda()[:, :256]  # row stride 320 floats
    ids, w = hpc.topk_router(view, 8, True)
    rid, rw = orouter.ref_topk_router(lg, 8, True)
    assert torch.equal(ids.cpu(), rid)
    assert allclose(rw, w.cpu(), rtol=1e-5, atol=1e-7)
    # outputs into caller tensors, -inf masked experts are never selected while finite ones remain
    lg2 = torch.randn(7, 64, generator=g)
    lg2[:, 10:50] = float("-inf")
    oi, ow = torch.empty(7, 4, dtype=torch.int32, device="cuda"), torch.empty(7, 4, device="cuda")
    ids2, w2 = hpc.topk_router(lg2.cuda(), 4, False, oi, ow)
    assert ids2.data_ptr() == oi.data_ptr() and w2.data_ptr() == ow.data_ptr()
    rid2, rw2 = orouter.ref_topk_router(lg2, 4, False)
    assert torch.equal(ids2.cpu(), rid2) and allclose(rw2, w2.cpu(), rtol=1e-5, atol=1e-7)
    assert not bool(((ids2 >= 10) & (ids2 < 50)).any())


@pytest.mark.gpu
def test_topk_router_signed_zeros_and_nans():
    """torch.topk's conventions the integer keys must reproduce: -0.0 == +0.0 (the tie goes to the smaller expert id)
    and every NaN, whatever its sign bit, ranks above +inf."""
    import hpc
    from oracle import router as orouter

    torch.manual_seed(3)
    lg = torch.randn(6, 64) - 3.0
    lg[0, [5, 9, 40]] = torch.tensor([-0.0, 0.0, -0.0])          # zeros are the row maximum: ids 5, 9, 40 in id order
    lg[1, [7, 3]] = torch.tensor([0.0, -0.0])
    neg_nan = torch.tensor([0xFFC00001], dtype=torch.uint32).view(torch.float32)[0]
    lg[2, 11], lg[2, 30] = neg_nan, float("inf")                  # a negative-sign NaN above +inf
    lg[3, 2], lg[3, 50] = float("nan"), neg_nan                   # two NaNs: the smaller id first
    lg[4, :] = 0.0
    lg[4, ::2] = -0.0                                             # a whole row of signed zeros
    for renorm in (False, True):
        rid, rw = orouter.ref_topk_router(lg, 8, renorm)
        mid, mw = hpc.topk_router(lg.cuda(), 8, renorm)
        assert torch.equal(mid.cpu(), rid.to(torch.int32)), (mid.cpu(), rid)
        fin = torch.isfinite(rw) & torch.isfinite(mw.cpu())
        assert torch.allclose(mw.cpu()[fin], rw[fin], atol=1e-6)
    assert mid[0, :3].tolist() == [5, 9, 40] and mid[1, :2].tolist() == [3, 7]
    assert mid[2, :2].tolist() == [11, 30] and mid[3, :2].tolist() == [2, 50]
    assert mid[4].tolist() == list(range(8))


@pytest.mark.gpu
def test_topk_router_error_paths():
    import hpc

    lg = torch.randn(4, 64, device="cuda")
    with pytest.raises(RuntimeError):
        hpc.topk_router(lg.double(), 4)
    with pytest.raises(RuntimeError):
        hpc.topk_router(lg, 0)
    with pytest.raises(RuntimeError):
        hpc.topk_router(lg, 65)
    with pytest.raises(RuntimeError):
        hpc.topk_router(torch.randn(4, 62, device="cuda"), 4)
    with pytest.raises(RuntimeError):
        hpc.topk_router(lg.cpu(), 4)


@pytest.mark.gpu
@pytest.mark.parametrize("num_tokens", [9, 200])
def test_router_chain_gemm_topk_fused_moe(num_tokens):
    """x -> gemm_bf16xfp32 (router logits, fp32) -> topk_router -> fuse_moe_blockwise_fp8, every stage on the device,
    against the oracle chain fed with the device logits (the GEMM has its own parity test): indices bit-exact,
    MoE output at the reference tolerance."""
    import hpc
    from oracle import fuse_moe as omoe
    from oracle import gemm as ogemm
    from oracle import router as orouter

    torch.manual_seed(3)
    E, k, H, I = 64, 8, 512, 256
    xb = torch.randn(num_tokens, H).bfloat16()
    wr = torch.randn(E, H)
    wh, wl = ogemm.split_weight(wr)
    logits = hpc.gemm_bf16xfp32(xb.cuda(), wh.cuda(), wl.cuda(), 1 / 256, True)
    assert allclose(ogemm.two_plane(xb, wh, wl, 1 / 256), logits.cpu(), rtol=1e-4, atol=2e-3)
    ids, sc = hpc.topk_router(logits, k, True)
    rid, rsc = orouter.ref_topk_router(logits.cpu(), k, True)
    assert torch.equal(ids.cpu(), rid) and allclose(rsc, sc.cpu(), rtol=1e-5, atol=1e-7)
    x8, xs = (xb.float() / 100).to(F8), torch.randn(num_tokens, H // 128)
    guw, guws = torch.randn(E, 2 * I, H).to(F8), torch.randn(E, 2 * I // 128, 4)
    dw, dws = torch.randn(E, H, I).to(F8), torch.randn(E, H // 128, 4)
    my = hpc.fuse_moe_blockwise_fp8(x8.cuda(), xs.cuda(), guw.cuda(), guws.cuda(), dw.cuda(), dws.cuda(), ids, sc, 0, E)
    gt = omoe.fuse_moe_blockwise_fp8(x8, xs, guw, guws, dw, dws, rid, rsc, 0, E)
    torch.cuda.synchronize()
    assert allclose(gt.float(), my.cpu().float(), rtol=0.01, atol=0.01)
