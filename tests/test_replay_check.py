"""The memory-checked deterministic replay harness (tests/replay_check.py; the MI355X counterpart of the reference's
compute-sanitizer mode, reference conftest.py:14-152): alive on real ops, and it catches planted faults."""
import math

import pytest
import torch

import replay_check as rc
import replay_planted

F8 = torch.float8_e4m3fn


@pytest.mark.gpu
def test_replay_catches_a_store_past_the_end_and_nondeterminism():
    x = torch.randn(64, device="cuda")
    out = torch.empty(64, dtype=F8, device="cuda")
    rc.replay_call("replay_planted:well_behaved", (x, torch.empty_like(x)), {}, replay_planted.well_behaved)
    with pytest.raises(AssertionError, match="guard band above the buffer was overwritten"):
        rc.replay_call("replay_planted:store_past_the_end", (x, out), {}, replay_planted.store_past_the_end)
    with pytest.raises(AssertionError, match="result: bytes differ"):
        rc.replay_call("replay_planted:depends_on_the_process", (x,), {}, replay_planted.depends_on_the_process)


@pytest.mark.gpu
def test_replay_of_hot_path_ops_is_byte_identical_and_in_bounds():
    """decode attention with split requests (arrival counters, in-launch merge), the fused MoE on the 256 x 256 kernel
    incl. a half tile, scaled_fp8_quant with a caller-provided output: each call replayed in a fresh process with guard
    bands around every argument storage.  (`HPC_REPLAY_CHECK=1 pytest tests -m gpu -k ...` replays every public call
    of the selected tests the same way.)"""
    import hpc

    torch.manual_seed(5)
    dev = torch.device("cuda", 0)
    # --- FP8 decode, one long request split over many workgroups
    B, Hkv, Hq, D, P = 5, 4, 32, 128, 64
    lens = torch.tensor([9000, 3, 130, 64, 2500], dtype=torch.int32)
    nb = (lens + P - 1) // P
    nblk = int(nb.sum()) + 3
    qb = torch.randn(B, Hq, D, dtype=torch.bfloat16) / math.sqrt(D)
    qs = qb.float().abs().max(-1)[0] / 10
    q8 = (qb / qs[:, :, None]).to(F8)
    kv = (torch.randn(nblk, 2, P, Hkv, D, dtype=torch.bfloat16) / math.sqrt(D)).to(F8).to(dev)
    perm = torch.randperm(nblk).to(torch.int32)
    bid = torch.zeros(B, int(nb.max()), dtype=torch.int32)
    o = 0
    for i, n in enumerate(nb.tolist()):
        bid[i, :n] = perm[o:o + n]
        o += n
    y = rc.replay_call("attention_decode_fp8",
                       (q8.to(dev), kv[:, 0], kv[:, 1], bid.to(dev), lens.to(dev), qs.to(dev), torch.tensor([0.7], device=dev),
                        torch.tensor([1.3], device=dev)),
                       dict(mtp=0, new_kv_included=True, quant_type=hpc.QuantType.QPERTOKEN_PERHEAD_KPERTENSOR_VPERTENSOR,
                            splitk=True, output=torch.zeros(B, Hq, D, dtype=torch.bfloat16, device=dev)),
                       hpc.attention_decode_fp8)
    assert bool(torch.isfinite(y.float()).all())
    # --- fused MoE: ~300 rows per expert (256 x 256 kernel, last tile = half tile), activation in the epilogue
    T, k, E, H, I = 600, 2, 4, 512, 256
    ids = torch.sort(torch.multinomial(torch.ones(T, E), k).to(torch.int32), dim=1)[0]
    sc = torch.rand(T, k)
    sc = sc / sc.sum(1, keepdim=True)
    args = ((torch.randn(T, H) / 100).to(F8), torch.rand(T, H // 128) + 0.5, torch.randn(E, 2 * I, H).to(F8),
            torch.rand(E, 2 * I // 128, 4) + 0.5, torch.randn(E, H, I).to(F8), torch.rand(E, H // 128, 4) + 0.5, ids, sc)
    rc.replay_call("fuse_moe_blockwise_fp8", tuple(a.to(dev) for a in args) + (0, E), {}, hpc.fuse_moe_blockwise_fp8)
    # --- scaled_fp8_quant into a caller-provided output, ragged element count
    x = torch.randn(37, 123, device=dev)
    rc.replay_call("scaled_fp8_quant", (x, torch.full((), 0.01, device=dev), torch.empty(37, 123, dtype=F8, device=dev)), {},
                   hpc.scaled_fp8_quant)
