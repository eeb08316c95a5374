"""Top-k router oracle - TEST INFRASTRUCTURE (see oracle/__init__.py).

The reference has no router kernel and no router test (hpc/gemm.py:16-61 stops at the GEMM; its MoE tests draw
topk_ids with torch.multinomial).  BASELINE north_star asks for a fused top-k router, so the parity definition
is the stable PyTorch-eager formulation below: parity unpinned against reference code (there is none), pinned
against torch.topk / torch.softmax - `ref_topk_router` agrees with torch.topk on every tie-free row
(tests/test_router.py) and resolves ties to the smaller expert id, which torch.topk leaves unspecified."""
import torch


def ref_topk_router(logits, topk, renormalize=True):
    """logits f32 [m, n] -> (ids int32 [m, topk] best first, weights f32 [m, topk])."""
    lf = logits.float()
    order = torch.sort(lf, dim=-1, descending=True, stable=True).indices
    ids = order[:, :topk]
    p = torch.softmax(lf, dim=-1)
    w = torch.gather(p, 1, ids)
    if renormalize:
        w = w / w.sum(-1, keepdim=True)
    return ids.to(torch.int32), w
