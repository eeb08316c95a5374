"""hpc.gemm — router GEMM with an fp32 weight carried as two bf16 planes (reference hpc/gemm.py:7-80)."""
import torch
from torch import Tensor

from . import _C  # noqa: F401  (loads the libraries that register torch.ops.hpc.*)


def get_gemm_bf16xfp32_workspace(max_weight_hidden_size: int, max_tokens: int = 131072) -> Tensor:
    """Zeroed split-K arrival counters, one per (16-token, 64-row) tile (reference hpc/gemm.py:7-13)."""
    nm_max = (max_tokens + 15) // 16
    nn_max = (max_weight_hidden_size + 63) // 64
    return torch.zeros((nm_max, nn_max), dtype=torch.int32, device="cuda")


def gemm_bf16xfp32(
    x: Tensor,
    w_high: Tensor,
    w_low: Tensor,
    scale: float,
    use_fp32_output: bool = False,
    use_splitk: bool = True,
    split_flag: Tensor = None,
) -> Tensor:
    """y = x @ (w_high + scale * w_low)^T with fp32 accumulation, where w_high = bf16(w_fp32) and
    w_low = bf16((w_fp32 - w_high) / scale), scale = 1/256.
    x [m, k] bf16; w_high, w_low [n, k] bf16 (n % 64 == 0, k % 64 == 0); returns [m, n] bf16 (or fp32
    with use_fp32_output).  split_flag (optional, from get_gemm_bf16xfp32_workspace) must be zero on
    entry and is zero again on exit."""
    return torch.ops.hpc.gemm_bf16xfp32(x, w_high, w_low, scale, use_fp32_output, use_splitk, split_flag)


@torch.library.register_fake("hpc::gemm_bf16xfp32")
def _gemm_bf16xfp32_fake(a, b_high, b_low, scale, use_fp32_output=False, use_splitk=True, split_flag=None):
    return torch.empty((a.shape[0], b_high.shape[0]), dtype=torch.float32 if use_fp32_output else a.dtype,
                       device=a.device)


def topk_router(logits: Tensor, topk: int, renormalize: bool = True, topk_ids: Tensor = None,
                topk_scale: Tensor = None):
    """Fused softmax + top-k over the router logits (the fp32 output of gemm_bf16xfp32): returns
    (topk_ids int32 [m, topk], topk_scale float32 [m, topk]) - what fuse_moe / fuse_moe_blockwise_fp8 take.

    No reference counterpart (the reference stops at the GEMM); semantics = stable PyTorch:
    ids = argsort(logits, descending, stable)[:, :topk] (ties -> smaller expert id, bit-exact),
    p = softmax(logits.float(), -1), scale = p[ids] (renormalize=False) or p[ids] / p[ids].sum(-1) (True).
    logits [m, num_expert] float32, num_expert % 4 == 0 and <= 1024, topk <= 64."""
    return torch.ops.hpc.topk_router(logits, topk, renormalize, topk_ids, topk_scale)


@torch.library.register_fake("hpc::topk_router")
def _topk_router_fake(logits, topk, renormalize=True, topk_ids=None, topk_scale=None):
    m = logits.shape[0]
    return (torch.empty((m, topk), dtype=torch.int32, device=logits.device),
            torch.empty((m, topk), dtype=torch.float32, device=logits.device))
