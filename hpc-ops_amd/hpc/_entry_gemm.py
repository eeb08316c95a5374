"""torch.ops.hpc.gemm_bf16xfp32 (reference src/gemm/sm90/entry.cc:86-153: same schema, checks and
scratch allocation); compute in csrc/gemm_bf16xfp32.hip."""
import torch

from . import _C

_T = _C.torch_lib
_T.define(
    "gemm_bf16xfp32(Tensor x, Tensor w_high, Tensor w_low, "
    "float scale, bool use_fp32_output, bool use_splitk, Tensor? split_flag) -> (Tensor)"
)


def _gemm_bf16xfp32_entry(x, w_high, w_low, scale, use_fp32_output=False, use_splitk=True, split_flag=None):
    _C.require(x.is_cuda, "x must be a device tensor")
    _C.require(x.is_contiguous(), "x tensor must be contiguous")
    _C.require(w_high.is_contiguous(), "w_high tensor must be contiguous")
    _C.require(w_low.is_contiguous(), "w_low tensor must be contiguous")
    _C.require(x.dtype == torch.bfloat16, "x dtype must be bfloat16")
    _C.require(w_high.dtype == torch.bfloat16, "w_high dtype must be bfloat16")
    _C.require(w_low.dtype == torch.bfloat16, "w_low dtype must be bfloat16")
    m, k = x.shape
    n = w_high.size(0)
    _C.require(n % 64 == 0, "n must to be divided by 64.")
    _C.require(w_high.size(1) == k and tuple(w_low.shape) == tuple(w_high.shape), "weight planes must be [n, k]")
    splits = _C.lib.hpc_gemm_bf16xfp32_splits(m, n, k, int(bool(use_splitk)))
    split_y = flag = None
    flag_ld = 0
    if splits > 1:
        split_y = torch.empty((splits, m, n), dtype=torch.float32, device=x.device)
        # counters: m <= 256 runs on 16-row tiles (flat [m_tiles, n/16]), larger m on a [ceil(m/64), n/64] grid
        if m <= 256:
            tm = 16 if m <= 16 else (32 if m <= 32 else 64)
            rows, flag_ld = (m + tm - 1) // tm, n // 16
        else:
            rows, flag_ld = (m + 63) // 64, n // 64
        if split_flag is not None:
            _C.require(split_flag.dtype == torch.int32 and split_flag.is_contiguous(),
                       "split_flag must be a contiguous int32 tensor")
            if m > 256:
                _C.require(split_flag.dim() == 2 and split_flag.size(1) >= flag_ld and split_flag.size(0) >= rows,
                           "split_flag is too small for this problem")
                flag_ld = split_flag.size(1)
            else:
                _C.require(split_flag.numel() >= rows * flag_ld, "split_flag is too small for this problem")
            flag = split_flag
        else:
            flag = torch.zeros((rows, flag_ld), dtype=torch.int32, device=x.device)
    y = torch.empty((m, n), dtype=torch.float32 if use_fp32_output else torch.bfloat16, device=x.device)
    rc = _C.lib.hpc_gemm_bf16xfp32_async(_C.ptr(y), _C.ptr(split_y), _C.ptr(flag), _C.ptr(x), _C.ptr(w_high),
                                         _C.ptr(w_low), m, n, k, float(scale), int(bool(use_fp32_output)), splits,
                                         flag_ld, _C.stream_of(x))
    _C.check(rc, "gemm_bf16xfp32 launch failed!")
    return y


_T.impl("gemm_bf16xfp32", _gemm_bf16xfp32_entry, "CUDA")


# ---- fused softmax + top-k router (no reference op: pinned to stable torch semantics, oracle/router.py) ----
_T.define("topk_router(Tensor logits, int topk, bool renormalize, Tensor? topk_ids, Tensor? topk_scale) -> (Tensor, Tensor)")


def _topk_router_entry(logits, topk, renormalize=True, topk_ids=None, topk_scale=None):
    import ctypes

    _C.require(logits.is_cuda, "logits must be a device tensor")
    _C.require(logits.dtype == torch.float32, "logits dtype must be float32 (the router GEMM's fp32 output)")
    _C.require(logits.dim() == 2 and logits.stride(1) == 1, "logits must be [num_tokens, num_expert] with unit expert stride")
    m, n = logits.shape
    _C.require(n % 4 == 0 and n <= 1024, "num_expert must be a multiple of 4 and <= 1024")
    _C.require(1 <= topk <= min(n, 64), "topk must be in 1..min(num_expert, 64)")
    _C.require(logits.stride(0) % 4 == 0 and logits.data_ptr() % 16 == 0, "logits rows must be 16-byte aligned")
    if topk_ids is None:
        topk_ids = torch.empty((m, topk), dtype=torch.int32, device=logits.device)
    if topk_scale is None:
        topk_scale = torch.empty((m, topk), dtype=torch.float32, device=logits.device)
    _C.require(topk_ids.dtype == torch.int32 and topk_ids.is_contiguous() and tuple(topk_ids.shape) == (m, topk),
               "topk_ids must be a contiguous int32 [num_tokens, topk] tensor")
    _C.require(topk_scale.dtype == torch.float32 and topk_scale.is_contiguous() and tuple(topk_scale.shape) == (m, topk),
               "topk_scale must be a contiguous float32 [num_tokens, topk] tensor")
    rc = _C.lib.hpc_topk_router_async(ctypes.cast(topk_ids.data_ptr(), ctypes.POINTER(ctypes.c_int)), _C.ptr(topk_scale),
                                      _C.ptr(logits), m, n, logits.stride(0), int(topk), int(bool(renormalize)),
                                      _C.stream_of(logits))
    _C.check(rc, "topk_router")
    return topk_ids, topk_scale


_T.impl("topk_router", _topk_router_entry, "CUDA")
