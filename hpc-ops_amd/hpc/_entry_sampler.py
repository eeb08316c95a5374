"""torch.ops.hpc.fused_sampler / fused_sampler_temperature_sample (reference src/sampler/entry.cc:13-275:
same schemas, checks and messages); compute in csrc/sampler.hip."""
import torch

from . import _C

_T = _C.torch_lib
_T.define(
    "fused_sampler(Tensor logits, Tensor? penalty_mask, Tensor? slot_id, "
    "Tensor? repetition_penalty, float repetition_penalty_val, "
    "Tensor? temperature, float temperature_val, "
    "int softmax_policy, "
    "Tensor? topk, int topk_val, "
    "Tensor? topp, float topp_val, "
    "int max_topk, "
    "Tensor? gumbel_noise=None, int seed=0) -> Tensor"
)
_T.define(
    "fused_sampler_temperature_sample(Tensor logits, Tensor? temperature, "
    "float temperature_val, Tensor? gumbel_noise=None, "
    "Tensor? draft_token_ids=None, "
    "int seed=0) -> Tensor"
)


def _check_logits(logits, who):
    _C.require(logits.is_cuda, "logits must be a device tensor")
    _C.require(logits.dim() == 2, "logits tensor must be dim == 2")
    _C.require(logits.dtype in (torch.float32, torch.bfloat16), "logits dtype must be float32 or bfloat16")
    b, v = logits.shape
    _C.require(logits.stride(1) == 1,
               f"{who}: logits must have contiguous inner dim (stride(1)=1), got stride(1)={logits.stride(1)}")
    _C.require(logits.stride(0) >= v, f"{who}: logits stride(0)={logits.stride(0)} must be >= vocab_size={v}")
    _C.require(v % 8 == 0 and v < (1 << 20), f"{who}: unsupported vocab_size {v} (must be a multiple of 8, < 2^20)")
    return b, v


def _check_1d_float(t, name, b):
    if t is None:
        return
    _C.require(t.is_contiguous(), f"{name} tensor must be contiguous")
    _C.require(t.dtype == torch.float32, f"{name} dtype must be float32")
    _C.require(t.dim() == 1, f"{name} tensor must be 1D")
    _C.require(t.size(0) == b, f"{name} size must be [batch_size={b}], got [{t.size(0)}]")


def _check_noise(g, b, v):
    if g is None:
        return
    _C.require(g.is_contiguous(), "gumbel_noise tensor must be contiguous")
    _C.require(g.dtype == torch.float32, "gumbel_noise dtype must be float32")
    _C.require(g.dim() == 2, "gumbel_noise must be 2D")
    _C.require(g.size(0) == b and g.size(1) == v, f"gumbel_noise shape must be [{b}, {v}]")


def _fused_sampler_entry(logits, penalty_mask, slot_id, repetition_penalty, repetition_penalty_val, temperature,
                         temperature_val, softmax_policy, topk, topk_val, topp, topp_val, max_topk,
                         gumbel_noise=None, seed=0):
    b, v = _check_logits(logits, "fused_sampler")
    _C.require(0 <= softmax_policy <= 2, "softmax_policy must be one of 0(NONE)/1(BEFORE_TOPK)/2(AFTER_TOPK)")
    _C.require((penalty_mask is None) == (slot_id is None),
               "penalty_mask and slot_id must both be provided or both be omitted")
    if penalty_mask is not None:
        _C.require(penalty_mask.is_contiguous(), "penalty_mask tensor must be contiguous")
        _C.require(penalty_mask.dtype == torch.uint8, "penalty_mask dtype must be uint8")
        _C.require(penalty_mask.dim() == 2, "penalty_mask must be 2D [MAX_BS, ceil(V/8)]")
        _C.require(penalty_mask.size(1) >= (v + 7) // 8,
                   f"penalty_mask dim 1 must be >= {(v + 7) // 8}, got {penalty_mask.size(1)}")
        _C.require(slot_id.is_contiguous(), "slot_id tensor must be contiguous")
        _C.require(slot_id.dtype == torch.int32, "slot_id dtype must be int32")
        _C.require(slot_id.dim() == 1, "slot_id must be 1D")
        _C.require(slot_id.size(0) == b, f"slot_id.size(0) must equal batch_size={b}, got {slot_id.size(0)}")
        _C.require(penalty_mask.size(0) >= b, f"penalty_mask.size(0)(MAX_BS) must be >= batch_size={b}")
    _check_1d_float(repetition_penalty, "repetition_penalty", b)
    _check_1d_float(temperature, "temperature", b)
    _check_1d_float(topp, "topp", b)
    topk_bytes = 0
    if topk is not None:
        _C.require(topk.is_contiguous(), "topk tensor must be contiguous")
        _C.require(topk.dim() == 1, "topk tensor must be 1D")
        _C.require(topk.size(0) == b, f"topk size must be [batch_size={b}], got [{topk.size(0)}]")
        _C.require(topk.dtype in (torch.int32, torch.int64), "topk dtype must be int32 or int64")
        topk_bytes = 4 if topk.dtype == torch.int32 else 8
    has_rp = repetition_penalty is not None or repetition_penalty_val > 0.0
    has_topk = topk is not None or topk_val > 0
    has_topp = topp is not None or topp_val > 0.0
    _C.require(not has_rp or penalty_mask is not None,
               "repetition_penalty is enabled but penalty_mask/slot_id are missing")
    _C.require(not has_topp or has_topk, "topp requires topk to be enabled (kernel does not support bare topp)")
    _C.require(not has_topp or softmax_policy != 0,
               "topp requires softmax_policy != NONE (BEFORE_TOPK or AFTER_TOPK)")
    _C.require(softmax_policy == 0 or has_topp,
               "softmax_policy != NONE requires topp to be enabled (softmax has no effect on sampling without topp)")
    _C.require(max_topk in (32, 64), f"max_topk must be 32 or 64, got {max_topk}")
    _check_noise(gumbel_noise, b, v)
    if gumbel_noise is None:
        _C.require(seed > 0, f"fused_sampler: seed must be > 0 when gumbel_noise is not provided, got seed={seed}")
    token_ids = torch.empty((b, 1), dtype=torch.int32, device=logits.device)
    if b == 0:
        return token_ids
    ws = torch.empty((_C.lib.hpc_fused_sampler_workspace_bytes(b, v, max_topk),), dtype=torch.uint8,
                     device=logits.device)
    rc = _C.lib.hpc_fused_sampler_async(
        _C.ptr(token_ids), _C.ptr(ws), _C.ptr(logits), 0 if logits.dtype == torch.float32 else 1,
        _C.ptr(penalty_mask), penalty_mask.stride(0) if penalty_mask is not None else 0, _C.ptr(slot_id),
        _C.ptr(repetition_penalty), float(repetition_penalty_val), _C.ptr(temperature), float(temperature_val),
        int(softmax_policy), _C.ptr(topk), topk_bytes, int(topk_val), _C.ptr(topp), float(topp_val),
        _C.ptr(gumbel_noise), b, v, logits.stride(0), int(max_topk), int(seed) if gumbel_noise is None else 0,
        _C.stream_of(logits))
    _C.check(rc, "fused_sampler_async")
    return token_ids


def _temperature_entry(logits, temperature, temperature_val, gumbel_noise=None, draft_token_ids=None, seed=0):
    b, v = _check_logits(logits, "fused_sampler_temperature_sample")
    if temperature is not None:
        _check_1d_float(temperature, "temperature", b)
        if temperature.numel():
            tmin = float(temperature.min())
            _C.require(tmin > 0.0, "fused_sampler_temperature_sample: every temperature tensor element must be > 0, "
                                   f"got min={tmin}")
    else:
        _C.require(temperature_val > 0.0,
                   f"fused_sampler_temperature_sample: scalar temperature must be > 0, got {temperature_val}")
    _check_noise(gumbel_noise, b, v)
    if draft_token_ids is not None:
        _C.require(draft_token_ids.is_contiguous(), "draft_token_ids tensor must be contiguous")
        _C.require(draft_token_ids.dtype == torch.int64, "draft_token_ids dtype must be int64")
        _C.require(draft_token_ids.dim() == 1, "draft_token_ids must be 1D")
        _C.require(draft_token_ids.size(0) == b,
                   f"draft_token_ids size must be [batch_size={b}], got [{draft_token_ids.size(0)}]")
    if gumbel_noise is None:
        _C.require(seed > 0, "fused_sampler_temperature_sample: seed must be > 0 when gumbel_noise is not "
                             f"provided, got seed={seed}")
    token_ids = torch.empty((b, 1), dtype=torch.int32, device=logits.device)
    if b == 0:
        return token_ids
    ws = torch.empty((b * _C.lib.hpc_sampler_segments(v),), dtype=torch.int64, device=logits.device)
    rc = _C.lib.hpc_fused_sampler_temperature_async(
        _C.ptr(token_ids), _C.ptr(ws), _C.ptr(logits), 0 if logits.dtype == torch.float32 else 1, logits.stride(0),
        _C.ptr(temperature), float(temperature_val), _C.ptr(gumbel_noise), _C.ptr(draft_token_ids), b, v,
        int(seed) if gumbel_noise is None else 0, _C.stream_of(logits))
    _C.check(rc, "fused_sampler_temperature_async")
    return token_ids


_T.impl("fused_sampler", _fused_sampler_entry, "CUDA")
_T.impl("fused_sampler_temperature_sample", _temperature_entry, "CUDA")
