"""hpc.sampler — fused sampler at the end of a decode step (reference hpc/sampler.py:8-330)."""
from enum import IntEnum
from typing import Optional, Tuple, Union

import torch
from torch import Tensor

from . import _C  # noqa: F401  (loads the libraries that register torch.ops.hpc.*)


class SoftmaxPolicy(IntEnum):
    """Where the sampler runs softmax (at most once per step; reference hpc/sampler.py:8-28):
    NONE — top-k / Gumbel-max work on logits; BEFORE_TOPK — softmax over the full vocabulary, top-k and
    top-p on probabilities; AFTER_TOPK — top-k on logits, softmax over the surviving top-k, top-p on it.
    top-p requires a policy != NONE."""

    NONE = 0
    BEFORE_TOPK = 1
    AFTER_TOPK = 2


def _to_tensor_scalar_tuple(x) -> Tuple[Optional[Tensor], Union[int, float]]:
    if isinstance(x, torch.Tensor):
        if x.dtype == torch.float:
            return (x, 0.0)
        if x.dtype in (torch.int32, torch.int64):
            return (x, 0)
        raise ValueError(f"Unsupported dtype {x.dtype}")
    return (None, x)


def fused_sampler(
    logits: Tensor,
    *,
    penalty_mask: Optional[Tensor] = None,
    slot_id: Optional[Tensor] = None,
    repetition_penalty: Union[Tensor, float] = 0.0,
    temperature: Union[Tensor, float] = 0.0,
    softmax_policy: SoftmaxPolicy = SoftmaxPolicy.NONE,
    topk: Union[Tensor, int] = 0,
    topp: Union[Tensor, float] = 0.0,
    max_topk: int = 32,
    gumbel_noise: Optional[Tensor] = None,
    draft_token_ids: Optional[Tensor] = None,
    seed: int = 0,
) -> Tensor:
    """repetition_penalty -> temperature -> [softmax] -> topk -> [softmax] -> topp -> Gumbel-max ->
    penalty write-back; every stage except the final Gumbel-max is optional and sampling always happens
    inside the top-``max_topk`` (32 / 64) candidates.

    logits [B, V] float32 / bfloat16 (inner stride 1; V % 8 == 0); penalty_mask [MAX_BS, ceil(V/8)] uint8
    bit mask + slot_id [B] int32 (together or not at all; the sampled token's bit is OR-ed in);
    repetition_penalty / temperature / topp: scalar or [B] float32 (0 disables); topk: scalar or [B]
    int32/int64 (<= max_topk; 0 = max_topk); gumbel_noise [B, V] float32 makes sampling reproducible
    against the PyTorch model, otherwise Philox noise is drawn from ``seed`` (> 0); draft_token_ids [B]
    int64 (-1 = none) masks one token per row, temperature-only fast path only.
    Returns token_ids int32 [B, 1].  (reference hpc/sampler.py:42-200)"""
    if isinstance(softmax_policy, int) and not isinstance(softmax_policy, SoftmaxPolicy):
        softmax_policy = SoftmaxPolicy(softmax_policy)
    if max_topk not in (32, 64):
        raise ValueError(f"fused_sampler: max_topk must be 32 or 64, got {max_topk}.")

    def _is_scalar_zero(x):
        return (not isinstance(x, Tensor)) and float(x) == 0.0

    temp_is_tensor = isinstance(temperature, Tensor)
    fast_path = (
        penalty_mask is None
        and slot_id is None
        and _is_scalar_zero(repetition_penalty)
        and _is_scalar_zero(topp)
        and (not isinstance(topk, Tensor))
        and int(topk) == 0
        and softmax_policy == SoftmaxPolicy.NONE
        and (temp_is_tensor or float(temperature) > 0.0)
    )
    if fast_path:
        temp_tensor, temp_scalar = _to_tensor_scalar_tuple(temperature)
        return torch.ops.hpc.fused_sampler_temperature_sample(
            logits, temp_tensor, float(temp_scalar), gumbel_noise, draft_token_ids, seed)
    if draft_token_ids is not None:
        raise ValueError(
            "draft_token_ids currently requires the temperature-only fast path. Disable the other sampler features "
            "(penalty_mask/slot_id/repetition_penalty/topk/topp/softmax_policy) to use draft-mask sampling.")
    return torch.ops.hpc.fused_sampler(
        logits, penalty_mask, slot_id,
        *_to_tensor_scalar_tuple(repetition_penalty),
        *_to_tensor_scalar_tuple(temperature),
        int(softmax_policy),
        *_to_tensor_scalar_tuple(topk),
        *_to_tensor_scalar_tuple(topp),
        max_topk, gumbel_noise, seed,
    )


@torch.library.register_fake("hpc::fused_sampler")
def _fused_sampler_fake(logits, penalty_mask, slot_id, repetition_penalty, repetition_penalty_val, temperature,
                        temperature_val, softmax_policy, topk, topk_val, topp, topp_val, max_topk,
                        gumbel_noise=None, seed=0):
    return torch.empty((logits.shape[0], 1), dtype=torch.int32, device=logits.device)


@torch.library.register_fake("hpc::fused_sampler_temperature_sample")
def _fused_sampler_temperature_fake(logits, temperature, temperature_val, gumbel_noise=None, draft_token_ids=None,
                                    seed=0):
    return torch.empty((logits.shape[0], 1), dtype=torch.int32, device=logits.device)
