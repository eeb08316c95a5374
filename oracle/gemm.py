"""Router GEMM oracle — TEST INFRASTRUCTURE (see oracle/__init__.py).
The reference's parity definition is inline in tests/test_gemm_bf16xfp32.py:29-45: weights are split as
w_high = bf16(w), w_low = bf16((w - w_high) / scale) and the kernel output is compared with
torch.matmul(x.float(), w.t()) at rtol 0.08 / atol 0.01.  `split_weight` and `ground_truth` restate those
lines; `two_plane` is what the kernel computes exactly (fp64 accumulate) for tight checks."""
import torch


def split_weight(w, scale=1.0 / 256):
    w_high = w.to(torch.bfloat16)
    w_low = ((w - w_high.float()) / scale).to(torch.bfloat16)
    return w_high, w_low


def ground_truth(x, w):
    return torch.matmul(x.float(), w.t())


def two_plane(x, w_high, w_low, scale):
    xd = x.double()
    return (xd @ w_high.double().t() + scale * (xd @ w_low.double().t())).float()
