"""Parity of hpc.fused_rmsnorm_with_scale (HIP, via the C-ABI) with the oracle.
Same cases and tolerances as reference tests/test_normalization.py:31-66, plus config C1
(BASELINE.json configs[0]: hidden 4096, 1024 tokens) and hidden 8192."""
import pytest
import torch

from utils import allclose


@pytest.mark.gpu
@pytest.mark.parametrize("batch_size", [1, 2, 4, 5, 8, 14, 16, 17, 32, 64, 1024])
@pytest.mark.parametrize("hidden_states", [5120, 320, 4096, 8192])
@pytest.mark.parametrize("scale", [2.5])
@pytest.mark.parametrize("is_moe", [False, True])
def test_fused_rmsnorm_with_scale(batch_size, hidden_states, scale, is_moe):
    import hpc
    from oracle import normalization as onorm

    torch.manual_seed(0)
    rmsnorm_weight = torch.rand((1, hidden_states), dtype=torch.bfloat16)
    x = torch.randn(batch_size, hidden_states, dtype=torch.bfloat16)
    scale_cpu = torch.tensor([scale, 2 * scale] if is_moe else [scale], dtype=torch.float32)
    eps = 1e-6

    gt = onorm.rmsnorm_with_scale_fp8(x, rmsnorm_weight, scale_cpu[0], eps)
    gt_2 = onorm.rmsnorm_with_scale_fp8(x, rmsnorm_weight, scale_cpu[1], eps) if is_moe else gt
    gt_fp32 = onorm.rmsnorm_fp32(x, rmsnorm_weight, eps)

    output = hpc.normalization.fused_rmsnorm_with_scale(
        x.cuda(), rmsnorm_weight.cuda(), scale=scale_cpu.cuda(), eps=eps, is_moe=is_moe
    )
    torch.cuda.synchronize()
    if is_moe:
        y_fp32, y_fp8, y_fp8_2 = [t.cpu() for t in output]
    else:
        y_fp32, y_fp8, y_fp8_2 = gt_fp32, output.cpu(), output.cpu()

    assert y_fp8.dtype == torch.float8_e4m3fn
    assert allclose(gt_fp32, y_fp32)
    assert allclose(gt_2, y_fp8_2.to(torch.bfloat16), atol=0.15, rtol=0.0125)
    assert allclose(gt, y_fp8.to(torch.bfloat16), atol=0.15, rtol=0.0125)


@pytest.mark.gpu
def test_rmsnorm_default_scale_and_errors():
    import hpc

    x = torch.randn(3, 4096, dtype=torch.bfloat16, device="cuda")
    w = torch.rand(4096, dtype=torch.bfloat16, device="cuda")
    y = hpc.fused_rmsnorm_with_scale(x, w)  # default eps / CPU scale tensor moved to device
    assert y.dtype == torch.float8_e4m3fn and y.shape == x.shape
    with pytest.raises(RuntimeError):
        hpc.fused_rmsnorm_with_scale(x.float(), w)
    with pytest.raises(RuntimeError):  # hidden not a multiple of 8 -> launcher refuses
        hpc.fused_rmsnorm_with_scale(x[:, :4095].contiguous(), w[:4095].contiguous())
