"""gemm_bf16xfp32 parity (grid of reference tests/test_gemm_bf16xfp32.py:13-45, large m trimmed to what
the CPU oracle finishes in seconds)."""
import pytest
import torch

from oracle import gemm as orc
from utils import allclose


def make(m, n, k, seed=10086):
    g = torch.Generator().manual_seed(seed)
    x = torch.randn(m, k, generator=g).bfloat16()
    w = torch.randn(n, k, generator=g)
    return x, w


def test_oracle_two_plane_recovers_fp32_weight():
    """CPU: the two bf16 planes carry the fp32 weight to ~2^-16 relative, so the kernel's exact target
    (two_plane) sits far inside the reference tolerance around the fp32 ground truth."""
    x, w = make(6, 192, 512)
    wh, wl = orc.split_weight(w)
    assert float(((wh.float() + wl.float() / 256) - w).abs().max()) < 4e-5
    assert torch.allclose(orc.two_plane(x, wh, wl, 1 / 256), orc.ground_truth(x, w), rtol=1e-3, atol=2e-3)


@pytest.mark.gpu
@pytest.mark.parametrize("m", [1, 6, 16, 32, 48, 64, 96, 144, 208, 304, 624, 1024, 4096, 12303])
@pytest.mark.parametrize("n", [192, 512, 2048])
@pytest.mark.parametrize("use_fp32_output", [True, False])
@pytest.mark.parametrize("use_split_flag", [True, False])
def test_gemm_bf16xfp32(m, n, use_fp32_output, use_split_flag):
    import hpc

    k = 4096
    if m > 1024 and (n != 512 or not use_split_flag):
        pytest.skip("large-m cases run once per output type")
    x, w = make(m, n, k)
    scale = 1 / 256
    wh, wl = orc.split_weight(w, scale)
    gt = orc.ground_truth(x, w)
    exact = orc.two_plane(x, wh, wl, scale)
    flag = hpc.get_gemm_bf16xfp32_workspace(n) if use_split_flag else None
    my = hpc.gemm_bf16xfp32(x.cuda(), wh.cuda(), wl.cuda(), scale, use_fp32_output, True, flag)
    assert my.dtype == (torch.float32 if use_fp32_output else torch.bfloat16)
    if use_split_flag:
        assert (flag == 0).all()
    assert allclose(gt, my.float().cpu(), rtol=0.08, atol=0.01)  # the reference's bar
    # our own, tighter: fp32 accumulation of the two planes (bf16 output rounding when requested)
    if use_fp32_output:
        assert allclose(exact, my.cpu(), rtol=1e-4, atol=2e-3)
    else:
        assert allclose(exact.bfloat16(), my.cpu(), rtol=8e-3, atol=2e-3)


@pytest.mark.gpu
@pytest.mark.parametrize("m,n,k", [(5, 64, 64), (17, 128, 192), (33, 64, 8192), (200, 256, 1088)])
def test_gemm_bf16xfp32_shapes_and_determinism(m, n, k):
    """odd m / small and non-power-of-two k, no split-K, and bit-identical results call to call."""
    import hpc

    x, w = make(m, n, k, seed=3)
    wh, wl = orc.split_weight(w)
    exact = orc.two_plane(x, wh, wl, 1 / 256)
    d = [t.cuda() for t in (x, wh, wl)]
    a = hpc.gemm_bf16xfp32(*d, 1 / 256, True, True, None)
    b = hpc.gemm_bf16xfp32(*d, 1 / 256, True, True, None)
    c = hpc.gemm_bf16xfp32(*d, 1 / 256, True, False, None)
    assert torch.equal(a, b)
    assert allclose(exact, a.cpu(), rtol=1e-4, atol=2e-3) and allclose(exact, c.cpu(), rtol=1e-4, atol=2e-3)


@pytest.mark.gpu
def test_gemm_bf16xfp32_checks():
    import hpc

    x = torch.zeros(4, 128, dtype=torch.bfloat16, device="cuda")
    w = torch.zeros(96, 128, dtype=torch.bfloat16, device="cuda")
    with pytest.raises(RuntimeError):
        hpc.gemm_bf16xfp32(x, w, w, 1 / 256)  # n % 64
    w = torch.zeros(64, 128, dtype=torch.bfloat16, device="cuda")
    with pytest.raises(RuntimeError):
        hpc.gemm_bf16xfp32(x.float(), w, w, 1 / 256)


@pytest.mark.dev
@pytest.mark.gpu
@pytest.mark.parametrize("m,n,k", [(304, 192, 4096), (1000, 256, 2048), (4096, 256, 4096), (2100, 128, 7168)])
def test_gemm_bf16xfp32_tile_kernel_equals_the_direct_load_kernel(m, n, k):
    """Round 5: above 256 tokens the LDS-staged tile kernel (64 weight rows x 128 tokens, every operand through LDS) replaced
    the 64 x 64 kernel that loaded operands straight into MFMA layout (development key 40 = 1).  Same k order, same split
    points, same fixed-order reduction: bit-identical at equal split counts (the count is a function of (m, n, k) only)."""
    import hpc
    from utils import dev_set

    x, w = make(m, n, k, seed=7)
    wh, wl = orc.split_weight(w)
    d = [t.cuda() for t in (x, wh, wl)]
    new = hpc.gemm_bf16xfp32(*d, 1 / 256, True, True, None)
    dev_set(40, 1)
    try:
        old = hpc.gemm_bf16xfp32(*d, 1 / 256, True, True, None)
    finally:
        dev_set(40, 0)
    assert torch.equal(new, old)
    assert allclose(orc.two_plane(x, wh, wl, 1 / 256), new.cpu(), rtol=1e-4, atol=2e-3)


def test_gemm_bf16xfp32_split_rule():
    """CPU: `hpc_gemm_bf16xfp32_splits` (callers size the split workspace from it).  Above 256 tokens: tiles of 64 weight rows
    x 128 tokens, splits until there is one workgroup per CU (256), at most 8, every split keeps >= 256 of k; a launch with
    one to two workgroups per CU unsplit is split once; no split-K -> 1."""
    import ctypes
    from pathlib import Path

    lib = ctypes.CDLL(str(Path(__file__).resolve().parent.parent / "hpc-ops_amd" / "hpc" / "libhpc_amd.so"))
    f = lib.hpc_gemm_bf16xfp32_splits
    f.argtypes = [ctypes.c_int] * 4
    f.restype = ctypes.c_int
    assert f(4096, 256, 4096, 0) == 1
    assert f(4096, 256, 4096, 1) == 2      # 128 tiles
    assert f(2048, 256, 4096, 1) == 4      # 64 tiles
    assert f(1024, 256, 4096, 1) == 8      # 32 tiles: the cap
    assert f(304, 256, 4096, 1) == 8       # 12 tiles: the cap, not 16
    assert f(8192, 256, 7168, 1) == 2      # 256 tiles: one more split for two workgroups per CU
    assert f(16384, 256, 4096, 1) == 1     # 512 tiles
    assert f(4096, 256, 512, 1) == 2       # every split keeps 256 of k ...
    assert f(4096, 256, 256, 1) == 1       # ... or there is none
    for m in (1, 16, 64, 256):             # the skinny kernels keep their own rule
        assert 1 <= f(m, 256, 4096, 1) <= 16
