"""Test helpers; `allclose` has the contract of reference tests/utils.py:176-189
(dtype/device/shape asserts + torch.allclose in fp32, worst offenders printed on failure)."""
import pytest
import torch


def allclose(ref_tensor, real_tensor, atol=1e-8, rtol=1e-5):
    assert ref_tensor.dtype == real_tensor.dtype, (ref_tensor.dtype, real_tensor.dtype)
    assert ref_tensor.device == real_tensor.device, (ref_tensor.device, real_tensor.device)
    assert ref_tensor.shape == real_tensor.shape, (ref_tensor.shape, real_tensor.shape)
    a, b = ref_tensor.to(torch.float32), real_tensor.to(torch.float32)
    ok = torch.allclose(a, b, atol=atol, rtol=rtol)
    if not ok:
        err = (a - b).abs()
        bad = err > (atol + rtol * a.abs())
        flat = torch.where(bad.reshape(-1))[0]
        top = torch.topk(err.reshape(-1), min(10, err.numel())).indices
        print(f"\nallclose FAILED: {int(bad.sum())}/{a.numel()} elements out of tolerance "
              f"(atol={atol}, rtol={rtol}); max abs err {float(err.max()):.6g}")
        for i in top.tolist():
            idx = tuple(int(v) for v in torch.unravel_index(torch.tensor(i), a.shape))
            print(f"  {idx}: ref={float(a.reshape(-1)[i]):.6g} real={float(b.reshape(-1)[i]):.6g}")
        if flat.numel():
            pass
    return ok


def to_cpu(*ts):
    return [t.cpu() if t is not None else None for t in ts]


def moe_allclose(ref, real, rtol=0.01, atol=0.01, max_literal_outliers=0.005, max_rel_rms=5e-3, max_abs_frac=0.02):
    """Tolerance of the fused-MoE parity checks at hidden sizes beyond the reference test's (512).

    The reference bar is rtol = atol = 0.01 at hidden 512 / ffn <= 512 (tests/test_fuse_moe_blockwise.py:268,350),
    where outputs are O(1).  With the same generator at hidden 4096 / ffn 11008 every expert contribution is
    O(10-30) and is rounded to bf16 (ulp 0.06-0.25) after each GEMM; a bf16 rounding of a gate_up element that
    flips with the fp32 summation order can move one e4m3 code of the quantised activation (a 6 % step of that
    element), so a small fraction of the outputs differs by a few bf16 ulps whatever the kernel does.  The bar:
      * at least 1 - max_literal_outliers (99.5 %) of the elements meet the reference's literal (0.01, 0.01);
      * the relative RMS error of every row is <= max_rel_rms;
      * no element is off by more than max_abs_frac of the largest |ref| (a wrong expert / scale / row is O(1))."""
    a, b = ref.float(), real.float()
    err = (a - b).abs()
    literal_miss = float((err > atol + rtol * a.abs()).float().mean())
    rel_rms = float((err.pow(2).mean(-1).sqrt() / a.pow(2).mean(-1).sqrt().clamp_min(1e-3)).max())
    worst = float(err.max() / a.abs().max().clamp_min(1e-3))
    ok = literal_miss <= max_literal_outliers and rel_rms <= max_rel_rms and worst <= max_abs_frac
    if not ok:
        print(f"\nmoe_allclose FAILED: max abs err {float(err.max()):.4g} ({worst:.4g} of max |ref|), worst row relative "
              f"rms {rel_rms:.4g}, fraction outside the literal (0.01, 0.01) bar {literal_miss:.5f}")
    return ok


def dev_set(key, value):
    """Set a development register (csrc/hpc_dev.h).  They exist only in the development build of the library
    (`HPC_AMD_DEV=1` -> hpc/libhpc_amd_dev.so); the product has none, so against the product a non-zero value SKIPS
    the test case - tests/test_dev_build.py re-runs every test marked `dev` in a subprocess with HPC_AMD_DEV=1 - and
    a zero (= the shipped configuration) is a no-op."""
    import hpc

    if hpc._C.DEV_BUILD:
        assert hpc._C.lib.hpc_dev_tuning_set(key, value) == 0
    elif value != 0:
        pytest.skip("pins a kernel variant through a development register: runs against the development build "
                    "(tests/test_dev_build.py)")
