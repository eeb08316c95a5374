"""Test helpers; `allclose` has the contract of reference tests/utils.py:176-189
(dtype/device/shape asserts + torch.allclose in fp32, worst offenders printed on failure)."""
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
