"""The C++ host side (hpc-ops_amd/csrc/torch_binding.cpp -> hpc/_hpc_torch.so): TORCH_LIBRARY_FRAGMENT(hpc)
registrations of the hot-path ops and the MulticastCommunicator torch class, mirroring reference
src/attention/entry.cc:822-874, src/fuse_moe/entry.cc:644-684, src/normalization/entry.cc:59-65 and
src/communicator/entry.cc:79-90.  The Python entries (hpc/_entry_*.py) are the fallback: both must give the same
results and raise the same errors."""
import math
import multiprocessing
import os
import subprocess
import sys
from pathlib import Path

import pytest
import torch

ROOT = Path(__file__).resolve().parent.parent
SHIM = ROOT / "hpc-ops_amd" / "hpc" / "_hpc_torch.so"
pytestmark = pytest.mark.skipif(not SHIM.exists(), reason="hpc/_hpc_torch.so not built (python hpc-ops_amd/build.py)")

NATIVE = {"assign_attention_decode_task", "attention_decode_bf16", "attention_decode_fp8", "fuse_moe_blockwise_fp8",
          "fuse_moe_blockwise", "fused_rmsnorm_with_scale"}


def test_shim_registers_ops_and_class():
    import hpc

    assert set(hpc._C.NATIVE_OPS) == NATIVE
    for name in NATIVE:
        assert hasattr(torch.ops.hpc, name)
    # schemas verbatim (reference src/attention/entry.cc:851-873)
    s = str(torch.ops.hpc.attention_decode_fp8.default._schema)
    assert "!" in s.split("kcache")[0].rsplit("Tensor", 1)[1] and "int quant_type" in s and "Tensor? task_map" in s
    cls = torch.classes.hpc.MulticastCommunicator
    c = cls(0, 1, -1, f"t_shim_{os.getpid()}")
    assert (c.GetRank(), c.GetWorldSize(), c.GetDeviceId()) == (0, 1, -1)
    c.Barrier()


def _class_worker(rank, world, name, q):
    try:
        sys.path.insert(0, str(ROOT / "hpc-ops_amd"))
        import hpc  # noqa: F401  (loads the shim)

        c = torch.classes.hpc.MulticastCommunicator(rank, world, -1, name)
        for _ in range(3):
            c.Barrier()
        q.put((rank, c.GetRank(), c.GetWorldSize()))
    except Exception as e:  # noqa: BLE001
        q.put((rank, repr(e), -1))


def test_torch_class_rendezvous_two_processes():
    """host-only rendezvous (no GPU) of two processes through torch.classes.hpc.MulticastCommunicator"""
    ctx = multiprocessing.get_context("spawn")
    q = ctx.Queue()
    name = f"t_shimcls_{os.getpid()}"
    ps = [ctx.Process(target=_class_worker, args=(r, 2, name, q)) for r in range(2)]
    for p in ps:
        p.start()
    res = sorted(q.get(timeout=120) for _ in ps)
    for p in ps:
        p.join(timeout=30)
    assert res == [(0, 0, 2), (1, 1, 2)], res


def test_python_entries_are_the_fallback():
    """HPC_AMD_PY_ENTRIES=1: the package imports without the shim and defines the same ops in Python"""
    code = ("import sys; sys.path.insert(0, %r); import torch, hpc; "
            "assert not hpc._C.NATIVE_OPS; assert hasattr(torch.ops.hpc, 'attention_decode_fp8'); print('ok')"
            % str(ROOT / "hpc-ops_amd"))
    r = subprocess.run([sys.executable, "-c", code], env=dict(os.environ, HPC_AMD_PY_ENTRIES="1"), capture_output=True,
                       text=True, timeout=300)
    assert r.returncode == 0 and "ok" in r.stdout, r.stderr[-2000:]


_GPU_CASE = r"""
import sys, math, torch
sys.path.insert(0, %(pkg)r); sys.path.insert(0, %(root)r)
import hpc
torch.manual_seed(7)
dev = torch.device("cuda", 0)
B, Hkv, Hq, D, P = 9, 4, 32, 128, 64
lens = torch.tensor([900, 1, 63, 2500, 64, 130, 7000, 33, 512], dtype=torch.int32)
nb = (lens + P - 1) // P
nblk = int(nb.sum()) + 4
qb = torch.randn(B, Hq, D, dtype=torch.bfloat16) / math.sqrt(D)
qs = qb.float().abs().max(-1)[0] / 10
q8 = (qb / qs[:, :, None]).to(torch.float8_e4m3fn)
kv = (torch.randn(nblk, 2, P, Hkv, D, dtype=torch.bfloat16) / math.sqrt(D)).to(torch.float8_e4m3fn).to(dev)
perm = torch.randperm(nblk).to(torch.int32)
bid = torch.zeros(B, int(nb.max()), dtype=torch.int32)
o = 0
for i, n in enumerate(nb.tolist()):
    bid[i, :n] = perm[o:o + n]; o += n
lens_d = lens.to(dev)
tm = hpc.get_attention_decode_task_workspace(B, 8192, Hkv, 64)
hpc.assign_attention_decode_task(lens_d, tm, Hkv, 1, True, 64)
y8 = hpc.attention_decode_fp8(q8.to(dev), kv[:, 0], kv[:, 1], bid.to(dev), lens_d, qs.to(dev), torch.tensor([0.7], device=dev),
                              torch.tensor([1.3], device=dev), mtp=0, new_kv_included=True,
                              quant_type=hpc.QuantType.QPERTOKEN_PERHEAD_KPERTENSOR_VPERTENSOR, splitk=True, task_map=tm)
kvb = torch.randn(nblk, 2, P, Hkv, D, dtype=torch.bfloat16, device=dev)
yb = hpc.attention_decode_bf16(qb.to(dev), kvb[:, 0], kvb[:, 1], bid.to(dev), lens_d, mtp=0, new_kv_included=True)  # no task map: on-the-fly schedule
x = torch.randn(64, 4096, dtype=torch.bfloat16, device=dev)
w = torch.rand(4096, dtype=torch.bfloat16, device=dev)
n8, nf, n2 = torch.ops.hpc.fused_rmsnorm_with_scale(x, w, torch.tensor([2.5, 5.0], device=dev), 1e-6, True)
T, k, E, H, I = 40, 4, 16, 256, 128
ids = torch.sort(torch.multinomial(torch.ones(T, E), k).to(torch.int32), dim=1)[0]
sc = torch.rand(T, k); sc = sc / sc.sum(1, keepdim=True)
f8 = torch.float8_e4m3fn
xm, xs = (torch.randn(T, H) / 100).to(f8), torch.rand(T, H // 128) + 0.5
guw, guws = torch.randn(E, 2 * I, H).to(f8), torch.rand(E, 2 * I // 128, 4) + 0.5
dw, dws = torch.randn(E, H, I).to(f8), torch.rand(E, H // 128, 4) + 0.5
ym = hpc.fuse_moe_blockwise_fp8(xm.to(dev), xs.to(dev), guw.to(dev), guws.to(dev), dw.to(dev), dws.to(dev), ids.to(dev), sc.to(dev), 0, E)
errs = []
for bad in (lambda: hpc.attention_decode_fp8(q8.to(dev).view(torch.uint8).to(torch.bfloat16), kv[:, 0], kv[:, 1], bid.to(dev), lens_d, qs.to(dev),
                                             torch.tensor([0.7], device=dev), torch.tensor([1.3], device=dev), mtp=0, new_kv_included=True,
                                             quant_type=hpc.QuantType.QPERTOKEN_PERHEAD_KPERTENSOR_VPERTENSOR, splitk=True, task_map=tm),
            lambda: hpc.attention_decode_fp8(q8.to(dev), kv[:, 0], kv[:, 1], bid.to(dev), lens_d, qs.to(dev), torch.tensor([0.7], device=dev),
                                             torch.tensor([1.3], device=dev), mtp=2, new_kv_included=True,
                                             quant_type=hpc.QuantType.QPERTOKEN_PERHEAD_KPERTENSOR_VPERTENSOR, splitk=True, task_map=tm)):
    try:
        bad(); errs.append("no error")
    except RuntimeError as e:
        errs.append(str(e).splitlines()[0][:60])
torch.cuda.synchronize()
torch.save({"native": sorted(hpc._C.NATIVE_OPS), "y8": y8.cpu(), "yb": yb.cpu(), "n8": n8.cpu().view(torch.uint8), "nf": nf.cpu(),
            "n2": n2.cpu().view(torch.uint8), "ym": ym.cpu(), "tm": tm.cpu(), "errs": errs}, sys.argv[1])
"""


@pytest.mark.gpu
def test_native_and_python_entries_agree(tmp_path):
    """the same seeded decode / RMSNorm / fused-MoE calls through the C++ registrations and through the Python entries:
    identical outputs (same kernels behind the same C-ABI) and the same error texts"""
    outs = {}
    for mode in ("native", "python"):
        path = tmp_path / f"{mode}.pt"
        env = dict(os.environ, HPC_AMD_PY_ENTRIES="1" if mode == "python" else "0")
        code = _GPU_CASE % {"pkg": str(ROOT / "hpc-ops_amd"), "root": str(ROOT)}
        r = subprocess.run([sys.executable, "-c", code, str(path)], env=env, capture_output=True, text=True, timeout=600)
        assert r.returncode == 0, r.stderr[-3000:]
        outs[mode] = torch.load(path)
    assert set(outs["native"]["native"]) == NATIVE and outs["python"]["native"] == []
    for k in ("y8", "yb", "n8", "nf", "n2", "ym", "tm"):
        assert torch.equal(outs["native"][k], outs["python"][k]), k
    assert outs["native"]["errs"] == outs["python"]["errs"], (outs["native"]["errs"], outs["python"]["errs"])
    assert "no error" not in outs["native"]["errs"]
