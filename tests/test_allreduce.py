"""Fused AllReduce + residual + RMSNorm and the communicator.

* CPU: socket rendezvous / allgather / barrier of the communicator with 2 processes (no GPU).
* GPU: world_size 1 in-process, and world_size 2 with two processes sharing the one GPU of the test
  box (peer access goes through HIP IPC exactly as across xGMI), several consecutive calls so the
  Lamport slots rotate and get cleaned.  Generator / tolerances follow reference
  tests/test_fuse_allreduce_rmsnorm_{low_latency,high_throughput}.py (seed 10001, atol=rtol=0.1).
"""
import ctypes
import math
import multiprocessing
import os
import sys
from pathlib import Path

import pytest
import torch

ROOT = Path(__file__).resolve().parent.parent


def _paths():
    for p in (str(ROOT / "hpc-ops_amd"), str(ROOT), str(ROOT / "tests")):
        if p not in sys.path:
            sys.path.insert(0, p)


def _host_comm_worker(rank, world, name, q):
    try:
        lib = ctypes.CDLL(str(ROOT / "hpc-ops_amd" / "hpc" / "libhpc_amd.so"))
        lib.hpc_comm_create.argtypes = [ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_char_p]
        h = lib.hpc_comm_create(rank, world, -1, name.encode())
        assert h > 0, h
        mine = (ctypes.c_char * 8)(*bytes([rank + 1] * 8))
        out = (ctypes.c_char * (8 * world))()
        lib.hpc_comm_allgather.argtypes = [ctypes.c_int, ctypes.c_void_p, ctypes.c_int64, ctypes.c_void_p]
        assert lib.hpc_comm_allgather(h, mine, 8, out) == 0
        assert bytes(out) == b"".join(bytes([r + 1] * 8) for r in range(world))
        for _ in range(3):
            assert lib.hpc_comm_barrier(h) == 0
        assert lib.hpc_comm_destroy(h) == 0
        q.put((rank, "ok"))
    except Exception as e:  # noqa: BLE001
        q.put((rank, repr(e)))


@pytest.mark.parametrize("world", [1, 2, 4])
@pytest.mark.parametrize("scheme", ["unix", "tcp"])
def test_communicator_rendezvous_host_only(world, scheme):
    ctx = multiprocessing.get_context("spawn")
    q = ctx.Queue()
    name = (f"tcp://127.0.0.1:{29000 + os.getpid() % 2000 + world}" if scheme == "tcp"
            else f"t_host_{os.getpid()}_{world}")
    ps = [ctx.Process(target=_host_comm_worker, args=(r, world, name, q)) for r in range(world)]
    for p in ps:
        p.start()
    res = [q.get(timeout=120) for _ in ps]
    for p in ps:
        p.join(timeout=60)
    assert sorted(res) == [(r, "ok") for r in range(world)], res


def _ar_task(rank, world_size, cases, name, q, device_index):
    try:
        _paths()
        import hpc
        from oracle import allreduce as oar
        from utils import allclose

        dev = torch.device("cuda", device_index)
        torch.cuda.set_device(dev)
        comm = hpc.MulticastCommunicator(rank, world_size, device_index, name)
        for mode, N, H, nblk, iters in cases:
            N_pad = (N + world_size - 1) // world_size * world_size
            if mode == "ll":
                M_pad = 2 * math.ceil(N / world_size) * world_size * 3
                ws_buf, hdl = hpc.empty_multimem(comm, [M_pad, H], dtype=torch.bfloat16, device=dev)
                ws_buf.view(torch.int32).fill_(-(2 ** 31))
                mc = hdl.get_multimem_buff([M_pad, H], dtype=torch.bfloat16)
                slot_bytes = (M_pad * H * 2 // 3) // 16 * 16
                flags = torch.tensor([0, 2, slot_bytes, 0, 0, 0, 0, 0, 0], dtype=torch.int32, device=dev)
            else:
                in_x, in_hdl = hpc.empty_multimem(comm, [N_pad, H], dtype=torch.bfloat16, device=dev)
                out_x, out_hdl = hpc.empty_multimem(comm, [N_pad, H], dtype=torch.bfloat16, device=dev)
            for it in range(iters):
                n_it = N if it % 2 == 0 else max(N - 3, 1)  # vary the row count across calls
                torch.manual_seed(10001 + it)
                inputs = [torch.randn((N_pad, H), dtype=torch.bfloat16) for _ in range(world_size)]
                residual = torch.randn((N_pad, H), dtype=torch.bfloat16)
                weight = torch.randn((H,), dtype=torch.bfloat16)
                if it == 1:  # exercise the -0.0 sentinel path
                    for x in inputs:
                        x[0, :64] = -0.0
                ref_res, ref_out = oar.ref_allreduce_rmsnorm([x[:n_it] for x in inputs], residual[:n_it],
                                                             weight, 1e-6)
                res_d, w_d = residual.to(dev), weight.to(dev)
                comm.Barrier()
                if mode == "ll":
                    x_d = inputs[rank][:n_it].contiguous().to(dev)
                    out = torch.empty_like(x_d)
                    out_res = torch.empty_like(x_d)
                    hpc.fuse_allreduce_rmsnorm_low_latency(
                        x_d, mc, hdl.data_buffer_ptrs_dev, ws_buf, flags, world_size, rank,
                        res_d[:n_it].contiguous(), w_d, 1e-6, nblk, out, out_res, True)
                    torch.cuda.synchronize()
                    assert allclose(ref_res, out_res.cpu(), atol=0.1, rtol=0.1), f"{mode} residual it{it}"
                    assert allclose(ref_out, out.cpu(), atol=0.1, rtol=0.1), f"{mode} output it{it}"
                else:
                    in_x.zero_()
                    in_x[:n_it] = inputs[rank][:n_it].to(dev)
                    out_x.fill_(7.0)
                    out_res = torch.empty_like(res_d)
                    start, end = N_pad // world_size * rank, N_pad // world_size * (rank + 1)
                    off = start * H * 2
                    torch.cuda.synchronize()
                    comm.Barrier()
                    hpc.fuse_allreduce_rmsnorm_high_throughput(
                        in_x[start:end], in_hdl.get_multimem_buff(in_x[start:end].shape, dtype=in_x.dtype,
                                                                  storage_offset=off),
                        res_d[start:end], w_d, 1e-6, in_hdl.signal_buffer_ptrs_dev, rank, world_size, nblk,
                        out_x[start:end], out_hdl.get_multimem_buff(out_x[start:end].shape, dtype=out_x.dtype,
                                                                    storage_offset=off),
                        out_res[start:end])
                    torch.cuda.synchronize()
                    lo, hi = start, min(end, n_it)
                    if hi > lo:
                        assert allclose(ref_res[lo:hi], out_res[lo:hi].cpu(), atol=0.1, rtol=0.1), f"ht res it{it}"
                    assert allclose(ref_out, out_x[:n_it].cpu(), atol=0.1, rtol=0.1), f"ht output it{it}"
                comm.Barrier()
        from hpc import _C
        assert _C.lib.hpc_allreduce_timeouts() == 0, "a bounded spin timed out"
        comm.Barrier()
        q.put((rank, "ok"))
    except Exception as e:  # noqa: BLE001
        import traceback
        q.put((rank, traceback.format_exc() or repr(e)))


CASES = [("ht", 128, 8192, 16, 3), ("ht", 77, 5120, 78, 2), ("ll", 128, 8192, 16, 5), ("ll", 77, 7168, 78, 4),
         ("ll", 8, 4096, 4, 3)]


def _spawn(world_size):
    ctx = multiprocessing.get_context("spawn")
    q = ctx.Queue()
    name = f"t_ar_{os.getpid()}_{world_size}"
    ps = [ctx.Process(target=_ar_task, args=(r, world_size, CASES, name, q, 0)) for r in range(world_size)]
    for p in ps:
        p.start()
    res = []
    for _ in ps:
        res.append(q.get(timeout=600))
    for p in ps:
        p.join(timeout=60)
    assert sorted(res) == [(r, "ok") for r in range(world_size)], res


@pytest.mark.gpu
def test_allreduce_rmsnorm_world1():
    _spawn(1)


@pytest.mark.gpu
def test_allreduce_rmsnorm_world2_shared_gpu():
    """two ranks on the single GPU of the test box: exercises IPC handles, pointer tables, signal
    barriers and the Lamport protocol across processes (the 8-GPU run is the driver's)."""
    _spawn(2)
