"""hpc.communicator — MulticastCommunicator (reference hpc/communicator.py + src/communicator/entry.cc:18-90).

Same constructor and methods as the reference torch class (GetRank / GetWorldSize / GetDeviceId /
Barrier / CreateTensorSync), implemented over libhpc_amd.so's socket rendezvous + HIP IPC
(csrc/communicator.cc).  One process per GPU.  There is no multicast object on xGMI: the `-1`
entry returned by CreateTensorSync aliases the local buffer and the kernels resolve peer addresses
through the registry / pointer tables.
"""
import ctypes

import torch

from . import _C


class _DevBuffer:
    """Minimal __cuda_array_interface__ carrier: lets torch alias device memory owned by the
    communicator (uint8, 1-D)."""

    def __init__(self, ptr: int, nbytes: int, owner):
        self._owner = owner
        self.__cuda_array_interface__ = {
            "shape": (nbytes,), "typestr": "|u1", "data": (ptr, False), "version": 3, "strides": None,
        }


class PeerBuffer:
    """Address-only stand-in for a peer's buffer when torch cannot alias it (the peer GPU is not
    visible to this process).  Supports what MulticastHandle needs: data_ptr() and 1-D slicing."""

    def __init__(self, ptr: int, nbytes: int):
        self._ptr, self._n = ptr, nbytes

    def data_ptr(self):
        return self._ptr

    def numel(self):
        return self._n

    def __getitem__(self, s):
        start, stop, step = s.indices(self._n)
        assert step == 1
        return PeerBuffer(self._ptr + start, max(stop - start, 0))


class MulticastCommunicator:
    def __init__(self, rank: int, world_size: int, device_id: int = -1, comm_name: str = "hpc_comm"):
        if device_id is None or device_id < 0:
            device_id = torch.cuda.current_device() if torch.cuda.is_available() else -1
        self._h = _C.lib.hpc_comm_create(int(rank), int(world_size), int(device_id), comm_name.encode())
        if self._h <= 0:
            raise RuntimeError(f"MulticastCommunicator: rendezvous '{comm_name}' failed ({self._h})")
        self._rank, self._world, self._device = int(rank), int(world_size), int(device_id)

    def GetRank(self) -> int:
        return self._rank

    def GetWorldSize(self) -> int:
        return self._world

    def GetDeviceId(self) -> int:
        return self._device

    def Barrier(self) -> None:
        if _C.lib.hpc_comm_barrier(self._h) != 0:
            raise RuntimeError("MulticastCommunicator.Barrier failed")

    def CreateTensorSync(self, size: int):
        """Collective.  Returns {rank -> uint8 buffer of `size` bytes} for every rank plus -1 (the
        reference's multicast view; here an alias of the local buffer)."""
        ptrs = (ctypes.c_void_p * self._world)()
        rc = _C.lib.hpc_comm_create_tensor_sync(self._h, int(size), ptrs)
        if rc != 0:
            raise RuntimeError(f"CreateTensorSync({size}) failed ({rc})")
        out = {}
        dev = torch.device("cuda", self._device)
        for r in range(self._world):
            p = int(ptrs[r])
            if r == self._rank:
                with torch.cuda.device(dev):
                    out[r] = torch.as_tensor(_DevBuffer(p, int(size), self), device=dev)
            else:
                out[r] = PeerBuffer(p, int(size))
        out[-1] = out[self._rank]
        return out

    def close(self):
        if getattr(self, "_h", 0) > 0:
            _C.lib.hpc_comm_destroy(self._h)
            self._h = 0


def lookup_peers(tensor_or_ptr):
    """[address in rank r's buffer for r in range(world)] for an address inside a local symmetric
    buffer, plus this rank; raises if the address is not symmetric memory."""
    p = tensor_or_ptr if isinstance(tensor_or_ptr, int) else tensor_or_ptr.data_ptr()
    ptrs = (ctypes.c_void_p * 64)()
    rank = ctypes.c_int(-1)
    n = _C.lib.hpc_comm_lookup_peers(ctypes.c_void_p(p), ptrs, ctypes.byref(rank))
    if n <= 0:
        raise RuntimeError("tensor is not inside a buffer created by MulticastCommunicator.CreateTensorSync")
    return [int(ptrs[i]) for i in range(n)], rank.value
