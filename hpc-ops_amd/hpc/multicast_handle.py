"""hpc.multicast_handle — MulticastHandle (reference hpc/multicast_handle.py:7-200), same public
attributes; peers are address-only (PeerBuffer) because nothing at the Python level ever touches
their contents."""
import math
from typing import Any, Optional, Sequence, Tuple

import torch


class MulticastHandle:
    def __init__(self, multicomm, size: Tuple[int], dtype: torch.dtype = None):
        self._rank = multicomm.GetRank()
        self._nranks = multicomm.GetWorldSize()
        self._data_bytes = math.prod(size) * dtype.itemsize
        # layout of one symmetric allocation per rank: [data | pad to 16 B | signal pad].  The signal pad keeps the reference's
        # size rule (72 = its largest P2P domain, x workgroups, x 4 bytes: one uint32 flag per (workgroup, peer)) so that
        # `signal_size` reads the same; "workgroups" is the CU count here.
        signal_at = -(-self._data_bytes // 16) * 16
        cus = torch.cuda.get_device_properties(multicomm.GetDeviceId()).multi_processor_count
        self._signal_bytes = 72 * cus * 4
        # rank -> uint8 tensor of that rank's allocation (peers: address-only), key -1 -> the "multicast" alias (= local)
        mapped = multicomm.CreateTensorSync(signal_at + self._signal_bytes)
        mapped[self._rank].zero_()
        self._data = {r: mapped[r][: self._data_bytes] for r in (*range(self._nranks), -1)}
        self._signal = {r: mapped[r][signal_at:] for r in (*range(self._nranks), -1)}
        self._mapped = mapped  # keeps the allocations alive
        self._data_ptrs = torch.tensor([self._data[r].data_ptr() for r in range(self._nranks)], dtype=torch.int64)
        self._signal_ptrs = torch.tensor([self._signal[r].data_ptr() for r in range(self._nranks)], dtype=torch.int64)
        device = mapped[self._rank].device
        self._data_ptrs_dev = self._data_ptrs.to(device=device)
        self._signal_ptrs_dev = self._signal_ptrs.to(device=device)
        torch.cuda.synchronize(device)
        multicomm.Barrier()  # every rank's pad is zeroed before anyone posts a flag

    @property
    def rank(self) -> int:
        return self._rank

    @property
    def world_size(self) -> int:
        return self._nranks

    @property
    def buffer_size(self) -> int:
        return self._data_bytes

    @property
    def signal_size(self) -> int:
        return self._signal_bytes

    @property
    def data_buffer_ptrs(self) -> torch.Tensor:
        return self._data_ptrs

    @property
    def signal_buffer_ptrs(self) -> torch.Tensor:
        return self._signal_ptrs

    @property
    def data_buffer_ptrs_dev(self) -> torch.Tensor:
        return self._data_ptrs_dev

    @property
    def signal_buffer_ptrs_dev(self) -> torch.Tensor:
        return self._signal_ptrs_dev

    def _view(self, base, sizes, dtype, storage_offset, limit=None):
        if len(sizes) == 1 and isinstance(sizes[0], Sequence):
            sizes = tuple(sizes[0])
        else:
            sizes = tuple(sizes)
        if dtype is None:
            dtype = torch.get_default_dtype()
        limit = self.buffer_size if limit is None else limit
        want = math.prod(sizes) * dtype.itemsize
        assert storage_offset + want <= limit, (
            f"view of {want} bytes at byte offset {storage_offset} does not fit the {limit}-byte symmetric region")
        return base[storage_offset : storage_offset + want].view(dtype).view(sizes)

    def get_buffer(self, rank: int, *sizes: Any, dtype: Optional[torch.dtype] = None,
                   storage_offset: int = 0) -> torch.Tensor:
        """View of rank `rank`'s data buffer; only the local rank's buffer is a torch tensor."""
        assert 0 <= rank <= self.world_size
        assert rank == self.rank, "peer buffers are address-only on xGMI (use data_buffer_ptrs)"
        return self._view(self._data[rank], sizes, dtype, storage_offset)

    def get_signal(self, rank: int, *sizes: Any, storage_offset: int = 0) -> torch.Tensor:
        """uint32 view of rank `rank`'s signal pad (reference hpc/multicast_handle.py:128-149); like get_buffer only the
        local rank's pad is a torch tensor here (peers' pads are reached through signal_buffer_ptrs)."""
        assert 0 <= rank <= self.world_size
        assert rank == self.rank, "peer signal pads are address-only on xGMI (use signal_buffer_ptrs)"
        return self._view(self._signal[rank], sizes, torch.uint32, storage_offset, self.signal_size)

    def get_multimem_buff(self, *sizes: Any, dtype: Optional[torch.dtype] = None,
                          storage_offset: int = 0) -> torch.Tensor:
        """The reference returns a view of the NVLS multicast mapping; xGMI has none, so this is the
        same view of the local buffer - the all-reduce entries translate it to peer addresses."""
        return self._view(self._data[-1], sizes, dtype, storage_offset)

    def get_multimem_signal(self, *sizes: Any, storage_offset: int = 0) -> torch.Tensor:
        """uint32 view of the "multicast" signal pad (reference :173-194): the local pad, see get_multimem_buff."""
        return self._view(self._signal[-1], sizes, torch.uint32, storage_offset, self.signal_size)

    def barrier(self, channel: int = 0, timeout_ms: int = 0):
        """Reference :196-198: a placeholder that launches nothing (`pass`); same signature, same (no) effect.  Ranks
        synchronise through MulticastCommunicator.Barrier() and the signal-pad barriers inside the fused kernels."""
        return None
