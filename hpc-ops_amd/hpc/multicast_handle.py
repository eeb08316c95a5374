"""hpc.multicast_handle — MulticastHandle (reference hpc/multicast_handle.py:7-200), same public
attributes; peers are address-only (PeerBuffer) because nothing at the Python level ever touches
their contents."""
from itertools import accumulate
from operator import mul
from typing import Any, Optional, Sequence, Tuple

import torch


class MulticastHandle:
    def __init__(self, multicomm, size: Tuple[int], dtype: torch.dtype = None):
        self.rank_ = multicomm.GetRank()
        self.world_size_ = multicomm.GetWorldSize()
        numel = list(accumulate(size, func=mul))[-1]
        buffer_size = numel * dtype.itemsize
        self.buffer_size_ = buffer_size
        signal_offset = (buffer_size + 15) // 16 * 16
        # reference: 72 (max P2P domain) * SM count * 4 bytes of per-(block, peer) flags
        max_num_blocks = torch.cuda.get_device_properties(multicomm.GetDeviceId()).multi_processor_count
        self.signal_size_ = 72 * max_num_blocks * 4
        total_size = signal_offset + self.signal_size_
        self.org_buffer_dict_ = multicomm.CreateTensorSync(total_size)
        self.org_buffer_dict_[self.rank][:] = 0
        self.data_buffer_list_ = [self.org_buffer_dict_[i][: self.buffer_size_] for i in range(self.world_size)]
        self.multimem_data_buffer_ = self.org_buffer_dict_[-1][: self.buffer_size_]
        self.signal_buffer_list_ = [self.org_buffer_dict_[i][signal_offset:] for i in range(self.world_size)]
        self.multimem_signal_buffer_ = self.org_buffer_dict_[-1][signal_offset:]
        self.data_buffer_ptrs_ = torch.empty(self.world_size, dtype=torch.int64, device="cpu")
        self.signal_buffer_ptrs_ = torch.empty(self.world_size, dtype=torch.int64, device="cpu")
        for i in range(self.world_size):
            self.data_buffer_ptrs_[i] = self.data_buffer_list_[i].data_ptr()
            self.signal_buffer_ptrs_[i] = self.signal_buffer_list_[i].data_ptr()
        device = self.org_buffer_dict_[self.rank].device
        self.data_buffer_ptrs_dev_ = self.data_buffer_ptrs_.to(device=device)
        self.signal_buffer_ptrs_dev_ = self.signal_buffer_ptrs_.to(device=device)
        torch.cuda.synchronize(device)
        multicomm.Barrier()  # every rank's pad is zeroed before anyone posts a flag

    @property
    def rank(self) -> int:
        return self.rank_

    @property
    def world_size(self) -> int:
        return self.world_size_

    @property
    def buffer_size(self) -> int:
        return self.buffer_size_

    @property
    def signal_size(self) -> int:
        return self.signal_size_

    @property
    def data_buffer_ptrs(self) -> torch.Tensor:
        return self.data_buffer_ptrs_

    @property
    def signal_buffer_ptrs(self) -> torch.Tensor:
        return self.signal_buffer_ptrs_

    @property
    def data_buffer_ptrs_dev(self) -> torch.Tensor:
        return self.data_buffer_ptrs_dev_

    @property
    def signal_buffer_ptrs_dev(self) -> torch.Tensor:
        return self.signal_buffer_ptrs_dev_

    def _view(self, base, sizes, dtype, storage_offset, limit=None):
        if len(sizes) == 1 and isinstance(sizes[0], Sequence):
            sizes = tuple(sizes[0])
        else:
            sizes = tuple(sizes)
        if dtype is None:
            dtype = torch.get_default_dtype()
        limit = self.buffer_size if limit is None else limit
        numel = list(accumulate(sizes, func=mul))[-1]
        ask = numel * dtype.itemsize
        assert storage_offset + ask <= limit, (
            f"The requested buffer size(got {storage_offset} + {ask} = {storage_offset + ask}) exceeds the size of the "
            f"hold buffer(got {limit}).")
        return base[storage_offset : storage_offset + ask].view(dtype).view(sizes)

    def get_buffer(self, rank: int, *sizes: Any, dtype: Optional[torch.dtype] = None,
                   storage_offset: int = 0) -> torch.Tensor:
        """View of rank `rank`'s data buffer; only the local rank's buffer is a torch tensor."""
        assert 0 <= rank <= self.world_size
        assert rank == self.rank, "peer buffers are address-only on xGMI (use data_buffer_ptrs)"
        return self._view(self.data_buffer_list_[rank], sizes, dtype, storage_offset)

    def get_signal(self, rank: int, *sizes: Any, storage_offset: int = 0) -> torch.Tensor:
        """uint32 view of rank `rank`'s signal pad (reference hpc/multicast_handle.py:128-149); like get_buffer only the
        local rank's pad is a torch tensor here (peers' pads are reached through signal_buffer_ptrs)."""
        assert 0 <= rank <= self.world_size
        assert rank == self.rank, "peer signal pads are address-only on xGMI (use signal_buffer_ptrs)"
        return self._view(self.signal_buffer_list_[rank], sizes, torch.uint32, storage_offset, self.signal_size)

    def get_multimem_buff(self, *sizes: Any, dtype: Optional[torch.dtype] = None,
                          storage_offset: int = 0) -> torch.Tensor:
        """The reference returns a view of the NVLS multicast mapping; xGMI has none, so this is the
        same view of the local buffer - the all-reduce entries translate it to peer addresses."""
        return self._view(self.multimem_data_buffer_, sizes, dtype, storage_offset)

    def get_multimem_signal(self, *sizes: Any, storage_offset: int = 0) -> torch.Tensor:
        """uint32 view of the "multicast" signal pad (reference :173-194): the local pad, see get_multimem_buff."""
        return self._view(self.multimem_signal_buffer_, sizes, torch.uint32, storage_offset, self.signal_size)

    def barrier(self, channel: int = 0, timeout_ms: int = 0):
        """Reference :196-198: a placeholder that launches nothing (`pass`); same signature, same (no) effect.  Ranks
        synchronise through MulticastCommunicator.Barrier() and the signal-pad barriers inside the fused kernels."""
        return None
