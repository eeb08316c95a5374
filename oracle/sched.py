"""Decode scheduler oracle bindings — TEST INFRASTRUCTURE (see oracle/__init__.py).

`task_map_oracle` runs oracle/sched_oracle.c (our C restatement of reference
src/attention/decode/assign_task.cu:362-492 + src/attention/entry.cc:727-778);
`task_map_ref` runs the REFERENCE's own function compiled by oracle/Makefile into
oracle/_ref/libsched_ref.so (present wherever `make -C oracle` ran with /root/reference mounted;
the prebuilt .so travels to the GPU box).  Both return the host task-map image of the reference CPU
entry as an int32 array [rows, 12].
"""
import ctypes
from pathlib import Path

import numpy as np

_HERE = Path(__file__).resolve().parent
_IP = ctypes.POINTER(ctypes.c_int)


def _load(path):
    if not path.exists():
        raise FileNotFoundError(f"{path} missing: run `make -C oracle` (or __graft_entry__.build())")
    return ctypes.CDLL(str(path))


def _ip(a):
    return a.ctypes.data_as(_IP)


def tiles_per_bin(lens, bins, num_head_kv, num_seq_q, new_kv_included, min_process_len, tilen=64):
    lib = _load(_HERE / "_build" / "libsched_oracle.so")
    lens = np.ascontiguousarray(lens, dtype=np.int32)
    return lib.sched_oracle_tiles_per_bin(_ip(lens), bins, len(lens), num_head_kv, num_seq_q, tilen,
                                          int(new_kv_included), min_process_len)


def task_map_oracle(lens, bins, num_head_kv, num_seq_q, new_kv_included, min_process_len, tilen=64):
    lib = _load(_HERE / "_build" / "libsched_oracle.so")
    lens = np.ascontiguousarray(lens, dtype=np.int32)
    per = tiles_per_bin(lens, bins, num_head_kv, num_seq_q, new_kv_included, min_process_len, tilen)
    rows = lib.sched_oracle_map_rows(per, bins, len(lens), num_head_kv)
    out = np.zeros((rows, 12), dtype=np.int32)
    got = lib.sched_oracle_task_map(_ip(lens), bins, len(lens), num_head_kv, num_seq_q, tilen,
                                    int(new_kv_included), min_process_len, _ip(out))
    assert got == rows
    return out


def have_ref():
    return (_HERE / "_ref" / "libsched_ref.so").exists()


def task_map_ref(lens, bins, num_head_kv, num_seq_q, new_kv_included, min_process_len, tilen=64):
    lib = _load(_HERE / "_ref" / "libsched_ref.so")
    lens = np.ascontiguousarray(lens, dtype=np.int32)
    per = tiles_per_bin(lens, bins, num_head_kv, num_seq_q, new_kv_included, min_process_len, tilen)
    rows = 1 + bins * (per + 1) + (num_head_kv * len(lens) * 4 + 47) // 48
    out = np.zeros((rows, 12), dtype=np.int32)
    got = lib.sched_ref_task_map(_ip(lens), bins, len(lens), num_head_kv, num_seq_q, tilen,
                                 int(new_kv_included), min_process_len, _ip(out), rows)
    assert got == rows, (got, rows)
    return out


def mask_pad(task_map, bins):
    """Zero ints 9..11 of every task record (uninitialised in the reference, see sched_oracle.c)."""
    tm = task_map.copy()
    per1 = int(tm[0, 0])
    tm[1 : 1 + bins * per1, 9:12] = 0
    return tm
