"""Regenerates tests/golden/*.npz.  Runs ONLY in the build container (needs /root/reference).

Scheduler: runs the REFERENCE's own assign_attention_decode_task_sync (compiled from
/root/reference/src/attention/decode/assign_task.cu by oracle/Makefile into oracle/_ref/) on every
case of sched_cases.py, checks our C restatement (oracle/sched_oracle.c) against it, and stores
the reference output (pad ints masked) - full image for small maps, SHA-256 for all.
"""
import hashlib
import sys
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parents[2]
sys.path.insert(0, str(ROOT))
sys.path.insert(0, str(ROOT / "tests" / "golden"))

from oracle import sched  # noqa: E402
from sched_cases import cases  # noqa: E402


def main():
    assert sched.have_ref(), "run `make -C oracle` with /root/reference present first"
    store = {}
    for name, lens, bins, hkv, sq, nkv, minlen in cases():
        ref = sched.mask_pad(sched.task_map_ref(lens, bins, hkv, sq, nkv, minlen), bins)
        ora = sched.task_map_oracle(lens, bins, hkv, sq, nkv, minlen)
        assert np.array_equal(ref, ora), f"oracle != reference for {name}"
        store[name + "__sha"] = np.frombuffer(hashlib.sha256(ref.tobytes()).digest(), np.uint8)
        if ref.size <= 12 * 600:
            store[name + "__map"] = ref
    np.savez_compressed(ROOT / "tests" / "golden" / "sched_golden.npz", **store)
    print("wrote sched_golden.npz:", len(store), "entries")


if __name__ == "__main__":
    main()
