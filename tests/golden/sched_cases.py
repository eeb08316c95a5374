"""Scheduler test cases shared by make_golden.py and the tests (inputs only, deterministic)."""
import numpy as np


def cases():
    """yield (name, lens int32[B], bins, num_head_kv, num_seq_q, new_kv_included, min_process_len)"""
    out = []
    # the reference's own test grid (tests/test_attention_decode_bf16.py:206-217): lens =
    # randint(1, max_seq) + num_seq_q, new_kv_included=True, min_process_len=64
    for B in (1, 16, 200):
        for sq in (1, 2, 3):
            for max_seq in (1024, 4096):
                for hkv in (1, 4):
                    rng = np.random.default_rng(41 + B * 1000 + sq * 100 + max_seq + hkv)
                    lens = rng.integers(1, max_seq, B).astype(np.int32) + sq
                    for bins in (512, 312):  # MI355X Sq<=2 choice; H20 78 SMs x 4
                        out.append((f"grid_B{B}_sq{sq}_s{max_seq}_h{hkv}_bins{bins}", lens, bins, hkv,
                                    sq, True, 64))
    # reference benchmark cases (benchmark/attention_decode/bench_attention_decode_fp8.py:57-67)
    bench = {
        "skewed_mix": [128] * 32 + [4096] * 32,
        "skewed_extreme": [64] * 15 + [16384],
        "two_32k_30x4k": [32768] * 2 + [4096] * 30,
        "one_64k_31x4k": [65536] + [4096] * 31,
        "uniform_8k_x64": [8192] * 64,
    }
    for name, lens in bench.items():
        for hkv, bins in ((1, 512), (8, 512), (8, 256)):
            out.append((f"bench_{name}_h{hkv}_bins{bins}", np.array(lens, np.int32), bins, hkv, 1,
                        True, 512))
    # edge cases: empty requests, exact tile multiples, Q rows straddling a chunk boundary
    # (L % 64 < Sq on the last chunk -> overflow fix-up), new_kv_included False, tiny bins
    out.append(("edge_zeros", np.array([0, 5, 0, 64, 0, 129, 0], np.int32), 8, 2, 1, True, 64))
    out.append(("edge_all_zero", np.array([0, 0, 0], np.int32), 4, 2, 1, True, 64))
    out.append(("edge_exact", np.array([64, 128, 192, 256], np.int32), 5, 3, 2, True, 64))
    out.append(("edge_overflow", np.array([130, 65, 129, 193, 66, 257, 1025], np.int32), 16, 1, 3,
                True, 64))
    out.append(("edge_overflow_nokv", np.array([127, 62, 126, 190, 63, 254], np.int32), 12, 2, 3,
                False, 64))
    out.append(("edge_one_bin", np.array([1000, 3], np.int32), 1, 2, 1, True, 64))
    out.append(("edge_more_bins_than_tiles", np.array([70, 3, 200], np.int32), 64, 1, 4, True, 64))
    out.append(("edge_minlen", np.array([700, 900, 64, 5000], np.int32), 32, 2, 1, True, 512))
    rng = np.random.default_rng(7)
    for i in range(12):  # randomised overflow hunting: many lens with L % 64 in [1, 5]
        B = int(rng.integers(1, 40))
        lens = (rng.integers(1, 40, B) * 64 + rng.integers(0, 6, B)).astype(np.int32)
        sq = int(rng.integers(1, 6))
        lens = np.maximum(lens, sq)
        out.append((f"rand_overflow_{i}", lens, int(rng.integers(2, 200)), int(rng.integers(1, 9)),
                    sq, True, 64))
    return out
