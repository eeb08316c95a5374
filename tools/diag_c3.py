"""diagnostic: kernel-level durations of the short / skewed fp8 decode cases (run under rocprofv3)."""
import sys
sys.path.insert(0, "tools"); sys.path.insert(0, "hpc-ops_amd"); sys.path.insert(0, ".")
import torch
import suite
for nm, ll, hkv, hq in (("extreme", [64] * 15 + [16384], 8, 64), ("extreme_h1", [64] * 15 + [16384], 1, 8),
                        ("mix", [128] * 32 + [4096] * 32, 8, 64)):
    suite.decode_case(nm, torch.tensor(ll, dtype=torch.int32), hkv, hq, "NHD", True, 512)
