import sys
sys.path.insert(0, "hpc-ops_amd"); sys.path.insert(0, ".")
import torch, hpc, bench
for mode in (2, 1, 2, 1):
    hpc._C.lib.hpc_tuning_set(7, mode)
    print("mapping", "by_position" if mode == 2 else "by_head", bench.extra_prefill(torch.device("cuda:0"), hpc), flush=True)
