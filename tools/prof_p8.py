"""Development tool: where a k-tile of the 256 x 256 grouped GEMM (csrc/group_gemm_p8.hip) spends its time.
The development build's profiling variants (development key 22 = 1 / 11, Cfg::kProf) log s_memtime stamps at the section
boundaries of every wave of the first 16 work items: A = behind the barrier that opens an MMA section, B = behind its last
MFMA, C = behind the barrier that closes it, D = end of the following load section's issue (in front of its wait).
usage: python tools/prof_p8.py [--half] [key22 ...]      (default: 1 = product loop, 3 = round-4 loop)"""
import os
os.environ.setdefault("HPC_AMD_DEV", "1")
import sys
from pathlib import Path
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT / "hpc-ops_amd")); sys.path.insert(0, str(ROOT))
import ctypes
import torch, bench, hpc
from hpc import _C
dev = torch.device("cuda", 0)
HALF = "--half" in sys.argv
args = [a for a in sys.argv[1:] if not a.startswith("--")]
w = bench.C4
m_ = bench.c4_inputs(dev, w)
E = w["num_expert"]
rows = 128 if HALF else 512
M = E * rows
F8 = torch.float8_e4m3fn
wt, wsc = m_["guw"], m_["guws"]
n, k = wt.shape[1], wt.shape[2]
seqlens = torch.full((E,), rows, dtype=torch.int32)
cu = torch.cat([torch.zeros(1, dtype=torch.int32), torch.cumsum(seqlens, 0).to(torch.int32)])
tile = hpc.aligned_size(rows)
m_pad = E * ((rows + tile - 1) // tile) * tile + 64
x = (torch.randn(M, k, device=dev) / 10).to(F8)
xs_t = torch.rand(k // 128, m_pad, device=dev) + 0.5
out = torch.empty(M, n, dtype=torch.bfloat16, device=dev)
sl, cud = seqlens.to(dev), cu.to(dev)
call = lambda: hpc.group_gemm_blockwise_fp8(x, wt, sl, cud, xs_t, wsc, num_seq_per_group_avg=rows, output=out)
buf = torch.zeros(16 * 8 * 64, dtype=torch.int32, device=dev)
_C.lib.hpc_dev_p8_prof_buffer.argtypes = [ctypes.c_void_p]
if HALF:
    _C.lib.hpc_dev_tuning_set(3, 4)  # always the 256 x 256 kernel: every item is a half tile
for key in [int(a) for a in args] or [1, 3]:
    _C.lib.hpc_dev_tuning_set(22, key)
    _C.lib.hpc_dev_p8_prof_buffer(ctypes.c_void_p(buf.data_ptr()))
    for _ in range(3):
        buf.zero_()
        call()
    torch.cuda.synchronize()
    us = bench.timed(call, iters=10, warm=2, graph=True)
    _C.lib.hpc_dev_p8_prof_buffer(ctypes.c_void_p(0))
    log = buf.cpu().view(16, 8, 16, 4).to(torch.int64) & 0xFFFFFFFF  # [wg][wave][entry][A B C D]
    res = {}
    for grp, waves in (("waves 0-3", range(0, 4)), ("waves 4-7", range(4, 8))):
        acc = {}
        for g in range(16):
            for wv in waves:
                e = log[g, wv]
                for i in range(15):  # entry i: sec = 8 + i; A, B, C of MMA section sec - 1, D of load section sec
                    A, B, C, D = (int(v) for v in e[i])
                    A1 = int(e[i + 1][0])
                    d = lambda a, b: (b - a) & 0xFFFFFFFF  # noqa: E731
                    mma_t = "X" if (8 + i - 1) % 2 == 0 else "Y"
                    ld_t = "X" if (8 + i) % 2 == 0 else "Y"
                    for name, val in ((f"mma{mma_t}", d(A, B)), (f"bar_out{mma_t}", d(B, C)), (f"load{ld_t}_issue", d(C, D)),
                                      (f"load{ld_t}_wait+bar_in", d(D, A1))):
                        if val < 100000:
                            acc.setdefault(name, []).append(val)
        res[grp] = {kk: sum(v) / len(v) for kk, v in acc.items()}
    # raw timeline of workgroup 0: stamps relative to the first one logged, waves 0 and 4 (one per group, same SIMD pairings)
    t0 = min(int(log[0, wv, 0, 0]) for wv in range(8))
    for wv in (0, 4, 1, 5):
        row = " ".join("[" + " ".join(f"{(int(v) - t0) & 0xFFFFFFFF:6d}" for v in log[0, wv, i]) + "]" for i in range(6))
        print(f"   raw wg0 wave{wv} (A B C of the MMA section before | D of this load section) x 6 entries: {row}")
    torch.save(log, str(ROOT / "gpurun_out" / f"r5_p8_prof_raw_{key}{'_half' if HALF else ''}.pt"))
    print(f"[22={key}{' half' if HALF else ''}] {us:8.1f} us  {2.0 * M * n * k / us / 1e6:7.1f} TFLOP/s (profiling build)")
    for grp, r in res.items():
        tot = sum(r.values())
        print(f"   {grp}: " + "  ".join(f"{kk} {v:6.0f}" for kk, v in sorted(r.items())) + f"  | k-tile {tot:6.0f} ticks")
_C.lib.hpc_dev_tuning_set(22, 0)
_C.lib.hpc_dev_tuning_set(3, 0)
