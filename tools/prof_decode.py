"""Development tool: where the waves of the FP8 decode kernel (attention_decode_v2.hip) spend their cycles.
Runs the profiling build (per-wave s_memtime sums around the phases of a wave-iteration) on the C3 mix and on
uniform 8k lengths, with and without real KV loads, and prints averages + the spread of finish times.
usage: python tools/prof_decode.py [four_heads] [dump]"""
import os
os.environ.setdefault("HPC_AMD_DEV", "1")  # development build of the library: tuning registers
import ctypes, sys
from pathlib import Path
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT / "hpc-ops_amd")); sys.path.insert(0, str(ROOT))
import torch, bench, hpc
from hpc import _C
dev = torch.device("cuda", 0)
B, D = 64, 128
lib = _C.lib
lib.hpc_dev_decode_prof_buffer.argtypes = [ctypes.c_void_p]
NWG = 512
buf = torch.zeros(NWG * 4 * 12, dtype=torch.int64, device=dev)

def run(name, lens_c, nomem):
    w = dict(bench.C3)
    inp = bench.c3_inputs(dev, w, lens=lens_c)
    tm = hpc.get_attention_decode_task_workspace(B, int(lens_c.max()), 8, 64)
    hpc.assign_attention_decode_task(inp["kv_lens"], tm, 8, 1, True, 64)
    o = torch.empty(B, 64, D, dtype=torch.bfloat16, device=dev)
    call = lambda: hpc.attention_decode_fp8(inp["q"], inp["k_cache"], inp["v_cache"], inp["block_ids"], inp["kv_lens"],
                                            inp["q_scale"], inp["k_scale"], inp["v_scale"], 0, True,
                                            hpc.QuantType.QPERTOKEN_PERHEAD_KPERTENSOR_VPERTENSOR, True, tm, None, o)
    lib.hpc_dev_tuning_set(15, 1 if nomem else 0)
    for _ in range(3): call()
    torch.cuda.synchronize()
    us_plain = bench.timed(call, iters=20, warm=2)
    lib.hpc_dev_decode_prof_buffer(buf.data_ptr())
    buf.zero_()
    for _ in range(2): call()
    torch.cuda.synchronize()
    us_prof = bench.timed(call, iters=10, warm=1)
    lib.hpc_dev_decode_prof_buffer(None)
    lib.hpc_dev_tuning_set(15, 0)
    if DUMP:
        torch.save(buf.cpu().view(NWG * 4, 12).clone(), str(ROOT / "gpurun_out" / f"prof_{name}_nomem{int(nomem)}_map{MAP}.pt"))
    t = buf.cpu().view(NWG * 4, 12).double()
    wg_of = (torch.arange(NWG * 4) // 4)[t[:, 9] > 0]
    raw11 = buf.cpu().view(NWG * 4, 12)[:, 11][t[:, 9] > 0]
    t = t[t[:, 9] > 0]
    tot = t[:, 2]
    names = ["wait_k", "wait_v", "lds_write", "issue+step", "compute", "finish"]
    r0, r1 = t[:, 0], t[:, 1]
    span = (r1.max() - r0.min()) / 100.0  # s_memrealtime: 100 MHz
    print(f"== {name} nomem={int(nomem)}: eager {us_plain:.1f} us, prof build {us_prof:.1f} us, waves {len(t)}, kernel span {span:.1f} us")
    print(f"   per wave: cycles {tot.mean():.0f} (min {tot.min():.0f} max {tot.max():.0f}), WIs {t[:, 9].mean():.1f} (min {t[:, 9].min():.0f} max {t[:, 9].max():.0f}), "
          f"cycles per WI {(tot / t[:, 9]).mean():.0f}")
    acc = 0.0
    for i, nm in enumerate(names):
        f = (t[:, 3 + i] / tot).mean()
        acc += f
        print(f"   {nm:<11} {f:6.3f} of wave cycles   ({(t[:, 3 + i] / t[:, 9]).mean():7.0f} cycles per WI)")
    print(f"   other       {1 - acc:6.3f}  (prologue: plan, first loads; epilogue)")
    start = (r0 - r0.min()) / 100.0
    end = (r1 - r0.min()) / 100.0
    q = torch.tensor([0.0, 0.1, 0.5, 0.9, 0.99, 1.0], dtype=torch.double)
    print("   wave start us (quantiles 0/10/50/90/99/100):", [round(float(x), 1) for x in torch.quantile(start, q)])
    print("   wave end   us (quantiles 0/10/50/90/99/100):", [round(float(x), 1) for x in torch.quantile(end, q)])
    pair_of = raw11 & 0xff
    for half in (0, 1):  # mean end time / us per wave-iteration by grid half (rows) and head pair (columns)
        cells = []
        for p_ in range(4):
            m = ((wg_of >= NWG // 2) == bool(half)) & (pair_of == p_)
            if m.any():
                cells.append(f"{float(end[m].mean()):6.1f}/{float(((end - start)[m] / t[:, 9][m]).mean()):.2f}")
        print(f"   {'second' if half else 'first '} half, pairs 0..3: end us / us per WI:", "  ".join(cells))
    xcc_of = (raw11 >> 32) & 0xf
    cells = []
    for x_ in range(8):  # by XCD: mean end / us per wave-iteration, first- and second-half workgroups
        m1 = (xcc_of == x_) & (wg_of < NWG // 2); m2 = (xcc_of == x_) & (wg_of >= NWG // 2)
        if m1.any() and m2.any():
            prs = sorted(set(int(v) for v in pair_of[xcc_of == x_].tolist()))
            cells.append(f"x{x_}(pairs {prs}) {float(end[m1].mean()):.0f}/{float(((end - start)[m1] / t[:, 9][m1]).mean()):.2f} {float(end[m2].mean()):.0f}/{float(((end - start)[m2] / t[:, 9][m2]).mean()):.2f}")
    print("   by XCD (first half end/us per WI, second half):", "  ".join(cells))
    clk = tot / ((r1 - r0) / 100.0).clamp_min(1e-3)  # shader cycles per us
    print(f"   shader clock seen by the waves: {clk.median():.0f} MHz")

DUMP = "dump" in sys.argv
MAP = 2 if "four_heads" in sys.argv else 0   # "four_heads": the four-head form (key 29 = 2) instead of head pairs
lib.hpc_dev_tuning_set(29, MAP)
mixed = bench.c3_lens()
ONLY = [a_[5:] for a_ in sys.argv if a_.startswith("only=")]
for nm, lens in (("mixed", mixed), ("uniform8k", torch.full((B,), 8192, dtype=torch.int32))):
    if ONLY and nm not in ONLY: continue
    for nomem in ((False,) if DUMP else (False, True)):
        run(nm, lens, nomem)
