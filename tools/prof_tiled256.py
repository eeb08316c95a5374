"""Development tool: read the per-segment cycle sums of the profiling build (tools/prof_tiled256.sh)."""
import ctypes, sys
from pathlib import Path
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT / "hpc-ops_amd")); sys.path.insert(0, str(ROOT))
import torch, hpc
from hpc import _C
dev = torch.device("cuda", 0)
F8 = torch.float8_e4m3fn
E, k, H, I, T = 64, 8, 4096, 11008, 4096
torch.manual_seed(41)
guw = torch.randint(-80, 80, (E, 2 * I, H), dtype=torch.int8, device=dev).view(F8)
dw = torch.randint(-80, 80, (E, H, I), dtype=torch.int8, device=dev).view(F8)
ids = torch.sort(torch.multinomial(torch.ones(T, E, device=dev), k).to(torch.int32), dim=1)[0]
sc = torch.rand(T, k, device=dev)
x = (torch.randn(T, H, device=dev) / 100).to(F8)
guws = torch.rand(E, 2 * I // 128, (H // 128 + 3) // 4 * 4, device=dev) * 0.02
dws = torch.rand(E, H // 128, (I // 128 + 3) // 4 * 4, device=dev) * 0.02
xs = torch.rand(T, H // 128, device=dev)
gus, ds, ams = torch.rand(E, device=dev) * 0.01, torch.rand(E, device=dev) * 0.01, torch.ones(1, device=dev)
buf = (ctypes.c_ulonglong * 16)()
fn = _C.lib.hpc_debug_tiled256_prof
for name, call in (("pertensor", lambda: hpc.fuse_moe_pertensor_fp8(x, guw, dw, gus, ds, ams, ids, sc, 0, E)),
                   ("blockwise", lambda: hpc.fuse_moe_blockwise_fp8(x, xs, guw, guws, dw, dws, ids, sc, 0, E))):
    call(); fn(buf, 1)
    for _ in range(3): call()
    fn(buf, 1)
    for w, o in (("wave0", 0), ("wave7", 8)):
        n = max(buf[o + 4], 1)
        seg = [buf[o + i] / n for i in range(4)]
        print(f"prof {name} {w}: k-steps {n}  first-half {seg[0]:.0f}  waitcnt {seg[1]:.0f}  barrier {seg[2]:.0f}  second-half {seg[3]:.0f}  sum {sum(seg):.0f} (s_memtime ticks per k-step)")
