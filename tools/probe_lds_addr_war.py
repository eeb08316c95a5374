"""Development tool (GPU box): tools/probes/probe_lds_addr_war.hip.  usage: python tools/probe_lds_addr_war.py [out.txt]"""
import ctypes, subprocess, sys
from pathlib import Path
import torch
ROOT = Path(__file__).resolve().parent.parent
bindir = ROOT / "tools" / "probes" / "bin"
bindir.mkdir(exist_ok=True)
so, src = bindir / "libprobe_lds_addr_war.so", ROOT / "tools" / "probes" / "probe_lds_addr_war.hip"
if not so.exists() or so.stat().st_mtime < src.stat().st_mtime:
    subprocess.check_call(["hipcc", "--offload-arch=gfx950", "-O2", "-shared", "-fPIC", str(src), "-o", str(so)])
lib = ctypes.CDLL(str(so))
lib.lds_war_launch.restype = ctypes.c_int
lib.lds_war_launch.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_void_p]
F, W = [0, 4, 16], [0, 1, 2, 4, 8, 16]
lines = []
def say(s=""):
    print(s, flush=True); lines.append(s)
ITERS, WGS = 2000, 256
for partner in (0, 1):
    out = torch.zeros(len(F) * len(W) * 4 * 2, dtype=torch.int32, device="cuda")
    rc = lib.lds_war_launch(out.data_ptr(), WGS, ITERS, partner, None)
    torch.cuda.synchronize(); assert rc == 0
    o = out.cpu().view(len(F), len(W), 4, 2)
    say(f"## SIMD partner {'running MFMAs + FMAs at priority 1' if partner else 'idle'}: of {WGS * 4 * 16 * ITERS} trials per cell and lane quarter")
    say("ds_reads queued in front, wait states | register != written value, lanes 0-15 / 16-31 / 32-47 / 48-63 | DS data wrong, same quarters")
    for fi, f in enumerate(F):
        for wi, w in enumerate(W):
            say(f"{f:4d} {w:4d} | " + " ".join(f"{int(o[fi, wi, q, 0]):8d}" for q in range(4)) + " | " + " ".join(f"{int(o[fi, wi, q, 1]):8d}" for q in range(4)))
if len(sys.argv) > 1:
    Path(sys.argv[1]).write_text("\n".join(lines) + "\n")
