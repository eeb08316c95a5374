"""Development tool (GPU box): tools/probes/probe_mfma_war.hip - a VALU write into a source register of the
v_mfma_f32_16x16x128_f8f6f4 issued D wait states before it, with S MFMAs of backlog in the matrix pipe: how often does
the MFMA's result change?  usage: python tools/probe_mfma_war.py [out.txt]"""
import ctypes, subprocess, sys
from pathlib import Path
import torch
ROOT = Path(__file__).resolve().parent.parent
bindir = ROOT / "tools" / "probes" / "bin"
bindir.mkdir(exist_ok=True)
so, src = bindir / "libprobe_mfma_war.so", ROOT / "tools" / "probes" / "probe_mfma_war.hip"
if not so.exists() or so.stat().st_mtime < src.stat().st_mtime:
    subprocess.check_call(["hipcc", "--offload-arch=gfx950", "-O2", "-shared", "-fPIC", str(src), "-o", str(so)])
lib = ctypes.CDLL(str(so))
lib.war_probe_launch.restype = ctypes.c_int
lib.war_probe_launch.argtypes = [ctypes.c_void_p] + [ctypes.c_int] * 4 + [ctypes.c_void_p]
B, W, R = [0, 2, 4, 6, 8, 10], [0, 1, 2, 4], ["B v0", "B v1", "A v0"]
lines = []
def say(s=""):
    print(s, flush=True); lines.append(s)
ITERS, WGS = 200, 256
for threads in (256, 512):
    for ww in (0, 1):
        out = torch.zeros(len(B) * len(W) * len(R), dtype=torch.int32, device="cuda")
        rc = lib.war_probe_launch(out.data_ptr(), WGS, threads, ITERS, ww, None)
        torch.cuda.synchronize(); assert rc == 0
        o = out.cpu().view(len(B), len(W), len(R))
        total = WGS * threads * ITERS * 4
        say(f"## {threads // 64 // 4} wave(s) per SIMD, {'WITH' if ww else 'without'} the write: changed result registers of {total} per cell")
        say("    gap  wait | " + " | ".join(f"{r:>9}" for r in R))
        say(f"cells with a changed result: {int((o != 0).sum())} of {o.numel()}")
        for bi, b in enumerate(B):
            for wi, w in enumerate(W):
                if ww: say(f"{b:7d} {w:5d} | " + " | ".join(f"{int(o[bi, wi, ri]):9d}" for ri in range(len(R))))
if len(sys.argv) > 1:
    Path(sys.argv[1]).write_text("\n".join(lines) + "\n")
