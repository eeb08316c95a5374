"""Development tool: offline look at the per-wave dumps of tools/prof_decode.py (gpurun_out/prof_*.pt):
finish times by XCD, head pair, SE/CU, and by position of the range."""
import sys, torch
for path in sys.argv[1:]:
    t = torch.load(path).to(torch.int64)
    t = t[t[:, 9] > 0]
    r0, r1 = t[:, 0].double(), t[:, 1].double()
    base = r0.min()
    end = (r1 - base) / 100.0
    start = (r0 - base) / 100.0
    hw = t[:, 10]
    meta = t[:, 11]
    xcc = (meta >> 32) & 0xf
    rng = (meta >> 8) & 0xffffff
    pr = meta & 0xff
    cu = (hw >> 8) & 0xf
    sh = (hw >> 12) & 0x1
    se = (hw >> 13) & 0x7
    simd = (hw >> 4) & 0x3
    print(f"== {path}: {len(t)} waves, end min {end.min():.1f} med {end.median():.1f} max {end.max():.1f}")
    def by(name, key):
        ks = sorted(set(key.tolist()))
        if len(ks) > 40:
            return
        print(f"  by {name}: " + "  ".join(f"{k}:{end[key == k].mean():.0f}/{end[key == k].max():.0f}({int((key == k).sum())})" for k in ks))
    by("xcc", xcc); by("pair", pr); by("se", se); by("cu", cu); by("simd", simd); by("rng%8", rng % 8)
    # range position: first / last ranges
    nr = int(rng.max()) + 1
    q = (rng * 8 // nr)
    by("range octile", q)
    # correlation of end with start
    print("  corr(end, start) = %.3f" % float(torch.corrcoef(torch.stack([end, start]))[0, 1]))
    wis = t[:, 9].double()
    print("  corr(end, WIs) = %.3f" % float(torch.corrcoef(torch.stack([end, wis]))[0, 1]))
    # same-CU partner effects: group by (xcc, se, sh, cu)
    cuid = ((xcc * 8 + se) * 2 + sh) * 16 + cu
    ids = sorted(set(cuid.tolist()))
    print(f"  distinct CUs seen: {len(ids)}; waves per CU min/max: {min(int((cuid == i).sum()) for i in ids)}/{max(int((cuid == i).sum()) for i in ids)}")
    m = torch.tensor([float(end[cuid == i].mean()) for i in ids])
    print(f"  per-CU mean end: min {m.min():.1f} med {m.median():.1f} max {m.max():.1f}")
