"""Literal-bar report for the fused MoE: for every large row of the reference's grid
(tests/test_fuse_moe_blockwise.py:265-272) and for smoke()'s long-group case, the number of output elements outside
the reference's literal allclose(rtol=0.01, atol=0.01), for the shipped blockwise-rescale form and - in a development
build - for the FMA form (development key 18 = 2).  Prints one line per case; writes gpurun_out/moe_literal.json."""
import json
import os
import sys
from pathlib import Path

import torch

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT / "hpc-ops_amd"))
sys.path.insert(0, str(ROOT))
sys.path.insert(0, str(ROOT / "tests"))

import hpc  # noqa: E402
from oracle import fuse_moe as omoe  # noqa: E402
from test_fuse_moe_blockwise import _inputs  # noqa: E402


def misses(gt, my):
    a, b = gt.float(), my.float()
    err = (a - b).abs()
    bad = err > 0.01 + 0.01 * a.abs()
    worst = None
    if bad.any():
        i = int(torch.argmax(err * bad))
        worst = (float(a.reshape(-1)[i]), float(b.reshape(-1)[i]))
    return int(bad.sum()), worst


def forms():
    setter = getattr(hpc._C.lib, "hpc_dev_tuning_set", None)
    yield "shipped", (lambda: None), (lambda: None)
    if setter is not None and os.environ.get("MOE_REPORT_FMA", "1") == "1":
        yield "fma", (lambda: setter(18, 2)), (lambda: setter(18, 0))


def main():
    rows = []
    cases = [(T, inter, r, s, sh) for T in (1024, 2048, 4096) for inter in (512, 256) for (r, s) in ((0, 1), (1, 4), (0, 8))
             for sh in (False, True)]
    for T, inter, rank_ep, size_ep, shared in cases:
        args = _inputs(T, 8, 512, inter, 128, size_ep, shared)
        gt = omoe.fuse_moe_blockwise_fp8(*args[:8], rank_ep, 128, args[8])
        dev = [t.cuda() if t is not None else None for t in args]
        row = {"T": T, "inter": inter, "rank_ep": rank_ep, "size_ep": size_ep, "shared": shared, "numel": gt.numel(),
               "max_abs_ref": float(gt.float().abs().max())}
        for name, on, off in forms():
            on()
            try:
                my = hpc.fuse_moe_blockwise_fp8(*dev[:8], rank_ep, 128, dev[8]).cpu()
            finally:
                off()
            n, worst = misses(gt, my)
            row[name] = n
            if worst:
                row[name + "_worst_ref_real"] = worst
        rows.append(row)
        print(row, flush=True)
    Path(ROOT / "gpurun_out").mkdir(exist_ok=True)
    (ROOT / "gpurun_out" / "moe_literal.json").write_text(json.dumps(rows, indent=1))


if __name__ == "__main__":
    main()
