"""Literal-bar report for the fused MoE: for every large row of the reference's grid
(tests/test_fuse_moe_blockwise.py:265-272, reference generator, seed 41) the number of output elements / token rows
outside the reference's literal allclose(rtol=0.01, atol=0.01) and the largest such error, for
  hip_vs_eager   - the HIP op against the reference TEST's eager model (oracle.fuse_moe.fuse_moe_blockwise_fp8),
  ka_vs_eager    - the CPU restatement of the reference KERNEL's arithmetic (kernel_arith=True) against the same model,
  hip_vs_ka      - the HIP op against that restatement.
Writes gpurun_out/moe_literal.json (copied to profiles/round4_moe_literal.json)."""
import json
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT / "hpc-ops_amd"))
sys.path.insert(0, str(ROOT))
sys.path.insert(0, str(ROOT / "tests"))

import hpc  # noqa: E402
from oracle import fuse_moe as omoe  # noqa: E402
from test_fuse_moe_blockwise import _inputs, _literal_misses  # noqa: E402


def main():
    rows = []
    cases = [(T, inter, r, s, sh) for T in (1024, 2048, 4096) for inter in (512, 256) for r in (0, 1) for s in (1, 4, 8)
             for sh in (False, True)]
    for T, inter, rank_ep, size_ep, shared in cases:
        args = _inputs(T, 8, 512, inter, 128, size_ep, shared)
        eager = omoe.fuse_moe_blockwise_fp8(*args[:8], rank_ep, 128, args[8])
        ka = omoe.fuse_moe_blockwise_fp8(*args[:8], rank_ep, 128, args[8], kernel_arith=True)
        dev = [t.cuda() if t is not None else None for t in args]
        my = hpc.fuse_moe_blockwise_fp8(*dev[:8], rank_ep, 128, dev[8]).cpu()
        row = {"T": T, "inter": inter, "rank_ep": rank_ep, "size_ep": size_ep, "shared": shared, "numel": eager.numel(),
               "hip_vs_eager": _literal_misses(eager, my), "ka_vs_eager": _literal_misses(eager, ka),
               "hip_vs_ka": _literal_misses(ka, my)}
        rows.append(row)
        print(row, flush=True)
    tot = {k: [sum(r[k][0] for r in rows), sum(r[k][1] for r in rows), max(r[k][2] for r in rows)]
           for k in ("hip_vs_eager", "ka_vs_eager", "hip_vs_ka")}
    print("TOTAL (elements, rows, max err) over %d cases / %d elements:" % (len(rows), sum(r["numel"] for r in rows)), tot)
    Path(ROOT / "gpurun_out").mkdir(exist_ok=True)
    (ROOT / "gpurun_out" / "moe_literal.json").write_text(json.dumps({"cases": rows, "total": tot,
                                                                     "version": hpc.__version__}, indent=1))


if __name__ == "__main__":
    main()
