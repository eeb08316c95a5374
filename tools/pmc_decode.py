"""Development tool: a few plain launches of FP8 decode attention (C3 mix and uniform 8k; both kernel generations)
so a rocprofv3 --pmc pass can attribute SQ / TCC counters to the decode kernels.
usage: rocprofv3 --pmc <counters> --kernel-trace --output-format csv -d <dir> -- python tools/pmc_decode.py [mixed|uniform8k]
       python tools/pmc_decode.py --summarise <out.json> <dir> [<dir>...]"""
import os
os.environ.setdefault("HPC_AMD_DEV", "1")  # development build of the library: tuning registers
import csv, json, sys
from collections import defaultdict
from pathlib import Path
ROOT = Path(__file__).resolve().parent.parent
if len(sys.argv) > 1 and sys.argv[1] == "--summarise":
    acc = defaultdict(lambda: defaultdict(list))
    for d in sys.argv[3:]:
        for f in Path(d).rglob("*counter_collection.csv"):
            for r in csv.DictReader(open(f)):
                if "decode" in r["Kernel_Name"] and "combine" not in r["Kernel_Name"]:
                    name = r["Kernel_Name"].split("(")[0].replace("void ", "")
                    acc[name][r["Counter_Name"]].append(float(r["Counter_Value"]))
    out = {n: {c: sum(v) / len(v) for c, v in cs.items()} for n, cs in acc.items()}
    json.dump(out, open(sys.argv[2], "w"), indent=1)
    print(json.dumps(out, indent=1))
    sys.exit(0)
sys.path.insert(0, str(ROOT / "hpc-ops_amd")); sys.path.insert(0, str(ROOT))
import torch, bench, hpc
from hpc import _C
dev = torch.device("cuda", 0)
case = sys.argv[1] if len(sys.argv) > 1 else "mixed"
lens_c = bench.c3_lens() if case == "mixed" else torch.full((64,), 8192, dtype=torch.int32)
inp = bench.c3_inputs(dev, bench.C3, lens=lens_c)
tm = hpc.get_attention_decode_task_workspace(64, int(lens_c.max()), 8, 64)
hpc.assign_attention_decode_task(inp["kv_lens"], tm, 8, 1, True, 64)
o = torch.empty(64, 64, 128, dtype=torch.bfloat16, device=dev)
for gen in (1, 0):
    _C.lib.hpc_dev_tuning_set(12, gen)
    for _ in range(3):
        hpc.attention_decode_fp8(inp["q"], inp["k_cache"], inp["v_cache"], inp["block_ids"], inp["kv_lens"], inp["q_scale"],
                                 inp["k_scale"], inp["v_scale"], 0, True, hpc.QuantType.QPERTOKEN_PERHEAD_KPERTENSOR_VPERTENSOR,
                                 True, tm, None, o)
    torch.cuda.synchronize()
print("done", case, "kv bytes", int(lens_c.sum()) * 8 * 256)
