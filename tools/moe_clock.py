"""Development tool (GPU box): the core clock and the socket power UNDER the grouped GEMMs of the fused MoE (VERDICT round 5,
missing #6: DESIGN 3.3 says the chip runs ~1.65 GHz under this load - "the 5 PF peak assumes 2.4" - from s_memtime ticks per
k-tile; no file recorded it).  Three independent readings, side by side, for an idle chip, a memory-bound kernel (decode) and
the two GEMMs:
  1. s_memtime / s_memrealtime: eight one-wave workgroups (one per XCD; tools/probes/probe_clock.hip) run BESIDE the kernel on a
     second stream and stamp both counters every 50 us; core clock of a window = d(s_memtime) / d(s_memrealtime) x the reference
     clock, which is calibrated against HIP events in the same run;
  2. sysfs (pp_dpm_sclk / hwmon freq1_input, power1_average | power1_input) sampled by a host thread every ~20 ms;
  3. `amd-smi metric` / `rocm-smi` once during the load, when the tools answer for an ordinary user.
And the ceiling those clocks imply, measured instead of computed: tools/probes/probe_mfma.hip runs NOTHING BUT back-to-back
v_mfma_f32_16x16x128_f8f6f4 (the instruction of the grouped GEMMs; register operands, no memory, no LDS, no VALU, 2 waves per
SIMD on every CU) with pseudo-random fp8 operands and with all-zero operands: the TFLOP/s the chip sustains on this instruction
under its power limit - the denominator a software schedule can actually be held against.
usage: python tools/moe_clock.py [out.json]      (writes profiles/round6_moe_clock.json by default)"""
import ctypes, glob, json, os, subprocess, sys, threading, time
from pathlib import Path
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT / "hpc-ops_amd")); sys.path.insert(0, str(ROOT))
import torch, bench, hpc

dev = torch.device("cuda", 0)
out_path = Path(sys.argv[1]) if len(sys.argv) > 1 else ROOT / "profiles" / "round6_moe_clock.json"
bindir = ROOT / "tools" / "probes" / "bin"
bindir.mkdir(exist_ok=True)
so = bindir / "libprobe_clock.so"
src = ROOT / "tools" / "probes" / "probe_clock.hip"
if not so.exists() or so.stat().st_mtime < src.stat().st_mtime:
    subprocess.check_call(["hipcc", "--offload-arch=gfx950", "-O2", "-shared", "-fPIC", str(src), "-o", str(so)])
so2 = bindir / "libprobe_mfma.so"
src2 = ROOT / "tools" / "probes" / "probe_mfma.hip"
if not so2.exists() or so2.stat().st_mtime < src2.stat().st_mtime:
    subprocess.check_call(["hipcc", "--offload-arch=gfx950", "-O3", "-shared", "-fPIC", str(src2), "-o", str(so2)])
mfma = ctypes.CDLL(str(so2))
mfma.mfma_probe_launch.restype = ctypes.c_int
mfma.mfma_probe_launch.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_void_p]
probe = ctypes.CDLL(str(so))
probe.clock_probe_launch.restype = ctypes.c_int
probe.clock_probe_launch.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_void_p]
NWG, PERIOD = 8, 5000  # 8 workgroups, one stamp per 5000 reference ticks (50 us at 100 MHz)


# ---- sysfs sampler -----------------------------------------------------------------------------------------------------
# Every amdgpu card the box exposes is read: an ordinary user cannot tell from sysfs alone which card is the GPU the process
# was given (the first run of this tool read a neighbour that sat at 1 414 MHz / 263 W through everything); the card whose
# power MOVES with the load is the one.
def sysfs_sources():
    cards = {}
    for card in sorted(glob.glob("/sys/class/drm/card*/device")):
        if not os.path.exists(card + "/pp_dpm_sclk"):
            continue
        srcs = {"pp_dpm_sclk": card + "/pp_dpm_sclk"}
        for hw in glob.glob(card + "/hwmon/hwmon*"):
            for f in ("freq1_input", "power1_average", "power1_input"):
                if os.path.exists(f"{hw}/{f}"):
                    srcs[f] = f"{hw}/{f}"
        cards[card.split("/")[4]] = srcs
    return cards


def read_sysfs(srcs):
    r = {}
    for k, p in srcs.items():
        try:
            t = open(p).read()
        except OSError:
            continue
        try:
            if k == "pp_dpm_sclk":
                cur = [ln for ln in t.splitlines() if ln.strip().endswith("*")]
                if cur:
                    r["sclk_mhz_dpm"] = float(cur[0].split(":")[1].strip().rstrip("*").strip().lower().replace("mhz", ""))
            elif k == "freq1_input":
                r["sclk_mhz_hwmon"] = float(t) / 1e6
            else:
                r["power_w"] = float(t) / 1e6
        except ValueError:
            pass
    return r


class Sampler(threading.Thread):
    def __init__(self, cards):
        super().__init__(daemon=True)
        self.cards, self.rows, self.stop = cards, {c: [] for c in cards}, False

    def run(self):
        while not self.stop:
            for c, srcs in self.cards.items():
                self.rows[c].append(read_sysfs(srcs))
            time.sleep(0.02)

    def summary(self):
        out = {}
        for c, rows in self.rows.items():
            o = {"samples": len(rows)}
            for k in ("sclk_mhz_dpm", "sclk_mhz_hwmon", "power_w"):
                v = sorted(r[k] for r in rows if k in r)
                if v:
                    o[k] = {"min": v[0], "median": v[len(v) // 2], "max": v[-1]}
            out[c] = o
        return out


def smi_once():
    for cmd in (["amd-smi", "metric", "-g", "0", "--clock", "--power", "--json"], ["rocm-smi", "--showclocks", "--showpower", "--json"]):
        try:
            p = subprocess.run(cmd, capture_output=True, text=True, timeout=20)
            if p.returncode == 0 and p.stdout.strip():
                return {"command": " ".join(cmd), "output": p.stdout.strip()[:3000]}
        except Exception:  # noqa: BLE001
            continue
    return None


# ---- one measurement: `work` (a callable that enqueues ~`ms` milliseconds of GPU work on the current stream) with the probe beside it
def measure(name, work, ms, srcs, with_smi=False):
    samples = int(ms * 1e3 / 50) + 40
    buf = torch.zeros(NWG * samples * 2, dtype=torch.int64, device=dev)
    side = torch.cuda.Stream()
    torch.cuda.synchronize()
    samp = Sampler(srcs)
    samp.start()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    with torch.cuda.stream(side):
        e0.record()
        assert probe.clock_probe_launch(buf.data_ptr(), NWG, samples, PERIOD, side.cuda_stream) == 0
        e1.record()
    w0, w1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    w0.record()
    if work is not None:
        work()
    w1.record()
    smi = smi_once() if with_smi else None
    torch.cuda.synchronize()
    samp.stop = True
    samp.join()
    b = buf.cpu().view(NWG, samples, 2)
    probe_ms, work_ms = e0.elapsed_time(e1), w0.elapsed_time(w1)
    # reference clock: realtime ticks of the whole probe over its event time (launch overhead makes this a slight underestimate)
    ref_hz = float((b[:, -1, 1] - b[:, 0, 1]).double().mean()) / (probe_ms * 1e-3)
    res = {"case": name, "work_ms": round(work_ms, 2), "probe_ms": round(probe_ms, 2), "reference_clock_mhz_measured": round(ref_hz / 1e6, 2)}
    # windows that lie inside the work: skip the first 15 % and the last 15 % of the work's duration
    n_in = int(min(work_ms, probe_ms) * 1e3 / 50)
    lo, hi = (max(2, int(n_in * 0.15)), max(4, int(n_in * 0.85))) if work is not None else (2, samples - 2)
    per_xcd = []
    for x in range(NWG):
        dt = (b[x, hi, 0] - b[x, lo, 0]).item()
        dr = (b[x, hi, 1] - b[x, lo, 1]).item()
        per_xcd.append(round(dt / dr * 100.0, 1))  # MHz at a 100 MHz reference
        # (the probe cannot say which XCD a workgroup landed on; workgroups of one launch go round the XCDs in index order)
    res["core_clock_mhz_per_probe_workgroup"] = per_xcd
    res["core_clock_ghz"] = round(sum(per_xcd) / len(per_xcd) / 1e3, 3)
    # finest-grained view of workgroup 0: min / median / max over 1 ms windows
    win = 20
    w = [((b[0, i + win, 0] - b[0, i, 0]).item() / (b[0, i + win, 1] - b[0, i, 1]).item()) * 100.0 for i in range(lo, hi - win, win)]
    if w:
        w.sort()
        res["core_clock_mhz_1ms_windows"] = {"min": round(w[0], 1), "median": round(w[len(w) // 2], 1), "max": round(w[-1], 1), "n": len(w)}
    res["sysfs"] = samp.summary()
    if smi:
        res["smi"] = smi
    print(json.dumps(res), flush=True)
    return res


srcs = sysfs_sources()
results = {"taken_at": subprocess.run(["git", "rev-parse", "--short=7", "HEAD"], cwd=ROOT, capture_output=True, text=True).stdout.strip() or "gpurun snapshot",
           "method": __doc__.split("usage:")[0].strip(), "sysfs_sources": srcs, "cases": []}

# idle
results["cases"].append(measure("idle", None, 30, srcs))

# the two GEMMs of the fused MoE at the routed sizes of the bench generator (tools/tune_ggemm.py's cases) and the fused op
w = bench.C4
m = bench.c4_inputs(dev, w)
E = w["num_expert"]
step = lambda: hpc.fuse_moe_blockwise_fp8(m["x"], m["x_scale"], m["guw"], m["guws"], m["dw"], m["dws"], m["ids"], m["scale"], 0, E)  # noqa: E731
step(); torch.cuda.synchronize()
us = bench.timed(step, iters=5, warm=2, graph=True)
g = bench.capture(step, reps=10)
reps = max(2, int(400e3 / (us * 10)))  # ~0.4 s of back-to-back fused ops


def moe_work():
    for _ in range(reps):
        g.replay()


r = measure("fused MoE blockwise T=4096 (gate-up + down GEMM > 95 % of the time)", moe_work, reps * 10 * us / 1e3, srcs, with_smi=True)
r["us_per_fused_op_alone"] = round(us, 1)
results["cases"].append(r)

# zero-filled weights: the same instruction stream, no toggling in the multipliers ("4-8 % faster")
mz = dict(m)
mz["guw"], mz["dw"] = torch.zeros_like(m["guw"]), torch.zeros_like(m["dw"])
stepz = lambda: hpc.fuse_moe_blockwise_fp8(mz["x"], mz["x_scale"], mz["guw"], mz["guws"], mz["dw"], mz["dws"], mz["ids"], mz["scale"], 0, E)  # noqa: E731
stepz(); torch.cuda.synchronize()
usz = bench.timed(stepz, iters=5, warm=2, graph=True)
gz = bench.capture(stepz, reps=10)


def moez_work():
    for _ in range(reps):
        gz.replay()


r = measure("fused MoE blockwise T=4096, ZERO weights (same instructions, idle multipliers)", moez_work, reps * 10 * usz / 1e3, srcs)
r["us_per_fused_op_alone"] = round(usz, 1)
results["cases"].append(r)
del mz, gz

# ---- the matrix pipe on its own: what the power limit leaves of the 5 PF -------------------------------------------------
cus = torch.cuda.get_device_properties(0).multi_processor_count
sink = torch.zeros(16, dtype=torch.float32, device=dev)
for label, mode in (("pure MFMA loop (16x16x128 f8f6f4), pseudo-random fp8 operands", 0), ("pure MFMA loop (16x16x128 f8f6f4), all-zero operands", 1)):
    def run(iters):
        assert mfma.mfma_probe_launch(sink.data_ptr(), cus, iters, mode, torch.cuda.current_stream().cuda_stream) == 0
    run(200); torch.cuda.synchronize()
    t0, t1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0.record(); run(4000); t1.record(); torch.cuda.synchronize()
    per_iter_ms = t0.elapsed_time(t1) / 4000
    iters = int(300.0 / per_iter_ms)
    holder = {}

    def mfma_work():
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); run(iters); b.record()
        holder["ev"] = (a, b)

    r = measure(label, mfma_work, iters * per_iter_ms, srcs, with_smi=(mode == 0))
    ms = holder["ev"][0].elapsed_time(holder["ev"][1])
    flops = cus * 8.0 * iters * 32 * 2 * 16 * 16 * 128
    r["tflops"] = round(flops / ms / 1e9, 1)
    r["frac_of_5PF"] = round(flops / ms / 1e9 / 5000.0, 4)
    r["tflops_at_2.4GHz_by_cycle_count"] = round(cus * 4 * 2048 * 2.4e9 / 1e12, 1)
    print(json.dumps({"case": label, "tflops": r["tflops"], "clock_ghz": r["core_clock_ghz"]}), flush=True)
    results["cases"].append(r)

# a memory-bound kernel for contrast: the FP8 decode headline
inp = bench.c3_inputs(dev)
tm = hpc.get_attention_decode_task_workspace(bench.C3["batch"], int(inp["kv_lens"].max()), bench.C3["num_head_kv"], bench.C3["min_process_len"])
hpc.assign_attention_decode_task(inp["kv_lens"], tm, bench.C3["num_head_kv"], 1, True, bench.C3["min_process_len"])
o8 = torch.empty(bench.C3["batch"], bench.C3["num_head_q"], 128, dtype=torch.bfloat16, device=dev)
dstep = lambda: hpc.attention_decode_fp8(inp["q"], inp["k_cache"], inp["v_cache"], inp["block_ids"], inp["kv_lens"], inp["q_scale"],  # noqa: E731
                                         inp["k_scale"], inp["v_scale"], 0, True, hpc.QuantType.QPERTOKEN_PERHEAD_KPERTENSOR_VPERTENSOR,
                                         True, tm, None, o8)
dstep(); torch.cuda.synchronize()
usd = bench.timed(dstep, graph=True, reps=10)
gd = bench.capture(dstep, reps=10)
dreps = max(2, int(300e3 / (usd * 10)))


def dec_work():
    for _ in range(dreps):
        gd.replay()


r = measure("FP8 decode attention, BASELINE configs[2] (HBM-bound)", dec_work, dreps * 10 * usd / 1e3, srcs)
r["us_per_call_alone"] = round(usd, 1)
results["cases"].append(r)

moe = results["cases"][1]
pure = next((c for c in results["cases"] if c["case"].startswith("pure MFMA loop") and "random" in c["case"]), {})
pure0 = next((c for c in results["cases"] if c["case"].startswith("pure MFMA loop") and "zero" in c["case"]), {})
results["summary"] = {"clock_ghz_under_moe": moe["core_clock_ghz"], "clock_ghz_idle_probe": results["cases"][0]["core_clock_ghz"],
                      "fp8_dense_peak_at_that_clock_tflops": round(5000.0 * moe["core_clock_ghz"] / 2.4, 1),
                      "pure_mfma_random_operands_tflops": pure.get("tflops"), "pure_mfma_random_operands_clock_ghz": pure.get("core_clock_ghz"),
                      "pure_mfma_zero_operands_tflops": pure0.get("tflops"), "pure_mfma_zero_operands_clock_ghz": pure0.get("core_clock_ghz"),
                      "note": "5 PF = 256 CUs x 4 SIMDs x 2048 flop/clk (16x16x128 f8f6f4: 65536 flop / 32 clk) x 2.4 GHz; the matrix pipe's rate scales with the core clock"}
out_path.write_text(json.dumps(results, indent=1) + "\n")
print("wrote", out_path)
