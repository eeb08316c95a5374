"""N > 1 path of bench.py on CPU: two gloo ranks, MAX-over-ranks timing and whole-job aggregation
(decode attention shards over requests with no exchange: replicas, weak scaling - DESIGN.md section 4)."""
import multiprocessing
import os
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent


def _worker(rank, world, port, q):
    try:
        sys.path.insert(0, str(ROOT))
        os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
        import torch
        import torch.distributed as dist

        import bench

        dist.init_process_group(backend="gloo", rank=rank, world_size=world)
        wall = bench.max_over_ranks(1.0 + rank, torch.device("cpu"))
        dist.barrier()
        dist.destroy_process_group()
        q.put((rank, wall, bench.whole_job_value(2.0e9, world, wall, 10)))
    except Exception as e:  # noqa: BLE001
        q.put((rank, repr(e), None))


def test_bench_timing_aggregation_gloo_world2():
    ctx = multiprocessing.get_context("spawn")
    q = ctx.Queue()
    port = 29600 + os.getpid() % 300
    ps = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in ps:
        p.start()
    res = sorted(q.get(timeout=180) for _ in ps)
    for p in ps:
        p.join(timeout=60)
    assert [r[1] for r in res] == [2.0, 2.0], res  # the slowest rank defines the timed region
    assert all(abs(r[2] - 2.0e9 * 2 / 0.2 / 1e9) < 1e-6 for r in res)


def test_bench_gpus_flag_starts_that_many_ranks():
    """`python bench.py --gpus 2` must itself start 2 ranks (re-exec under torch.distributed.run when no
    launcher environment is present) and report n_gpus = 2 on rank 0; checked on CPU through the gloo
    self-test path that shares the launch plumbing, the timed-region bracket and the aggregation."""
    import json
    import subprocess

    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    r = subprocess.run([sys.executable, str(ROOT / "bench.py"), "--gpus", "2", "--steps", "5", "--warmup", "1",
                        "--cpu-selftest"], capture_output=True, text=True, timeout=300, env=env)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout  # exactly one JSON line, from rank 0
    rec = json.loads(lines[0])
    assert rec["n_gpus"] == 2 and rec["steps"] == 5 and rec["warmup"] == 1
    # the slowest rank (rank 1 sleeps 2 ms per step) defines the timed region
    assert rec["ms_per_step"] >= 2.0


def test_bench_under_launcher_does_not_relaunch():
    """the driver's form: already under torch.distributed.run (RANK / WORLD_SIZE set) -> no re-exec"""
    sys.path.insert(0, str(ROOT))
    import bench

    cmd = bench.relaunch_cmd(4, ["--gpus", "4", "--steps", "3"])
    assert cmd[1:3] == ["-m", "torch.distributed.run"] and "--nproc-per-node=4" in cmd and "127.0.0.1" in cmd
    old = dict(os.environ)
    try:
        os.environ.pop("RANK", None)
        assert not bench.under_launcher()
        os.environ.update(RANK="0", WORLD_SIZE="4")
        assert bench.under_launcher()
    finally:
        os.environ.clear()
        os.environ.update(old)
