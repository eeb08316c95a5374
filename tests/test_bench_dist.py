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
