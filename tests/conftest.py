import os
import sys
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parent.parent
# in-tree package (mirrors the reference tests' sys.path.insert of build/lib.*/)
sys.path.insert(0, str(ROOT / "hpc-ops_amd"))
sys.path.insert(0, str(ROOT))
sys.path.insert(0, str(ROOT / "tests"))

os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
if os.environ.get("PYTEST_XDIST_WORKER"):
    # xdist workers share the host cores: the PyTorch-eager oracles get slower, not faster, beyond a few dozen threads
    _nw = int(os.environ.get("PYTEST_XDIST_WORKER_COUNT", "1") or 1)
    os.environ.setdefault("OMP_NUM_THREADS", str(max(2, min(32, (os.cpu_count() or 8) // max(_nw, 1)))))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    config.addinivalue_line("markers", "exclusive_gpu: the test needs the GPU to itself (several processes whose grids must be "
                                       "co-resident); serialised against every other gpu test across xdist workers")
    config.addinivalue_line("markers", "gate_free: a gpu test that only drives a nested pytest run and must not hold the GPU "
                                       "gate itself (its inner tests take it)")
    config.addinivalue_line("markers", "dev: pins kernel variants through development registers, which only the "
                                       "development build of the library has (HPC_AMD_DEV=1 -> libhpc_amd_dev.so); "
                                       "tests/test_dev_build.py re-runs these in a subprocess against that build")


    # HPC_REPLAY_CHECK=1: every public hpc.* call is replayed in a fresh process with guard bands around its arguments
    # and must reproduce byte for byte (tests/replay_check.py; reference conftest.py: SANITIZER_CHECK=memcheck,...)
    if os.environ.get("HPC_REPLAY_CHECK") == "1":
        import replay_check

        replay_check.install()


def pytest_collection_modifyitems(config, items):
    try:
        import torch

        has_gpu = torch.cuda.is_available()
    except Exception:
        has_gpu = False
    if has_gpu:
        return
    skip = pytest.mark.skip(reason="no GPU in this container")
    for item in items:
        if "gpu" in item.keywords:
            item.add_marker(skip)


# ---- one GPU, several pytest processes ---------------------------------------------------------------------------------
# Every gpu test holds a SHARED lock on a fixed lock file while it runs; `exclusive_gpu` tests take it EXCLUSIVELY (through
# a second "gate" lock, so that a stream of overlapping shared holders cannot starve the exclusive waiter: new shared
# holders queue at the gate while the exclusive test waits for the running ones to drain).  The files are fixed paths in
# the temp directory, so nested pytest runs (tests/test_dev_build.py) and plain serial runs take part without any plumbing;
# flock locks die with their process.
def _gate_files():
    import tempfile

    base = os.path.join(tempfile.gettempdir(), "hpc_amd_gpu_gate_%d" % os.getuid())
    return base + ".gate", base + ".main"


@pytest.fixture(autouse=True)
def _gpu_gate(request):
    if "gpu" not in request.keywords or request.node.get_closest_marker("gate_free") is not None:
        yield
        return
    import fcntl

    exclusive = request.node.get_closest_marker("exclusive_gpu") is not None
    gate_path, main_path = _gate_files()
    with open(gate_path, "a") as gate, open(main_path, "a") as main:
        if exclusive:
            fcntl.flock(gate, fcntl.LOCK_EX)
            fcntl.flock(main, fcntl.LOCK_EX)
        else:
            fcntl.flock(gate, fcntl.LOCK_SH)
            fcntl.flock(main, fcntl.LOCK_SH)
            fcntl.flock(gate, fcntl.LOCK_UN)
        try:
            yield
        finally:
            fcntl.flock(main, fcntl.LOCK_UN)
            if exclusive:
                fcntl.flock(gate, fcntl.LOCK_UN)
