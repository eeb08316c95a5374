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


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
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
