"""The `hpc` package is installable (reference: setup.py / CMakeLists.txt / Makefile build an `hpc` wheel): hpc-ops_amd/setup.py
+ pyproject.toml package hpc/*.py with the two product libraries (built by build.py) and nothing of the development build.
The wheel itself is built by `pip wheel hpc-ops_amd --no-build-isolation --no-deps` (minutes: not part of the suite); here the
metadata and the file list the build would package are checked."""
import subprocess
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
PKG = ROOT / "hpc-ops_amd"


def test_setup_metadata_and_package_data():
    out = subprocess.run([sys.executable, "setup.py", "--name", "--version"], cwd=str(PKG), capture_output=True, text=True, check=True).stdout.split()
    assert out[0] == "hpc" and out[1].startswith("0.0.1.dev0+g")
    src = (PKG / "setup.py").read_text()
    for lib in ("libhpc_amd.so", "_hpc_torch.so"):
        assert lib in src and (PKG / "hpc" / lib).exists(), lib
    assert "*_dev.so" in src  # the development build is excluded
    assert (PKG / "pyproject.toml").exists()
    # every public module of the reference package is a module of ours (the wheel packages hpc/*.py)
    ours = {p.name for p in (PKG / "hpc").glob("*.py")}
    for mod in ("attention.py", "fuse_moe.py", "group_gemm.py", "gemm.py", "allreduce.py", "normalization.py", "rope.py", "sampler.py",
                "act.py", "multicast_handle.py"):
        assert mod in ours, mod
