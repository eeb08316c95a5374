"""Packaging of the `hpc` package (reference: setup.py + CMakeLists.txt:10-106 + Makefile build an installable `hpc` wheel
around hpc/_C.abi3.so).  Here the native code is built by build.py - hipcc for gfx950 straight from csrc/*.hip, g++ for the
C++ torch shim - into hpc/libhpc_amd.so and hpc/_hpc_torch.so, which ship as package data:

    cd hpc-ops_amd && pip wheel . --no-build-isolation --no-deps -w dist     # or: pip install . --no-build-isolation

The development build (libhpc_amd_dev.so: tuning registers, profiling variants) is not packaged.  The in-tree layout
(tests and bench.py put hpc-ops_amd/ on sys.path) stays the primary way to run inside this repository."""
import subprocess
import sys
from pathlib import Path

from setuptools import setup
from setuptools.command.build_py import build_py

HERE = Path(__file__).resolve().parent


def _version() -> str:
    try:
        h = subprocess.check_output(["git", "rev-parse", "--short=7", "HEAD"], cwd=HERE, stderr=subprocess.DEVNULL, text=True).strip()
    except Exception:  # noqa: BLE001
        h = "unknown"
    return f"0.0.1.dev0+g{h}"


class BuildNative(build_py):
    """build_py that first compiles the native libraries in-tree (no CMake, no CUTLASS: see build.py)"""

    def run(self):
        subprocess.check_call([sys.executable, str(HERE / "build.py"), "--no-dev"], cwd=str(HERE))
        for lib in ("libhpc_amd.so", "_hpc_torch.so"):
            if not (HERE / "hpc" / lib).exists():
                raise RuntimeError(f"build.py did not produce hpc/{lib}")
        super().run()


setup(
    name="hpc",
    version=_version(),
    description="MI355X (gfx950) drop-in for the decode-step operator path of Tencent/hpc-ops: hand-written HIP behind the reference's hpc.* API",
    packages=["hpc"],
    package_dir={"hpc": "hpc"},
    package_data={"hpc": ["libhpc_amd.so", "_hpc_torch.so"]},
    exclude_package_data={"hpc": ["*_dev.so"]},
    python_requires=">=3.10",
    install_requires=["torch"],
    extras_require={"test": ["pytest>=7", "pytest-xdist>=3", "pytest-timeout", "numpy"]},  # pytest.ini: addopts = -n 6
    cmdclass={"build_py": BuildNative},
    zip_safe=False,
)
