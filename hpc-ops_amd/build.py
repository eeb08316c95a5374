"""Build libhpc_amd.so (hand-written HIP for gfx950) in-tree with hipcc.

Replaces the reference's CMake/CUTLASS build (CMakeLists.txt:10-106, setup.py) — there is no
CUTLASS, no nvcc and no CMake here: every csrc/*.hip is compiled straight to an object by hipcc
for --offload-arch=gfx950 and linked into hpc/libhpc_amd.so, which the Python package loads with
ctypes.  hipcc cross-compiles without a GPU, so this runs in the CPU-only build container.

Two libraries come out of the same sources:
  * hpc/libhpc_amd.so      - the PRODUCT: no development registers (csrc/hpc_dev.h: hpc_dev_tuning_get is the
                             constant 0, every A/B variant and timing-only kernel folds away), no HPC_AMD_TUNING;
  * hpc/libhpc_amd_dev.so  - the DEVELOPMENT build (-DHPC_DEV): registers + development entry points, for tools/ and
                             the tests marked `dev` (`HPC_AMD_DEV=1` makes the Python package load it).
Each has its C++ torch shim (hpc/_hpc_torch.so / hpc/_hpc_torch_dev.so).

    python hpc-ops_amd/build.py [-j N] [--force] [--no-dev]
"""
import argparse
import concurrent.futures as cf
import os
import subprocess
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent
CSRC = ROOT / "csrc"
OBJ = ROOT / "build"
LIB = ROOT / "hpc" / "libhpc_amd.so"
LIB_DEV = ROOT / "hpc" / "libhpc_amd_dev.so"
INCLUDE = ROOT.parent / "include"


def _git_hash() -> str:
    try:
        return subprocess.check_output(
            ["git", "rev-parse", "--short=7", "HEAD"], cwd=ROOT, stderr=subprocess.DEVNULL, text=True
        ).strip()
    except Exception:
        return "unknown"


def _flags():
    # NOTE: version macros are only applied to library.hip so that a new commit does not force a
    # rebuild of every kernel.
    return [
        "--offload-arch=gfx950",
        "-O3",
        "-std=c++17",
        "-fPIC",
        "-fno-gpu-rdc",
        "-Wall",
        "-Wno-unused-function",
        "-I" + str(INCLUDE),
        "-I" + str(CSRC),
    ]


# Per-file flags.  group_gemm_p8: no SLP vectorizer - in the k-loop of the ride-along body it pairs the four fused
# multiply-adds of a block's rescale into <2 x float> operations that the backend splits again, each with a `v_mov` splat of
# the scale in front (VALU instructions in the section that feeds the matrix pipe, 26 instead of 8 spilled SGPRs, 249
# instead of 243 registers); the hand-written schedule wants the scalar form it was written in.
_PER_FILE_FLAGS = {"group_gemm_p8": ["-fno-slp-vectorize"]}


def _deps_newer(obj: Path, src: Path) -> bool:
    if not obj.exists():
        return True
    t = obj.stat().st_mtime
    deps = [src] + list(CSRC.glob("*.h")) + list(INCLUDE.glob("*.h")) + [Path(__file__)]
    return any(d.stat().st_mtime > t for d in deps)


def _compile(src: Path, force: bool, dev: bool = False) -> Path:
    obj = (OBJ / "dev" if dev else OBJ) / (src.stem + ".o")
    if not force and not _deps_newer(obj, src):
        return obj
    cmd = ["hipcc"] + _flags() + (["-DHPC_DEV=1"] if dev else []) + _PER_FILE_FLAGS.get(src.stem, [])
    if src.stem == "library":
        h = _git_hash()
        cmd += ['-DHPC_VERSION_STR="0.0.1.dev0+g%s"' % h, '-DHPC_GIT_HASH_STR="%s"' % h]
    cmd += ["-c", str(src), "-o", str(obj)]
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError("hipcc failed for %s:\n%s\n%s" % (src.name, r.stdout, r.stderr))
    if r.stderr.strip():
        sys.stderr.write(r.stderr)
    return obj


def build(jobs: int = 0, force: bool = False, dev: bool = False) -> Path:
    """dev=False: hpc/libhpc_amd.so (the product); dev=True: hpc/libhpc_amd_dev.so (-DHPC_DEV)."""
    lib = LIB_DEV if dev else LIB
    (OBJ / "dev" if dev else OBJ).mkdir(parents=True, exist_ok=True)
    srcs = sorted(CSRC.glob("*.hip")) + sorted(CSRC.glob("*.cc"))
    jobs = jobs or min(len(srcs), os.cpu_count() or 4)
    with cf.ThreadPoolExecutor(max_workers=jobs) as ex:
        objs = list(ex.map(lambda s: _compile(s, force, dev), srcs))
    need_link = force or not lib.exists() or any(o.stat().st_mtime > lib.stat().st_mtime for o in objs)
    if need_link:
        cmd = ["hipcc", "--offload-arch=gfx950", "-shared", "-fPIC", "-o", str(lib)] + [
            str(o) for o in objs
        ] + ["-lpthread"]
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError("link failed:\n%s\n%s" % (r.stdout, r.stderr))
    return lib


SHIM = ROOT / "hpc" / "_hpc_torch.so"
SHIM_DEV = ROOT / "hpc" / "_hpc_torch_dev.so"


def build_torch_shim(force: bool = False, dev: bool = False):
    """hpc/_hpc_torch.so: the C++ host side (csrc/torch_*.cpp: TORCH_LIBRARY(hpc) + the registration of every op + the
    MulticastCommunicator torch class) on top of the C-ABI.  Host-only C++ compiled by g++ against the installed torch
    headers - the reference registers its ops the same way (src/*/entry.cc).  The objects do not depend on which
    C-ABI library they are linked against, so the development shim re-links the same objects."""
    import torch
    from torch.utils import cpp_extension as ce

    srcs = sorted(CSRC.glob("torch_*.cpp"))
    shim, lib = (SHIM_DEV, LIB_DEV) if dev else (SHIM, LIB)
    deps = srcs + [CSRC / "torch_common.h", INCLUDE / "hpc_amd.h", Path(__file__)]
    if not force and shim.exists() and all(d.stat().st_mtime <= shim.stat().st_mtime for d in deps) \
            and shim.stat().st_mtime >= lib.stat().st_mtime:
        return shim
    tlib = Path(ce.library_paths()[0])
    abi = int(getattr(torch._C, "_GLIBCXX_USE_CXX11_ABI", True))
    cc = ["g++", "-O2", "-std=c++17", "-fPIC", "-D__HIP_PLATFORM_AMD__=1", "-DUSE_ROCM=1",
          f"-D_GLIBCXX_USE_CXX11_ABI={abi}", "-w"]
    cc += ["-I" + p for p in ce.include_paths()] + ["-I/opt/rocm/include", "-I" + str(INCLUDE), "-I" + str(CSRC)]
    (OBJ / "torch").mkdir(parents=True, exist_ok=True)

    def one(src):
        obj = OBJ / "torch" / (src.stem + ".o")
        if not force and obj.exists() and all(d.stat().st_mtime <= obj.stat().st_mtime
                                              for d in (src, CSRC / "torch_common.h", INCLUDE / "hpc_amd.h", Path(__file__))):
            return obj
        r = subprocess.run(cc + ["-c", str(src), "-o", str(obj)], capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError("torch shim build failed (%s):\n%s\n%s" % (src.name, r.stdout[-3000:], r.stderr[-3000:]))
        return obj

    with cf.ThreadPoolExecutor(max_workers=len(srcs)) as ex:
        objs = list(ex.map(one, srcs))
    cmd = ["g++", "-shared", "-fPIC", "-o", str(shim)] + [str(o) for o in objs]
    cmd += ["-L" + str(tlib), "-ltorch", "-ltorch_cpu", "-ltorch_hip", "-lc10", "-lc10_hip",
            "-L" + str(lib.parent), "-l:" + lib.name, "-Wl,-rpath,$ORIGIN", "-Wl,-rpath," + str(tlib)]
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError("torch shim link failed:\n%s\n%s" % (r.stdout[-3000:], r.stderr[-3000:]))
    return shim


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("-j", type=int, default=0)
    ap.add_argument("--force", action="store_true")
    ap.add_argument("--no-dev", action="store_true", help="skip the development build")
    a = ap.parse_args()
    print(build(a.j, a.force))
    print(build_torch_shim(a.force))
    if not a.no_dev:
        print(build(a.j, a.force, dev=True))
        print(build_torch_shim(a.force, dev=True))
