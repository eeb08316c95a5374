"""CPU: libhpc_amd.so loads and exports every symbol declared in include/hpc_amd.h."""
import ctypes
import re
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent


def _declared_symbols():
    text = (ROOT / "include" / "hpc_amd.h").read_text()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(hpc_[a-z0-9_]+)\s*\(", text)))


def test_library_exports_every_declared_symbol():
    libpath = ROOT / "hpc-ops_amd" / "hpc" / "libhpc_amd.so"
    assert libpath.exists(), "run `python hpc-ops_amd/build.py` (or __graft_entry__.build()) first"
    lib = ctypes.CDLL(str(libpath))
    syms = _declared_symbols()
    assert len(syms) >= 4
    missing = [s for s in syms if not hasattr(lib, s)]
    assert not missing, f"declared in include/hpc_amd.h but not exported: {missing}"


def test_version_and_built_json():
    import json

    import hpc

    assert isinstance(hpc.__version__, str) and hpc.__version__
    info = json.loads(hpc.__built_json__)
    assert info["offload-arch"] == "gfx950"
    assert info["version"] == hpc.__version__


def test_python_surface_is_reexported():
    import hpc

    for name in ["fused_rmsnorm_with_scale"]:
        assert callable(getattr(hpc, name)), name
