"""hpc — MI355X (gfx950) drop-in for the decode-step hot path of Tencent/hpc-ops.

Import surface of the reference package (reference hpc/__init__.py): every public callable of every
public submodule is also reachable as `hpc.<name>`, and `hpc.__version__` / `hpc.__built_json__` come from
the native library.  That library is libhpc_amd.so (hand-written HIP, C-ABI in include/hpc_amd.h), loaded by
hpc/_C.py - importing this package fails if it has not been built; ops are registered under
torch.ops.hpc.* with the reference's schemas.
"""
import importlib
from pathlib import Path

import torch

from . import _C  # loads libhpc_amd.so or raises - there is no fallback path

__all__ = []
for _src in sorted(Path(__file__).parent.glob("[!_]*.py")):  # _C comes in through the public modules
    _mod = importlib.import_module(f"{__name__}.{_src.stem}")
    for _name, _obj in vars(_mod).items():
        if callable(_obj) and not _name.startswith("_"):
            globals()[_name] = _obj
            __all__.append(_name)
del _src, _mod, _name, _obj

__version__ = torch.ops.hpc.version()
__built_json__ = torch.ops.hpc.built_json()
