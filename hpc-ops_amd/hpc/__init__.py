"""hpc — MI355X (gfx950) drop-in for the decode-step hot path of Tencent/hpc-ops.

Same import surface as the reference package (reference hpc/__init__.py:12-52): every public
callable of every hpc/*.py module is re-exported at package level, `hpc.__version__` and
`hpc.__built_json__` come from the native library.  The native library is libhpc_amd.so
(hand-written HIP, C-ABI in include/hpc_amd.h) loaded by hpc/_C.py; ops are registered under
torch.ops.hpc.* with the reference's schemas.
"""
import importlib
import sys
from pathlib import Path
from types import ModuleType
from typing import Dict

import torch

from . import _C  # loads libhpc_amd.so or raises — no fallback path exists

_pkg_dir = Path(__file__).parent


def _discover_modules() -> Dict[str, ModuleType]:
    modules = {}
    for file in sorted(_pkg_dir.iterdir()):
        if file.suffix != ".py" or file.name.startswith("_"):
            continue
        module_name = file.stem
        modules[module_name] = importlib.import_module(f".{module_name}", package=__package__)
    return modules


def _export_functions(modules: Dict[str, ModuleType]):
    for module_name, module in modules.items():
        funcs = {
            name: obj
            for name, obj in vars(module).items()
            if callable(obj) and not name.startswith("_")
        }
        globals().update(funcs)
        __all__.extend(funcs.keys())


__all__ = []

_export_functions(_discover_modules())

__version__ = torch.ops.hpc.version()
__built_json__ = torch.ops.hpc.built_json()
