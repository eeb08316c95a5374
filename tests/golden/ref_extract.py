"""Loads the PyTorch-eager oracle FUNCTIONS of the reference's own tests (build container only).

The reference test modules cannot be imported (they import the compiled `hpc` CUDA extension at the
top), but their in-file oracles are plain torch functions.  This helper cuts the requested function
definitions out of a reference test file with `ast`, rewrites the hard-coded device="cuda" to "cpu",
and executes ONLY those definitions - the reference's own code, run on the CPU.
Used by make_golden.py; never used at test time (the GPU box has no /root/reference)."""
import ast
import math
from pathlib import Path
from typing import Tuple

import torch
import torch.nn.functional as F

REF = Path("/root/reference")


def load(rel_path, names, extra=None):
    src = (REF / rel_path).read_text()
    tree = ast.parse(src)
    ns = {"torch": torch, "math": math, "F": F, "Tuple": Tuple}
    ns.update(extra or {})
    found = []
    for node in tree.body:
        if isinstance(node, ast.FunctionDef) and node.name in names:
            code = ast.get_source_segment(src, node).replace('device="cuda"', 'device="cpu"')
            exec(compile(code, f"{rel_path}:{node.name}", "exec"), ns)  # noqa: S102
            found.append(node.name)
    missing = set(names) - set(found)
    assert not missing, f"{rel_path}: {missing} not found"
    return ns
