"""Every torch.ops.hpc.* op must carry the reference's schema: argument names, types, defaults, mutability marks and
return annotation (keyword call sites on torch.ops.hpc.* depend on the names).  The reference strings were extracted
from /root/reference/src/**/entry.cc by tests/golden/extract_schemas.py into tests/golden/ref_schemas.json."""
import json
from pathlib import Path

import pytest
import torch

import hpc  # noqa: F401  (registers the ops)

REF = json.loads((Path(__file__).parent / "golden" / "ref_schemas.json").read_text())
# ops of the reference that are outside the decode-step path (stem_* block-sparse-attention helpers)
OUT_OF_SCOPE = {"stem_oam_gemm", "stem_oam_prep_paged_kv", "stem_oam_prep_varlen_q", "stem_tpd"}
# ops without a reference counterpart (BASELINE north_star asks for a top-k router; the reference stops at the GEMM)
# internal ops of the host library: drop / list its cached decode scratch buffers (hpc.release_decode_workspaces)
OURS_ONLY = {"topk_router", "_release_decode_workspaces", "_decode_workspaces"}


def _canon(schema: str) -> str:
    return str(torch._C.parse_schema("hpc::" + schema if not schema.startswith("hpc::") else schema))


@pytest.mark.parametrize("name", sorted(n for n, v in REF.items() if v["schema"] and n not in OUT_OF_SCOPE))
def test_schema_equals_reference(name):
    assert hasattr(torch.ops.hpc, name), "op hpc::%s is not registered" % name
    ours = str(getattr(torch.ops.hpc, name).default._schema)
    assert ours == _canon(REF[name]["schema"]), "%s differs from %s" % (name, REF[name]["at"])


def test_no_unknown_ops():
    """Everything registered under hpc:: is either a reference op or a documented addition."""
    names = {n.split("::")[1].split(".")[0] for n in torch._C._dispatch_get_all_op_names() if n.startswith("hpc::")}
    extra = names - set(REF) - OURS_ONLY
    assert not extra, extra
    for n in ("version", "built_json"):
        assert n in names


# ---- the Python surface: every public function / method of the reference package with the same argument names and defaults
PY_REF = json.loads((Path(__file__).parent / "golden" / "ref_py_signatures.json").read_text())


def _our_signatures():
    import sys

    sys.path.insert(0, str(Path(__file__).parent / "golden"))
    from extract_schemas import py_signatures

    ours = {}
    for f in sorted((Path(hpc.__file__).parent).glob("*.py")):
        for name, sig in py_signatures(f).items():
            ours.setdefault(name, sig)
    return ours


@pytest.mark.parametrize("name", sorted(n for n in PY_REF if not n.startswith("stem_")))
def test_python_signature_equals_reference(name):
    """call sites written against the reference package (positional or keyword) keep working: same names, same order, same
    defaults (reference hpc/*.py, extracted by tests/golden/extract_schemas.py)"""
    ours = _our_signatures()
    assert name in ours, "hpc.%s (reference hpc/%s) is missing" % (name, PY_REF[name]["module"])
    assert ours[name] == PY_REF[name]["args"], name
