"""CPU (hipcc cross-compiles): register budget of the decode kernels that ship (csrc/attention_decode_v2.hip), as build.py
compiles them.  Every form of the head-pair pipeline runs two workgroups of four waves per CU - 256 registers per lane, nothing
in scratch - and the hand-counted vmcnt pipeline assumes that no load destination is spilled; a source or toolchain change that
pushes a form over the budget would still pass every parity test and quietly lose its second workgroup (or spill inside the
wave-iteration).  This test pins the budget per shipped instantiation:
  decode2_kernel<2, bf16, prof, quad, ktok, hnd, solo>: the fp8 head-pair form (the headline), its bf16 form, per-token K scales,
  one kv head per workgroup (17-32 q rows) with per-tensor scales, with per-token K scales, and in bf16."""
import importlib.util
import re
import shutil
import subprocess
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parents[1]


def _build_module():
    spec = importlib.util.spec_from_file_location("hpc_amd_build_for_test_decode", ROOT / "hpc-ops_amd" / "build.py")
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


# mangled template arguments (kAux, kBf16, kProf, kQuad, kKtok, kHnd, kSolo) -> (name, max VGPRs, max spilled VGPRs)
SHIPPED = {
    "ILi2ELb0ELb0ELb0ELb0ELb0ELb0EE": ("fp8 head pairs (headline)", 232, 0),
    "ILi2ELb1ELb0ELb0ELb0ELb0ELb0EE": ("bf16 head pairs", 236, 0),
    "ILi2ELb0ELb0ELb0ELb1ELb0ELb0EE": ("fp8 head pairs, per-token K scales", 236, 0),
    "ILi2ELb0ELb0ELb0ELb0ELb0ELb1EE": ("fp8 one head per workgroup", 256, 0),
    "ILi2ELb0ELb0ELb0ELb1ELb0ELb1EE": ("fp8 one head per workgroup, per-token K scales", 256, 4),
    "ILi2ELb1ELb0ELb0ELb0ELb0ELb1EE": ("bf16 one head per workgroup", 244, 0),
}


@pytest.mark.skipif(shutil.which("hipcc") is None, reason="hipcc not on PATH")
def test_decode_kernels_keep_two_workgroups_per_cu(tmp_path):
    b = _build_module()
    src = ROOT / "hpc-ops_amd" / "csrc" / "attention_decode_v2.hip"
    out = tmp_path / "v2.s"
    cmd = ["hipcc"] + b._flags() + b._PER_FILE_FLAGS.get("attention_decode_v2", []) + ["-S", "--cuda-device-only", "-o", str(out), str(src)]
    subprocess.run(cmd, check=True, capture_output=True, text=True)
    asm = out.read_text()
    found = {}
    for m in re.finditer(r"\.name:\s+(\S+)\n(?:.*\n)*?\s+\.vgpr_count:\s+(\d+)\n\s+\.vgpr_spill_count:\s+(\d+)", asm):
        name, vgpr, spill = m.group(1), int(m.group(2)), int(m.group(3))
        for key, (what, max_vgpr, max_spill) in SHIPPED.items():
            if "decode2_kernel" + key in name:
                found[key] = (vgpr, spill)
                assert vgpr <= max_vgpr and spill <= max_spill, (what, name, vgpr, spill)
    assert set(found) == set(SHIPPED), (sorted(found), "missing an instantiation the launcher ships")
    # the headline form: nothing in scratch at all
    hl = re.search(r"_ZN3hpc7decode214decode2_kernelILi2ELb0ELb0ELb0ELb0ELb0ELb0EEEvNS0_4ArgsE:(.*?)s_endpgm", asm, re.S)
    assert hl is not None and "scratch_" not in hl.group(1)
