"""CPU (hipcc cross-compiles): the ISA of the 256 x 256 grouped GEMM (csrc/group_gemm_p8.hip) as build.py compiles it.

Round 6 found that this file's hand schedule does not survive every choice hipcc is free to make: with the SLP vectorizer on, the
ride-along body's rescale FMAs were paired into <2 x float> operations (v_mov splats + v_pk_fma_f32 in the section that feeds the
matrix pipe), the kernel went to 249 registers and 26 spilled SGPRs - and that build computed a wrong output row on every call
(profiles/round6_moe_ext_ab.txt).  build.py therefore compiles this file with -fno-slp-vectorize; this test pins what that
build must look like, so that a toolchain or flag change that brings the pattern back fails here and not as a nondeterministic
wrong answer on a GPU."""
import importlib.util
import re
import shutil
import subprocess
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parents[1]


def _build_module():
    spec = importlib.util.spec_from_file_location("hpc_amd_build_for_test", ROOT / "hpc-ops_amd" / "build.py")
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def test_build_flags_for_the_grouped_gemm():
    b = _build_module()
    assert "-fno-slp-vectorize" in b._PER_FILE_FLAGS.get("group_gemm_p8", [])


@pytest.mark.skipif(shutil.which("hipcc") is None, reason="hipcc not on PATH")
def test_grouped_gemm_isa_has_no_spills_and_no_packed_rescale(tmp_path):
    b = _build_module()
    src = ROOT / "hpc-ops_amd" / "csrc" / "group_gemm_p8.hip"
    out = tmp_path / "p8.s"
    cmd = ["hipcc"] + b._flags() + b._PER_FILE_FLAGS["group_gemm_p8"] + ["-S", "--cuda-device-only", "-o", str(out), str(src)]
    subprocess.run(cmd, check=True, capture_output=True, text=True)
    asm = out.read_text()
    kernels = re.findall(r"\.name:\s+(\S*gemm_fp8_p8_kernel\S*)", asm)
    assert len(set(kernels)) >= 6, kernels
    # per kernel: registers within the two-waves-per-SIMD budget, nothing in scratch
    for m in re.finditer(r"\.name:\s+(\S+)\n(?:.*\n)*?\s+\.vgpr_count:\s+(\d+)\n\s+\.vgpr_spill_count:\s+(\d+)", asm):
        name, vgpr, spill = m.group(1), int(m.group(2)), int(m.group(3))
        if "gemm_fp8_p8_kernel" in name:
            assert vgpr <= 256 and spill == 0, (name, vgpr, spill)
    assert asm.count("scratch_store") == 0 and asm.count("scratch_load") == 0
    # the rescale is scalar FMAs: no packed fp32 FMA anywhere in the file (the per-tensor epilogue's `tot *= scale` is a packed
    # multiply by construction - a vector times a scalar - and stays)
    assert "v_pk_fma_f32" not in asm
    # the pattern of the bad build: a VALU write into a source register of the MFMA issued just before it
    lines = [l.split(";")[0].strip() for l in asm.splitlines()]
    lines = [l for l in lines if l and not l.startswith(".") and not l.endswith(":")]

    def regs(tok):
        m = re.match(r"v\[(\d+):(\d+)\]", tok)
        if m:
            return set(range(int(m.group(1)), int(m.group(2)) + 1))
        m = re.match(r"v(\d+)$", tok)
        return {int(m.group(1))} if m else set()

    bad = []
    for i, l in enumerate(lines):
        if l.startswith("v_mfma"):
            ops = [o.strip() for o in l.split(None, 1)[1].split(",")]
            src_regs = regs(ops[1]) | regs(ops[2])
            for n in lines[i + 1: i + 3]:
                if n.startswith("v_mfma") or n.startswith("s_barrier") or n.startswith("s_cbranch"):
                    break
                if n.startswith("v_") and not n.startswith(("v_cmp", "v_readlane", "v_readfirstlane")):
                    dst = n.split(None, 1)[1].split(",")[0].strip()
                    if regs(dst) & src_regs:
                        bad.append((l, n))
    assert not bad, bad[:4]
