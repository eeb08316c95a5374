"""The product library (hpc/libhpc_amd.so) carries no development registers: no setter, no HPC_AMD_TUNING, every
variant branch folded away at compile time (csrc/hpc_dev.h).  The tests that pin kernel variants (marked `dev`) skip
their variant cases against the product; here they run - all of them, every case - in ONE subprocess against the
development build (HPC_AMD_DEV=1 -> hpc/libhpc_amd_dev.so, same sources compiled with -DHPC_DEV)."""
import ctypes
import os
import subprocess
import sys
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parent.parent


def _exported_symbols(path):
    """names in the ELF dynamic symbol table (nm -D; ctypes' hasattr is an exact dlsym and cannot enumerate)"""
    out = subprocess.run(["nm", "-D", "--defined-only", str(path)], capture_output=True, text=True, check=True).stdout
    return {line.split()[-1] for line in out.splitlines() if line.strip()}


def test_product_library_has_no_development_registers():
    """no exported symbol of the product starts with hpc_dev_ (the old probe named
    `hpc_dev_allreduce_loopback`, which is not a symbol of either build - the real ones end in _ht / _ll - so it could
    never fail); the development build is checked to export exactly those names, so the scan itself is pinned."""
    prod = _exported_symbols(ROOT / "hpc-ops_amd" / "hpc" / "libhpc_amd.so")
    leaked = sorted(s for s in prod if s.startswith("hpc_dev_") or s.startswith("hpc_debug_"))
    assert not leaked, f"development symbols exported by the product library: {leaked}"
    dev = _exported_symbols(ROOT / "hpc-ops_amd" / "hpc" / "libhpc_amd_dev.so")
    for sym in ("hpc_dev_tuning_set", "hpc_dev_tuning_get", "hpc_dev_decode_prof_buffer", "hpc_dev_allreduce_loopback_ht",
                "hpc_dev_allreduce_loopback_ll"):
        assert sym in dev, sym + " missing from the development build"
        assert sym not in prod
    assert any(s.startswith("hpc_") for s in prod)  # the scan sees the C-ABI at all
    blob = (ROOT / "hpc-ops_amd" / "hpc" / "libhpc_amd.so").read_bytes()
    assert b"HPC_AMD_TUNING" not in blob
    header = (ROOT / "include" / "hpc_amd.h").read_text()
    assert "tuning" not in header and "hpc_dev" not in header


def test_development_library_holds_64_registers():
    """keys 16-31 were silently dropped by a 16-entry table once and key 32 by a 32-entry one, which turned A/B runs
    into no-ops."""
    lib = ctypes.CDLL(str(ROOT / "hpc-ops_amd" / "hpc" / "libhpc_amd_dev.so"))
    for key in (0, 15, 16, 17, 20, 31, 32, 63):
        assert lib.hpc_dev_tuning_set(key, 7 + key) == 0
        assert lib.hpc_dev_tuning_get(key) == 7 + key
        assert lib.hpc_dev_tuning_set(key, 0) == 0
    assert lib.hpc_dev_tuning_set(64, 1) == -2 and lib.hpc_dev_tuning_get(64) == 0
    assert lib.hpc_dev_tuning_set(-1, 1) == -2


@pytest.mark.gate_free  # the inner tests take the GPU gate of tests/conftest.py themselves (an exclusive inner test would
@pytest.mark.gpu        # otherwise wait for ever for the shared lock this test holds)
def test_dev_build_suite():
    """every test marked `dev`, against the development build (its own xdist workers: the suite's wall-clock)"""
    if os.environ.get("HPC_AMD_DEV") == "1":
        pytest.skip("already inside the development-build run")
    env = {k: v for k, v in os.environ.items() if not k.startswith("PYTEST_XDIST")}
    env["HPC_AMD_DEV"] = "1"
    r = subprocess.run([sys.executable, "-m", "pytest", str(ROOT / "tests"), "-m", "gpu and dev", "-q", "-x", "-n", "4",
                        "-p", "no:cacheprovider"], env=env, cwd=str(ROOT), capture_output=True, text=True, timeout=3000)
    tail = "\n".join((r.stdout + r.stderr).splitlines()[-25:])
    print(tail)
    assert r.returncode == 0, tail
