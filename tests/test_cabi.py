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


def _prototypes():
    """name -> (return type, [parameter types]) parsed from include/hpc_amd.h."""
    text = (ROOT / "include" / "hpc_amd.h").read_text()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    text = re.sub(r"//[^\n]*", "", text)
    protos = {}
    for ret, name, params in re.findall(r"([A-Za-z_][A-Za-z0-9_ \*]*?)\b(hpc_[a-z0-9_]+)\s*\(([^;{]*?)\)\s*;", text):
        params = params.strip()
        plist = [] if params in ("", "void") else [" ".join(p.split()) for p in params.split(",")]
        protos[name] = (" ".join(ret.split()), plist)
    return protos


def _kind(ctype: str) -> str:
    """C parameter / return type -> 'ptr' | 'i32' | 'i64' | 'f32'."""
    t = ctype.replace("const", " ").strip()
    if "*" in t or re.search(r"\bhipStream_t\b|\bhpc_stream_t\b", t):
        return "ptr"
    base = t.split()[:-1] if len(t.split()) > 1 else t.split()  # drop the parameter name
    base = " ".join(base)
    if re.search(r"\b(int64_t|uint64_t|long long|size_t|long)\b", base):
        return "i64"
    if re.search(r"\bfloat\b", base):
        return "f32"
    if re.search(r"\b(int|int32_t|uint32_t|unsigned)\b", base):
        return "i32"
    raise AssertionError(f"unclassified C type: {ctype!r}")


def _ckind(ct) -> str:
    if ct is None:
        return "void"
    if ct in (ctypes.c_void_p, ctypes.c_char_p) or hasattr(ct, "contents") or issubclass(ct, ctypes._Pointer):
        return "ptr"
    return {ctypes.c_int: "i32", ctypes.c_uint: "i32", ctypes.c_int64: "i64", ctypes.c_uint64: "i64",
            ctypes.c_longlong: "i64", ctypes.c_float: "f32", ctypes.c_size_t: "i64"}[ct]


def test_header_is_valid_c99_and_cxx(tmp_path):
    import shutil
    import subprocess

    src = tmp_path / "use_header.c"
    src.write_text('#include "hpc_amd.h"\nint main(void) { return 0; }\n')
    inc = str(ROOT / "include")
    if shutil.which("gcc"):
        subprocess.run(["gcc", "-std=c99", "-Wall", "-Wextra", "-pedantic", "-Werror", "-I", inc, "-fsyntax-only", str(src)],
                       check=True)
    if shutil.which("g++"):
        subprocess.run(["g++", "-std=c++17", "-Wall", "-Werror", "-I", inc, "-fsyntax-only", "-x", "c++", str(src)],
                       check=True)


def test_ctypes_signatures_match_the_header():
    """Every entry point the Python side binds has the argument kinds (pointer / int / int64 / float) and
    count the header declares - a mismatch here is a silent ABI bug (truncated strides, shifted arguments)."""
    from hpc import _C

    protos = _prototypes()
    assert len(protos) >= 40
    bad = []
    for name, (ret, params) in sorted(protos.items()):
        fn = getattr(_C.lib, name)
        if fn.argtypes is None:  # declared in the header but not bound from Python: nothing to compare
            continue
        want = [_kind(p) for p in params]
        have = [_ckind(a) for a in fn.argtypes]
        if want != have:
            bad.append((name, want, have))
        want_ret = "void" if ret.strip() == "void" else _kind(ret + " r")
        have_ret = _ckind(fn.restype)
        if want_ret != have_ret:
            bad.append((name + " (return)", want_ret, have_ret))
    assert not bad, "\n".join(f"{n}: header {w} vs ctypes {h}" for n, w, h in bad)


def test_prefill_refuses_strides_the_kernel_cannot_address():
    """Host-side argument checks run before any device call: the FP8 prefill kernels form page offsets as unsigned
    32 x 32 -> 64 products, so a page (block) stride of 4 GB or more is refused - HPC_ERR_UNSUPPORTED, not a launch."""
    from ctypes import c_int, c_int64, c_void_p

    lib = ctypes.CDLL(str(ROOT / "hpc-ops_amd" / "hpc" / "libhpc_amd.so"))
    fn = lib.hpc_attention_with_kvcache_prefill_fp8_async
    fn.restype = c_int
    fn.argtypes = [c_void_p] * 10 + [c_int] * 12 + [c_int64] * 9 + [c_void_p]
    p = c_void_p(4096)  # never dereferenced on the host
    ints = (1, 2, 128, 128, 128, 128, 8, 1, 64, 4, 1024, 1024)  # quant_type 1, B 2, Sq 128, D 128/128, 8/1 heads, pages of 64
    ok_strides = (64 * 128, 128, 128, 64 * 128, 128, 128, 0, 0, 0)
    big = 1 << 32
    for which in (0, 3):  # K block stride, V block stride
        strides = list(ok_strides)
        strides[which] = big
        assert fn(*([p] * 10), *ints, *strides, None) == -1
    strides = list(ok_strides)
    strides[1] = big // 64  # token stride x page size reaches 4 GB
    assert fn(*([p] * 10), *ints, *strides, None) == -1
