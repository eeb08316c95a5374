"""Memory-checked deterministic replay of every public hpc.* call - the MI355X counterpart of the reference's
sanitizer mode (reference conftest.py:14-152: `SANITIZER_CHECK=memcheck,...` monkey-patches every public `hpc.*`
function, dumps its arguments and result, and replays the call under compute-sanitizer asserting byte-equal outputs).

There is no compute-sanitizer on ROCm, so the replay process checks what can be checked from the outside:
  * determinism: the replayed call - fresh process, arguments restored from the dump taken BEFORE the call - must give
    byte-identical results and leave byte-identical arguments (a race, a read of uninitialised scratch, a stale
    counter or a stray write into an input shows up as a difference);
  * guard bands: in the replay every distinct storage behind the tensor arguments is re-created inside a fresh
    allocation with `GUARD` poisoned bytes on either side (views keep their sizes / strides / offsets), and the
    bands must be intact afterwards - an out-of-bounds store within 64 KB of any argument (caller-provided outputs,
    KV caches, task maps, workspaces ...) is caught;
  * allocator isolation: the replay runs with PYTORCH_NO_HIP_MEMORY_CACHING=1, so tensors the op allocates itself
    (outputs, scratch) are separate hipMalloc allocations - an access far outside one faults instead of landing in a
    neighbouring block of the caching allocator's pool.

Enable with `HPC_REPLAY_CHECK=1 pytest tests -m gpu ...` (tests/conftest.py installs the hook; every call is replayed:
slow, meant for a subset), or call `install()` / `replay_call()` directly.  tests/test_replay_check.py keeps the harness
alive on a few ops and proves that it catches a planted out-of-bounds store and a planted nondeterminism."""
import os
import subprocess
import sys
import tempfile
import types
from pathlib import Path

import torch

ROOT = Path(__file__).resolve().parent.parent
GUARD = 1 << 16
POISON = 0xA5


def _map(obj, fn):
    if isinstance(obj, torch.Tensor):
        return fn(obj)
    if isinstance(obj, (list, tuple)):
        return type(obj)(_map(o, fn) for o in obj)
    if isinstance(obj, dict):
        return {k: _map(v, fn) for k, v in obj.items()}
    return obj


def _tensors(obj, out):
    _map(obj, lambda t: out.append(t) or t)
    return out


def guard_arguments(args, kwargs, device):
    """Re-create every distinct storage behind the tensor arguments inside a guarded allocation on `device`; returns the
    rebuilt (args, kwargs) and the list of (guarded buffer, payload bytes) to check afterwards."""
    rebuilt, guards = {}, []

    def move(t):
        if t.device.type != "cuda":
            return t
        st = t.untyped_storage()
        key = st.data_ptr()
        if key not in rebuilt:
            n = st.nbytes()
            buf = torch.full((n + 2 * GUARD,), POISON, dtype=torch.uint8, device=device)
            src = torch.empty(0, dtype=torch.uint8, device=t.device).set_(st, 0, (n,), (1,))
            buf[GUARD:GUARD + n].copy_(src)
            rebuilt[key] = buf
            guards.append((buf, n))
        buf = rebuilt[key]
        out = torch.empty(0, dtype=t.dtype, device=device)
        # same sizes / strides, storage offset shifted by the guard (in elements: GUARD is a multiple of every itemsize)
        out.set_(buf.untyped_storage(), t.storage_offset() + GUARD // t.element_size(), t.size(), t.stride())
        return out

    return _map(args, move), _map(kwargs, move), guards


def guards_intact(guards):
    bad = []
    for i, (buf, n) in enumerate(guards):
        lo, hi = buf[:GUARD], buf[GUARD + n:]
        if not bool((lo == POISON).all()) or not bool((hi == POISON).all()):
            where = "below" if not bool((lo == POISON).all()) else "above"
            bad.append(f"storage {i} ({n} bytes): guard band {where} the buffer was overwritten")
    return bad


def _equal(a, b, what, bad):
    if isinstance(a, torch.Tensor):
        if not isinstance(b, torch.Tensor) or a.shape != b.shape or a.dtype != b.dtype:
            bad.append(f"{what}: shape / dtype differs")
        elif a.numel() and not torch.equal(a.reshape(-1).contiguous().view(torch.uint8).cpu(),
                                           b.reshape(-1).contiguous().view(torch.uint8).cpu()):
            bad.append(f"{what}: bytes differ")
    elif isinstance(a, (list, tuple)):
        if not isinstance(b, (list, tuple)) or len(a) != len(b):
            bad.append(f"{what}: length differs")
        else:
            for i, (x, y) in enumerate(zip(a, b)):
                _equal(x, y, f"{what}[{i}]", bad)
    elif isinstance(a, dict):
        for k in a:
            _equal(a[k], b.get(k) if isinstance(b, dict) else None, f"{what}[{k!r}]", bad)
    elif a != b:
        bad.append(f"{what}: {a!r} != {b!r}")


_REPLAY = r"""
import sys
sys.path.insert(0, %(pkg)r); sys.path.insert(0, %(tests)r)
import torch
import hpc
import replay_check as rc
din = torch.load(%(before)r, weights_only=False)
dout = torch.load(%(after)r, weights_only=False)
dev = torch.device("cuda", 0)
args, kwargs, guards = rc.guard_arguments(din["args"], din["kwargs"], dev)
name = din["func_name"]  # "function" of the hpc package, or "module:function" (the planted faults of tests/replay_planted.py)
fn = getattr(__import__(name.split(":")[0]), name.split(":")[1]) if ":" in name else getattr(hpc, name)
ret = fn(*args, **kwargs)
torch.cuda.synchronize()
bad = rc.guards_intact(guards)
rc._equal(dout["ret"], ret, "result", bad)
rc._equal(dout["args"], args, "args after the call", bad)
rc._equal(dout["kwargs"], kwargs, "kwargs after the call", bad)
if bad:
    print("REPLAY-CHECK FAILED for %%s:\n  " %% name + "\n  ".join(bad))
    sys.exit(3)
print("replay ok:", din["func_name"], len(guards), "guarded storages")
"""


def replay_call(func_name, args, kwargs, call):
    """Run `call(*args, **kwargs)` here, then replay it in a fresh process from the arguments as they were before the
    call; raises AssertionError when the replay differs or a guard band was touched."""
    d = tempfile.mkdtemp(prefix=f"hpc_replay_{func_name}_", dir="/dev/shm" if os.path.isdir("/dev/shm") else None)
    before, after = os.path.join(d, "before.pt"), os.path.join(d, "after.pt")
    try:
        torch.save({"func_name": func_name, "args": args, "kwargs": kwargs}, before)
        ret = call(*args, **kwargs)
        torch.cuda.synchronize()
        torch.save({"func_name": func_name, "ret": ret, "args": args, "kwargs": kwargs}, after)
        code = _REPLAY % {"pkg": str(ROOT / "hpc-ops_amd"), "tests": str(ROOT / "tests"), "before": before, "after": after}
        env = dict(os.environ, PYTORCH_NO_HIP_MEMORY_CACHING="1", HPC_REPLAY_CHECK="0")
        r = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=900)
        assert r.returncode == 0, f"replay of hpc.{func_name} failed (rc {r.returncode}):\n{r.stdout[-3000:]}\n{r.stderr[-3000:]}"
        return ret
    finally:
        for f in (before, after):
            if os.path.exists(f):
                os.unlink(f)
        os.rmdir(d)


# calls whose arguments cannot be replayed in another process (communicator handles, symmetric memory)
SKIP = {"MulticastCommunicator", "MulticastHandle", "empty_multimem", "fuse_allreduce_rmsnorm_high_throughput",
        "fuse_allreduce_rmsnorm_low_latency", "lookup_peers", "release_decode_workspaces"}


def install(module=None):
    """Wrap every public function of the hpc package (reference conftest.py:85-138)."""
    import hpc

    module = module or hpc
    for name in dir(module):
        if name.startswith("_") or name.endswith("fake") or name in SKIP:
            continue
        fn = getattr(module, name)
        if not isinstance(fn, types.FunctionType):
            continue

        def wrapped(*args, __fn=fn, __name=name, **kwargs):
            has_cuda = any(t.is_cuda for t in _tensors((args, kwargs), []))
            # samplers that draw their own noise advance a per-process launch counter: not replayable in a fresh process
            own_noise = __name.startswith("fused_sampler") and kwargs.get("gumbel_noise") is None
            if not has_cuda or own_noise or torch.cuda.is_current_stream_capturing():
                return __fn(*args, **kwargs)
            return replay_call(__name, args, kwargs, __fn)

        setattr(module, name, wrapped)
