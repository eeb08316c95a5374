"""torch.compile-ability of the surface: every op with tensor outputs has a fake (meta) implementation
(`torch.library.register_fake` in hpc/*.py, like the reference's hpc/*.py) whose arity matches the op's schema and whose
output shapes / dtypes are the ones the C++ entries allocate (csrc/torch_*.cpp).  Runs on the CPU box: FakeTensorMode
carries "cuda" fake tensors without a device."""
import pytest
import torch
from torch._subclasses import FakeTensorMode

import hpc  # noqa: F401

F8, BF, F32, I32, I64, U8 = torch.float8_e4m3fn, torch.bfloat16, torch.float32, torch.int32, torch.int64, torch.uint8


def T(*shape, dtype=BF):
    return torch.empty(shape, dtype=dtype, device="cuda")


def meta(out):
    outs = out if isinstance(out, (tuple, list)) else (out,)
    return [(tuple(t.shape), t.dtype) for t in outs]


def _cases():
    B, Hq, Hkv, D, P, nblk, mb = 4, 32, 4, 128, 64, 40, 6
    tok, E, k, H, I = 24, 8, 2, 512, 256
    kc, vc = T(nblk, P, Hkv, D, dtype=F8), T(nblk, P, Hkv, D, dtype=F8)
    kcb, vcb = T(nblk, P, Hkv, D), T(nblk, P, Hkv, D)
    ids, lens = T(B, mb, dtype=I32), T(B, dtype=I32)
    ops = torch.ops.hpc
    yield "attention_decode_bf16", lambda: ops.attention_decode_bf16(T(B, Hq, D), kcb, vcb, ids, lens, 0, True, True, None, None, None), \
        [((B, Hq, D), BF)]
    yield "attention_decode_fp8", lambda: ops.attention_decode_fp8(T(B, Hq, D, dtype=F8), kc, vc, ids, lens, T(B, Hq, dtype=F32),
                                                                   T(1, dtype=F32), T(1, dtype=F32), 0, True, 1, True, None, None, None), \
        [((B, Hq, D), BF)]
    cu = T(B + 1, dtype=I32)
    yield "attention_with_kvcache_prefill_fp8", lambda: ops.attention_with_kvcache_prefill_fp8(
        T(tok, Hq, D, dtype=F8), kc, vc, T(B, Hq, 128, dtype=F32), T(1, dtype=F32), T(1, dtype=F32), cu, ids, lens, 100, 1, None), \
        [((tok, Hq, D), BF)]
    yield "attention_with_kvcache_blocksparse_prefill_fp8", lambda: ops.attention_with_kvcache_blocksparse_prefill_fp8(
        T(tok, Hq, D, dtype=F8), kc, vc, T(B, Hq, 128, dtype=F32), T(1, dtype=F32), T(1, dtype=F32), cu, ids, lens, 100, 1,
        T(B, Hq, 1, 3, dtype=U8), None), [((tok, Hq, D), BF)]
    yield "attention_with_kvcache_prefill_bf16", lambda: ops.attention_with_kvcache_prefill_bf16(T(tok, Hq, D), kcb, vcb, cu, ids, lens,
                                                                                                 100, None), [((tok, Hq, D), BF)]
    yield "attention_prefill_bf16", lambda: ops.attention_prefill_bf16(T(tok, Hq, D), T(tok, Hkv, D), T(tok, Hkv, D), lens, cu, 100, None), \
        [((tok, Hq, D), BF)]
    x8, xs = T(tok, H, dtype=F8), T(tok, H // 128, dtype=F32)
    guw, guws = T(E, 2 * I, H, dtype=F8), T(E, 2 * I // 128, 4, dtype=F32)
    dw, dws = T(E, H, I, dtype=F8), T(E, H // 128, 4, dtype=F32)
    tids, tsc = T(tok, k, dtype=I32), T(tok, k, dtype=F32)
    for name in ("fuse_moe_blockwise_fp8", "fuse_moe_blockwise"):
        yield name, (lambda n=name: getattr(ops, n)(x8, xs, guw, guws, dw, dws, tids, tsc, None, 0, E, None)), [((tok, H), BF)]
    es = T(E, dtype=F32)
    for name in ("fuse_moe", "fuse_moe_pertensor_fp8"):
        yield name, (lambda n=name: getattr(ops, n)(x8, guw, dw, es, es, T(1, dtype=F32), tids, tsc, None, 0, E, True, None)), \
            [((tok, H), BF)]
    yield "count_and_gather", lambda: ops.count_and_gather(x8, tids, E, 0, I, 8), \
        [((tok * k, H), F8), ((tok * k, I), BF), ((tok, k), I32), ((E,), I32), ((E + 1,), I32), ((E,), I32), ((E + 1,), I32),
         ((E * 2, 128), torch.int8), ((E * 2, 128), torch.int8)]
    yield "reduce", lambda: ops.reduce(T(tok * k, H), tids, tsc, None), [((tok, H), BF)]
    sl, cs = T(E, dtype=I32), T(E + 1, dtype=I32)
    yield "group_gemm_blockwise_fp8", lambda: ops.group_gemm_blockwise_fp8(x8, guw, sl, cs, T(H // 128, 64, dtype=F32), guws, 8, None,
                                                                           None, None), [((tok, 2 * I), BF)]
    for name in ("group_gemm_fp8", "group_gemm_pertensor_fp8"):
        yield name, (lambda n=name: getattr(ops, n)(x8, guw, sl, cs, es, 8, None, None, None)), [((tok, 2 * I), BF)]
    yield "group_gemm_fp8_cp_async", lambda: ops.group_gemm_fp8_cp_async(x8, guw, es, sl, cs, sl, cs), [((tok, 2 * I), BF)]
    yield "group_gemm_fp8_scatter_cp_async", lambda: ops.group_gemm_fp8_scatter_cp_async(x8, guw, es, T(40, dtype=I32), sl, cs, sl, cs), \
        [((40, 2 * I), BF)]
    yield "reformat_x_scale", lambda: ops.reformat_x_scale(T(E * 16, 4, dtype=F32), sl, cs, None, 16), [((4, E * 16), F32)]
    gu = T(tok, 2 * I)
    yield "act_mul_and_quant", lambda: ops.act_mul_and_quant(gu, T(1, dtype=F32), True, None), [((tok, I), F8)]
    yield "scaled_fp8_quant", lambda: ops.scaled_fp8_quant(T(3, 5, dtype=torch.float16), T(dtype=F32), None), [((3, 5), F8), ((), F32)]
    yield "masked_act_mul_and_quant", lambda: ops.masked_act_mul_and_quant(gu, T(1, dtype=F32), sl, None), [((tok, I), F8)]
    yield "masked_act_mul_and_blockwise_quant", lambda: ops.masked_act_mul_and_blockwise_quant(gu, sl, None, None), \
        [((tok, I), F8), ((tok, I // 128), F32)]
    yield "fused_rmsnorm_with_scale", lambda: ops.fused_rmsnorm_with_scale(T(7, H), T(H), T(2, dtype=F32), 1e-6, True), \
        [((7, H), F8), ((7, H), F32), ((7, H), F8)]
    hid = (Hq + 2 * Hkv) * D
    cossin, nseq, qidx, kvi = T(512, D, dtype=F32), T(B, dtype=I32), T(B + 1, dtype=I32), T(B, mb, dtype=I32)
    yield "rope_norm_store_kv", lambda: ops.rope_norm_store_kv(kcb, vcb, T(tok, hid), cossin, nseq, qidx, kvi, True, None, None), \
        [((tok, Hq, D), BF)]
    yield "rope_norm_store_kv_fp8 (decode, dynamic q scale)", lambda: ops.rope_norm_store_kv_fp8(
        kc, vc, T(B, hid), cossin, nseq, qidx, kvi, False, T(1, dtype=F32), T(1, dtype=F32), 1, 0, None, None, None, None), \
        [((B, Hq, D), F8), ((B, Hq), F32), ((B, Hkv), I32)]
    yield "rope_norm_store_kv_fp8 (prefill, dynamic q scale)", lambda: ops.rope_norm_store_kv_fp8(
        kc, vc, T(tok, hid), cossin, nseq, qidx, kvi, True, T(1, dtype=F32), T(1, dtype=F32), 1, 100, None, None, None, None), \
        [((tok, Hq, D), F8), ((B, Hq, 128), F32), ((B, Hkv), I32)]
    yield "gemm_bf16xfp32", lambda: ops.gemm_bf16xfp32(T(16, 1024), T(256, 1024), T(256, 1024), 1 / 256, True, True, None), [((16, 256), F32)]
    yield "topk_router", lambda: ops.topk_router(T(16, 64, dtype=F32), 8, True, None, None), [((16, 8), I32), ((16, 8), F32)]
    V = 4096
    yield "fused_sampler", lambda: ops.fused_sampler(T(3, V, dtype=F32), None, None, None, 0.0, None, 0.7, 2, None, 20, None, 0.9, 32, None, 7), \
        [((3, 1), I32)]
    yield "fused_sampler_temperature_sample", lambda: ops.fused_sampler_temperature_sample(T(3, V, dtype=F32), None, 0.7, None, None, 7), \
        [((3, 1), I32)]


CASES = None


def _table():
    global CASES
    if CASES is None:
        with FakeTensorMode():
            CASES = {name: (fn, want) for name, fn, want in _cases()}
    return CASES


def _names():
    # collected without entering fake mode twice: the generator only builds lambdas and expected metadata
    with FakeTensorMode():
        return [name for name, _, _ in _cases()]


@pytest.mark.parametrize("name", _names())
def test_fake_implementation(name):
    with FakeTensorMode():
        for n, fn, want in _cases():
            if n == name:
                assert meta(fn()) == want
                return
    raise AssertionError(name)


def test_every_op_with_tensor_outputs_is_covered():
    covered = {n.split(" ")[0] for n in _names()}
    names = {n.split("::")[1].split(".")[0] for n in torch._C._dispatch_get_all_op_names() if n.startswith("hpc::")}
    # not compute ops: strings / internal housekeeping / the scheduler (host-visible output, no fake in the reference either) /
    # in-place collectives without outputs
    skip = {"version", "built_json", "_release_decode_workspaces", "_decode_workspaces", "assign_attention_decode_task",
            "fuse_allreduce_rmsnorm_high_throughput", "fuse_allreduce_rmsnorm_low_latency"}
    assert names - skip <= covered, names - skip - covered
