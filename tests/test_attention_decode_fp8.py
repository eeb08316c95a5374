"""Parity of hpc.attention_decode_fp8 (HIP via the C-ABI) with the CPU oracles.
Generators / tolerances follow reference
tests/test_attention_decode_qpertoken_perhead_kvpertensor_fp8.py:82-298 (atol 0.2) and
tests/test_attention_decode_qkpertoken_perhead_vperhead_fp8.py:262-470 (atol 0.1)."""
import math

import pytest
import torch

from utils import allclose, dev_set


def _case(num_batch, num_seq_q, lens_before, block_size, kv_head_q_head, k_per_token, seed=41):
    torch.manual_seed(seed)
    num_head_kv, num_head_q = kv_head_q_head
    D = 128
    q = torch.randn((num_batch * num_seq_q, num_head_q, D), dtype=torch.bfloat16) / math.sqrt(D)
    q_scale = q.float().abs().max(-1)[0] / 10
    q8 = (q / q_scale[:, :, None]).to(torch.float8_e4m3fn)
    nblocks = (lens_before + num_seq_q + block_size - 1) // block_size
    total_blocks = int(nblocks.sum())
    max_num_blocks = int(total_blocks * 1.2) + 4
    scale_rows = block_size * 4 // D if k_per_token else 0
    gen_dev = "cuda" if torch.cuda.is_available() else "cpu"
    if k_per_token:
        kv = torch.randn(max_num_blocks, 2, block_size + scale_rows, num_head_kv, D, dtype=torch.bfloat16,
                         device=gen_dev).cpu()
    else:
        kv = (torch.randn(max_num_blocks, 2, block_size, num_head_kv, D, dtype=torch.bfloat16, device=gen_dev)
              / math.sqrt(D)).cpu()
    packed = torch.randperm(max_num_blocks)[:total_blocks].to(torch.int32)
    block_ids = torch.full((num_batch, int(nblocks.max())), -999999, dtype=torch.int32)
    cu = 0
    for i in range(num_batch):
        nb = int(nblocks[i])
        block_ids[i, :nb] = packed[cu : cu + nb]
        cu += nb
    return q8, q_scale, kv, block_ids, nblocks


def _run(num_batch, num_seq_q, lens_before, block_size, kv_head_q_head, k_per_token, new_kv_included,
         use_dynamic_sched, kvcache_shape, atol):
    import hpc
    from oracle import attention as oattn

    num_head_kv, num_head_q = kv_head_q_head
    q8, q_scale, kv, block_ids, nblocks = _case(
        num_batch, num_seq_q, lens_before, block_size, kv_head_q_head, k_per_token)
    if k_per_token:
        kc, _ = oattn.quant_paged_cache_pertoken(kv[:, 0], block_size)
        vc, v_scale = oattn.quant_paged_cache_perhead(kv[:, 1], block_size)
        kv8 = torch.empty_like(kv, dtype=torch.float8_e4m3fn)
        kv8[:, 0] = kc
        kv8[:, 1] = vc
        k_scale = kv8[:, 0, block_size:]
        qt = hpc.QuantType.QPERTOKEN_PERHEAD_KPERTOKEN_PERHEAD_VPERHEAD
    else:
        kv8 = kv.to(torch.float8_e4m3fn)
        k_scale = torch.randn(1, dtype=torch.float32)
        v_scale = torch.randn(1, dtype=torch.float32)
        qt = hpc.QuantType.QPERTOKEN_PERHEAD_KPERTENSOR_VPERTENSOR
    gt = oattn.ref_attn_fp8(q8, kv8[:, :, :block_size], block_ids, nblocks, num_seq_q, lens_before,
                            q_scale, k_scale, v_scale, k_per_token)

    kv_dev = kv8.cuda()
    if kvcache_shape == "HND":
        kv_dev = kv_dev.view(torch.uint8).permute(0, 1, 3, 2, 4).contiguous().permute(0, 1, 3, 2, 4).view(
            torch.float8_e4m3fn)
    kcache, vcache = kv_dev[:, 0, :block_size], kv_dev[:, 1, :block_size]
    ks_dev = kv_dev[:, 0, block_size:] if k_per_token else k_scale.cuda()
    lens_dev = lens_before.cuda()
    lens_in = lens_dev + num_seq_q if new_kv_included else lens_dev
    task_map = None
    if use_dynamic_sched:
        task_map = hpc.get_attention_decode_task_workspace(
            num_batch, int(lens_before.max()) + num_seq_q, num_head_kv, min_process_len=1024)
        hpc.assign_attention_decode_task(lens_in, task_map, num_head_kv, num_seq_q, new_kv_included,
                                         min_process_len=1024)
    my = hpc.attention_decode_fp8(
        q8.cuda(), kcache, vcache, block_ids.cuda(), lens_in, q_scale.cuda(), ks_dev, v_scale.cuda(),
        mtp=num_seq_q - 1, new_kv_included=new_kv_included, quant_type=qt, splitk=True,
        task_map=task_map)
    torch.cuda.synchronize()
    assert my.dtype == torch.bfloat16
    assert allclose(gt, my.cpu(), atol=atol)


@pytest.mark.gpu
@pytest.mark.parametrize("num_batch", [1, 16, 50])  # (200: test_attn_fp8_reference_grid_batch_200)
@pytest.mark.parametrize("num_seq_q", [1, 2, 3, 4])
@pytest.mark.parametrize("max_seq_kv", [1024, 4096])
@pytest.mark.parametrize("kv_head_q_head", [(1, 8), (4, 32)])
@pytest.mark.parametrize("use_dynamic_sched", [True, False])
@pytest.mark.parametrize("kvcache_shape", ["NHD", "HND"])
def test_attn_fp8_kvpertensor(num_batch, num_seq_q, max_seq_kv, kv_head_q_head, use_dynamic_sched, kvcache_shape):
    """The reference grid (tests/test_attention_decode_qpertoken_perhead_kvpertensor_fp8.py:263-273): num_seq_q
    1 ... 4 x max_seq_kv 1024 / 4096 x 1 / 8 and 4 / 32 heads x dynamic / static schedule x NHD / HND pages."""
    if num_batch == 50 and max_seq_kv == 4096 and kvcache_shape == "HND":
        pytest.skip("sampling: the 50 x 4096 cases run on NHD pages (the suite's wall-clock)")
    torch.manual_seed(41)
    lens = torch.randint(1, max_seq_kv, (num_batch,), dtype=torch.int32)
    _run(num_batch, num_seq_q, lens, 64, kv_head_q_head, False, True, use_dynamic_sched,
         kvcache_shape, 0.2)


@pytest.mark.gpu
@pytest.mark.parametrize("num_batch", [1, 16])
@pytest.mark.parametrize("num_seq_q", [1, 2, 4])
@pytest.mark.parametrize("kv_head_q_head", [(1, 8), (4, 32), (2, 8)])
@pytest.mark.parametrize("use_dynamic_sched", [True, False])
@pytest.mark.parametrize("kvcache_shape", ["NHD", "HND"])
def test_attn_fp8_kpertoken(num_batch, num_seq_q, kv_head_q_head, use_dynamic_sched, kvcache_shape):
    torch.manual_seed(41)
    lens = torch.randint(1, 2048, (num_batch,), dtype=torch.int32)
    _run(num_batch, num_seq_q, lens, 64, kv_head_q_head, True, True, use_dynamic_sched,
         kvcache_shape, 0.1)


@pytest.mark.gpu
@pytest.mark.parametrize("k_per_token,block_size", [(False, 16), (False, 32), (True, 32)])
def test_attn_fp8_small_pages_and_split(k_per_token, block_size):
    lens = torch.tensor([9000, 3, 130, 65, 2049], dtype=torch.int32)
    _run(5, 2, lens, block_size, (2, 16), k_per_token, False, True, "NHD", 0.2 if not k_per_token else 0.1)


@pytest.mark.dev
@pytest.mark.gpu
@pytest.mark.parametrize("k_per_token", [False, True])
@pytest.mark.parametrize("num_seq_q", [1, 3])
@pytest.mark.parametrize("solo", [True, False])
def test_attn_fp8_bins_of_short_requests(k_per_token, num_seq_q, solo):
    """Many short requests + one long one: bins packed with 1-4 tile tasks run one task per wave
    (no workgroup merge); tuning key 5 = 1 forces the team path on the same inputs."""
    import hpc

    lens = torch.tensor([3, 64, 65, 128, 200, 250, 17, 1] * 6 + [5000], dtype=torch.int32)
    dev_set(5, 0 if solo else 1)
    try:
        _run(len(lens), num_seq_q, lens, 64, (2, 16), k_per_token, True, True, "NHD", 0.1 if k_per_token else 0.2)
    finally:
        dev_set(5, 0)


@pytest.mark.dev
@pytest.mark.gpu
@pytest.mark.parametrize("num_seq_q,heads,block_size", [(1, (8, 64), 64), (2, (4, 32), 32), (4, (2, 8), 16), (1, (4, 16), 64)])
def test_attn_fp8_head_pair_kernel_hnd_form(num_seq_q, heads, block_size):
    """Development key 55 = 1: HND pages on the head-pair kernel (a load instruction = 8 tokens x 128 B of one head, the stage
    image and everything behind it the NHD form's) - measured a wash against the first-generation kernel, so the product keeps
    that one (profiles/round6_decode_ab.txt, call 4).  Split requests, short ones, empty ones, pages of 16 / 32 / 64 tokens."""
    lens = torch.tensor([20000, 3, 9000, 130, 64, 63, 65, 4097, 700, 1, 127, 129, 31000, 2, 0, 255, 256, 257], dtype=torch.int32)
    dev_set(55, 1)
    try:
        for _ in range(2):  # twice: the arrival counters are left zero
            _run(len(lens), num_seq_q, lens, block_size, heads, False, True, True, "HND", 0.2)
    finally:
        dev_set(55, 0)


@pytest.mark.dev
@pytest.mark.gpu
@pytest.mark.parametrize("key", [0, 1])
def test_attn_fp8_underloaded_launch_does_not_split_short_requests(key):
    """A launch with fewer tiles than it could spread (here 40 requests of 200 ... 700 tokens on 4 kv heads: the plan's floor of
    8 tiles sizes the ranges) moves every range boundary that falls inside a request of <= 16 tiles to that request's nearer
    end (round 6, attention_decode_v2.hip: `snap`): nothing is split, so no fp32 partial is written - the poisoned partial
    region of the cached workspace stays untouched.  Development key 61 = 1 keeps the split plan of rounds 2-5 (partials
    appear).  Both equal the oracle; the reference benchmark's uniform_512 case went 31.9 -> 23.5 us on 8 / 64 heads."""
    import hpc
    from oracle import attention as oattn

    hpc.release_decode_workspaces()
    torch.manual_seed(7)
    lens = torch.randint(200, 700, (40,), dtype=torch.int32)
    heads, P = (4, 32), 64
    q8, q_scale, kv, block_ids, nblocks = _case(40, 1, lens, P, heads, False)
    kv8 = kv.to(torch.float8_e4m3fn)
    ks, vs = torch.rand(1) + 0.5, torch.rand(1) + 0.5
    gt = oattn.ref_attn_fp8(q8, kv8[:, :, :P], block_ids, nblocks, 1, lens, q_scale, ks, vs, False)
    kvd = kv8.cuda()
    lens_in = (lens + 1).cuda()
    args = (q8.cuda(), kvd[:, 0, :P], kvd[:, 1, :P], block_ids.cuda(), lens_in, q_scale.cuda(), ks.cuda(), vs.cuda())
    zero = hpc._C.lib.hpc_attention_decode_workspace_zero_bytes()
    tm = hpc.get_attention_decode_task_workspace(40, int(lens.max()) + 1, heads[0], min_process_len=64)
    hpc.assign_attention_decode_task(lens_in, tm, heads[0], 1, True, min_process_len=64)

    def run():
        y = hpc.attention_decode_fp8(*args, mtp=0, new_kv_included=True,
                                     quant_type=hpc.QuantType.QPERTOKEN_PERHEAD_KPERTENSOR_VPERTENSOR, splitk=True, task_map=tm)
        torch.cuda.synchronize()
        assert allclose(gt, y.cpu(), atol=0.2)

    dev_set(61, key)
    try:
        run()  # creates the cached scratch
        cached = torch.ops.hpc._decode_workspaces()
        assert len(cached) >= 1
        for ws in cached:
            ws[zero:].view(torch.int32).fill_(0x7FC12345)
        torch.cuda.synchronize()
        run()
        touched = sum(int((ws[zero:].view(torch.int32) != 0x7FC12345).sum()) for ws in cached)
        assert (touched == 0) if key == 0 else (touched > 0)
        run()  # the arrival counters were left zero
    finally:
        dev_set(61, 0)
        hpc.release_decode_workspaces()


@pytest.mark.dev
@pytest.mark.gpu
@pytest.mark.parametrize("num_seq_q,heads,block_size,shape,key", [(3, (8, 64), 64, "NHD", 0), (4, (4, 32), 32, "HND", 0), (4, (1, 8), 64, "NHD", 0),
                                                                  (3, (2, 16), 32, "NHD", 0), (2, (3, 24), 64, "NHD", 2), (1, (2, 8), 64, "HND", 2),
                                                                  (4, (3, 12), 32, "NHD", 2), (3, (2, 16), 64, "NHD", 1), (4, (4, 32), 64, "HND", 1)])
def test_attn_fp8_one_head_per_workgroup_form(num_seq_q, heads, block_size, shape, key, k_per_token=False):
    """Speculative steps with 17 ... 32 q rows per kv head (num_seq_q 3 / 4 at 8 q heads per kv head, 2 at 16) run ONE kv head per
    workgroup on the head-pair kernel's pipeline since round 6 (attention_decode_v2.hip, kSolo: 64-token wave-iterations of
    128-byte rows, both q-row halves on the same K / V; any page layout and head count).  Development key 60 = 2 sends every
    other per-tensor call there that the pair form does not take (odd head counts, HND pages); key 60 = 1 keeps the
    first generation's two-block form under test.  Split, short and empty requests; twice (arrival counters left zero)."""
    lens = torch.tensor([20000, 3, 9000, 130, 64, 63, 65, 4097, 700, 1, 127, 129, 31000, 2, 0, 255, 256, 257], dtype=torch.int32)
    dev_set(60, key)
    try:
        for _ in range(2):
            _run(len(lens), num_seq_q, lens, block_size, heads, k_per_token, True, True, shape, 0.1 if k_per_token else 0.2)
    finally:
        dev_set(60, 0)


@pytest.mark.dev
@pytest.mark.gpu
@pytest.mark.parametrize("num_seq_q,heads,block_size,shape,key", [(4, (4, 32), 64, "NHD", 0), (4, (1, 8), 32, "NHD", 0), (3, (2, 16), 64, "HND", 0),
                                                                  (3, (8, 64), 32, "HND", 0), (4, (4, 32), 64, "NHD", 1)])
def test_attn_fp8_one_head_per_workgroup_form_per_token_k_scales(num_seq_q, heads, block_size, shape, key):
    """quant_type 0 with 17 ... 32 q rows per kv head on the same form: the 64 K scales of a wave-iteration are two 128-byte
    pieces of the head's tail rows (one page of 64 tokens, or two of 32), V scales per head."""
    test_attn_fp8_one_head_per_workgroup_form(num_seq_q, heads, block_size, shape, key, k_per_token=True)


@pytest.mark.gpu
@pytest.mark.parametrize("mode", ["last_arriver", "poisoned_partials"])
def test_attn_fp8_split_request_merge_variants(mode):
    """Requests cut by range boundaries (second-generation kernel) are merged inside the launch by the chunk that
    arrives last.  The call's scratch is [arrival counters (zero on first use, left zero) | partial slots]: the
    partial region may hold anything - here words of the form (epoch << 16) | n for every epoch, the pattern a
    tagged counter would take for 'n chunks arrived' - and the same call twice must agree (counters left clean)."""
    import hpc

    lens = torch.tensor([20000, 3, 9000, 130, 64, 4097, 700, 31000], dtype=torch.int32)
    try:
        if mode == "poisoned_partials":
            _run(len(lens), 1, lens, 64, (4, 32), False, True, True, "NHD", 0.2)  # creates the cached scratch
            zero = hpc._C.lib.hpc_attention_decode_workspace_zero_bytes()
            cached = torch.ops.hpc._decode_workspaces()  # the host library's per-(device, stream) scratch buffers
            assert len(cached) >= 1
            for ws in cached:
                assert int(ws[:zero].view(torch.int32).abs().max()) == 0  # the call left its counters zero
                words = ws[zero:].view(torch.int32)
                idx = torch.arange(words.numel(), device=ws.device, dtype=torch.int32)
                words.copy_(((idx % 32767 + 1) << 16) | (idx % 7 + 1))
            torch.cuda.synchronize()
        for _ in range(2):
            _run(len(lens), 1, lens, 64, (4, 32), False, True, True, "NHD", 0.2)
    finally:
        pass


@pytest.mark.gpu
def test_attn_fp8_head_pair_kernel_honours_min_process_len():
    """The head-pair kernel plans its ranges inside the launch; of the task map it takes header int 6 = the scheduler
    call's min_process_len (csrc/sched_task_info.h): a workgroup's range is never shorter than that many KV tokens.
    With the bound above the batch's total length every head pair is ONE range: no request is cut, so no fp32 partial
    is written (the poisoned partial region of the cached workspace stays untouched); with the bound at 64 the 9000-token
    request is cut and partials appear.  Both answers equal the oracle."""
    import hpc
    from oracle import attention as oattn

    hpc.release_decode_workspaces()
    torch.manual_seed(5)
    lens = torch.tensor([9000, 3000, 500], dtype=torch.int32)
    heads, P = (4, 32), 64
    q8, q_scale, kv, block_ids, nblocks = _case(3, 1, lens, P, heads, False)
    kv8 = kv.to(torch.float8_e4m3fn)
    ks, vs = torch.rand(1) + 0.5, torch.rand(1) + 0.5
    gt = oattn.ref_attn_fp8(q8, kv8[:, :, :P], block_ids, nblocks, 1, lens, q_scale, ks, vs, False)
    kvd = kv8.cuda()
    lens_in = (lens + 1).cuda()
    args = (q8.cuda(), kvd[:, 0, :P], kvd[:, 1, :P], block_ids.cuda(), lens_in, q_scale.cuda(), ks.cuda(), vs.cuda())
    zero = hpc._C.lib.hpc_attention_decode_workspace_zero_bytes()

    def run(mpl):
        tm = hpc.get_attention_decode_task_workspace(3, int(lens.max()) + 1, heads[0], min_process_len=mpl)
        hpc.assign_attention_decode_task(lens_in, tm, heads[0], 1, True, min_process_len=mpl)
        assert int(tm.view(torch.int32)[6]) == mpl
        y = hpc.attention_decode_fp8(*args, mtp=0, new_kv_included=True,
                                     quant_type=hpc.QuantType.QPERTOKEN_PERHEAD_KPERTENSOR_VPERTENSOR, splitk=True, task_map=tm)
        torch.cuda.synchronize()
        assert allclose(gt, y.cpu(), atol=0.2)

    run(64)  # creates the cached scratch
    cached = torch.ops.hpc._decode_workspaces()
    assert len(cached) >= 1

    def poison():
        for ws in cached:
            ws[zero:].view(torch.int32).fill_(0x7FC12345)
        torch.cuda.synchronize()

    def touched():
        return sum(int((ws[zero:].view(torch.int32) != 0x7FC12345).sum()) for ws in cached)

    poison()
    run(16384)  # >= the 12503 tokens of the batch: one range per head pair, nothing is split
    assert touched() == 0
    run(64)     # the long request is cut at range boundaries: partials are written
    assert touched() > 0
    for ws in cached:
        assert int(ws[:zero].view(torch.int32).abs().max()) == 0
    hpc.release_decode_workspaces()


def _mixed_lens(num_batch, seed, hi):
    g = torch.Generator().manual_seed(seed)
    lens = torch.randint(1, hi, (num_batch,), dtype=torch.int32, generator=g)
    lens[torch.randperm(num_batch, generator=g)[: max(1, num_batch // 9)]] = 0  # finished / empty requests
    lens[0] = hi * 4  # one long request that is split over many ranges
    return lens


_MANY_HEADS = [(1, 64, (4, 32)), (2, 32, (2, 16)), (2, 16, (8, 32)), (1, 64, (1, 8)), (1, 32, (16, 64)), (4, 64, (8, 32))]


@pytest.mark.gpu
@pytest.mark.parametrize("num_batch,num_seq_q,block_size,heads",
                         [(nb,) + c for nb in (65, 200) for c in _MANY_HEADS] +
                         [(1000,) + c for c in (_MANY_HEADS[0], _MANY_HEADS[3], _MANY_HEADS[4])])  # (1000 requests: 7-11 s each)
def test_attn_fp8_many_requests(num_batch, num_seq_q, block_size, heads):
    """More than 64 requests: the in-kernel planner of the second-generation kernel puts several requests on a
    lane of its prefix scan and walks them (reference grid: num_batch = 200,
    tests/test_attention_decode_qpertoken_perhead_kvpertensor_fp8.py:263; kernel limit 1024).  Mixed lengths,
    empty requests, both page-lookup paths, 1 / 2 / 4 / 8 head pairs per workgroup (8 / 4 / 2 / 1 partner waves each);
    one kv head runs the first-generation kernel."""
    lens = _mixed_lens(num_batch, 1000 + num_batch, 600 if num_batch >= 1000 else 1500)
    _run(num_batch, num_seq_q, lens, block_size, heads, False, False, True, "NHD", 0.2)


@pytest.mark.gpu
@pytest.mark.parametrize("num_seq_q", [1, 2])
@pytest.mark.parametrize("kv_head_q_head", [(1, 8), (4, 32)])
def test_attn_fp8_reference_grid_batch_200(num_seq_q, kv_head_q_head):
    """The reference grid's largest batch (tests/test_attention_decode_qpertoken_perhead_kvpertensor_fp8.py:263)."""
    torch.manual_seed(41)
    lens = torch.randint(1, 1024, (200,), dtype=torch.int32)
    _run(200, num_seq_q, lens, 64, kv_head_q_head, False, True, True, "NHD", 0.2)


@pytest.mark.gpu
def test_attn_fp8_workspace_reused_across_shapes():
    """One cached scratch buffer serves calls of different shapes back to back: the arrival counters sit in a fixed
    region at its start, so a call can never find them on top of another call's partials."""
    for num_batch, heads, hi in ((64, (8, 64), 3000), (200, (2, 16), 900), (7, (4, 32), 20000), (64, (8, 64), 3000),
                                 (33, (1, 8), 5000)):
        lens = _mixed_lens(num_batch, num_batch, hi)
        _run(num_batch, 1, lens, 64, heads, False, True, True, "NHD", 0.2)


@pytest.mark.gpu
@pytest.mark.parametrize("block_size", [32, 64])
@pytest.mark.parametrize("num_seq_q", [1, 2, 4])
def test_attn_fp8_single_kv_head(block_size, num_seq_q):
    """One kv head (the reference benchmark's default 1/8 heads, bench_attention_decode_fp8.py:44-49) - served by the
    first-generation kernel (its token rows are contiguous there; a token-pair form of the second-generation kernel
    was built in round 3 and measured slower at this size: DESIGN.md) - split requests, short requests, odd lengths."""
    lens = torch.tensor([20000, 3, 9000, 130, 64, 63, 65, 4097, 700, 1, 127, 129, 31000, 2], dtype=torch.int32)
    _run(len(lens), num_seq_q, lens, block_size, (1, 8 if num_seq_q <= 2 else 4), False, True, True, "NHD", 0.2)


@pytest.mark.dev
@pytest.mark.gpu
@pytest.mark.parametrize("num_seq_q,heads,k_per_token,shape", [(1, (1, 8), False, "NHD"), (2, (4, 32), False, "HND"), (1, (2, 8), True, "NHD"),
                                                             (4, (1, 4), False, "NHD")])
def test_attn_fp8_first_generation_combine_kernel_form(num_seq_q, heads, k_per_token, shape):
    """The first-generation kernel merges the chunks of a split request inside the launch (round 5: the chunk that
    arrives last folds all of them - one atomic add per chunk on the zero-once counters at the start of the workspace;
    team and one-task-per-wave forms, one / two q blocks).  Development key 33 = 1 keeps the round-1 form (fp32 partials
    + decode_combine_kernel as a second launch), which still serves calls with more (kv head, request) pairs than
    counters.  Both against the oracle on the same inputs, and the in-launch form twice in a row (counters left zero)."""
    lens = torch.tensor([20000, 3, 9000, 130, 64, 63, 65, 4097, 700, 1, 127, 129, 31000, 2, 0, 255, 256, 257], dtype=torch.int32)
    atol = 0.1 if k_per_token else 0.2
    for key in (1, 0, 0):
        dev_set(33, key)
        try:
            _run(len(lens), num_seq_q, lens, 64, heads, k_per_token, True, True, shape, atol)
        finally:
            dev_set(33, 0)


@pytest.mark.dev
@pytest.mark.gpu
@pytest.mark.parametrize("form", ["four_heads"])  # (the head-pair form serves these shapes by default: every other test)
@pytest.mark.parametrize("num_seq_q,block_size,heads", [(1, 64, (8, 64)), (1, 64, (4, 32)), (2, 32, (4, 16)), (1, 16, (4, 16)),
                                                         (2, 16, (8, 32)), (1, 32, (12, 48)), (1, 64, (16, 64))])
def test_attn_fp8_four_heads_per_workgroup(form, num_seq_q, block_size, heads):
    """Calls with <= 8 q rows per kv head and a multiple of 4 kv heads can run the four-head form of the
    second-generation kernel (two kv heads share the 16 columns of an MFMA tile; development key 29 = 2 - measured
    slower than head pairs, so not the default).  Both against the oracle: full tiles (8 rows) and half-filled ones (4 rows: group 4, one q token), every
    page size, split requests, short and empty requests, one to three head quads."""
    import hpc

    lens = torch.tensor([20000, 3, 9000, 130, 64, 63, 65, 4097, 700, 1, 0, 127, 129, 15, 16, 17, 31000, 2], dtype=torch.int32)
    dev_set(29, 0 if form == "head_pairs" else 2)
    try:
        for new_kv_included in (True, False):
            _run(len(lens), num_seq_q, lens, block_size, heads, False, new_kv_included, True, "NHD", 0.2)
    finally:
        dev_set(29, 0)


@pytest.mark.dev
@pytest.mark.gpu
@pytest.mark.parametrize("keys", [{32: 115}, {30: 118, 31: 108, 32: 125}])
@pytest.mark.parametrize("num_batch,hi", [(64, 6000)])  # (development variants that lost their A/B: a small grid)
def test_attn_fp8_uneven_ranges(keys, num_batch, hi):
    """Development variants of the in-kernel plan: longer ranges for the first half of the grid (key 32) and unequal
    numbers of ranges per head pair (keys 30 / 31).  Every request still has to be covered exactly once and split
    requests have to find all their chunks (the chunk count of a request comes from the same range arithmetic)."""
    import hpc

    lens = _mixed_lens(num_batch, 7 * num_batch, hi)
    for k, v in keys.items():
        dev_set(k, v)
    try:
        _run(num_batch, 1, lens, 64, (8, 64), False, True, True, "NHD", 0.2)
    finally:
        for k in keys:
            dev_set(k, 0)


@pytest.mark.dev
@pytest.mark.gpu
@pytest.mark.parametrize("heads", [(8, 64), (4, 32), (16, 64)])
def test_attn_fp8_workgroup_to_slice_mapping_is_only_a_mapping(heads):
    """Round 5: the second workgroup of every CU serves head pair p ^ mask (the slice of the token rows across byte-address
    bit 9 from its CU mate's; csrc/attention_decode_v2.hip).  Which workgroup computes which (range, pair) must not change a
    bit of the result: the product's mask against none (development key 36 = 1) and against the other masks of the sweep,
    and an XCD -> slice table (key 38) - same plan, same arithmetic, `torch.equal`."""
    import hpc

    num_head_kv, num_head_q = heads
    num_batch, block_size = 48, 64
    lens = _mixed_lens(num_batch, 11, 9000)
    lens[3] = 30000  # split over many ranges: the merge finds its chunks under every mapping
    q8, q_scale, kv, block_ids, nblocks = _case(num_batch, 1, lens, block_size, heads, False)
    kv_dev = kv.to(torch.float8_e4m3fn).cuda()
    ks, vs = (torch.rand(1) + 0.5).cuda(), (torch.rand(1) + 0.5).cuda()
    lens_in = (lens + 1).cuda()
    tm = hpc.get_attention_decode_task_workspace(num_batch, int(lens.max()) + 1, num_head_kv, 64)
    hpc.assign_attention_decode_task(lens_in, tm, num_head_kv, 1, True, 64)
    qd, qs, bd = q8.cuda(), q_scale.cuda(), block_ids.cuda()

    def call():
        y = hpc.attention_decode_fp8(qd, kv_dev[:, 0, :block_size], kv_dev[:, 1, :block_size], bd, lens_in, qs, ks, vs, mtp=0,
                                     new_kv_included=True, quant_type=hpc.QuantType.QPERTOKEN_PERHEAD_KPERTENSOR_VPERTENSOR,
                                     splitk=True, task_map=tm)
        torch.cuda.synchronize()
        return y

    ref = call()
    variants = [{36: 1}, {36: 2}, {36: 3}] + ([{38: 16205392}, {36: 4}] if num_head_kv == 8 else [])
    for keys in variants:
        for k, v in keys.items():
            dev_set(k, v)
        try:
            assert torch.equal(call(), ref), keys
        finally:
            for k in keys:
                dev_set(k, 0)


@pytest.mark.gpu
def test_attn_fp8_two_graphs_captured_before_any_replay():
    """The usual serving pattern: one hipGraph per batch size, all captured first, replayed later in any order.  The
    scratch of a decode call starts with arrival counters that must be zero on first use; a buffer first used inside a
    capture has its zero-fill only RECORDED in that graph, so every capture must own its buffer (and its zero node):
    the second graph, replayed before the first one ever ran, must not find un-zeroed counters."""
    import hpc
    from oracle import attention as oattn

    hpc.release_decode_workspaces()
    cases = []
    for num_batch, hi, seed in ((6, 30000, 1), (9, 20000, 2)):
        torch.manual_seed(seed)
        lens = _mixed_lens(num_batch, seed, hi)
        lens[0] = hi  # a long request among short ones: split over many workgroups, merged by the last arriver
        q8, q_scale, kv, block_ids, nblocks = _case(num_batch, 1, lens, 64, (4, 32), False)
        kv8 = kv.to(torch.float8_e4m3fn)
        ks, vs = torch.rand(1) + 0.5, torch.rand(1) + 0.5
        gt = oattn.ref_attn_fp8(q8, kv8, block_ids, nblocks, 1, lens, q_scale, ks, vs, False)
        kvd = kv8.cuda()
        dev = dict(q=q8.cuda(), k=kvd[:, 0], v=kvd[:, 1], bid=block_ids.cuda(), lens=(lens + 1).cuda(), qs=q_scale.cuda(),
                   ks=ks.cuda(), vs=vs.cuda(), out=torch.zeros(num_batch, 32, 128, dtype=torch.bfloat16, device="cuda"))
        cases.append((dev, gt))
    graphs = []
    for dev, _ in cases:  # capture only: nothing runs
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            hpc.attention_decode_fp8(dev["q"], dev["k"], dev["v"], dev["bid"], dev["lens"], dev["qs"], dev["ks"], dev["vs"],
                                     mtp=0, new_kv_included=True,
                                     quant_type=hpc.QuantType.QPERTOKEN_PERHEAD_KPERTENSOR_VPERTENSOR, splitk=True,
                                     output=dev["out"])
        graphs.append(g)
    for i in (1, 0, 1, 0):  # the graph captured second runs first
        cases[i][0]["out"].zero_()
        graphs[i].replay()
        torch.cuda.synchronize()
        assert allclose(cases[i][1], cases[i][0]["out"].cpu(), atol=0.2), i
    hpc.release_decode_workspaces()


@pytest.mark.dev
@pytest.mark.gpu
@pytest.mark.parametrize("heads,shape", [((8, 64), "NHD"), ((1, 8), "NHD"), ((4, 32), "HND")])
def test_attn_fp8_stale_arrival_counter_is_reported(heads, shape):
    """both kernel generations take their split-request tickets on the zero-once counters at the start of
    the workspace; a counter that is NOT zero on entry (a caller-owned buffer that was never cleared) means the last arriver
    is never recognised and rows of y silently stay unwritten.  The development build counts arrivals whose ticket exceeds
    the request's chunk count (hpc_dev_decode_ticket_overruns).  Through the C-ABI with a caller-owned workspace: a clean
    buffer -> no overruns, the op's own result, counters left zero; the same call on dirtied counters -> reported."""
    import ctypes

    import hpc
    from hpc import _C

    if not _C.DEV_BUILD:
        pytest.skip("the overrun counter exists in the development build only (tests/test_dev_build.py runs this case)")
    lib = _C.lib
    num_head_kv, num_head_q = heads
    lens = torch.tensor([20000, 3, 9000, 130, 31000, 64], dtype=torch.int32)
    B, bs = len(lens), 64
    q8, q_scale, kv, block_ids, nblocks = _case(B, 1, lens, bs, heads, False)
    kv_dev = kv.to(torch.float8_e4m3fn).cuda()
    if shape == "HND":
        kv_dev = kv_dev.view(torch.uint8).permute(0, 1, 3, 2, 4).contiguous().permute(0, 1, 3, 2, 4).view(torch.float8_e4m3fn)
    kc, vc = kv_dev[:, 0], kv_dev[:, 1]
    qd, bd, lens_in, qs = q8.cuda(), block_ids.cuda(), (lens + 1).cuda(), q_scale.cuda()
    ks, vs = torch.rand(1).cuda() + 0.5, torch.rand(1).cuda() + 0.5
    task_map = hpc.get_attention_decode_task_workspace(B, int(lens.max()) + 1, num_head_kv, min_process_len=1024)
    hpc.assign_attention_decode_task(lens_in, task_map, num_head_kv, 1, True, min_process_len=1024)
    want = hpc.attention_decode_fp8(qd, kc, vc, bd, lens_in, qs, ks, vs, mtp=0, new_kv_included=True,
                                    quant_type=hpc.QuantType.QPERTOKEN_PERHEAD_KPERTENSOR_VPERTENSOR, splitk=True,
                                    task_map=task_map)
    torch.cuda.synchronize()
    nbins = lib.hpc_attention_decode_num_bins(1, torch.cuda.current_device())
    zero_bytes = lib.hpc_attention_decode_workspace_zero_bytes()
    ws = torch.empty(lib.hpc_attention_decode_workspace_bytes(nbins, B, num_head_kv, 1, num_head_q // num_head_kv),
                     dtype=torch.uint8, device="cuda")

    def ip(t):
        return ctypes.cast(t.data_ptr(), _C.IP)

    def call(y):
        st = torch.cuda.current_stream().cuda_stream
        rc = lib.hpc_attention_decode_fp8_async(
            y.data_ptr(), ws.data_ptr(), ip(task_map), qd.data_ptr(), kc.data_ptr(), vc.data_ptr(), ip(bd), ip(lens_in),
            qs.data_ptr(), ks.data_ptr(), vs.data_ptr(), 1, 1, nbins, B, 1, num_head_q, num_head_kv, 128, 128, bs,
            bd.shape[1], qs.stride(0), y.stride(0), qd.stride(0), kc.stride(0), kc.stride(1), kc.stride(2), vc.stride(0),
            vc.stride(1), vc.stride(2), 0, 0, 0, st)
        assert rc == 0
        torch.cuda.synchronize()

    assert lib.hpc_dev_decode_ticket_overruns(1) >= 0
    ws[:zero_bytes].zero_()
    y = torch.full_like(want, float("nan"))
    call(y)
    assert lib.hpc_dev_decode_ticket_overruns(1) == 0
    assert torch.equal(y, want)
    assert int(ws[:zero_bytes].view(torch.int32).abs().sum()) == 0  # left zero: the buffer can be reused as it is
    ws[:zero_bytes].view(torch.int32).fill_(1)  # a buffer nobody cleared
    call(torch.full_like(want, float("nan")))
    assert lib.hpc_dev_decode_ticket_overruns(1) > 0
