// C++ host side, part 1: the library definition, version / built_json, the decode-attention ops and the scheduler,
// the blockwise fused MoE, RMSNorm + quant, and the MulticastCommunicator torch class - on top of the C-ABI of
// include/hpc_amd.h (libhpc_amd.so).  Parts 2-4: torch_moe.cpp, torch_prefill.cpp, torch_misc.cpp.  Together they
// register EVERY op of the package from C++, like the reference does (src/*/entry.cc TORCH_LIBRARY_FRAGMENT blocks):
// there are no Python-side op implementations.
//
// Mirrors the reference's registration blocks - src/C/C.cc:5 (TORCH_LIBRARY(hpc, m)), src/C/version.cc, built_json.cu,
// src/attention/entry.cc:822-874 (decode ops + scheduler), src/fuse_moe/entry.cc:644-684 (fuse_moe_blockwise[_fp8]),
// src/normalization/entry.cc:59-65, src/communicator/entry.cc:79-90 (m.class_<MulticastCommunicator>) - with the same
// schemas, check messages (TORCH_CHECK -> RuntimeError) and ownership rules (outputs allocated here unless the caller
// passes one; scratch from the caching allocator).  Built in-tree by hpc-ops_amd/build.py into hpc/_hpc_torch.so and
// loaded by hpc/_C.py with torch.ops.load_library.  No kernels here: host code only.
#include <torch/custom_class.h>

#include <map>
#include <mutex>

#include "torch_common.h"

using namespace hpc_torch;

namespace {

// ---- decode attention ---------------------------------------------------------------------------------------
// Scratch of a decode call: [arrival counters (zero on first use, left zero by every call) | partials] - a buffer of
// the per-(device, stream, hipGraph capture) cache of torch_common.h (the reference allocates lse / split_out per call
// and zeroes split_flag per call, src/attention/entry.cc:660-663, 690-694).
at::Tensor decode_workspace(const at::Tensor& like, int64_t nbytes) {
  return cached_scratch(kScratchDecode, like, std::max<int64_t>(nbytes, 1 << 20), hpc_attention_decode_workspace_zero_bytes());
}

struct DecodeCommon {
  int num_batch, num_seq_q, group;
};
DecodeCommon decode_common_checks(const at::Tensor& q, const at::Tensor& kcache, const at::Tensor& vcache,
                                  const at::Tensor& block_ids, const at::Tensor& num_seq_kvcache, int64_t mtp,
                                  int64_t max_mtp) {
  TORCH_CHECK(q.is_cuda(), "q tensor must be cuda");
  TORCH_CHECK(kcache.is_cuda(), "kcache tensor must be cuda");
  TORCH_CHECK(vcache.is_cuda(), "vcache tensor must be cuda");
  TORCH_CHECK(block_ids.is_cuda(), "block_ids tensor must be cuda");
  TORCH_CHECK(block_ids.is_contiguous(), "block_ids tensor must be contiguous");
  TORCH_CHECK(num_seq_kvcache.is_contiguous(), "num_seq_kvcache tensor must be contiguous");
  TORCH_CHECK(block_ids.scalar_type() == at::kInt, "block_ids dtype must be int32");
  TORCH_CHECK(num_seq_kvcache.scalar_type() == at::kInt, "num_seq_kvcache dtype must be int32");
  TORCH_CHECK(0 <= mtp && mtp <= max_mtp, "we only support mtp 0..", max_mtp, ".");
  DecodeCommon c;
  c.num_batch = static_cast<int>(num_seq_kvcache.size(0));
  c.num_seq_q = static_cast<int>(q.size(0) / std::max(c.num_batch, 1));
  TORCH_CHECK(c.num_seq_q == mtp + 1, "every request num_seq_q must be mtp + 1");
  TORCH_CHECK(q.size(2) == 128, "we only support head dim 128.");
  TORCH_CHECK(q.stride(2) == 1 && q.stride(1) == 128, "q heads must be contiguous");
  TORCH_CHECK(kcache.stride(3) == 1 && vcache.stride(3) == 1, "kv cache dims must be contiguous");
  c.group = static_cast<int>(q.size(1) / kcache.size(2));
  TORCH_CHECK(c.group == 4 || c.group == 8, "we only support num_head_q / num_head_k == 4 or 8.");
  return c;
}

// byte size + scheduler byte size of the task-map workspace (reference hpc/attention.py:540-571)
std::pair<int64_t, int64_t> task_workspace_bytes(int num_cu, int64_t max_num_batch, int64_t max_seqlen, int64_t num_head_kv,
                                                 int64_t min_process_len) {
  const int64_t k_task = 48, k_max_cta = 4, k_tile = 64;
  const int64_t max_cta = num_cu * k_max_cta;
  const int64_t total_tiles = max_num_batch * num_head_kv * ((max_seqlen + k_tile - 1) / k_tile);
  int64_t max_tasks = 0;
  for (int cta_per_cu = 4; cta_per_cu >= 1; --cta_per_cu) {
    const int64_t ctas = static_cast<int64_t>(num_cu) * cta_per_cu;
    const int64_t per = std::max((total_tiles + ctas - 1) / ctas, min_process_len / k_tile);
    max_tasks = std::max(max_tasks, (per + 1) * ctas + 1);
  }
  const int64_t chunk_bytes = (max_num_batch * num_head_kv * 4 + k_task - 1) / k_task * k_task;
  const int64_t cta_pad = (max_cta + 11) / 12 * 12 * 4;
  const int64_t sched = max_tasks * k_task + chunk_bytes;
  return {sched + 2 * cta_pad, sched};
}

int num_bins_of(int num_seq_q, const at::Tensor& on) {
  const int n = hpc_attention_decode_num_bins(num_seq_q, on.is_cuda() ? on.device().index() : -1);
  TORCH_CHECK(n > 0, "we only support num_seq_q 1..5 (and a HIP device must be present)");
  return n;
}

at::Tensor assign_task_cuda(const at::Tensor& num_seq_kvcache, int64_t num_head_kv, int64_t num_seq_q, bool new_kv_included,
                            int64_t min_process_len, const c10::optional<at::Tensor>& task_map) {
  // reference assign_attention_decode_task_cuda_entry (entry.cc:780-817)
  TORCH_CHECK(num_seq_kvcache.is_cuda(), "num_seq_kvcache tensor must be cuda");
  TORCH_CHECK(num_seq_kvcache.scalar_type() == at::kInt, "num_seq_kvcache dtype must be int32");
  TORCH_CHECK(num_seq_kvcache.is_contiguous(), "num_seq_kvcache tensor must be contiguous");
  TORCH_CHECK(task_map.has_value(), "assign_attention_decode_task_cuda must use task_map output.");
  TORCH_CHECK(num_seq_kvcache.size(0) <= 4096, "assign_attention_decode_task_cuda only support batch_size <= 4096");
  const int bins = num_bins_of(static_cast<int>(num_seq_q), num_seq_kvcache);
  const int rc = hpc_assign_attention_decode_task_async(
      static_cast<int*>(task_map->data_ptr()), static_cast<const int*>(num_seq_kvcache.data_ptr()), bins,
      static_cast<int>(num_seq_kvcache.size(0)), static_cast<int>(num_head_kv), static_cast<int>(num_seq_q),
      new_kv_included ? 1 : 0, static_cast<int>(min_process_len), stream_of(num_seq_kvcache));
  HPC_LAUNCH_CHECK(rc, "assign_attention_decode_task_async");
  return *task_map;
}

at::Tensor assign_task_cpu(const at::Tensor& num_seq_kvcache, int64_t num_head_kv, int64_t num_seq_q, bool new_kv_included,
                           int64_t min_process_len, const c10::optional<at::Tensor>& /*placehold*/) {
  // reference assign_attention_decode_task_cpu_entry (entry.cc:727-778)
  TORCH_CHECK(num_seq_kvcache.device().is_cpu(), "num_seq_kvcache tensor must be cpu");
  TORCH_CHECK(num_seq_kvcache.scalar_type() == at::kInt, "num_seq_kvcache dtype must be int32");
  const at::Tensor lens = num_seq_kvcache.contiguous();
  const int max_bins = hpc_attention_decode_num_bins(static_cast<int>(num_seq_q), -1);
  TORCH_CHECK(max_bins > 0, "we only support num_seq_q 1..5 (and a HIP device must be present)");
  const int* lp = static_cast<const int*>(lens.data_ptr());
  const int nb = static_cast<int>(lens.numel());
  // the same bin count the device scheduler picks (a small batch is planned on fewer bins): byte-identical maps
  const int bins = hpc_attention_decode_effective_bins(lp, nb, static_cast<int>(num_head_kv), static_cast<int>(num_seq_q),
                                                       new_kv_included ? 1 : 0, max_bins);
  TORCH_CHECK(bins > 0, "assign_attention_decode_task: invalid arguments");
  const int rows = hpc_assign_attention_decode_task_rows(lp, bins, nb, static_cast<int>(num_head_kv), static_cast<int>(num_seq_q),
                                                         new_kv_included ? 1 : 0, static_cast<int>(min_process_len));
  HPC_LAUNCH_CHECK(rows > 0 ? 0 : (rows < 0 ? rows : -2), "assign_attention_decode_task_sync");
  at::Tensor task_map = at::zeros({rows, 48}, at::TensorOptions().dtype(at::kChar));
  const int rc = hpc_assign_attention_decode_task_sync(lp, bins, nb, static_cast<int>(num_head_kv), static_cast<int>(num_seq_q),
                                                       new_kv_included ? 1 : 0, static_cast<int>(min_process_len),
                                                       static_cast<int*>(task_map.data_ptr()), rows);
  HPC_LAUNCH_CHECK(rc == rows ? 0 : (rc < 0 ? rc : -2), "assign_attention_decode_task_sync");
  return task_map;
}

// No task_map given (reference static split-K path, entry.cc:507-563): build the dynamic schedule on the fly
at::Tensor schedule_on_the_fly(const at::Tensor& num_seq_kvcache, const at::Tensor& block_ids, int64_t block_size,
                               int64_t num_head_kv, int64_t num_seq_q, bool new_kv_included) {
  const int num_cu = hpc_get_cu_count(num_seq_kvcache.device().index());
  TORCH_CHECK(num_cu > 0, "hpc_get_cu_count failed (no HIP device?)");
  const int64_t max_seq = block_ids.size(1) * block_size + num_seq_q;
  const auto sz = task_workspace_bytes(num_cu, num_seq_kvcache.size(0), max_seq, num_head_kv, 512);
  at::Tensor ws = at::zeros({sz.first}, num_seq_kvcache.options().dtype(at::kChar));
  at::Tensor hdr = ws.view(at::kInt);  // header ints 2..4 (device-side fills: capturable in a hipGraph)
  hdr.narrow(0, 2, 1).fill_(num_head_kv);
  hdr.narrow(0, 3, 1).fill_(num_seq_kvcache.size(0));
  hdr.narrow(0, 4, 1).fill_(sz.second);
  return assign_task_cuda(num_seq_kvcache, num_head_kv, num_seq_q, new_kv_included, 512, ws);
}

at::Tensor attention_decode_bf16(const at::Tensor& q, at::Tensor& kcache, at::Tensor& vcache, const at::Tensor& block_ids,
                                 const at::Tensor& num_seq_kvcache, int64_t mtp, bool new_kv_included, bool use_splitk,
                                 const c10::optional<at::Tensor>& task_map_in, const c10::optional<at::Tensor>& /*split_flag*/,
                                 const c10::optional<at::Tensor>& output) {
  const DecodeCommon c = decode_common_checks(q, kcache, vcache, block_ids, num_seq_kvcache, mtp, 4);
  TORCH_CHECK(q.scalar_type() == at::kBFloat16, "q dtype must be bfloat16");
  TORCH_CHECK(kcache.scalar_type() == at::kBFloat16 && vcache.scalar_type() == at::kBFloat16, "kv cache dtype must be bfloat16");
  const int64_t block_size = kcache.size(1);
  TORCH_CHECK(block_size == 16 || block_size == 32 || block_size == 64, "kvcache paged blocksize must be 16, 32 or 64.");
  const int64_t num_head_q = q.size(1), num_head_kv = kcache.size(2);
  at::Tensor task_map;
  if (task_map_in.has_value()) {
    task_map = *task_map_in;
    TORCH_CHECK(task_map.is_cuda(), "task_map tensor must be cuda");
    TORCH_CHECK(task_map.is_contiguous(), "task_map tensor must be contiguous");
    TORCH_CHECK(task_map.scalar_type() == at::kChar || task_map.scalar_type() == at::kInt,
                "task_map dtype must be int8 (raw workspace) or int32 (typed)");
    TORCH_CHECK(use_splitk, "attention_decode_bf16: splitk must be true with a task_map.");
  } else {
    task_map = schedule_on_the_fly(num_seq_kvcache, block_ids, block_size, num_head_kv, c.num_seq_q, new_kv_included);
  }
  at::Tensor y = output.has_value() ? *output
                                    : at::empty({static_cast<int64_t>(c.num_batch) * c.num_seq_q, num_head_q, vcache.size(3)},
                                                q.options().dtype(at::kBFloat16));
  const int bins = num_bins_of(c.num_seq_q, q);
  const int64_t ws_bytes = hpc_attention_decode_workspace_bytes(bins, c.num_batch, static_cast<int>(num_head_kv), c.num_seq_q, c.group);
  at::Tensor ws = decode_workspace(q, ws_bytes);
  const int rc = hpc_attention_decode_bf16_async(
      ptr(y), ptr(ws), static_cast<const int*>(task_map.data_ptr()), ptr(q), ptr(kcache), ptr(vcache),
      static_cast<const int*>(block_ids.data_ptr()),
      num_seq_kvcache.is_cuda() ? static_cast<const int*>(num_seq_kvcache.data_ptr()) : nullptr, new_kv_included ? 1 : 0, bins,
      c.num_batch, c.num_seq_q, static_cast<int>(num_head_q),
      static_cast<int>(num_head_kv), static_cast<int>(q.size(2)), static_cast<int>(vcache.size(3)), static_cast<int>(block_size),
      static_cast<int>(block_ids.size(1)), static_cast<int>(y.stride(0)), static_cast<int>(q.stride(0)), kcache.stride(0),
      kcache.stride(1), kcache.stride(2), vcache.stride(0), vcache.stride(1), vcache.stride(2), stream_of(q));
  HPC_LAUNCH_CHECK(rc, "attn decode kernel");
  return y;
}

at::Tensor attention_decode_fp8(const at::Tensor& q, at::Tensor& kcache, at::Tensor& vcache, const at::Tensor& block_ids,
                                const at::Tensor& num_seq_kvcache, const at::Tensor& qscale, const at::Tensor& kscale,
                                const at::Tensor& vscale, int64_t mtp, bool new_kv_included, int64_t quant_type, bool /*use_splitk*/,
                                const c10::optional<at::Tensor>& task_map_in, const c10::optional<at::Tensor>& /*split_flag*/,
                                const c10::optional<at::Tensor>& output) {
  // reference attention_decode_fp8_entry, src/attention/entry.cc:569-725
  const DecodeCommon c = decode_common_checks(q, kcache, vcache, block_ids, num_seq_kvcache, mtp, 3);
  TORCH_CHECK(q.scalar_type() == at::kFloat8_e4m3fn, "q dtype must be fp8_e4m3fn");
  TORCH_CHECK(kcache.element_size() == 1, "kcache tensor element type size must be fp8_e4m3");
  TORCH_CHECK(vcache.element_size() == 1, "vcache tensor element type size must be fp8_e4m3");
  TORCH_CHECK(qscale.scalar_type() == at::kFloat && vscale.scalar_type() == at::kFloat, "qscale / vscale must be float32");
  TORCH_CHECK(quant_type == 0 || quant_type == 1,
              "quant_type must be QPERTOKEN_PERHEAD_KPERTOKEN_PERHEAD_VPERHEAD or QPERTOKEN_PERHEAD_KPERTENSOR_VPERTENSOR");
  TORCH_CHECK(num_seq_kvcache.is_cuda(), "num_seq_kvcache tensor must be cuda");
  const int64_t block_size = kcache.size(1);
  int64_t ks[3] = {0, 0, 0};
  if (quant_type == 0) {
    TORCH_CHECK(block_size == 32 || block_size == 64, "kvcache paged blocksize must be 32 or 64.");
    TORCH_CHECK(kscale.dim() == 4 && kscale.stride(3) == 1 && kscale.element_size() == 1,
                "per-token kscale must be the byte view of the K-cache tail rows");
    ks[0] = kscale.stride(0), ks[1] = kscale.stride(1), ks[2] = kscale.stride(2);
  } else {
    TORCH_CHECK(block_size == 16 || block_size == 32 || block_size == 64, "kvcache paged blocksize must be 16, 32 or 64.");
    TORCH_CHECK(kscale.scalar_type() == at::kFloat && kscale.numel() >= 1, "kscale must be float32 [1]");
  }
  const int64_t num_head_q = q.size(1), num_head_kv = kcache.size(2);
  at::Tensor task_map;
  if (task_map_in.has_value()) {
    task_map = *task_map_in;
    TORCH_CHECK(task_map.is_cuda() && task_map.is_contiguous(), "task_map tensor must be cuda, contiguous");
  } else {
    task_map = schedule_on_the_fly(num_seq_kvcache, block_ids, block_size, num_head_kv, c.num_seq_q, new_kv_included);
  }
  at::Tensor y = output.has_value() ? *output
                                    : at::empty({static_cast<int64_t>(c.num_batch) * c.num_seq_q, num_head_q, vcache.size(3)},
                                                q.options().dtype(at::kBFloat16));
  const int bins = num_bins_of(c.num_seq_q, q);
  const int64_t ws_bytes = hpc_attention_decode_workspace_bytes(bins, c.num_batch, static_cast<int>(num_head_kv), c.num_seq_q, c.group);
  at::Tensor ws = decode_workspace(q, ws_bytes);
  const int rc = hpc_attention_decode_fp8_async(
      ptr(y), ptr(ws), static_cast<const int*>(task_map.data_ptr()), ptr(q), ptr(kcache), ptr(vcache),
      static_cast<const int*>(block_ids.data_ptr()), static_cast<const int*>(num_seq_kvcache.data_ptr()),
      static_cast<const float*>(qscale.data_ptr()), ptr(kscale), static_cast<const float*>(vscale.data_ptr()), new_kv_included ? 1 : 0,
      static_cast<int>(quant_type), bins, c.num_batch, c.num_seq_q, static_cast<int>(num_head_q), static_cast<int>(num_head_kv),
      static_cast<int>(q.size(2)), static_cast<int>(vcache.size(3)), static_cast<int>(block_size), static_cast<int>(block_ids.size(1)),
      static_cast<int>(qscale.stride(0)), static_cast<int>(y.stride(0)), static_cast<int>(q.stride(0)), kcache.stride(0), kcache.stride(1),
      kcache.stride(2), vcache.stride(0), vcache.stride(1), vcache.stride(2), ks[0], ks[1], ks[2], stream_of(q));
  HPC_LAUNCH_CHECK(rc, "attn decode kernel");
  return y;
}

// ---- fused MoE, blockwise FP8 (reference fuse_moe_blockwise_entry, src/fuse_moe/entry.cc:445-560) -------------------
at::Tensor fuse_moe_blockwise(const at::Tensor& x, const at::Tensor& x_scale, const at::Tensor& gate_up_weight,
                              const at::Tensor& gate_up_weight_scale, const at::Tensor& down_weight,
                              const at::Tensor& down_weight_scale, const at::Tensor& topk_ids, const at::Tensor& topk_scale,
                              const c10::optional<at::Tensor>& shared_output, int64_t rank_ep, int64_t num_expert_total,
                              const c10::optional<at::Tensor>& output) {
  const auto f8 = at::kFloat8_e4m3fn;
  TORCH_CHECK(x.scalar_type() == f8 && gate_up_weight.scalar_type() == f8 && down_weight.scalar_type() == f8,
              "x, gate_up_weight and down_weight dtype must be fp8_e4m3");
  TORCH_CHECK(topk_ids.scalar_type() == at::kInt, "topk_ids dtype must be int32");
  TORCH_CHECK(gate_up_weight_scale.scalar_type() == at::kFloat && down_weight_scale.scalar_type() == at::kFloat &&
                  topk_scale.scalar_type() == at::kFloat && x_scale.scalar_type() == at::kFloat,
              "gate_up_scale, down_scale, x_scale and topk_scale dtype must be float32");
  cuda_contig(x, "x");
  cuda_contig(x_scale, "x_scale");
  cuda_contig(gate_up_weight, "gate_up_weight");
  cuda_contig(gate_up_weight_scale, "gate_up_weight_scale");
  cuda_contig(down_weight, "down_weight");
  cuda_contig(down_weight_scale, "down_weight_scale");
  cuda_contig(topk_ids, "topk_ids");
  cuda_contig(topk_scale, "topk_scale");
  TORCH_CHECK(x.size(0) == topk_ids.size(0), "x and topk_ids must share the same num_tokens");
  TORCH_CHECK(topk_ids.sizes() == topk_scale.sizes(), "topk_ids and topk_scale must share the same shape");
  TORCH_CHECK(x.size(1) == gate_up_weight.size(2), "x and weight must share the same k");
  TORCH_CHECK(gate_up_weight.size(0) == down_weight.size(0), "gate_up_weight and down_weight must share the same num_expert");
  TORCH_CHECK(x_scale.size(0) == x.size(0) && x_scale.size(1) == x.size(1) / 128, "x_scale must be per 128 blockwise quant");
  TORCH_CHECK(gate_up_weight_scale.size(1) == gate_up_weight.size(1) / 128 &&
                  gate_up_weight_scale.size(2) == (gate_up_weight.size(2) / 128 + 3) / 4 * 4,
              "gate_up_weight must be per 128 blockwise quant and must be aligned to 4");
  TORCH_CHECK(down_weight_scale.size(1) == down_weight.size(1) / 128 &&
                  down_weight_scale.size(2) == (down_weight.size(2) / 128 + 3) / 4 * 4,
              "down_weight must be per 128 blockwise quant and must be aligned to 4");
  TORCH_CHECK(down_weight.size(1) == x.size(1) && down_weight.size(2) * 2 == gate_up_weight.size(1),
              "down_weight must be [num_expert, hidden, intermediate]");
  const int64_t num_tokens = x.size(0), hidden = x.size(1);
  const int64_t num_experts = gate_up_weight.size(0), inter2 = gate_up_weight.size(1), num_topk = topk_ids.size(1);
  TORCH_CHECK(num_topk <= 128, "num_topk must less than or equal to 128");
  if (shared_output.has_value()) {
    cuda_contig(*shared_output, "shared_output");
    TORCH_CHECK(shared_output->scalar_type() == at::kBFloat16, "shared_output tensor dtype must be bfloat16");
    TORCH_CHECK(shared_output->dim() == 2 && shared_output->size(0) == num_tokens && shared_output->size(1) == hidden,
                "shared_output tensor shape must be same as x tensor");
  }
  at::Tensor y;
  if (output.has_value()) {
    TORCH_CHECK(output->dim() == 2 && output->size(0) == num_tokens && output->size(1) == hidden,
                "output shape must be [num_tokens, hidden_size]");
    TORCH_CHECK(output->scalar_type() == at::kBFloat16 && output->is_cuda(), "output must be a cuda bfloat16 tensor");
    y = *output;
  } else {
    y = at::empty({num_tokens, hidden}, x.options().dtype(at::kBFloat16));
  }
  const int64_t nbytes = hpc_fuse_moe_blockwise_workspace_bytes(static_cast<int>(num_tokens), static_cast<int>(num_topk),
                                                                static_cast<int>(hidden), static_cast<int>(inter2),
                                                                static_cast<int>(num_experts));
  at::Tensor ws = at::empty({std::max<int64_t>(nbytes, 256)}, x.options().dtype(at::kByte));
  const int rc = hpc_fuse_moe_blockwise_async(
      ptr(y), ptr(ws), ptr(x), ptr(x_scale), ptr(gate_up_weight), ptr(gate_up_weight_scale), ptr(down_weight), ptr(down_weight_scale),
      ptr(topk_ids), ptr(topk_scale), ptr(shared_output), static_cast<int>(num_tokens), static_cast<int>(hidden),
      static_cast<int>(inter2), static_cast<int>(num_topk), static_cast<int>(num_expert_total), static_cast<int>(num_experts),
      static_cast<int>(gate_up_weight_scale.size(2)), static_cast<int>(down_weight_scale.size(2)), static_cast<int>(rank_ep),
      stream_of(x));
  HPC_LAUNCH_CHECK(rc, "fuse_moe_blockwise_async");
  return y;
}

// ---- RMSNorm + fp8 quant (reference fused_rmsnorm_with_scale_entry, src/normalization/entry.cc:20-57) ---------------
std::tuple<at::Tensor, at::Tensor, at::Tensor> fused_rmsnorm_with_scale(const at::Tensor& input, const at::Tensor& weight,
                                                                        const at::Tensor& scale, double eps, bool is_moe) {
  TORCH_CHECK(input.scalar_type() == at::kBFloat16 && weight.scalar_type() == at::kBFloat16, "input and weight must be bfloat16.");
  TORCH_CHECK(input.is_contiguous() && weight.is_contiguous(), "input/weight must be contiguous");
  TORCH_CHECK(scale.scalar_type() == at::kFloat, "scale must be float32");
  TORCH_CHECK(scale.numel() >= (is_moe ? 2 : 1), "scale has too few elements");
  at::Tensor out = at::empty_like(input, input.options().dtype(at::kFloat8_e4m3fn));
  // the reference allocates all three outputs unconditionally (entry.cc:29-31)
  at::Tensor out_fp32 = at::empty_like(input, input.options().dtype(at::kFloat));
  at::Tensor out_scale2 = at::empty_like(input, input.options().dtype(at::kFloat8_e4m3fn));
  const int64_t hidden = input.size(-1);
  const int64_t batch = hidden ? input.numel() / hidden : 0;
  const int rc = hpc_fused_rmsnorm_with_scale_async(ptr(input), ptr(weight), ptr(out), is_moe ? ptr(out_fp32) : nullptr,
                                                    is_moe ? ptr(out_scale2) : nullptr, ptr(scale), static_cast<float>(eps),
                                                    static_cast<int>(batch), static_cast<int>(hidden), is_moe ? 1 : 0, stream_of(input));
  HPC_LAUNCH_CHECK(rc, "fused_rmsnorm_with_scale_async");
  return std::make_tuple(out, out_fp32, out_scale2);
}

// ---- MulticastCommunicator (reference src/communicator/entry.cc:18-74) -----------------------------------------------
// Same constructor and methods as the reference torch class.  There is no multicast object on xGMI: the -1 entry
// of CreateTensorSync aliases the local buffer.  Peer buffers live in other processes' GPUs; torch can alias them
// as tensors only when the peer device is visible to this process (one process per GPU: it is not), so peers are
// returned as 0-byte placeholder tensors and their addresses through PeerAddresses().
struct MulticastCommunicator : torch::CustomClassHolder {
  int handle = 0, rank = 0, world = 0, device = -1;
  std::vector<int64_t> last_peer_ptrs;

  MulticastCommunicator(int64_t rank_, int64_t world_size, int64_t device_id, std::string comm_name) {
    handle = hpc_comm_create(static_cast<int>(rank_), static_cast<int>(world_size), static_cast<int>(device_id), comm_name.c_str());
    TORCH_CHECK(handle > 0, "MulticastCommunicator: rendezvous '", comm_name, "' failed (", handle, ")");
    rank = static_cast<int>(rank_), world = static_cast<int>(world_size), device = static_cast<int>(device_id);
  }
  ~MulticastCommunicator() override {
    if (handle > 0) hpc_comm_destroy(handle);
  }
  int64_t GetRank() { return rank; }
  int64_t GetWorldSize() { return world; }
  int64_t GetDeviceId() { return device; }
  void Barrier() { TORCH_CHECK(hpc_comm_barrier(handle) == 0, "MulticastCommunicator.Barrier failed"); }
  c10::Dict<int64_t, at::Tensor> CreateTensorSync(int64_t size) {
    std::vector<void*> ptrs(static_cast<size_t>(world), nullptr);
    const int rc = hpc_comm_create_tensor_sync(handle, size, ptrs.data());
    TORCH_CHECK(rc == 0, "CreateTensorSync(", size, ") failed (", rc, ")");
    c10::Dict<int64_t, at::Tensor> out;
    last_peer_ptrs.assign(static_cast<size_t>(world), 0);
    const auto opts = at::TensorOptions().dtype(at::kByte).device(at::kCUDA, static_cast<c10::DeviceIndex>(device));
    for (int r = 0; r < world; ++r) {
      last_peer_ptrs[static_cast<size_t>(r)] = reinterpret_cast<int64_t>(ptrs[static_cast<size_t>(r)]);
      if (r == rank)
        out.insert(r, at::from_blob(ptrs[static_cast<size_t>(r)], {size}, [](void*) {}, opts));  // owned by the communicator
      else
        out.insert(r, at::empty({0}, opts));
    }
    out.insert(-1, out.at(rank));
    return out;
  }
  std::vector<int64_t> PeerAddresses() { return last_peer_ptrs; }
};

}  // namespace

TORCH_LIBRARY(hpc, m) {
  // reference src/C/version.cc:14, src/C/built_json.cu:45
  m.def("version", []() { return std::string(hpc_version()); });
  m.def("built_json", []() { return std::string(hpc_built_json()); });
  m.def(
      "assign_attention_decode_task(Tensor num_seq_kvcache, int num_head_kv, int num_seq_q, bool new_kv_included, "
      "int min_process_len, Tensor? task_map) -> (Tensor)");
  m.def(
      "attention_decode_bf16(Tensor q, Tensor! kcache, Tensor! vcache, Tensor block_ids, Tensor num_seq_kvcache, "
      "int mtp, bool new_kv_included, bool use_splitk, Tensor? task_map, Tensor? split_flag, Tensor? output) -> (Tensor)");
  m.def(
      "attention_decode_fp8(Tensor q, Tensor! kcache, Tensor! vcache, Tensor block_ids, Tensor num_seq_kvcache, "
      "Tensor qscale, Tensor kscale, Tensor vscale, int mtp, bool new_kv_included, int quant_type, bool use_splitk, "
      "Tensor? task_map, Tensor? split_flag, Tensor? output) -> (Tensor)");
  m.def(
      "fuse_moe_blockwise_fp8(Tensor x, Tensor x_scale, Tensor gate_up_weight, Tensor gate_up_weight_scale, Tensor "
      "down_weight, Tensor down_weight_scale, Tensor topk_ids, Tensor topk_scale, Tensor ? shared_output, int rank_ep, "
      "int num_expert_total, Tensor ? output) -> (Tensor)");
  m.def(
      "fuse_moe_blockwise(Tensor x, Tensor x_scale, Tensor gate_up_weight, Tensor gate_up_weight_scale, Tensor "
      "down_weight, Tensor down_weight_scale, Tensor topk_ids, Tensor topk_scale, Tensor ? shared_output, int rank_ep, "
      "int num_expert_total, Tensor ? output) -> (Tensor)");
  m.def("fused_rmsnorm_with_scale(Tensor input, Tensor weight, Tensor scale, float eps, bool is_moe) -> (Tensor, Tensor, Tensor)");
  m.def("_release_decode_workspaces() -> ()", []() { release_cached_scratch(); });
  m.def("_decode_workspaces() -> Tensor[]", []() { return list_cached_scratch(kScratchDecode); });
  m.class_<MulticastCommunicator>("MulticastCommunicator")
      .def(torch::init<int64_t, int64_t, int64_t, std::string>(), "",
           {torch::arg("rank"), torch::arg("world_size"), torch::arg("device_id") = -1, torch::arg("comm_name") = "hpc_comm"})
      .def("GetRank", &MulticastCommunicator::GetRank)
      .def("GetWorldSize", &MulticastCommunicator::GetWorldSize)
      .def("GetDeviceId", &MulticastCommunicator::GetDeviceId)
      .def("Barrier", &MulticastCommunicator::Barrier)
      .def("CreateTensorSync", &MulticastCommunicator::CreateTensorSync)
      .def("PeerAddresses", &MulticastCommunicator::PeerAddresses);
}

TORCH_LIBRARY_IMPL(hpc, CUDA, m) {
  m.impl("assign_attention_decode_task", &assign_task_cuda);
  m.impl("attention_decode_bf16", &attention_decode_bf16);
  m.impl("attention_decode_fp8", &attention_decode_fp8);
  m.impl("fuse_moe_blockwise_fp8", &fuse_moe_blockwise);
  m.impl("fuse_moe_blockwise", &fuse_moe_blockwise);
  m.impl("fused_rmsnorm_with_scale", &fused_rmsnorm_with_scale);
}

TORCH_LIBRARY_IMPL(hpc, CPU, m) { m.impl("assign_attention_decode_task", &assign_task_cpu); }
