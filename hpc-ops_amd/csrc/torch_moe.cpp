// C++ host side of the MoE building blocks: reduce, count_and_gather, the grouped GEMMs (blockwise / per-tensor /
// the raw cp.async ops), the per-tensor fused MoE, the activation + quantisation ops.
//
// Mirrors the reference's entries - src/fuse_moe/entry.cc:18-443, 563-684 (fuse_moe_entry, count_and_gather_entry,
// reduce_entry), src/group_gemm/entry.cc:14-250 (group_gemm_*_entry, reformat_x_scale_entry),
// src/group_gemm/cp_async/entry.cc:14-160, src/activation/entry.cc:14-224 - same schemas (verbatim strings,
// tests/test_schemas.py), check messages and ownership rules; compute behind the C-ABI (csrc/fuse_moe.hip,
// csrc/group_gemm_*.hip).  No kernels here.
#include "torch_common.h"

using namespace hpc_torch;

namespace {

const auto kF8 = at::kFloat8_e4m3fn;

// tileM ladder of the reference (src/fuse_moe/entry.cc:525-543): defines the tile-padded column layout of
// transposed x_scale tensors
int aligned_size(int64_t avg) {
  static const int lim[8] = {8, 16, 32, 48, 64, 96, 128, 144}, val[8] = {8, 16, 32, 48, 64, 48, 32, 48};
  for (int i = 0; i < 8; ++i)
    if (avg <= lim[i]) return val[i];
  return 64;
}

// Scan of ceil(seqlens / 128) for the tiled (large-group) GEMM kernels; an undefined tensor keeps the streaming one.
// The caller holds the tensor until its GEMM launch has been enqueued.
at::Tensor cu_tiles128(const at::Tensor& seqlens, int64_t m, int64_t num_group, hpc_stream_t stream) {
  if (m / std::max<int64_t>(num_group, 1) <= 20) return at::Tensor();
  at::Tensor tiles = at::empty({num_group}, seqlens.options().dtype(at::kInt));
  at::Tensor cu = at::empty({num_group + 1}, seqlens.options().dtype(at::kInt));
  HPC_LAUNCH_CHECK(hpc_moe_tiles_async(ptr(seqlens), i32(num_group), 128, ptr(tiles), ptr(cu), stream), "group_gemm tiles");
  return cu;
}
const void* ptr_or_null(const at::Tensor& t) { return t.defined() ? t.data_ptr() : nullptr; }

// ---- reduce (reference reduce_entry, src/fuse_moe/entry.cc:563-642) ------------------------------------------------
at::Tensor reduce(const at::Tensor& x, const at::Tensor& topk_pos, const at::Tensor& topk_scale,
                  const c10::optional<at::Tensor>& shared_output) {
  cuda_contig(x, "x");
  cuda_contig(topk_pos, "topk_pos");
  cuda_contig(topk_scale, "topk_scale");
  TORCH_CHECK(x.scalar_type() == at::kBFloat16, "x dtype must be bfloat16");
  TORCH_CHECK(topk_pos.scalar_type() == at::kInt && topk_scale.scalar_type() == at::kFloat,
              "topk_pos must be int32 and topk_scale float32");
  TORCH_CHECK(topk_pos.sizes() == topk_scale.sizes(), "topk_pos and topk_scale must share the same shape");
  if (shared_output.has_value()) {
    cuda_contig(*shared_output, "shared_output");
    TORCH_CHECK(shared_output->scalar_type() == at::kBFloat16, "shared_output dtype must be bfloat16");
  }
  const int64_t num_tokens = topk_pos.size(0), num_topk = topk_pos.size(1);
  at::Tensor y = at::empty({num_tokens, x.size(1)}, x.options());
  const int rc = hpc_moe_reduce_async(ptr(y), ptr(x), ptr(topk_pos), ptr(topk_scale), ptr(shared_output), i32(num_tokens),
                                      i32(num_topk), i32(x.size(1)), stream_of(x));
  HPC_LAUNCH_CHECK(rc, "reduce_async");
  return y;
}

// ---- group_gemm_blockwise_fp8 (reference src/group_gemm/entry.cc:91-168) --------------------------------------------
at::Tensor group_gemm_blockwise_fp8(const at::Tensor& x, const at::Tensor& weight, const at::Tensor& seqlens,
                                    const at::Tensor& cu_seqlens, const at::Tensor& x_scale, const at::Tensor& w_scale,
                                    int64_t num_seq_per_group_avg, const c10::optional<at::Tensor>& output,
                                    const c10::optional<at::Tensor>& /*tma_desc*/,
                                    const c10::optional<at::Tensor>& /*task_map_workspace*/) {
  TORCH_CHECK(x.is_cuda(), "x tensor must be cuda");
  TORCH_CHECK(weight.is_cuda(), "weight tensor must be cuda");
  TORCH_CHECK(seqlens.is_cuda(), "seqlens tensor must be cuda");
  TORCH_CHECK(cu_seqlens.is_cuda(), "cu_seqlens tensor must be cuda");
  TORCH_CHECK(x.is_contiguous() && weight.is_contiguous(), "x / weight tensor must be contiguous");
  TORCH_CHECK(x.scalar_type() == kF8 && weight.scalar_type() == kF8, "x and weight dtype must be fp8_e4m3");
  TORCH_CHECK(seqlens.scalar_type() == at::kInt && cu_seqlens.scalar_type() == at::kInt,
              "seqlens and cu_seqlens dtype must be int32");
  TORCH_CHECK(x_scale.scalar_type() == at::kFloat && w_scale.scalar_type() == at::kFloat,
              "x_scale and w_scale dtype must be float32");
  TORCH_CHECK(x_scale.is_contiguous() && w_scale.is_contiguous(), "scales must be contiguous");
  TORCH_CHECK(seqlens.size(0) == weight.size(0), "seqlens and weight must share the same num_group");
  TORCH_CHECK(x.size(1) == weight.size(2), "x and weight must share the same k");
  TORCH_CHECK(w_scale.size(2) % 4 == 0, "w_scale must be multiple of 4");
  const int64_t m = x.size(0), k = x.size(1), n = weight.size(1), num_group = seqlens.size(0), m_pad = x_scale.size(1);
  at::Tensor y = output.has_value() ? *output : at::empty({m, n}, x.options().dtype(at::kBFloat16));
  const int tile_m = aligned_size(num_seq_per_group_avg);
  at::Tensor tiles = at::empty({num_group}, seqlens.options());
  at::Tensor cu_tiles = at::empty({num_group + 1}, seqlens.options());
  const hpc_stream_t s = stream_of(x);
  HPC_LAUNCH_CHECK(hpc_moe_tiles_async(ptr(seqlens), i32(num_group), tile_m, ptr(tiles), ptr(cu_tiles), s), "group_gemm tiles");
  const at::Tensor cu128 = cu_tiles128(seqlens, m, num_group, s);
  const int rc = hpc_group_gemm_blockwise_fp8_async(ptr(y), ptr(x), ptr(weight), ptr(seqlens), ptr(cu_seqlens), ptr(x_scale),
                                                    ptr(w_scale), nullptr, ptr(cu_tiles), i32(num_group), i32(m), i32(n), i32(k),
                                                    i32(w_scale.size(2)), tile_m, 1, m_pad, ptr_or_null(cu128), s);
  HPC_LAUNCH_CHECK(rc, "group_gemm_blockwise_fp8_async");
  return y;
}

// ---- per-tensor fused MoE (reference fuse_moe_entry, src/fuse_moe/entry.cc:18-443) ---------------------------------
at::Tensor fuse_moe(const at::Tensor& x, const at::Tensor& gate_up_weight, const at::Tensor& down_weight,
                    const at::Tensor& gate_up_scale, const at::Tensor& down_scale, const at::Tensor& act_and_mul_scale,
                    const at::Tensor& topk_ids, const at::Tensor& topk_scale, const c10::optional<at::Tensor>& shared_output,
                    int64_t rank_ep, int64_t /*num_expert_total*/, bool use_bf16_mul, const c10::optional<at::Tensor>& output) {
  TORCH_CHECK(x.scalar_type() == kF8 && gate_up_weight.scalar_type() == kF8 && down_weight.scalar_type() == kF8,
              "x, gate_up_weight and down_weight dtype must be fp8_e4m3");
  TORCH_CHECK(topk_ids.scalar_type() == at::kInt, "topk_ids dtype must be int32");
  TORCH_CHECK(gate_up_scale.scalar_type() == at::kFloat && down_scale.scalar_type() == at::kFloat &&
                  act_and_mul_scale.scalar_type() == at::kFloat && topk_scale.scalar_type() == at::kFloat,
              "gate_up_scale, down_scale, act_and_mul_scale and topk_scale dtype must be float32");
  cuda_contig(x, "x");
  cuda_contig(gate_up_weight, "gate_up_weight");
  cuda_contig(gate_up_scale, "gate_up_scale");
  cuda_contig(down_weight, "down_weight");
  cuda_contig(down_scale, "down_scale");
  cuda_contig(topk_ids, "topk_ids");
  cuda_contig(topk_scale, "topk_scale");
  cuda_contig(act_and_mul_scale, "act_and_mul_scale");
  TORCH_CHECK(x.size(0) == topk_ids.size(0), "x and topk_ids must share the same num_seq");
  TORCH_CHECK(topk_ids.sizes() == topk_scale.sizes(), "topk_ids and topk_scale must share the same shape");
  TORCH_CHECK(x.size(1) == gate_up_weight.size(2), "x and weight must share the same k");
  TORCH_CHECK(gate_up_weight.size(0) == down_weight.size(0), "gate_up_weight and down_weight must share the same num_expert");
  const int64_t num_tokens = x.size(0), hidden = x.size(1);
  const int64_t num_experts = gate_up_weight.size(0), inter2 = gate_up_weight.size(1), num_topk = topk_ids.size(1);
  TORCH_CHECK(num_topk <= 128, "num_topk must less than or equal to 128");
  TORCH_CHECK(gate_up_scale.numel() >= num_experts && down_scale.numel() >= num_experts, "one scale per local expert is required");
  if (shared_output.has_value()) {
    cuda_contig(*shared_output, "shared_output");
    TORCH_CHECK(shared_output->scalar_type() == at::kBFloat16, "shared_output tensor dtype must be bfloat16");
    TORCH_CHECK(shared_output->dim() == 2 && shared_output->size(0) == num_tokens && shared_output->size(1) == hidden,
                "shared_output tensor shape must be same as x tensor");
  }
  at::Tensor y;
  if (output.has_value()) {
    TORCH_CHECK(output->dim() == 2 && output->size(0) == num_tokens && output->size(1) == hidden &&
                    output->scalar_type() == at::kBFloat16 && output->is_cuda(),
                "output must be a cuda bfloat16 [num_tokens, hidden_size] tensor");
    y = *output;
  } else {
    y = at::empty({num_tokens, hidden}, x.options().dtype(at::kBFloat16));
  }
  const int64_t nbytes = hpc_fuse_moe_blockwise_workspace_bytes(i32(num_tokens), i32(num_topk), i32(hidden),
                                                                i32((inter2 + 255) / 256 * 256), i32(num_experts));
  at::Tensor ws = at::empty({std::max<int64_t>(nbytes, 256)}, x.options().dtype(at::kByte));
  const int rc = hpc_fuse_moe_pertensor_async(ptr(y), ptr(ws), ptr(x), ptr(gate_up_weight), ptr(down_weight), ptr(gate_up_scale),
                                              ptr(down_scale), ptr(act_and_mul_scale), ptr(topk_ids), ptr(topk_scale),
                                              ptr(shared_output), i32(num_tokens), i32(hidden), i32(inter2), i32(num_topk),
                                              i32(num_experts), i32(rank_ep), use_bf16_mul ? 1 : 0, stream_of(x));
  HPC_LAUNCH_CHECK(rc, "fuse_moe_async");
  return y;
}

// ---- count_and_gather (reference src/fuse_moe/entry.cc: count_and_gather_entry) --------------------------------------
std::tuple<at::Tensor, at::Tensor, at::Tensor, at::Tensor, at::Tensor, at::Tensor, at::Tensor, at::Tensor, at::Tensor>
count_and_gather(const at::Tensor& x, const at::Tensor& topk_ids, int64_t num_expert, int64_t rank_ep, int64_t intermediate_size,
                 int64_t num_seq_per_group_avg) {
  cuda_contig(x, "x");
  cuda_contig(topk_ids, "topk_ids");
  TORCH_CHECK(x.size(0) == topk_ids.size(0), "x and topk_ids must share the same k");
  TORCH_CHECK(topk_ids.scalar_type() == at::kInt && x.element_size() == 1, "x must be fp8, topk_ids int32");
  const int64_t num_seq = x.size(0), hidden = x.size(1), num_topk = topk_ids.size(1);
  const auto i32o = x.options().dtype(at::kInt);
  at::Tensor gate_up_input = at::empty({num_seq * num_topk, hidden}, x.options());
  at::Tensor gate_up_output = at::empty({num_seq * num_topk, intermediate_size}, x.options().dtype(at::kBFloat16));
  at::Tensor topk_pos = at::empty({num_seq, num_topk}, i32o);
  at::Tensor seqlens = at::zeros({num_expert}, i32o);
  at::Tensor cu_seqlens = at::empty({num_expert + 1}, i32o);
  at::Tensor tiles = at::empty({num_expert}, i32o);
  at::Tensor cu_tiles = at::empty({num_expert + 1}, i32o);
  at::Tensor row_index = at::empty({num_seq * num_topk}, i32o);
  at::Tensor tmas = at::empty({num_expert * 2, 128}, x.options().dtype(at::kChar));  // unused: no TMA on gfx950
  const int tile_m = aligned_size(num_seq_per_group_avg);
  const hpc_stream_t s = stream_of(x);
  HPC_LAUNCH_CHECK(hpc_moe_count_and_slot_async(ptr(topk_ids), i32(num_seq), i32(num_topk), i32(num_expert), i32(rank_ep), tile_m,
                                                ptr(seqlens), ptr(cu_seqlens), ptr(tiles), ptr(cu_tiles), ptr(topk_pos),
                                                ptr(row_index), s),
                   "count_and_gather_async");
  HPC_LAUNCH_CHECK(hpc_moe_gather_rows_async(ptr(x), ptr(topk_pos), i32(num_seq), i32(num_topk), i32(hidden), ptr(gate_up_input), s),
                   "count_and_gather_async");
  return std::make_tuple(gate_up_input, gate_up_output, topk_pos, seqlens, cu_seqlens, tiles, cu_tiles, tmas, tmas.clone());
}

// ---- group_gemm_fp8 / group_gemm_pertensor_fp8 (reference src/group_gemm/entry.cc:14-89) ----------------------------
at::Tensor group_gemm_fp8(const at::Tensor& x, const at::Tensor& weight, const at::Tensor& seqlens, const at::Tensor& cu_seqlens,
                          const at::Tensor& y_scale, int64_t /*num_seq_per_group_avg*/, const c10::optional<at::Tensor>& output,
                          const c10::optional<at::Tensor>& /*tma_desc*/, const c10::optional<at::Tensor>& /*task_map_workspace*/) {
  TORCH_CHECK(x.is_cuda(), "x tensor must be cuda");
  TORCH_CHECK(weight.is_cuda(), "weight tensor must be cuda");
  TORCH_CHECK(seqlens.is_cuda(), "seqlens tensor must be cuda");
  TORCH_CHECK(cu_seqlens.is_cuda(), "cu_seqlens tensor must be cuda");
  TORCH_CHECK(y_scale.is_cuda(), "y_scale tensor must be cuda");
  TORCH_CHECK(x.is_contiguous() && weight.is_contiguous(), "x / weight tensor must be contiguous");
  TORCH_CHECK(x.scalar_type() == kF8 && weight.scalar_type() == kF8, "x and weight dtype must be fp8_e4m3");
  TORCH_CHECK(seqlens.scalar_type() == at::kInt && cu_seqlens.scalar_type() == at::kInt,
              "seqlens and cu_seqlens dtype must be int32");
  TORCH_CHECK(y_scale.scalar_type() == at::kFloat && y_scale.numel() >= weight.size(0), "y_scale must be float32 [num_group]");
  TORCH_CHECK(seqlens.size(0) == weight.size(0), "seqlens and weight must share the same num_group");
  TORCH_CHECK(x.size(1) == weight.size(2), "x and weight must share the same k");
  const int64_t m = x.size(0), k = x.size(1), n = weight.size(1), num_group = seqlens.size(0);
  at::Tensor y = output.has_value() ? *output : at::empty({m, n}, x.options().dtype(at::kBFloat16));
  const hpc_stream_t s = stream_of(x);
  const at::Tensor cu128 = cu_tiles128(seqlens, m, num_group, s);
  const int rc = hpc_group_gemm_pertensor_fp8_async(ptr(y), ptr(x), ptr(weight), ptr(seqlens), ptr(cu_seqlens), ptr(y_scale), nullptr,
                                                    i32(num_group), i32(m), i32(m), i32(n), i32(k), ptr_or_null(cu128), s);
  HPC_LAUNCH_CHECK(rc, "group_gemm_fp8_async");
  return y;
}

// ---- the reference's raw cp.async group-GEMM ops (src/group_gemm/cp_async/entry.cc) ----------------------------------
at::Tensor group_gemm_cp_async(const at::Tensor& x, const at::Tensor& weight, const at::Tensor& y_scale,
                               const c10::optional<at::Tensor>& row_indices, const at::Tensor& seqlens,
                               const at::Tensor& cu_seqlens) {
  cuda_contig(x, "x");
  cuda_contig(weight, "weight");
  TORCH_CHECK(x.scalar_type() == kF8 && weight.scalar_type() == kF8, "x / weight must be fp8_e4m3");
  TORCH_CHECK(y_scale.scalar_type() == at::kFloat && seqlens.scalar_type() == at::kInt && cu_seqlens.scalar_type() == at::kInt,
              "y_scale must be float32, seqlens / cu_seqlens int32");
  const int64_t num_group = weight.size(0), n = weight.size(1), k = weight.size(2);
  const int64_t m = row_indices.has_value() ? row_indices->size(0) : x.size(0);
  at::Tensor y = at::empty({m, n}, x.options().dtype(at::kBFloat16));
  const hpc_stream_t s = stream_of(x);
  const at::Tensor cu128 = cu_tiles128(seqlens, m, num_group, s);
  const int rc = hpc_group_gemm_pertensor_fp8_async(ptr(y), ptr(x), ptr(weight), ptr(seqlens), ptr(cu_seqlens), ptr(y_scale),
                                                    ptr(row_indices), i32(num_group), i32(m), i32(x.size(0)), i32(n), i32(k),
                                                    ptr_or_null(cu128), s);
  HPC_LAUNCH_CHECK(rc, "group_gemm_fp8_cp_async");
  return y;
}
at::Tensor group_gemm_fp8_cp_async(const at::Tensor& x, const at::Tensor& weight, const at::Tensor& y_scale, const at::Tensor& seqlens,
                                   const at::Tensor& cu_seqlens, const at::Tensor& /*tiles*/, const at::Tensor& /*cu_tiles*/,
                                   bool /*use_task_map*/) {
  // (the caller's 64-row tile tables are not needed here)
  return group_gemm_cp_async(x, weight, y_scale, c10::nullopt, seqlens, cu_seqlens);
}
at::Tensor group_gemm_fp8_scatter_cp_async(const at::Tensor& x, const at::Tensor& weight, const at::Tensor& y_scale,
                                           const at::Tensor& row_indices, const at::Tensor& seqlens, const at::Tensor& cu_seqlens,
                                           const at::Tensor& /*tiles*/, const at::Tensor& /*cu_tiles*/, bool /*use_task_map*/) {
  TORCH_CHECK(row_indices.is_cuda() && row_indices.scalar_type() == at::kInt && row_indices.is_contiguous(),
              "row_indices must be a contiguous cuda int32 tensor");
  return group_gemm_cp_async(x, weight, y_scale, row_indices, seqlens, cu_seqlens);
}

// ---- reformat_x_scale (reference src/group_gemm/entry.cc:170-222) -----------------------------------------------------
at::Tensor reformat_x_scale(const at::Tensor& x_scale, const at::Tensor& seqlens, const at::Tensor& cu_seqlens,
                            const c10::optional<at::Tensor>& out_x_scale, int64_t num_seq_per_group_avg) {
  const at::Tensor* ts[3] = {&x_scale, &seqlens, &cu_seqlens};
  const char* names[3] = {"x_scale", "seqlens", "cu_seqlens"};
  for (int i = 0; i < 3; ++i) {
    TORCH_CHECK(ts[i]->is_cuda(), names[i], " tensor must be cuda");
    TORCH_CHECK(ts[i]->is_contiguous(), names[i], " tensor a must be contiguous");
  }
  TORCH_CHECK(x_scale.scalar_type() == at::kFloat && x_scale.dim() == 2, "x_scale must be float32 [rows, K/128]");
  const int64_t m = x_scale.size(0), n = x_scale.size(1), num_group = seqlens.size(0);
  const int64_t avg = num_seq_per_group_avg;
  const int tilem = avg <= 8 ? 8 : avg <= 16 ? 16 : avg <= 32 ? 32 : avg <= 48 ? 48 : 64;
  TORCH_CHECK((m / num_group) % tilem == 0,
              "The sparse pad length of x_scale for each group must be aligned to multiple of 8/16/32/48/64 according to "
              "num_seq_per_group_avg");
  at::Tensor out = out_x_scale.has_value() ? *out_x_scale : at::empty({n, m}, x_scale.options());
  TORCH_CHECK(out.is_contiguous() && out.scalar_type() == at::kFloat && out.numel() >= n * m,
              "out_x_scale must be a contiguous float32 [K/128, rows] tensor");
  HPC_LAUNCH_CHECK(hpc_reformat_x_scale_async(ptr(out), ptr(x_scale), ptr(seqlens), ptr(cu_seqlens), i32(num_group), i32(m), i32(n),
                                              tilem, stream_of(x_scale)),
                   "reformat_x_scale");
  return out;
}

// ---- activation + quantisation (reference src/activation/entry.cc) ----------------------------------------------------
at::Tensor act_mul_and_quant(const at::Tensor& gate_up, const at::Tensor& scale, bool use_bf16_mul,
                             const c10::optional<at::Tensor>& output) {
  // reference act_mul_and_quant_entry, src/activation/entry.cc:17-47: ANY rank - rows = product of the leading dims,
  // the output has the input's shape with the last dim halved.  The reference writes through a caller's `output`
  // unchecked; here a wrong dtype / size / layout is refused instead of becoming an out-of-bounds device write.
  cuda_contig(gate_up, "gate_up");
  TORCH_CHECK(gate_up.scalar_type() == at::kBFloat16 && gate_up.dim() >= 1, "gate_up must be bfloat16 [..., 2*C]");
  TORCH_CHECK(gate_up.size(-1) % 2 == 0, "the last dim of gate_up must be even (gate | up)");
  TORCH_CHECK(scale.is_cuda() && scale.scalar_type() == at::kFloat && scale.numel() >= 1,
              "scale must be a cuda float32 tensor with at least one element");
  const int64_t inter = gate_up.size(-1) / 2;
  const int64_t rows = inter > 0 ? gate_up.numel() / (2 * inter) : 0;
  std::vector<int64_t> out_shape(gate_up.sizes().begin(), gate_up.sizes().end());
  out_shape.back() = inter;
  at::Tensor out = output.has_value() ? *output : at::empty(out_shape, gate_up.options().dtype(kF8));
  if (output.has_value()) {
    TORCH_CHECK(out.is_cuda() && out.is_contiguous(), "output must be a contiguous cuda tensor");
    TORCH_CHECK(out.scalar_type() == kF8, "output dtype must be float8_e4m3fn");
    TORCH_CHECK(out.numel() == rows * inter, "output must hold gate_up's shape with the last dim halved");
  }
  if (rows == 0 || inter == 0) return out;
  const int rc = hpc_act_mul_and_quant_async(ptr(out), ptr(gate_up), ptr(scale), nullptr, i32(rows), i32(inter), use_bf16_mul ? 1 : 0,
                                             stream_of(gate_up));
  HPC_LAUNCH_CHECK(rc, "act_mul_and_quant_async");
  return out;
}

// reference scaled_fp8_quant_entry, src/activation/entry.cc:158-200: out = e4m3(input * (1 / scale[0])), returns (output, scale)
std::tuple<at::Tensor, at::Tensor> scaled_fp8_quant(const at::Tensor& input, const c10::optional<at::Tensor>& scale,
                                                    const c10::optional<at::Tensor>& output) {
  TORCH_CHECK(input.is_cuda(), "input must be a CUDA tensor");
  TORCH_CHECK(input.is_contiguous(), "input must be contiguous");
  TORCH_CHECK(input.numel() > 0, "input must be non-empty");
  const auto st = input.scalar_type();
  TORCH_CHECK(st == at::kFloat || st == at::kHalf || st == at::kBFloat16, "input dtype must be float32, float16, or bfloat16");
  at::Tensor out = output.has_value() ? *output : at::empty_like(input, input.options().dtype(kF8));
  TORCH_CHECK(out.is_cuda(), "output must be a CUDA tensor");
  TORCH_CHECK(out.is_contiguous(), "output must be contiguous");
  TORCH_CHECK(out.sizes() == input.sizes(), "output shape must match input shape");
  TORCH_CHECK(out.scalar_type() == kF8, "output dtype must be float8_e4m3fn");
  TORCH_CHECK(scale.has_value(), "scale is required for scaled_fp8_quant");
  const at::Tensor& sc = *scale;
  TORCH_CHECK(sc.is_cuda(), "scale must be a CUDA tensor");
  TORCH_CHECK(sc.scalar_type() == at::kFloat, "scale dtype must be float32");
  TORCH_CHECK(sc.numel() == 1, "scale must contain one element");
  const int in_dtype = st == at::kBFloat16 ? 0 : (st == at::kHalf ? 1 : 2);
  HPC_LAUNCH_CHECK(hpc_scaled_fp8_quant_async(ptr(out), ptr(input), ptr(sc), input.numel(), in_dtype, stream_of(input)),
                   "scaled_fp8_quant_async");
  return std::make_tuple(out, sc);
}

// masked (DeepEP-layout) variants, reference src/activation/entry.cc:50-156
struct Masked {
  int64_t total, inter, per;
};
Masked masked_common(const at::Tensor& input, const at::Tensor& num_per_expert) {
  TORCH_CHECK(input.is_contiguous(), "input tensor must be contiguous");
  TORCH_CHECK(num_per_expert.is_contiguous(), "num_per_expert tensor must be contiguous");
  TORCH_CHECK(input.is_cuda(), "input tensor's device must be cuda");
  TORCH_CHECK(num_per_expert.is_cuda(), "num_per_expert tensor's device must be cuda");
  TORCH_CHECK(input.scalar_type() == at::kBFloat16 && input.dim() == 2, "input must be bfloat16 [N, 2*C]");
  TORCH_CHECK(num_per_expert.scalar_type() == at::kInt, "num_per_expert must be int32");
  const int64_t num_experts = num_per_expert.size(0), total = input.size(0);
  TORCH_CHECK(num_experts > 0 && total % num_experts == 0, "rows must be num_expert * num_token_padded_per_expert");
  return {total, input.size(1) / 2, total / num_experts};
}
at::Tensor masked_act_mul_and_quant(const at::Tensor& input, const at::Tensor& scale, const at::Tensor& num_per_expert,
                                    const c10::optional<at::Tensor>& output) {
  const Masked c = masked_common(input, num_per_expert);
  TORCH_CHECK(scale.is_contiguous() && scale.is_cuda(), "scale tensor must be contiguous, on cuda");
  TORCH_CHECK(c.inter % 8 == 0, "hidden dim must be divided by 8");
  TORCH_CHECK(scale.numel() == 1 && scale.scalar_type() == at::kFloat, "only support per tensor qunat");
  at::Tensor out = output.has_value() ? *output : at::empty({c.total, c.inter}, input.options().dtype(kF8));
  HPC_LAUNCH_CHECK(hpc_masked_act_mul_and_quant_async(ptr(out), ptr(input), ptr(scale), ptr(num_per_expert), i32(c.total), i32(c.inter),
                                                      i32(c.per), stream_of(input)),
                   "masked_act_mul_and_quant_async");
  return out;
}
std::tuple<at::Tensor, at::Tensor> masked_act_mul_and_blockwise_quant(const at::Tensor& input, const at::Tensor& num_per_expert,
                                                                      const c10::optional<at::Tensor>& output,
                                                                      const c10::optional<at::Tensor>& output_scale) {
  const Masked c = masked_common(input, num_per_expert);
  TORCH_CHECK(c.inter % 128 == 0, "hidden dim must be divided by 128");
  at::Tensor out = output.has_value() ? *output : at::empty({c.total, c.inter}, input.options().dtype(kF8));
  at::Tensor osc = output_scale.has_value() ? *output_scale : at::empty({c.total, c.inter / 128}, input.options().dtype(at::kFloat));
  TORCH_CHECK(osc.is_contiguous() && osc.scalar_type() == at::kFloat, "output_scale must be contiguous float32");
  HPC_LAUNCH_CHECK(hpc_masked_act_mul_and_blockwise_quant_async(ptr(out), ptr(osc), ptr(input), ptr(num_per_expert), i32(c.total),
                                                                i32(c.inter), i32(c.per), stream_of(input)),
                   "masked_act_mul_and_blockwise_quant_async");
  return std::make_tuple(out, osc);
}

}  // namespace

// schema strings verbatim from the reference (tests/golden/ref_schemas.json, tests/test_schemas.py)
TORCH_LIBRARY_FRAGMENT(hpc, m) {
  m.def("reduce(Tensor x, Tensor topk_pos, Tensor topk_scale, Tensor ? shared_output) -> (Tensor)");
  m.def(
      "group_gemm_blockwise_fp8(Tensor x, Tensor weight, Tensor seqlens, Tensor cu_seqlens, Tensor xscale, Tensor wscale,"
      "int num_seq_per_group_avg, Tensor? output, Tensor? tma_desc, Tensor? task_map_workspace) -> (Tensor)");
  m.def(
      "fuse_moe(Tensor x, Tensor gate_up_weight, Tensor down_weight, Tensor gate_up_scale, Tensor down_scale, Tensor "
      "act_and_mul_scale, Tensor topk_ids, Tensor topk_scale, Tensor ? shared_output, int rank_ep, int num_expert_total, bool "
      "use_bf16_mul, Tensor ? output) -> (Tensor)");
  m.def(
      "fuse_moe_pertensor_fp8(Tensor x, Tensor gate_up_weight, Tensor down_weight, Tensor gate_up_scale, Tensor down_scale, "
      "Tensor act_and_mul_scale, Tensor topk_ids, Tensor topk_scale, Tensor ? shared_output, int rank_ep, int num_expert_total, "
      "bool use_bf16_mul, Tensor ? output) -> (Tensor)");
  m.def(
      "count_and_gather(Tensor x, Tensor topk_ids, int num_expert, int rank_ep, int intermediate_size, int "
      "num_seq_per_group_avg) -> (Tensor, Tensor, Tensor, Tensor, Tensor, Tensor, Tensor, Tensor, Tensor)");
  m.def(
      "group_gemm_fp8(Tensor x, Tensor weight, Tensor seqlens, Tensor cu_seqlens, Tensor y_scale, int num_seq_per_group_avg, "
      "Tensor? output, Tensor? tma_desc, Tensor? task_map_workspace) -> (Tensor)");
  m.def(
      "group_gemm_pertensor_fp8(Tensor x, Tensor weight, Tensor seqlens, Tensor cu_seqlens, Tensor y_scale, int "
      "num_seq_per_group_avg, Tensor? output, Tensor? tma_desc, Tensor? task_map_workspace) -> (Tensor)");
  m.def(
      "group_gemm_fp8_cp_async(Tensor x, Tensor weight, Tensor y_scale, Tensor seqlens, Tensor cu_seqlens, Tensor tiles, Tensor "
      "cu_tiles, bool use_task_map=False) -> (Tensor)");
  m.def(
      "group_gemm_fp8_scatter_cp_async(Tensor x, Tensor weight, Tensor y_scale, Tensor row_indices, Tensor seqlens, Tensor "
      "cu_seqlens, Tensor tiles, Tensor cu_tiles, bool use_task_map=False) -> (Tensor)");
  m.def("reformat_x_scale(Tensor x_scale, Tensor seqlens, Tensor cu_seqlens, Tensor? out_x_scale, int num_seq_per_group_avg) -> (Tensor)");
  m.def("act_mul_and_quant(Tensor input, Tensor scale, bool use_bf16_mul, Tensor? output) -> (Tensor)");
  m.def("scaled_fp8_quant(Tensor input, Tensor? scale, Tensor? output) -> (Tensor, Tensor)");
  m.def("masked_act_mul_and_quant(Tensor input, Tensor scale, Tensor num_per_expert, Tensor? output) -> (Tensor)");
  m.def(
      "masked_act_mul_and_blockwise_quant(Tensor input, Tensor num_per_expert, Tensor? output, Tensor? output_scale) -> "
      "(Tensor output, Tensor output_scale)");
}

TORCH_LIBRARY_IMPL(hpc, CUDA, m) {
  m.impl("reduce", &reduce);
  m.impl("group_gemm_blockwise_fp8", &group_gemm_blockwise_fp8);
  m.impl("fuse_moe", &fuse_moe);
  m.impl("fuse_moe_pertensor_fp8", &fuse_moe);
  m.impl("count_and_gather", &count_and_gather);
  m.impl("group_gemm_fp8", &group_gemm_fp8);
  m.impl("group_gemm_pertensor_fp8", &group_gemm_fp8);
  m.impl("group_gemm_fp8_cp_async", &group_gemm_fp8_cp_async);
  m.impl("group_gemm_fp8_scatter_cp_async", &group_gemm_fp8_scatter_cp_async);
  m.impl("reformat_x_scale", &reformat_x_scale);
  m.impl("act_mul_and_quant", &act_mul_and_quant);
  m.impl("scaled_fp8_quant", &scaled_fp8_quant);
  m.impl("masked_act_mul_and_quant", &masked_act_mul_and_quant);
  m.impl("masked_act_mul_and_blockwise_quant", &masked_act_mul_and_blockwise_quant);
}
