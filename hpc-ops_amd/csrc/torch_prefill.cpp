// C++ host side of the prefill attention ops (FP8 paged, FP8 paged block-sparse, bf16 paged, bf16 contiguous).
//
// Mirrors the reference's entries src/attention/entry.cc:15-409 (attention_prefill_bf16_entry,
// attention_with_kvcache_prefill_bf16_entry, attention_with_kvcache_prefill_fp8_entry,
// attention_with_kvcache_blocksparse_prefill_fp8_entry) and registrations :822-850 - schemas verbatim, same checks and
// output rules; compute behind the C-ABI (csrc/attention_prefill.hip, csrc/attention_prefill_bf16.hip).
#include "torch_common.h"

using namespace hpc_torch;

namespace {

const auto kF8 = at::kFloat8_e4m3fn;

at::Tensor prefill_output(const at::Tensor& q, const c10::optional<at::Tensor>& output, int64_t dim_v) {
  const int64_t total_q = q.size(0), num_head_q = q.size(1);
  if (!output.has_value()) return at::empty({total_q, num_head_q, dim_v}, q.options().dtype(at::kBFloat16));
  TORCH_CHECK(output->is_cuda() && output->device() == q.device(), "output tensor must be on the same device as q");
  TORCH_CHECK(output->scalar_type() == at::kBFloat16, "output dtype must be bfloat16");
  TORCH_CHECK(output->is_contiguous(), "output tensor must be contiguous");
  TORCH_CHECK(output->dim() == 3 && output->size(0) == total_q && output->size(1) == num_head_q && output->size(2) == dim_v,
              "output must have shape [total_seq_q, num_head_q, num_dim_v]");
  return *output;
}

void int32_contig(const at::Tensor& t, const char* name) {
  TORCH_CHECK(t.scalar_type() == at::kInt && t.is_contiguous(), name, " must be contiguous int32");
}

// reference attention_with_kvcache_prefill_fp8_entry (:152-262) / ..._blocksparse_... (:264-409)
at::Tensor prefill_fp8(const at::Tensor& q, const at::Tensor& kcache, const at::Tensor& vcache, const at::Tensor& qscale,
                       const at::Tensor& kscale, const at::Tensor& vscale, const at::Tensor& cu_seqlens_q, const at::Tensor& block_ids,
                       const at::Tensor& seqlens_kvcache, int64_t max_seqlens_q, int64_t quant_type,
                       const c10::optional<at::Tensor>& output, const c10::optional<at::Tensor>& block_mask, bool blocksparse) {
  const at::Tensor* ts[9] = {&q, &kcache, &vcache, &qscale, &kscale, &vscale, &cu_seqlens_q, &block_ids, &seqlens_kvcache};
  const char* names[9] = {"q", "kcache", "vcache", "qscale", "kscale", "vscale", "cu_seqlens_q", "block_ids", "seqlens_kvcache"};
  for (int i = 0; i < 9; ++i) TORCH_CHECK(ts[i]->is_cuda(), names[i], " tensor must be cuda");
  TORCH_CHECK(quant_type == 0 || quant_type == 1, "quant_type only support 0/1");
  TORCH_CHECK(kscale.element_size() == 1 || kscale.element_size() == 4, "kscale dtype must be float or fp8");
  TORCH_CHECK(q.scalar_type() == kF8, "q dtype must be float8_e4m3fn");
  TORCH_CHECK(kcache.scalar_type() == kF8, "kcache dtype must be float8_e4m3fn");
  TORCH_CHECK(vcache.scalar_type() == kF8, "vcache dtype must be float8_e4m3fn");
  TORCH_CHECK(q.dim() == 3 && q.stride(2) == 1 && q.stride(1) == q.size(2), "q must be [total_seq, Hq, D] row-major");
  const int64_t total_q = q.size(0), num_head_q = q.size(1), dim_qk = q.size(2);
  const int64_t num_batch = cu_seqlens_q.size(0) - 1;
  const int64_t block_size = kcache.size(1), num_head_kv = kcache.size(2), dim_v = vcache.size(3);
  TORCH_CHECK(dim_qk == 128 && dim_v == 128, "attention_with_kvcache_prefill_fp8: expected dim_qk=128 and dim_v=128, got dim_qk=",
              dim_qk, " dim_v=", dim_v);
  TORCH_CHECK(kcache.stride(3) == 1 && vcache.stride(3) == 1, "kv cache head dim must be contiguous");
  TORCH_CHECK(qscale.scalar_type() == at::kFloat && qscale.dim() == 3 && qscale.is_contiguous() && qscale.size(0) == num_batch &&
                  qscale.size(1) == num_head_q && qscale.size(2) >= max_seqlens_q,
              "qscale must be float32 [num_batch, num_head_q, max_seqlens_q_pad]");
  int32_contig(cu_seqlens_q, "cu_seqlens_q");
  int32_contig(block_ids, "block_ids");
  int32_contig(seqlens_kvcache, "seqlens_kvcache");
  TORCH_CHECK(vscale.scalar_type() == at::kFloat, "vscale must be float32");
  int64_t ks[3] = {0, 0, 0};
  if (quant_type == 0) {
    TORCH_CHECK(kscale.dim() == 4 && kscale.stride(3) == 1, "per-token kscale must be the K-cache tail rows view");
    const int64_t es = kscale.element_size();
    ks[0] = kscale.stride(0) * es, ks[1] = kscale.stride(1) * es, ks[2] = kscale.stride(2) * es;
    TORCH_CHECK(vscale.numel() >= num_head_kv, "vscale must hold one value per kv head");
  } else {
    TORCH_CHECK(kscale.scalar_type() == at::kFloat && kscale.numel() >= 1, "kscale must be float32 [1]");
  }
  at::Tensor y = prefill_output(q, output, dim_v);
  if (total_q == 0) return y;
  if (blocksparse) {
    TORCH_CHECK(128 % block_size == 0, "unsupported block_size for FP8 blocksparse prefill");
    const int64_t tiles_m = (max_seqlens_q + 127) / 128;
    int64_t tiles_kv = 0;
    if (block_mask.has_value()) {
      TORCH_CHECK(block_mask->device() == q.device(), "block_mask tensor must be on the same device as q");
      TORCH_CHECK(block_mask->scalar_type() == at::kByte, "block_mask dtype must be uint8");
      TORCH_CHECK(block_mask->is_contiguous(), "block_mask tensor must be contiguous");
      TORCH_CHECK(block_mask->dim() == 4 && block_mask->size(0) == num_batch && block_mask->size(1) == num_head_q &&
                      block_mask->size(2) == tiles_m,
                  "block_mask must have shape [", num_batch, ", ", num_head_q, ", ", tiles_m,
                  ", Kb] where Kb = ceil(max_kv_len / kTileN=128)");
      tiles_kv = block_mask->size(3);
      TORCH_CHECK(tiles_kv > 0, "block_mask Kb dim must be > 0");
    }
    const int rc = hpc_attention_with_kvcache_blocksparse_prefill_fp8_async(
        ptr(y), ptr(q), ptr(kcache), ptr(vcache), ptr(qscale), ptr(kscale), ptr(vscale), ptr(cu_seqlens_q), ptr(block_ids),
        ptr(seqlens_kvcache), ptr(block_mask), i32(tiles_m), i32(tiles_kv), i32(quant_type), i32(num_batch), i32(max_seqlens_q),
        i32(qscale.size(2)), i32(dim_qk), i32(dim_v), i32(num_head_q), i32(num_head_kv), i32(block_size), i32(block_ids.size(1)),
        i32(y.stride(0)), i32(q.stride(0)), kcache.stride(0), kcache.stride(1), kcache.stride(2), vcache.stride(0), vcache.stride(1),
        vcache.stride(2), ks[0], ks[1], ks[2], stream_of(q));
    HPC_LAUNCH_CHECK(rc, "attention_with_kvcache_blocksparse_prefill_fp8");
    return y;
  }
  const int rc = hpc_attention_with_kvcache_prefill_fp8_async(
      ptr(y), ptr(q), ptr(kcache), ptr(vcache), ptr(qscale), ptr(kscale), ptr(vscale), ptr(cu_seqlens_q), ptr(block_ids),
      ptr(seqlens_kvcache), i32(quant_type), i32(num_batch), i32(max_seqlens_q), i32(qscale.size(2)), i32(dim_qk), i32(dim_v),
      i32(num_head_q), i32(num_head_kv), i32(block_size), i32(block_ids.size(1)), i32(y.stride(0)), i32(q.stride(0)),
      kcache.stride(0), kcache.stride(1), kcache.stride(2), vcache.stride(0), vcache.stride(1), vcache.stride(2), ks[0], ks[1], ks[2],
      stream_of(q));
  HPC_LAUNCH_CHECK(rc, "attention_with_kvcache_prefill_fp8");
  return y;
}

at::Tensor attention_with_kvcache_prefill_fp8(const at::Tensor& q, const at::Tensor& kcache, const at::Tensor& vcache,
                                              const at::Tensor& qscale, const at::Tensor& kscale, const at::Tensor& vscale,
                                              const at::Tensor& cu_seqlens_q, const at::Tensor& block_ids,
                                              const at::Tensor& num_seq_kvcache, int64_t max_seqlens_q, int64_t quant_type,
                                              const c10::optional<at::Tensor>& output) {
  return prefill_fp8(q, kcache, vcache, qscale, kscale, vscale, cu_seqlens_q, block_ids, num_seq_kvcache, max_seqlens_q, quant_type,
                     output, c10::nullopt, false);
}
at::Tensor attention_with_kvcache_blocksparse_prefill_fp8(const at::Tensor& q, const at::Tensor& kcache, const at::Tensor& vcache,
                                                          const at::Tensor& qscale, const at::Tensor& kscale, const at::Tensor& vscale,
                                                          const at::Tensor& cu_seqlens_q, const at::Tensor& block_ids,
                                                          const at::Tensor& num_seq_kvcache, int64_t max_seqlens_q, int64_t quant_type,
                                                          const c10::optional<at::Tensor>& block_mask,
                                                          const c10::optional<at::Tensor>& output) {
  return prefill_fp8(q, kcache, vcache, qscale, kscale, vscale, cu_seqlens_q, block_ids, num_seq_kvcache, max_seqlens_q, quant_type,
                     output, block_mask, true);
}

// reference attention_prefill_bf16_entry, src/attention/entry.cc:15-81
at::Tensor attention_prefill_bf16(const at::Tensor& q, const at::Tensor& k, const at::Tensor& v, const at::Tensor& seqlens_q,
                                  const at::Tensor& cu_seqlens_q, int64_t max_seqlens_q, const c10::optional<at::Tensor>& output) {
  const at::Tensor* ts[5] = {&q, &k, &v, &seqlens_q, &cu_seqlens_q};
  const char* names[5] = {"q", "k", "v", "seqlens_q", "cu_seqlens_q"};
  for (int i = 0; i < 5; ++i) TORCH_CHECK(ts[i]->is_cuda(), names[i], " tensor must be cuda");
  for (int i = 0; i < 3; ++i)
    TORCH_CHECK(ts[i]->scalar_type() == at::kBFloat16 && ts[i]->dim() == 3 && ts[i]->stride(2) == 1 && ts[i]->stride(1) == ts[i]->size(2),
                names[i], " must be bfloat16 [total_seq, heads, dim] with contiguous heads");
  int32_contig(cu_seqlens_q, "cu_seqlens_q");
  TORCH_CHECK(q.size(2) == 128 && k.size(2) == 128 && v.size(2) == 128, "attention_prefill_bf16: expected dim_qk=128 and dim_v=128");
  TORCH_CHECK(k.size(0) == q.size(0) && v.size(0) == q.size(0) && k.size(1) == v.size(1),
              "k / v must hold one row per q token and the same number of kv heads");
  at::Tensor y = prefill_output(q, output, v.size(2));
  if (q.size(0) == 0) return y;
  const int rc = hpc_attention_prefill_bf16_async(ptr(y), ptr(q), ptr(k), ptr(v), ptr(cu_seqlens_q), i32(cu_seqlens_q.size(0) - 1),
                                                  i32(max_seqlens_q), 128, 128, i32(q.size(1)), i32(k.size(1)), i32(y.stride(0)),
                                                  i32(q.stride(0)), i32(k.stride(0)), i32(v.stride(0)), stream_of(q));
  HPC_LAUNCH_CHECK(rc, "attention_prefill_bf16");
  return y;
}

// reference attention_with_kvcache_prefill_bf16_entry, src/attention/entry.cc:83-150
at::Tensor attention_with_kvcache_prefill_bf16(const at::Tensor& q, const at::Tensor& kcache, const at::Tensor& vcache,
                                               const at::Tensor& cu_seqlens_q, const at::Tensor& block_ids,
                                               const at::Tensor& num_seq_kvcache, int64_t max_seqlens_q,
                                               const c10::optional<at::Tensor>& output) {
  const at::Tensor* ts[6] = {&q, &kcache, &vcache, &cu_seqlens_q, &block_ids, &num_seq_kvcache};
  const char* names[6] = {"q", "kcache", "vcache", "cu_seqlens_q", "block_ids", "seqlens_kvcache"};
  for (int i = 0; i < 6; ++i) TORCH_CHECK(ts[i]->is_cuda(), names[i], " tensor must be cuda");
  for (int i = 0; i < 3; ++i) TORCH_CHECK(ts[i]->scalar_type() == at::kBFloat16, names[i], " dtype must be bfloat16");
  TORCH_CHECK(q.dim() == 3 && q.stride(2) == 1 && q.stride(1) == q.size(2), "q must be [total_seq, Hq, D] row-major");
  const int64_t dim_qk = q.size(2), dim_v = vcache.size(3);
  TORCH_CHECK(dim_qk == 128 && dim_v == 128, "attention_with_kvcache_prefill_bf16: expected dim_qk=128 and dim_v=128, got dim_qk=",
              dim_qk, " dim_v=", dim_v);
  TORCH_CHECK(kcache.stride(3) == 1 && vcache.stride(3) == 1, "kv cache head dim must be contiguous");
  int32_contig(cu_seqlens_q, "cu_seqlens_q");
  int32_contig(block_ids, "block_ids");
  int32_contig(num_seq_kvcache, "seqlens_kvcache");
  at::Tensor y = prefill_output(q, output, dim_v);
  if (q.size(0) == 0) return y;
  const int rc = hpc_attention_with_kvcache_prefill_bf16_async(
      ptr(y), ptr(q), ptr(kcache), ptr(vcache), ptr(cu_seqlens_q), ptr(block_ids), ptr(num_seq_kvcache), i32(cu_seqlens_q.size(0) - 1),
      i32(max_seqlens_q), i32(dim_qk), i32(dim_v), i32(q.size(1)), i32(kcache.size(2)), i32(kcache.size(1)), i32(block_ids.size(1)),
      i32(y.stride(0)), i32(q.stride(0)), kcache.stride(0), kcache.stride(1), kcache.stride(2), vcache.stride(0), vcache.stride(1),
      vcache.stride(2), stream_of(q));
  HPC_LAUNCH_CHECK(rc, "attention_with_kvcache_prefill_bf16");
  return y;
}

}  // namespace

// schema strings verbatim from the reference (src/attention/entry.cc:823-850; tests/test_schemas.py)
TORCH_LIBRARY_FRAGMENT(hpc, m) {
  m.def(
      "attention_prefill_bf16(Tensor q, Tensor k, Tensor v, Tensor seqlens_q, Tensor cu_seqlens_q, int max_seqlens_q, Tensor? "
      "output) -> (Tensor)");
  m.def(
      "attention_with_kvcache_prefill_bf16(Tensor q, Tensor kcache, Tensor vcache,"
      "Tensor cu_seqlens_q, Tensor block_ids, Tensor num_seq_kvcache, int max_seqlens_q, Tensor? output) -> (Tensor)");
  m.def(
      "attention_with_kvcache_prefill_fp8(Tensor q, Tensor kcache, Tensor vcache,"
      "Tensor qscale, Tensor kscale, Tensor vscale, Tensor cu_seqlens_q,"
      "Tensor block_ids, Tensor num_seq_kvcache, int max_seqlens_q, int quant_type,"
      "Tensor? output) -> (Tensor)");
  m.def(
      "attention_with_kvcache_blocksparse_prefill_fp8(Tensor q, Tensor kcache, Tensor vcache,"
      "Tensor qscale, Tensor kscale, Tensor vscale, Tensor cu_seqlens_q,"
      "Tensor block_ids, Tensor num_seq_kvcache, int max_seqlens_q, int quant_type,"
      "Tensor? block_mask, Tensor? output) -> (Tensor)");
}

TORCH_LIBRARY_IMPL(hpc, CUDA, m) {
  m.impl("attention_prefill_bf16", &attention_prefill_bf16);
  m.impl("attention_with_kvcache_prefill_bf16", &attention_with_kvcache_prefill_bf16);
  m.impl("attention_with_kvcache_prefill_fp8", &attention_with_kvcache_prefill_fp8);
  m.impl("attention_with_kvcache_blocksparse_prefill_fp8", &attention_with_kvcache_blocksparse_prefill_fp8);
}
