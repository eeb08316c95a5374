// C++ host side of the ops either side of attention + MoE in a decode step: router GEMM (+ the fused softmax / top-k
// router), RoPE + QK-norm + paged KV store (bf16 / fp8), the fused samplers, and the fused AllReduce + residual +
// RMSNorm (high-throughput and low-latency).
//
// Mirrors the reference's entries src/gemm/sm90/entry.cc:86-153, src/rope/entry.cc:14-240, src/sampler/entry.cc:13-275,
// src/allreduce/entry.cc:14-215 - schemas verbatim (tests/test_schemas.py), same checks / messages / output rules;
// compute behind the C-ABI (csrc/gemm_bf16xfp32.hip, topk_router.hip, rope.hip, sampler.hip, allreduce.hip).
#include <map>
#include <mutex>

#include "torch_common.h"

using namespace hpc_torch;

namespace {

const auto kF8 = at::kFloat8_e4m3fn;

// ---- router GEMM (reference gemm_bf16xfp32_entry, src/gemm/sm90/entry.cc:86-147) ------------------------------------
at::Tensor gemm_bf16xfp32(const at::Tensor& x, const at::Tensor& w_high, const at::Tensor& w_low, double scale, bool use_fp32_output,
                          bool use_splitk, const c10::optional<at::Tensor>& split_flag) {
  TORCH_CHECK(x.is_cuda(), "x must be a device tensor");
  TORCH_CHECK(x.is_contiguous(), "x tensor must be contiguous");
  TORCH_CHECK(w_high.is_contiguous(), "w_high tensor must be contiguous");
  TORCH_CHECK(w_low.is_contiguous(), "w_low tensor must be contiguous");
  TORCH_CHECK(x.scalar_type() == at::kBFloat16, "x dtype must be bfloat16");
  TORCH_CHECK(w_high.scalar_type() == at::kBFloat16, "w_high dtype must be bfloat16");
  TORCH_CHECK(w_low.scalar_type() == at::kBFloat16, "w_low dtype must be bfloat16");
  const int64_t m = x.size(0), k = x.size(1), n = w_high.size(0);
  TORCH_CHECK(n % 64 == 0, "n must to be divided by 64.");
  TORCH_CHECK(w_high.size(1) == k && w_low.sizes() == w_high.sizes(), "weight planes must be [n, k]");
  const int splits = hpc_gemm_bf16xfp32_splits(i32(m), i32(n), i32(k), use_splitk ? 1 : 0);
  at::Tensor split_y, flag;
  int64_t flag_ld = 0;
  if (splits > 1) {
    split_y = at::empty({splits, m, n}, x.options().dtype(at::kFloat));
    // counters: m <= 256 runs on 16-row tiles (flat [m_tiles, n / 16]), larger m on a [ceil(m / 64), n / 64] grid
    int64_t rows;
    if (m <= 256) {
      const int64_t tm = m <= 16 ? 16 : (m <= 32 ? 32 : 64);
      rows = (m + tm - 1) / tm, flag_ld = n / 16;
    } else {
      rows = (m + 63) / 64, flag_ld = n / 64;
    }
    if (split_flag.has_value()) {
      TORCH_CHECK(split_flag->scalar_type() == at::kInt && split_flag->is_contiguous(), "split_flag must be a contiguous int32 tensor");
      if (m > 256) {
        TORCH_CHECK(split_flag->dim() == 2 && split_flag->size(1) >= flag_ld && split_flag->size(0) >= rows,
                    "split_flag is too small for this problem");
        flag_ld = split_flag->size(1);
      } else {
        TORCH_CHECK(split_flag->numel() >= rows * flag_ld, "split_flag is too small for this problem");
      }
      flag = *split_flag;
    } else {
      // the tickets are zero again when the kernel exits (the last arriver of a tile resets its counter): a zero-once
      // buffer per (device, stream, capture) instead of an allocation + a zero-fill launch per call
      flag = cached_scratch(kScratchGemmFlags, x, std::max<int64_t>(rows * flag_ld * 4, 1 << 16), 1 << 30).view(at::kInt);
    }
  }
  at::Tensor y = at::empty({m, n}, x.options().dtype(use_fp32_output ? at::kFloat : at::kBFloat16));
  const int rc = hpc_gemm_bf16xfp32_async(ptr(y), split_y.defined() ? split_y.data_ptr() : nullptr,
                                          flag.defined() ? flag.data_ptr() : nullptr, ptr(x), ptr(w_high), ptr(w_low), i32(m), i32(n),
                                          i32(k), static_cast<float>(scale), use_fp32_output ? 1 : 0, splits, i32(flag_ld), stream_of(x));
  HPC_LAUNCH_CHECK(rc, "gemm_bf16xfp32");
  return y;
}

// ---- fused softmax + top-k router (no reference op: pinned to stable torch semantics, oracle/router.py) ----------------
std::tuple<at::Tensor, at::Tensor> topk_router(const at::Tensor& logits, int64_t topk, bool renormalize,
                                               const c10::optional<at::Tensor>& topk_ids_in,
                                               const c10::optional<at::Tensor>& topk_scale_in) {
  TORCH_CHECK(logits.is_cuda(), "logits must be a device tensor");
  TORCH_CHECK(logits.scalar_type() == at::kFloat, "logits dtype must be float32 (the router GEMM's fp32 output)");
  TORCH_CHECK(logits.dim() == 2 && logits.stride(1) == 1, "logits must be [num_tokens, num_expert] with unit expert stride");
  const int64_t m = logits.size(0), n = logits.size(1);
  TORCH_CHECK(n % 4 == 0 && n <= 1024, "num_expert must be a multiple of 4 and <= 1024");
  TORCH_CHECK(1 <= topk && topk <= std::min<int64_t>(n, 64), "topk must be in 1..min(num_expert, 64)");
  TORCH_CHECK(logits.stride(0) % 4 == 0 && reinterpret_cast<uintptr_t>(logits.data_ptr()) % 16 == 0,
              "logits rows must be 16-byte aligned");
  at::Tensor ids = topk_ids_in.has_value() ? *topk_ids_in : at::empty({m, topk}, logits.options().dtype(at::kInt));
  at::Tensor sc = topk_scale_in.has_value() ? *topk_scale_in : at::empty({m, topk}, logits.options());
  TORCH_CHECK(ids.scalar_type() == at::kInt && ids.is_contiguous() && ids.dim() == 2 && ids.size(0) == m && ids.size(1) == topk,
              "topk_ids must be a contiguous int32 [num_tokens, topk] tensor");
  TORCH_CHECK(sc.scalar_type() == at::kFloat && sc.is_contiguous() && sc.dim() == 2 && sc.size(0) == m && sc.size(1) == topk,
              "topk_scale must be a contiguous float32 [num_tokens, topk] tensor");
  const int rc = hpc_topk_router_async(static_cast<int*>(ids.data_ptr()), static_cast<float*>(sc.data_ptr()),
                                       static_cast<const float*>(logits.data_ptr()), i32(m), i32(n), logits.stride(0), i32(topk),
                                       renormalize ? 1 : 0, stream_of(logits));
  HPC_LAUNCH_CHECK(rc, "topk_router");
  return std::make_tuple(ids, sc);
}

// ---- RoPE + QK-norm + paged KV store (reference src/rope/entry.cc:14-222) ---------------------------------------------
struct RopeDims {
  int64_t num_q, num_kv;
};
RopeDims rope_common(const at::Tensor& kcache, const at::Tensor& vcache, const at::Tensor& qkv, const at::Tensor& cos_sin,
                     const at::Tensor& num_seqlen_per_req, const at::Tensor& q_index, const at::Tensor& kvcache_indices,
                     const c10::optional<at::Tensor>& q_norm_weight, const c10::optional<at::Tensor>& k_norm_weight,
                     int64_t qk_norm_policy) {
  TORCH_CHECK(qkv.is_cuda() && qkv.is_contiguous(), "qkv tensor must be contiguous");
  TORCH_CHECK(cos_sin.is_contiguous() && cos_sin.scalar_type() == at::kFloat, "cos_sin tensor must be contiguous float32");
  TORCH_CHECK(num_seqlen_per_req.is_contiguous() && num_seqlen_per_req.scalar_type() == at::kInt,
              "num_seqlen_per_req tensor must be contiguous int32");
  TORCH_CHECK(q_index.is_contiguous() && q_index.scalar_type() == at::kInt, "q_index must be contiguous int32");
  TORCH_CHECK(kvcache_indices.is_contiguous() && kvcache_indices.scalar_type() == at::kInt,
              "kvcache_indices tensor must be contiguous int32");
  TORCH_CHECK(0 <= qk_norm_policy && qk_norm_policy <= 2, "qk_norm_policy must be 0, 1 or 2");
  TORCH_CHECK(qkv.scalar_type() == at::kBFloat16, "qkv must be bfloat16");
  const int64_t num_kv = kcache.size(2), qk_dim = kcache.size(3), v_dim = vcache.size(3);
  TORCH_CHECK(qk_dim == 128 && v_dim == 128, "head dims must be 128");
  const int64_t hidden = qkv.size(1);
  const int64_t num_q = (hidden - num_kv * qk_dim - num_kv * v_dim) / qk_dim;
  TORCH_CHECK(num_q > 0 && (num_q + 2 * num_kv) * qk_dim == hidden, "qkv hidden size does not match the caches");
  for (const at::Tensor* c : {&kcache, &vcache})
    TORCH_CHECK(c->stride(3) == 1 && c->stride(2) == 128 && c->stride(1) == num_kv * 128,
                "kv cache pages must be [block_size, num_kv_heads, 128] contiguous");
  for (const auto* wt : {&q_norm_weight, &k_norm_weight})
    if (wt->has_value())
      TORCH_CHECK((*wt)->scalar_type() == at::kFloat && (*wt)->numel() == 128, "norm weights must be float32 [128]");
  if (qk_norm_policy)
    TORCH_CHECK(q_norm_weight.has_value() && k_norm_weight.has_value(),
                "q_norm_weight / k_norm_weight are required when qk_norm_policy != 0");
  return {num_q, num_kv};
}

at::Tensor rope_norm_store_kv(at::Tensor& kcache, at::Tensor& vcache, const at::Tensor& qkv, const at::Tensor& cos_sin,
                              const at::Tensor& num_seqlen_per_req, const at::Tensor& q_index, const at::Tensor& kvcache_indices,
                              bool is_prefill, const c10::optional<at::Tensor>& q_norm_weight,
                              const c10::optional<at::Tensor>& k_norm_weight, const c10::optional<at::Tensor>& out_q_in,
                              const c10::optional<at::Tensor>& out_k, const c10::optional<at::Tensor>& out_v, int64_t qk_norm_policy) {
  const RopeDims d = rope_common(kcache, vcache, qkv, cos_sin, num_seqlen_per_req, q_index, kvcache_indices, q_norm_weight,
                                 k_norm_weight, qk_norm_policy);
  TORCH_CHECK(kcache.scalar_type() == at::kBFloat16 && vcache.scalar_type() == at::kBFloat16, "caches must be bfloat16");
  const int64_t rows = qkv.size(0);
  at::Tensor out_q;
  if (out_q_in.has_value()) {
    TORCH_CHECK(out_q_in->is_contiguous(), "out_q tensor must be contiguous");
    out_q = *out_q_in;
  } else {
    out_q = at::empty({rows, d.num_q, 128}, qkv.options());
  }
  if (out_k.has_value()) TORCH_CHECK(out_k->is_contiguous(), "out_k tensor must be contiguous");
  if (out_v.has_value()) TORCH_CHECK(out_v->is_contiguous(), "out_v tensor must be contiguous");
  const int rc = hpc_rope_norm_store_kv_async(ptr(out_q), ptr(kcache), ptr(vcache), ptr(out_k), ptr(out_v), ptr(qkv), ptr(cos_sin),
                                              ptr(num_seqlen_per_req), ptr(q_index), ptr(kvcache_indices), ptr(q_norm_weight),
                                              ptr(k_norm_weight), kcache.stride(0), vcache.stride(0), i32(num_seqlen_per_req.size(0)),
                                              i32(kvcache_indices.size(1)), i32(kcache.size(1)), i32(rows), i32(d.num_q), i32(d.num_kv),
                                              128, 128, is_prefill ? 1 : 0, i32(qk_norm_policy), stream_of(qkv));
  HPC_LAUNCH_CHECK(rc, "rope_norm_store_kv_async");
  return out_q;
}

std::tuple<at::Tensor, at::Tensor, at::Tensor> rope_norm_store_kv_fp8(
    at::Tensor& kcache, at::Tensor& vcache, const at::Tensor& qkv, const at::Tensor& cos_sin, const at::Tensor& num_seqlen_per_req,
    const at::Tensor& q_index, const at::Tensor& kvcache_indices, bool is_prefill, const at::Tensor& k_scale, const at::Tensor& v_scale,
    int64_t quant_policy, int64_t max_seqlens, c10::optional<double> upper_max, const c10::optional<at::Tensor>& q_scale_inv,
    const c10::optional<at::Tensor>& q_norm_weight, const c10::optional<at::Tensor>& k_norm_weight,
    const c10::optional<at::Tensor>& out_q_in, const c10::optional<at::Tensor>& out_k, const c10::optional<at::Tensor>& out_v,
    int64_t qk_norm_policy) {
  const RopeDims d = rope_common(kcache, vcache, qkv, cos_sin, num_seqlen_per_req, q_index, kvcache_indices, q_norm_weight,
                                 k_norm_weight, qk_norm_policy);
  TORCH_CHECK(k_scale.dim() == 1 && k_scale.size(0) == 1, "k_scale must contain 1 element");
  TORCH_CHECK(v_scale.dim() == 1 && v_scale.size(0) == 1, "v_scale must contain 1 element");
  TORCH_CHECK(quant_policy == 1 || quant_policy == 2, "quant_policy must be 1 or 2");
  TORCH_CHECK(kcache.element_size() == 1 && vcache.element_size() == 1, "caches must be 1-byte dtype");
  float fp8_max = 448.0f;
  if (upper_max.has_value()) {
    TORCH_CHECK(!(*upper_max > fp8_max), "upper_max should not be larger than fp8_max");
    fp8_max = static_cast<float>(*upper_max);
  }
  const int64_t rows = qkv.size(0), num_req = num_seqlen_per_req.size(0);
  at::Tensor out_q;
  if (out_q_in.has_value()) {
    TORCH_CHECK(out_q_in->is_contiguous() && out_q_in->scalar_type() == kF8, "out_q must be contiguous float8_e4m3fn");
    out_q = *out_q_in;
  } else {
    out_q = at::empty({rows, d.num_q, 128}, qkv.options().dtype(kF8));
  }
  // q_scale: dynamic per-token-per-head scales get real storage, static q scales an undefined tensor (reference :150-165)
  at::Tensor q_scale;
  int64_t pad = 0;
  if (quant_policy == 1) {
    if (is_prefill) {
      pad = (max_seqlens + 127) / 128 * 128;
      q_scale = at::empty({num_req, d.num_q, pad}, qkv.options().dtype(at::kFloat));
    } else {
      q_scale = at::empty({rows, d.num_q}, qkv.options().dtype(at::kFloat));
    }
  } else {
    TORCH_CHECK(q_scale_inv.has_value() && q_scale_inv->scalar_type() == at::kFloat, "q_scale_inv required for quant_policy=2");
  }
  at::Tensor split_k_flag = at::empty({num_req, d.num_kv}, qkv.options().dtype(at::kInt));
  if (out_k.has_value()) TORCH_CHECK(out_k->is_contiguous() && out_k->scalar_type() == kF8, "out_k must be contiguous float8_e4m3fn");
  if (out_v.has_value()) TORCH_CHECK(out_v->is_contiguous() && out_v->scalar_type() == kF8, "out_v must be contiguous float8_e4m3fn");
  const int rc = hpc_rope_norm_store_kv_fp8_async(
      ptr(out_q), ptr(kcache), ptr(vcache), ptr(out_k), ptr(out_v), ptr(split_k_flag), q_scale.defined() ? q_scale.data_ptr() : nullptr,
      ptr(qkv), ptr(cos_sin), ptr(num_seqlen_per_req), ptr(q_index), ptr(kvcache_indices), ptr(q_norm_weight), ptr(k_norm_weight),
      ptr(k_scale), ptr(v_scale), ptr(q_scale_inv), fp8_max, i32(pad), kcache.stride(0), vcache.stride(0), i32(num_req),
      i32(kvcache_indices.size(1)), i32(kcache.size(1)), i32(rows), i32(d.num_q), i32(d.num_kv), 128, 128, is_prefill ? 1 : 0,
      i32(qk_norm_policy), i32(quant_policy), stream_of(qkv));
  HPC_LAUNCH_CHECK(rc, "rope_norm_store_kv_fp8_async");
  return std::make_tuple(out_q, q_scale, split_k_flag);
}

// ---- fused samplers (reference src/sampler/entry.cc:13-255) -----------------------------------------------------------
struct LogitsDims {
  int64_t b, v;
};
LogitsDims check_logits(const at::Tensor& logits, const char* who) {
  TORCH_CHECK(logits.is_cuda(), "logits must be a device tensor");
  TORCH_CHECK(logits.dim() == 2, "logits tensor must be dim == 2");
  TORCH_CHECK(logits.scalar_type() == at::kFloat || logits.scalar_type() == at::kBFloat16, "logits dtype must be float32 or bfloat16");
  const int64_t b = logits.size(0), v = logits.size(1);
  TORCH_CHECK(logits.stride(1) == 1, who, ": logits must have contiguous inner dim (stride(1)=1), got stride(1)=", logits.stride(1));
  TORCH_CHECK(logits.stride(0) >= v, who, ": logits stride(0)=", logits.stride(0), " must be >= vocab_size=", v);
  TORCH_CHECK(v % 8 == 0 && v < (1 << 20), who, ": unsupported vocab_size ", v, " (must be a multiple of 8, < 2^20)");
  return {b, v};
}
void check_1d_float(const c10::optional<at::Tensor>& t, const char* name, int64_t b) {
  if (!t.has_value()) return;
  TORCH_CHECK(t->is_contiguous(), name, " tensor must be contiguous");
  TORCH_CHECK(t->scalar_type() == at::kFloat, name, " dtype must be float32");
  TORCH_CHECK(t->dim() == 1, name, " tensor must be 1D");
  TORCH_CHECK(t->size(0) == b, name, " size must be [batch_size=", b, "], got [", t->size(0), "]");
}
void check_noise(const c10::optional<at::Tensor>& g, int64_t b, int64_t v) {
  if (!g.has_value()) return;
  TORCH_CHECK(g->is_contiguous(), "gumbel_noise tensor must be contiguous");
  TORCH_CHECK(g->scalar_type() == at::kFloat, "gumbel_noise dtype must be float32");
  TORCH_CHECK(g->dim() == 2, "gumbel_noise must be 2D");
  TORCH_CHECK(g->size(0) == b && g->size(1) == v, "gumbel_noise shape must be [", b, ", ", v, "]");
}

at::Tensor fused_sampler(const at::Tensor& logits, const c10::optional<at::Tensor>& penalty_mask, const c10::optional<at::Tensor>& slot_id,
                         const c10::optional<at::Tensor>& repetition_penalty, double repetition_penalty_val,
                         const c10::optional<at::Tensor>& temperature, double temperature_val, int64_t softmax_policy,
                         const c10::optional<at::Tensor>& topk, int64_t topk_val, const c10::optional<at::Tensor>& topp, double topp_val,
                         int64_t max_topk, const c10::optional<at::Tensor>& gumbel_noise, int64_t seed) {
  const LogitsDims d = check_logits(logits, "fused_sampler");
  const int64_t b = d.b, v = d.v;
  TORCH_CHECK(0 <= softmax_policy && softmax_policy <= 2, "softmax_policy must be one of 0(NONE)/1(BEFORE_TOPK)/2(AFTER_TOPK)");
  TORCH_CHECK(penalty_mask.has_value() == slot_id.has_value(), "penalty_mask and slot_id must both be provided or both be omitted");
  if (penalty_mask.has_value()) {
    TORCH_CHECK(penalty_mask->is_contiguous(), "penalty_mask tensor must be contiguous");
    TORCH_CHECK(penalty_mask->scalar_type() == at::kByte, "penalty_mask dtype must be uint8");
    TORCH_CHECK(penalty_mask->dim() == 2, "penalty_mask must be 2D [MAX_BS, ceil(V/8)]");
    TORCH_CHECK(penalty_mask->size(1) >= (v + 7) / 8, "penalty_mask dim 1 must be >= ", (v + 7) / 8, ", got ", penalty_mask->size(1));
    TORCH_CHECK(slot_id->is_contiguous(), "slot_id tensor must be contiguous");
    TORCH_CHECK(slot_id->scalar_type() == at::kInt, "slot_id dtype must be int32");
    TORCH_CHECK(slot_id->dim() == 1, "slot_id must be 1D");
    TORCH_CHECK(slot_id->size(0) == b, "slot_id.size(0) must equal batch_size=", b, ", got ", slot_id->size(0));
    TORCH_CHECK(penalty_mask->size(0) >= b, "penalty_mask.size(0)(MAX_BS) must be >= batch_size=", b);
  }
  check_1d_float(repetition_penalty, "repetition_penalty", b);
  check_1d_float(temperature, "temperature", b);
  check_1d_float(topp, "topp", b);
  int topk_bytes = 0;
  if (topk.has_value()) {
    TORCH_CHECK(topk->is_contiguous(), "topk tensor must be contiguous");
    TORCH_CHECK(topk->dim() == 1, "topk tensor must be 1D");
    TORCH_CHECK(topk->size(0) == b, "topk size must be [batch_size=", b, "], got [", topk->size(0), "]");
    TORCH_CHECK(topk->scalar_type() == at::kInt || topk->scalar_type() == at::kLong, "topk dtype must be int32 or int64");
    topk_bytes = topk->scalar_type() == at::kInt ? 4 : 8;
  }
  const bool has_rp = repetition_penalty.has_value() || repetition_penalty_val > 0.0;
  const bool has_topk = topk.has_value() || topk_val > 0;
  const bool has_topp = topp.has_value() || topp_val > 0.0;
  TORCH_CHECK(!has_rp || penalty_mask.has_value(), "repetition_penalty is enabled but penalty_mask/slot_id are missing");
  TORCH_CHECK(!has_topp || has_topk, "topp requires topk to be enabled (kernel does not support bare topp)");
  TORCH_CHECK(!has_topp || softmax_policy != 0, "topp requires softmax_policy != NONE (BEFORE_TOPK or AFTER_TOPK)");
  TORCH_CHECK(softmax_policy == 0 || has_topp,
              "softmax_policy != NONE requires topp to be enabled (softmax has no effect on sampling without topp)");
  TORCH_CHECK(max_topk == 32 || max_topk == 64, "max_topk must be 32 or 64, got ", max_topk);
  check_noise(gumbel_noise, b, v);
  if (!gumbel_noise.has_value())
    TORCH_CHECK(seed > 0, "fused_sampler: seed must be > 0 when gumbel_noise is not provided, got seed=", seed);
  at::Tensor token_ids = at::empty({b, 1}, logits.options().dtype(at::kInt));
  if (b == 0) return token_ids;
  at::Tensor ws = at::empty({hpc_fused_sampler_workspace_bytes(i32(b), i32(v), i32(max_topk))}, logits.options().dtype(at::kByte));
  const int rc = hpc_fused_sampler_async(
      ptr(token_ids), ptr(ws), ptr(logits), logits.scalar_type() == at::kFloat ? 0 : 1, ptr(penalty_mask),
      penalty_mask.has_value() ? penalty_mask->stride(0) : 0, ptr(slot_id), ptr(repetition_penalty),
      static_cast<float>(repetition_penalty_val), ptr(temperature), static_cast<float>(temperature_val), i32(softmax_policy), ptr(topk),
      topk_bytes, i32(topk_val), ptr(topp), static_cast<float>(topp_val), ptr(gumbel_noise), i32(b), i32(v), logits.stride(0),
      i32(max_topk), gumbel_noise.has_value() ? 0ull : static_cast<uint64_t>(seed), stream_of(logits));
  HPC_LAUNCH_CHECK(rc, "fused_sampler_async");
  return token_ids;
}

at::Tensor fused_sampler_temperature_sample(const at::Tensor& logits, const c10::optional<at::Tensor>& temperature,
                                            double temperature_val, const c10::optional<at::Tensor>& gumbel_noise,
                                            const c10::optional<at::Tensor>& draft_token_ids, int64_t seed) {
  const LogitsDims d = check_logits(logits, "fused_sampler_temperature_sample");
  const int64_t b = d.b, v = d.v;
  if (temperature.has_value()) {
    check_1d_float(temperature, "temperature", b);
    if (temperature->numel()) {
      const float tmin = temperature->min().item<float>();
      TORCH_CHECK(tmin > 0.f, "fused_sampler_temperature_sample: every temperature tensor element must be > 0, got min=", tmin);
    }
  } else {
    TORCH_CHECK(temperature_val > 0.0, "fused_sampler_temperature_sample: scalar temperature must be > 0, got ", temperature_val);
  }
  check_noise(gumbel_noise, b, v);
  if (draft_token_ids.has_value()) {
    TORCH_CHECK(draft_token_ids->is_contiguous(), "draft_token_ids tensor must be contiguous");
    TORCH_CHECK(draft_token_ids->scalar_type() == at::kLong, "draft_token_ids dtype must be int64");
    TORCH_CHECK(draft_token_ids->dim() == 1, "draft_token_ids must be 1D");
    TORCH_CHECK(draft_token_ids->size(0) == b, "draft_token_ids size must be [batch_size=", b, "], got [", draft_token_ids->size(0), "]");
  }
  if (!gumbel_noise.has_value())
    TORCH_CHECK(seed > 0, "fused_sampler_temperature_sample: seed must be > 0 when gumbel_noise is not provided, got seed=", seed);
  at::Tensor token_ids = at::empty({b, 1}, logits.options().dtype(at::kInt));
  if (b == 0) return token_ids;
  at::Tensor ws = at::empty({b * hpc_sampler_segments(i32(v))}, logits.options().dtype(at::kLong));
  const int rc = hpc_fused_sampler_temperature_async(
      ptr(token_ids), ptr(ws), ptr(logits), logits.scalar_type() == at::kFloat ? 0 : 1, logits.stride(0), ptr(temperature),
      static_cast<float>(temperature_val), ptr(gumbel_noise), ptr(draft_token_ids), i32(b), i32(v),
      gumbel_noise.has_value() ? 0ull : static_cast<uint64_t>(seed), stream_of(logits));
  HPC_LAUNCH_CHECK(rc, "fused_sampler_temperature_async");
  return token_ids;
}

// ---- fused AllReduce + residual + RMSNorm (reference src/allreduce/entry.cc:14-194) -----------------------------------
void bf16_contig(const at::Tensor& t, const char* name) {
  TORCH_CHECK(t.is_cuda() && t.is_contiguous(), name, " tensor must be a contiguous cuda tensor");
  TORCH_CHECK(t.scalar_type() == at::kBFloat16, name, " must be bfloat16");
  TORCH_CHECK(reinterpret_cast<uintptr_t>(t.data_ptr()) % 16 == 0, name, " must be 16-byte aligned");
}

// peers' addresses of an address inside a local symmetric buffer (the communicator's registry; the reference gets them
// for free from the NVLS multicast mapping)
int lookup_peers(const at::Tensor& t, void* (&ptrs)[64]) {
  int rank = -1;
  const int n = hpc_comm_lookup_peers(t.data_ptr(), ptrs, &rank);
  TORCH_CHECK(n > 0, "tensor is not inside a buffer created by MulticastCommunicator.CreateTensorSync");
  TORCH_CHECK(rank >= 0 && rank < n, "symmetric buffer has no local rank");
  return n * 64 + rank;  // (world, rank) packed
}

// host copy of the device table of signal-pad addresses + the capacity (uint32 words) of this rank's pad, read ONCE per
// table (a per-call device-to-host copy would synchronise the stream and break graph capture)
struct SignalInfo {
  std::vector<void*> ptrs;
  int pad_words;
};
const SignalInfo& signal_info(const at::Tensor& signal, int64_t world_size, int64_t rank) {
  static std::mutex mu;
  static auto& cache = *new std::map<std::tuple<void*, int64_t, int64_t>, SignalInfo>();
  std::lock_guard<std::mutex> lock(mu);
  const auto key = std::make_tuple(signal.data_ptr(), world_size, rank);
  auto it = cache.find(key);
  if (it == cache.end()) {
    const at::Tensor host = signal.cpu();
    SignalInfo info;
    for (int64_t i = 0; i < world_size; ++i) info.ptrs.push_back(reinterpret_cast<void*>(host.data_ptr<int64_t>()[i]));
    const int64_t left = hpc_comm_region_bytes_left(info.ptrs[static_cast<size_t>(rank)]);
    TORCH_CHECK(left >= 4 * world_size, "signal pad is not inside a symmetric buffer of this process (or too small)");
    info.pad_words = static_cast<int>(std::min<int64_t>(left / 4, 2147483647));
    it = cache.emplace(key, std::move(info)).first;
  }
  return it->second;
}

void fuse_allreduce_rmsnorm_high_throughput(const at::Tensor& input, const at::Tensor& mc_input, const at::Tensor& in_residual,
                                            const at::Tensor& weight, const at::Tensor& signal, int64_t rank, int64_t world_size,
                                            int64_t num_max_blocks, double rms_norm_eps, const at::Tensor& output,
                                            const at::Tensor& mc_output, const at::Tensor& out_residual) {
  bf16_contig(input, "x");
  bf16_contig(mc_input, "multicast_x");
  bf16_contig(in_residual, "residual");
  bf16_contig(weight, "weight");
  bf16_contig(output, "output_x");
  bf16_contig(mc_output, "output_multicast_x");
  bf16_contig(out_residual, "output_residual");
  TORCH_CHECK(signal.scalar_type() == at::kLong && signal.numel() >= world_size, "signal must be int64 [world_size]");
  TORCH_CHECK(1 <= world_size && world_size <= 8, "world_size must be in 1..8");
  const int64_t rows = input.size(0), hidden = input.size(1);
  TORCH_CHECK(hidden % 8 == 0 && hidden <= 16384, "hidden_size must be a multiple of 8 and <= 16384");
  TORCH_CHECK(in_residual.dim() == 2 && in_residual.size(0) == rows && in_residual.size(1) == hidden && weight.numel() == hidden,
              "shape mismatch");
  void *in_ptrs[64], *out_ptrs[64];
  const int wi = lookup_peers(mc_input, in_ptrs), wo = lookup_peers(mc_output, out_ptrs);
  TORCH_CHECK(wi / 64 == world_size && wo / 64 == world_size && wi % 64 == rank && wo % 64 == rank,
              "multicast views do not belong to a communicator of this world_size / rank");
  const SignalInfo& sig = signal_info(signal, world_size, rank);
  const int rc = hpc_fuse_allreduce_rmsnorm_high_throughput_async(
      in_ptrs, out_ptrs, sig.ptrs.data(), ptr(in_residual), ptr(out_residual), ptr(weight), static_cast<float>(rms_norm_eps), i32(rows),
      i32(hidden), i32(rank), i32(world_size), i32(num_max_blocks), sig.pad_words, stream_of(input));
  HPC_LAUNCH_CHECK(rc, "fuse_allreduce_rmsnorm_high_throughput_async");
}

void fuse_allreduce_rmsnorm_low_latency(const at::Tensor& input_x, const at::Tensor& /*multicast_x*/, const at::Tensor& data_buffer_ptrs,
                                        at::Tensor& multinode_x, const at::Tensor& buffer_flags, int64_t world_size, int64_t rank,
                                        bool rmsnorm_fusion, bool /*launch_with_pdl*/, bool use_two_shot, at::Tensor& output_x,
                                        at::Tensor& residual_out, const at::Tensor& residual_in, const at::Tensor& weight_gamma,
                                        double rms_norm_eps) {
  bf16_contig(input_x, "input_x");
  bf16_contig(multinode_x, "multinode_x");
  bf16_contig(output_x, "output_x");
  bf16_contig(residual_in, "residual_in");
  bf16_contig(residual_out, "residual_out");
  bf16_contig(weight_gamma, "weight_gamma");
  TORCH_CHECK(data_buffer_ptrs.scalar_type() == at::kLong && data_buffer_ptrs.is_cuda() && data_buffer_ptrs.is_contiguous(),
              "data_buffer_ptrs must be a contiguous cuda int64 tensor");
  TORCH_CHECK(buffer_flags.is_cuda() && buffer_flags.is_contiguous() && buffer_flags.numel() >= 9 && buffer_flags.element_size() == 4,
              "buffer_flags must be 9 x uint32 on the device");
  TORCH_CHECK(input_x.dim() == 2, "input_x must be 2D [num_tokens, token_dim]");
  // rmsnorm_fusion = false: the all-reduce alone (the reference kernel's `wait_for_results` path, low_latency.cu:119-140);
  // use_two_shot = false: the one-shot Lamport form (the reference takes the flag and runs its two-shot kernel either way)
  const int64_t num_tokens = input_x.size(0), hidden = input_x.size(1);
  TORCH_CHECK(hidden % 8 == 0, "token_dim must be divisible by 8");
  TORCH_CHECK(output_x.dim() == 2 && output_x.size(0) == num_tokens && output_x.size(1) == hidden, "output_x shape mismatch");
  TORCH_CHECK(1 <= world_size && world_size <= 64 && 0 <= rank && rank < world_size, "bad world_size / rank");
  if (rmsnorm_fusion) {
    TORCH_CHECK(residual_in.dim() == 2 && residual_in.size(0) == num_tokens && residual_in.size(1) == hidden &&
                    residual_out.dim() == 2 && residual_out.size(0) == num_tokens && residual_out.size(1) == hidden,
                "residual shape mismatch");
    TORCH_CHECK(weight_gamma.dim() == 1 && weight_gamma.size(0) == hidden, "weight_gamma shape mismatch");
  }
  const int rc = hpc_allreduce_low_latency_async(
      ptr(output_x), ptr(residual_out), ptr(input_x), ptr(data_buffer_ptrs), ptr(multinode_x), ptr(buffer_flags), ptr(residual_in),
      ptr(weight_gamma), static_cast<float>(rms_norm_eps), i32(num_tokens), i32(hidden), i32(rank), i32(world_size),
      multinode_x.numel() * multinode_x.element_size(), rmsnorm_fusion ? 1 : 0, use_two_shot ? 1 : 0, stream_of(input_x));
  HPC_LAUNCH_CHECK(rc, "fuse_allreduce_rmsnorm_low_latency_async");
}

}  // namespace

// schema strings verbatim from the reference (tests/golden/ref_schemas.json, tests/test_schemas.py)
TORCH_LIBRARY_FRAGMENT(hpc, m) {
  m.def("gemm_bf16xfp32(Tensor x, Tensor w_high, Tensor w_low, float scale, bool use_fp32_output, bool use_splitk, Tensor? split_flag) -> (Tensor)");
  m.def("topk_router(Tensor logits, int topk, bool renormalize, Tensor? topk_ids, Tensor? topk_scale) -> (Tensor, Tensor)");
  m.def(
      "rope_norm_store_kv(Tensor! kcache, Tensor! vcache, Tensor qkv, Tensor cos_sin, Tensor num_seqlen_per_req, Tensor q_index, "
      "Tensor kvcache_indices, bool is_prefill, Tensor? q_norm_weight, Tensor? k_norm_weight, Tensor? out_q=None, Tensor? out_k=None, "
      "Tensor? out_v=None, int qk_norm_policy=0) -> Tensor");
  m.def(
      "rope_norm_store_kv_fp8(Tensor! kcache, Tensor! vcache, Tensor qkv, Tensor cos_sin, Tensor num_seqlen_per_req, Tensor q_index, "
      "Tensor kvcache_indices, bool is_prefill, Tensor k_scale, Tensor v_scale, int quant_policy, int max_seqlens, float? upper_max, "
      "Tensor? q_scale_inv, Tensor? q_norm_weight, Tensor? k_norm_weight, Tensor? out_q=None, Tensor? out_k=None, Tensor? out_v=None, "
      "int qk_norm_policy=0) -> (Tensor, Tensor, Tensor)");
  m.def(
      "fused_sampler(Tensor logits, Tensor? penalty_mask, Tensor? slot_id, Tensor? repetition_penalty, float repetition_penalty_val, "
      "Tensor? temperature, float temperature_val, int softmax_policy, Tensor? topk, int topk_val, Tensor? topp, float topp_val, "
      "int max_topk, Tensor? gumbel_noise=None, int seed=0) -> Tensor");
  m.def(
      "fused_sampler_temperature_sample(Tensor logits, Tensor? temperature, float temperature_val, Tensor? gumbel_noise=None, "
      "Tensor? draft_token_ids=None, int seed=0) -> Tensor");
  m.def(
      "fuse_allreduce_rmsnorm_high_throughput(Tensor input, Tensor mc_input, Tensor in_residual, Tensor weight, Tensor signal, int "
      "rank, int world_size, int num_max_blocks, float rms_norm_eps, Tensor output, Tensor mc_output, Tensor out_residual) -> ()");
  m.def(
      "fuse_allreduce_rmsnorm_low_latency(Tensor input_x, Tensor multicast_x, Tensor data_buffer_ptrs, Tensor! multinode_x, Tensor "
      "buffer_flags, int world_size, int rank, bool rmsnorm_fusion, bool launch_with_pdl, bool use_two_shot, Tensor! output_x, Tensor! "
      "residual_out, Tensor residual_in, Tensor weight_gamma, float rms_norm_eps) -> ()");
}

TORCH_LIBRARY_IMPL(hpc, CUDA, m) {
  m.impl("gemm_bf16xfp32", &gemm_bf16xfp32);
  m.impl("topk_router", &topk_router);
  m.impl("rope_norm_store_kv", &rope_norm_store_kv);
  m.impl("rope_norm_store_kv_fp8", &rope_norm_store_kv_fp8);
  m.impl("fused_sampler", &fused_sampler);
  m.impl("fused_sampler_temperature_sample", &fused_sampler_temperature_sample);
  m.impl("fuse_allreduce_rmsnorm_high_throughput", &fuse_allreduce_rmsnorm_high_throughput);
  m.impl("fuse_allreduce_rmsnorm_low_latency", &fuse_allreduce_rmsnorm_low_latency);
}
