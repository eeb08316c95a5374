// Fused softmax + top-k router for gfx950: the step between the router GEMM (gemm_bf16xfp32.hip) and
// fuse_moe* - it produces the `topk_ids` / `topk_scale` tensors those ops consume.
//
// The reference has no such kernel (hpc/gemm.py:16-61 stops at the GEMM and callers use torch.topk +
// softmax in eager mode); BASELINE north_star asks for one on wavefront shuffles.  Semantics are pinned
// to the stable PyTorch formulation (oracle/router.py):
//     order   = stable descending sort of the logits  (ties -> smaller expert id first; NaN sorts
//               above +inf like torch.topk)
//     ids     = order[:, :k]                                   (int32, best first)
//     p       = softmax(logits, dim=-1) in fp32
//     weights = p[ids]                  (renormalize = 0)      or p[ids] / sum_j p[ids_j]  (renormalize = 1)
// Selecting on the logits instead of on p is the same choice mathematically (softmax is monotonic)
// and makes the indices independent of the exp implementation: the bar is torch.equal on ids.
//
// MI355X design: one wave per token row, the row held in registers (16 B per lane and 256 experts),
// no LDS.  A round of selection = every lane's best remaining element as a 64-bit composite
// (order-preserving key << 32 | ~expert id) followed by a wave-wide max through 6 shuffle steps; the
// winner is struck out in its owner lane.  k rounds give the top-k in rank order; the row maximum is
// round 0's winner, the softmax denominator one more shuffle reduction.
#include "hpc_common.h"
#include "../../include/hpc_amd.h"

namespace hpc {
namespace router {

constexpr int kThreads = 256;
constexpr int kWavesPerBlock = kThreads / 64;
constexpr int kMaxTopk = 64;

// Order-preserving integer key with torch.topk's conventions: -0.0 and +0.0 are EQUAL (the tie goes to the smaller
// expert id, like every other tie) and every NaN - whatever its sign bit - ranks above +inf.
__device__ __forceinline__ uint32_t key_of(float x) {
  uint32_t u = __float_as_uint(x + 0.0f);  // -0.0 + 0.0 = +0.0
  if (x != x) u = 0x7fc00000u;             // canonical positive NaN
  return (u & 0x80000000u) ? ~u : (u | 0x80000000u);  // monotonic: larger float <-> larger key
}

__device__ __forceinline__ uint64_t wave_max_u64(uint64_t v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    const uint32_t lo = __shfl_xor(static_cast<uint32_t>(v), o, 64);
    const uint32_t hi = __shfl_xor(static_cast<uint32_t>(v >> 32), o, 64);
    const uint64_t other = (static_cast<uint64_t>(hi) << 32) | lo;
    v = other > v ? other : v;
  }
  return v;
}

// kVec: float4 vectors per lane (experts <= 256 * kVec); element c of vector j of lane l is expert
// 256 j + 4 l + c, so a wave reads whole 1 KB row segments.
template <int kVec>
__global__ __launch_bounds__(kThreads) void topk_router_kernel(const float* __restrict__ logits,
                                                               int* __restrict__ ids,
                                                               float* __restrict__ weights, int num_tokens,
                                                               int num_expert, long ld, int topk,
                                                               int renormalize) {
  const int lane = threadIdx.x & 63;
  const int row = blockIdx.x * kWavesPerBlock + (threadIdx.x >> 6);
  if (row >= num_tokens) return;
  const float* src = logits + static_cast<long>(row) * ld;
  float x[kVec][4];
#pragma unroll
  for (int j = 0; j < kVec; ++j) {
    const int e0 = 256 * j + 4 * lane;
    if (e0 < num_expert) {
      const f32x4 v = *reinterpret_cast<const f32x4*>(src + e0);
#pragma unroll
      for (int c = 0; c < 4; ++c) x[j][c] = v[c];
    } else {
#pragma unroll
      for (int c = 0; c < 4; ++c) x[j][c] = -__builtin_inff();
    }
  }
  // composites of this lane's elements; 0 = struck out / not an expert
  uint64_t comp[kVec][4];
#pragma unroll
  for (int j = 0; j < kVec; ++j)
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      const int e = 256 * j + 4 * lane + c;
      comp[j][c] = e < num_expert ? (static_cast<uint64_t>(key_of(x[j][c])) << 32) | (0xffffffffu - e) : 0ull;
    }

  uint64_t mine = 0;  // lane r keeps the winner of round r (two rounds per lane when topk > 64: not supported)
  float row_max = 0.f;
  for (int r = 0; r < topk; ++r) {
    uint64_t best = 0;
#pragma unroll
    for (int j = 0; j < kVec; ++j)
#pragma unroll
      for (int c = 0; c < 4; ++c) best = comp[j][c] > best ? comp[j][c] : best;
    const uint64_t win = wave_max_u64(best);
#pragma unroll
    for (int j = 0; j < kVec; ++j)
#pragma unroll
      for (int c = 0; c < 4; ++c) comp[j][c] = comp[j][c] == win ? 0ull : comp[j][c];  // composites are unique
    if (lane == r) mine = win;
    if (r == 0) {
      const uint32_t k32 = static_cast<uint32_t>(win >> 32);
      const uint32_t bits = (k32 & 0x80000000u) ? (k32 & 0x7fffffffu) : ~k32;
      row_max = __uint_as_float(bits);
    }
  }
  // softmax statistics over the whole row (max = the first winner; -inf rows / NaN follow IEEE like torch)
  float denom = 0.f;
#pragma unroll
  for (int j = 0; j < kVec; ++j)
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      const int e = 256 * j + 4 * lane + c;
      if (e < num_expert) denom += expf(x[j][c] - row_max);
    }
  denom = wave_sum(denom);
  float p = 0.f;
  int id = 0;
  if (lane < topk) {
    id = static_cast<int>(0xffffffffu - static_cast<uint32_t>(mine));
    const uint32_t k32 = static_cast<uint32_t>(mine >> 32);
    const uint32_t bits = (k32 & 0x80000000u) ? (k32 & 0x7fffffffu) : ~k32;
    p = expf(__uint_as_float(bits) - row_max);
  }
  const float sel = wave_sum(p);
  if (lane < topk) {
    ids[static_cast<long>(row) * topk + lane] = id;
    weights[static_cast<long>(row) * topk + lane] = p / (renormalize ? sel : denom);
  }
}

}  // namespace router
}  // namespace hpc

extern "C" int hpc_topk_router_async(int* topk_ids, float* topk_scale, const float* logits, int num_tokens,
                                     int num_expert, int64_t ld_logits, int topk, int renormalize,
                                     hipStream_t stream) {
  using namespace hpc::router;
  if (!topk_ids || !topk_scale || !logits) return HPC_ERR_INVALID;
  if (num_tokens < 0 || num_expert <= 0 || topk <= 0 || topk > num_expert) return HPC_ERR_INVALID;
  if (topk > kMaxTopk || num_expert > 1024 || (num_expert & 3) || (ld_logits & 3) || ld_logits < num_expert)
    return HPC_ERR_UNSUPPORTED;  // 16-byte row segments; one winner per lane
  if ((reinterpret_cast<uintptr_t>(logits) & 15) != 0) return HPC_ERR_UNSUPPORTED;
  if (num_tokens == 0) return HPC_OK;
  const int grid = (num_tokens + kWavesPerBlock - 1) / kWavesPerBlock;
  const int vec = (num_expert + 255) / 256;
#define HPC_ROUTER_LAUNCH(V)                                                                              \
  topk_router_kernel<V><<<grid, kThreads, 0, stream>>>(logits, topk_ids, topk_scale, num_tokens, num_expert, \
                                                       ld_logits, topk, renormalize)
  if (vec == 1) {
    HPC_ROUTER_LAUNCH(1);
  } else if (vec == 2) {
    HPC_ROUTER_LAUNCH(2);
  } else {
    HPC_ROUTER_LAUNCH(4);
  }
#undef HPC_ROUTER_LAUNCH
  HPC_CHECK_LAUNCH();
  return HPC_OK;
}
