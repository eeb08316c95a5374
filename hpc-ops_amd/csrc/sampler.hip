// Fused sampler: repetition penalty -> temperature -> [softmax] -> top-k -> [softmax] -> top-p ->
// Gumbel-max -> penalty write-back, and the temperature-only fast path - gfx950.
//
// Replaces reference src/sampler/fused_sampler.cu (cluster-cooperative BlockRadixSort top-K, :161-296,
// launcher :700-775), src/sampler/fused_sampler_temperature.cu and src/sampler/sampler_rng.cuh.
//
// MI355X design.  A decode step samples B <= a few hundred rows of V ~ 1.2e5 logits: 0.5 MB per row,
// nowhere near an HBM problem - it is a latency problem, so a row is spread over S >= 16 workgroups
// and the selection is EXACT but cheap:
//   phase 1 (grid S x B): each thread keeps its <= 32 elements of the segment in registers as
//     order-preserving integer keys.  The K-th largest of the 256 per-thread maxima is a lower bound
//     of the segment's K-th largest element, so only the (typically ~K..2K) elements above it are
//     compacted into an LDS list; a byte-wise radix select over 52-bit composites (key << 20 | ~index -
//     all distinct, ties resolved towards the smaller token id) picks the segment's top K.  There is
//     no sort and no atomics storm: histograms only ever see the short lists.
//   phase 2 (grid B): radix select over the S*K candidates, one wave bitonic-sorts the K winners (one
//     per lane) and runs softmax / top-p / Gumbel-max / tie-break exactly in the order of the
//     reference's PyTorch model (tests/test_sampler.py:47-165); lane 0 writes the token and ORs the
//     penalty bit.
// The fast path (temperature only) is a two-step arg-max of logits / T + Gumbel noise.
// Self-drawn noise: Philox-4x32-10 keyed by the caller's seed, counter = (token, row, launch offset).
#include "hpc_common.h"
#include "../../include/hpc_amd.h"

#include <atomic>

namespace hpc {
namespace sampler {

constexpr int kThreads = 256;
constexpr int kSegMax = 8192;   // elements of one segment (32 per thread)
constexpr int kElems = kSegMax / kThreads;
constexpr int kIdxBits = 20;    // token ids < 2^20
constexpr uint64_t kIdxMask = (1ull << kIdxBits) - 1;
constexpr int kSelBytes = 7;    // 52-bit composites
constexpr float kNegInf = -__builtin_inff();

struct Args {
  const void* logits;
  int dtype;  // 0 fp32, 1 bf16
  long row_stride;
  int B, V, S, seg;
  uint8_t* mask;
  long mask_stride;
  const int* slot;
  const float* rp_arr;
  float rp_val;
  const float* t_arr;
  float t_val;
  int policy;
  const void* topk_ptr;
  int topk_bytes, topk_val;
  const float* topp_arr;
  float topp_val;
  const float* noise;
  const int64_t* draft;
  int max_topk;
  uint64_t seed, offset;
  int* out;
  uint64_t* cand;    // [B][S][K] composites
  float* seg_stat;   // [B][S][2]: max, sum exp(x - max)   (softmax-before-topk only)
};

__device__ __forceinline__ uint32_t key_of(float x) {
  const uint32_t u = __float_as_uint(x);
  return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
__device__ __forceinline__ float val_of(uint32_t k) {
  return __uint_as_float((k & 0x80000000u) ? (k & 0x7fffffffu) : ~k);
}
__device__ __forceinline__ uint64_t comp_of(uint32_t key, uint32_t idx) {
  return (static_cast<uint64_t>(key) << kIdxBits) | (kIdxMask - idx);
}

// ---- Philox-4x32-10 ---------------------------------------------------------------------------------
__device__ __forceinline__ uint32_t philox_word0(uint64_t seed, uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3) {
  uint32_t k0 = static_cast<uint32_t>(seed), k1 = static_cast<uint32_t>(seed >> 32);
#pragma unroll
  for (int r = 0; r < 10; ++r) {
    const uint64_t p0 = static_cast<uint64_t>(0xD2511F53u) * c0, p1 = static_cast<uint64_t>(0xCD9E8D57u) * c2;
    const uint32_t n0 = static_cast<uint32_t>(p1 >> 32) ^ c1 ^ k0, n2 = static_cast<uint32_t>(p0 >> 32) ^ c3 ^ k1;
    c1 = static_cast<uint32_t>(p1);
    c3 = static_cast<uint32_t>(p0);
    c0 = n0;
    c2 = n2;
    k0 += 0x9E3779B9u;
    k1 += 0xBB67AE85u;
  }
  return c0;
}
// Gumbel(0) noise for (row, token): g = -log(max(-log(U), 1e-20)), U in (0, 1]  (reference sampler_rng.cuh:34-38)
__device__ __forceinline__ float own_gumbel(const Args& a, int b, int idx) {
  const uint32_t w = philox_word0(a.seed, static_cast<uint32_t>(idx), static_cast<uint32_t>(b),
                                  static_cast<uint32_t>(a.offset), static_cast<uint32_t>(a.offset >> 32));
  const float u = (static_cast<float>(w >> 8) + 1.0f) * (1.0f / 16777216.0f);  // (0, 1]
  return -logf(fmaxf(-logf(u), 1e-20f));
}

__device__ __forceinline__ float load_logit(const Args& a, int b, int i) {
  if (a.dtype == 0) return static_cast<const float*>(a.logits)[static_cast<long>(b) * a.row_stride + i];
  const uint16_t h = static_cast<const uint16_t*>(a.logits)[static_cast<long>(b) * a.row_stride + i];
  return __uint_as_float(static_cast<uint32_t>(h) << 16);
}

// ---- byte-wise radix select: the K-th largest of n distinct composites in `list` (LDS) ---------------
// All 256 threads call it; n > K.  hist: 256 counters, misc: ints 0, 1 and 4 are used here.
__device__ uint64_t select_kth(const uint64_t* list, int n, int K, unsigned* hist, int* misc) {
  const int tid = threadIdx.x, lane = tid & 63;
  uint64_t prefix = 0, mask = 0;
  int need = K;
  for (int p = kSelBytes - 1; p >= 0; --p) {
    hist[tid] = 0;
    __syncthreads();
    for (int i = tid; i < n; i += kThreads) {
      const uint64_t c = list[i];
      if ((c & mask) == prefix) atomicAdd(&hist[(c >> (8 * p)) & 255], 1u);
    }
    __syncthreads();
    if (tid < 64) {  // digit d with count(> d) < need <= count(>= d)
      unsigned h[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) h[j] = hist[4 * lane + j];
      const unsigned mine = h[0] + h[1] + h[2] + h[3];
      unsigned incl = mine;  // sum over lanes >= lane
#pragma unroll
      for (int o = 1; o < 64; o <<= 1) {
        const unsigned v = __shfl_down(incl, o, 64);
        if (lane + o < 64) incl += v;
      }
      unsigned gt = incl - mine;  // elements in higher lanes
#pragma unroll
      for (int j = 3; j >= 0; --j) {
        if (gt < static_cast<unsigned>(need) && static_cast<unsigned>(need) <= gt + h[j]) {
          misc[0] = 4 * lane + j;
          misc[1] = need - static_cast<int>(gt);
          misc[4] = static_cast<unsigned>(need) == gt + h[j];  // the whole bin is wanted: no need to refine it
        }
        gt += h[j];
      }
    }
    __syncthreads();
    prefix |= static_cast<uint64_t>(misc[0]) << (8 * p);
    mask |= 255ull << (8 * p);
    need = misc[1];
    // early exit (the usual case after the 3-4 bytes that carry the float value): every element of the
    // chosen bin belongs to the top K, so the bin's lower bound is already the threshold
    if (misc[4]) break;
  }
  return prefix;
}

// ---- phase 1: per-segment top-K candidates ------------------------------------------------------------
__global__ __launch_bounds__(kThreads) void sampler_segment_kernel(const Args a) {
  __shared__ uint64_t s_list[kSegMax];
  __shared__ unsigned s_hist[256];
  __shared__ int s_misc[6];
  __shared__ float s_red[8];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int s = blockIdx.x, b = blockIdx.y;
  const int K = a.max_topk;
  const int i0 = s * a.seg, i1 = min(a.V, i0 + a.seg);
  const int n = max(i1 - i0, 0);

  float rp = a.rp_arr ? a.rp_arr[b] : a.rp_val;
  const bool has_rp = a.mask && rp > 0.f;
  const float inv_rp = has_rp ? static_cast<float>(1.0 / static_cast<double>(rp)) : 1.f;
  const float temp = a.t_arr ? a.t_arr[b] : a.t_val;
  const uint8_t* mrow = a.mask ? a.mask + static_cast<long>(a.slot[b]) * a.mask_stride : nullptr;

  uint32_t key[kElems];
  uint32_t best = 0;
#pragma unroll
  for (int j = 0; j < kElems; ++j) {
    const int i = i0 + j * kThreads + tid;
    key[j] = 0;
    if (i < i1) {
      float x = load_logit(a, b, i);
      if (has_rp && ((mrow[i >> 3] >> (i & 7)) & 1)) x = x > 0.f ? x * inv_rp : x * rp;
      if (temp > 0.f) x = x / temp;
      key[j] = key_of(x);
      best = max(best, key[j]);
    }
  }

  // softmax-before-topk statistics of the segment (max, sum exp(x - max))
  if (a.policy == 1) {
    float m = best ? val_of(best) : kNegInf;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o, 64));
    if (lane == 0) s_red[wave] = m;
    __syncthreads();
    m = fmaxf(fmaxf(s_red[0], s_red[1]), fmaxf(s_red[2], s_red[3]));
    // a segment whose logits are all -inf (grammar / vocabulary masks) must contribute (max = -inf,
    // sum = 0), not exp(-inf - (-inf)) = NaN: subtract 0 instead of the -inf maximum
    const float m_sub = m == kNegInf ? 0.f : m;
    float e = 0.f;
#pragma unroll
    for (int j = 0; j < kElems; ++j)
      if (key[j]) e += expf(val_of(key[j]) - m_sub);
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) e += __shfl_xor(e, o, 64);
    if (lane == 0) s_red[4 + wave] = e;
    __syncthreads();
    if (tid == 0) {
      float* st = a.seg_stat + (static_cast<long>(b) * a.S + s) * 2;
      st[0] = m;
      st[1] = (s_red[4] + s_red[5]) + (s_red[6] + s_red[7]);
    }
  }

  // lower bound of the K-th largest element: the K-th largest per-thread maximum
  uint32_t bound = 0;
  if (n > K) {
    s_list[tid] = comp_of(best, tid);
    __syncthreads();
    bound = static_cast<uint32_t>(select_kth(s_list, kThreads, K, s_hist, s_misc) >> kIdxBits);
  }
  if (tid == 0) s_misc[2] = 0;
  __syncthreads();
#pragma unroll
  for (int j = 0; j < kElems; ++j) {
    if (key[j] && key[j] >= bound) {
      const int pos = atomicAdd(&s_misc[2], 1);
      s_list[pos] = comp_of(key[j], i0 + j * kThreads + tid);
    }
  }
  __syncthreads();
  const int cnt = s_misc[2];
  uint64_t thr = 0;
  if (cnt > K) thr = select_kth(s_list, cnt, K, s_hist, s_misc);
  if (tid == 0) s_misc[3] = 0;
  __syncthreads();
  uint64_t* out = a.cand + (static_cast<long>(b) * a.S + s) * K;
  for (int i = tid; i < cnt; i += kThreads) {
    const uint64_t c = s_list[i];
    if (c >= thr) out[atomicAdd(&s_misc[3], 1)] = c;
  }
  __syncthreads();
  // fewer than K elements: pad with distinct composites below every real one (key part 0)
  for (int i = min(cnt, K) + tid; i < K; i += kThreads) out[i] = static_cast<uint64_t>(s * K + i);
}

// ---- phase 2: merge, sort the K winners, sample ---------------------------------------------------------
__global__ __launch_bounds__(kThreads) void sampler_final_kernel(const Args a) {
  __shared__ uint64_t s_list[kSegMax];
  __shared__ uint64_t s_sel[64];
  __shared__ unsigned s_hist[256];
  __shared__ int s_misc[6];
  const int tid = threadIdx.x, lane = tid & 63;
  const int b = blockIdx.x;
  const int K = a.max_topk;
  const int n = a.S * K;
  const uint64_t* cand = a.cand + static_cast<long>(b) * n;
  for (int i = tid; i < n; i += kThreads) s_list[i] = cand[i];
  if (tid < 64) s_sel[tid] = 0;
  if (tid == 0) s_misc[2] = 0;
  __syncthreads();
  uint64_t thr = 0;
  if (n > K) thr = select_kth(s_list, n, K, s_hist, s_misc);
  for (int i = tid; i < n; i += kThreads) {
    const uint64_t c = s_list[i];
    if (c >= thr) s_sel[atomicAdd(&s_misc[2], 1)] = c;
  }
  __syncthreads();
  if (tid >= 64) return;

  // ---- one wave: bitonic sort (descending) of one composite per lane --------------------------------------
  uint64_t c = s_sel[lane];
#pragma unroll
  for (int k = 2; k <= 64; k <<= 1)
#pragma unroll
    for (int j = k >> 1; j > 0; j >>= 1) {
      const uint32_t lo = __shfl_xor(static_cast<uint32_t>(c), j, 64);
      const uint32_t hi = __shfl_xor(static_cast<uint32_t>(c >> 32), j, 64);
      const uint64_t o = (static_cast<uint64_t>(hi) << 32) | lo;
      const bool up = (lane & k) != 0;       // this block sorts ascending
      const bool lower = (lane & j) == 0;    // lane holds the lower position of the pair
      const bool take_max = (lower != up);   // descending blocks keep the max in the lower lane
      c = take_max ? (c > o ? c : o) : (c < o ? c : o);
    }
  const uint32_t key = static_cast<uint32_t>(c >> kIdxBits);
  const int tok = static_cast<int>(kIdxMask - (c & kIdxMask));
  const bool real = key != 0;
  const float x = real ? val_of(key) : kNegInf;

  int kb = a.topk_ptr ? (a.topk_bytes == 8 ? static_cast<int>(static_cast<const int64_t*>(a.topk_ptr)[b])
                                           : static_cast<const int*>(a.topk_ptr)[b])
                      : a.topk_val;
  if (kb <= 0 || kb > K) kb = K;
  const bool in_k = lane < kb && real;
  const float tp = a.topp_arr ? a.topp_arr[b] : a.topp_val;

  float p = 0.f, g = x;
  if (a.policy == 2) {  // softmax over the surviving top-k
    const float m = __shfl(x, 0, 64);
    float e = in_k ? expf(x - m) : 0.f;
    float sum = e;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) sum += __shfl_xor(sum, o, 64);
    p = e / sum;
    g = logf(p);
  } else if (a.policy == 1) {  // probabilities of the full-vocabulary softmax
    float gm = kNegInf;
    for (int s = 0; s < a.S; ++s) gm = fmaxf(gm, a.seg_stat[(static_cast<long>(b) * a.S + s) * 2]);
    float gs = 0.f;
    for (int s = 0; s < a.S; ++s) {
      const float* st = a.seg_stat + (static_cast<long>(b) * a.S + s) * 2;
      gs += st[1] * expf(st[0] - gm);
    }
    p = in_k ? expf(x - gm) / gs : 0.f;
    g = p > 0.f ? logf(p) : kNegInf;
  }
  bool keep = in_k;
  if (tp > 0.f) {
    // cumulative probability in rank order, summed serially like the PyTorch model (cumsum - p)
    float cum = 0.f, mine_excl = 0.f;
    for (int i = 0; i < kb; ++i) {
      const float pi = __shfl(p, i, 64);
      cum += pi;
      if (lane == i) mine_excl = cum - pi;
    }
    keep = in_k && (lane == 0 || mine_excl < tp);
  }
  const float noise = in_k ? (a.noise ? a.noise[static_cast<long>(b) * a.V + tok] : own_gumbel(a, b, tok)) : 0.f;
  float score = keep ? g + noise : kNegInf;
  float best = score;
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) best = fmaxf(best, __shfl_xor(best, o, 64));
  // ties (and the all -inf row) resolve towards the smaller token id among the top-k entries
  int cand_tok = (in_k && score == best) ? tok : 0x7fffffff;
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) cand_tok = min(cand_tok, __shfl_xor(cand_tok, o, 64));
  if (cand_tok == 0x7fffffff) cand_tok = __shfl(tok, 0, 64);
  if (lane == 0) {
    a.out[b] = cand_tok;
    if (a.mask) {
      uint8_t* row = a.mask + static_cast<long>(a.slot[b]) * a.mask_stride;
      const uintptr_t addr = reinterpret_cast<uintptr_t>(row + (cand_tok >> 3));
      atomicOr(reinterpret_cast<unsigned*>(addr & ~uintptr_t(3)), (1u << (cand_tok & 7)) << (8 * (addr & 3)));
    }
  }
}

// ---- temperature-only fast path: argmax(logits / T + gumbel) ---------------------------------------------
// composite = key(score) << 32 | ~index  -> max picks the highest score, first index on ties
__global__ __launch_bounds__(kThreads) void temperature_segment_kernel(const Args a) {
  __shared__ uint64_t s_red[4];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int s = blockIdx.x, b = blockIdx.y;
  const int i0 = s * a.seg, i1 = min(a.V, i0 + a.seg);
  const float temp = a.t_arr ? a.t_arr[b] : a.t_val;
  const int64_t dr = a.draft ? a.draft[b] : -1;
  uint64_t best = 0;
  for (int i = i0 + tid; i < i1; i += kThreads) {
    float x = load_logit(a, b, i) / temp;
    if (static_cast<int64_t>(i) == dr) x = kNegInf;
    x += a.noise ? a.noise[static_cast<long>(b) * a.V + i] : own_gumbel(a, b, i);
    const uint64_t c = (static_cast<uint64_t>(key_of(x)) << 32) | (0xffffffffu - static_cast<uint32_t>(i));
    best = c > best ? c : best;
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    const uint32_t lo = __shfl_xor(static_cast<uint32_t>(best), o, 64);
    const uint32_t hi = __shfl_xor(static_cast<uint32_t>(best >> 32), o, 64);
    const uint64_t v = (static_cast<uint64_t>(hi) << 32) | lo;
    best = v > best ? v : best;
  }
  if (lane == 0) s_red[wave] = best;
  __syncthreads();
  if (tid == 0) {
    uint64_t m = s_red[0];
    for (int w = 1; w < 4; ++w) m = s_red[w] > m ? s_red[w] : m;
    a.cand[static_cast<long>(b) * a.S + s] = m;
  }
}

__global__ __launch_bounds__(64) void temperature_final_kernel(const Args a) {
  const int b = blockIdx.x, lane = threadIdx.x;
  uint64_t best = 0;
  for (int s = lane; s < a.S; s += 64) {
    const uint64_t c = a.cand[static_cast<long>(b) * a.S + s];
    best = c > best ? c : best;
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    const uint32_t lo = __shfl_xor(static_cast<uint32_t>(best), o, 64);
    const uint32_t hi = __shfl_xor(static_cast<uint32_t>(best >> 32), o, 64);
    const uint64_t v = (static_cast<uint64_t>(hi) << 32) | lo;
    best = v > best ? v : best;
  }
  if (lane == 0) a.out[b] = static_cast<int>(0xffffffffu - static_cast<uint32_t>(best));
}

// launch counter: the same seed gives fresh noise on every launch (reference sampler_rng.cuh:41-52)
std::atomic<uint64_t> g_launch_offset{0};
inline uint64_t next_offset() { return g_launch_offset.fetch_add(128, std::memory_order_relaxed); }

inline int segments_of(int V) {
  const int s = (V + kSegMax - 1) / kSegMax;
  return s < 16 ? 16 : s;
}

}  // namespace sampler
}  // namespace hpc

extern "C" int hpc_sampler_segments(int vocab_size) { return hpc::sampler::segments_of(vocab_size); }

// workspace: candidates [B][S][max_topk] u64 + segment softmax statistics [B][S][2] f32
extern "C" int64_t hpc_fused_sampler_workspace_bytes(int batch_size, int vocab_size, int max_topk) {
  const int64_t S = hpc::sampler::segments_of(vocab_size);
  return static_cast<int64_t>(batch_size) * S * max_topk * 8 + static_cast<int64_t>(batch_size) * S * 8;
}

// reference: fused_sampler_async (src/sampler/sampler.h:17-31, src/sampler/fused_sampler.cu:700-775)
extern "C" int hpc_fused_sampler_async(
    void* token_ids, void* workspace, const void* logits, int logits_dtype, void* penalty_mask,
    int64_t penalty_mask_row_bytes, const void* slot_id, const void* rp_arr, float rp_val, const void* temp_arr,
    float temp_val, int softmax_policy, const void* topk_ptr, int topk_int_bytes, int topk_val,
    const void* topp_arr, float topp_val, const void* gumbel_noise, int batch_size, int vocab_size,
    int64_t logits_row_stride, int max_topk, uint64_t rng_seed, hipStream_t stream) {
  using namespace hpc::sampler;
  if (!token_ids || !workspace || !logits) return HPC_ERR_INVALID;
  if (batch_size < 0 || vocab_size <= 0 || (max_topk != 32 && max_topk != 64)) return HPC_ERR_INVALID;
  if (batch_size == 0) return HPC_OK;
  if ((vocab_size & 7) || vocab_size >= (1 << kIdxBits)) return HPC_ERR_UNSUPPORTED;
  if ((penalty_mask == nullptr) != (slot_id == nullptr)) return HPC_ERR_INVALID;
  if (softmax_policy < 0 || softmax_policy > 2 || (logits_dtype != 0 && logits_dtype != 1)) return HPC_ERR_INVALID;
  if (batch_size > 65535) return HPC_ERR_UNSUPPORTED;
  Args a{};
  a.logits = logits;
  a.dtype = logits_dtype;
  a.row_stride = logits_row_stride;
  a.B = batch_size;
  a.V = vocab_size;
  a.S = segments_of(vocab_size);
  a.seg = ((vocab_size + a.S - 1) / a.S + kThreads - 1) / kThreads * kThreads;
  a.mask = static_cast<uint8_t*>(penalty_mask);
  a.mask_stride = penalty_mask_row_bytes;
  a.slot = static_cast<const int*>(slot_id);
  a.rp_arr = static_cast<const float*>(rp_arr);
  a.rp_val = rp_val;
  a.t_arr = static_cast<const float*>(temp_arr);
  a.t_val = temp_val;
  a.policy = softmax_policy;
  a.topk_ptr = topk_ptr;
  a.topk_bytes = topk_int_bytes;
  a.topk_val = topk_val;
  a.topp_arr = static_cast<const float*>(topp_arr);
  a.topp_val = topp_val;
  a.noise = static_cast<const float*>(gumbel_noise);
  a.draft = nullptr;
  a.max_topk = max_topk;
  a.seed = rng_seed;
  a.offset = gumbel_noise ? 0 : next_offset();
  a.out = static_cast<int*>(token_ids);
  a.cand = static_cast<uint64_t*>(workspace);
  a.seg_stat = reinterpret_cast<float*>(a.cand + static_cast<int64_t>(batch_size) * a.S * max_topk);
  sampler_segment_kernel<<<dim3(a.S, batch_size), kThreads, 0, stream>>>(a);
  HPC_CHECK_LAUNCH();
  sampler_final_kernel<<<batch_size, kThreads, 0, stream>>>(a);
  HPC_CHECK_LAUNCH();
  return HPC_OK;
}

// reference: fused_sampler_temperature_async (src/sampler/sampler.h:33-45,
// src/sampler/fused_sampler_temperature.cu:480-574).  workspace: batch_size * segments u64.
extern "C" int hpc_fused_sampler_temperature_async(void* token_ids, void* workspace, const void* logits,
                                                   int logits_dtype, int64_t logits_row_stride,
                                                   const void* temp_arr, float temp_val,
                                                   const void* gumbel_noise, const void* draft_token_ids,
                                                   int batch_size, int vocab_size, uint64_t rng_seed,
                                                   hipStream_t stream) {
  using namespace hpc::sampler;
  if (!token_ids || !workspace || !logits) return HPC_ERR_INVALID;
  if (batch_size < 0 || vocab_size <= 0 || (logits_dtype != 0 && logits_dtype != 1)) return HPC_ERR_INVALID;
  if (batch_size == 0) return HPC_OK;
  if ((vocab_size & 7) || vocab_size >= (1 << kIdxBits)) return HPC_ERR_UNSUPPORTED;
  if (batch_size > 65535) return HPC_ERR_UNSUPPORTED;
  Args a{};
  a.logits = logits;
  a.dtype = logits_dtype;
  a.row_stride = logits_row_stride;
  a.B = batch_size;
  a.V = vocab_size;
  a.S = segments_of(vocab_size);
  a.seg = (vocab_size + a.S - 1) / a.S;
  a.t_arr = static_cast<const float*>(temp_arr);
  a.t_val = temp_val;
  a.noise = static_cast<const float*>(gumbel_noise);
  a.draft = static_cast<const int64_t*>(draft_token_ids);
  a.seed = rng_seed;
  a.offset = gumbel_noise ? 0 : next_offset();
  a.out = static_cast<int*>(token_ids);
  a.cand = static_cast<uint64_t*>(workspace);
  temperature_segment_kernel<<<dim3(a.S, batch_size), kThreads, 0, stream>>>(a);
  HPC_CHECK_LAUNCH();
  temperature_final_kernel<<<batch_size, 64, 0, stream>>>(a);
  HPC_CHECK_LAUNCH();
  return HPC_OK;
}
