// RoPE (neox) + optional QK RMSNorm + paged KV-cache write, bf16 and fp8 outputs - gfx950.
//
// Replaces reference src/rope/rope.cu (rope_norm_store_kv_kernel :99, rope_norm_store_kv_fp8_kernel
// :420-786, launchers :790-950).  This is the producer of the decode-attention inputs: it defines
// the KV-cache "wire format" those kernels read.
//
// MI355X design: pure HBM-bound elementwise work, so no LDS staging of whole rows (the reference
// stages the row in shared memory for its 32-lane warps): one 64-lane wave handles EIGHT heads of a
// token, lane j of an 8-lane group owns elements 8j..8j+7 of both halves of a 128-wide head (the
// neox partners) - two 16-byte loads / stores per lane, RMSNorm and the dynamic-scale abs-max are
// 8-lane shuffle reductions.  V heads ride in the same grid (copy / static quant).  One extra
// workgroup per request zeroes the tail of its last page (so stale tokens can never be attended
// to) and its split_k_flag row, like the reference's "clear blocks" (rope.cu:462-503).
#include "hpc_common.h"
#include "../../include/hpc_amd.h"

namespace hpc {
namespace rope {

struct Args {
  const uint16_t* qkv;       // [rows, (Hq + 2*Hkv) * 128] bf16
  const float* cos_sin;      // [max_pos, 128]: cos[0..63], sin[0..63]
  const int* seqlen;         // [num_req] total length incl. the new tokens
  const int* q_index;        // [num_req + 1]
  const int* kv_indices;     // [num_req, max_blocks]
  const float* q_norm_w;     // [128] or null
  const float* k_norm_w;
  void* out_q;               // [rows, Hq, 128] bf16 / e4m3
  void* kcache;              // [blocks, P, Hkv, 128]
  void* vcache;
  void* out_k;               // optional bypass [rows, Hkv, 128]
  void* out_v;
  int* split_k_flag;         // fp8: [num_req, Hkv]
  float* q_scale;            // fp8 dynamic: decode [rows, Hq]; prefill [num_req, Hq, max_seqlen_pad]
  const float* k_scale;      // fp8: [1]
  const float* v_scale;
  const float* q_scale_inv;  // fp8 static: [1]
  float upper_max;
  int num_req, num_rows, num_q_heads, num_kv_heads, block_size, max_blocks, max_seqlen_pad;
  long k_block_stride, v_block_stride;  // elements
  int norm_policy, quant_policy, is_prefill;
};

constexpr int kThreads = 256;
constexpr int kD = 128;
constexpr int kHeadsPerWave = 8;  // 8 lanes per head, 16 B of each half per lane

__device__ __forceinline__ float oct_sum(float v) {  // over the 8 lanes of a head
#pragma unroll
  for (int o = 4; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ __forceinline__ float oct_max(float v) {
#pragma unroll
  for (int o = 4; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
  return v;
}
__device__ __forceinline__ void unpack8(const u32x4 r, float* x) {
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    x[2 * i] = bf16lo_to_f32(r[i]);
    x[2 * i + 1] = bf16hi_to_f32(r[i]);
  }
}
__device__ __forceinline__ void ld8f(const float* p, float* x) {
  const f32x4 lo = *reinterpret_cast<const f32x4*>(p), hi = *reinterpret_cast<const f32x4*>(p + 4);
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    x[i] = lo[i];
    x[4 + i] = hi[i];
  }
}

template <bool kFp8, int kU>  // kU groups of 8 heads per wave (same token): amortises the position lookup
__global__ __launch_bounds__(kThreads) void rope_kernel(const Args a) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int hsub = lane >> 3, j = lane & 7;
  const int heads_total = a.num_q_heads + 2 * a.num_kv_heads;
  const int units_per_row = (heads_total + kHeadsPerWave * kU - 1) / (kHeadsPerWave * kU);
  constexpr int kEB = kFp8 ? 1 : 2;

  // ---- clear blocks: one workgroup per request ---------------------------------------------------
  const int num_work = a.num_rows * units_per_row;
  const int compute_blocks = (num_work + 3) / 4;
  if (static_cast<int>(blockIdx.x) >= compute_blocks) {
    const int req = blockIdx.x - compute_blocks;
    if (req >= a.num_req) return;
    if (kFp8 && a.split_k_flag)
      for (int h = threadIdx.x; h < a.num_kv_heads; h += kThreads) a.split_k_flag[req * a.num_kv_heads + h] = 0;
    const int last = a.seqlen[req] - 1;
    if (last < 0 || a.out_k || a.out_v) return;
    const int pb = last % a.block_size;
    const int phys = a.kv_indices[static_cast<long>(req) * a.max_blocks + last / a.block_size];
    const long row_bytes = static_cast<long>(a.num_kv_heads) * kD * kEB;
    const long from = (pb + 1) * row_bytes, to = a.block_size * row_bytes;
    uint8_t* kb = static_cast<uint8_t*>(a.kcache) + phys * a.k_block_stride * kEB;
    uint8_t* vb = static_cast<uint8_t*>(a.vcache) + phys * a.v_block_stride * kEB;
    const u32x4 z = u32x4{0u, 0u, 0u, 0u};
    for (long o = from + threadIdx.x * 16; o < to; o += kThreads * 16) {
      st16(kb + o, z);
      st16(vb + o, z);
    }
    return;
  }

  // ---- compute: wave -> (row, kU x 8 consecutive heads) -------------------------------------------
  const int work = blockIdx.x * 4 + wave;
  if (work >= num_work) return;
  const int row = work / units_per_row;
  const int head0 = (work % units_per_row) * kHeadsPerWave * kU + hsub;
  // the token's heads do not depend on its position: get them in flight before the lookup
  u32x4 raw1[kU], raw2[kU];
#pragma unroll
  for (int u = 0; u < kU; ++u) {
    const int head = head0 + u * kHeadsPerWave;
    const uint16_t* src = a.qkv + static_cast<long>(row) * heads_total * kD + (head < heads_total ? head : 0) * kD;
    raw1[u] = ld16(src + 8 * j);
    raw2[u] = ld16(src + 64 + 8 * j);
  }
  const cint_ptr qidx = as_const(a.q_index);
  if (row >= qidx[a.num_req]) return;  // rows past the last request
  // request of this row: last b with q_index[b] <= row (wave-uniform binary search)
  int lo = 0, hi = a.num_req;
  while (lo < hi) {
    const int mid = (lo + hi) >> 1;
    if (qidx[mid + 1] <= row) lo = mid + 1; else hi = mid;
  }
  const int req = lo;
  const int sl = as_const(a.seqlen)[req];
  const int pos = sl - (qidx[req + 1] - row);  // absolute position of this token
  if (pos < 0) return;  // align-8 padding requests (length 0) own the padding rows: nothing to do
  float c[8], sn[8];
  ld8f(a.cos_sin + static_cast<long>(pos) * kD + 8 * j, c);
  ld8f(a.cos_sin + static_cast<long>(pos) * kD + 64 + 8 * j, sn);
  const int phys = (a.out_k && a.out_v) ? 0 : as_const(a.kv_indices)[static_cast<long>(req) * a.max_blocks + pos / a.block_size];
  const long tok_off = static_cast<long>(pos % a.block_size) * a.num_kv_heads * kD;

#pragma unroll
  for (int u = 0; u < kU; ++u) {
    const int head = head0 + u * kHeadsPerWave;
    if (head >= heads_total) continue;  // ragged last group (the 8-lane groups are independent)
    float x1[8], x2[8];
    unpack8(raw1[u], x1);
    unpack8(raw2[u], x2);
    const bool is_q = head < a.num_q_heads;
    const bool is_k = !is_q && head < a.num_q_heads + a.num_kv_heads;
    const int kvh = is_q ? 0 : (is_k ? head - a.num_q_heads : head - a.num_q_heads - a.num_kv_heads);

    if (is_q || is_k) {
      const float* nw = is_q ? a.q_norm_w : a.k_norm_w;
      auto rms = [&]() {
        float ss = 0.f;
#pragma unroll
        for (int i = 0; i < 8; ++i) ss += x1[i] * x1[i] + x2[i] * x2[i];
        ss = oct_sum(ss);
        const float r = rsqrtf(ss * (1.0f / kD) + 1e-6f);
        float w1[8], w2[8];
        ld8f(nw + 8 * j, w1);
        ld8f(nw + 64 + 8 * j, w2);
#pragma unroll
        for (int i = 0; i < 8; ++i) {
          x1[i] *= r * w1[i];
          x2[i] *= r * w2[i];
        }
      };
      if (a.norm_policy == 2) rms();
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        const float p = x1[i] * c[i] - x2[i] * sn[i], q = x2[i] * c[i] + x1[i] * sn[i];
        x1[i] = p;
        x2[i] = q;
      }
      if (a.norm_policy == 1) rms();
    }

    // ---- scale (fp8) and destination ------------------------------------------------------------------
    float mult = 1.0f;
    uint8_t* dst;
    if (is_q) {
      if constexpr (kFp8) {
        if (a.quant_policy == 1) {
          float am = 0.f;
#pragma unroll
          for (int i = 0; i < 8; ++i) am = fmaxf(am, fmaxf(fabsf(x1[i]), fabsf(x2[i])));
          const float sc = oct_max(am) / a.upper_max;
          if (j == 0) {
            if (a.is_prefill)
              a.q_scale[(static_cast<long>(req) * a.num_q_heads + head) * a.max_seqlen_pad + (row - qidx[req])] = sc;
            else
              a.q_scale[static_cast<long>(row) * a.num_q_heads + head] = sc;
          }
          mult = 1.0f / sc;
        } else {
          mult = a.q_scale_inv[0];
        }
      }
      dst = static_cast<uint8_t*>(a.out_q) + (static_cast<long>(row) * a.num_q_heads + head) * kD * kEB;
    } else {
      if constexpr (kFp8) mult = 1.0f / (is_k ? a.k_scale[0] : a.v_scale[0]);
      void* bypass = is_k ? a.out_k : a.out_v;
      if (bypass) {
        dst = static_cast<uint8_t*>(bypass) + (static_cast<long>(row) * a.num_kv_heads + kvh) * kD * kEB;
      } else {
        const long off = phys * (is_k ? a.k_block_stride : a.v_block_stride) + tok_off + kvh * kD;
        dst = static_cast<uint8_t*>(is_k ? a.kcache : a.vcache) + off * kEB;
      }
    }
    if constexpr (kFp8) {
      u32x2 o1, o2;
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        o1[i] = quant_4xe4m3(x1[4 * i] * mult, x1[4 * i + 1] * mult, x1[4 * i + 2] * mult, x1[4 * i + 3] * mult);
        o2[i] = quant_4xe4m3(x2[4 * i] * mult, x2[4 * i + 1] * mult, x2[4 * i + 2] * mult, x2[4 * i + 3] * mult);
      }
      *reinterpret_cast<u32x2*>(dst + 8 * j) = o1;
      *reinterpret_cast<u32x2*>(dst + 64 + 8 * j) = o2;
    } else {
      u32x4 o1, o2;
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        o1[i] = pack_bf16x2(x1[2 * i], x1[2 * i + 1]);
        o2[i] = pack_bf16x2(x2[2 * i], x2[2 * i + 1]);
      }
      st16(dst + 16 * j, o1);
      st16(dst + 128 + 16 * j, o2);
    }
  }
}

}  // namespace rope
}  // namespace hpc

namespace {
int launch_rope(bool fp8, hpc::rope::Args& a, hipStream_t stream) {
  using namespace hpc::rope;
  if (!a.qkv || !a.cos_sin || !a.seqlen || !a.q_index || !a.kv_indices || !a.out_q) return HPC_ERR_INVALID;
  if ((!a.kcache || !a.vcache) && (!a.out_k || !a.out_v)) return HPC_ERR_INVALID;
  if (a.norm_policy < 0 || a.norm_policy > 2) return HPC_ERR_INVALID;
  if (a.norm_policy && (!a.q_norm_w || !a.k_norm_w)) return HPC_ERR_INVALID;
  if (a.num_req <= 0 || a.num_rows < 0 || a.block_size <= 0) return HPC_ERR_INVALID;
  const int heads_total = a.num_q_heads + 2 * a.num_kv_heads;
  const int octs = (heads_total + kHeadsPerWave - 1) / kHeadsPerWave;
  // large token counts: 2 head groups per wave (position lookup amortised, more bytes in flight per wave)
  const bool wide = static_cast<long>(a.num_rows) * octs >= 16384 && octs >= 2;
  const int units = wide ? (octs + 1) / 2 : octs;
  const int grid = (a.num_rows * units + 3) / 4 + a.num_req;
  if (fp8) {
    if (wide) rope_kernel<true, 2><<<grid, kThreads, 0, stream>>>(a);
    else rope_kernel<true, 1><<<grid, kThreads, 0, stream>>>(a);
  } else {
    if (wide) rope_kernel<false, 2><<<grid, kThreads, 0, stream>>>(a);
    else rope_kernel<false, 1><<<grid, kThreads, 0, stream>>>(a);
  }
  HPC_CHECK_LAUNCH();
  return HPC_OK;
}
}  // namespace

extern "C" int hpc_rope_norm_store_kv_async(
    void* out_q, void* kcache, void* vcache, void* out_k, void* out_v, const void* qkv,
    const void* cos_sin, const void* num_seqlen_per_req, const void* q_index,
    const void* kvcache_indices, const void* q_norm_weight, const void* k_norm_weight,
    int64_t kcache_block_stride, int64_t vcache_block_stride, int num_req, int max_blocks_per_req,
    int kv_block_size, int num_rows, int num_q_heads, int num_kv_heads, int qk_head_dim,
    int v_head_dim, int is_prefill, int qk_norm_policy, hipStream_t stream) {
  if (qk_head_dim != 128 || v_head_dim != 128) return HPC_ERR_UNSUPPORTED;
  hpc::rope::Args a{};
  a.qkv = static_cast<const uint16_t*>(qkv);
  a.cos_sin = static_cast<const float*>(cos_sin);
  a.seqlen = static_cast<const int*>(num_seqlen_per_req);
  a.q_index = static_cast<const int*>(q_index);
  a.kv_indices = static_cast<const int*>(kvcache_indices);
  a.q_norm_w = static_cast<const float*>(q_norm_weight);
  a.k_norm_w = static_cast<const float*>(k_norm_weight);
  a.out_q = out_q;
  a.kcache = kcache;
  a.vcache = vcache;
  a.out_k = out_k;
  a.out_v = out_v;
  a.upper_max = 448.0f;
  a.num_req = num_req;
  a.num_rows = num_rows;
  a.num_q_heads = num_q_heads;
  a.num_kv_heads = num_kv_heads;
  a.block_size = kv_block_size;
  a.max_blocks = max_blocks_per_req;
  a.k_block_stride = kcache_block_stride;
  a.v_block_stride = vcache_block_stride;
  a.norm_policy = qk_norm_policy;
  a.is_prefill = is_prefill;
  return launch_rope(false, a, stream);
}

extern "C" int hpc_rope_norm_store_kv_fp8_async(
    void* out_q, void* kcache, void* vcache, void* out_k, void* out_v, void* split_k_flag,
    void* q_scale, const void* qkv, const void* cos_sin, const void* num_seqlen_per_req,
    const void* q_index, const void* kvcache_indices, const void* q_norm_weight,
    const void* k_norm_weight, const void* k_scale, const void* v_scale, const void* q_scale_inv,
    float upper_max, int max_seqlens_pad, int64_t kcache_block_stride, int64_t vcache_block_stride,
    int num_req, int max_blocks_per_req, int kv_block_size, int num_rows, int num_q_heads,
    int num_kv_heads, int qk_head_dim, int v_head_dim, int is_prefill, int qk_norm_policy,
    int quant_policy, hipStream_t stream) {
  if (qk_head_dim != 128 || v_head_dim != 128) return HPC_ERR_UNSUPPORTED;
  if (quant_policy != 1 && quant_policy != 2) return HPC_ERR_INVALID;
  if (!k_scale || !v_scale || !split_k_flag) return HPC_ERR_INVALID;
  if (quant_policy == 1 && !q_scale) return HPC_ERR_INVALID;
  if (quant_policy == 2 && !q_scale_inv) return HPC_ERR_INVALID;
  hpc::rope::Args a{};
  a.qkv = static_cast<const uint16_t*>(qkv);
  a.cos_sin = static_cast<const float*>(cos_sin);
  a.seqlen = static_cast<const int*>(num_seqlen_per_req);
  a.q_index = static_cast<const int*>(q_index);
  a.kv_indices = static_cast<const int*>(kvcache_indices);
  a.q_norm_w = static_cast<const float*>(q_norm_weight);
  a.k_norm_w = static_cast<const float*>(k_norm_weight);
  a.out_q = out_q;
  a.kcache = kcache;
  a.vcache = vcache;
  a.out_k = out_k;
  a.out_v = out_v;
  a.split_k_flag = static_cast<int*>(split_k_flag);
  a.q_scale = static_cast<float*>(q_scale);
  a.k_scale = static_cast<const float*>(k_scale);
  a.v_scale = static_cast<const float*>(v_scale);
  a.q_scale_inv = static_cast<const float*>(q_scale_inv);
  a.upper_max = upper_max;
  a.num_req = num_req;
  a.num_rows = num_rows;
  a.num_q_heads = num_q_heads;
  a.num_kv_heads = num_kv_heads;
  a.block_size = kv_block_size;
  a.max_blocks = max_blocks_per_req;
  a.max_seqlen_pad = max_seqlens_pad;
  a.k_block_stride = kcache_block_stride;
  a.v_block_stride = vcache_block_stride;
  a.norm_policy = qk_norm_policy;
  a.quant_policy = quant_policy;
  a.is_prefill = is_prefill;
  return launch_rope(true, a, stream);
}
