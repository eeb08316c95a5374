// Fused MoE (FP8, 128-block scales) pipeline pieces for gfx950: routing prep, SiLU*up + block
// quant, top-k reduce, and the 5-stage orchestrator.
//
// Replaces reference src/fuse_moe/count_and_gather_for_blockwise.cu (count / scan / gather),
// src/activation/activation.cu:282-355 (act_mul_and_blockwise_quant_fusemoe_kernel),
// src/fuse_moe/reduce.cu:17-81 and src/fuse_moe/fuse_moe.cu:62-116 (fuse_moe_blockwise_async).
//
// MI355X design notes:
//  * Routing is DETERMINISTIC: slot = arrival order in flattened (token, k) order, exactly the
//    reference test oracle's order (tests/test_fuse_moe_blockwise.py:55-66); the reference kernel
//    slots with atomics (count_and_gather_for_blockwise.cu:208-214).  One workgroup per local expert
//    scans topk_ids with wave ballots + prefix popcounts, so topk_pos is bit-exact and repeatable.
//  * The fused path is gather-free: the first GEMM reads token rows through row_index (pos ->
//    token) and x_scale rows directly - no copy of x, no transposed scale scatter.  The standalone
//    count_and_gather op still produces the reference's gathered / transposed layouts.
//  * Everything is launched back to back on one stream (hipGraph-capturable, no host sync).
#include "hpc_common.h"
#include "group_gemm.h"
#include "hpc_dev.h"
#include "../../include/hpc_amd.h"

extern "C" int hpc_group_gemm_blockwise_fp8_async(
    void* y_ptr, const void* x_ptr, const void* w_ptr, const void* seqlens_ptr,
    const void* cu_seqlens_ptr, const void* xscale_ptr, const void* wscale_ptr,
    const void* row_index_ptr, const void* col_base_ptr, int num_group, int m, int n, int k,
    int num_block_k_pad4, int tile_m, int64_t xscale_row_stride, int64_t xscale_kb_stride,
    const void* cu_tiles128_ptr, hipStream_t stream);
int hpc_group_gemm_blockwise_fp8_act(void* act_out, void* act_scale, const void* x_ptr, const void* w_ptr,
                                     const void* seqlens_ptr, const void* cu_seqlens_ptr, const void* xscale_ptr,
                                     const void* wscale_ptr, const void* row_index_ptr, int num_group, int m, int n,
                                     int k, int num_block_k_pad4, int64_t xscale_row_stride, int64_t xscale_kb_stride,
                                     const void* cu_tiles128_ptr, hipStream_t stream);

int hpc_group_gemm_pertensor_fp8_act(void* act_out, const void* x_ptr, const void* w_ptr, const void* seqlens_ptr,
                                     const void* cu_seqlens_ptr, const void* yscale_ptr, const void* row_index_ptr,
                                     const void* act_mul_scale_ptr, int use_bf16_mul, int num_group, int m, int x_rows,
                                     int n, int k, const void* cu_tiles128_ptr, hipStream_t stream);
bool hpc_ggemm_p8_selected(int num_group, int m, int n, int k, const void* cu_tiles128);

extern "C" int hpc_group_gemm_pertensor_fp8_async(void* y_ptr, const void* x_ptr, const void* w_ptr,
                                                  const void* seqlens_ptr, const void* cu_seqlens_ptr,
                                                  const void* yscale_ptr, const void* row_index_ptr,
                                                  int num_group, int m, int x_rows, int n, int k,
                                                  const void* cu_tiles128_ptr, hipStream_t stream);

namespace hpc {
namespace moe {

constexpr int kThreads = 256;

__device__ __forceinline__ int block_sum(int v, int* red) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  const int wave = threadIdx.x >> 6;
  __syncthreads();
  if ((threadIdx.x & 63) == 0) red[wave] = v;
  __syncthreads();
  return red[0] + red[1] + red[2] + red[3];
}

// Four consecutive routed ids starting at i (all inside the array).  kVec = the id array is 16-byte aligned: one
// 16-byte load; otherwise four 4-byte loads.  No branch on the data path: the loops below keep several of these in flight.
template <bool kVec>
__device__ __forceinline__ void load_ids4(const int* __restrict__ ids, int i, int (&id)[4]) {
  if constexpr (kVec) {
    const int4 v = *reinterpret_cast<const int4*>(ids + i);
    id[0] = v.x; id[1] = v.y; id[2] = v.z; id[3] = v.w;
  } else {
#pragma unroll
    for (int j = 0; j < 4; ++j) id[j] = ids[i + j];
  }
}

// A wave's share of the id array: wave w of kW owns [w * share_len(n), ...) - a multiple of 4 ids, so that 16-byte loads
// stay aligned; the n % 4 ids behind the last full group of four are looked at one by one.
template <int kW>
__device__ __forceinline__ int share_len(int n) { return (((n + kW - 1) / kW) + 3) & ~3; }

template <int kW>
__device__ __forceinline__ int block_sum_w(int v, int* red) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  const int wave = threadIdx.x >> 6;
  __syncthreads();
  if ((threadIdx.x & 63) == 0) red[wave] = v;
  __syncthreads();
  int t = 0;
#pragma unroll
  for (int w = 0; w < kW; ++w) t += red[w];
  return t;
}

// seqlens[e] = #{i : ids[i] == first_expert + e}.  One workgroup of kW waves per expert; a lane compares 4 ids per
// 16-byte load.  Round 5: one id per load and lane in a 4-wave workgroup took 32 us for the 32768 routed ids of the fused
// MoE's 4096-token case - a lone wave per SIMD issues an instruction every ~5 cycles, so the loop count per wave is
// what the kernel costs: 16 waves and 4 ids per lane cut it 16-fold.
template <bool kVec, int kW>
__global__ __launch_bounds__(kW * 64) void count_kernel(const int* __restrict__ ids, int n,
                                                        int first_expert, int* __restrict__ seqlens) {
  __shared__ int red[kW];
  const int target = first_expert + blockIdx.x;
  const int n4 = n & ~3;
  int c = 0;
#pragma unroll 2
  for (int i = threadIdx.x * 4; i < n4; i += kW * 256) {
    int id[4];
    load_ids4<kVec>(ids, i, id);
#pragma unroll
    for (int j = 0; j < 4; ++j) c += id[j] == target;
  }
  if (n4 + static_cast<int>(threadIdx.x) < n) c += ids[n4 + threadIdx.x] == target;
  c = block_sum_w<kW>(c, red);
  if (threadIdx.x == 0) seqlens[blockIdx.x] = c;
}

// Stable slotting: entry i (token i / num_topk) routed to local expert e gets position cu_seqlens[e] + #{i' < i routed to e}.
// One workgroup of kW waves per expert.  Round 5: every wave owns a contiguous share of the id array - it counts its
// share's matches first (the shares before it give its starting offset: ONE barrier), then walks the share again,
// 256 ids per step (4 per lane, one ballot per id column), with no further barrier.  The one-id-per-lane form with two
// barriers per 256 ids took 64 us at 32768 ids; like the count, the walk is paced by instruction issue of a lone wave
// per SIMD (~0.75 us per 256 ids), so large batches get 16 waves per expert.
template <bool kVec, int kW>
__global__ __launch_bounds__(kW * 64) void slot_kernel(
    const int* __restrict__ ids, int n, int num_topk, int first_expert, int num_expert, int tile_m,
    const int* __restrict__ seqlens, int* __restrict__ cu_seqlens, int* __restrict__ tiles,
    int* __restrict__ cu_tiles, int* __restrict__ topk_pos, int* __restrict__ row_index) {
  __shared__ int red[kW];
  __shared__ int wave_cnt[kW];
  const int e = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  // exclusive prefixes of the row and tile counts of the experts before this one: every workgroup writes its own
  // entries (round 5: thread 0 of workgroup 0 used to walk all experts one dependent load at a time - the longest
  // chain of the kernel), the last one also the totals
  int part = 0, part_t = 0;
  for (int j = tid; j < e; j += kW * 64) {
    const int c = seqlens[j];
    part += c;
    part_t += (c + tile_m - 1) / tile_m;
  }
  const int cu = block_sum_w<kW>(part, red);
  const int cu_t = block_sum_w<kW>(part_t, red);
  if (tid == 0) {
    const int c = seqlens[e], t = (c + tile_m - 1) / tile_m;
    cu_seqlens[e] = cu;
    tiles[e] = t;
    cu_tiles[e] = cu_t;
    if (e == num_expert - 1) {
      cu_seqlens[num_expert] = cu + c;
      cu_tiles[num_expert] = cu_t + t;
    }
  }
  const int target = first_expert + e;
  const int n4 = n & ~3;
  const int slen = share_len<kW>(n);
  const int q_begin = wave * slen < n4 ? wave * slen : n4;
  const int q_end = q_begin + slen < n4 ? q_begin + slen : n4;  // groups of four; the n % 4 ids behind n4: last wave, below
  int mine = 0;
#pragma unroll 2
  for (int i = q_begin + lane * 4; i < q_end; i += 256) {
    int id[4];
    load_ids4<kVec>(ids, i, id);
#pragma unroll
    for (int j = 0; j < 4; ++j) mine += id[j] == target;
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) mine += __shfl_xor(mine, o, 64);
  if (lane == 0) wave_cnt[wave] = mine;
  __syncthreads();
  int running = cu;
#pragma unroll
  for (int w = 0; w < kW; ++w) running += w < wave ? wave_cnt[w] : 0;

  const unsigned long long below = (1ull << lane) - 1ull;
  const int last_expert = first_expert + num_expert;
  // token (i / num_topk) and owner (i % num_expert) of a lane's first entry are divided out ONCE and then advanced by
  // additions
  const int first_i = q_begin + lane * 4;
  int tok0 = first_i / num_topk, sub0 = first_i - tok0 * num_topk;
  int own0 = first_i % num_expert;
  const int step_tok = 256 / num_topk, step_sub = 256 - step_tok * num_topk;
  const int step_own = 256 % num_expert;
  int nxt[4] = {-1, -1, -1, -1};  // the next step's ids are requested before this step's positions are stored
  if (first_i < q_end) load_ids4<kVec>(ids, first_i, nxt);
  for (int base = q_begin; base < q_end; base += 256) {
    const int i0 = base + lane * 4;
    const bool live = i0 < q_end;
    int id[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      id[j] = nxt[j];
      nxt[j] = -1;
    }
    if (i0 + 256 < q_end) load_ids4<kVec>(ids, i0 + 256, nxt);
    int pre = 0, tot = 0;  // matches in lower lanes / in these 256 ids
    bool match[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      match[j] = id[j] == target;
      const unsigned long long bal = __ballot(match[j]);
      pre += __builtin_popcountll(bal & below);
      tot += __builtin_popcountll(bal);
    }
    int pos = running + pre;
    int tok = tok0, sub = sub0, own = own0;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      if (match[j]) {
        topk_pos[i0 + j] = pos;
        row_index[pos] = tok;
        ++pos;
      }
      // non-local experts: exactly one block owns entry i
      if (live && (id[j] < first_expert || id[j] >= last_expert) && own == e) topk_pos[i0 + j] = -1;
      if (++sub == num_topk) { sub = 0; ++tok; }
      if (++own == num_expert) own = 0;
    }
    running += tot;
    tok0 += step_tok;
    sub0 += step_sub;
    if (sub0 >= num_topk) { sub0 -= num_topk; ++tok0; }
    own0 += step_own;
    if (own0 >= num_expert) own0 -= num_expert;
  }
  // the n % 4 ids behind the last group of four: after every group in index order, so the last wave takes them
  // (its `running` has arrived at the count of everything before them)
  if (wave == kW - 1 && n4 < n) {
    const int i = n4 + lane;
    const int id = i < n ? ids[i] : -1;
    const bool match = id == target;
    const unsigned long long bal = __ballot(match);
    if (match) {
      const int pos = running + __builtin_popcountll(bal & below);
      topk_pos[i] = pos;
      row_index[pos] = i / num_topk;
    }
    if (i < n && (id < first_expert || id >= last_expert) && i % num_expert == e) topk_pos[i] = -1;
  }
}

// tiles[g] = ceil(seqlens[g] / tile_m), cu_tiles = exclusive scan (standalone group-GEMM entry;
// reference src/group_gemm/group_gemm_blockwise_fp8.cu:19-86 computes the same inside its kernel)
__global__ void tiles_kernel(const int* __restrict__ seqlens, int num_group, int tile_m,
                             int* __restrict__ tiles, int* __restrict__ cu_tiles) {
  if (threadIdx.x != 0 || blockIdx.x != 0) return;
  int ct = 0;
  for (int j = 0; j < num_group; ++j) {
    const int t = (seqlens[j] + tile_m - 1) / tile_m;
    tiles[j] = t;
    cu_tiles[j] = ct;
    ct += t;
  }
  cu_tiles[num_group] = ct;
}

// Copies routed rows into the expert-contiguous buffer and scatters x_scale into the reference's
// transposed, tile-padded layout xs_t[kb][cu_tiles[e]*tile_m + slot]
// (reference blockwise_gather_kernel, count_and_gather_for_blockwise.cu:181-376).
__global__ __launch_bounds__(kThreads) void gather_kernel(
    const uint8_t* __restrict__ x, const float* __restrict__ x_scale, const int* __restrict__ ids,
    const int* __restrict__ topk_pos, const int* __restrict__ cu_seqlens,
    const int* __restrict__ cu_tiles, int n, int num_topk, int first_expert, int hidden, int tile_m,
    int m_pad, uint8_t* __restrict__ xg, float* __restrict__ xs_t) {
  const int i = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (i >= n) return;
  const int pos = topk_pos[i];
  if (pos < 0) return;
  const int lane = threadIdx.x & 63;
  const int tok = i / num_topk, e = ids[i] - first_expert;
  const uint8_t* src = x + static_cast<long>(tok) * hidden;
  uint8_t* dst = xg + static_cast<long>(pos) * hidden;
  for (int c = lane * 16; c < hidden; c += 64 * 16) st16(dst + c, ld16(src + c));
  const int col = cu_tiles[e] * tile_m + (pos - cu_seqlens[e]);
  const int nkb = hidden >> 7;
  for (int kb = lane; kb < nkb; kb += 64)
    xs_t[static_cast<long>(kb) * m_pad + col] = x_scale[static_cast<long>(tok) * nkb + kb];
}

// a = silu(gate) * up in fp32 on the bf16 GEMM output, per 128 columns: scale = amax/448,
// q = e4m3(a / (scale + 1e-8)).  16 lanes x 8 columns = one quant block.
// out_scale[row * os_row_stride + jb * os_blk_stride].
__global__ __launch_bounds__(kThreads) void act_mul_blockwise_quant_kernel(
    const uint16_t* __restrict__ gate_up, const int* __restrict__ num_rows_ptr, int max_rows,
    int inter, uint8_t* __restrict__ out, float* __restrict__ out_scale, long os_row_stride,
    long os_blk_stride, const int* __restrict__ row_to_col, const int* __restrict__ num_per_expert,
    int rows_per_expert) {
  const int row = blockIdx.y;
  const int rows = num_rows_ptr ? min(*num_rows_ptr, max_rows) : max_rows;
  if (row >= rows) return;
  // masked (DeepEP-layout) form: expert e owns rows [e * rows_per_expert, ...), the first num_per_expert[e] valid
  if (num_per_expert && row % rows_per_expert >= num_per_expert[row / rows_per_expert]) return;
  const int c8 = blockIdx.x * kThreads + threadIdx.x;  // chunk of 8 columns
  const bool ok = c8 * 8 < inter;
  const uint16_t* gp = gate_up + static_cast<long>(row) * 2 * inter;
  float a[8];
  float amax = 0.f;
  if (ok) {
    const u32x4 gv = ld16(gp + c8 * 8), uv = ld16(gp + inter + c8 * 8);
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const float g0 = bf16lo_to_f32(gv[j]), g1 = bf16hi_to_f32(gv[j]);
      const float u0 = bf16lo_to_f32(uv[j]), u1 = bf16hi_to_f32(uv[j]);
      a[2 * j] = g0 / (1.0f + __expf(-g0)) * u0;
      a[2 * j + 1] = g1 / (1.0f + __expf(-g1)) * u1;
      amax = fmaxf(amax, fmaxf(fabsf(a[2 * j]), fabsf(a[2 * j + 1])));
    }
  }
#pragma unroll
  for (int o = 8; o > 0; o >>= 1) amax = fmaxf(amax, __shfl_xor(amax, o, 64));
  if (!ok) return;
  const float scale = amax / 448.0f;
  const float inv = 1.0f / (scale + 1e-8f);
  u32x2 q;
  q[0] = quant_4xe4m3(a[0] * inv, a[1] * inv, a[2] * inv, a[3] * inv);
  q[1] = quant_4xe4m3(a[4] * inv, a[5] * inv, a[6] * inv, a[7] * inv);
  *reinterpret_cast<u32x2*>(out + static_cast<long>(row) * inter + c8 * 8) = q;
  if ((threadIdx.x & 15) == 0) {
    const long r = row_to_col ? row_to_col[row] : row;
    out_scale[r * os_row_stride + (c8 >> 4) * os_blk_stride] = scale;
  }
}

// Per-tensor variant: q = e4m3( silu(gate) * up * scale[0] ); with use_bf16_mul the product is formed
// in bf16 like the reference kernel (src/activation/activation.cu:19-75, :54-65).
__global__ __launch_bounds__(kThreads) void act_mul_quant_kernel(
    const uint16_t* __restrict__ gate_up, const float* __restrict__ scale,
    const int* __restrict__ num_rows_ptr, int max_rows, int inter, int use_bf16_mul,
    uint8_t* __restrict__ out, const int* __restrict__ num_per_expert, int rows_per_expert) {
  const int row = blockIdx.y;
  const int rows = num_rows_ptr ? min(*num_rows_ptr, max_rows) : max_rows;
  if (row >= rows) return;
  if (num_per_expert && row % rows_per_expert >= num_per_expert[row / rows_per_expert]) return;
  const int c8 = blockIdx.x * kThreads + threadIdx.x;
  if (c8 * 8 >= inter) return;
  const float sc = scale[0];
  const uint16_t* gp = gate_up + static_cast<long>(row) * 2 * inter;
  const u32x4 gv = ld16(gp + c8 * 8), uv = ld16(gp + inter + c8 * 8);
  float a[8];
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const float g0 = bf16lo_to_f32(gv[j]), g1 = bf16hi_to_f32(gv[j]);
    const float u0 = bf16lo_to_f32(uv[j]), u1 = bf16hi_to_f32(uv[j]);
    float s0 = g0 / (1.0f + __expf(-g0)), s1 = g1 / (1.0f + __expf(-g1));
    if (use_bf16_mul) {
      s0 = bf16_to_f32(f32_to_bf16(bf16_to_f32(f32_to_bf16(s0)) * u0));
      s1 = bf16_to_f32(f32_to_bf16(bf16_to_f32(f32_to_bf16(s1)) * u1));
    } else {
      s0 *= u0;
      s1 *= u1;
    }
    a[2 * j] = s0 * sc;
    a[2 * j + 1] = s1 * sc;
  }
  u32x2 q;
  q[0] = quant_4xe4m3(a[0], a[1], a[2], a[3]);
  q[1] = quant_4xe4m3(a[4], a[5], a[6], a[7]);
  *reinterpret_cast<u32x2*>(out + static_cast<long>(row) * inter + c8 * 8) = q;
}

// out = e4m3(float(in) * (1.0f / scale[0]))  (reference scaled_fp8_quant_kernel, src/activation/activation.cu:461-505:
// the reciprocal is taken once, IEEE division, and every element is MULTIPLIED by it - so is this).
// kIn: 0 = bf16, 1 = fp16, 2 = fp32.  A thread converts 8 elements per trip (16 B in for the 2-byte types, 2 x 16 B
// for fp32, 8 B out); the < 8 elements of a ragged tail are done one by one by the first threads of the grid.
template <int kIn>
__device__ __forceinline__ float quant_in_to_f32(const void* in, long i) {
  if constexpr (kIn == 0) return bf16_to_f32(static_cast<const uint16_t*>(in)[i]);
  if constexpr (kIn == 1) return static_cast<float>(static_cast<const _Float16*>(in)[i]);
  return static_cast<const float*>(in)[i];
}
template <int kIn>
__global__ __launch_bounds__(kThreads) void scaled_fp8_quant_kernel(const void* __restrict__ in,
                                                                    const float* __restrict__ scale,
                                                                    long numel, uint8_t* __restrict__ out) {
  const float inv = 1.0f / scale[0];
  const long n8 = numel >> 3;
  const long tid = static_cast<long>(blockIdx.x) * kThreads + threadIdx.x;
  for (long i = tid; i < n8; i += static_cast<long>(gridDim.x) * kThreads) {
    float a[8];
    if constexpr (kIn == 0) {
      const u32x4 v = ld16(static_cast<const uint16_t*>(in) + i * 8);
#pragma unroll
      for (int j = 0; j < 4; ++j) a[2 * j] = bf16lo_to_f32(v[j]), a[2 * j + 1] = bf16hi_to_f32(v[j]);
    } else if constexpr (kIn == 1) {
      typedef _Float16 h8 __attribute__((ext_vector_type(8)));
      const h8 v = *reinterpret_cast<const h8*>(static_cast<const _Float16*>(in) + i * 8);
#pragma unroll
      for (int j = 0; j < 8; ++j) a[j] = static_cast<float>(v[j]);
    } else {
      typedef float f4 __attribute__((ext_vector_type(4)));
      const f4 v0 = *reinterpret_cast<const f4*>(static_cast<const float*>(in) + i * 8);
      const f4 v1 = *reinterpret_cast<const f4*>(static_cast<const float*>(in) + i * 8 + 4);
#pragma unroll
      for (int j = 0; j < 4; ++j) a[j] = v0[j], a[4 + j] = v1[j];
    }
    u32x2 q;
    q[0] = quant_4xe4m3(a[0] * inv, a[1] * inv, a[2] * inv, a[3] * inv);
    q[1] = quant_4xe4m3(a[4] * inv, a[5] * inv, a[6] * inv, a[7] * inv);
    *reinterpret_cast<u32x2*>(out + i * 8) = q;
  }
  const long t = n8 * 8 + tid;
  if (t < numel) {
    const float v = quant_in_to_f32<kIn>(in, t) * inv;
    out[t] = static_cast<uint8_t>(quant_4xe4m3(v, 0.f, 0.f, 0.f) & 0xff);
  }
}

// rows only (per-tensor count_and_gather): xg[pos] = x[token]
__global__ __launch_bounds__(kThreads) void gather_rows_kernel(const uint8_t* __restrict__ x,
                                                               const int* __restrict__ topk_pos, int n,
                                                               int num_topk, int hidden,
                                                               uint8_t* __restrict__ xg) {
  const int i = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (i >= n) return;
  const int pos = topk_pos[i];
  if (pos < 0) return;
  const int lane = threadIdx.x & 63;
  const uint8_t* src = x + static_cast<long>(i / num_topk) * hidden;
  uint8_t* dst = xg + static_cast<long>(pos) * hidden;
  for (int c = lane * 16; c < hidden; c += 64 * 16) st16(dst + c, ld16(src + c));
}

// y[t] = bf16( sum_j topk_scale[t,j] * float(x[topk_pos[t,j]]) + float(shared[t]) ), pos < 0 skipped
// (reference reduce_kernel, src/fuse_moe/reduce.cu:17-81).
__global__ __launch_bounds__(kThreads) void reduce_kernel(
    const uint16_t* __restrict__ x, const int* __restrict__ topk_pos,
    const float* __restrict__ topk_scale, const uint16_t* __restrict__ shared, int num_topk,
    int hidden, uint16_t* __restrict__ y) {
  __shared__ int s_pos[128];
  __shared__ float s_scale[128];
  const int t = blockIdx.y;
  if (threadIdx.x < num_topk) {
    s_pos[threadIdx.x] = topk_pos[static_cast<long>(t) * num_topk + threadIdx.x];
    s_scale[threadIdx.x] = topk_scale[static_cast<long>(t) * num_topk + threadIdx.x];
  }
  __syncthreads();
  const int c = (blockIdx.x * kThreads + threadIdx.x) * 8;
  if (c >= hidden) return;
  float acc[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) acc[i] = 0.f;
  for (int j = 0; j < num_topk; ++j) {
    const int pos = s_pos[j];
    if (pos < 0) continue;
    const float sc = s_scale[j];
    const u32x4 v = ld16(x + static_cast<long>(pos) * hidden + c);
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      acc[2 * i] = fmaf(bf16lo_to_f32(v[i]), sc, acc[2 * i]);
      acc[2 * i + 1] = fmaf(bf16hi_to_f32(v[i]), sc, acc[2 * i + 1]);
    }
  }
  if (shared) {
    const u32x4 v = ld16(shared + static_cast<long>(t) * hidden + c);
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      acc[2 * i] += bf16lo_to_f32(v[i]);
      acc[2 * i + 1] += bf16hi_to_f32(v[i]);
    }
  }
  u32x4 o;
#pragma unroll
  for (int i = 0; i < 4; ++i) o[i] = pack_bf16x2(acc[2 * i], acc[2 * i + 1]);
  st16(y + static_cast<long>(t) * hidden + c, o);
}

}  // namespace moe
}  // namespace hpc

using namespace hpc::moe;

// ---- routing prep --------------------------------------------------------------------------------
extern "C" int hpc_moe_count_and_slot_async(const void* topk_ids, int num_tokens, int num_topk,
                                            int num_expert, int rank_ep, int tile_m,
                                            void* seqlens, void* cu_seqlens, void* tiles,
                                            void* cu_tiles, void* topk_pos, void* row_index,
                                            hipStream_t stream) {
  if (!topk_ids || !seqlens || !cu_seqlens || !tiles || !cu_tiles || !topk_pos || !row_index)
    return HPC_ERR_INVALID;
  if (num_expert <= 0 || num_topk <= 0 || tile_m <= 0 || num_tokens < 0) return HPC_ERR_INVALID;
  const int n = num_tokens * num_topk;
  const int first = rank_ep * num_expert;
  // 16-byte id loads when the array allows them (any tensor torch hands over does; a caller's odd offset gets 4-byte loads);
  // 16 waves per expert from 4096 routed ids on (the walk is paced by instruction issue per wave), 4 below
  const bool vec = (reinterpret_cast<uintptr_t>(topk_ids) & 15) == 0;
  const bool wide = n >= 4096;
  const int* ids = static_cast<const int*>(topk_ids);
#define HPC_ROUTING_PREP(VEC, W)                                                                                       \
  do {                                                                                                                 \
    count_kernel<VEC, W><<<num_expert, W * 64, 0, stream>>>(ids, n, first, static_cast<int*>(seqlens));                \
    HPC_CHECK_LAUNCH();                                                                                                \
    slot_kernel<VEC, W><<<num_expert, W * 64, 0, stream>>>(                                                            \
        ids, n, num_topk, first, num_expert, tile_m, static_cast<const int*>(seqlens), static_cast<int*>(cu_seqlens),  \
        static_cast<int*>(tiles), static_cast<int*>(cu_tiles), static_cast<int*>(topk_pos), static_cast<int*>(row_index)); \
  } while (0)
  if (vec && wide) HPC_ROUTING_PREP(true, 16);
  else if (vec) HPC_ROUTING_PREP(true, 4);
  else if (wide) HPC_ROUTING_PREP(false, 16);
  else HPC_ROUTING_PREP(false, 4);
#undef HPC_ROUTING_PREP
  HPC_CHECK_LAUNCH();
  return HPC_OK;
}

extern "C" int hpc_moe_gather_blockwise_async(const void* x, const void* x_scale,
                                              const void* topk_ids, const void* topk_pos,
                                              const void* cu_seqlens, const void* cu_tiles,
                                              int num_tokens, int num_topk, int num_expert,
                                              int rank_ep, int hidden, int tile_m, int m_pad,
                                              void* x_gathered, void* xscale_t, hipStream_t stream) {
  if (!x || !x_scale || !topk_ids || !topk_pos || !cu_seqlens || !cu_tiles || !x_gathered || !xscale_t)
    return HPC_ERR_INVALID;
  if (hidden & 127) return HPC_ERR_UNSUPPORTED;
  const int n = num_tokens * num_topk;
  if (n == 0) return HPC_OK;
  gather_kernel<<<(n + 3) / 4, kThreads, 0, stream>>>(
      static_cast<const uint8_t*>(x), static_cast<const float*>(x_scale),
      static_cast<const int*>(topk_ids), static_cast<const int*>(topk_pos),
      static_cast<const int*>(cu_seqlens), static_cast<const int*>(cu_tiles), n, num_topk,
      rank_ep * num_expert, hidden, tile_m, m_pad, static_cast<uint8_t*>(x_gathered),
      static_cast<float*>(xscale_t));
  HPC_CHECK_LAUNCH();
  return HPC_OK;
}

extern "C" int hpc_moe_tiles_async(const void* seqlens, int num_group, int tile_m, void* tiles,
                                   void* cu_tiles, hipStream_t stream) {
  if (!seqlens || !tiles || !cu_tiles || num_group <= 0 || tile_m <= 0) return HPC_ERR_INVALID;
  tiles_kernel<<<1, 64, 0, stream>>>(static_cast<const int*>(seqlens), num_group, tile_m,
                                     static_cast<int*>(tiles), static_cast<int*>(cu_tiles));
  HPC_CHECK_LAUNCH();
  return HPC_OK;
}

// ---- x_scale [rows, K/128] (per group: rows cu_seqlens[g] ..) -> transposed, tile-padded, compact
//      [K/128, m_pad]: column = (sum_{j<g} ceil(seqlens[j] / tilem)) * tilem + slot.
// reference: reformat_x_scale_async (src/group_gemm/group_gemm.h:27-29), the DeepEP-layout companion of
// group_gemm_blockwise_fp8.  Padding columns are left untouched.
namespace {
__global__ void reformat_x_scale_kernel(float* __restrict__ out, const float* __restrict__ xs,
                                        const int* __restrict__ seqlens, const int* __restrict__ cu_seqlens,
                                        int m, int n, int tilem) {
  const int g = blockIdx.x;
  int col0 = 0;
  for (int j = 0; j < g; ++j) col0 += (seqlens[j] + tilem - 1) / tilem;
  col0 *= tilem;
  const int cnt = seqlens[g], row0 = cu_seqlens[g];
  for (int i = threadIdx.x; i < cnt * n; i += blockDim.x) {
    const int slot = i / n, kb = i % n;  // coalesced reads; the transposed writes are small
    out[static_cast<long>(kb) * m + col0 + slot] = xs[static_cast<long>(row0 + slot) * n + kb];
  }
}
}  // namespace

extern "C" int hpc_reformat_x_scale_async(void* output_ptr, const void* xscale_ptr, const void* seqlens_ptr,
                                          const void* cu_seqlens_ptr, int num_group, int m, int n, int tilem,
                                          hipStream_t stream) {
  if (!output_ptr || !xscale_ptr || !seqlens_ptr || !cu_seqlens_ptr) return HPC_ERR_INVALID;
  if (num_group <= 0 || m <= 0 || n <= 0 || tilem <= 0) return HPC_ERR_INVALID;
  reformat_x_scale_kernel<<<num_group, 256, 0, stream>>>(static_cast<float*>(output_ptr),
                                                         static_cast<const float*>(xscale_ptr),
                                                         static_cast<const int*>(seqlens_ptr),
                                                         static_cast<const int*>(cu_seqlens_ptr), m, n, tilem);
  HPC_CHECK_LAUNCH();
  return HPC_OK;
}

// ---- activation + block quant ---------------------------------------------------------------------
extern "C" int hpc_act_mul_and_blockwise_quant_async(void* out_ptr, void* out_scale_ptr,
                                                     const void* gate_up_ptr,
                                                     const void* num_rows_ptr, int max_rows,
                                                     int intermediate_size, int64_t scale_row_stride,
                                                     int64_t scale_block_stride,
                                                     const void* row_to_col_ptr, hipStream_t stream) {
  if (!out_ptr || !out_scale_ptr || !gate_up_ptr) return HPC_ERR_INVALID;
  if (intermediate_size <= 0 || (intermediate_size & 127)) return HPC_ERR_UNSUPPORTED;
  if (max_rows <= 0) return HPC_OK;
  dim3 grid((intermediate_size / 8 + kThreads - 1) / kThreads, max_rows);
  act_mul_blockwise_quant_kernel<<<grid, kThreads, 0, stream>>>(
      static_cast<const uint16_t*>(gate_up_ptr), static_cast<const int*>(num_rows_ptr), max_rows,
      intermediate_size, static_cast<uint8_t*>(out_ptr), static_cast<float*>(out_scale_ptr),
      scale_row_stride, scale_block_stride, static_cast<const int*>(row_to_col_ptr), nullptr, 1);
  HPC_CHECK_LAUNCH();
  return HPC_OK;
}

// Masked (DeepEP-layout) forms: gate_up [num_expert * rows_per_expert, 2 I]; only the first num_per_expert[e]
// rows of every expert are computed, the others are left untouched.
// reference: masked_act_mul_and_quant_async / masked_act_mul_and_blockwise_quant_async
// (src/activation/activation.h, entry src/activation/entry.cc:50-110).
extern "C" int hpc_masked_act_mul_and_blockwise_quant_async(void* out_ptr, void* out_scale_ptr,
                                                            const void* gate_up_ptr, const void* num_per_expert_ptr,
                                                            int num_total_tokens, int intermediate_size,
                                                            int num_tokens_per_expert, hipStream_t stream) {
  if (!out_ptr || !out_scale_ptr || !gate_up_ptr || !num_per_expert_ptr) return HPC_ERR_INVALID;
  if (intermediate_size <= 0 || (intermediate_size & 127) || num_tokens_per_expert <= 0) return HPC_ERR_UNSUPPORTED;
  if (num_total_tokens <= 0) return HPC_OK;
  dim3 grid((intermediate_size / 8 + kThreads - 1) / kThreads, num_total_tokens);
  act_mul_blockwise_quant_kernel<<<grid, kThreads, 0, stream>>>(
      static_cast<const uint16_t*>(gate_up_ptr), nullptr, num_total_tokens, intermediate_size,
      static_cast<uint8_t*>(out_ptr), static_cast<float*>(out_scale_ptr), intermediate_size / 128, 1, nullptr,
      static_cast<const int*>(num_per_expert_ptr), num_tokens_per_expert);
  HPC_CHECK_LAUNCH();
  return HPC_OK;
}

extern "C" int hpc_masked_act_mul_and_quant_async(void* out_ptr, const void* gate_up_ptr, const void* scale_ptr,
                                                  const void* num_per_expert_ptr, int num_total_tokens,
                                                  int intermediate_size, int num_tokens_per_expert,
                                                  hipStream_t stream) {
  if (!out_ptr || !gate_up_ptr || !scale_ptr || !num_per_expert_ptr) return HPC_ERR_INVALID;
  if (intermediate_size <= 0 || (intermediate_size & 7) || num_tokens_per_expert <= 0) return HPC_ERR_UNSUPPORTED;
  if (num_total_tokens <= 0) return HPC_OK;
  dim3 grid((intermediate_size / 8 + kThreads - 1) / kThreads, num_total_tokens);
  act_mul_quant_kernel<<<grid, kThreads, 0, stream>>>(
      static_cast<const uint16_t*>(gate_up_ptr), static_cast<const float*>(scale_ptr), nullptr, num_total_tokens,
      intermediate_size, 0, static_cast<uint8_t*>(out_ptr), static_cast<const int*>(num_per_expert_ptr),
      num_tokens_per_expert);
  HPC_CHECK_LAUNCH();
  return HPC_OK;
}

extern "C" int hpc_act_mul_and_quant_async(void* out_ptr, const void* gate_up_ptr, const void* scale_ptr,
                                           const void* num_rows_ptr, int max_rows,
                                           int intermediate_size, int use_bf16_mul, hipStream_t stream) {
  if (!out_ptr || !gate_up_ptr || !scale_ptr) return HPC_ERR_INVALID;
  if (intermediate_size <= 0 || (intermediate_size & 7)) return HPC_ERR_UNSUPPORTED;
  if (max_rows <= 0) return HPC_OK;
  dim3 grid((intermediate_size / 8 + kThreads - 1) / kThreads, max_rows);
  act_mul_quant_kernel<<<grid, kThreads, 0, stream>>>(
      static_cast<const uint16_t*>(gate_up_ptr), static_cast<const float*>(scale_ptr),
      static_cast<const int*>(num_rows_ptr), max_rows, intermediate_size, use_bf16_mul ? 1 : 0,
      static_cast<uint8_t*>(out_ptr), nullptr, 1);
  HPC_CHECK_LAUNCH();
  return HPC_OK;
}

extern "C" int hpc_scaled_fp8_quant_async(void* out_ptr, const void* in_ptr, const void* scale_ptr,
                                          int64_t numel, int in_dtype, hipStream_t stream) {
  if (!out_ptr || !in_ptr || !scale_ptr) return HPC_ERR_INVALID;
  if (in_dtype < 0 || in_dtype > 2) return HPC_ERR_UNSUPPORTED;
  if (numel <= 0) return HPC_ERR_INVALID;  // reference: "input must be non-empty" (src/activation/entry.cc:165)
  const long n8 = numel / 8;
  const long want = n8 / kThreads + 1;
  const int grid = static_cast<int>(want < 4096 ? want : 4096);
  const float* sp = static_cast<const float*>(scale_ptr);
  uint8_t* op = static_cast<uint8_t*>(out_ptr);
  if (in_dtype == 0)
    scaled_fp8_quant_kernel<0><<<grid, kThreads, 0, stream>>>(in_ptr, sp, numel, op);
  else if (in_dtype == 1)
    scaled_fp8_quant_kernel<1><<<grid, kThreads, 0, stream>>>(in_ptr, sp, numel, op);
  else
    scaled_fp8_quant_kernel<2><<<grid, kThreads, 0, stream>>>(in_ptr, sp, numel, op);
  HPC_CHECK_LAUNCH();
  return HPC_OK;
}

extern "C" int hpc_moe_gather_rows_async(const void* x, const void* topk_pos, int num_tokens,
                                         int num_topk, int hidden, void* x_gathered, hipStream_t stream) {
  if (!x || !topk_pos || !x_gathered) return HPC_ERR_INVALID;
  if (hidden & 15) return HPC_ERR_UNSUPPORTED;
  const int n = num_tokens * num_topk;
  if (n <= 0) return HPC_OK;
  gather_rows_kernel<<<(n + 3) / 4, kThreads, 0, stream>>>(
      static_cast<const uint8_t*>(x), static_cast<const int*>(topk_pos), n, num_topk, hidden,
      static_cast<uint8_t*>(x_gathered));
  HPC_CHECK_LAUNCH();
  return HPC_OK;
}

// ---- top-k reduce ------------------------------------------------------------------------------------
extern "C" int hpc_moe_reduce_async(void* y_ptr, const void* x_ptr, const void* topk_pos_ptr,
                                    const void* topk_scale_ptr, const void* shared_output_ptr,
                                    int num_tokens, int num_topk, int hidden_size,
                                    hipStream_t stream) {
  if (!y_ptr || !x_ptr || !topk_pos_ptr || !topk_scale_ptr) return HPC_ERR_INVALID;
  if (num_topk <= 0 || num_topk > 128 || (hidden_size & 7)) return HPC_ERR_UNSUPPORTED;
  if (num_tokens <= 0) return HPC_OK;
  dim3 grid((hidden_size / 8 + kThreads - 1) / kThreads, num_tokens);
  reduce_kernel<<<grid, kThreads, 0, stream>>>(
      static_cast<const uint16_t*>(x_ptr), static_cast<const int*>(topk_pos_ptr),
      static_cast<const float*>(topk_scale_ptr), static_cast<const uint16_t*>(shared_output_ptr),
      num_topk, hidden_size, static_cast<uint16_t*>(y_ptr));
  HPC_CHECK_LAUNCH();
  return HPC_OK;
}

// ---- fused pipeline -------------------------------------------------------------------------------------
namespace {
struct MoeWs {
  int64_t seqlens, cu_seqlens, tiles, cu_tiles, topk_pos, row_index, gate_up_out, down_in,
      down_in_scale, down_out, total;
};
inline int64_t align256(int64_t v) { return (v + 255) / 256 * 256; }
MoeWs moe_ws_layout(int num_tokens, int num_topk, int hidden, int inter2, int num_expert) {
  MoeWs w;
  const int64_t m = static_cast<int64_t>(num_tokens) * num_topk;
  const int64_t inter = inter2 / 2;
  int64_t off = 0;
  auto take = [&](int64_t bytes) {
    const int64_t o = off;
    off += align256(bytes);
    return o;
  };
  w.seqlens = take(4 * num_expert);
  w.cu_seqlens = take(4 * (num_expert + 1));
  w.tiles = take(4 * num_expert);
  w.cu_tiles = take(4 * (num_expert + 1));
  w.topk_pos = take(4 * m);
  w.row_index = take(4 * m);
  w.gate_up_out = take(2 * m * inter2);
  w.down_in = take(m * inter);
  w.down_in_scale = take(4 * m * (inter / 128));
  w.down_out = take(2 * m * hidden);
  w.total = off;
  return w;
}
}  // namespace

extern "C" int64_t hpc_fuse_moe_blockwise_workspace_bytes(int num_tokens, int num_topk,
                                                          int hidden_size, int intermediate_size2,
                                                          int num_expert) {
  if (num_tokens < 0 || num_topk <= 0 || hidden_size <= 0 || intermediate_size2 <= 0 || num_expert <= 0)
    return HPC_ERR_INVALID;
  return moe_ws_layout(num_tokens, num_topk, hidden_size, intermediate_size2, num_expert).total;
}

extern "C" int hpc_fuse_moe_blockwise_async(
    void* y_ptr, void* workspace, const void* x_ptr, const void* x_scale_ptr,
    const void* gate_up_weight_ptr, const void* gate_up_weight_scale_ptr,
    const void* down_weight_ptr, const void* down_weight_scale_ptr, const void* topk_ids_ptr,
    const void* topk_scale_ptr, const void* shared_output_ptr, int num_tokens, int hidden_size,
    int intermediate_size2, int num_topk, int num_expert_total, int num_expert, int gate_up_ws_pad4,
    int down_ws_pad4, int rank_ep, hipStream_t stream) {
  if (!y_ptr || !workspace || !x_ptr || !x_scale_ptr || !gate_up_weight_ptr ||
      !gate_up_weight_scale_ptr || !down_weight_ptr || !down_weight_scale_ptr || !topk_ids_ptr ||
      !topk_scale_ptr)
    return HPC_ERR_INVALID;
  if ((hidden_size & 127) || (intermediate_size2 & 255) || num_topk > 128) return HPC_ERR_UNSUPPORTED;
  if (num_tokens <= 0) return HPC_OK;
  (void)num_expert_total;
  const int inter = intermediate_size2 / 2;
  const int m = num_tokens * num_topk;
  const MoeWs w = moe_ws_layout(num_tokens, num_topk, hidden_size, intermediate_size2, num_expert);
  char* ws = static_cast<char*>(workspace);
  int rc = hpc_moe_count_and_slot_async(topk_ids_ptr, num_tokens, num_topk, num_expert, rank_ep, 128,
                                        ws + w.seqlens, ws + w.cu_seqlens, ws + w.tiles,
                                        ws + w.cu_tiles, ws + w.topk_pos, ws + w.row_index, stream);
  if (rc) return rc;
  // gate_up: rows come straight from x through row_index, scales from x_scale[token][kb].  With the 256 x 256 tile
  // kernel the activation + quantisation runs in the GEMM's epilogue (a tile = 128 gate rows + the 128 up rows of
  // the same columns) and the bf16 gate-up matrix is never written (development key 19 = 1: keep them apart)
  if ((inter & 127) == 0 && hpc_dev_tuning_get(19) != 1 &&
      hpc_ggemm_p8_selected(num_expert, m, intermediate_size2, hidden_size, ws + w.cu_tiles)) {
    rc = hpc_group_gemm_blockwise_fp8_act(ws + w.down_in, ws + w.down_in_scale, x_ptr, gate_up_weight_ptr,
                                          ws + w.seqlens, ws + w.cu_seqlens, x_scale_ptr, gate_up_weight_scale_ptr,
                                          ws + w.row_index, num_expert, m, intermediate_size2, hidden_size,
                                          gate_up_ws_pad4, hidden_size / 128, 1, ws + w.cu_tiles, stream);
    if (rc) return rc;
  } else {
    rc = hpc_group_gemm_blockwise_fp8_async(ws + w.gate_up_out, x_ptr, gate_up_weight_ptr,
                                            ws + w.seqlens, ws + w.cu_seqlens, x_scale_ptr,
                                            gate_up_weight_scale_ptr, ws + w.row_index, nullptr,
                                            num_expert, m, intermediate_size2, hidden_size,
                                            gate_up_ws_pad4, 16, hidden_size / 128, 1, ws + w.cu_tiles, stream);
    if (rc) return rc;
    rc = hpc_act_mul_and_blockwise_quant_async(
        ws + w.down_in, ws + w.down_in_scale, ws + w.gate_up_out,
        reinterpret_cast<const int*>(ws + w.cu_seqlens) + num_expert, m, inter, inter / 128, 1, nullptr,
        stream);
    if (rc) return rc;
  }
  rc = hpc_group_gemm_blockwise_fp8_async(ws + w.down_out, ws + w.down_in, down_weight_ptr,
                                          ws + w.seqlens, ws + w.cu_seqlens, ws + w.down_in_scale,
                                          down_weight_scale_ptr, nullptr, nullptr, num_expert, m,
                                          hidden_size, inter, down_ws_pad4, 16, inter / 128, 1,
                                          ws + w.cu_tiles, stream);
  if (rc) return rc;
  return hpc_moe_reduce_async(y_ptr, ws + w.down_out, ws + w.topk_pos, topk_scale_ptr,
                              shared_output_ptr, num_tokens, num_topk, hidden_size, stream);
}

// Per-tensor pipeline (reference fuse_moe_async, src/fuse_moe/fuse_moe.cu:14-60, and the gather-free
// cp.async pipeline cp_async/fuse_moe.cu:16-63): one fp32 scale per expert on each GEMM, one
// activation scale.  Same workspace layout as the blockwise pipeline (the scale buffer is unused).
extern "C" int hpc_fuse_moe_pertensor_async(
    void* y_ptr, void* workspace, const void* x_ptr, const void* gate_up_weight_ptr,
    const void* down_weight_ptr, const void* gate_up_scale_ptr, const void* down_scale_ptr,
    const void* act_and_mul_scale_ptr, const void* topk_ids_ptr, const void* topk_scale_ptr,
    const void* shared_output_ptr, int num_tokens, int hidden_size, int intermediate_size2,
    int num_topk, int num_expert, int rank_ep, int use_bf16_mul, hipStream_t stream) {
  if (!y_ptr || !workspace || !x_ptr || !gate_up_weight_ptr || !down_weight_ptr || !gate_up_scale_ptr ||
      !down_scale_ptr || !act_and_mul_scale_ptr || !topk_ids_ptr || !topk_scale_ptr)
    return HPC_ERR_INVALID;
  if ((hidden_size & 63) || (intermediate_size2 & 127) || num_topk > 128) return HPC_ERR_UNSUPPORTED;
  if (num_tokens <= 0) return HPC_OK;
  const int inter = intermediate_size2 / 2;
  const int m = num_tokens * num_topk;
  const MoeWs w = moe_ws_layout(num_tokens, num_topk, hidden_size, (intermediate_size2 + 255) / 256 * 256,
                                num_expert);
  char* ws = static_cast<char*>(workspace);
  int rc = hpc_moe_count_and_slot_async(topk_ids_ptr, num_tokens, num_topk, num_expert, rank_ep, 128,
                                        ws + w.seqlens, ws + w.cu_seqlens, ws + w.tiles,
                                        ws + w.cu_tiles, ws + w.topk_pos, ws + w.row_index, stream);
  if (rc) return rc;
  // With the 256 x 256 tile kernel the activation + quantisation runs in the gate-up GEMM's epilogue (a tile = 128
  // gate rows + the 128 up rows of the same columns): the bf16 gate-up matrix is never written (development key
  // 19 = 1 keeps the two kernels apart)
  if ((inter & 127) == 0 && (hidden_size & 127) == 0 && hpc_dev_tuning_get(19) != 1 &&
      hpc_ggemm_p8_selected(num_expert, m, intermediate_size2, hidden_size, ws + w.cu_tiles)) {
    rc = hpc_group_gemm_pertensor_fp8_act(ws + w.down_in, x_ptr, gate_up_weight_ptr, ws + w.seqlens, ws + w.cu_seqlens,
                                          gate_up_scale_ptr, ws + w.row_index, act_and_mul_scale_ptr, use_bf16_mul,
                                          num_expert, m, num_tokens, intermediate_size2, hidden_size, ws + w.cu_tiles,
                                          stream);
    if (rc) return rc;
  } else {
    rc = hpc_group_gemm_pertensor_fp8_async(ws + w.gate_up_out, x_ptr, gate_up_weight_ptr, ws + w.seqlens,
                                            ws + w.cu_seqlens, gate_up_scale_ptr, ws + w.row_index,
                                            num_expert, m, num_tokens, intermediate_size2, hidden_size,
                                            ws + w.cu_tiles, stream);
    if (rc) return rc;
    rc = hpc_act_mul_and_quant_async(ws + w.down_in, ws + w.gate_up_out, act_and_mul_scale_ptr,
                                     reinterpret_cast<const int*>(ws + w.cu_seqlens) + num_expert, m, inter,
                                     use_bf16_mul, stream);
    if (rc) return rc;
  }
  rc = hpc_group_gemm_pertensor_fp8_async(ws + w.down_out, ws + w.down_in, down_weight_ptr, ws + w.seqlens,
                                          ws + w.cu_seqlens, down_scale_ptr, nullptr, num_expert, m, m,
                                          hidden_size, inter, ws + w.cu_tiles, stream);
  if (rc) return rc;
  return hpc_moe_reduce_async(y_ptr, ws + w.down_out, ws + w.topk_pos, topk_scale_ptr,
                              shared_output_ptr, num_tokens, num_topk, hidden_size, stream);
}
