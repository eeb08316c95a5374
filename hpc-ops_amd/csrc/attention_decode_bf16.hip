// Paged decode attention, bf16, D=128, GQA group 4/8, dynamic split-KV task map - gfx950.
//
// Replaces reference src/attention/decode/sm90/dynamic/smallm_bf16_dim128_dynamic*.cu(h)
// (+ the static variants, which become "schedule on the fly") and
// src/attention/decode/splitk_combine_kernels.cuh.
//
// MI355X design (HBM-bound op: 256 B of KV per token per kv-head, ~8 FLOP/B):
//  * KV is streamed HBM -> VGPR with 16-byte loads and never touches LDS: a 64-token tile is
//    consumed by exactly one wave, so an LDS round trip would be pure overhead.  Each wave keeps a
//    whole K tile (64 VGPRs) and a whole V tile (64 VGPRs) in flight and refills a buffer as soon
//    as the MFMAs have consumed it; 8 waves/CU => ~128-256 KB of loads in flight per CU.
//  * KV tokens sit on the MFMA M axis and the (padded to 16) q rows of one GQA group on N:
//    S^T = K Q^T and O^T = V^T P^T with v_mfma_f32_16x16x32_bf16.  In this orientation the
//    softmax row of a q row lives in 4 lanes (lane&15 == row), the P operand of the second GEMM
//    is already in the right lanes (no LDS, no shuffles), and rescaling O is per-lane.
//  * V^T operands need 8 tokens of one dim per lane while memory has dims contiguous: each lane
//    loads 8 token rows x 16 B (full 256-B rows per 16 lanes -> perfectly coalesced) and
//    transposes its private 8x8 bf16 block with 32 v_perm_b32.  Output dims come out permuted
//    (MFMA row m of block j <-> dim 8m+j), undone for free at the epilogue.
//  * A workgroup (= one scheduler bin) is 4 independent waves that take tiles t = w, w+4, ... of
//    the current task, each with its own online softmax; they merge once per task through LDS.
//    Requests that fit one bin are written straight to y; split requests leave fp32 partials
//    (2 slots per bin) that the combine kernel merges with base-2 LSE weights.
#include "hpc_common.h"
#include "../../include/hpc_amd.h"
#include "sched_task_info.h"

namespace hpc {
namespace decode {

using sched::kTaskStride;

struct Args {
  const uint16_t* q;
  const uint16_t* kcache;
  const uint16_t* vcache;
  const int* block_ids;
  const int* task_map;
  uint16_t* y;
  float* part_o;    // [bins][2][16*kNB][128]
  float* part_lse;  // [bins][2][16*kNB]
  int* first_bin;   // [Hkv*B]
  int num_batch, num_seq_q, num_head_kv, g_shift, page_shift, max_blocks;
  int ldq, ldy;
  long k_block_stride, k_token_stride, k_head_stride;
  long v_block_stride, v_token_stride, v_head_stride;
  float scale_log2;
};

constexpr int kThreads = 256;
constexpr int kWaves = 4;
constexpr float kNegInf = -__builtin_inff();

__device__ __forceinline__ uint32_t perm_lo(uint32_t hi_src, uint32_t lo_src) {
  return __builtin_amdgcn_perm(hi_src, lo_src, 0x05040100u);  // {hi_src.lo16, lo_src.lo16}
}
__device__ __forceinline__ uint32_t perm_hi(uint32_t hi_src, uint32_t lo_src) {
  return __builtin_amdgcn_perm(hi_src, lo_src, 0x07060302u);  // {hi_src.hi16, lo_src.hi16}
}

union Frag {
  u32x4 u;
  bf16x8 b;
};

template <int kNB>
__global__ __launch_bounds__(kThreads, kNB == 1 ? 2 : 1) void decode_bf16_kernel(const Args a) {
  static_assert(kNB == 1, "only one 16-row q block for now");
  __shared__ __attribute__((aligned(16))) float s_o[kWaves][16][128 + 4];
  __shared__ float s_m[kWaves][16];
  __shared__ float s_l[kWaves][16];

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int n = lane & 15;   // q row of this lane (MFMA N index)
  const int g = lane >> 4;   // lane group (MFMA k-slot group / C row group)
  const int bin = blockIdx.x;

  const cint_ptr tmap = as_const(a.task_map);
  const int per1 = tmap[0];
  const int num_bins = tmap[1];
  const cint_ptr chunk_tab = tmap + sched::chunk_table_off(per1 - 1, num_bins);
  cint_ptr task_ptr = tmap + static_cast<long>(kTaskStride) * (1 + static_cast<long>(bin) * per1);

  const int G = 1 << a.g_shift;
  const int rows_valid = a.num_seq_q << a.g_shift;
  const int sq = n >> a.g_shift;  // which of the Sq new tokens this q row belongs to
  const int page_mask = (1 << a.page_shift) - 1;

  for (int itask = 0; itask < per1; ++itask, task_ptr += kTaskStride) {
    const int h = __builtin_amdgcn_readfirstlane(task_ptr[0]);
    const int b = __builtin_amdgcn_readfirstlane(task_ptr[1]);
    if (h < 0 || b < 0) break;
    const int ichunk = __builtin_amdgcn_readfirstlane(task_ptr[2]);
    const int iseq_start = __builtin_amdgcn_readfirstlane(task_ptr[3]);
    const int num_seqkv = __builtin_amdgcn_readfirstlane(task_ptr[4]);
    const int num_seqkvcache = __builtin_amdgcn_readfirstlane(task_ptr[5]);
    const int ntile = __builtin_amdgcn_readfirstlane(task_ptr[6]);
    const int ntile_full = __builtin_amdgcn_readfirstlane(task_ptr[7]);

    // ---- Q fragments (B operand of S^T = K Q^T): lane (n, g) holds dims (4j+g)*8 .. +7 --------
    Frag qf[4];
    {
      const bool ok = n < rows_valid;
      const long qoff = static_cast<long>(b * a.num_seq_q + sq) * a.ldq +
                        ((h << a.g_shift) + (n & (G - 1))) * 128;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        qf[j].u = u32x4{0u, 0u, 0u, 0u};
        if (ok) qf[j].u = ld16(a.q + qoff + (4 * j + g) * 8);
      }
    }

    const cint_ptr bid_row = as_const(a.block_ids) + static_cast<long>(b) * a.max_blocks;
    const int last_blk16 = (num_seqkv - 1) >> 4;  // chunk-local, clamp for masked token blocks
    const int base_blk16 = iseq_start >> 4;

    // page ids of the four 16-token blocks of tile t (wave-uniform -> SGPRs)
    auto tile_pages = [&](int t, int (&pid)[4], int (&inpage)[4]) {
#pragma unroll
      for (int tb = 0; tb < 4; ++tb) {
        int blk = t * 4 + tb;
        blk = blk < last_blk16 ? blk : last_blk16;
        const int gtok = (base_blk16 + blk) << 4;
        pid[tb] = __builtin_amdgcn_readfirstlane(bid_row[gtok >> a.page_shift]);
        inpage[tb] = gtok & page_mask;
      }
    };

    // Loads go through buffer descriptors: one wave-uniform 64-bit base per 16-token block, a
    // loop-invariant 32-bit lane offset and immediate / scalar offsets for everything else.
    Frag kf[4][4];  // [token block][k step]
    Frag vf[2][8];  // [k step of PV][token slot]
    const int k_voff = (n * static_cast<int>(a.k_token_stride) + g * 8) * 2;
    const int v_voff = (g * 4 * static_cast<int>(a.v_token_stride) + n * 8) * 2;
    const int v_tok_bytes = static_cast<int>(a.v_token_stride) * 2;
    // `nrec` = 0 turns the loads of a non-existent next tile into no-ops (out-of-range buffer
    // reads return 0 without touching memory) - the loop body stays branch-free, which keeps
    // hipcc's counted vmcnt waits intact.
    auto load_k = [&](const int (&pid)[4], const int (&inpage)[4], unsigned nrec) {
#pragma unroll
      for (int tb = 0; tb < 4; ++tb) {
        const auto rs = make_rsrc(a.kcache + pid[tb] * a.k_block_stride +
                                  inpage[tb] * a.k_token_stride + h * a.k_head_stride, nrec);
#pragma unroll
        for (int j = 0; j < 4; ++j) kf[tb][j].u = buf_ld16(rs, k_voff + 64 * j, 0);
      }
    };
    auto load_v = [&](const int (&pid)[4], const int (&inpage)[4], unsigned nrec) {
#pragma unroll
      for (int ks = 0; ks < 2; ++ks)
#pragma unroll
        for (int hb = 0; hb < 2; ++hb) {
          const int tb = 2 * ks + hb;
          const auto rs = make_rsrc(a.vcache + pid[tb] * a.v_block_stride +
                                    inpage[tb] * a.v_token_stride + h * a.v_head_stride, nrec);
#pragma unroll
          for (int r = 0; r < 4; ++r) vf[ks][hb * 4 + r].u = buf_ld16(rs, v_voff, r * v_tok_bytes);
        }
    };

    f32x4 o[8];
#pragma unroll
    for (int jj = 0; jj < 8; ++jj) o[jj] = f32x4{0.f, 0.f, 0.f, 0.f};
    float m_run = kNegInf;
    float l_run = 0.f;  // per-lane partial of the row sum (4 lanes share a row)

    int t = wave;
    int pid[4], inpage[4];
    {
      const unsigned nrec = t < ntile ? 0xffffffffu : 0u;
      tile_pages(t < ntile ? t : ntile - 1, pid, inpage);
      // pin the issue order K then V: the tile loop's first wait is for K only (vmcnt counts
      // in order), so a reordered prologue would degrade it to vmcnt(0)
      __builtin_amdgcn_sched_barrier(0);
      load_k(pid, inpage, nrec);
      __builtin_amdgcn_sched_barrier(0);
      load_v(pid, inpage, nrec);
      __builtin_amdgcn_sched_barrier(0);
    }
    for (; t < ntile; t += kWaves) {
      const unsigned nrec = t + kWaves < ntile ? 0xffffffffu : 0u;
      tile_pages(t + kWaves < ntile ? t + kWaves : ntile - 1, pid, inpage);

      // ---- S^T = K Q^T ------------------------------------------------------------------
      f32x4 s[4];
#pragma unroll
      for (int tb = 0; tb < 4; ++tb) {
        s[tb] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int j = 0; j < 4; ++j)
          s[tb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kf[tb][j].b, qf[j].b, s[tb], 0, 0, 0);
      }
      load_k(pid, inpage, nrec);

      // ---- online softmax in base 2 ---------------------------------------------------------
      float mt = kNegInf;
      if (t >= ntile_full) {  // tile holds masked keys (tail of the request / causal rows)
        const int lim = min(num_seqkv - 1, num_seqkvcache + sq);
#pragma unroll
        for (int tb = 0; tb < 4; ++tb)
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const int tok = t * 64 + tb * 16 + g * 4 + r;
            s[tb][r] = tok <= lim ? s[tb][r] * a.scale_log2 : kNegInf;
            mt = fmaxf(mt, s[tb][r]);
          }
      } else {
#pragma unroll
        for (int tb = 0; tb < 4; ++tb)
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            s[tb][r] *= a.scale_log2;
            mt = fmaxf(mt, s[tb][r]);
          }
      }
      mt = fmaxf(mt, __shfl_xor(mt, 16, 64));
      mt = fmaxf(mt, __shfl_xor(mt, 32, 64));
      const float m_new = fmaxf(m_run, mt);
      const float m_use = m_new == kNegInf ? 0.f : m_new;  // all-masked rows: p = exp2(-inf) = 0
      const float alpha = __builtin_amdgcn_exp2f(m_run - m_use);
      m_run = m_new;
      float psum = 0.f;
      Frag pf[2];
#pragma unroll
      for (int tb = 0; tb < 4; ++tb) {
        float p[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          p[r] = __builtin_amdgcn_exp2f(s[tb][r] - m_use);
          psum += p[r];
        }
        pf[tb >> 1].u[(tb & 1) * 2] = pack_bf16x2(p[0], p[1]);
        pf[tb >> 1].u[(tb & 1) * 2 + 1] = pack_bf16x2(p[2], p[3]);
      }
      l_run = l_run * alpha + psum;
#pragma unroll
      for (int jj = 0; jj < 8; ++jj) o[jj] *= alpha;

      // ---- O^T += V^T P^T (private 8x8 bf16 transposes feed the A operand) -----------------------
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) {
#pragma unroll
        for (int jj = 0; jj < 8; ++jj) {
          Frag vt;
#pragma unroll
          for (int p2 = 0; p2 < 4; ++p2) {
            const uint32_t lo = vf[ks][2 * p2].u[jj >> 1], hi = vf[ks][2 * p2 + 1].u[jj >> 1];
            vt.u[p2] = (jj & 1) ? perm_hi(hi, lo) : perm_lo(hi, lo);
          }
          o[jj] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(vt.b, pf[ks].b, o[jj], 0, 0, 0);
          if (jj & 1) __builtin_amdgcn_sched_barrier(0);  // keep the transposes next to their MFMAs
        }
      }
      load_v(pid, inpage, nrec);
    }

    // ---- merge the 4 waves of the workgroup --------------------------------------------------------
    l_run += __shfl_xor(l_run, 16, 64);
    l_run += __shfl_xor(l_run, 32, 64);
    if (g == 0) {
      s_m[wave][n] = m_run;
      s_l[wave][n] = l_run;
    }
#pragma unroll
    for (int jj = 0; jj < 8; ++jj)
#pragma unroll
      for (int r = 0; r < 4; ++r) s_o[wave][n][8 * (g * 4 + r) + jj] = o[jj][r];
    __syncthreads();
    {
      const int row = tid >> 4;   // q row
      const int c8 = tid & 15;    // chunk of 8 dims
      float mw[kWaves], M = kNegInf;
#pragma unroll
      for (int w = 0; w < kWaves; ++w) {
        mw[w] = s_m[w][row];
        M = fmaxf(M, mw[w]);
      }
      const float Mu = M == kNegInf ? 0.f : M;
      float L = 0.f, acc[8];
#pragma unroll
      for (int i = 0; i < 8; ++i) acc[i] = 0.f;
#pragma unroll
      for (int w = 0; w < kWaves; ++w) {
        const float wgt = __builtin_amdgcn_exp2f(mw[w] - Mu);
        L += wgt * s_l[w][row];
        const f32x4 x0 = *reinterpret_cast<const f32x4*>(&s_o[w][row][c8 * 8]);
        const f32x4 x1 = *reinterpret_cast<const f32x4*>(&s_o[w][row][c8 * 8 + 4]);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          acc[i] = fmaf(wgt, x0[i], acc[i]);
          acc[4 + i] = fmaf(wgt, x1[i], acc[4 + i]);
        }
      }
      const float inv = L > 0.f ? 1.0f / L : 0.f;
#pragma unroll
      for (int i = 0; i < 8; ++i) acc[i] *= inv;
      const int nchunks = chunk_tab[h * a.num_batch + b];
      if (row < rows_valid) {
        if (nchunks == 1) {
          const int rs = row >> a.g_shift;
          uint16_t* dst = a.y + static_cast<long>(b * a.num_seq_q + rs) * a.ldy +
                          ((h << a.g_shift) + (row & (G - 1))) * 128 + c8 * 8;
          u32x4 pk;
#pragma unroll
          for (int i = 0; i < 4; ++i) pk[i] = pack_bf16x2(acc[2 * i], acc[2 * i + 1]);
          st16(dst, pk);
        } else {
          const long slot = static_cast<long>(bin) * 2 + (ichunk == 0 ? 1 : 0);
          float* po = a.part_o + (slot * 16 + row) * 128 + c8 * 8;
          *reinterpret_cast<f32x4*>(po) = f32x4{acc[0], acc[1], acc[2], acc[3]};
          *reinterpret_cast<f32x4*>(po + 4) = f32x4{acc[4], acc[5], acc[6], acc[7]};
          if (c8 == 0) a.part_lse[slot * 16 + row] = L > 0.f ? M + __builtin_amdgcn_logf(L) : kNegInf;
        }
      }
      if (tid == 0 && nchunks > 1 && ichunk == 0) a.first_bin[h * a.num_batch + b] = bin;
    }
    // gfx950 counts stores in vmcnt too; a store still pending when the next task starts makes
    // hipcc treat the counter as out-of-order and degrade every wait in the tile loop to
    // vmcnt(0).  Retire the epilogue stores here, where nothing else is in flight.
    __builtin_amdgcn_s_waitcnt(0x0F70);  // vmcnt(0)
    __syncthreads();
  }
}

// ---- split-KV combine: y = sum_c 2^(lse_c - max) O_c / sum_c 2^(lse_c - max) ----------------------------
// (reference splitk_combine_kernels.cuh:140-322). One workgroup per (kv head, request); chunk c of
// a request lives in bin first_bin + c, slot 1 for c == 0 and slot 0 otherwise.
__global__ __launch_bounds__(kThreads) void decode_combine_kernel(const Args a) {
  const int hb = blockIdx.x;
  const int per1 = a.task_map[0];
  const int num_bins = a.task_map[1];
  const int* chunk_tab = a.task_map + sched::chunk_table_off(per1 - 1, num_bins);
  const int nchunks = chunk_tab[hb];
  if (nchunks <= 1) return;
  const int h = hb / a.num_batch, b = hb % a.num_batch;
  const int fb = a.first_bin[hb];
  const int tid = threadIdx.x;
  const int row = tid >> 4, c8 = tid & 15;
  const int rows_valid = a.num_seq_q << a.g_shift;
  if (row >= rows_valid) return;
  const int G = 1 << a.g_shift;

  float M = kNegInf;
  for (int c = 0; c < nchunks; ++c) {
    const long slot = static_cast<long>(fb + c) * 2 + (c == 0 ? 1 : 0);
    M = fmaxf(M, a.part_lse[slot * 16 + row]);
  }
  const float Mu = M == kNegInf ? 0.f : M;
  float W = 0.f, acc[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) acc[i] = 0.f;
  for (int c = 0; c < nchunks; ++c) {
    const long slot = static_cast<long>(fb + c) * 2 + (c == 0 ? 1 : 0);
    const float wgt = __builtin_amdgcn_exp2f(a.part_lse[slot * 16 + row] - Mu);
    const float* po = a.part_o + (slot * 16 + row) * 128 + c8 * 8;
    const f32x4 x0 = *reinterpret_cast<const f32x4*>(po);
    const f32x4 x1 = *reinterpret_cast<const f32x4*>(po + 4);
    W += wgt;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      acc[i] = fmaf(wgt, x0[i], acc[i]);
      acc[4 + i] = fmaf(wgt, x1[i], acc[4 + i]);
    }
  }
  const float inv = W > 0.f ? 1.0f / W : 0.f;
  const int rs = row >> a.g_shift;
  uint16_t* dst = a.y + static_cast<long>(b * a.num_seq_q + rs) * a.ldy +
                  ((h << a.g_shift) + (row & (G - 1))) * 128 + c8 * 8;
  u32x4 pk;
#pragma unroll
  for (int i = 0; i < 4; ++i) pk[i] = pack_bf16x2(acc[2 * i] * inv, acc[2 * i + 1] * inv);
  st16(dst, pk);
}

}  // namespace decode
}  // namespace hpc

extern "C" int64_t hpc_attention_decode_workspace_bytes(int num_bins, int num_batch, int num_head_kv,
                                                        int num_seq_q, int heads_per_group) {
  if (num_bins <= 0 || num_batch <= 0 || num_head_kv <= 0) return HPC_ERR_INVALID;
  const int64_t rows = 16;  // q rows per group block (padded)
  (void)num_seq_q;
  (void)heads_per_group;
  const int64_t part_o = static_cast<int64_t>(num_bins) * 2 * rows * 128 * 4;
  const int64_t part_lse = static_cast<int64_t>(num_bins) * 2 * rows * 4;
  const int64_t first_bin = static_cast<int64_t>(num_batch) * num_head_kv * 4;
  return part_o + part_lse + ((first_bin + 15) / 16) * 16;
}

extern "C" int hpc_attention_decode_bf16_async(
    void* y_ptr, void* workspace, const int* task_map_ptr, const void* q_ptr, const void* kcache_ptr,
    const void* vcache_ptr, const int* block_ids_ptr, int num_bins, int num_batch, int num_seq_q,
    int num_head_q, int num_head_kv, int num_dim_qk, int num_dim_v, int block_size,
    int num_seq_max_blocks, int ldY, int ldQ, int64_t kcache_block_stride,
    int64_t kcache_token_stride, int64_t kcache_head_stride, int64_t vcache_block_stride,
    int64_t vcache_token_stride, int64_t vcache_head_stride, hipStream_t stream) {
  using namespace hpc::decode;
  if (!y_ptr || !workspace || !task_map_ptr || !q_ptr || !kcache_ptr || !vcache_ptr || !block_ids_ptr)
    return HPC_ERR_INVALID;
  if (num_dim_qk != 128 || num_dim_v != 128) return HPC_ERR_UNSUPPORTED;
  if (block_size != 16 && block_size != 32 && block_size != 64) return HPC_ERR_UNSUPPORTED;
  if (num_head_kv <= 0 || num_head_q % num_head_kv) return HPC_ERR_INVALID;
  const int group = num_head_q / num_head_kv;
  if (group != 4 && group != 8) return HPC_ERR_UNSUPPORTED;
  if (num_seq_q < 1 || num_seq_q * group > 16) return HPC_ERR_UNSUPPORTED;
  if (num_batch <= 0 || num_bins <= 0) return HPC_ERR_INVALID;
  if ((ldQ & 7) || (ldY & 7) || (kcache_token_stride & 7) || (vcache_token_stride & 7) ||
      (kcache_head_stride & 7) || (vcache_head_stride & 7) || (kcache_block_stride & 7) ||
      (vcache_block_stride & 7))
    return HPC_ERR_UNSUPPORTED;  // 16-byte vector accesses

  Args a;
  a.q = static_cast<const uint16_t*>(q_ptr);
  a.kcache = static_cast<const uint16_t*>(kcache_ptr);
  a.vcache = static_cast<const uint16_t*>(vcache_ptr);
  a.block_ids = block_ids_ptr;
  a.task_map = task_map_ptr;
  a.y = static_cast<uint16_t*>(y_ptr);
  char* ws = static_cast<char*>(workspace);
  a.part_o = reinterpret_cast<float*>(ws);
  ws += static_cast<int64_t>(num_bins) * 2 * 16 * 128 * 4;
  a.part_lse = reinterpret_cast<float*>(ws);
  ws += static_cast<int64_t>(num_bins) * 2 * 16 * 4;
  a.first_bin = reinterpret_cast<int*>(ws);
  a.num_batch = num_batch;
  a.num_seq_q = num_seq_q;
  a.num_head_kv = num_head_kv;
  a.g_shift = group == 8 ? 3 : 2;
  a.page_shift = block_size == 64 ? 6 : (block_size == 32 ? 5 : 4);
  a.max_blocks = num_seq_max_blocks;
  a.ldq = ldQ;
  a.ldy = ldY;
  a.k_block_stride = kcache_block_stride;
  a.k_token_stride = kcache_token_stride;
  a.k_head_stride = kcache_head_stride;
  a.v_block_stride = vcache_block_stride;
  a.v_token_stride = vcache_token_stride;
  a.v_head_stride = vcache_head_stride;
  a.scale_log2 = 0.08838834764831845f * 1.4426950408889634f;  // 1/sqrt(128) * log2(e)

  decode_bf16_kernel<1><<<num_bins, kThreads, 0, stream>>>(a);
  HPC_CHECK_LAUNCH();
  decode_combine_kernel<<<num_batch * num_head_kv, kThreads, 0, stream>>>(a);
  HPC_CHECK_LAUNCH();
  return HPC_OK;
}
