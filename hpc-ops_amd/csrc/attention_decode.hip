// Paged decode attention for gfx950: bf16 and fp8 (e4m3) KV, D=128, GQA group 4/8, dynamic
// split-KV task map, plus the split-KV combine kernel.
//
// Replaces reference src/attention/decode/sm90/{dynamic,static}/smallm_{bf16,fp8_*}_dim128_*.cu(h)
// and src/attention/decode/splitk_combine_kernels.cuh (static split-K becomes "schedule on the
// fly", see hpc/_entry_attention.py).
//
// MI355X design (HBM-bound op: 256/512 B of KV per token per kv-head, ~8 FLOP/B):
//  * KV is streamed HBM -> VGPR with 16-byte (fp8 V: 8-byte) non-temporal buffer loads, every load
//    covering full 128/256-byte row segments (the shape that streams fastest from NHD pages); a
//    64-token tile is consumed by exactly one wave, so there is no shared LDS staging and no
//    barrier in the tile loop (K only bounces through a 4 KB wave-private tile to reach the MFMA
//    operand layout).  Each wave keeps a whole K tile and a whole V tile in flight and refills a
//    buffer as soon as the MFMAs have consumed it; 8 waves/CU => ~128-256 KB in flight per CU.
//  * KV tokens sit on the MFMA M axis and the (padded to 16) q rows of one GQA group on N:
//    S^T = K Q^T and O^T = V^T P^T with v_mfma_f32_16x16x32_{bf16,fp8_fp8}.  In this orientation
//    the softmax row of a q row lives in 4 lanes (lane&15 == row), the P operand of the second
//    GEMM is already in the right lanes (no LDS, no shuffles), and rescaling O is per-lane.
//  * V^T operands need 8 tokens of one dim per lane while memory has dims contiguous: each lane
//    loads 8 token rows x 8 dims (full rows per 16 lanes -> perfectly coalesced) and transposes
//    its private 8x8 block with v_perm_b32.  Output dims come out permuted (MFMA row m of block j
//    <-> dim 8m+j), undone for free at the epilogue.
//  * A workgroup (= one scheduler bin) is 4 independent waves that take tiles t = w, w+4, ... of
//    the current task, each with its own online softmax; they merge once per task through LDS.
//    Requests that fit one bin are written straight to y; split requests leave fp32 partials
//    (2 slots per bin) that the combine kernel merges with base-2 LSE weights.  A bin packed with
//    short tasks (<= 4 tiles each, mixed-length batches) instead gives every wave a whole task
//    ("solo": no workgroup barrier, 4 tasks side by side) - see run_task below.
//  * fp8 numerics follow the reference kernels: scores scaled by
//    qscale[row]*kscale/sqrt(d) in the exp2 domain, P~ = e4m3(256 * 2^(s - running max)),
//    O = sum(P~ V) / sum(p) * vscale / 256.
#include <type_traits>

#include "hpc_common.h"
#include "hpc_dev.h"
#include "../../include/hpc_amd.h"
#include "sched_task_info.h"
#include "attention_decode_v2.h"

namespace hpc {
namespace decode {

using sched::kTaskStride;
#ifdef HPC_DEV
__device__ int g_ticket_overruns;  // arrivals whose ticket exceeded the request's chunk count (a counter not zero on entry)
#endif

struct Args {
  const void* q;
  const void* kcache;
  const void* vcache;
  const int* block_ids;
  const int* task_map;
  uint16_t* y;
  float* part_o;    // [bins][2][16*kNB][128]
  float* part_lse;  // [bins][2][16*kNB]
  int* first_bin;   // [Hkv*B]  (combine-kernel form only)
  int* arrive;      // [Hkv*B] arrival counters of split requests (zero before the call, left zero), or null: the
                    // chunks of a request are merged by decode_combine_kernel in a second launch
  const float* qscale;  // fp8: [B*Sq, qscale_stride]
  const float* kscale;  // fp8: [1] (per tensor) or base of the K-scale rows of the cache (per token)
  const float* vscale;  // fp8: [1] or [Hkv]
  int num_batch, num_seq_q, num_head_kv, g_shift, page_shift, max_blocks;
  int ldq, ldy, qscale_stride;
  int solo_ok;  // bins full of short tasks may run one task per wave (tuning key 5 = 1 turns it off)
  long k_block_stride, k_token_stride, k_head_stride;  // elements
  long v_block_stride, v_token_stride, v_head_stride;
  long ks_block_stride, ks_row_stride, ks_head_stride;  // bytes, per-token K scales
  float scale_log2;
};

constexpr int kThreads = 256;
constexpr int kWaves = 4;
constexpr float kNegInf = -__builtin_inff();

union Frag16 {
  u32x4 u;
  bf16x8 b;
};

__device__ __forceinline__ long pack64(uint32_t lo, uint32_t hi) {
  return static_cast<long>(static_cast<uint64_t>(lo) | (static_cast<uint64_t>(hi) << 32));
}

// kQuant (fp8 only): 1 = q per-token/per-head, k/v per tensor; 0 = k per-token/per-head (scales in
// the page tail rows), v per head.
template <bool kFp8, int kQuant, int kNB, int kAux>
__global__ __launch_bounds__(kThreads, kNB == 1 ? 2 : 1) void decode_kernel(const Args a) {
  constexpr int kEB = kFp8 ? 1 : 2;  // bytes per element
  constexpr int kKC = kFp8 ? 2 : 4;  // 16-byte K chunks per lane per 16-token block
  __shared__ __attribute__((aligned(16))) float s_o[kWaves][16][128 + 4];
  __shared__ float s_m[kWaves][16];
  __shared__ float s_l[kWaves][16];
  // per-wave private staging of one 16-token K block (double-buffered): K is fetched with FULL-ROW
  // loads (16 lanes x 16 B = one 256/128-byte row segment per quarter-wave, the access shape that
  // streams 10-15 % faster from NHD pages than MFMA-fragment-shaped 16 rows x 64 B) and is turned
  // into the MFMA A-operand layout by a write/read through this 4 KB tile.  No barrier: a wave's
  // LDS operations complete in order.
  constexpr int kKRow = (kFp8 ? 128 : 256) + 16;  // padded row: conflict-free b128 reads
  __shared__ __attribute__((aligned(16))) uint8_t s_kt[kWaves][2][16 * kKRow];

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int n = lane & 15;  // q row of this lane inside a 16-row block (MFMA N index)
  const int g = lane >> 4;  // lane group (MFMA k-slot group / C row group)
  const int bin = blockIdx.x;

  const cint_ptr tmap = as_const(a.task_map);
  const int per1 = tmap[0];
  const int num_bins = tmap[1];
  if (bin >= num_bins) return;  // the scheduler planned a small batch on fewer bins than the launch holds (sched_task_info.h)
  const cint_ptr chunk_tab = tmap + sched::chunk_table_off(per1 - 1, num_bins);
  cint_ptr task_ptr = tmap + static_cast<long>(kTaskStride) * (1 + static_cast<long>(bin) * per1);

  const int G = 1 << a.g_shift;
  const int rows_valid = a.num_seq_q << a.g_shift;
  const int page_mask = (1 << a.page_shift) - 1;
  // fp32 partials of split requests go THROUGH to memory (sc1 stores, aux = 16) and are read back with sc1 loads by the
  // workgroup that arrives last at the request: per-XCD L2s are not coherent with each other (attention_decode_v2.hip)
  const auto part_rs = make_rsrc(a.part_o);
  const auto lse_rs = make_rsrc(a.part_lse);
  __shared__ int s_ticket;
  const uint8_t* qbase = static_cast<const uint8_t*>(a.q);
  const uint8_t* kbase = static_cast<const uint8_t*>(a.kcache);
  const uint8_t* vbase = static_cast<const uint8_t*>(a.vcache);

  // ---- how many tasks does this bin hold, and how long is the longest? ------------------------------
  // (lane-parallel scan of the bin's records; the list ends at the first h < 0 / b < 0 record)
  int ntasks = 0, max_ntile = 0;
  for (int base = 0; base < per1; base += 64) {
    const int idx = base + lane;
    int hh = -1, bb = -1, nt = 0;
    if (idx < per1) {
      const int* rec = a.task_map + static_cast<long>(kTaskStride) * (1 + static_cast<long>(bin) * per1 + idx);
      hh = rec[0];
      bb = rec[1];
      nt = rec[6];
    }
    const uint64_t bad = __ballot(hh < 0 || bb < 0);
    const int first = bad ? __builtin_ctzll(bad) : 64;
    int mx = lane < first ? nt : 0;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) mx = max(mx, __shfl_xor(mx, o, 64));
    max_ntile = max(max_ntile, mx);
    ntasks += first;
    if (first < 64) break;
  }
  ntasks = __builtin_amdgcn_readfirstlane(ntasks);
  max_ntile = __builtin_amdgcn_readfirstlane(max_ntile);

  // One task, two ways to run it.  TEAM: the 4 waves take tiles w, w+4, ... and merge through LDS
  // (long tasks: KV streaming).  SOLO: one wave runs the whole task alone and finishes it without
  // any workgroup barrier, so 4 short tasks of a bin run side by side - a bin packed with short
  // requests (mixed-length batches) would otherwise serialise ~5 us of latency chain per task.
  auto run_task = [&](cint_ptr task_ptr, auto solo_c) {
    constexpr bool kSolo = decltype(solo_c)::value;
    const int t_first = kSolo ? 0 : wave, t_step = kSolo ? 1 : kWaves;
    const int h = __builtin_amdgcn_readfirstlane(task_ptr[0]);
    const int b = __builtin_amdgcn_readfirstlane(task_ptr[1]);
    const int ichunk = __builtin_amdgcn_readfirstlane(task_ptr[2]);
    const int iseq_start = __builtin_amdgcn_readfirstlane(task_ptr[3]);
    const int num_seqkv = __builtin_amdgcn_readfirstlane(task_ptr[4]);
    const int num_seqkvcache = __builtin_amdgcn_readfirstlane(task_ptr[5]);
    const int ntile = __builtin_amdgcn_readfirstlane(task_ptr[6]);
    const int ntile_full = __builtin_amdgcn_readfirstlane(task_ptr[7]);

    // ---- Q fragments (B operand of S^T = K Q^T) and per-row score scales -----------------------
    // bf16: lane (n, g) holds dims (4j+g)*8..+7 for k-step j; fp8: 16-byte chunks g and g+4.
    u32x4 qf[kNB][kKC];
    float row_scale[kNB];
    int row_sq[kNB];
#pragma unroll
    for (int nb = 0; nb < kNB; ++nb) {
      const int row = nb * 16 + n;
      const bool ok = row < rows_valid;
      const int sq = row >> a.g_shift;
      const int hq = (h << a.g_shift) + (row & (G - 1));
      row_sq[nb] = sq;
      const long qoff = (static_cast<long>(b * a.num_seq_q + sq) * a.ldq + hq * 128) * kEB;
#pragma unroll
      for (int c = 0; c < kKC; ++c) {
        qf[nb][c] = u32x4{0u, 0u, 0u, 0u};
        if (ok) qf[nb][c] = ld16(qbase + qoff + (kFp8 ? (g + 4 * c) * 16 : (4 * c + g) * 16));
      }
      row_scale[nb] = a.scale_log2;
      if constexpr (kFp8) {
        float qs = ok ? a.qscale[static_cast<long>(b * a.num_seq_q + sq) * a.qscale_stride + hq] : 0.f;
        if constexpr (kQuant == 1) qs *= a.kscale[0];
        row_scale[nb] *= qs;
      }
    }
    float out_scale = 1.0f;  // fp8: vscale / 256
    if constexpr (kFp8) out_scale = (kQuant == 1 ? a.vscale[0] : a.vscale[h]) * (1.0f / 256.0f);

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
    // `nrec` = 0 turns the loads of a non-existent next tile into no-ops (out-of-range buffer
    // reads return 0 without touching memory) - the loop body stays branch-free, which keeps
    // hipcc's counted vmcnt waits intact.
    u32x4 kf[4][kKC];                                   // [token block][16-byte chunk]
    u32x4 vf16[kFp8 ? 1 : 2][kFp8 ? 1 : 8];             // bf16: [k step of PV][token slot]
    u32x2 vf8[kFp8 ? 2 : 1][kFp8 ? 8 : 1];              // fp8
    f32x4 ksc[(kFp8 && kQuant == 0) ? 4 : 1];           // per-token K scales of this lane's tokens
    constexpr int kChunks = kFp8 ? 8 : 16;      // 16-byte chunks per K row
    constexpr int kRowsPerLd = 64 / kChunks;    // rows covered by one wave-wide load
    const int k_chunk = lane % kChunks, k_rsub = lane / kChunks;
    const int k_voff = k_rsub * static_cast<int>(a.k_token_stride) * kEB + k_chunk * 16;
    const int k_ld_bytes = kRowsPerLd * static_cast<int>(a.k_token_stride) * kEB;
    const int v_voff = g * 4 * static_cast<int>(a.v_token_stride) * kEB + n * 8 * kEB;
    const int v_tok_bytes = static_cast<int>(a.v_token_stride) * kEB;
    auto load_k = [&](const int (&pid)[4], const int (&inpage)[4], unsigned nrec) {
#pragma unroll
      for (int tb = 0; tb < 4; ++tb) {
        const auto rs = make_rsrc(kbase + (pid[tb] * a.k_block_stride + inpage[tb] * a.k_token_stride +
                                           h * a.k_head_stride) * kEB, nrec);
#pragma unroll
        for (int c = 0; c < kKC; ++c) kf[tb][c] = buf_ld16<kAux>(rs, k_voff, c * k_ld_bytes);
        if constexpr (kFp8 && kQuant == 0) {
          // scales of tokens inpage+g*4 .. +3: tail row (tok >> 5), byte (tok & 31) * 4
          const auto rk = make_rsrc(reinterpret_cast<const uint8_t*>(a.kscale) + pid[tb] * a.ks_block_stride +
                                    (inpage[tb] >> 5) * a.ks_row_stride + h * a.ks_head_stride +
                                    (inpage[tb] & 31) * 4, nrec);
          const u32x4 raw = buf_ld16<0>(rk, g * 16, 0);
          ksc[tb] = f32x4{__uint_as_float(raw[0]), __uint_as_float(raw[1]), __uint_as_float(raw[2]),
                          __uint_as_float(raw[3])};
        }
      }
    };
    auto load_v = [&](const int (&pid)[4], const int (&inpage)[4], unsigned nrec) {
#pragma unroll
      for (int ks = 0; ks < 2; ++ks)
#pragma unroll
        for (int hb = 0; hb < 2; ++hb) {
          const int tb = 2 * ks + hb;
          const auto rs = make_rsrc(vbase + (pid[tb] * a.v_block_stride + inpage[tb] * a.v_token_stride +
                                             h * a.v_head_stride) * kEB, nrec);
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            if constexpr (kFp8)
              vf8[ks][hb * 4 + r] = buf_ld8<kAux>(rs, v_voff, r * v_tok_bytes);
            else
              vf16[ks][hb * 4 + r] = buf_ld16<kAux>(rs, v_voff, r * v_tok_bytes);
          }
        }
    };

    f32x4 o[kNB][8];
    float m_run[kNB], l_run[kNB];  // l: per-lane partial of the row sum (4 lanes share a row)
#pragma unroll
    for (int nb = 0; nb < kNB; ++nb) {
#pragma unroll
      for (int jj = 0; jj < 8; ++jj) o[nb][jj] = f32x4{0.f, 0.f, 0.f, 0.f};
      m_run[nb] = kNegInf;
      l_run[nb] = 0.f;
    }

    int t = t_first;
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
    for (; t < ntile; t += t_step) {
      const unsigned nrec = t + t_step < ntile ? 0xffffffffu : 0u;
      tile_pages(t + t_step < ntile ? t + t_step : ntile - 1, pid, inpage);

      // ---- S^T = K Q^T ------------------------------------------------------------------
      f32x4 s[kNB][4];
#pragma unroll
      for (int tb = 0; tb < 4; ++tb) {
        // full-row layout -> MFMA A-operand layout through the wave's private LDS tile
        uint8_t* kt = s_kt[wave][tb & 1];
#pragma unroll
        for (int c = 0; c < kKC; ++c)
          *reinterpret_cast<u32x4*>(kt + (c * kRowsPerLd + k_rsub) * kKRow + k_chunk * 16) = kf[tb][c];
        u32x4 ka[kKC];
#pragma unroll
        for (int c = 0; c < kKC; ++c)
          ka[c] = *reinterpret_cast<const u32x4*>(kt + n * kKRow + (kFp8 ? (g + 4 * c) : (4 * c + g)) * 16);
#pragma unroll
        for (int nb = 0; nb < kNB; ++nb) {
          f32x4 acc = f32x4{0.f, 0.f, 0.f, 0.f};
          if constexpr (kFp8) {
#pragma unroll
            for (int c = 0; c < 2; ++c) {
              acc = __builtin_amdgcn_mfma_f32_16x16x32_fp8_fp8(
                  pack64(ka[c][0], ka[c][1]), pack64(qf[nb][c][0], qf[nb][c][1]), acc, 0, 0, 0);
              acc = __builtin_amdgcn_mfma_f32_16x16x32_fp8_fp8(
                  pack64(ka[c][2], ka[c][3]), pack64(qf[nb][c][2], qf[nb][c][3]), acc, 0, 0, 0);
            }
          } else {
#pragma unroll
            for (int j = 0; j < 4; ++j) {
              Frag16 kfr, qa;
              kfr.u = ka[j];
              qa.u = qf[nb][j];
              acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kfr.b, qa.b, acc, 0, 0, 0);
            }
          }
          s[nb][tb] = acc;
        }
      }
      f32x4 ksc_cur[(kFp8 && kQuant == 0) ? 4 : 1];
      if constexpr (kFp8 && kQuant == 0) {
#pragma unroll
        for (int tb = 0; tb < 4; ++tb) ksc_cur[tb] = ksc[tb];
      }
      load_k(pid, inpage, nrec);

      // ---- online softmax in base 2 (per 16-row q block) ----------------------------------------
      uint32_t pf[kNB][2][kFp8 ? 2 : 4];  // P^T operand of the PV MFMAs, per k-step
      const bool masked = t >= ntile_full;  // tile holds masked keys (request tail / causal rows)
#pragma unroll
      for (int nb = 0; nb < kNB; ++nb) {
        float mt = kNegInf;
        const int lim = min(num_seqkv - 1, num_seqkvcache + row_sq[nb]);
#pragma unroll
        for (int tb = 0; tb < 4; ++tb)
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            float x = s[nb][tb][r] * row_scale[nb];
            if constexpr (kFp8 && kQuant == 0) x *= ksc_cur[tb][r];
            if (masked) {
              const int tok = t * 64 + tb * 16 + g * 4 + r;
              x = tok <= lim ? x : kNegInf;
            }
            s[nb][tb][r] = x;
            mt = fmaxf(mt, x);
          }
        mt = fmaxf(mt, __shfl_xor(mt, 16, 64));
        mt = fmaxf(mt, __shfl_xor(mt, 32, 64));
        const float m_new = fmaxf(m_run[nb], mt);
        const float m_use = m_new == kNegInf ? 0.f : m_new;  // all-masked rows: p = exp2(-inf) = 0
        const float alpha = __builtin_amdgcn_exp2f(m_run[nb] - m_use);
        m_run[nb] = m_new;
        float psum = 0.f;
#pragma unroll
        for (int tb = 0; tb < 4; ++tb) {
          float p[4];
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            p[r] = __builtin_amdgcn_exp2f(s[nb][tb][r] - m_use);
            psum += p[r];
          }
          if constexpr (kFp8) {
            pf[nb][tb >> 1][tb & 1] =
                cvt_4xe4m3(p[0] * 256.f, p[1] * 256.f, p[2] * 256.f, p[3] * 256.f);
          } else {
            pf[nb][tb >> 1][(tb & 1) * 2] = pack_bf16x2(p[0], p[1]);
            pf[nb][tb >> 1][(tb & 1) * 2 + 1] = pack_bf16x2(p[2], p[3]);
          }
        }
        l_run[nb] = l_run[nb] * alpha + psum;
#pragma unroll
        for (int jj = 0; jj < 8; ++jj) o[nb][jj] *= alpha;
      }

      // ---- O^T += V^T P^T (private 8x8 transposes feed the A operand) -----------------------
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) {
        if constexpr (kFp8) {
          // 8 tokens x 8 dims bytes -> 8 dims x 8 tokens.  stage 1 interleaves token pairs,
          // stage 2 gathers 4 tokens of one dim per dword.
          uint32_t st[2][2][4];  // [dims half][token quad][pair-interleaved dword]
#pragma unroll
          for (int dh = 0; dh < 2; ++dh)
#pragma unroll
            for (int tq = 0; tq < 2; ++tq) {
              const uint32_t r0 = vf8[ks][tq * 4 + 0][dh], r1 = vf8[ks][tq * 4 + 1][dh];
              const uint32_t r2 = vf8[ks][tq * 4 + 2][dh], r3 = vf8[ks][tq * 4 + 3][dh];
              st[dh][tq][0] = __builtin_amdgcn_perm(r1, r0, 0x05010400u);  // d0:t0 t1, d1:t0 t1
              st[dh][tq][1] = __builtin_amdgcn_perm(r1, r0, 0x07030602u);  // d2, d3
              st[dh][tq][2] = __builtin_amdgcn_perm(r3, r2, 0x05010400u);  // d0:t2 t3, d1:t2 t3
              st[dh][tq][3] = __builtin_amdgcn_perm(r3, r2, 0x07030602u);
            }
#pragma unroll
          for (int jj = 0; jj < 8; ++jj) {
            const int dh = jj >> 2, dp = (jj >> 1) & 1;  // dims half, dim pair inside the half
            const uint32_t sel = (jj & 1) ? 0x07060302u : 0x05040100u;
            const uint32_t lo = __builtin_amdgcn_perm(st[dh][0][2 + dp], st[dh][0][dp], sel);
            const uint32_t hi = __builtin_amdgcn_perm(st[dh][1][2 + dp], st[dh][1][dp], sel);
#pragma unroll
            for (int nb = 0; nb < kNB; ++nb)
              o[nb][jj] = __builtin_amdgcn_mfma_f32_16x16x32_fp8_fp8(
                  pack64(lo, hi), pack64(pf[nb][ks][0], pf[nb][ks][1]), o[nb][jj], 0, 0, 0);
            if (jj & 1) __builtin_amdgcn_sched_barrier(0);
          }
        } else {
#pragma unroll
          for (int jj = 0; jj < 8; ++jj) {
            Frag16 vt;
#pragma unroll
            for (int p2 = 0; p2 < 4; ++p2) {
              const uint32_t lo = vf16[ks][2 * p2][jj >> 1], hi = vf16[ks][2 * p2 + 1][jj >> 1];
              vt.u[p2] = __builtin_amdgcn_perm(hi, lo, (jj & 1) ? 0x07060302u : 0x05040100u);
            }
#pragma unroll
            for (int nb = 0; nb < kNB; ++nb) {
              Frag16 pa;
              pa.u = u32x4{pf[nb][ks][0], pf[nb][ks][1], pf[nb][ks][2], pf[nb][ks][3]};
              o[nb][jj] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(vt.b, pa.b, o[nb][jj], 0, 0, 0);
            }
            if (jj & 1) __builtin_amdgcn_sched_barrier(0);  // keep the transposes next to their MFMAs
          }
        }
      }
      load_v(pid, inpage, nrec);
    }

    const int nchunks = chunk_tab[h * a.num_batch + b];
    // final scaling + store of 8 output dims of one q row: bf16 y when the request is whole, fp32
    // partial + lse otherwise (slot 1 of the bin for a request's first chunk, slot 0 for a later one)
    auto emit = [&](int nb, int row16, int c8, float M, float L, float (&acc)[8]) {
      const int row = nb * 16 + row16;
      const float inv = (L > 0.f ? 1.0f / L : 0.f) * out_scale;
#pragma unroll
      for (int i = 0; i < 8; ++i) acc[i] *= inv;
      if (row >= rows_valid) return;
      if (nchunks == 1) {
        const int rs = row >> a.g_shift;
        uint16_t* dst = a.y + static_cast<long>(b * a.num_seq_q + rs) * a.ldy +
                        ((h << a.g_shift) + (row & (G - 1))) * 128 + c8 * 8;
        u32x4 pk;
#pragma unroll
        for (int i = 0; i < 4; ++i) pk[i] = pack_bf16x2(acc[2 * i], acc[2 * i + 1]);
        st16(dst, pk);
      } else {
        const int slot = bin * 2 + (ichunk == 0 ? 1 : 0);
        const int prow = slot * kNB * 16 + row;
        const int off = (prow * 128 + c8 * 8) * 4;
        __builtin_amdgcn_raw_buffer_store_b128(u32x4{__float_as_uint(acc[0]), __float_as_uint(acc[1]), __float_as_uint(acc[2]),
                                                     __float_as_uint(acc[3])}, part_rs, off, 0, 16);
        __builtin_amdgcn_raw_buffer_store_b128(u32x4{__float_as_uint(acc[4]), __float_as_uint(acc[5]), __float_as_uint(acc[6]),
                                                     __float_as_uint(acc[7])}, part_rs, off + 16, 0, 16);
        if (c8 == 0)
          __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(L > 0.f ? M + __builtin_amdgcn_logf(L) : kNegInf), lse_rs, prow * 4, 0, 16);
      }
    };
#pragma unroll
    for (int nb = 0; nb < kNB; ++nb) {
      float l = l_run[nb];
      l += __shfl_xor(l, 16, 64);
      l += __shfl_xor(l, 32, 64);
      if (g == 0) {
        s_m[wave][n] = m_run[nb];
        s_l[wave][n] = l;
      }
#pragma unroll
      for (int jj = 0; jj < 8; ++jj)
#pragma unroll
        for (int r = 0; r < 4; ++r) s_o[wave][n][8 * (g * 4 + r) + jj] = o[nb][jj][r];
      if constexpr (kSolo) {
        // ---- the wave finishes its own task: re-read its tile row-major (LDS ops of a wave are in order)
#pragma unroll 1
        for (int it = 0; it < 4; ++it) {
          const int row16 = it * 4 + (lane >> 4), c8 = lane & 15;
          const f32x4 x0 = *reinterpret_cast<const f32x4*>(&s_o[wave][row16][c8 * 8]);
          const f32x4 x1 = *reinterpret_cast<const f32x4*>(&s_o[wave][row16][c8 * 8 + 4]);
          float acc[8] = {x0[0], x0[1], x0[2], x0[3], x1[0], x1[1], x1[2], x1[3]};
          emit(nb, row16, c8, s_m[wave][row16], s_l[wave][row16], acc);
        }
        __builtin_amdgcn_s_waitcnt(0x0F70);  // vmcnt(0): see below
      } else {
        // ---- merge the 4 waves of the workgroup, one 16-row q block at a time -----------------------
        __syncthreads();
        {
          const int row16 = tid >> 4;  // q row inside the block
          const int c8 = tid & 15;     // chunk of 8 dims
          float mw[kWaves], M = kNegInf;
#pragma unroll
          for (int w = 0; w < kWaves; ++w) {
            mw[w] = s_m[w][row16];
            M = fmaxf(M, mw[w]);
          }
          const float Mu = M == kNegInf ? 0.f : M;
          float L = 0.f, acc[8];
#pragma unroll
          for (int i = 0; i < 8; ++i) acc[i] = 0.f;
#pragma unroll
          for (int w = 0; w < kWaves; ++w) {
            const float wgt = __builtin_amdgcn_exp2f(mw[w] - Mu);
            L += wgt * s_l[w][row16];
            const f32x4 x0 = *reinterpret_cast<const f32x4*>(&s_o[w][row16][c8 * 8]);
            const f32x4 x1 = *reinterpret_cast<const f32x4*>(&s_o[w][row16][c8 * 8 + 4]);
#pragma unroll
            for (int i = 0; i < 4; ++i) {
              acc[i] = fmaf(wgt, x0[i], acc[i]);
              acc[4 + i] = fmaf(wgt, x1[i], acc[4 + i]);
            }
          }
          emit(nb, row16, c8, M, L, acc);
        }
        // gfx950 counts stores in vmcnt too; a store still pending when the next task starts makes
        // hipcc treat the counter as out-of-order and degrade every wait in the tile loop to
        // vmcnt(0).  Retire the epilogue stores here, where nothing else is in flight.
        __builtin_amdgcn_s_waitcnt(0x0F70);  // vmcnt(0)
        __syncthreads();
      }
    }
    if (nchunks > 1 && a.arrive) {
      // ---- split request, merged inside the launch (the reference's static path does the same: the last CTA of a
      // request reduces, static_splitk_kernels.cuh:362-377; math of splitk_combine_kernels.cuh:140-322) ----------------
      // This chunk's partial has reached memory (sc1 stores, vmcnt(0) above - team: every thread's, behind the
      // barrier).  One atomic add on the request's counter (zero on first use: the contract of
      // hpc_attention_decode_workspace_zero_bytes(); the last arriver puts the zero back); the chunk that arrives LAST
      // folds all of them - chunk c of a request lives in bin (bin - ichunk) + c, slot 1 for c == 0 and slot 0 otherwise -
      // with 8-16 chunks' loads in flight per thread and a running maximum (one pass), and writes y.
      int* cnt = a.arrive + h * a.num_batch + b;
      int ticket = 0;
      if constexpr (kSolo) {
        if (lane == 0) ticket = __hip_atomic_fetch_add(cnt, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) + 1;
        ticket = __builtin_amdgcn_readfirstlane(ticket);
      } else {
        if (tid == 0) s_ticket = __hip_atomic_fetch_add(cnt, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) + 1;
        __syncthreads();
        ticket = s_ticket;
      }
#ifdef HPC_DEV
      if ((kSolo ? lane == 0 : tid == 0) && ticket > nchunks) atomicAdd(&g_ticket_overruns, 1);  // the counter was not zero on entry
#endif
      if (ticket == nchunks) {
        const int fb = bin - ichunk;
        auto merge_row = [&](int nb, int row16, int c8) {
          const int row = nb * 16 + row16;
          if (row >= rows_valid) return;
          float Mr = kNegInf, W = 0.f, acc[8];
#pragma unroll
          for (int i = 0; i < 8; ++i) acc[i] = 0.f;
          // chunks in flight per thread: 16 with one q block (144 registers: the tile loop's are dead here), 8 with more
          // (the two- and three-block kernels sit at 330-506 registers) - a request cut into many chunks (one long request
          // among short ones on a single kv head: 171 chunks of a 64k request) is merged in chunks / kU round trips
          constexpr int kU = kNB == 1 ? 16 : 8;
          for (int c0 = 0; c0 < nchunks; c0 += kU) {
            float l8[kU];
            u32x4 x0[kU], x1[kU];
#pragma unroll
            for (int u = 0; u < kU; ++u) {
              const int c = c0 + u < nchunks ? c0 + u : nchunks - 1;
              const int prow = ((fb + c) * 2 + (c == 0 ? 1 : 0)) * kNB * 16 + row;
              l8[u] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(lse_rs, prow * 4, 0, 16));
              const int off = (prow * 128 + c8 * 8) * 4;
              x0[u] = __builtin_amdgcn_raw_buffer_load_b128(part_rs, off, 0, 16);
              x1[u] = __builtin_amdgcn_raw_buffer_load_b128(part_rs, off + 16, 0, 16);
            }
            float mb = Mr;
#pragma unroll
            for (int u = 0; u < kU; ++u) mb = fmaxf(mb, c0 + u < nchunks ? l8[u] : kNegInf);
            const float mu = mb == kNegInf ? 0.f : mb;
            const float sc_old = __builtin_amdgcn_exp2f(Mr - mu);  // Mr = -inf: 0 (nothing folded yet)
            Mr = mb;
            W *= sc_old;
#pragma unroll
            for (int i = 0; i < 8; ++i) acc[i] *= sc_old;
#pragma unroll
            for (int u = 0; u < kU; ++u) {
              const float wgt = c0 + u < nchunks ? __builtin_amdgcn_exp2f(l8[u] - mu) : 0.f;
              W += wgt;
#pragma unroll
              for (int i = 0; i < 4; ++i) {
                acc[i] = fmaf(wgt, __uint_as_float(x0[u][i]), acc[i]);
                acc[4 + i] = fmaf(wgt, __uint_as_float(x1[u][i]), acc[4 + i]);
              }
            }
          }
          const float inv = W > 0.f ? 1.0f / W : 0.f;
          const int rs = row >> a.g_shift;
          uint16_t* dst = a.y + static_cast<long>(b * a.num_seq_q + rs) * a.ldy + ((h << a.g_shift) + (row & (G - 1))) * 128 + c8 * 8;
          u32x4 pk;
#pragma unroll
          for (int i = 0; i < 4; ++i) pk[i] = pack_bf16x2(acc[2 * i] * inv, acc[2 * i + 1] * inv);
          st16(dst, pk);
        };
#pragma unroll 1
        for (int nb = 0; nb < kNB; ++nb) {
          if constexpr (kSolo) {
#pragma unroll 1
            for (int it = 0; it < 4; ++it) merge_row(nb, it * 4 + (lane >> 4), lane & 15);
          } else {
            merge_row(nb, tid >> 4, tid & 15);
          }
        }
        if (kSolo ? lane == 0 : tid == 0) __hip_atomic_store(cnt, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);  // ready for the next call
        __builtin_amdgcn_s_waitcnt(0x0F70);  // retire the stores before the next task's loads (vmcnt counts both)
      }
      if constexpr (!kSolo) __syncthreads();  // s_ticket is free again
    } else if ((kSolo ? lane == 0 : tid == 0) && nchunks > 1 && ichunk == 0) {
      a.first_bin[h * a.num_batch + b] = bin;
      __builtin_amdgcn_s_waitcnt(0x0F70);
    }
  };

  constexpr int kSoloMaxTiles = 4;  // up to here a lone wave is no slower than the team (1 tile per wave)
  // The solo form exists for one and two 16-row q blocks only.  With three (bf16, num_seq_q = 5: 40 q rows) the
  // kernel sits at the 512-register limit and the build that held BOTH forms faulted in the solo path only
  // (round 3, tools/dbg_sq5.py: "Memory access fault ... on address 0xaa000" = a null base + offset, team path of
  // the same binary correct) when the bf16 packs compiled to v_cvt_pk_bf16_f32 (510 registers, no scratch), and
  // ran with the integer pack sequence (512 registers + 76 bytes of scratch): an allocation-dependent failure of
  // the solo instantiation at the register limit, not an out-of-bounds access of the algorithm.  Five-token
  // speculative steps of batches full of <= 256-token requests are not a case worth a 512-register second body.
  if constexpr (kNB < 3) {
    const bool solo = ntasks >= 2 && max_ntile <= kSoloMaxTiles && a.solo_ok;
    if (solo) {
      for (int it = wave; it < ntasks; it += kWaves) run_task(task_ptr + it * kTaskStride, std::true_type{});
      return;
    }
  }
  for (int it = 0; it < ntasks; ++it) run_task(task_ptr + it * kTaskStride, std::false_type{});
}

// ---- split-KV combine: y = sum_c 2^(lse_c - max) O_c / sum_c 2^(lse_c - max) ----------------------------
// (reference splitk_combine_kernels.cuh:140-322). One workgroup per (kv head, request, q block);
// chunk c of a request lives in bin first_bin + c, slot 1 for c == 0 and slot 0 otherwise.
__global__ __launch_bounds__(kThreads) void decode_combine_kernel(const Args a, int num_nb) {
  const int hb = blockIdx.x / num_nb, nb = blockIdx.x % num_nb;
  const cint_ptr tmap = as_const(a.task_map);
  const int per1 = tmap[0];
  const int num_bins = tmap[1];
  const int nchunks = (tmap + sched::chunk_table_off(per1 - 1, num_bins))[hb];
  if (nchunks <= 1) return;
  const int h = hb / a.num_batch, b = hb % a.num_batch;
  const int fb = a.first_bin[hb];
  const int tid = threadIdx.x;
  const int row = nb * 16 + (tid >> 4), c8 = tid & 15;
  const int rows_valid = a.num_seq_q << a.g_shift;
  if (row >= rows_valid) return;
  const int G = 1 << a.g_shift;
  const long rows_per_slot = num_nb * 16;

  // chunks in batches with all loads of a batch in flight (a serial load->use chain costs one L2 /
  // HBM round trip per chunk - 19 us for a 32-chunk request)
  auto slot_of = [&](int c) { return static_cast<long>(fb + c) * 2 + (c == 0 ? 1 : 0); };
  float M = kNegInf;
  for (int c0 = 0; c0 < nchunks; c0 += 8) {
    float l8[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const int c = c0 + u < nchunks ? c0 + u : nchunks - 1;
      l8[u] = a.part_lse[slot_of(c) * rows_per_slot + row];
    }
#pragma unroll
    for (int u = 0; u < 8; ++u) M = fmaxf(M, l8[u]);
  }
  const float Mu = M == kNegInf ? 0.f : M;
  float W = 0.f, acc[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) acc[i] = 0.f;
  for (int c0 = 0; c0 < nchunks; c0 += 4) {
    float l4[4];
    f32x4 x0[4], x1[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int c = c0 + u < nchunks ? c0 + u : nchunks - 1;
      const long slot = slot_of(c);
      l4[u] = a.part_lse[slot * rows_per_slot + row];
      const float* po = a.part_o + (slot * rows_per_slot + row) * 128 + c8 * 8;
      x0[u] = *reinterpret_cast<const f32x4*>(po);
      x1[u] = *reinterpret_cast<const f32x4*>(po + 4);
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const float wgt = c0 + u < nchunks ? __builtin_amdgcn_exp2f(l4[u] - Mu) : 0.f;
      W += wgt;
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        acc[i] = fmaf(wgt, x0[u][i], acc[i]);
        acc[4 + i] = fmaf(wgt, x1[u][i], acc[4 + i]);
      }
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

template <bool kFp8, int kQuant>
int launch(const Args& a, int num_bins, int num_nb, hipStream_t stream) {
  const bool temporal = hpc_dev_tuning_get(0) == 1;
#define HPC_DECODE_LAUNCH(NB)                                                                   \
  if (temporal)                                                                                 \
    decode_kernel<kFp8, kQuant, NB, 0><<<num_bins, kThreads, 0, stream>>>(a);                   \
  else                                                                                          \
    decode_kernel<kFp8, kQuant, NB, 2><<<num_bins, kThreads, 0, stream>>>(a)
  if (num_nb == 1) {
    HPC_DECODE_LAUNCH(1);
  } else if (num_nb == 2) {
    HPC_DECODE_LAUNCH(2);
  } else if (num_nb == 3) {
    if constexpr (kFp8) {
      return HPC_ERR_UNSUPPORTED;  // fp8 supports num_seq_q <= 4
    } else {
      HPC_DECODE_LAUNCH(3);
    }
  } else {
    return HPC_ERR_UNSUPPORTED;
  }
#undef HPC_DECODE_LAUNCH
  HPC_CHECK_LAUNCH();
  if (!a.arrive) {  // more (kv head, request) pairs than arrival counters: the chunks are merged by a second launch
    decode_combine_kernel<<<a.num_batch * a.num_head_kv * num_nb, kThreads, 0, stream>>>(a, num_nb);
    HPC_CHECK_LAUNCH();
  }
  return HPC_OK;
}

struct Common {
  int num_nb;
  int code;
};

inline Common fill_common(Args& a, void* y_ptr, void* workspace, const int* task_map_ptr,
                          const void* q_ptr, const void* kcache_ptr, const void* vcache_ptr,
                          const int* block_ids_ptr, int num_bins, int num_batch, int num_seq_q,
                          int num_head_q, int num_head_kv, int num_dim_qk, int num_dim_v,
                          int block_size, int num_seq_max_blocks, int ldY, int ldQ, int64_t kbs,
                          int64_t kts, int64_t khs, int64_t vbs, int64_t vts, int64_t vhs, int align) {
  Common c{0, HPC_OK};
  if (!y_ptr || !workspace || !task_map_ptr || !q_ptr || !kcache_ptr || !vcache_ptr || !block_ids_ptr) {
    c.code = HPC_ERR_INVALID;
    return c;
  }
  if (num_dim_qk != 128 || num_dim_v != 128 ||
      (block_size != 16 && block_size != 32 && block_size != 64)) {
    c.code = HPC_ERR_UNSUPPORTED;
    return c;
  }
  if (num_head_kv <= 0 || num_head_q % num_head_kv || num_batch <= 0 || num_bins <= 0) {
    c.code = HPC_ERR_INVALID;
    return c;
  }
  const int group = num_head_q / num_head_kv;
  if ((group != 4 && group != 8) || num_seq_q < 1 || num_seq_q > 5) {
    c.code = HPC_ERR_UNSUPPORTED;
    return c;
  }
  const int m = align - 1;  // elements per 16 bytes - 1
  if ((ldQ & m) || (ldY & 7) || (kts & m) || (vts & m) || (khs & m) || (vhs & m) || (kbs & m) || (vbs & m)) {
    c.code = HPC_ERR_UNSUPPORTED;  // 16-byte vector accesses
    return c;
  }
  c.num_nb = (num_seq_q * group + 15) / 16;
  a.q = q_ptr;
  a.kcache = kcache_ptr;
  a.vcache = vcache_ptr;
  a.block_ids = block_ids_ptr;
  a.task_map = task_map_ptr;
  a.y = static_cast<uint16_t*>(y_ptr);
  // the workspace starts with the arrival counters of split requests (zero-once region, shared with the second
  // generation: one of the two runs per call); development key 33 = 1: the round-1 form (combine kernel)
  a.arrive = static_cast<int64_t>(num_batch) * num_head_kv * 4 <= hpc::decode2::kCounterBytes && hpc_dev_tuning_get(33) != 1
                 ? static_cast<int*>(workspace) : nullptr;
  char* ws = static_cast<char*>(workspace) + hpc::decode2::kCounterBytes;
  a.part_o = reinterpret_cast<float*>(ws);
  ws += static_cast<int64_t>(num_bins) * 2 * 16 * c.num_nb * 128 * 4;
  a.part_lse = reinterpret_cast<float*>(ws);
  ws += static_cast<int64_t>(num_bins) * 2 * 16 * c.num_nb * 4;
  a.first_bin = reinterpret_cast<int*>(ws);
  a.qscale = a.kscale = a.vscale = nullptr;
  a.num_batch = num_batch;
  a.num_seq_q = num_seq_q;
  a.num_head_kv = num_head_kv;
  a.g_shift = group == 8 ? 3 : 2;
  a.page_shift = block_size == 64 ? 6 : (block_size == 32 ? 5 : 4);
  a.max_blocks = num_seq_max_blocks;
  a.solo_ok = hpc_dev_tuning_get(5) != 1;
  a.ldq = ldQ;
  a.ldy = ldY;
  a.qscale_stride = 0;
  a.k_block_stride = kbs;
  a.k_token_stride = kts;
  a.k_head_stride = khs;
  a.v_block_stride = vbs;
  a.v_token_stride = vts;
  a.v_head_stride = vhs;
  a.ks_block_stride = a.ks_row_stride = a.ks_head_stride = 0;
  a.scale_log2 = 0.08838834764831845f * 1.4426950408889634f;  // 1/sqrt(128) * log2(e)
  return c;
}

}  // namespace decode
}  // namespace hpc

namespace {
int64_t v1_workspace_bytes(int num_bins, int num_batch, int num_head_kv, int num_seq_q, int heads_per_group) {
  const int64_t rows = (static_cast<int64_t>(num_seq_q) * heads_per_group + 15) / 16 * 16;
  const int64_t part_o = static_cast<int64_t>(num_bins) * 2 * rows * 128 * 4;
  const int64_t part_lse = static_cast<int64_t>(num_bins) * 2 * rows * 4;
  const int64_t first_bin = static_cast<int64_t>(num_batch) * num_head_kv * 4;
  return part_o + part_lse + ((first_bin + 15) / 16) * 16;
}
}  // namespace

// Scratch of one decode call: [arrival counters of the second-generation kernel: hpc_attention_decode_workspace_zero_bytes()
// bytes that must be zero the first time the buffer is used and are left zero by every call] [the first-generation
// kernel's region: 2 partial slots per bin, first-bin table] [the second-generation FP8 / NHD kernel's partial slots:
// 2 per workgroup x 2 heads; its grid never exceeds num_bins workgroups].
extern "C" int64_t hpc_attention_decode_workspace_bytes(int num_bins, int num_batch, int num_head_kv,
                                                        int num_seq_q, int heads_per_group) {
  if (num_bins <= 0 || num_batch <= 0 || num_head_kv <= 0 || num_seq_q <= 0 || heads_per_group <= 0)
    return HPC_ERR_INVALID;
  return hpc::decode2::kCounterBytes + v1_workspace_bytes(num_bins, num_batch, num_head_kv, num_seq_q, heads_per_group) +
         hpc::decode2::workspace_bytes(num_bins);
}
extern "C" int64_t hpc_attention_decode_workspace_zero_bytes(void) { return hpc::decode2::kCounterBytes; }

// development (tools/prof_decode.py): device buffer [workgroups][4 waves][12] uint64 that the profiling build of the
// second-generation kernel fills with per-wave s_memtime sums; null = the shipped kernel
#ifdef HPC_DEV
// development build: number of arrivals (either kernel generation) that drew a ticket above their request's chunk count
// since the last reset - a stale arrival counter (a caller-owned workspace that was never zeroed makes the
// split requests' rows of y silently stay unwritten).  Synchronises the device; < 0 on error.
extern "C" int hpc_dev_decode_ticket_overruns(int reset) {
  if (hipDeviceSynchronize() != hipSuccess) return -1;
  int v = 0;
  if (hipMemcpyFromSymbol(&v, HIP_SYMBOL(hpc::decode::g_ticket_overruns), sizeof(int)) != hipSuccess) return -1;
  const int zero = 0;
  if (reset && hipMemcpyToSymbol(HIP_SYMBOL(hpc::decode::g_ticket_overruns), &zero, sizeof(int)) != hipSuccess) return -1;
  const int v2 = hpc::decode2::ticket_overruns(reset != 0);
  return v2 < 0 ? -1 : v + v2;
}
static void* g_decode_prof = nullptr;
extern "C" int hpc_dev_decode_prof_buffer(void* p) {
  g_decode_prof = p;
  return 0;
}
#else
static constexpr void* g_decode_prof = nullptr;  // production build: the profiling instantiations are not emitted
#endif

// Second generation (attention_decode_v2.hip: head pairs per load, deep prefetch, in-kernel plan and merge) when the
// layout allows it.  Returns HPC_OK when it launched, 1 when the call is not its case (the caller goes on to the
// first generation), a negative code on a launch error.  Strides in `b` are BYTES.
static int try_second_generation(hpc::decode2::Args& b, void* workspace, int num_bins, int num_batch, int num_seq_q,
                                 int num_head_q, int num_head_kv, int block_size, int64_t k_head_stride_bytes,
                                 int64_t v_head_stride_bytes, hipStream_t stream) {
  b.part_o = b.part_lse = nullptr;
  b.arrive = nullptr;
  b.dev_nomem = hpc_dev_tuning_get(15);
  b.min_range_cost = hpc_dev_tuning_get(20) > 0 ? hpc_dev_tuning_get(20) : 8;  // development key 20 overrides (15 x 64 + 1 x 16k tokens: 87 us without a floor, 47 / 49 / 61 / 105 us at 8 / 16 / 32 / 64)
  b.prof = g_decode_prof;
  if (hpc_dev_tuning_get(12) == 1) return 1;  // development key 12 = 1: first generation only
  const int mode = hpc::decode2::mode_of(b, num_head_q, block_size, k_head_stride_bytes, v_head_stride_bytes);
  int dev = 0;
  if (mode == 0 || hipGetDevice(&dev) != hipSuccess) return 1;
  const int unit = mode == 3 ? num_head_kv : num_head_kv / (mode == 2 ? 4 : 2);  // workgroup = (token range, head pair, quad or head)
  int num_wg = 2 * hpc_get_cu_count(dev);  // two 4-wave workgroups per CU (<= 256 registers, 65 KB of LDS each)
  const int wg_dev = hpc_dev_tuning_get(14);
  if (wg_dev > 0) num_wg = wg_dev;
  if (num_wg > num_bins) num_wg = num_bins;  // the scratch is sized for num_bins workgroups
  num_wg -= num_wg % unit;
  if (num_wg <= 0) return 1;  // fewer bins than head pairs: the first generation takes it
  char* base = static_cast<char*>(workspace);
  char* part = base + hpc::decode2::kCounterBytes +
               v1_workspace_bytes(num_bins, num_batch, num_head_kv, num_seq_q, num_head_q / num_head_kv);
  return hpc::decode2::launch(b, base, part, num_wg, mode, stream);
}

extern "C" int hpc_attention_decode_bf16_async(
    void* y_ptr, void* workspace, const int* task_map_ptr, const void* q_ptr, const void* kcache_ptr,
    const void* vcache_ptr, const int* block_ids_ptr, const int* num_seq_kvcache_ptr, int new_kv_included, int num_bins,
    int num_batch, int num_seq_q, int num_head_q, int num_head_kv, int num_dim_qk, int num_dim_v, int block_size,
    int num_seq_max_blocks, int ldY, int ldQ, int64_t kcache_block_stride,
    int64_t kcache_token_stride, int64_t kcache_head_stride, int64_t vcache_block_stride,
    int64_t vcache_token_stride, int64_t vcache_head_stride, hipStream_t stream) {
  using namespace hpc::decode;
  Args a;
  const Common c = fill_common(a, y_ptr, workspace, task_map_ptr, q_ptr, kcache_ptr, vcache_ptr,
                               block_ids_ptr, num_bins, num_batch, num_seq_q, num_head_q,
                               num_head_kv, num_dim_qk, num_dim_v, block_size, num_seq_max_blocks,
                               ldY, ldQ, kcache_block_stride, kcache_token_stride,
                               kcache_head_stride, vcache_block_stride, vcache_token_stride,
                               vcache_head_stride, 8);
  if (c.code != HPC_OK) return c.code;
  if (num_seq_kvcache_ptr && hpc_dev_tuning_get(28) != 1) {  // development key 28 = 1: bf16 on the first generation only
    hpc::decode2::Args b;
    b.q = q_ptr;
    b.kcache = kcache_ptr;
    b.vcache = vcache_ptr;
    b.block_ids = block_ids_ptr;
    b.lens = num_seq_kvcache_ptr;
    b.task_map = task_map_ptr;
    b.y = static_cast<uint16_t*>(y_ptr);
    b.qscale = b.kscale = b.vscale = nullptr;
    b.num_batch = num_batch;
    b.num_seq_q = num_seq_q;
    b.num_head_kv = num_head_kv;
    b.g_shift = a.g_shift;
    b.page_shift = a.page_shift;
    b.max_blocks = num_seq_max_blocks;
    b.ldq = ldQ * 2;  // bytes
    b.ldy = ldY;
    b.qscale_stride = 0;
    b.new_kv_included = new_kv_included;
    b.bf16 = 1;
    b.k_block_stride = kcache_block_stride * 2;
    b.k_token_stride = kcache_token_stride * 2;
    b.v_block_stride = vcache_block_stride * 2;
    b.v_token_stride = vcache_token_stride * 2;
    b.ks_block_stride = b.ks_row_stride = b.ks_head_stride = 0;
    b.scale_log2 = a.scale_log2;
    const int rc = try_second_generation(b, workspace, num_bins, num_batch, num_seq_q, num_head_q, num_head_kv, block_size,
                                         kcache_head_stride * 2, vcache_head_stride * 2, stream);
    if (rc <= 0) return rc;
  }
  return launch<false, 1>(a, num_bins, c.num_nb, stream);
}

extern "C" int hpc_attention_decode_fp8_async(
    void* y_ptr, void* workspace, const int* task_map_ptr, const void* q_ptr, const void* kcache_ptr,
    const void* vcache_ptr, const int* block_ids_ptr, const int* num_seq_kvcache_ptr, const float* qscale_ptr,
    const void* kscale_ptr, const float* vscale_ptr, int new_kv_included, int quant_type, int num_bins, int num_batch,
    int num_seq_q, int num_head_q, int num_head_kv, int num_dim_qk, int num_dim_v, int block_size,
    int num_seq_max_blocks, int qscale_pad_stride, int ldY, int ldQ, int64_t kcache_block_stride,
    int64_t kcache_token_stride, int64_t kcache_head_stride, int64_t vcache_block_stride,
    int64_t vcache_token_stride, int64_t vcache_head_stride, int64_t kscale_block_stride,
    int64_t kscale_row_stride, int64_t kscale_head_stride, hipStream_t stream) {
  using namespace hpc::decode;
  if (!qscale_ptr || !kscale_ptr || !vscale_ptr) return HPC_ERR_INVALID;
  if (quant_type != 0 && quant_type != 1) return HPC_ERR_UNSUPPORTED;
  if (num_seq_q > 4) return HPC_ERR_UNSUPPORTED;
  Args a;
  const Common c = fill_common(a, y_ptr, workspace, task_map_ptr, q_ptr, kcache_ptr, vcache_ptr,
                               block_ids_ptr, num_bins, num_batch, num_seq_q, num_head_q,
                               num_head_kv, num_dim_qk, num_dim_v, block_size, num_seq_max_blocks,
                               ldY, ldQ, kcache_block_stride, kcache_token_stride,
                               kcache_head_stride, vcache_block_stride, vcache_token_stride,
                               vcache_head_stride, 16);
  if (c.code != HPC_OK) return c.code;
  if (quant_type == 0 && block_size < 32) return HPC_ERR_UNSUPPORTED;  // scale rows hold 32 tokens
  a.qscale = qscale_ptr;
  a.kscale = static_cast<const float*>(kscale_ptr);
  a.vscale = vscale_ptr;
  a.qscale_stride = qscale_pad_stride;
  a.ks_block_stride = kscale_block_stride;
  a.ks_row_stride = kscale_row_stride;
  a.ks_head_stride = kscale_head_stride;
  // second generation (head pairs per load, deep prefetch, in-kernel plan) when the layout allows it
  {
    hpc::decode2::Args b;
    b.q = q_ptr;
    b.kcache = kcache_ptr;
    b.vcache = vcache_ptr;
    b.block_ids = block_ids_ptr;
    b.lens = num_seq_kvcache_ptr;
    b.task_map = task_map_ptr;
    b.y = static_cast<uint16_t*>(y_ptr);
    b.qscale = qscale_ptr;
    b.kscale = static_cast<const float*>(kscale_ptr);
    b.vscale = vscale_ptr;
    b.num_batch = num_batch;
    b.num_seq_q = num_seq_q;
    b.num_head_kv = num_head_kv;
    b.g_shift = a.g_shift;
    b.page_shift = a.page_shift;
    b.max_blocks = num_seq_max_blocks;
    b.ldq = ldQ;
    b.ldy = ldY;
    b.qscale_stride = qscale_pad_stride;
    b.new_kv_included = new_kv_included;
    b.bf16 = 0;
    b.k_block_stride = kcache_block_stride;
    b.k_token_stride = kcache_token_stride;
    b.v_block_stride = vcache_block_stride;
    b.v_token_stride = vcache_token_stride;
    b.ks_block_stride = kscale_block_stride;
    b.ks_row_stride = kscale_row_stride;
    b.ks_head_stride = kscale_head_stride;
    b.scale_log2 = a.scale_log2;
    // quant_type 0 (per-token K scales, per-head V scales) runs there too since round 6 (development key 54 = 1: first generation)
    b.ktok = quant_type == 0 ? 1 : 0;
    if (quant_type == 1 || hpc_dev_tuning_get(54) != 1) {
      const int rc = try_second_generation(b, workspace, num_bins, num_batch, num_seq_q, num_head_q, num_head_kv, block_size,
                                           kcache_head_stride, vcache_head_stride, stream);
      if (rc <= 0) return rc;
    }
  }
  if (quant_type == 1) return launch<true, 1>(a, num_bins, c.num_nb, stream);
  return launch<true, 0>(a, num_bins, c.num_nb, stream);
}
