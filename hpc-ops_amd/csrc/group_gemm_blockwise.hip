// Grouped FP8 GEMM with 128-block scales, weight-streaming ("decode / low-latency") form - gfx950.
//
//   Y[m, n] = bf16( sum_kb ( sum_{k in kb} X[m,k] W[g,n,k] ) * xs[m,kb] * ws[g, n/128, kb] )
// for the rows m of group (expert) g.  Replaces reference src/group_gemm/kernels.cuh:532-892
// (group_gemm_blockwise_fp8_kernel), group_gemm_blockwise_fp8.cu:368-457 and the gather-free
// "scatter-A" variant cp_async/group_gemm_fp8_scatter.cu:70-302 (here: optional row_index).
//
// MI355X design: at decode batch sizes every expert sees a handful of tokens, so the op is a
// stream of the weights (HBM-bound).  Same swap as the reference - weights on the
// MFMA M axis, tokens on N (v_mfma_f32_16x16x32_fp8_fp8, 16 / 32 / 48 tokens per pass) - but no TMA /
// warp specialisation: each WAVE owns 16 weight rows of one expert and streams them HBM -> VGPR
// with 16-byte non-temporal full-row buffer loads (3-4 stages of 256 B per row = 12-16 KB in flight
// per wave, 8 waves per CU).  The activation tile of the expert is shared by the waves of a
// workgroup through a small double-buffered LDS tile, fetched at the same prefetch depth as the
// weights (vmcnt retires in order, so a shallower activation pipeline would drain the weight
// stream).  Per 128-wide k block the fp32 partial is rescaled by xs*ws and accumulated (reference
// kernels.cuh:806-836).  Groups above ~40 rows go to the tiled kernels (group_gemm_tiled256.hip,
// group_gemm_tiled.hip) - see launch_stream_gemm at the end of this file.
#include "hpc_common.h"
#include "hpc_dev.h"
#include "../../include/hpc_amd.h"
#include "group_gemm.h"

namespace hpc {
namespace ggemm {



constexpr int kThreads = 256;
// stages in flight per wave (template kDepth, default 4); one stage = 2 k-blocks = 256 B per weight row
constexpr int kXRow = 272;    // LDS bytes per staged activation row (256 + 16: conflict-free b128 reads)

__device__ __forceinline__ long pack64(uint32_t lo, uint32_t hi) {
  return static_cast<long>(static_cast<uint64_t>(lo) | (static_cast<uint64_t>(hi) << 32));
}

// kMT = token blocks of 16 served per pass over the weights (1, 2 or 4); kR = 16-row weight blocks
// per wave (1 or 2): the workgroup tile is 64*kR rows.
//
// Workgroup = 4 waves = one 64*kR-row weight tile of one expert (inside one 128-row scale block).  Each
// wave streams its own 16*kR weight rows HBM -> VGPR in MFMA A-operand layout (non-temporal buffer
// loads, kDepth stages = 32 KB in flight per wave, 256 B per row per stage issued back to back).
// The activation tile (16*kMT tokens x 256 B) and its scales are shared: every wave fetches a
// quarter at the same prefetch depth as the weights (vmcnt retires in order - a shallower
// activation pipeline would drain the weight stream), drops it into a double-buffered LDS tile,
// and after ONE barrier per stage all waves read their B operands from LDS.
// kWaves = 4 or 8 waves per workgroup; with 8 the activation staging is spread over twice the lanes
// (half the staging registers per lane) - that is what makes 64 tokens per pass fit without spills.
template <int kMT, int kR, int kWaves = 4, int kDepth = 4>
__global__ __launch_bounds__(64 * kWaves, (kWaves == 8 || kMT * kR >= 8) ? 1 : 2) void gemm_blockwise_stream_kernel(
    const Args a) {
  constexpr int kTok = 16 * kMT;
  constexpr int kXI = 4 * kMT / kWaves;  // activation staging instructions (4 rows x 256 B each) per wave
  static_assert(kXI >= 1, "need at least one staging instruction per wave");
  __shared__ __attribute__((aligned(16))) uint8_t s_x[2][kTok * kXRow];
  __shared__ float s_xs[2][2][kTok];
  // wave-private 16-row x 256-byte weight tile (double-buffered): weights are fetched with full-row
  // loads (16 lanes x 16 B per row segment - the access shape that streams fastest, see
  // attention_decode.hip) and reach the MFMA A-operand layout through this tile; no barrier needed.
  __shared__ __attribute__((aligned(16))) uint8_t s_w[kWaves][2][16 * kXRow];

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int r16 = lane & 15, g4 = lane >> 4;
  const int e = blockIdx.y;
  const int m_cnt = as_const(a.seqlens)[e];
  if (m_cnt <= 0) return;
  const int n0 = (blockIdx.x * kWaves + wave) * 16 * kR;  // N % 128 == 0 is checked by the launcher
  const int m0 = as_const(a.cu_seqlens)[e];
  const int K = a.K, KB = a.KB;  // KB = ceil(K / 128)
  const int nstage = (KB + 1) >> 1;

  const uint8_t* wbase = a.w + (static_cast<long>(e) * a.N + n0) * K;
  const unsigned w_bytes = 16u * static_cast<unsigned>(K);
  const int w_voff = g4 * K + r16 * 16;  // lane -> (row 4*qd + g4, 16-byte chunk r16) of the stage slab
  const cint_ptr ws_row = as_const(reinterpret_cast<const int*>(a.ws)) +
                          static_cast<long>(e) * a.ws_group_stride + (n0 >> 7) * a.ws_ntile_stride;
  const int xs_kb_bytes = static_cast<int>(a.xs_kb_stride * 4);
  const long col0 = a.col_base ? static_cast<long>(as_const(a.col_base)[e]) * a.tile_m : 0;

  const int npass = (m_cnt + kTok - 1) / kTok;
  for (int p = 0; p < npass; ++p) {
    // The weight loads of the first kDepth stages go out BEFORE the lane's activation roles are known (round 6): those hang on a
    // chain of dependent loads (row_index -> activation row -> its address) that the weights do not need, and a workgroup only
    // lives for ~16 stages - two memory round trips in front of its first weight byte were ~10 % of its life.
    u32x4 wb[kDepth][2][kR][2];  // [stage][k-block of the stage][row block][64-byte half]
    auto issue_w = [&](int d, int st) {
      // Weights: whole stage on/off (a descriptor with num_records = 0 fetches nothing).  When K is
      // not a multiple of 256 the tail of the last stage reads into the next weight row - finite
      // e4m3 data that meets ZERO activations: the activation loads below are bounded per lane.
      const int koff = 2 * st * 128;
      const unsigned w_on = st < nstage ? w_bytes : 0u;
#pragma unroll
      for (int rb = 0; rb < kR; ++rb) {
        const auto rw = make_rsrc(wbase + static_cast<long>(rb) * 16 * K, w_on);
#pragma unroll
        for (int qd = 0; qd < 4; ++qd) wb[d][qd >> 1][rb][qd & 1] = buf_ld16<2>(rw, w_voff, koff + qd * 4 * K);
      }
    };
    // ---- this lane's roles in staging the activation tile ------------------------------------------
    // x rows: instruction i = wave*kXI + j loads tile rows 4i .. 4i+3, lane -> (row 4i + g4, chunk r16)
    // (the row_index lookups are requested first and consumed AFTER the weight loads have been issued: they retire first - in
    // order - so waiting for them leaves the kDepth stages of weights in flight)
    const bool xs_role = wave < 2 && lane < kTok;
    int xrow_j[kXI], xrow_s = 0, sc_s = 0;
#pragma unroll
    for (int j = 0; j < kXI; ++j) {
      const int slot = p * kTok + 4 * (wave * kXI + j) + g4;
      const int sc = slot < m_cnt ? slot : m_cnt - 1;
      xrow_j[j] = a.row_index ? a.row_index[m0 + sc] : m0 + sc;
    }
    if (xs_role) {
      const int slot = p * kTok + lane;
      sc_s = slot < m_cnt ? slot : m_cnt - 1;
      xrow_s = a.row_index ? a.row_index[m0 + sc_s] : m0 + sc_s;
    }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int d = 0; d < kDepth; ++d) issue_w(d, d);
    __builtin_amdgcn_sched_barrier(0);
    unsigned x_voff[kXI];
    int x_lds[kXI];
#pragma unroll
    for (int j = 0; j < kXI; ++j) {
      const int trow = 4 * (wave * kXI + j) + g4;
      x_voff[j] = static_cast<unsigned>(xrow_j[j]) * static_cast<unsigned>(K) + r16 * 16;
      x_lds[j] = trow * kXRow + r16 * 16;
    }
    // scales: waves 0/1 load k-block 0/1 of the stage, lane -> token `lane`
    unsigned xs_voff = 0;
    if (xs_role) {
      const long term = a.col_base ? col0 + sc_s : static_cast<long>(xrow_s);
      xs_voff = static_cast<unsigned>(term * a.xs_row_stride * 4);
    }

    u32x4 xb[kDepth][kXI];
    float xsb[kDepth];
    auto issue_x = [&](int d, int st) {
      const int kb0 = 2 * st;
      const int koff = kb0 * 128;
      // activation quarter: one 16-byte chunk of the 256-byte slab per lane; chunks at k >= K get an
      // out-of-range offset and read as zero
      const auto rx = make_rsrc(a.x, a.x_bytes);
      const bool x_ok = koff + r16 * 16 < K;
#pragma unroll
      for (int j = 0; j < kXI; ++j) xb[d][j] = buf_ld16<0>(rx, x_ok ? x_voff[j] : 0xffffff00u, koff);
      const auto rs = make_rsrc(a.xs, (a.has_xs && xs_role && kb0 + wave < KB) ? 0xffffffffu : 0u);
      xsb[d] = __uint_as_float(
          __builtin_amdgcn_raw_buffer_load_b32(rs, xs_voff, (kb0 + wave) * xs_kb_bytes, 0));
    };

    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int d = 0; d < kDepth; ++d) {
      issue_x(d, d);
      __builtin_amdgcn_sched_barrier(0);  // keep issue order = consumption order (in-order vmcnt)
    }

    f32x4 tot[kR][kMT];
#pragma unroll
    for (int rb = 0; rb < kR; ++rb)
#pragma unroll
      for (int mt = 0; mt < kMT; ++mt) tot[rb][mt] = f32x4{0.f, 0.f, 0.f, 0.f};

    for (int st0 = 0; st0 < nstage; st0 += kDepth) {
#pragma unroll
      for (int d = 0; d < kDepth; ++d) {
        const int st = st0 + d;
        const int buf = (kDepth & 1) ? (st & 1) : (d & 1);  // stage parity (== d parity when kDepth is even)
        // stage my quarter of the activation tile, then one barrier for the whole workgroup
#pragma unroll
        for (int j = 0; j < kXI; ++j)
          *reinterpret_cast<u32x4*>(&s_x[buf][x_lds[j]]) = xb[d][j];
        if (xs_role) s_xs[buf][wave][lane] = xsb[d];
        // weights: full-row layout -> MFMA A-operand layout (lane (r16, g4): row r16, chunk kbl*8+h*4+g4)
        u32x4 wf[kR][2][2];
#pragma unroll
        for (int rb = 0; rb < kR; ++rb) {
          uint8_t* wt = s_w[wave][(rb + buf) & 1];
#pragma unroll
          for (int qd = 0; qd < 4; ++qd)
            *reinterpret_cast<u32x4*>(wt + (4 * qd + g4) * kXRow + r16 * 16) = wb[d][qd >> 1][rb][qd & 1];
#pragma unroll
          for (int kbl = 0; kbl < 2; ++kbl)
#pragma unroll
            for (int h = 0; h < 2; ++h)
              wf[rb][kbl][h] = *reinterpret_cast<const u32x4*>(wt + r16 * kXRow + (kbl * 8 + h * 4 + g4) * 16);
        }
        __syncthreads();
#pragma unroll
        for (int kbl = 0; kbl < 2; ++kbl) {
          const int kb = 2 * st + kbl;
          const int kbc = kb < KB ? kb : KB - 1;
          const float wsk = kb < KB ? __int_as_float(ws_row[kbc * a.ws_kb_stride]) : 0.f;
          // token blocks two at a time: their MFMA chains (4 dependent k-steps each) interleave
          constexpr int kPair = (kMT % 2 == 0) ? 2 : 1;
#pragma unroll
          for (int mt0 = 0; mt0 < kMT; mt0 += kPair) {
            u32x4 b0[kPair], b1[kPair];
            float f[kPair];
#pragma unroll
            for (int q = 0; q < kPair; ++q) {
              const uint8_t* xp = &s_x[buf][((mt0 + q) * 16 + r16) * kXRow + kbl * 128 + g4 * 16];
              b0[q] = *reinterpret_cast<const u32x4*>(xp);
              b1[q] = *reinterpret_cast<const u32x4*>(xp + 64);
              f[q] = a.has_xs ? s_xs[buf][kbl][(mt0 + q) * 16 + r16] * wsk : wsk;
            }
#pragma unroll
            for (int rb = 0; rb < kR; ++rb) {
              f32x4 part[kPair];
#pragma unroll
              for (int q = 0; q < kPair; ++q) part[q] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
              for (int q = 0; q < kPair; ++q)
                part[q] = __builtin_amdgcn_mfma_f32_16x16x32_fp8_fp8(
                    pack64(wf[rb][kbl][0][0], wf[rb][kbl][0][1]), pack64(b0[q][0], b0[q][1]), part[q], 0, 0, 0);
#pragma unroll
              for (int q = 0; q < kPair; ++q)
                part[q] = __builtin_amdgcn_mfma_f32_16x16x32_fp8_fp8(
                    pack64(wf[rb][kbl][0][2], wf[rb][kbl][0][3]), pack64(b0[q][2], b0[q][3]), part[q], 0, 0, 0);
#pragma unroll
              for (int q = 0; q < kPair; ++q)
                part[q] = __builtin_amdgcn_mfma_f32_16x16x32_fp8_fp8(
                    pack64(wf[rb][kbl][1][0], wf[rb][kbl][1][1]), pack64(b1[q][0], b1[q][1]), part[q], 0, 0, 0);
#pragma unroll
              for (int q = 0; q < kPair; ++q)
                part[q] = __builtin_amdgcn_mfma_f32_16x16x32_fp8_fp8(
                    pack64(wf[rb][kbl][1][2], wf[rb][kbl][1][3]), pack64(b1[q][2], b1[q][3]), part[q], 0, 0, 0);
#pragma unroll
              for (int q = 0; q < kPair; ++q)
#pragma unroll
                for (int i = 0; i < 4; ++i) tot[rb][mt0 + q][i] = fmaf(part[q][i], f[q], tot[rb][mt0 + q][i]);
            }
            __builtin_amdgcn_sched_barrier(0);  // stop hipcc hoisting every LDS read of the stage (VGPRs)
          }
        }
        issue_w(d, st + kDepth);
        issue_x(d, st + kDepth);
      }
    }

#pragma unroll
    for (int mt = 0; mt < kMT; ++mt) {
      const int slot = p * kTok + mt * 16 + r16;
      if (slot < m_cnt) {
#pragma unroll
        for (int rb = 0; rb < kR; ++rb) {
          u32x2 pk;
          pk[0] = pack_bf16x2(tot[rb][mt][0], tot[rb][mt][1]);
          pk[1] = pack_bf16x2(tot[rb][mt][2], tot[rb][mt][3]);
          *reinterpret_cast<u32x2*>(a.y + static_cast<long>(m0 + slot) * a.N + n0 + rb * 16 + g4 * 4) = pk;
        }
      }
    }
    __builtin_amdgcn_s_waitcnt(0x0F70);  // retire stores before the next pass (see attention_decode.hip)
    __syncthreads();
  }
}


// Round 6: the stream's stage loop re-ordered for the latency of ONE stage (<= 32 tokens per pass, 4 waves x 16 rows).
// In the kernel above a wave's stage is a serial chain - wait for the stage's loads, stage writes, operand reads, barrier, then per
// k-block a scalar load of the weight scale that is waited for on the spot, an LDS read of the token scale, four dependent MFMAs -
// and the refill of the stage's registers is issued at its very END: ~1 200 of the ~2 000 cycles a stage may take at the HBM rate
// pass before the memory pipeline hears from the wave again, so fewer than the kDepth stages are really in flight (0.76-0.78 of
// 8 TB/s at T = 16 ... 64 against 0.87 for a pure stream of the same weights with the same 16 KB per wave in flight,
// tools/probes/probe_wstream.hip).  Here
//  * the refill goes out as soon as the stage's registers have been written to LDS (they are dead then), in front of the barrier
//    and the MFMAs;
//  * the weight scales of the stage are requested at its top, branch-free (clamped index, zeroed afterwards), and the token
//    scales ride in the same LDS read batch as the B operands;
//  * the token-scale load uses a wave-uniform descriptor (the lane-dependent one above cost a waterfall loop per stage);
//  * the weight loads of the first kDepth stages are issued before the lane's activation roles are known (see above).
// Same arithmetic in the same order: bit-identical to the kernel above.  Development key 56 = 1 keeps that one.
//  * (kK128) a k-block is ONE v_mfma_f32_16x16x128_f8f6f4 per token block - the 256 x 256 kernel's instruction, operand
//    convention (lane (r16, g4): chunks g4 and g4 + 4 of its row on both sides) and arithmetic: results are bit-identical to
//    that kernel's bodies - instead of a chain of four dependent K = 32 MFMAs (kK128 = false: bit-identical to the kernel above).
template <int kMT, int kDepth, bool kK128 = true>
__global__ __launch_bounds__(256, 2) void gemm_blockwise_stream2_kernel(const Args a) {
  constexpr int kWaves = 4;
  constexpr int kTok = 16 * kMT;
  constexpr int kXI = 4 * kMT / kWaves;  // activation staging instructions (4 rows x 256 B each) per wave
  static_assert(kXI >= 1, "need at least one staging instruction per wave");
  __shared__ __attribute__((aligned(16))) uint8_t s_x[2][kTok * kXRow];
  __shared__ float s_xs[2][2][kTok];
  __shared__ __attribute__((aligned(16))) uint8_t s_w[kWaves][2][16 * kXRow];

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int r16 = lane & 15, g4 = lane >> 4;
  const int e = blockIdx.y;
  const int m_cnt = as_const(a.seqlens)[e];
  if (m_cnt <= 0) return;
  const int n0 = (blockIdx.x * kWaves + wave) * 16;
  const int m0 = as_const(a.cu_seqlens)[e];
  const int K = a.K, KB = a.KB;
  const int nstage = (KB + 1) >> 1;

  const uint8_t* wbase = a.w + (static_cast<long>(e) * a.N + n0) * K;
  const unsigned w_bytes = 16u * static_cast<unsigned>(K);
  const int w_voff = g4 * K + r16 * 16;
  const cint_ptr ws_row = as_const(reinterpret_cast<const int*>(a.ws)) +
                          static_cast<long>(e) * a.ws_group_stride + (n0 >> 7) * a.ws_ntile_stride;
  const int xs_kb_bytes = static_cast<int>(a.xs_kb_stride * 4);
  const long col0 = a.col_base ? static_cast<long>(as_const(a.col_base)[e]) * a.tile_m : 0;
  const bool has_xs = a.has_xs != 0;

  const int npass = (m_cnt + kTok - 1) / kTok;
  for (int p = 0; p < npass; ++p) {
    const bool xs_role = wave < 2 && lane < kTok;
    int xrow_j[kXI], xrow_s = 0, sc_s = 0;
#pragma unroll
    for (int j = 0; j < kXI; ++j) {
      const int slot = p * kTok + 4 * (wave * kXI + j) + g4;
      const int sc = slot < m_cnt ? slot : m_cnt - 1;
      xrow_j[j] = a.row_index ? a.row_index[m0 + sc] : m0 + sc;
    }
    {
      const int slot = p * kTok + (lane < kTok ? lane : 0);
      sc_s = slot < m_cnt ? slot : m_cnt - 1;
      xrow_s = a.row_index ? a.row_index[m0 + sc_s] : m0 + sc_s;
    }
    u32x4 wb[kDepth][4];
    auto issue_w = [&](int d, int st) {
      const int koff = 2 * st * 128;
      const auto rw = make_rsrc(wbase, st < nstage ? w_bytes : 0u);
#pragma unroll
      for (int qd = 0; qd < 4; ++qd) wb[d][qd] = buf_ld16<2>(rw, w_voff, koff + qd * 4 * K);
    };
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int d = 0; d < kDepth; ++d) issue_w(d, d);
    __builtin_amdgcn_sched_barrier(0);
    unsigned x_voff[kXI];
    int x_lds[kXI];
#pragma unroll
    for (int j = 0; j < kXI; ++j) {
      x_voff[j] = static_cast<unsigned>(xrow_j[j]) * static_cast<unsigned>(K) + r16 * 16;
      x_lds[j] = (4 * (wave * kXI + j) + g4) * kXRow + r16 * 16;
    }
    // scales: waves 0 / 1 load k-block 0 / 1 of the stage, lane -> token `lane` (lanes past the tile: token 0, never stored)
    const long term = a.col_base ? col0 + sc_s : static_cast<long>(xrow_s);
    const unsigned xs_voff = static_cast<unsigned>(term * a.xs_row_stride * 4);
    const int xs_wave = wave < 2 ? wave : 0;

    u32x4 xb[kDepth][kXI];
    float xsb[kDepth];
    auto issue_x = [&](int d, int st) {
      const int kb0 = 2 * st;
      const int koff = kb0 * 128;
      const auto rx = make_rsrc(a.x, a.x_bytes);
      const bool x_ok = koff + r16 * 16 < K;
#pragma unroll
      for (int j = 0; j < kXI; ++j) xb[d][j] = buf_ld16<0>(rx, x_ok ? x_voff[j] : 0xffffff00u, koff);
      // wave-uniform, and pinned so that hipcc sees it (a descriptor it cannot prove uniform costs a waterfall loop per load)
      const auto rs = make_rsrc(a.xs, __builtin_amdgcn_readfirstlane((has_xs && wave < 2 && kb0 + xs_wave < KB) ? 0xffffffffu : 0u));
      xsb[d] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rs, xs_voff, (kb0 + xs_wave) * xs_kb_bytes, 0));
    };
#pragma unroll
    for (int d = 0; d < kDepth; ++d) {
      issue_x(d, d);
      __builtin_amdgcn_sched_barrier(0);  // keep issue order = consumption order (in-order vmcnt)
    }

    f32x4 tot[kMT];
#pragma unroll
    for (int mt = 0; mt < kMT; ++mt) tot[mt] = f32x4{0.f, 0.f, 0.f, 0.f};

    for (int st0 = 0; st0 < nstage; st0 += kDepth) {
#pragma unroll
      for (int d = 0; d < kDepth; ++d) {
        const int st = st0 + d;
        const int buf = (kDepth & 1) ? (st & 1) : (d & 1);
        // this stage's two weight scales: requested now, used after the barrier (clamped index, zeroed past the end: no branch)
        const int kb_a = 2 * st, kb_b = 2 * st + 1;
        const int wsi_a = ws_row[(kb_a < KB ? kb_a : KB - 1) * a.ws_kb_stride];
        const int wsi_b = ws_row[(kb_b < KB ? kb_b : KB - 1) * a.ws_kb_stride];
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int j = 0; j < kXI; ++j) *reinterpret_cast<u32x4*>(&s_x[buf][x_lds[j]]) = xb[d][j];
        if (xs_role) s_xs[buf][wave][lane] = xsb[d];
        uint8_t* wt = s_w[wave][buf];
#pragma unroll
        for (int qd = 0; qd < 4; ++qd) *reinterpret_cast<u32x4*>(wt + (4 * qd + g4) * kXRow + r16 * 16) = wb[d][qd];
        __builtin_amdgcn_sched_barrier(0);
        // the stage's registers are in LDS: refill them now - the loads travel while this stage is computed
        issue_w(d, st + kDepth);
        issue_x(d, st + kDepth);
        __builtin_amdgcn_sched_barrier(0);
        u32x4 wf[2][2];
#pragma unroll
        for (int kbl = 0; kbl < 2; ++kbl)
#pragma unroll
          for (int h = 0; h < 2; ++h)
            wf[kbl][h] = *reinterpret_cast<const u32x4*>(wt + r16 * kXRow + (kbl * 8 + h * 4 + g4) * 16);
        __syncthreads();
        u32x4 b0[2][kMT], b1[2][kMT];
        float f[2][kMT];
        const float wsk[2] = {kb_a < KB ? __int_as_float(wsi_a) : 0.f, kb_b < KB ? __int_as_float(wsi_b) : 0.f};
        auto read_b = [&](int kbl) {
#pragma unroll
          for (int q = 0; q < kMT; ++q) {
            const uint8_t* xp = &s_x[buf][(q * 16 + r16) * kXRow + kbl * 128 + g4 * 16];
            b0[kbl][q] = *reinterpret_cast<const u32x4*>(xp);
            b1[kbl][q] = *reinterpret_cast<const u32x4*>(xp + 64);
            f[kbl][q] = has_xs ? s_xs[buf][kbl][q * 16 + r16] * wsk[kbl] : wsk[kbl];
          }
        };
        if constexpr (kMT == 1) {  // one batch of LDS reads for the whole stage (two token blocks: per k-block - registers)
          read_b(0);
          read_b(1);
        }
#pragma unroll
        for (int kbl = 0; kbl < 2; ++kbl) {
          if constexpr (kMT != 1) {
            read_b(kbl);
            __builtin_amdgcn_sched_barrier(0);
          }
          f32x4 part[kMT];
#pragma unroll
          for (int q = 0; q < kMT; ++q) part[q] = f32x4{0.f, 0.f, 0.f, 0.f};
          if constexpr (kK128) {
            const i32x8 av = {static_cast<int>(wf[kbl][0][0]), static_cast<int>(wf[kbl][0][1]), static_cast<int>(wf[kbl][0][2]),
                              static_cast<int>(wf[kbl][0][3]), static_cast<int>(wf[kbl][1][0]), static_cast<int>(wf[kbl][1][1]),
                              static_cast<int>(wf[kbl][1][2]), static_cast<int>(wf[kbl][1][3])};
#pragma unroll
            for (int q = 0; q < kMT; ++q) {
              const i32x8 bv = {static_cast<int>(b0[kbl][q][0]), static_cast<int>(b0[kbl][q][1]), static_cast<int>(b0[kbl][q][2]),
                                static_cast<int>(b0[kbl][q][3]), static_cast<int>(b1[kbl][q][0]), static_cast<int>(b1[kbl][q][1]),
                                static_cast<int>(b1[kbl][q][2]), static_cast<int>(b1[kbl][q][3])};
              part[q] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(av, bv, part[q], 0, 0, 0, 0, 0, 0);
            }
          } else {
#pragma unroll
            for (int q = 0; q < kMT; ++q)
              part[q] = __builtin_amdgcn_mfma_f32_16x16x32_fp8_fp8(pack64(wf[kbl][0][0], wf[kbl][0][1]),
                                                                   pack64(b0[kbl][q][0], b0[kbl][q][1]), part[q], 0, 0, 0);
#pragma unroll
            for (int q = 0; q < kMT; ++q)
              part[q] = __builtin_amdgcn_mfma_f32_16x16x32_fp8_fp8(pack64(wf[kbl][0][2], wf[kbl][0][3]),
                                                                   pack64(b0[kbl][q][2], b0[kbl][q][3]), part[q], 0, 0, 0);
#pragma unroll
            for (int q = 0; q < kMT; ++q)
              part[q] = __builtin_amdgcn_mfma_f32_16x16x32_fp8_fp8(pack64(wf[kbl][1][0], wf[kbl][1][1]),
                                                                   pack64(b1[kbl][q][0], b1[kbl][q][1]), part[q], 0, 0, 0);
#pragma unroll
            for (int q = 0; q < kMT; ++q)
              part[q] = __builtin_amdgcn_mfma_f32_16x16x32_fp8_fp8(pack64(wf[kbl][1][2], wf[kbl][1][3]),
                                                                   pack64(b1[kbl][q][2], b1[kbl][q][3]), part[q], 0, 0, 0);
          }
#pragma unroll
          for (int q = 0; q < kMT; ++q)
#pragma unroll
            for (int i = 0; i < 4; ++i) tot[q][i] = fmaf(part[q][i], f[kbl][q], tot[q][i]);
        }
      }
    }

#pragma unroll
    for (int mt = 0; mt < kMT; ++mt) {
      const int slot = p * kTok + mt * 16 + r16;
      if (slot < m_cnt) {
        u32x2 pk;
        pk[0] = pack_bf16x2(tot[mt][0], tot[mt][1]);
        pk[1] = pack_bf16x2(tot[mt][2], tot[mt][3]);
        *reinterpret_cast<u32x2*>(a.y + static_cast<long>(m0 + slot) * a.N + n0 + g4 * 4) = pk;
      }
    }
    __builtin_amdgcn_s_waitcnt(0x0F70);  // retire stores before the next pass (see attention_decode.hip)
    __syncthreads();
  }
}

}  // namespace ggemm
}  // namespace hpc

bool hpc_ggemm_p8_selected(int num_group, int m, int n, int k, const void* cu_tiles128) {
  const int tiled_mode = hpc_dev_tuning_get(3);
  // (up to 64 groups the kernel finds its work item from one round of lane-parallel loads, above that from one round
  // per 64 groups: tests/test_fuse_moe_blockwise.py::test_group_gemm_blockwise_many_groups, 65 ... 256 groups)
  // From 16 rows per group on (round 5; rounds 2-4: from 192): with the carried tails, the tail body for <= 64 rows and the
  // half-tile body the 256 x 256 kernel overtakes the 256 x 128 ring kernel everywhere and the streaming kernel from
  // ~16 rows per group (fused MoE, E64 / top-8 / H4096 / I11008, us: T = 128 1554-1561 against 1564-1676, T = 256
  // 1590-1596 against 1711-1836, T = 512 1753-1755 against 1876-1978, T = 1024 2069-2102 against 2432-2607; below
  // it loses: T = 64 1552-1563 against 1396-1461 - profiles/round5_moe_kernel_choice.txt).  Development key 25 restores
  // the old threshold.
  const int p8_from = hpc_dev_tuning_get(25) == 1 ? 192 : 16;
  return cu_tiles128 && n % 256 == 0 && k >= 128 &&
         (tiled_mode == 4 || (tiled_mode == 0 && m / num_group >= p8_from));
}

namespace {
int launch_stream_gemm(hpc::ggemm::Args& a, int num_group, int m, int n, const void* cu_tiles128,
                       hipStream_t stream) {
  using namespace hpc::ggemm;
  // groups above ~20 tokens: tiled kernels (need the scan of ceil(seqlens/128)): the 256 x 128 LDS-DMA ring
  // kernel when n allows (one pass over the weights for up to 128 tokens, 100 KB in flight per CU without
  // staging registers; measured on E64 / top-8: T = 128 (16 per group) 1.61 vs 1.50 ms for the streaming form,
  // T = 192 1.63 vs 1.79, T = 256 1.73 vs 1.85, T = 384 1.75 ms), else the 128 x 128 register-staged one
  // development key 3: 0 auto, 1 never tiled, 2 always 256 x 128 (when possible), 3 always 128 x 128,
  // 4 always 256 x 256 (when possible)
  const int tiled_mode = hpc_dev_tuning_get(3);
  // the 256 x 256 kernel from 16 rows per group on (hpc_ggemm_p8_selected: tail body for <= 64 rows, half-tile body for <= 128)
  if (tiled_mode != 1 && hpc_ggemm_p8_selected(num_group, m, n, a.K, cu_tiles128))
    return hpc_ggemm_launch_p8(a, static_cast<const int*>(cu_tiles128), num_group, m, n, stream);
  if (cu_tiles128 && n % 128 == 0 && tiled_mode != 1 && (tiled_mode >= 2 || m / num_group > 20)) {
    if (n % 256 == 0 && a.K >= 128 && tiled_mode != 3)
      return hpc_ggemm_launch_tiled256(a, static_cast<const int*>(cu_tiles128), num_group, m, n, stream);
    return hpc_ggemm_launch_tiled(a, static_cast<const int*>(cu_tiles128), num_group, m, n, stream);
  }
  // tokens served per pass over the weights, from the average group size (the reference picks its
  // tileM the same way, fuse_moe/entry.cc:525-543); larger groups take several passes
  const int avg = m / num_group;
  const int forced = hpc_dev_tuning_get(1);
  // forced: 1 / 2 / 3 / 4 = tokens-per-pass 16 / 32 / 48 / 64 with 16 rows per wave; 8 = 64 tokens, 32 rows per
  // wave; 16 / 32 = 64 / 32 tokens with 8 waves per workgroup
  // measured on E64 / top-8: 16 tokens per pass up to ~10 per group, 32 up to ~22, then 48 (one pass still
  // covers nearly every group of a 32-average batch; 64 per pass is register-bound and slower)
  const int mt = forced ? forced : (avg <= 10 ? 1 : (avg <= 22 ? 2 : 3));
  if (mt == 8 && n % 128 == 0) {
    dim3 grid(n / 128, num_group);
    gemm_blockwise_stream_kernel<4, 2><<<grid, kThreads, 0, stream>>>(a);
  } else if (mt == 16 && n % 128 == 0) {  // 64 tokens per pass, 8 waves x 16 rows
    dim3 grid(n / 128, num_group);
    gemm_blockwise_stream_kernel<4, 1, 8, 3><<<grid, 512, 0, stream>>>(a);
  } else if (mt == 32 && n % 128 == 0) {  // 32 tokens per pass, 8 waves x 16 rows
    dim3 grid(n / 128, num_group);
    gemm_blockwise_stream_kernel<2, 1, 8><<<grid, 512, 0, stream>>>(a);
  } else {
    dim3 grid(n / 64, num_group);
    const int k56 = hpc_dev_tuning_get(56);  // development key 56: 1 = the stage loop of rounds 1-5, 2 = the new loop on K = 32 MFMAs
    if (mt == 1 && k56 == 0)
      gemm_blockwise_stream2_kernel<1, 4><<<grid, kThreads, 0, stream>>>(a);
    else if (mt == 2 && k56 == 0)
      gemm_blockwise_stream2_kernel<2, 4><<<grid, kThreads, 0, stream>>>(a);
    else if (kHpcDevBuild && mt == 1 && k56 == 2)
      gemm_blockwise_stream2_kernel<1, 4, false><<<grid, kThreads, 0, stream>>>(a);
    else if (kHpcDevBuild && mt == 2 && k56 == 2)
      gemm_blockwise_stream2_kernel<2, 4, false><<<grid, kThreads, 0, stream>>>(a);
    else if (mt == 1)
      gemm_blockwise_stream_kernel<1, 1><<<grid, kThreads, 0, stream>>>(a);
    else if (mt == 3)
      gemm_blockwise_stream_kernel<3, 1, 4, 3><<<grid, kThreads, 0, stream>>>(a);
    else if (mt == 4)
      gemm_blockwise_stream_kernel<4, 1, 4, 3><<<grid, kThreads, 0, stream>>>(a);
    else
      gemm_blockwise_stream_kernel<2, 1><<<grid, kThreads, 0, stream>>>(a);
  }
  HPC_CHECK_LAUNCH();
  return HPC_OK;
}
}  // namespace

extern "C" int hpc_group_gemm_blockwise_fp8_async(
    void* y_ptr, const void* x_ptr, const void* w_ptr, const void* seqlens_ptr,
    const void* cu_seqlens_ptr, const void* xscale_ptr, const void* wscale_ptr,
    const void* row_index_ptr, const void* col_base_ptr, int num_group, int m, int n, int k,
    int num_block_k_pad4, int tile_m, int64_t xscale_row_stride, int64_t xscale_kb_stride,
    const void* cu_tiles128_ptr, hipStream_t stream) {
  using namespace hpc::ggemm;
  if (!y_ptr || !x_ptr || !w_ptr || !seqlens_ptr || !cu_seqlens_ptr || !xscale_ptr || !wscale_ptr)
    return HPC_ERR_INVALID;
  if (num_group <= 0 || n <= 0 || k <= 0) return HPC_ERR_INVALID;
  if (m <= 0) return HPC_OK;
  if ((n & 127) || (k & 127)) return HPC_ERR_UNSUPPORTED;  // 128x128 weight scale blocks
  if (num_block_k_pad4 < k / 128) return HPC_ERR_INVALID;
  if (static_cast<int64_t>(m) * k > 0xfffffe00ll) return HPC_ERR_UNSUPPORTED;  // 32-bit x offsets
  Args a;
  a.x = static_cast<const uint8_t*>(x_ptr);
  a.w = static_cast<const uint8_t*>(w_ptr);
  a.xs = static_cast<const float*>(xscale_ptr);
  a.ws = static_cast<const float*>(wscale_ptr);
  a.y = static_cast<uint16_t*>(y_ptr);
  a.seqlens = static_cast<const int*>(seqlens_ptr);
  a.cu_seqlens = static_cast<const int*>(cu_seqlens_ptr);
  a.row_index = static_cast<const int*>(row_index_ptr);
  a.col_base = static_cast<const int*>(col_base_ptr);
  a.N = n;
  a.K = k;
  a.KB = k / 128;
  a.tile_m = tile_m;
  a.ws_group_stride = (n / 128) * num_block_k_pad4;
  a.ws_ntile_stride = num_block_k_pad4;
  a.ws_kb_stride = 1;
  a.has_xs = 1;
  a.x_bytes = 0xfffffe00u;  // x may be indexed through row_index: rows beyond m exist (< 4 GB checked)
  a.xs_row_stride = xscale_row_stride;
  a.xs_kb_stride = xscale_kb_stride;
  return launch_stream_gemm(a, num_group, m, n, cu_tiles128_ptr, stream);
}

// Per-tensor variant: Y = bf16( (X W^T) * y_scale[g] ), no activation scales.
// reference: group_gemm_fp8_async / group_gemm_pertensor_fp8 (src/group_gemm/group_gemm.h,
// kernels.cuh:215-530) and the gather-free cp.async path (cp_async/group_gemm_fp8_scatter.cu).
// n % 64 == 0, k % 64 == 0.
extern "C" int hpc_group_gemm_pertensor_fp8_async(void* y_ptr, const void* x_ptr, const void* w_ptr,
                                                  const void* seqlens_ptr, const void* cu_seqlens_ptr,
                                                  const void* yscale_ptr, const void* row_index_ptr,
                                                  int num_group, int m, int x_rows, int n, int k,
                                                  const void* cu_tiles128_ptr, hipStream_t stream) {
  using namespace hpc::ggemm;
  if (!y_ptr || !x_ptr || !w_ptr || !seqlens_ptr || !cu_seqlens_ptr || !yscale_ptr) return HPC_ERR_INVALID;
  if (num_group <= 0 || n <= 0 || k <= 0 || x_rows <= 0) return HPC_ERR_INVALID;
  if (m <= 0) return HPC_OK;
  if ((n & 63) || (k & 63)) return HPC_ERR_UNSUPPORTED;
  if (static_cast<int64_t>(x_rows) * k > 0xfffffe00ll) return HPC_ERR_UNSUPPORTED;
  Args a;
  a.x = static_cast<const uint8_t*>(x_ptr);
  a.w = static_cast<const uint8_t*>(w_ptr);
  a.xs = static_cast<const float*>(yscale_ptr);  // unused (has_xs = 0), any valid pointer
  a.ws = static_cast<const float*>(yscale_ptr);
  a.y = static_cast<uint16_t*>(y_ptr);
  a.seqlens = static_cast<const int*>(seqlens_ptr);
  a.cu_seqlens = static_cast<const int*>(cu_seqlens_ptr);
  a.row_index = static_cast<const int*>(row_index_ptr);
  a.col_base = nullptr;
  a.N = n;
  a.K = k;
  a.KB = (k + 127) / 128;
  a.tile_m = 16;
  a.ws_group_stride = 1;
  a.ws_ntile_stride = 0;
  a.ws_kb_stride = 0;
  a.has_xs = 0;
  a.x_bytes = static_cast<unsigned>(static_cast<int64_t>(x_rows) * k);
  a.xs_row_stride = 0;
  a.xs_kb_stride = 0;
  return launch_stream_gemm(a, num_group, m, n, cu_tiles128_ptr, stream);
}

// Gate-up GEMM of the per-tensor fused MoE with out = e4m3(silu(gate) * up * act_mul_scale[0]) in its epilogue (256 x 256
// tile kernel; the caller has checked hpc_ggemm_p8_selected, inter % 128 == 0 and k % 128 == 0).  Same arguments as
// hpc_group_gemm_pertensor_fp8_async with n = 2 * inter; writes act_out e4m3 [m, inter].
int hpc_group_gemm_pertensor_fp8_act(void* act_out, const void* x_ptr, const void* w_ptr, const void* seqlens_ptr,
                                     const void* cu_seqlens_ptr, const void* yscale_ptr, const void* row_index_ptr,
                                     const void* act_mul_scale_ptr, int use_bf16_mul, int num_group, int m, int x_rows,
                                     int n, int k, const void* cu_tiles128_ptr, hipStream_t stream) {
  using namespace hpc::ggemm;
  if (static_cast<int64_t>(x_rows) * k > 0xfffffe00ll || static_cast<int64_t>(n) * k > 0xfffffe00ll) return HPC_ERR_UNSUPPORTED;
  Args a;
  a.x = static_cast<const uint8_t*>(x_ptr);
  a.w = static_cast<const uint8_t*>(w_ptr);
  a.xs = static_cast<const float*>(yscale_ptr);
  a.ws = static_cast<const float*>(yscale_ptr);
  a.y = nullptr;
  a.seqlens = static_cast<const int*>(seqlens_ptr);
  a.cu_seqlens = static_cast<const int*>(cu_seqlens_ptr);
  a.row_index = static_cast<const int*>(row_index_ptr);
  a.col_base = nullptr;
  a.N = n;
  a.K = k;
  a.KB = k / 128;
  a.tile_m = 16;
  a.ws_group_stride = 1;
  a.ws_ntile_stride = 0;
  a.ws_kb_stride = 0;
  a.has_xs = 0;
  a.x_bytes = static_cast<unsigned>(static_cast<int64_t>(x_rows) * k);
  a.xs_row_stride = 0;
  a.xs_kb_stride = 0;
  a.act_out = static_cast<uint8_t*>(act_out);
  a.act_mul_scale = static_cast<const float*>(act_mul_scale_ptr);
  a.use_bf16_mul = use_bf16_mul;
  return hpc_ggemm_launch_p8(a, static_cast<const int*>(cu_tiles128_ptr), num_group, m, n, stream);
}

// Gate-up GEMM of the fused MoE with SiLU(gate) * up + 128-block quantisation in its epilogue (256 x 256 tile kernel;
// the caller has checked hpc_ggemm_p8_selected and inter % 128 == 0).  Same arguments as
// hpc_group_gemm_blockwise_fp8_async with n = 2 * inter; writes act_out e4m3 [m, inter] and act_scale f32 [m, inter/128].
int hpc_group_gemm_blockwise_fp8_act(void* act_out, void* act_scale, const void* x_ptr, const void* w_ptr,
                                     const void* seqlens_ptr, const void* cu_seqlens_ptr, const void* xscale_ptr,
                                     const void* wscale_ptr, const void* row_index_ptr, int num_group, int m, int n,
                                     int k, int num_block_k_pad4, int64_t xscale_row_stride, int64_t xscale_kb_stride,
                                     const void* cu_tiles128_ptr, hipStream_t stream) {
  using namespace hpc::ggemm;
  if (static_cast<int64_t>(m) * k > 0xfffffe00ll || static_cast<int64_t>(n) * k > 0xfffffe00ll) return HPC_ERR_UNSUPPORTED;
  Args a;
  a.x = static_cast<const uint8_t*>(x_ptr);
  a.w = static_cast<const uint8_t*>(w_ptr);
  a.xs = static_cast<const float*>(xscale_ptr);
  a.ws = static_cast<const float*>(wscale_ptr);
  a.y = nullptr;
  a.seqlens = static_cast<const int*>(seqlens_ptr);
  a.cu_seqlens = static_cast<const int*>(cu_seqlens_ptr);
  a.row_index = static_cast<const int*>(row_index_ptr);
  a.col_base = nullptr;
  a.N = n;
  a.K = k;
  a.KB = k / 128;
  a.tile_m = 16;
  a.ws_group_stride = (n / 128) * num_block_k_pad4;
  a.ws_ntile_stride = num_block_k_pad4;
  a.ws_kb_stride = 1;
  a.has_xs = 1;
  a.x_bytes = 0xfffffe00u;
  a.xs_row_stride = xscale_row_stride;
  a.xs_kb_stride = xscale_kb_stride;
  a.act_out = static_cast<uint8_t*>(act_out);
  a.act_scale = static_cast<float*>(act_scale);
  return hpc_ggemm_launch_p8(a, static_cast<const int*>(cu_tiles128_ptr), num_group, m, n, stream);
}
