// Grouped FP8 GEMM, 256 x 128 tile with an LDS-DMA ring ("throughput" form, second generation) - gfx950.
//
// Same contract as group_gemm_tiled.hip / group_gemm_blockwise.hip (reference
// src/group_gemm/kernels.cuh:215-892).  The 128 x 128 register-staged kernel tops out near 0.6 PFLOP/s:
// one k-step of prefetch cannot cover HBM/L2 latency and a deeper register ring does not fit next to
// 128 accumulator registers.  Here a workgroup of 8 waves (4 x 2, 64 x 64 outputs per wave) owns a
// 256 (weight rows) x 128 (tokens) tile and streams 128-byte k-slabs of W and X straight into a 3-deep
// LDS ring with `buffer_load ... lds` (16 bytes per lane, no staging registers, no ds_write pass):
//   * the LDS image of a piece is lane-linear (HW rule), so the XOR swizzle that makes the MFMA operand
//     reads conflict-free is applied on the SOURCE side: lane l of an 8-row piece fetches chunk
//     (l % 8) ^ (l / 8) of row l / 8, and a reader asks for position chunk ^ (row & 7);
//   * two slabs stay in flight across the single barrier of a k-step: counted `s_waitcnt vmcnt(N)` and a
//     raw `s_barrier` (a __syncthreads() would drain the DMA queue);
//   * a 16 x 16 output block takes ONE v_mfma_f32_16x16x128_f8f6f4 per slab (plain fp8 x fp8: the only
//     fp8 MFMA form that runs at the 5 PFLOP/s rate; 16x16x32_fp8_fp8 runs at the bf16 rate);
//   * the per-token activation scales of the slab ride in the same ring (4-byte DMA), the weight-block
//     scale is a scalar load; fp32 partial of a 128-k block is rescaled into the running sum as before;
//   * DMA issue is spread over the k-step and the instruction order pinned (see the pipeline comment).
#include "hpc_common.h"
#include "hpc_dev.h"
#include "../../include/hpc_amd.h"
#include "group_gemm.h"

namespace hpc {
namespace ggemm {
namespace {

constexpr int kThreads = 512;
constexpr int kBN = 256, kBM = 128, kBK = 128;
constexpr int kWBytes = kBN * kBK, kXsBytes = 8 * 256;

#ifdef HPC_TILED256_PROFILE
// development build only (tools/prof_tiled256.sh): per-segment s_memtime sums of waves 0 and 7
__device__ unsigned long long g_t256_prof[16];
#define T256_STAMP(x) const unsigned long long x = __builtin_readcyclecounter()
#define T256_ACC(i, v) prof[i] += (v)
#else
#define T256_STAMP(x)
#define T256_ACC(i, v)
#endif

typedef __attribute__((address_space(3))) void lds_void;

__device__ __forceinline__ long pack64(uint32_t lo, uint32_t hi) {
  return static_cast<long>(static_cast<uint64_t>(lo) | (static_cast<uint64_t>(hi) << 32));
}

// Per-token-tile-size constants (namespace scope on purpose: the kernel's lambdas use them, and hipcc's
// host pass drops the kernel stub when a lambda in a templated __global__ refers to function-local
// constexpr values derived from the template arguments).
template <int kTok>
struct TileCfg {
  static constexpr int kSub = kBM / kTok;            // token tiles per 128-token tile of the scan
  static constexpr int kWM = kTok == 128 ? 2 : 1;    // waves along tokens
  static constexpr int kWN = 8 / kWM;                // waves along weight rows
  static constexpr int kIN = kBN / kWN / 16;         // 16-row blocks per wave
  static constexpr int kJN = kTok / kWM / 16;        // 16-token blocks per wave
  static constexpr int kXBytes = kTok * kBK;
  static constexpr int kStages = kTok >= 64 ? 3 : 4;
  static constexpr int kStageBytes = kWBytes + kXBytes + kXsBytes;
  static constexpr int kXQ = kTok == 128 ? 2 : 1;    // activation DMA pieces per wave (8 rows each)
  static constexpr int kXPieces = kTok / 8;
};

// kTok = tokens per tile: 128 (the form in use: 4 x 2 waves of 64 x 64, 3-slab ring) or 32 (experimental:
// 8 x 1 waves of 32 x 32, a 4 KB activation slab and FOUR slabs in the ring; extra token tiles of a group
// re-read the weight tile from the XCD's L2 - measured slower, see the launcher).
template <bool kHasXs, int kTok>
__global__ __launch_bounds__(kThreads, 1) void gemm_fp8_tiled256_kernel(const Args a, const int* __restrict__ cu_tiles,
                                                                        int num_group) {
  using C = TileCfg<kTok>;
  __shared__ __attribute__((aligned(1024))) uint8_t s_ring[C::kStages * C::kStageBytes];
  constexpr int kDmaPerStage = 4 + C::kXQ + (kHasXs ? 1 : 0);  // per wave

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int r16 = lane & 15, g4 = lane >> 4;
  // ---- which (group, token tile, weight tile)?  XCD-aware order -------------------------------------------
  // Work items are ordered group -> weight tile -> token tile, so the token tiles that share a weight
  // tile are neighbours; hardware deals consecutive workgroup ids round-robin over the 8 XCDs, so id L
  // runs item (L % 8) * chunk + L / 8: each XCD walks one contiguous eighth of the list and the 2-4 token
  // tiles of a weight tile (and the group's activation tiles) meet in ONE L2 instead of eight.
  const cint_ptr cut = as_const(cu_tiles);
  const int nt = a.N / kBN;
  const int total = cut[num_group] * nt * C::kSub;
  const int chunk = (total + 7) >> 3;
  const int lin = blockIdx.x;
  const int item = (lin & 7) * chunk + (lin >> 3);
  if ((lin >> 3) >= chunk || item >= total) return;
  int lo = 0, hi = num_group;  // first e with cu_tiles[e + 1] * nt > item
  while (lo < hi) {
    const int mid = (lo + hi) >> 1;
    if (cut[mid + 1] * nt * C::kSub <= item) lo = mid + 1; else hi = mid;
  }
  const int e = lo;
  const int mtiles = (cut[e + 1] - cut[e]) * C::kSub;
  const int rem = item - cut[e] * nt * C::kSub;
  const int m_cnt = as_const(a.seqlens)[e];
  const int m0 = as_const(a.cu_seqlens)[e];
  const int mt0 = (rem % mtiles) * kTok;
  if (mt0 >= m_cnt) return;  // empty token sub-tile of the last 128-token tile (workgroup-uniform)
  const int n0 = (rem / mtiles) * kBN;
  const int K = a.K, KB = a.KB;
  const int wn = wave / C::kWM, wm = wave % C::kWM;

  // ---- DMA roles ----------------------------------------------------------------------------------------
  // 8-row pieces (1 KB): lane -> row piece*8 + lane/8, LDS position lane%8 <- global chunk (lane%8) ^ (lane/8)
  const int p_row = lane >> 3, p_chunk = (lane & 7) ^ (lane >> 3);
  const uint8_t* wsrc = a.w + (static_cast<long>(e) * a.N + n0) * K;
  const unsigned w_bytes = static_cast<unsigned>(kBN) * static_cast<unsigned>(K);
  unsigned w_voff[4], x_voff[2] = {0u, 0u};
#pragma unroll
  for (int q = 0; q < 4; ++q) w_voff[q] = static_cast<unsigned>((wave * 4 + q) * 8 + p_row) * K + p_chunk * 16;
#pragma unroll
  for (int q = 0; q < C::kXQ; ++q) {
    const int slot = mt0 + ((wave * C::kXQ + q) % C::kXPieces) * 8 + p_row;
    const int sc = slot < m_cnt ? slot : m_cnt - 1;
    const int xrow = a.row_index ? a.row_index[m0 + sc] : m0 + sc;
    x_voff[q] = static_cast<unsigned>(xrow) * static_cast<unsigned>(K) + p_chunk * 16;
  }
  unsigned xs_voff = 0;
  if constexpr (kHasXs) {
    const int slot = mt0 + (wave & 1) * 64 + lane;
    const int sc = slot < m_cnt ? slot : m_cnt - 1;
    const long col0 = a.col_base ? static_cast<long>(as_const(a.col_base)[e]) * a.tile_m : 0;
    const long term = a.col_base ? col0 + sc : static_cast<long>(a.row_index ? a.row_index[m0 + sc] : m0 + sc);
    xs_voff = static_cast<unsigned>(term * a.xs_row_stride * 4);
  }
  const int xs_kb_bytes = static_cast<int>(a.xs_kb_stride * 4);

  // k-slab st -> ring slot st % C::kStages; past the end: empty descriptors (zeros).  The weight and the
  // activation DMAs of a slab are issued half a k-step apart (see the pipeline below).
  auto issue_w = [&](int st, int q) {  // piece q (of 4) of this wave's share of the weight slab
    const bool on = st < KB;
    const int koff = st * kBK;
    uint8_t* base = s_ring + (st % C::kStages) * C::kStageBytes;
    const auto rw = make_rsrc(wsrc, on ? w_bytes : 0u);
    // K % 128 == 64 (per-tensor scales): chunks past K get an out-of-range offset and land as zeros
    const bool k_ok = koff + p_chunk * 16 < K;
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rw, (lds_void*)(base + (wave * 4 + q) * 1024), 16,
                                             k_ok ? w_voff[q] : 0xffffff00u, koff, 0, 0);
  };
  auto issue_x = [&](int st) {
    const bool on = st < KB;
    const int koff = st * kBK;
    uint8_t* base = s_ring + (st % C::kStages) * C::kStageBytes;
    const bool k_ok = koff + p_chunk * 16 < K;
#pragma unroll
    for (int q = 0; q < 2; ++q) {
      if (q >= C::kXQ) break;
      // every wave issues the same number of DMAs (the vmcnt arithmetic needs that); waves without an
      // activation piece (kTok = 32: waves 4..7) fetch nothing into the unused half of the scale slots
      const int piece = wave * C::kXQ + q;
      const bool real = piece < C::kXPieces;
      const auto rxq = make_rsrc(a.x, on && real ? a.x_bytes : 0u);
      const int dst_off = real ? kWBytes + piece * 1024 : kWBytes + C::kXBytes + 1024;
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rxq, (lds_void*)(base + dst_off), 16, k_ok ? x_voff[q] : 0xffffff00u,
                                               koff, 0, 0);
    }
    if constexpr (kHasXs) {
      // readfirstlane: hipcc otherwise treats the record count as divergent and wraps the DMA in a waterfall loop
      const auto rs = make_rsrc(a.xs, __builtin_amdgcn_readfirstlane(on && wave < 2 && wave * 64 < kTok ? 0xffffffffu : 0u));
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lds_void*)(base + kWBytes + C::kXBytes + wave * 256), 4, xs_voff,
                                           st * xs_kb_bytes, 0, 0);
    }
  };

  const cint_ptr ws_row = as_const(reinterpret_cast<const int*>(a.ws)) + static_cast<long>(e) * a.ws_group_stride +
                          ((n0 + wn * 16 * C::kIN) >> 7) * a.ws_ntile_stride;

  f32x4 tot[C::kIN][C::kJN];
#pragma unroll
  for (int i = 0; i < C::kIN; ++i)
#pragma unroll
    for (int j = 0; j < C::kJN; ++j) tot[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  // operand read offsets inside a slab (swizzled): row r, chunk c -> r*128 + ((c ^ (r & 7)) << 4)
  int a_off[C::kIN][2], b_off[C::kJN][2];
#pragma unroll
  for (int c = 0; c < 2; ++c) {
#pragma unroll
    for (int i = 0; i < C::kIN; ++i) {
      const int ra = wn * 16 * C::kIN + i * 16 + r16;
      a_off[i][c] = ra * kBK + (((g4 + 4 * c) ^ (ra & 7)) << 4);
    }
#pragma unroll
    for (int j = 0; j < C::kJN; ++j) {
      const int rb = wm * 16 * C::kJN + j * 16 + r16;
      b_off[j][c] = kWBytes + rb * kBK + (((g4 + 4 * c) ^ (rb & 7)) << 4);
    }
  }

  // ---- software pipeline --------------------------------------------------------------------------------
  // One v_mfma_f32_16x16x128_f8f6f4 (plain fp8 x fp8, no MX scales) covers the whole 128-byte k-slab of a
  // 16 x 16 output block at TWICE the rate of four 16x16x32 fp8 MFMAs.  A lane supplies 32 bytes of its
  // row: the two 16-byte chunks g4 and g4 + 4 (any assignment of k-bytes to lane slots is fine as long as
  // both operands use the same one - a dot product does not care about the order of its terms).
  // A k-step is split over the wave's weight-row blocks: the first half multiplies blocks [0, kIN/2) while
  // the second half's weight fragments are read; the single barrier sits between the halves - past it
  // every wave has finished reading slab kb (its slot is refilled with slab kb+C::kStages) and slab kb+1
  // is visible, so the second half runs under the reads of the next slab's token fragments (double
  // buffered: the current ones are still in use) and first-half weight fragments.
  constexpr int kH = C::kIN / 2;
  auto read_a = [&](const uint8_t* slab, int i0, u32x4 (&af)[kH][2]) {
#pragma unroll
    for (int i = 0; i < kH; ++i)
#pragma unroll
      for (int c = 0; c < 2; ++c) af[i][c] = *reinterpret_cast<const u32x4*>(slab + a_off[i0 + i][c]);
  };
  auto read_b = [&](const uint8_t* slab, u32x4 (&bf)[C::kJN][2]) {
#pragma unroll
    for (int j = 0; j < C::kJN; ++j)
#pragma unroll
      for (int c = 0; c < 2; ++c) bf[j][c] = *reinterpret_cast<const u32x4*>(slab + b_off[j][c]);
  };
  auto mma_half = [&](int i0, const u32x4 (&af)[kH][2], const u32x4 (&bf)[C::kJN][2], const float (&f)[C::kJN]) {
#pragma unroll
    for (int i = 0; i < kH; ++i)
#pragma unroll
      for (int j = 0; j < C::kJN; ++j) {
        const i32x8 av = {static_cast<int>(af[i][0][0]), static_cast<int>(af[i][0][1]), static_cast<int>(af[i][0][2]),
                          static_cast<int>(af[i][0][3]), static_cast<int>(af[i][1][0]), static_cast<int>(af[i][1][1]),
                          static_cast<int>(af[i][1][2]), static_cast<int>(af[i][1][3])};
        const i32x8 bv = {static_cast<int>(bf[j][0][0]), static_cast<int>(bf[j][0][1]), static_cast<int>(bf[j][0][2]),
                          static_cast<int>(bf[j][0][3]), static_cast<int>(bf[j][1][0]), static_cast<int>(bf[j][1][1]),
                          static_cast<int>(bf[j][1][2]), static_cast<int>(bf[j][1][3])};
        if constexpr (kHasXs) {
          // fp32 partial of the 128-k block, rescaled into the running sum (reference kernels.cuh:473-476)
          const f32x4 part = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(av, bv, f32x4{0.f, 0.f, 0.f, 0.f}, 0, 0, 0,
                                                                              0, 0, 0);
#pragma unroll
          for (int r = 0; r < 4; ++r) tot[i0 + i][j][r] = fmaf(part[r], f[j], tot[i0 + i][j][r]);
        } else {
          // one scale per group: accumulate straight into the running sum, scale once in the epilogue
          // (the reference scales every k-tile: same value up to fp32 rounding)
          tot[i0 + i][j] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(av, bv, tot[i0 + i][j], 0, 0, 0, 0, 0, 0);
        }
      }
  };
  // DMA issue is spread over the k-step - eight waves firing all their slab DMAs right behind the barrier
  // queue up in the texture path, and the MFMAs behind them in program order starve: the weight DMAs of
  // slab kb+kStages go out between the second half's MFMAs (their slot is free past the barrier), its
  // activation (+ scale) DMAs between the first half's MFMAs of the next k-step.
  // s_waitcnt immediates: vmcnt = n (split 4 + 2 bits), expcnt untouched, lgkmcnt untouched (0xF) or 0
  constexpr int kDmaX = C::kXQ + (kHasXs ? 1 : 0);
  constexpr int kVmFirst = kDmaPerStage * (C::kStages - 1) - kDmaX, kVmLoop = kDmaPerStage * (C::kStages - 2);
  constexpr int kWaitFirst = 0x0F70 | (kVmFirst & 15) | ((kVmFirst >> 4) << 14);
  constexpr int kWaitLoop = 0x0070 | (kVmLoop & 15) | ((kVmLoop >> 4) << 14);

#pragma unroll
  for (int st = 0; st < C::kStages; ++st) {
#pragma unroll
    for (int q = 0; q < 4; ++q) issue_w(st, q);
    if (st + 1 < C::kStages) issue_x(st);  // the last slab's activation part goes out in the first k-step
  }
  __builtin_amdgcn_s_waitcnt(kWaitFirst);  // slab 0 has landed
  __builtin_amdgcn_s_barrier();
  u32x4 a_lo[kH][2], a_hi[kH][2], b_even[C::kJN][2], b_odd[C::kJN][2];
  read_a(s_ring, 0, a_lo);
  read_b(s_ring, b_even);

#ifdef HPC_TILED256_PROFILE
  unsigned long long prof[5] = {0, 0, 0, 0, 0};
#endif
  auto k_step = [&](int kb, const u32x4 (&b_cur)[C::kJN][2], u32x4 (&b_nxt)[C::kJN][2]) {
    T256_STAMP(t0);
    const uint8_t* slab = s_ring + (kb % C::kStages) * C::kStageBytes;
    const uint8_t* next = s_ring + ((kb + 1) % C::kStages) * C::kStageBytes;
    read_a(slab, kH, a_hi);
    float f[C::kJN];
#pragma unroll
    for (int j = 0; j < C::kJN; ++j) f[j] = 1.f;
    if constexpr (kHasXs) {
      const float wsk = __int_as_float(ws_row[kb * a.ws_kb_stride]);
#pragma unroll
      for (int j = 0; j < C::kJN; ++j)
        f[j] = wsk * *reinterpret_cast<const float*>(slab + kWBytes + C::kXBytes + (wm * 16 * C::kJN + j * 16 + r16) * 4);
    }
    issue_x(kb + C::kStages - 1);
    mma_half(0, a_lo, b_cur, f);
    // order: the LDS reads, then MFMAs with one DMA between every two of them
    __builtin_amdgcn_sched_group_barrier(0x100, kH * 2 + C::kJN, 0);
#pragma unroll
    for (int q = 0; q < kDmaX; ++q) {
      __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);
      __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
    }
    __builtin_amdgcn_sched_barrier(0);
    // slab kb+1 has landed when only the younger slabs are outstanding; lgkmcnt(0): my reads of slab kb are done
    T256_STAMP(t1);
    __builtin_amdgcn_s_waitcnt(kWaitLoop);
    T256_STAMP(t2);
    __builtin_amdgcn_s_barrier();
    T256_STAMP(t3);
    __builtin_amdgcn_sched_barrier(0);
    // program order = wanted order (a DMA writes LDS, so hipcc keeps it in place relative to the LDS reads):
    // four rounds of one weight DMA + a quarter of the next slab's fragment reads, an MFMA between rounds
    constexpr int kBPer = C::kJN * 2 / 4;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      issue_w(kb + C::kStages, q);
      if (q < kH * 2) a_lo[q >> 1][q & 1] = *reinterpret_cast<const u32x4*>(next + a_off[q >> 1][q & 1]);
#pragma unroll
      for (int t = q * kBPer; t < (q + 1) * kBPer; ++t)
        b_nxt[t >> 1][t & 1] = *reinterpret_cast<const u32x4*>(next + b_off[t >> 1][t & 1]);
    }
    mma_half(kH, a_hi, b_cur, f);
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
      __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
      __builtin_amdgcn_sched_group_barrier(0x100, (kH * 2 + C::kJN * 2 + 3) / 4, 0);
    }
    __builtin_amdgcn_sched_barrier(0);
    T256_STAMP(t4);
    T256_ACC(0, t1 - t0); T256_ACC(1, t2 - t1); T256_ACC(2, t3 - t2); T256_ACC(3, t4 - t3); T256_ACC(4, 1);
  };
  for (int kb = 0; kb < KB; kb += 2) {
    k_step(kb, b_even, b_odd);
    if (kb + 1 < KB) k_step(kb + 1, b_odd, b_even);
  }
#ifdef HPC_TILED256_PROFILE
  if (lane == 0 && (wave == 0 || wave == 7))
    for (int i = 0; i < 5; ++i) atomicAdd(&g_t256_prof[(wave ? 8 : 0) + i], prof[i]);
#endif
  __builtin_amdgcn_s_waitcnt(0x0F70);  // drain the (empty) tail DMAs before the workgroup's LDS is released

  if constexpr (!kHasXs) {
    const float gs = __int_as_float(ws_row[0]);  // per-tensor form: strides are zero, one scale per group
#pragma unroll
    for (int i = 0; i < C::kIN; ++i)
#pragma unroll
      for (int j = 0; j < C::kJN; ++j) tot[i][j] *= gs;
  }
  // ---- epilogue: lane holds rows n = wn*16*C::kIN + i*16 + g4*4 + r of token column wm*16*C::kJN + j*16 + r16 --------
#pragma unroll
  for (int j = 0; j < C::kJN; ++j) {
    const int slot = mt0 + wm * 16 * C::kJN + j * 16 + r16;
    if (slot < m_cnt) {
#pragma unroll
      for (int i = 0; i < C::kIN; ++i) {
        u32x2 pk;
        pk[0] = pack_bf16x2(tot[i][j][0], tot[i][j][1]);
        pk[1] = pack_bf16x2(tot[i][j][2], tot[i][j][3]);
        *reinterpret_cast<u32x2*>(a.y + static_cast<long>(m0 + slot) * a.N + n0 + wn * 16 * C::kIN + i * 16 + g4 * 4) = pk;
      }
    }
  }
}

}  // namespace
}  // namespace ggemm
}  // namespace hpc

#ifdef HPC_TILED256_PROFILE
extern "C" int hpc_debug_tiled256_prof(unsigned long long* out16, int reset) {
  using namespace hpc::ggemm;
  if (hipDeviceSynchronize() != hipSuccess) return HPC_ERR_LAUNCH;
  if (hipMemcpyFromSymbol(out16, HIP_SYMBOL(g_t256_prof), sizeof(unsigned long long) * 16) != hipSuccess) return HPC_ERR_LAUNCH;
  if (reset) {
    unsigned long long z[16] = {};
    if (hipMemcpyToSymbol(HIP_SYMBOL(g_t256_prof), z, sizeof(z)) != hipSuccess) return HPC_ERR_LAUNCH;
  }
  return HPC_OK;
}
#endif

int hpc_ggemm_launch_tiled256(const hpc::ggemm::Args& a, const int* cu_tiles, int num_group, int m, int n,
                              hipStream_t stream) {
  using namespace hpc::ggemm;
  if (n % kBN || a.K < kBK) return HPC_ERR_UNSUPPORTED;
  // The 32-token-tile form (4-slab ring: three weight slabs in flight) was meant for 40-128 tokens per
  // group; measured it is SLOWER (E64: 3.6 ms vs 2.2 ms at T = 256 .. 768) - its extra token tiles re-read
  // the weight tile through L2 and do a quarter of the MFMA work per slab - so it only runs on request.
  const bool narrow = hpc_dev_tuning_get(6) == 2;
  // 64-token tiles (8 x 1 waves of 32 x 64, 42 KB per slab, three slabs in the ring) - also measured SLOWER than
  // the 128-token tile on groups of 24-64 rows (E64: T = 256 2.19 vs 1.73 ms, T = 384 2.25 vs 1.75 ms): on request only
  const bool half = hpc_dev_tuning_get(6) == 3;
  const long max_tiles = m / kBM + num_group;  // upper bound of sum_g ceil(len_g / 128)
  const long items = max_tiles * (n / kBN) * (narrow ? 4 : (half ? 2 : 1)) + 8;  // + 8: the per-XCD chunks round up
  if (items > 0x7fffffffl) return HPC_ERR_UNSUPPORTED;
  dim3 grid(static_cast<unsigned>(items));
  if (a.has_xs) {
    if (narrow) gemm_fp8_tiled256_kernel<true, 32><<<grid, kThreads, 0, stream>>>(a, cu_tiles, num_group);
    else if (half) gemm_fp8_tiled256_kernel<true, 64><<<grid, kThreads, 0, stream>>>(a, cu_tiles, num_group);
    else gemm_fp8_tiled256_kernel<true, 128><<<grid, kThreads, 0, stream>>>(a, cu_tiles, num_group);
  } else {
    if (narrow) gemm_fp8_tiled256_kernel<false, 32><<<grid, kThreads, 0, stream>>>(a, cu_tiles, num_group);
    else if (half) gemm_fp8_tiled256_kernel<false, 64><<<grid, kThreads, 0, stream>>>(a, cu_tiles, num_group);
    else gemm_fp8_tiled256_kernel<false, 128><<<grid, kThreads, 0, stream>>>(a, cu_tiles, num_group);
  }
  HPC_CHECK_LAUNCH();
  return HPC_OK;
}
