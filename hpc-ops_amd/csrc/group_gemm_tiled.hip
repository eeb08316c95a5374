// Grouped FP8 GEMM, tiled ("throughput") form for large groups - gfx950.
//
// Same contract as group_gemm_blockwise.hip (reference src/group_gemm/kernels.cuh:215-892) but for the
// MFMA-bound regime (hundreds of tokens per expert): a workgroup computes a 128 (weight rows) x 128
// (tokens) output tile; weight and activation k-slabs (128 x 128 B each) are staged through a
// double-buffered LDS tile with full-row global loads issued one k-step ahead (register staging,
// write after the barrier), 2 x 2 waves each own a 64 x 64 sub-tile = 4 x 4 v_mfma_f32_16x16x32_fp8_fp8
// blocks.  Per 128-wide k block the fp32 partial is rescaled by xs*ws (or the per-group scale).
// Tiles are enumerated on the device: blockIdx.y walks the exclusive scan of ceil(seqlens/128).
#include "hpc_common.h"
#include "../../include/hpc_amd.h"
#include "group_gemm.h"

namespace hpc {
namespace ggemm {
namespace {

constexpr int kThreads = 256;
constexpr int kTile = 128;
constexpr int kRow = 128 + 16;  // padded LDS row (bytes)

__device__ __forceinline__ long pack64(uint32_t lo, uint32_t hi) {
  return static_cast<long>(static_cast<uint64_t>(lo) | (static_cast<uint64_t>(hi) << 32));
}

__global__ __launch_bounds__(kThreads, 2) void gemm_fp8_tiled_kernel(const Args a, const int* __restrict__ cu_tiles,
                                                                     int num_group) {
  __shared__ __attribute__((aligned(16))) uint8_t s_w[2][kTile * kRow];
  __shared__ __attribute__((aligned(16))) uint8_t s_x[2][kTile * kRow];

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int r16 = lane & 15, g4 = lane >> 4;
  // which (group, m-tile) is this?  binary search over the tile scan (wave-uniform)
  const cint_ptr cut = as_const(cu_tiles);
  const int tile_id = blockIdx.y;
  if (tile_id >= cut[num_group]) return;
  int lo = 0, hi = num_group;  // first e with cu_tiles[e + 1] > tile_id
  while (lo < hi) {
    const int mid = (lo + hi) >> 1;
    if (cut[mid + 1] <= tile_id) lo = mid + 1; else hi = mid;
  }
  const int e = lo;
  const int m_cnt = as_const(a.seqlens)[e];
  const int m0 = as_const(a.cu_seqlens)[e];
  const int mt0 = (tile_id - cut[e]) * kTile;  // first token slot of this tile
  const int n0 = blockIdx.x * kTile;
  const int K = a.K, KB = a.KB;

  // ---- staging roles: thread -> (row tid/8 + 32*i, 16-byte chunk tid%8), i = 0..3 ------------------
  const int ld_chunk = tid & 7, ld_row = tid >> 3;
  const uint8_t* wsrc = a.w + (static_cast<long>(e) * a.N + n0) * K;
  unsigned xoff[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int slot = mt0 + ld_row + 32 * i;
    const int sc = slot < m_cnt ? slot : m_cnt - 1;
    const int xrow = a.row_index ? a.row_index[m0 + sc] : m0 + sc;
    xoff[i] = static_cast<unsigned>(xrow) * static_cast<unsigned>(K) + ld_chunk * 16;
  }
  const auto rw = make_rsrc(wsrc, static_cast<unsigned>(kTile) * static_cast<unsigned>(K));
  const auto rx = make_rsrc(a.x, a.x_bytes);
  u32x4 wreg[4], xreg[4];
  auto issue = [&](int kb) {
    const int koff = kb * 128;
    const bool ok = koff + ld_chunk * 16 < K;  // K % 64 == 0: chunks past K read as zero
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      wreg[i] = buf_ld16<0>(rw, ok ? (ld_row + 32 * i) * K + ld_chunk * 16 : 0xffffff00u, koff);
      xreg[i] = buf_ld16<0>(rx, ok ? xoff[i] : 0xffffff00u, koff);
    }
  };
  auto stash = [&](int buf) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      *reinterpret_cast<u32x4*>(&s_w[buf][(ld_row + 32 * i) * kRow + ld_chunk * 16]) = wreg[i];
      *reinterpret_cast<u32x4*>(&s_x[buf][(ld_row + 32 * i) * kRow + ld_chunk * 16]) = xreg[i];
    }
  };

  // ---- scales: per lane the 4 token columns (block j, column r16) of its wave's 64-token half ------
  const int wn = wave >> 1, wm = wave & 1;  // wave's 64-row / 64-token quadrant
  long xs_term[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int slot = mt0 + wm * 64 + j * 16 + r16;
    const int sc = slot < m_cnt ? slot : m_cnt - 1;
    const long col0 = a.col_base ? static_cast<long>(as_const(a.col_base)[e]) * a.tile_m : 0;
    xs_term[j] = (a.col_base ? col0 + sc : static_cast<long>(a.row_index ? a.row_index[m0 + sc] : m0 + sc)) *
                 a.xs_row_stride;
  }
  const cint_ptr ws_row = as_const(reinterpret_cast<const int*>(a.ws)) +
                          static_cast<long>(e) * a.ws_group_stride + (n0 >> 7) * a.ws_ntile_stride;

  f32x4 tot[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) tot[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  issue(0);
  stash(0);
  __syncthreads();
  for (int kb = 0; kb < KB; ++kb) {
    const int buf = kb & 1;
    if (kb + 1 < KB) issue(kb + 1);
    float f[4];
    const float wsk = __int_as_float(ws_row[kb * a.ws_kb_stride]);
#pragma unroll
    for (int j = 0; j < 4; ++j) f[j] = a.has_xs ? a.xs[xs_term[j] + kb * a.xs_kb_stride] * wsk : wsk;

    f32x4 part[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j) part[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int c = 0; c < 2; ++c) {  // two 16-byte chunks per lane = four k-steps of 32
      u32x4 af[4], bf[4];
#pragma unroll
      for (int i = 0; i < 4; ++i)
        af[i] = *reinterpret_cast<const u32x4*>(&s_w[buf][(wn * 64 + i * 16 + r16) * kRow + (g4 + 4 * c) * 16]);
#pragma unroll
      for (int j = 0; j < 4; ++j)
        bf[j] = *reinterpret_cast<const u32x4*>(&s_x[buf][(wm * 64 + j * 16 + r16) * kRow + (g4 + 4 * c) * 16]);
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          part[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_fp8_fp8(pack64(af[i][0], af[i][1]),
                                                                 pack64(bf[j][0], bf[j][1]), part[i][j], 0, 0, 0);
          part[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_fp8_fp8(pack64(af[i][2], af[i][3]),
                                                                 pack64(bf[j][2], bf[j][3]), part[i][j], 0, 0, 0);
        }
    }
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int r = 0; r < 4; ++r) tot[i][j][r] = fmaf(part[i][j][r], f[j], tot[i][j][r]);
    if (kb + 1 < KB) stash(buf ^ 1);
    __syncthreads();
  }

  // ---- epilogue: lane holds rows n = i*16 + g4*4 + r of token column j*16 + r16 -----------------------
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int slot = mt0 + wm * 64 + j * 16 + r16;
    if (slot < m_cnt) {
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        u32x2 pk;
        pk[0] = pack_bf16x2(tot[i][j][0], tot[i][j][1]);
        pk[1] = pack_bf16x2(tot[i][j][2], tot[i][j][3]);
        *reinterpret_cast<u32x2*>(a.y + static_cast<long>(m0 + slot) * a.N + n0 + wn * 64 + i * 16 + g4 * 4) = pk;
      }
    }
  }
}

}  // namespace
}  // namespace ggemm
}  // namespace hpc

int hpc_ggemm_launch_tiled(const hpc::ggemm::Args& a, const int* cu_tiles, int num_group, int m, int n,
                           hipStream_t stream) {
  using namespace hpc::ggemm;
  if (n % kTile) return HPC_ERR_UNSUPPORTED;
  const int max_tiles = m / kTile + num_group;  // upper bound of sum_g ceil(len_g / 128)
  dim3 grid(n / kTile, max_tiles);
  gemm_fp8_tiled_kernel<<<grid, kThreads, 0, stream>>>(a, cu_tiles, num_group);
  HPC_CHECK_LAUNCH();
  return HPC_OK;
}
