// Grouped FP8 GEMM with 128-block scales, weight-streaming ("decode / low-latency") form - gfx950.
//
//   Y[m, n] = bf16( sum_kb ( sum_{k in kb} X[m,k] W[g,n,k] ) * xs[m,kb] * ws[g, n/128, kb] )
// for the rows m of group (expert) g.  Replaces reference src/group_gemm/kernels.cuh:532-892
// (group_gemm_blockwise_fp8_kernel), group_gemm_blockwise_fp8.cu:368-457 and the gather-free
// "scatter-A" variant cp_async/group_gemm_fp8_scatter.cu:70-302 (here: optional row_index).
//
// MI355X design: at decode batch sizes every expert sees a handful of tokens, so the op is a
// stream of the weights (HBM-bound).  Same swap as the reference - weights on the
// MFMA M axis, tokens on N (v_mfma_f32_16x16x32_fp8_fp8, 16 tokens per pass) - but no TMA /
// warp specialisation: each WAVE owns 32 weight rows of one expert and streams them HBM -> VGPR
// with 16-byte non-temporal buffer loads, 8 k-blocks (8 KB) deep, no LDS, no barriers.  The few
// activation rows come from L2 at the same prefetch depth (vmcnt retires in order, so a shallower
// X pipeline would drain the weight stream).  Per 128-wide k block the fp32 partial is rescaled by
// xs*ws and accumulated (reference kernels.cuh:806-836).  8 waves/CU x 16 KB of weights in flight.
#include "hpc_common.h"
#include "../../include/hpc_amd.h"

namespace hpc {
namespace ggemm {

struct Args {
  const uint8_t* x;       // [rows, K] e4m3
  const uint8_t* w;       // [G, N, K] e4m3
  const float* xs;        // activation scales, see xs_row_stride / xs_kb_stride
  const float* ws;        // [G, N/128, ws_ld]
  uint16_t* y;            // [M, N] bf16
  const int* seqlens;     // [G]
  const int* cu_seqlens;  // [G]   first row of group g in y (and in x when row_index == null)
  const int* row_index;   // null, or [M] -> row of x / xs for output row m (gather-free MoE)
  const int* col_base;    // null, or [G] (cu_tiles): transposed xs, column = col_base[g]*tile_m + slot
  int N, K, KB, ws_ld, tile_m;
  long xs_row_stride, xs_kb_stride;  // in floats
};

constexpr int kThreads = 256;
constexpr int kDepth = 4;  // k-blocks (4 KB of weights each) in flight per wave

__device__ __forceinline__ long pack64(uint32_t lo, uint32_t hi) {
  return static_cast<long>(static_cast<uint64_t>(lo) | (static_cast<uint64_t>(hi) << 32));
}

__global__ __launch_bounds__(kThreads, 2) void gemm_blockwise_stream_kernel(const Args a) {
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int r16 = lane & 15, g4 = lane >> 4;
  const int e = blockIdx.y;
  const int m_cnt = as_const(a.seqlens)[e];
  if (m_cnt <= 0) return;
  const int n0 = (blockIdx.x * 4 + wave) * 32;
  if (n0 >= a.N) return;
  const int m0 = as_const(a.cu_seqlens)[e];
  const int K = a.K, KB = a.KB;

  const uint8_t* wbase = a.w + (static_cast<long>(e) * a.N + n0) * K;
  const unsigned w_bytes = 16u * static_cast<unsigned>(K);
  const int w_voff = r16 * K + g4 * 16;
  const cint_ptr ws_row = as_const(reinterpret_cast<const int*>(a.ws)) +
                          (static_cast<long>(e) * (a.N >> 7) + (n0 >> 7)) * a.ws_ld;

  const int npass = (m_cnt + 15) >> 4;
  for (int p = 0; p < npass; ++p) {
    const int slot = p * 16 + r16;
    const bool valid = slot < m_cnt;
    const int sc = valid ? slot : 0;
    const int xrow = a.row_index ? a.row_index[m0 + sc] : m0 + sc;
    const unsigned x_voff = static_cast<unsigned>(xrow) * static_cast<unsigned>(K) + g4 * 16;
    const long xs_term = a.col_base ? static_cast<long>(as_const(a.col_base)[e]) * a.tile_m + sc
                                    : static_cast<long>(xrow);
    const unsigned xs_voff = static_cast<unsigned>(xs_term * a.xs_row_stride * 4);
    const int xs_kb_bytes = static_cast<int>(a.xs_kb_stride * 4);

    u32x4 wb[kDepth][2][2];
    u32x4 xb[kDepth][2];
    float xsb[kDepth];
    auto issue = [&](int d, int kn) {
      const unsigned on = kn < KB ? 1u : 0u;
      const int koff = kn * 128;
#pragma unroll
      for (int rb = 0; rb < 2; ++rb) {
        const auto rw = make_rsrc(wbase + static_cast<long>(rb) * 16 * K, on ? w_bytes : 0u);
        wb[d][rb][0] = buf_ld16<2>(rw, w_voff, koff);
        wb[d][rb][1] = buf_ld16<2>(rw, w_voff + 64, koff);
      }
      const auto rx = make_rsrc(a.x, on ? 0xffffffffu : 0u);
      xb[d][0] = buf_ld16<0>(rx, x_voff, koff);
      xb[d][1] = buf_ld16<0>(rx, x_voff + 64, koff);
      const auto rs = make_rsrc(a.xs, on ? 0xffffffffu : 0u);
      xsb[d] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rs, xs_voff, kn * xs_kb_bytes, 0));
    };

    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int d = 0; d < kDepth; ++d) {
      issue(d, d);
      __builtin_amdgcn_sched_barrier(0);  // keep issue order = consumption order (in-order vmcnt)
    }

    f32x4 tot[2];
    tot[0] = tot[1] = f32x4{0.f, 0.f, 0.f, 0.f};
    for (int kb0 = 0; kb0 < KB; kb0 += kDepth) {
#pragma unroll
      for (int d = 0; d < kDepth; ++d) {
        const int kb = kb0 + d;
        const int kbc = kb < KB ? kb : KB - 1;
        const float wsk = __int_as_float(ws_row[kbc]);
        const float f = xsb[d] * wsk;  // loads past K return 0 -> contribute nothing
#pragma unroll
        for (int rb = 0; rb < 2; ++rb) {
          f32x4 part = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
          for (int h = 0; h < 2; ++h) {
            part = __builtin_amdgcn_mfma_f32_16x16x32_fp8_fp8(
                pack64(wb[d][rb][h][0], wb[d][rb][h][1]), pack64(xb[d][h][0], xb[d][h][1]), part, 0, 0, 0);
            part = __builtin_amdgcn_mfma_f32_16x16x32_fp8_fp8(
                pack64(wb[d][rb][h][2], wb[d][rb][h][3]), pack64(xb[d][h][2], xb[d][h][3]), part, 0, 0, 0);
          }
#pragma unroll
          for (int i = 0; i < 4; ++i) tot[rb][i] = fmaf(part[i], f, tot[rb][i]);
        }
        issue(d, kb + kDepth);
      }
    }

    if (valid) {
#pragma unroll
      for (int rb = 0; rb < 2; ++rb) {
        u32x2 pk;
        pk[0] = pack_bf16x2(tot[rb][0], tot[rb][1]);
        pk[1] = pack_bf16x2(tot[rb][2], tot[rb][3]);
        *reinterpret_cast<u32x2*>(a.y + static_cast<long>(m0 + slot) * a.N + n0 + rb * 16 + g4 * 4) = pk;
      }
    }
    __builtin_amdgcn_s_waitcnt(0x0F70);  // retire stores before the next pass (see attention_decode.hip)
  }
}

}  // namespace ggemm
}  // namespace hpc

extern "C" int hpc_group_gemm_blockwise_fp8_async(
    void* y_ptr, const void* x_ptr, const void* w_ptr, const void* seqlens_ptr,
    const void* cu_seqlens_ptr, const void* xscale_ptr, const void* wscale_ptr,
    const void* row_index_ptr, const void* col_base_ptr, int num_group, int m, int n, int k,
    int num_block_k_pad4, int tile_m, int64_t xscale_row_stride, int64_t xscale_kb_stride,
    hipStream_t stream) {
  using namespace hpc::ggemm;
  if (!y_ptr || !x_ptr || !w_ptr || !seqlens_ptr || !cu_seqlens_ptr || !xscale_ptr || !wscale_ptr)
    return HPC_ERR_INVALID;
  if (num_group <= 0 || n <= 0 || k <= 0) return HPC_ERR_INVALID;
  if (m <= 0) return HPC_OK;
  if ((n & 127) || (k & 127)) return HPC_ERR_UNSUPPORTED;  // 128x128 weight scale blocks
  if (num_block_k_pad4 < k / 128) return HPC_ERR_INVALID;
  if (static_cast<int64_t>(m) * k > 0xffffffffll) return HPC_ERR_UNSUPPORTED;  // 32-bit x offsets
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
  a.ws_ld = num_block_k_pad4;
  a.tile_m = tile_m;
  a.xs_row_stride = xscale_row_stride;
  a.xs_kb_stride = xscale_kb_stride;
  dim3 grid((n + 127) / 128, num_group);
  gemm_blockwise_stream_kernel<<<grid, kThreads, 0, stream>>>(a);
  HPC_CHECK_LAUNCH();
  return HPC_OK;
}
