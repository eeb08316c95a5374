// Router ("route") GEMM: bf16 activations x fp32 weights carried as two bf16 planes - gfx950.
//
//   y[m, n] = sum_k x[m,k] * (w_high[n,k] + scale * w_low[n,k])        (fp32 accumulate)
// with w_high = bf16(w), w_low = bf16((w - w_high) / scale), scale = 1/256: the fp32 router weight at
// ~16 mantissa bits from two bf16 MFMA GEMMs.  Replaces reference src/gemm/sm90/gemm_bf16xfp32.cu
// (kernel :88-410, launcher :413-557; config table src/gemm/sm90/entry.cc:23-84).
//
// MI355X design: weights sit on the MFMA M axis, tokens on N.  Two kernels: the decode-step shape
// (m <= 256 tokens against n = #experts rows of k = 4096: 3 - 33 MB of weights, a streaming / latency
// problem) runs on skinny 16-row tiles with K split over waves and workgroups (below), both weight planes and
// the activations fetched straight into MFMA operand layout (v_mfma_f32_16x16x32_bf16: lane (r, g) holds 8
// consecutive k of row r), 64 k per step, register double-buffered; larger m runs on the LDS-staged tile kernel
// (64 weight rows x 128 tokens, every operand read once per workgroup as full lines; round 5 - the 64 x 64
// direct-load kernel of rounds 1-4 stays behind development key 40 as the A/B partner).  Split-K partials
// go to an fp32 workspace; the last workgroup to arrive at a tile (device-scope counter) sums the
// splits in fixed order - deterministic - writes y and leaves the counter at zero for the next call
// (same contract as the reference's split_flag).
#include <type_traits>

#include "hpc_common.h"
#include "hpc_dev.h"
#include "../../include/hpc_amd.h"

namespace hpc {
namespace rgemm {

struct Args {
  const uint16_t* x;
  const uint16_t* wh;
  const uint16_t* wl;
  void* y;
  float* split_y;  // [S, m, n] fp32 (S > 1)
  int* flag;       // per-tile arrival counters, row stride flag_ld (S > 1)
  int m, n, k, splits, flag_ld, fp32_out;
  int dev_skip;  // development key 41: bit 0 = no weight loads, bit 1 = no activation loads in the tile kernel (timing only)
  float scale;
};

constexpr int kThreads = 256;

__device__ __forceinline__ f32x4 mfma_bf16(const u32x4 a, const u32x4 b, const f32x4 c) {
  return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b),
                                                 c, 0, 0, 0);
}

template <int kMT>
__global__ __launch_bounds__(kThreads) void gemm_bf16xfp32_kernel(const Args a) {
  constexpr int kTM = 16 * kMT;
  __shared__ int s_last;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int r16 = lane & 15, g4 = lane >> 4;
  const int n0 = blockIdx.x * 64, m0 = blockIdx.y * kTM, split = blockIdx.z;
  const int K = a.k;
  const int chunks = K >> 6;  // 64 k per step
  const int c_begin = static_cast<int>(static_cast<long>(chunks) * split / a.splits);
  const int c_end = static_cast<int>(static_cast<long>(chunks) * (split + 1) / a.splits);

  const unsigned w_bytes = static_cast<unsigned>(a.n) * static_cast<unsigned>(K) * 2u;
  const unsigned x_bytes = static_cast<unsigned>(a.m) * static_cast<unsigned>(K) * 2u;
  const unsigned w_off = static_cast<unsigned>(n0 + wave * 16 + r16) * static_cast<unsigned>(K) * 2u + g4 * 16;
  unsigned x_off[kMT];
#pragma unroll
  for (int j = 0; j < kMT; ++j) {
    const int t = m0 + j * 16 + r16;
    x_off[j] = static_cast<unsigned>(t < a.m ? t : a.m - 1) * static_cast<unsigned>(K) * 2u + g4 * 16;
  }

  f32x4 acc_h[kMT], acc_l[kMT];
#pragma unroll
  for (int j = 0; j < kMT; ++j) acc_h[j] = acc_l[j] = f32x4{0.f, 0.f, 0.f, 0.f};

  // Two 64-k steps in flight.  Loads are UNCONDITIONAL (steps past the end use empty descriptors and
  // read zeros): a branch around a prefetch makes hipcc fall back to vmcnt(0) at the join.
  u32x4 bh[2][2], bl[2][2], bx[2][kMT][2];  // [buffer][..][32-k half of the step]
  auto issue = [&](int buf, int c) {
    const int koff = c * 128;  // bytes
    const bool on = c < c_end;
    const auto rh_ = make_rsrc(a.wh, on ? w_bytes : 0u), rl_ = make_rsrc(a.wl, on ? w_bytes : 0u);
    const auto rx_ = make_rsrc(a.x, on ? x_bytes : 0u);
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      bh[buf][h] = buf_ld16<0>(rh_, w_off, koff + h * 64);
      bl[buf][h] = buf_ld16<0>(rl_, w_off, koff + h * 64);
#pragma unroll
      for (int j = 0; j < kMT; ++j) bx[buf][j][h] = buf_ld16<0>(rx_, x_off[j], koff + h * 64);
    }
  };
  auto compute = [&](int buf) {
#pragma unroll
    for (int h = 0; h < 2; ++h)
#pragma unroll
      for (int j = 0; j < kMT; ++j) {
        acc_h[j] = mfma_bf16(bh[buf][h], bx[buf][j][h], acc_h[j]);
        acc_l[j] = mfma_bf16(bl[buf][h], bx[buf][j][h], acc_l[j]);
      }
  };
  issue(0, c_begin);
  issue(1, c_begin + 1);
  for (int c = c_begin; c < c_end; c += 2) {
    compute(0);
    issue(0, c + 2);
    compute(1);
    issue(1, c + 3);
  }

  // lane holds rows n = n0 + wave*16 + g4*4 + i of token column m0 + j*16 + r16
  const int nn = n0 + wave * 16 + g4 * 4;
  auto emit = [&](int t, const f32x4 v) {
    if (a.fp32_out) {
      *reinterpret_cast<f32x4*>(static_cast<float*>(a.y) + static_cast<long>(t) * a.n + nn) = v;
    } else {
      u32x2 pk;
      pk[0] = pack_bf16x2(v[0], v[1]);
      pk[1] = pack_bf16x2(v[2], v[3]);
      *reinterpret_cast<u32x2*>(static_cast<uint16_t*>(a.y) + static_cast<long>(t) * a.n + nn) = pk;
    }
  };
  if (a.splits == 1) {
#pragma unroll
    for (int j = 0; j < kMT; ++j) {
      const int t = m0 + j * 16 + r16;
      if (t < a.m) emit(t, acc_l[j] * a.scale + acc_h[j]);
    }
    return;
  }

  // ---- split-K: park the partial, count arrivals, the last workgroup reduces ---------------------------
  // Partials travel with system-scope (sc0 sc1) buffer stores / loads: written through to memory and
  // read around the per-XCD L2s, so no L2 write-back / invalidate fence is needed (those cost tens
  // of microseconds here - more than the GEMM).
  const unsigned plane = static_cast<unsigned>(a.m) * static_cast<unsigned>(a.n) * 4u;  // bytes, < 2^28 (launcher)
  const auto rp = make_rsrc(a.split_y, plane * static_cast<unsigned>(a.splits));
#pragma unroll
  for (int j = 0; j < kMT; ++j) {
    const int t = m0 + j * 16 + r16;
    const f32x4 v = acc_l[j] * a.scale + acc_h[j];
    if (t < a.m)
      __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), rp,
                                             (static_cast<unsigned>(t) * a.n + nn) * 4u, split * plane, 17);
  }
  __builtin_amdgcn_s_waitcnt(0x0F70);  // vmcnt(0): the write-through stores are acknowledged
  __syncthreads();
  int* flag = a.flag + static_cast<long>(blockIdx.y) * a.flag_ld + blockIdx.x;
  if (tid == 0) {
    const int old = atomicAdd(flag, 1);
    s_last = (old == a.splits - 1);
    if (s_last) atomicExch(flag, 0);  // every split has arrived: nobody touches the counter again in this call
  }
  __syncthreads();
  if (!s_last) return;
  // all partial loads of (up to) two token blocks in flight at once: a serial load->add chain costs
  // one ~2 us memory round trip per split.  Splits past a.splits use an empty descriptor (reads 0).
  constexpr int kMaxSplits = 16;
  const auto rnull = make_rsrc(a.split_y, 0u);
#pragma unroll
  for (int j0 = 0; j0 < kMT; j0 += 2) {
    u32x4 part[2][kMaxSplits];
#pragma unroll
    for (int jj = 0; jj < 2; ++jj) {
      const int t = m0 + (j0 + jj) * 16 + r16;
      const unsigned off = (static_cast<unsigned>(t < a.m ? t : 0) * a.n + nn) * 4u;
#pragma unroll
      for (int sp = 0; sp < kMaxSplits; ++sp)
        if (j0 + jj < kMT) part[jj][sp] = buf_ld16<17>(sp < a.splits ? rp : rnull, off, sp * plane);
    }
#pragma unroll
    for (int jj = 0; jj < 2; ++jj) {
      if (j0 + jj >= kMT) continue;
      const int t = m0 + (j0 + jj) * 16 + r16;
      f32x4 sum = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int sp = 0; sp < kMaxSplits; ++sp) sum += __builtin_bit_cast(f32x4, part[jj][sp]);
      if (t < a.m) emit(t, sum);
    }
  }
}

// ---- m > 256: 64 weight rows x 128 tokens per workgroup, every operand staged through LDS ------------------
// The 64 x 64 kernel above fetches every activation row four times (each of its waves loads the whole token tile into
// MFMA operand registers) as 64-byte pieces: at m = 4096 it is bound by what the CU's load path accepts (93 us for
// 17 GFLOP, round 1-4).  Here every operand is read ONCE per workgroup as full 128-byte lines - a thread fetches one 16-byte
// chunk of four token rows and of two weight rows of both planes per 64-k step, three steps ahead, parks it in registers and
// writes it one step ahead into a double-buffered LDS stage ([128 tokens | 2 planes x 64 weight rows][128 B], 16-byte chunks
// XOR-swizzled with the row index: conflict-free b128 writes and operand reads) - and leaves LDS as MFMA operands: a wave owns
// 32 weight rows x 64 tokens (8 operand reads for 16 MFMAs per 32-k half).  One barrier per step, 64 KB of LDS, two
// workgroups per CU.  (First form of this kernel: weight planes straight into A-operand registers, 16 rows x 64 B per load
// instruction - those loads alone cost 50 of 128 us at m = 8192 x k = 7168.)  Split-K and its last-arriver reduction as above
// (counters on the same [m tiles, n / 64] grid, m tiles of 128).
__global__ __launch_bounds__(kThreads, 2) void gemm_bf16xfp32_tile_kernel(const Args a) {
  constexpr int kTT = 128;  // tokens per workgroup
  __shared__ u32x4 s_x[2][kTT * 8];
  __shared__ u32x4 s_w[2][2][64 * 8];  // [buffer][plane][64 weight rows x 8 chunks]
  __shared__ int s_last;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int r16 = lane & 15, g4 = lane >> 4;
  const int wr = wave & 1, wt = wave >> 1;  // weight-row half, token half
  const int n0 = blockIdx.x * 64, m0 = blockIdx.y * kTT, split = blockIdx.z;
  const int K = a.k;
  const int chunks = K >> 6;
  const int c_begin = static_cast<int>(static_cast<long>(chunks) * split / a.splits);
  const int c_end = static_cast<int>(static_cast<long>(chunks) * (split + 1) / a.splits);

  const unsigned w_bytes = static_cast<unsigned>(a.n) * static_cast<unsigned>(K) * 2u;
  const unsigned x_bytes = static_cast<unsigned>(a.m) * static_cast<unsigned>(K) * 2u;
  // staging roles: thread t carries 16-byte chunk t % 8 of tokens t / 8 + 32 q (q = 0 .. 3) and of weight rows t / 8 + 32 q2
  // (q2 = 0, 1) of both planes: every load instruction fetches whole 128-byte lines
  unsigned xg_off[4], wg_off[2];
  int xs_idx[4], ws_idx[2];
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const int tl = (tid >> 3) + 32 * q, t = m0 + tl;
    xg_off[q] = static_cast<unsigned>(t < a.m ? t : a.m - 1) * static_cast<unsigned>(K) * 2u + (tid & 7) * 16;
    xs_idx[q] = tl * 8 + ((tid & 7) ^ (tl & 7));
  }
#pragma unroll
  for (int q = 0; q < 2; ++q) {
    const int rl = (tid >> 3) + 32 * q;
    wg_off[q] = static_cast<unsigned>(n0 + rl) * static_cast<unsigned>(K) * 2u + (tid & 7) * 16;
    ws_idx[q] = rl * 8 + ((tid & 7) ^ (rl & 7));
  }
  // operand reads: token wt * 64 + j * 16 + r16 / weight row wr * 32 + i * 16 + r16, chunk h * 4 + g4 (h = 1: index ^ 4)
  int xr_idx[4], wr_idx[2];
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int tl = wt * 64 + j * 16 + r16;
    xr_idx[j] = tl * 8 + (g4 ^ (tl & 7));
  }
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int rl = wr * 32 + i * 16 + r16;
    wr_idx[i] = rl * 8 + (g4 ^ (rl & 7));
  }

  f32x4 acc_h[2][4], acc_l[2][4];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc_h[i][j] = acc_l[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  u32x4 xr[2][4], wreg[2][2][2];  // staging registers: two steps in flight ([stage][q] / [stage][plane][q2])
  // (Measured and dropped: every workgroup starting its k walk at its own step - rows of x and of the weight planes are a
  //  power of two apart, so workgroups in step queue at the same lines - changes nothing: 49.4 against 50.2 us.)
  auto load = [&](int rb, int c) {
    const bool on = c < c_end;
    const int koff = on ? c * 128 : 0;
    const bool mem_w = on && !(a.dev_skip & 1);  // development key 41 bit 0: no weight loads (timing only)
    const auto rh_ = make_rsrc(a.wh, mem_w ? w_bytes : 0u), rl_ = make_rsrc(a.wl, mem_w ? w_bytes : 0u);
    const auto rx_ = make_rsrc(a.x, on && !(a.dev_skip & 2) ? x_bytes : 0u);  // development key 41 bit 1: no activation loads
#pragma unroll
    for (int q = 0; q < 4; ++q) xr[rb][q] = buf_ld16<0>(rx_, xg_off[q], koff);
#pragma unroll
    for (int q = 0; q < 2; ++q) {
      wreg[rb][0][q] = buf_ld16<0>(rh_, wg_off[q], koff);
      wreg[rb][1][q] = buf_ld16<0>(rl_, wg_off[q], koff);
    }
  };
  auto stage = [&](int sb, int rb) {
#pragma unroll
    for (int q = 0; q < 4; ++q) s_x[sb][xs_idx[q]] = xr[rb][q];
#pragma unroll
    for (int q = 0; q < 2; ++q) {
      s_w[sb][0][ws_idx[q]] = wreg[rb][0][q];
      s_w[sb][1][ws_idx[q]] = wreg[rb][1][q];
    }
  };
  auto compute = [&](int sb) {
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      u32x4 bx[4], ah[2], al[2];
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        ah[i] = s_w[sb][0][wr_idx[i] ^ (h << 2)];
        al[i] = s_w[sb][1][wr_idx[i] ^ (h << 2)];
      }
#pragma unroll
      for (int j = 0; j < 4; ++j) bx[j] = s_x[sb][xr_idx[j] ^ (h << 2)];
#pragma unroll
      for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int i = 0; i < 2; ++i) {
          acc_h[i][j] = mfma_bf16(ah[i], bx[j], acc_h[i][j]);
          acc_l[i][j] = mfma_bf16(al[i], bx[j], acc_l[i][j]);
        }
    }
  };
  // Step s is loaded into staging registers [s & 1] three steps before it is multiplied and written to LDS buffer s & 1 one
  // step before (that buffer was last read two steps earlier, behind a barrier).
  load(0, c_begin);
  load(1, c_begin + 1);
  stage(0, 0);
  load(0, c_begin + 2);
  __syncthreads();
  for (int c = c_begin; c < c_end; c += 2) {
    stage(1, 1);  // step c + 1
    load(1, c + 3);
    compute(0);
    __syncthreads();
    stage(0, 0);  // step c + 2
    load(0, c + 4);
    compute(1);
    __syncthreads();
  }

  // lane holds weight rows nn_i + 0..3 of token column t_j
  auto emit = [&](int t, int nn, const f32x4 v) {
    if (a.fp32_out) {
      *reinterpret_cast<f32x4*>(static_cast<float*>(a.y) + static_cast<long>(t) * a.n + nn) = v;
    } else {
      u32x2 pk;
      pk[0] = pack_bf16x2(v[0], v[1]);
      pk[1] = pack_bf16x2(v[2], v[3]);
      *reinterpret_cast<u32x2*>(static_cast<uint16_t*>(a.y) + static_cast<long>(t) * a.n + nn) = pk;
    }
  };
  if (a.splits == 1) {
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int t = m0 + wt * 64 + j * 16 + r16;
        if (t < a.m) emit(t, n0 + wr * 32 + i * 16 + g4 * 4, acc_l[i][j] * a.scale + acc_h[i][j]);
      }
    return;
  }
  const unsigned plane = static_cast<unsigned>(a.m) * static_cast<unsigned>(a.n) * 4u;
  const auto rp = make_rsrc(a.split_y, plane * static_cast<unsigned>(a.splits));
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int t = m0 + wt * 64 + j * 16 + r16;
      const f32x4 v = acc_l[i][j] * a.scale + acc_h[i][j];
      if (t < a.m)
        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), rp,
                                               (static_cast<unsigned>(t) * a.n + n0 + wr * 32 + i * 16 + g4 * 4) * 4u, split * plane, 17);
    }
  __builtin_amdgcn_s_waitcnt(0x0F70);  // vmcnt(0): the write-through stores are acknowledged
  __syncthreads();
  int* flag = a.flag + static_cast<long>(blockIdx.y) * a.flag_ld + blockIdx.x;
  if (tid == 0) {
    const int old = atomicAdd(flag, 1);
    s_last = (old == a.splits - 1);
    if (s_last) atomicExch(flag, 0);
  }
  __syncthreads();
  if (!s_last) return;
  // The last arriver sums the splits in split order.  All partial loads of a trip are in flight at once (a serial load -> add
  // chain costs one ~2.5 us system-scope round trip per trip): 32 loads per trip = all eight (row block, token block) units of
  // the wave with up to 4 splits, four units with up to 8, two with up to 16 (round 5: two units per trip whatever the split
  // count - four trips = ~10 of the 49 us at m = 4096).  Splits past a.splits use an empty descriptor (read 0, add nothing).
  const auto rnull = make_rsrc(a.split_y, 0u);
  auto reduce = [&](auto ks) {
    constexpr int kS = decltype(ks)::value, kG = 32 / kS;
#pragma unroll
    for (int u0 = 0; u0 < 8; u0 += kG) {
      u32x4 part[kG][kS];
#pragma unroll
      for (int g = 0; g < kG; ++g) {
        const int i = (u0 + g) >> 2, j = (u0 + g) & 3;
        const int t = m0 + wt * 64 + j * 16 + r16;
        const unsigned off = (static_cast<unsigned>(t < a.m ? t : 0) * a.n + n0 + wr * 32 + i * 16 + g4 * 4) * 4u;
#pragma unroll
        for (int sp = 0; sp < kS; ++sp) part[g][sp] = buf_ld16<17>(sp < a.splits ? rp : rnull, off, sp * plane);
      }
#pragma unroll
      for (int g = 0; g < kG; ++g) {
        const int i = (u0 + g) >> 2, j = (u0 + g) & 3;
        const int t = m0 + wt * 64 + j * 16 + r16;
        f32x4 sum = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int sp = 0; sp < kS; ++sp) sum += __builtin_bit_cast(f32x4, part[g][sp]);
        if (t < a.m) emit(t, n0 + wr * 32 + i * 16 + g4 * 4, sum);
      }
    }
  };
  if (a.splits <= 4)
    reduce(std::integral_constant<int, 4>{});
  else if (a.splits <= 8)
    reduce(std::integral_constant<int, 8>{});
  else
    reduce(std::integral_constant<int, 16>{});
}

// ---- decode-size m (<= 256): skinny tiles, K split across waves AND workgroups --------------------------
// The problem is a stream of the two weight planes (4 - 33 MB) against a few tokens: it needs every
// CU pulling bytes (one CU sustains only ~50 GB/s), so the tile is the MFMA minimum of 16 weight rows
// x 16*kMT tokens, the 4 waves of a workgroup interleave the 64-k steps of the workgroup's K range
// (partials meet in LDS, free), and K is further split over blockIdx.z just enough to reach ~256
// workgroups.  Cross-workgroup partials are one f32x4 per thread (kept small on purpose: they travel
// write-through / uncached), the last arriver sums them in split order.
template <int kMT>
__global__ __launch_bounds__(kThreads) void gemm_bf16xfp32_skinny_kernel(const Args a) {
  constexpr int kTM = 16 * kMT;
  __shared__ f32x4 s_part[4][kMT][64];
  __shared__ int s_last;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int r16 = lane & 15, g4 = lane >> 4;
  const int n0 = blockIdx.x * 16, m0 = blockIdx.y * kTM, split = blockIdx.z;
  const int K = a.k;
  const int chunks = K >> 6;
  const int c_begin = static_cast<int>(static_cast<long>(chunks) * split / a.splits);
  const int c_end = static_cast<int>(static_cast<long>(chunks) * (split + 1) / a.splits);

  const unsigned w_bytes = static_cast<unsigned>(a.n) * static_cast<unsigned>(K) * 2u;
  const unsigned x_bytes = static_cast<unsigned>(a.m) * static_cast<unsigned>(K) * 2u;
  const unsigned w_off = static_cast<unsigned>(n0 + r16) * static_cast<unsigned>(K) * 2u + g4 * 16;
  unsigned x_off[kMT];
#pragma unroll
  for (int j = 0; j < kMT; ++j) {
    const int t = m0 + j * 16 + r16;
    x_off[j] = static_cast<unsigned>(t < a.m ? t : a.m - 1) * static_cast<unsigned>(K) * 2u + g4 * 16;
  }
  f32x4 acc_h[kMT], acc_l[kMT];
#pragma unroll
  for (int j = 0; j < kMT; ++j) acc_h[j] = acc_l[j] = f32x4{0.f, 0.f, 0.f, 0.f};

  // two steps in flight per wave; unconditional loads, empty descriptors past the end
  u32x4 bh[2][2], bl[2][2], bx[2][kMT][2];
  auto issue = [&](int buf, int c) {
    const int koff = c * 128;
    const bool on = c < c_end;
    const auto rh_ = make_rsrc(a.wh, on ? w_bytes : 0u), rl_ = make_rsrc(a.wl, on ? w_bytes : 0u);
    const auto rx_ = make_rsrc(a.x, on ? x_bytes : 0u);
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      bh[buf][h] = buf_ld16<0>(rh_, w_off, koff + h * 64);
      bl[buf][h] = buf_ld16<0>(rl_, w_off, koff + h * 64);
#pragma unroll
      for (int j = 0; j < kMT; ++j) bx[buf][j][h] = buf_ld16<0>(rx_, x_off[j], koff + h * 64);
    }
  };
  auto compute = [&](int buf) {
#pragma unroll
    for (int h = 0; h < 2; ++h)
#pragma unroll
      for (int j = 0; j < kMT; ++j) {
        acc_h[j] = mfma_bf16(bh[buf][h], bx[buf][j][h], acc_h[j]);
        acc_l[j] = mfma_bf16(bl[buf][h], bx[buf][j][h], acc_l[j]);
      }
  };
  issue(0, c_begin + wave);
  issue(1, c_begin + wave + 4);
  for (int c = c_begin + wave; c < c_end; c += 8) {
    compute(0);
    issue(0, c + 8);
    compute(1);
    issue(1, c + 12);
  }
#pragma unroll
  for (int j = 0; j < kMT; ++j) s_part[wave][j][lane] = acc_l[j] * a.scale + acc_h[j];
  __syncthreads();

  // threads 0 .. 64*kMT-1 own one f32x4 of the tile each: (token block j = tid / 64, MFMA C lane)
  const bool owner = tid < kMT * 64;
  const int j = tid >> 6;  // wave-uniform
  const int t = m0 + j * 16 + r16;
  const int nn = n0 + g4 * 4;
  f32x4 sum = f32x4{0.f, 0.f, 0.f, 0.f};
  if (owner) sum = (s_part[0][j][lane] + s_part[1][j][lane]) + (s_part[2][j][lane] + s_part[3][j][lane]);
  auto emit = [&](const f32x4 v) {
    if (a.fp32_out) {
      *reinterpret_cast<f32x4*>(static_cast<float*>(a.y) + static_cast<long>(t) * a.n + nn) = v;
    } else {
      u32x2 pk;
      pk[0] = pack_bf16x2(v[0], v[1]);
      pk[1] = pack_bf16x2(v[2], v[3]);
      *reinterpret_cast<u32x2*>(static_cast<uint16_t*>(a.y) + static_cast<long>(t) * a.n + nn) = pk;
    }
  };
  if (a.splits == 1) {
    if (owner && t < a.m) emit(sum);
    return;
  }
  const unsigned plane = static_cast<unsigned>(a.m) * static_cast<unsigned>(a.n) * 4u;
  const auto rp = make_rsrc(a.split_y, plane * static_cast<unsigned>(a.splits));
  const unsigned off = (static_cast<unsigned>(t < a.m ? t : 0) * a.n + nn) * 4u;
  if (owner && t < a.m) __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, sum), rp, off, split * plane, 17);
  __builtin_amdgcn_s_waitcnt(0x0F70);  // write-through stores acknowledged
  __syncthreads();
  int* flag = a.flag + static_cast<long>(blockIdx.y) * a.flag_ld + blockIdx.x;
  if (tid == 0) {
    const int old = atomicAdd(flag, 1);
    s_last = (old == a.splits - 1);
    if (s_last) atomicExch(flag, 0);
  }
  __syncthreads();
  if (!s_last || !owner) return;
  constexpr int kMaxSplits = 16;
  const auto rnull = make_rsrc(a.split_y, 0u);
  u32x4 part[kMaxSplits];
#pragma unroll
  for (int sp = 0; sp < kMaxSplits; ++sp) part[sp] = buf_ld16<17>(sp < a.splits ? rp : rnull, off, sp * plane);
  f32x4 tot = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int sp = 0; sp < kMaxSplits; ++sp) tot += __builtin_bit_cast(f32x4, part[sp]);
  if (t < a.m) emit(tot);
}

constexpr int kSkinnyMaxM = 256;
inline int skinny_tm(int m) { return m <= 16 ? 16 : (m <= 32 ? 32 : 64); }

}  // namespace rgemm
}  // namespace hpc

// Split count the launcher will use for (m, n, k): callers size split_y = splits * m * n floats and
// (m <= 256) provide ceil(m / tm) * n / 16 zeroed counters, (m > 256) a [ceil(m/64)+, n/64+] counter grid (the tile kernel
// counts on its first ceil(m / 128) rows).
extern "C" int hpc_gemm_bf16xfp32_splits(int m, int n, int k, int use_splitk) {
  using namespace hpc::rgemm;
  if (m <= 0 || n <= 0 || k <= 0) return 1;
  if (!use_splitk) return 1;
  int s = 1;
  int dev = 0;
  const int cus = hipGetDevice(&dev) == hipSuccess && hpc_get_cu_count(dev) > 0 ? hpc_get_cu_count(dev) : 256;
  if (m <= kSkinnyMaxM) {
    const int tm = skinny_tm(m);
    const long tiles = static_cast<long>((m + tm - 1) / tm) * (n / 16);
    // ~one workgroup per CU; every wave keeps at least one 64-k step
    while (s < 16 && tiles * s < cus && (k >> 6) / (s * 2) >= 4) s *= 2;
    return s;
  }
  // m > 256: the tile kernel, 64 weight rows x 128 tokens.  Splits until there is ONE workgroup per CU, at most 8 (every split
  // costs the hand-off of its fp32 partials: m = 4096 x n = 256 35.7 us with 4 splits = two workgroups per CU, 28.7 us with 2;
  // m = 1024 29.2 us with 16 splits, 19.9 with 8 - profiles/round5_router_tile_ab.txt); a launch that has between one and two
  // workgroups per CU without splitting is split once more (two resident workgroups per CU overlap each other's load phases).
  const long tiles = static_cast<long>((m + 127) / 128) * (n / 64);
  // development key 45: cap on the split count above m = 256 (never above the 16 planes the reduce sums)
  const int cap = hpc_dev_tuning_get(45) > 0 ? (hpc_dev_tuning_get(45) < 16 ? hpc_dev_tuning_get(45) : 16) : 8;
  while (s < cap && tiles * s < cus && k / (s * 2) >= 256) s *= 2;
  if (tiles >= cus && tiles < 2 * cus && s == 1 && k >= 512) s = 2;
  return s;
}

// reference: gemm_bf16xfp32_async (src/gemm/gemm.h:13-17, src/gemm/sm90/gemm_bf16xfp32.cu:488-557)
extern "C" int hpc_gemm_bf16xfp32_async(void* y_ptr, void* splitk_y_ptr, void* split_flag_ptr,
                                        const void* x_ptr, const void* w_high_ptr, const void* w_low_ptr,
                                        int m, int n, int k, float scale, int use_fp32_output, int splits,
                                        int flag_ld, hipStream_t stream) {
  using namespace hpc::rgemm;
  if (!y_ptr || !x_ptr || !w_high_ptr || !w_low_ptr) return HPC_ERR_INVALID;
  if (m < 0 || n <= 0 || k <= 0 || splits < 1) return HPC_ERR_INVALID;
  if (m == 0) return HPC_OK;
  if ((n & 63) || (k & 63)) return HPC_ERR_UNSUPPORTED;
  if (static_cast<int64_t>(m) * k * 2 > 0xfffffff0ll || static_cast<int64_t>(n) * k * 2 > 0xfffffff0ll)
    return HPC_ERR_UNSUPPORTED;  // 32-bit buffer offsets
  if (splits > 1 && (!splitk_y_ptr || !split_flag_ptr)) return HPC_ERR_INVALID;
  if (splits > (k >> 6)) return HPC_ERR_INVALID;
  if (splits > 16) return HPC_ERR_INVALID;  // the last arriver of every kernel form sums at most 16 partial planes
  if (splits > 1 && static_cast<int64_t>(splits) * m * n * 4 > 0xfffffff0ll) return HPC_ERR_UNSUPPORTED;
  Args a;
  a.x = static_cast<const uint16_t*>(x_ptr);
  a.wh = static_cast<const uint16_t*>(w_high_ptr);
  a.wl = static_cast<const uint16_t*>(w_low_ptr);
  a.y = y_ptr;
  a.split_y = static_cast<float*>(splitk_y_ptr);
  a.flag = static_cast<int*>(split_flag_ptr);
  a.m = m;
  a.n = n;
  a.k = k;
  a.splits = splits;
  a.flag_ld = flag_ld;
  a.fp32_out = use_fp32_output;
  a.scale = scale;
  a.dev_skip = hpc_dev_tuning_get(41);
  if (m <= kSkinnyMaxM) {
    const int tm = skinny_tm(m);
    dim3 grid(n / 16, (m + tm - 1) / tm, splits);
    if (splits > 1 && flag_ld < n / 16) return HPC_ERR_INVALID;
    if (tm == 16)
      gemm_bf16xfp32_skinny_kernel<1><<<grid, kThreads, 0, stream>>>(a);
    else if (tm == 32)
      gemm_bf16xfp32_skinny_kernel<2><<<grid, kThreads, 0, stream>>>(a);
    else
      gemm_bf16xfp32_skinny_kernel<4><<<grid, kThreads, 0, stream>>>(a);
    HPC_CHECK_LAUNCH();
    return HPC_OK;
  }
  if (splits > 1 && flag_ld < n / 64) return HPC_ERR_INVALID;
  // Measured (profiles/round5_router_tile_ab.txt; n = 256, k = 4096 unless said, old -> new): m = 4096 94.7 -> 29.0 us, m = 1024
  // 32.5 -> 20.1, m = 304 23.6 -> 18.2, m = 8192 x k = 7168 304 -> 77.5 us (0.78 PFLOP/s), m = 16384 x n = 128 180 -> 51.5 us.  What is
  // left (timing-only variants, development key 41): the loop with no loads at all takes 62 of the 77.5 us (eight 16-byte LDS
  // stores + sixteen operand reads per wave for 32 MFMAs, one barrier per step), and below m ~ 1024 the call is its fixed
  // cost (launch, eight steps, split hand-off: 18-19 us).
  if (hpc_dev_tuning_get(40) == 1) {  // development key 40 = 1: the 64 x 64 kernel of rounds 1-4 (operands straight from memory)
    dim3 grid(n / 64, (m + 63) / 64, splits);
    if (grid.y > 65535) return HPC_ERR_UNSUPPORTED;
    gemm_bf16xfp32_kernel<4><<<grid, kThreads, 0, stream>>>(a);
    HPC_CHECK_LAUNCH();
    return HPC_OK;
  }
  dim3 grid(n / 64, (m + 127) / 128, splits);
  if (grid.y > 65535) return HPC_ERR_UNSUPPORTED;
  gemm_bf16xfp32_tile_kernel<<<grid, kThreads, 0, stream>>>(a);
  HPC_CHECK_LAUNCH();
  return HPC_OK;
}
