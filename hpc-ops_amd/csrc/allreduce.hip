// Fused AllReduce + residual add + RMSNorm over xGMI peer memory (bf16) - gfx950.
//
// Replaces reference src/allreduce/fuse_allreduce_rmsnorm_high_throughput.cu:15-154 (one-kernel
// two-shot over NVLS multimem) and fuse_allreduce_rmsnorm_low_latency.cu:16-453 (Lamport two-kernel).
//
// MI355X design: xGMI is a point-to-point full mesh (7 links x ~153 GB/s per GPU) with no
// multicast and no in-fabric reduction, so both modes are explicit two-shot exchanges over the
// peers' symmetric (uncached) buffers, every phase using all links at once, one peer per link:
//  * high throughput: rank r owns a contiguous token slice.  reduce-scatter = r READS its slice
//    from every peer (system-scope loads), adds the residual, normalises, and all-gather = r
//    WRITES the normalised rows into every peer's output buffer.  Two signal-pad barriers
//    (per-block flags in the peers' pads, CAS 0->1 post / 1->0 consume like the reference,
//    high_throughput.cu:37-43) bracket the kernel.
//  * low latency: Lamport protocol - the data is its own flag.  Buffers are pre-filled with the
//    -0.0 word pattern (0x80000000); a 16-byte vector is "arrived" when none of its words is the
//    sentinel (real -0.0 words are rewritten to +0.0 by the sender).  Token t is owned by rank
//    t % ws: kernel 1 cleans the slot that will be used next and pushes the local rows to their
//    owners; kernel 2 has one workgroup per row that either reduces + broadcasts (owner) or polls
//    the broadcast copy, then adds the residual and normalises.  Three slots rotate; the slot
//    state lives in the caller's `buffer_flags` exactly like the reference (low_latency.h:208-304).
//    No workgroup ever waits on another workgroup of the same GPU, so the protocol cannot
//    deadlock on residency; every spin is bounded.
#include "hpc_common.h"
#include "../../include/hpc_amd.h"

namespace hpc {
namespace ar {

constexpr int kThreads = 256;
constexpr int kMaxWs = 8;
constexpr int kMaxVec = 8;  // 16-byte vectors per thread per row: hidden <= 8 * 256 * 8 = 16384
constexpr uint32_t kSentinel = 0x80000000u;
constexpr long kSpinLimit = 1L << 22;  // ~seconds; a dead peer yields a counted timeout, not a hang

__device__ int g_timeouts = 0;

// system-scope (sc0 sc1) accesses: the load bypasses L1/L2 so a peer's store is observed without a
// kernel boundary; the store is written through to its home memory at once (a write-back store
// would sit in this GPU's L2 until the kernel ends while the peer spins on it).
__device__ __forceinline__ u32x4 ld16_sys(const void* p) {
  u32x4 v;
  asm volatile("global_load_dwordx4 %0, %1, off sc0 sc1\n\ts_waitcnt vmcnt(0)" : "=&v"(v) : "v"(p) : "memory");
  return v;
}
__device__ __forceinline__ void st16_sys(void* p, u32x4 v) {
  asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1\n\ts_nop 1" : : "v"(p), "v"(v) : "memory");
}
// bulk read of peer data that was complete before this kernel's barrier (first touch -> L1 miss)
__device__ __forceinline__ u32x4 ld16_peer(const void* p) {
  return __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(p));
}
__device__ __forceinline__ u32x4 sanitize(u32x4 v) {
#pragma unroll
  for (int i = 0; i < 4; ++i) v[i] = v[i] == kSentinel ? 0u : v[i];
  return v;
}
__device__ __forceinline__ bool arrived(u32x4 v) {
  return v[0] != kSentinel && v[1] != kSentinel && v[2] != kSentinel && v[3] != kSentinel;
}
__device__ __forceinline__ u32x4 poll16(const void* p) {
  u32x4 v = ld16_sys(p);
  long spins = 0;
  while (!arrived(v)) {
    __builtin_amdgcn_s_sleep(2);
    v = ld16_sys(p);
    if (++spins > kSpinLimit) {
      atomicAdd(&g_timeouts, 1);
      break;
    }
  }
  return v;
}

__device__ __forceinline__ float block_sum(float v, float* red) {
  v = wave_sum(v);
  __syncthreads();
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
  __syncthreads();
  return red[0] + red[1] + red[2] + red[3];
}

// r = bf16(sum + residual) -> residual_out; y = bf16(float(r) * rsqrt(mean(r^2) + eps) * w)
// acc[i][0..7] hold the all-reduced row (fp32, already rounded to bf16 precision by the caller if
// the mode requires it); returns the normalised row packed in `out`.
__device__ __forceinline__ void residual_rmsnorm(float (&acc)[kMaxVec][8], int nvec, int hidden,
                                                 const uint16_t* res_row, uint16_t* res_out_row,
                                                 const uint16_t* w, float eps, u32x4 (&out)[kMaxVec],
                                                 float* red) {
  float ss = 0.f;
#pragma unroll
  for (int i = 0; i < kMaxVec; ++i) {
    const int v = threadIdx.x + i * kThreads;
    if (v < nvec) {
      const u32x4 rv = ld16(res_row + v * 8);
      u32x4 ro;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const float a = acc[i][2 * j] + bf16lo_to_f32(rv[j]);
        const float b = acc[i][2 * j + 1] + bf16hi_to_f32(rv[j]);
        ro[j] = pack_bf16x2(a, b);
        acc[i][2 * j] = bf16lo_to_f32(ro[j]);
        acc[i][2 * j + 1] = bf16hi_to_f32(ro[j]);
        ss = fmaf(acc[i][2 * j], acc[i][2 * j], ss);
        ss = fmaf(acc[i][2 * j + 1], acc[i][2 * j + 1], ss);
      }
      st16(res_out_row + v * 8, ro);
    }
  }
  ss = block_sum(ss, red);
  const float rms = rsqrtf(ss / static_cast<float>(hidden) + eps);
#pragma unroll
  for (int i = 0; i < kMaxVec; ++i) {
    const int v = threadIdx.x + i * kThreads;
    if (v < nvec) {
      const u32x4 wv = ld16(w + v * 8);
#pragma unroll
      for (int j = 0; j < 4; ++j)
        out[i][j] = pack_bf16x2(acc[i][2 * j] * rms * bf16lo_to_f32(wv[j]),
                                acc[i][2 * j + 1] * rms * bf16hi_to_f32(wv[j]));
    }
  }
}

// ---------------------------------------------------------------------------------------------
// High throughput
// ---------------------------------------------------------------------------------------------
struct HtArgs {
  const uint16_t* in[kMaxWs];   // peer p's copy of MY token slice (rows x hidden)
  uint16_t* out[kMaxWs];        // peer p's output rows of MY token slice
  uint32_t* sig[kMaxWs];        // peer p's signal pad
  const uint16_t* residual;
  uint16_t* out_residual;
  const uint16_t* w;
  float eps;
  int rows, hidden, rank, ws;
};

__device__ __forceinline__ void signal_barrier(const HtArgs& a) {
  // block b of rank r <-> block b of every peer (reference high_throughput.cu:37-43, utils.cuh:571-590)
  __syncthreads();
  const int t = threadIdx.x;
  if (t < a.ws) {
    uint32_t* post = a.sig[t] + blockIdx.x * a.ws + a.rank;  // my flag in peer t's pad
    long spins = 0;
    uint32_t expect = 0u;
    while (!__hip_atomic_compare_exchange_strong(post, &expect, 1u, __ATOMIC_RELEASE, __ATOMIC_RELAXED,
                                                 __HIP_MEMORY_SCOPE_SYSTEM)) {
      expect = 0u;
      __builtin_amdgcn_s_sleep(1);
      if (++spins > kSpinLimit) {
        atomicAdd(&g_timeouts, 1);
        break;
      }
    }
    uint32_t* wait = a.sig[a.rank] + blockIdx.x * a.ws + t;  // peer t's flag in my pad
    spins = 0;
    expect = 1u;
    while (!__hip_atomic_compare_exchange_strong(wait, &expect, 0u, __ATOMIC_ACQUIRE, __ATOMIC_RELAXED,
                                                 __HIP_MEMORY_SCOPE_SYSTEM)) {
      expect = 1u;
      __builtin_amdgcn_s_sleep(1);
      if (++spins > kSpinLimit) {
        atomicAdd(&g_timeouts, 1);
        break;
      }
    }
  }
  __syncthreads();
}

__global__ __launch_bounds__(kThreads) void ht_kernel(const HtArgs a) {
  __shared__ float red[4];
  const int nvec = a.hidden >> 3;
  signal_barrier(a);  // every rank's input is in place
  for (int row = blockIdx.x; row < a.rows; row += gridDim.x) {
    const long roff = static_cast<long>(row) * a.hidden;
    float acc[kMaxVec][8];
#pragma unroll
    for (int i = 0; i < kMaxVec; ++i) {
      const int v = threadIdx.x + i * kThreads;
#pragma unroll
      for (int j = 0; j < 8; ++j) acc[i][j] = 0.f;
      if (v < nvec) {
        for (int p = 0; p < a.ws; ++p) {
          const u32x4 x = ld16_peer(a.in[p] + roff + v * 8);
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            acc[i][2 * j] += bf16lo_to_f32(x[j]);
            acc[i][2 * j + 1] += bf16hi_to_f32(x[j]);
          }
        }
      }
    }
    u32x4 y[kMaxVec];
    residual_rmsnorm(acc, nvec, a.hidden, a.residual + roff, a.out_residual + roff, a.w, a.eps, y, red);
#pragma unroll
    for (int i = 0; i < kMaxVec; ++i) {
      const int v = threadIdx.x + i * kThreads;
      if (v < nvec)
        for (int p = 0; p < a.ws; ++p) st16_sys(a.out[p] + roff + v * 8, y[i]);
    }
  }
  __threadfence_system();  // my rows are visible in every peer before I signal
  signal_barrier(a);       // every rank's rows have landed here
}

// ---------------------------------------------------------------------------------------------
// Low latency (Lamport)
// ---------------------------------------------------------------------------------------------
struct LlArgs {
  const uint16_t* x;        // [rows, hidden] local input
  const long* peers;        // device table [ws] of the ranks' workspace base addresses
  uint8_t* local_ws;        // this rank's workspace
  uint32_t* flags;          // [9] cur, dirty, bytes/slot, dirty stages, 4 x bytes to clear, arrive
  const uint16_t* residual;
  uint16_t* residual_out;
  uint16_t* y;
  const uint16_t* w;
  float eps;
  int rows, hidden, rank, ws;
};

__global__ __launch_bounds__(kThreads) void ll_scatter_kernel(const LlArgs a) {
  const uint32_t cur = a.flags[0] % 3u;
  const uint32_t slot_bytes = a.flags[2];
  const uint32_t nxt = (cur + 1u) % 3u;
  // 1. clean the slot the NEXT call will use: nobody reads or writes it any more (everyone has
  //    finished the call before the previous one, or this call could not have started)
  {
    const uint32_t dirty = a.flags[4 + nxt];
    uint8_t* base = a.local_ws + static_cast<long>(nxt) * slot_bytes;
    const u32x4 s = u32x4{kSentinel, kSentinel, kSentinel, kSentinel};
    for (long o = (static_cast<long>(blockIdx.x) * kThreads + threadIdx.x) * 16; o < dirty;
         o += static_cast<long>(gridDim.x) * kThreads * 16)
      st16_sys(base + o, s);
  }
  // 2. push my rows to their owners: owner's slot [t / ws][rank][hidden]
  const int nvec = a.hidden >> 3;
  for (int t = blockIdx.x; t < a.rows; t += gridDim.x) {
    const int owner = t % a.ws;
    uint8_t* dst = reinterpret_cast<uint8_t*>(a.peers[owner]) + static_cast<long>(cur) * slot_bytes +
                   (static_cast<long>(t / a.ws) * a.ws + a.rank) * a.hidden * 2;
    const uint16_t* src = a.x + static_cast<long>(t) * a.hidden;
    for (int v = threadIdx.x; v < nvec; v += kThreads) st16_sys(dst + v * 16, sanitize(ld16(src + v * 8)));
  }
}

__global__ __launch_bounds__(kThreads) void ll_reduce_norm_kernel(const LlArgs a) {
  __shared__ float red[4];
  const uint32_t cur = a.flags[0] % 3u;
  const uint32_t slot_bytes = a.flags[2];
  const int nvec = a.hidden >> 3;
  const int n_pad = (a.rows + a.ws - 1) / a.ws * a.ws;
  const long row_bytes = static_cast<long>(a.hidden) * 2;
  const long bcast_off = static_cast<long>(cur) * slot_bytes + static_cast<long>(n_pad) * row_bytes;

  // slot bookkeeping: the last workgroup to arrive (all have read `cur` by then) rotates the slots
  __syncthreads();
  if (threadIdx.x == 0) {
    const uint32_t old = atomicAdd(&a.flags[8], 1u);
    if (old == gridDim.x - 1) {
      a.flags[4 + cur] = static_cast<uint32_t>(2 * n_pad * row_bytes);
      a.flags[8] = 0u;
      a.flags[1] = (cur + 2u) % 3u;
      __threadfence();
      a.flags[0] = (cur + 1u) % 3u;
    }
  }

  for (int t = blockIdx.x; t < a.rows; t += gridDim.x) {
    float acc[kMaxVec][8];
    if (t % a.ws == a.rank) {
      // owner: wait for every rank's copy of row t, reduce, broadcast
      const uint8_t* src = a.local_ws + static_cast<long>(cur) * slot_bytes +
                           static_cast<long>(t / a.ws) * a.ws * row_bytes;
#pragma unroll
      for (int i = 0; i < kMaxVec; ++i) {
        const int v = threadIdx.x + i * kThreads;
#pragma unroll
        for (int j = 0; j < 8; ++j) acc[i][j] = 0.f;
        if (v < nvec) {
          for (int p = 0; p < a.ws; ++p) {
            const u32x4 xv = poll16(src + p * row_bytes + v * 16);
#pragma unroll
            for (int j = 0; j < 4; ++j) {
              acc[i][2 * j] += bf16lo_to_f32(xv[j]);
              acc[i][2 * j + 1] += bf16hi_to_f32(xv[j]);
            }
          }
          u32x4 sum;
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            sum[j] = pack_bf16x2(acc[i][2 * j], acc[i][2 * j + 1]);
            acc[i][2 * j] = bf16lo_to_f32(sum[j]);
            acc[i][2 * j + 1] = bf16hi_to_f32(sum[j]);
          }
          sum = sanitize(sum);
          for (int p = 0; p < a.ws; ++p)
            if (p != a.rank)
              st16_sys(reinterpret_cast<uint8_t*>(a.peers[p]) + bcast_off + t * row_bytes + v * 16, sum);
        }
      }
    } else {
      const uint8_t* src = a.local_ws + bcast_off + t * row_bytes;
#pragma unroll
      for (int i = 0; i < kMaxVec; ++i) {
        const int v = threadIdx.x + i * kThreads;
#pragma unroll
        for (int j = 0; j < 8; ++j) acc[i][j] = 0.f;
        if (v < nvec) {
          const u32x4 xv = poll16(src + v * 16);
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            acc[i][2 * j] = bf16lo_to_f32(xv[j]);
            acc[i][2 * j + 1] = bf16hi_to_f32(xv[j]);
          }
        }
      }
    }
    u32x4 y[kMaxVec];
    const long roff = static_cast<long>(t) * a.hidden;
    residual_rmsnorm(acc, nvec, a.hidden, a.residual + roff, a.residual_out + roff, a.w, a.eps, y, red);
#pragma unroll
    for (int i = 0; i < kMaxVec; ++i) {
      const int v = threadIdx.x + i * kThreads;
      if (v < nvec) st16(a.y + roff + v * 8, y[i]);
    }
    __syncthreads();
  }
}

}  // namespace ar
}  // namespace hpc

using namespace hpc::ar;

extern "C" int hpc_allreduce_timeouts(void) {
  int v = 0;
  if (hipMemcpyFromSymbol(&v, HIP_SYMBOL(g_timeouts), sizeof(int)) != hipSuccess) return -1;
  return v;
}

extern "C" int hpc_fuse_allreduce_rmsnorm_high_throughput_async(
    const void* const* peer_x_ptrs, void* const* peer_out_ptrs, void* const* peer_signal_ptrs,
    const void* residual_ptr, void* out_residual_ptr, const void* weight_ptr, float rms_norm_eps,
    int num_rows, int hidden_size, int rank, int world_size, int num_max_blocks, hipStream_t stream) {
  if (!peer_x_ptrs || !peer_out_ptrs || !peer_signal_ptrs || !residual_ptr || !out_residual_ptr || !weight_ptr)
    return HPC_ERR_INVALID;
  if (world_size < 1 || world_size > kMaxWs || rank < 0 || rank >= world_size) return HPC_ERR_UNSUPPORTED;
  if ((hidden_size & 7) || hidden_size <= 0 || hidden_size > kMaxVec * kThreads * 8) return HPC_ERR_UNSUPPORTED;
  if (num_max_blocks <= 0) return HPC_ERR_INVALID;
  HtArgs a;
  for (int p = 0; p < kMaxWs; ++p) {
    a.in[p] = p < world_size ? static_cast<const uint16_t*>(peer_x_ptrs[p]) : nullptr;
    a.out[p] = p < world_size ? static_cast<uint16_t*>(peer_out_ptrs[p]) : nullptr;
    a.sig[p] = p < world_size ? static_cast<uint32_t*>(peer_signal_ptrs[p]) : nullptr;
  }
  a.residual = static_cast<const uint16_t*>(residual_ptr);
  a.out_residual = static_cast<uint16_t*>(out_residual_ptr);
  a.w = static_cast<const uint16_t*>(weight_ptr);
  a.eps = rms_norm_eps;
  a.rows = num_rows;
  a.hidden = hidden_size;
  a.rank = rank;
  a.ws = world_size;
  // Every rank must launch the same grid: the barriers pair block b with block b of each peer, so
  // the grid is a pure function of (rows, num_max_blocks).  One row is only 2*hidden bytes per
  // peer, so a 78-block grid (the reference's SM-count-sized default) is latency-bound on MI355X:
  // use at least 2 workgroups per CU worth of rows; the signal pad (72 * CUs words) bounds it.
  int grid = num_max_blocks > 512 ? num_max_blocks : 512;
  if (grid > num_rows) grid = num_rows > 0 ? num_rows : 1;
  if (grid * world_size > 72 * 256) grid = 72 * 256 / world_size;
  ht_kernel<<<grid, kThreads, 0, stream>>>(a);
  HPC_CHECK_LAUNCH();
  return HPC_OK;
}

extern "C" int hpc_fuse_allreduce_rmsnorm_low_latency_async(
    void* output_ptr, void* residual_out_ptr, const void* input_ptr, const void* data_buffer_ptrs_dev,
    void* local_workspace_ptr, void* buffer_flags_dev, const void* residual_in_ptr,
    const void* weight_ptr, float rms_norm_eps, int num_tokens, int hidden_size, int rank,
    int world_size, int64_t workspace_bytes, hipStream_t stream) {
  if (!output_ptr || !residual_out_ptr || !input_ptr || !data_buffer_ptrs_dev || !local_workspace_ptr ||
      !buffer_flags_dev || !residual_in_ptr || !weight_ptr)
    return HPC_ERR_INVALID;
  if (world_size < 1 || world_size > 64 || rank < 0 || rank >= world_size) return HPC_ERR_UNSUPPORTED;
  if ((hidden_size & 7) || hidden_size <= 0 || hidden_size > kMaxVec * kThreads * 8) return HPC_ERR_UNSUPPORTED;
  if (num_tokens <= 0) return HPC_OK;
  const int64_t n_pad = (num_tokens + world_size - 1) / world_size * world_size;
  const int64_t need = 3 * 2 * n_pad * hidden_size * 2;
  if (workspace_bytes < need) return HPC_ERR_INVALID;
  LlArgs a;
  a.x = static_cast<const uint16_t*>(input_ptr);
  a.peers = static_cast<const long*>(data_buffer_ptrs_dev);
  a.local_ws = static_cast<uint8_t*>(local_workspace_ptr);
  a.flags = static_cast<uint32_t*>(buffer_flags_dev);
  a.residual = static_cast<const uint16_t*>(residual_in_ptr);
  a.residual_out = static_cast<uint16_t*>(residual_out_ptr);
  a.y = static_cast<uint16_t*>(output_ptr);
  a.w = static_cast<const uint16_t*>(weight_ptr);
  a.eps = rms_norm_eps;
  a.rows = num_tokens;
  a.hidden = hidden_size;
  a.rank = rank;
  a.ws = world_size;
  const int grid = num_tokens < 2048 ? num_tokens : 2048;
  ll_scatter_kernel<<<grid, kThreads, 0, stream>>>(a);
  HPC_CHECK_LAUNCH();
  ll_reduce_norm_kernel<<<grid, kThreads, 0, stream>>>(a);
  HPC_CHECK_LAUNCH();
  return HPC_OK;
}
