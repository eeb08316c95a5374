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
//    deadlock on residency; every spin is bounded.  Round 5: when the grid is resident at once the two
//    phases run in ONE launch (ll_fused_kernel) - at decode sizes the second launch was a third of the call.
#include "hpc_common.h"
#include "hpc_dev.h"
#include "../../include/hpc_amd.h"

namespace hpc {
namespace ar {

constexpr int kThreads = 256;
constexpr int kMaxWs = 8;
constexpr int kMaxVec = 8;  // 16-byte vectors per thread per row: hidden <= 8 * 256 * 8 = 16384
constexpr uint32_t kSentinel = 0x80000000u;
constexpr long kSpinLimit = 1L << 22;  // ~seconds; a dead peer yields a counted timeout, not a hang

// Peer-visible timeout counter: a word of pinned host memory every GPU of the process can bump with
// a system-scope atomic and the host can read WITHOUT synchronising a stream.  The C entries refuse
// to launch once it is non-zero (HPC_ERR_TIMEOUT): after a spin gave up the Lamport slots / signal
// pads hold leftovers and every later result would be silently wrong.
__device__ __forceinline__ void note_timeout(int* counter) {
  __hip_atomic_fetch_add(counter, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}

// system-scope (sc0 sc1) accesses: the load bypasses L1/L2 so a peer's store is observed without a
// kernel boundary (and a line of a peer's buffer cached by an EARLIER call is never served again);
// the store is written through to its home memory at once (a write-back store would sit in this
// GPU's L2 until the kernel ends while the peer spins on it).
// Loads go through buffer descriptors with the cache-policy bits in `aux`, so hipcc tracks them in
// vmcnt like any other load: a thread issues ALL the loads of a phase (every peer, every vector)
// before it waits for the first one - with a hand-written `global_load ... ; s_waitcnt vmcnt(0)`
// pair there was exactly one remote request in flight per thread.
constexpr int kAuxSys = 17;  // sc0 | sc1
__device__ __forceinline__ void st16_sys(void* p, u32x4 v) {
  asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1\n\ts_nop 1" : : "v"(p), "v"(v) : "memory");
}
__device__ __forceinline__ u32x4 sanitize(u32x4 v) {
#pragma unroll
  for (int i = 0; i < 4; ++i) v[i] = v[i] == kSentinel ? 0u : v[i];
  return v;
}
__device__ __forceinline__ bool arrived(u32x4 v) {
  return v[0] != kSentinel && v[1] != kSentinel && v[2] != kSentinel && v[3] != kSentinel;
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
// `rpre` / `wpre` (round 6): the residual / weight vectors of the row when the caller has them in registers already - the
// high-throughput body issues the residual loads TOGETHER with the peers' rows (one memory round trip per row instead of two
// dependent ones) and loads the weight vectors once per workgroup; null = load here (the low-latency bodies: their rows
// arrive by polling, the residual load overlaps the poll rounds).
template <int kVec>
__device__ __forceinline__ void residual_rmsnorm(float (&acc)[kVec][8], int nvec, int hidden,
                                                 const uint16_t* res_row, uint16_t* res_out_row,
                                                 const uint16_t* w, float eps, u32x4 (&out)[kVec],
                                                 float* red, const u32x4* rpre = nullptr, const u32x4* wpre = nullptr) {
  float ss = 0.f;
#pragma unroll
  for (int i = 0; i < kVec; ++i) {
    const int v = threadIdx.x + i * kThreads;
    if (v < nvec) {
      const u32x4 rv = rpre ? rpre[i] : ld16(res_row + v * 8);
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
  for (int i = 0; i < kVec; ++i) {
    const int v = threadIdx.x + i * kThreads;
    if (v < nvec) {
      const u32x4 wv = wpre ? wpre[i] : ld16(w + v * 8);
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
  int* timeouts;                // host-pinned counter (see note_timeout)
  long spin_limit;
  float eps;
  int rows, hidden, rank, ws;
  int cas_barrier;  // development key 47 = 1: signal barriers with compare-and-swap (rounds 1-5)
  int sig_stride;   // words between the flag groups of consecutive blocks (hpc_..._signal_stride: spread over the whole pad)
};

__device__ __forceinline__ void signal_barrier(const HtArgs& a, int bid, bool closing = false) {
  // block b of rank r <-> block b of every peer (reference high_throughput.cu:37-43, utils.cuh:571-590: CAS 0 -> 1 to post,
  // CAS 1 -> 0 to consume, release / acquire at system scope).  Same protocol, same words, same values, but (round 6):
  //  * no read-modify-write: a word has exactly ONE writer per direction (only the poster ever turns it 0 -> 1, only its
  //    owner 1 -> 0), so "post" = wait until the word reads 0, then store 1, "consume" = wait until it reads 1, then store 0;
  //  * no system-scope release / acquire FENCES.  On gfx950 a system-scope release is `buffer_wbl2 sc0 sc1` - write back
  //    every dirty line of this XCD's L2 - and an acquire `buffer_inv sc0 sc1` - invalidate it: five to six L2-wide
  //    operations per workgroup and call, serialised per L2 (measured at world size 1, profiles/round6_allreduce_ab.txt:
  //    a call's fixed cost was 85 ns x the number of workgroups - 22 us at 256, 43 us at 512 - whatever the flag
  //    instruction and wherever the flags sit; without them T = 512 29.4 -> 9.9 us, T = 4096 79.4 -> 49.9 us per call).  None of it is needed: everything a PEER reads was stored with sc0 sc1
  //    (written through to its home memory, uncached there) and is read with sc0 sc1 loads (never served from a cache) -
  //    the idiom of the Lamport path and of the decode kernels' split-KV partials.  So: every thread retires its own stores
  //    (vmcnt(0): a write-through store is acknowledged by its home), the workgroup meets, and the flags are relaxed
  //    system-scope accesses.  The rows this rank keeps to itself (out_residual: ordinary write-back stores) are nobody
  //    else's business and reach memory at the end of the kernel like any other output.
  const int t = threadIdx.x;
  if (kHpcDevBuild && a.cas_barrier) {  // development key 47 = 1: the form of rounds 1-5 (CAS loops, release / acquire fences) - A/B
    if (closing) __threadfence_system();
    __syncthreads();
    if (t < a.ws) {
      uint32_t* post = a.sig[t] + bid * a.sig_stride + a.rank;
      long spins = 0;
      uint32_t expect = 0u;
      while (!__hip_atomic_compare_exchange_strong(post, &expect, 1u, __ATOMIC_RELEASE, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM)) {
        expect = 0u;
        __builtin_amdgcn_s_sleep(1);
        if (++spins > a.spin_limit) { note_timeout(a.timeouts); break; }
      }
      uint32_t* wait = a.sig[a.rank] + bid * a.sig_stride + t;
      spins = 0;
      expect = 1u;
      while (!__hip_atomic_compare_exchange_strong(wait, &expect, 0u, __ATOMIC_ACQUIRE, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM)) {
        expect = 1u;
        __builtin_amdgcn_s_sleep(1);
        if (++spins > a.spin_limit) { note_timeout(a.timeouts); break; }
      }
    }
    __syncthreads();
    return;
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // my write-through stores have been acknowledged by their home memory
  __syncthreads();
  if (t < a.ws) {
    uint32_t* post = a.sig[t] + bid * a.sig_stride + a.rank;  // my flag in peer t's pad
    long spins = 0;
    while (__hip_atomic_load(post, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) != 0u) {  // the peer has not consumed my last post yet
      __builtin_amdgcn_s_sleep(1);
      if (++spins > a.spin_limit) {
        note_timeout(a.timeouts);
        break;
      }
    }
    __hip_atomic_store(post, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    uint32_t* wait = a.sig[a.rank] + bid * a.sig_stride + t;  // peer t's flag in my pad
    spins = 0;
    while (__hip_atomic_load(wait, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) != 1u) {
      __builtin_amdgcn_s_sleep(1);
      if (++spins > a.spin_limit) {
        note_timeout(a.timeouts);
        break;
      }
    }
    __hip_atomic_store(wait, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
  }
  __syncthreads();
}

// kWs > 0: world size known at compile time - the peer loops are fully unrolled and a thread has
// 4 vectors x kWs peers = up to 32 remote 16-byte reads in flight before the first add (the row of
// a peer is only 2 * hidden bytes, so the reduce-scatter phase is a latency problem: one read in
// flight per thread moves ~0.25 of what the links can carry).  kWs == 0: any world size <= 8,
// runtime loop over the peers (still all vectors of a peer in flight).
// Loads are bounded buffer loads: lanes past the end of the row read 0 without a branch.
// kVec = 16-byte vectors per thread and row: 4 for hidden <= 8192 (x[4][8] + acc: ~200 VGPRs, two
// workgroups per CU), 8 up to 16384 (two passes of 4).
// The body takes the workgroup's index / the grid size as arguments: the product kernel passes blockIdx.x / gridDim.x,
// the development loopback kernel below runs the workgroups of ALL ranks of a world in one grid (blockIdx.y = rank).
template <int kWs, int kVec>
__device__ __forceinline__ void ht_body(const HtArgs& a, int bid, int nblk, float* red) {
  const int nvec = a.hidden >> 3;
  const unsigned row_bytes = static_cast<unsigned>(a.hidden) * 2u;
  // the weight row does not depend on the peers: its vectors are loaded once per workgroup, in flight across the opening
  // barrier (bounded loads: lanes past the row end read 0 and are never used).  Not with 8 vectors per thread beside the
  // rows of many peers (x[4][8] + acc[8][8] + residual: the registers of two workgroups per CU are spent).
  constexpr bool kHoistW = (kVec == 4 && kWs <= 4) || (kWs > 0 && kWs <= 2);
  u32x4 wv[kVec];
  if constexpr (kHoistW) {
    const auto rw = make_rsrc(uniform_ptr(a.w), row_bytes);
#pragma unroll
    for (int i = 0; i < kVec; ++i) wv[i] = buf_ld16<0>(rw, (threadIdx.x + i * kThreads) * 16, 0);
  }
  signal_barrier(a, bid);  // every rank's input is in place
  for (int row = bid; row < a.rows; row += nblk) {
    const long roff = static_cast<long>(row) * a.hidden;
    // the residual row rides with the peers' rows: ONE memory round trip per row (round 5 loaded it inside the
    // normalisation, behind the sum: two dependent round trips per row - ws = 1, T = 4096: 93.6 us for 268 MB)
    u32x4 rv[kVec];
    {
      const auto rr = make_rsrc(uniform_ptr(a.residual + roff), row_bytes);
#pragma unroll
      for (int i = 0; i < kVec; ++i) rv[i] = buf_ld16<0>(rr, (threadIdx.x + i * kThreads) * 16, 0);
    }
    float acc[kVec][8];
#pragma unroll
    for (int i = 0; i < kVec; ++i)
#pragma unroll
      for (int j = 0; j < 8; ++j) acc[i][j] = 0.f;
    // the whole byte offset goes into the VGPR offset: the descriptor's range check covers
    // voffset + immediate only, an SGPR offset would slip lanes past the row end through
    const int voff = threadIdx.x * 16;
#pragma unroll
    for (int half = 0; half < kVec / 4; ++half) {
      if constexpr (kWs > 0) {
        u32x4 x[4][kWs];
#pragma unroll
        for (int p = 0; p < kWs; ++p) {
          const auto rs = make_rsrc(uniform_ptr(a.in[p] + roff), row_bytes);
#pragma unroll
          for (int i = 0; i < 4; ++i) x[i][p] = buf_ld16<kAuxSys>(rs, voff + (half * 4 + i) * kThreads * 16, 0);
        }
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
          for (int p = 0; p < kWs; ++p)  // fixed rank order: deterministic sums
#pragma unroll
            for (int j = 0; j < 4; ++j) {
              acc[half * 4 + i][2 * j] += bf16lo_to_f32(x[i][p][j]);
              acc[half * 4 + i][2 * j + 1] += bf16hi_to_f32(x[i][p][j]);
            }
      } else {
        for (int p = 0; p < a.ws; ++p) {
          const auto rs = make_rsrc(uniform_ptr(a.in[p] + roff), row_bytes);
          u32x4 x[4];
#pragma unroll
          for (int i = 0; i < 4; ++i) x[i] = buf_ld16<kAuxSys>(rs, voff + (half * 4 + i) * kThreads * 16, 0);
#pragma unroll
          for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
              acc[half * 4 + i][2 * j] += bf16lo_to_f32(x[i][j]);
              acc[half * 4 + i][2 * j + 1] += bf16hi_to_f32(x[i][j]);
            }
        }
      }
    }
    u32x4 y[kVec];
    residual_rmsnorm(acc, nvec, a.hidden, a.residual + roff, a.out_residual + roff, a.w, a.eps, y, red, rv,
                     kHoistW ? wv : nullptr);
    const int nws = kWs > 0 ? kWs : a.ws;
#pragma unroll
    for (int i = 0; i < kVec; ++i) {
      const int v = threadIdx.x + i * kThreads;
      if (v < nvec)
        for (int p = 0; p < nws; ++p) st16_sys(a.out[p] + roff + v * 8, y[i]);
    }
  }
  signal_barrier(a, bid, true);  // my rows are in every peer (write-through stores, retired inside); every rank's rows have landed here
}
template <int kWs, int kVec>
__global__ __launch_bounds__(kThreads) void ht_kernel(const HtArgs a) {
  __shared__ float red[4];
  ht_body<kWs, kVec>(a, blockIdx.x, gridDim.x, red);
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
  int* timeouts;
  long spin_limit;
  float eps;
  int rows, hidden, rank, ws;
  // one_shot (the entry's `use_two_shot = false`; the reference takes the flag and runs its two-shot kernel whatever it says,
  // src/allreduce/entry.cc:84-181): every rank pushes its rows to EVERY rank - slot image [t][rank][hidden] - and every
  // rank sums the ws copies of a row itself, in rank order, in fp32, rounded to bf16 once: the arithmetic of the owner in
  // the two-shot form, so the results are bit-identical - with ws times the bytes on the links and one hop instead of two
  // (no owner, no broadcast, nothing to wait for but the peers' pushes).  Meant for a handful of rows; which form is faster
  // at which size is a question for a node with real links (none was available: DESIGN 3.4).
  // fuse_norm = 0 (the reference's rmsnorm_fusion = false, `wait_for_results` in its kernel, low_latency.cu:119-140): the
  // reduced rows are the output - no residual, no norm.
  int one_shot, fuse_norm;
};

__device__ __forceinline__ void ll_scatter_body(const LlArgs& a, int bid, int nblk) {
  const uint32_t cur = a.flags[0] % 3u;
  const uint32_t slot_bytes = a.flags[2];
  const uint32_t nxt = (cur + 1u) % 3u;
  // 1. clean the slot the NEXT call will use: nobody reads or writes it any more (everyone has
  //    finished the call before the previous one, or this call could not have started)
  {
    const uint32_t d0 = a.flags[4], d1 = a.flags[5], d2 = a.flags[6];  // one round trip with `cur`, not one behind it
    const uint32_t dirty = nxt == 0u ? d0 : nxt == 1u ? d1 : d2;
    uint8_t* base = a.local_ws + static_cast<long>(nxt) * slot_bytes;
    const u32x4 s = u32x4{kSentinel, kSentinel, kSentinel, kSentinel};
    for (long o = (static_cast<long>(bid) * kThreads + threadIdx.x) * 16; o < dirty;
         o += static_cast<long>(nblk) * kThreads * 16)
      st16_sys(base + o, s);
  }
  // 2. push my rows to their owners: owner's slot [t / ws][rank][hidden] (one shot: to every rank, slot [t][rank][hidden])
  const int nvec = a.hidden >> 3;
  for (int t = bid; t < a.rows; t += nblk) {
    const uint16_t* src = a.x + static_cast<long>(t) * a.hidden;
    if (a.one_shot) {
      const long off = static_cast<long>(cur) * slot_bytes + (static_cast<long>(t) * a.ws + a.rank) * a.hidden * 2;
      for (int v = threadIdx.x; v < nvec; v += kThreads) {
        const u32x4 xv = sanitize(ld16(src + v * 8));
        for (int p = 0; p < a.ws; ++p) st16_sys(reinterpret_cast<uint8_t*>(a.peers[p]) + off + v * 16, xv);
      }
      continue;
    }
    const int owner = t % a.ws;
    uint8_t* dst = reinterpret_cast<uint8_t*>(a.peers[owner]) + static_cast<long>(cur) * slot_bytes +
                   (static_cast<long>(t / a.ws) * a.ws + a.rank) * a.hidden * 2;
    for (int v = threadIdx.x; v < nvec; v += kThreads) st16_sys(dst + v * 16, sanitize(ld16(src + v * 8)));
  }
}
__global__ __launch_bounds__(kThreads) void ll_scatter_kernel(const LlArgs a) { ll_scatter_body(a, blockIdx.x, gridDim.x); }

// Poll `kN` x `nper` 16-byte vectors of this thread until none carries the sentinel: every poll round
// issues ALL the loads before it looks at the first one (local memory, system scope), instead of one
// load -> wait -> check per vector.  Out-of-row lanes read 0 (bounded descriptor) = "arrived".
template <int kVec, int kN>
__device__ __forceinline__ void poll_vectors(u32x4 (&v)[kVec][kN], const uint8_t* base, long stride,
                                             unsigned row_bytes, int nper, int* timeouts, long spin_limit) {
  long spins = 0;
  while (true) {
    bool ok = true;
#pragma unroll
    for (int p = 0; p < kN; ++p) {
      if (p >= nper) break;
      const auto rs = make_rsrc(uniform_ptr(base + p * stride), row_bytes);
#pragma unroll
      for (int i = 0; i < kVec; ++i) v[i][p] = buf_ld16<kAuxSys>(rs, (threadIdx.x + i * kThreads) * 16, 0);
    }
#pragma unroll
    for (int p = 0; p < kN; ++p) {
      if (p >= nper) break;
#pragma unroll
      for (int i = 0; i < kVec; ++i) ok = ok && arrived(v[i][p]);
    }
    if (ok) break;
    __builtin_amdgcn_s_sleep(2);
    asm volatile("" ::: "memory");  // the next round must really re-load
    if (++spins > spin_limit) {
      note_timeout(timeouts);
      break;
    }
  }
}

template <int kVec>
__device__ __forceinline__ void ll_reduce_norm_body(const LlArgs& a, int bid, int nblk, float* red) {
  const uint32_t cur = a.flags[0] % 3u;
  const uint32_t slot_bytes = a.flags[2];
  const int nvec = a.hidden >> 3;
  const int n_pad = (a.rows + a.ws - 1) / a.ws * a.ws;
  const long row_bytes = static_cast<long>(a.hidden) * 2;
  const long bcast_off = static_cast<long>(cur) * slot_bytes + static_cast<long>(n_pad) * row_bytes;

  // slot bookkeeping: the last workgroup to arrive (all have read `cur` by then) rotates the slots
  // (looking at the ticket only after the rows was measured: T 8 9.0 -> 8.8 us, T 128 11.5 -> 12.7 - not kept)
  __syncthreads();
  if (threadIdx.x == 0) {
    const uint32_t old = atomicAdd(&a.flags[8], 1u);
    if (old == static_cast<uint32_t>(nblk) - 1u) {
      a.flags[4 + cur] = static_cast<uint32_t>(a.one_shot ? static_cast<long>(a.rows) * a.ws * row_bytes : 2 * n_pad * row_bytes);
      a.flags[8] = 0u;
      a.flags[1] = (cur + 2u) % 3u;
      __threadfence();
      a.flags[0] = (cur + 1u) % 3u;
    }
  }

  for (int t = bid; t < a.rows; t += nblk) {
    float acc[kVec][8];
#pragma unroll
    for (int i = 0; i < kVec; ++i)
#pragma unroll
      for (int j = 0; j < 8; ++j) acc[i][j] = 0.f;
    if (a.one_shot || t % a.ws == a.rank) {
      // owner (one shot: every rank, of every row): wait for every rank's copy of row t (they land in MY memory), reduce,
      // broadcast (two-shot form only)
      const uint8_t* src = a.local_ws + static_cast<long>(cur) * slot_bytes +
                           static_cast<long>(a.one_shot ? t : t / a.ws) * a.ws * row_bytes;
      for (int p0 = 0; p0 < a.ws; p0 += 4) {  // 4 peers x all vectors in flight per round
        u32x4 xv[kVec][4];
        const int nper = a.ws - p0 < 4 ? a.ws - p0 : 4;
        poll_vectors<kVec, 4>(xv, src + p0 * row_bytes, row_bytes, static_cast<unsigned>(row_bytes), nper, a.timeouts, a.spin_limit);
#pragma unroll
        for (int p = 0; p < 4; ++p) {
          if (p >= nper) break;
#pragma unroll
          for (int i = 0; i < kVec; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
              acc[i][2 * j] += bf16lo_to_f32(xv[i][p][j]);
              acc[i][2 * j + 1] += bf16hi_to_f32(xv[i][p][j]);
            }
        }
      }
#pragma unroll
      for (int i = 0; i < kVec; ++i) {
        const int v = threadIdx.x + i * kThreads;
        if (v < nvec) {
          u32x4 sum;
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            sum[j] = pack_bf16x2(acc[i][2 * j], acc[i][2 * j + 1]);
            acc[i][2 * j] = bf16lo_to_f32(sum[j]);
            acc[i][2 * j + 1] = bf16hi_to_f32(sum[j]);
          }
          sum = sanitize(sum);
          if (!a.one_shot)
            for (int p = 0; p < a.ws; ++p)
              if (p != a.rank)
                st16_sys(reinterpret_cast<uint8_t*>(a.peers[p]) + bcast_off + t * row_bytes + v * 16, sum);
        }
      }
    } else {
      u32x4 xv[kVec][1];
      poll_vectors<kVec, 1>(xv, a.local_ws + bcast_off + t * row_bytes, 0, static_cast<unsigned>(row_bytes), 1, a.timeouts, a.spin_limit);
#pragma unroll
      for (int i = 0; i < kVec; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          acc[i][2 * j] = bf16lo_to_f32(xv[i][0][j]);
          acc[i][2 * j + 1] = bf16hi_to_f32(xv[i][0][j]);
        }
    }
    u32x4 y[kVec];
    const long roff = static_cast<long>(t) * a.hidden;
    if (a.fuse_norm) {
      residual_rmsnorm(acc, nvec, a.hidden, a.residual + roff, a.residual_out + roff, a.w, a.eps, y, red);
    } else {  // the all-reduce alone: acc holds the bf16-rounded sums
#pragma unroll
      for (int i = 0; i < kVec; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) y[i][j] = pack_bf16x2(acc[i][2 * j], acc[i][2 * j + 1]);
    }
#pragma unroll
    for (int i = 0; i < kVec; ++i) {
      const int v = threadIdx.x + i * kThreads;
      if (v < nvec) st16(a.y + roff + v * 8, y[i]);
    }
    __syncthreads();
  }
}
template <int kVec>
__global__ __launch_bounds__(kThreads) void ll_reduce_norm_kernel(const LlArgs a) {
  __shared__ float red[4];
  ll_reduce_norm_body<kVec>(a, blockIdx.x, gridDim.x, red);
}
// Both phases in ONE launch (round 5): workgroup b pushes rows b, b + grid, ... and then reduces / polls the same
// rows, so it only ever waits for workgroup b of the OTHER ranks, and those push before they wait - the property the
// two-kernel form has ("never waits on a workgroup of the same GPU") is kept.  Both bodies read the slot state before
// this workgroup arrives at the rotation counter, so the state they see is the same.  Used when the whole grid is
// resident at once (no assumption about the order in which a GPU starts workgroups); larger grids keep two launches.
template <int kVec>
__global__ __launch_bounds__(kThreads) void ll_fused_kernel(const LlArgs a) {
  __shared__ float red[4];
  ll_scatter_body(a, blockIdx.x, gridDim.x);
  ll_reduce_norm_body<kVec>(a, blockIdx.x, gridDim.x, red);
}

#ifdef HPC_DEV
// ---------------------------------------------------------------------------------------------
// Development loopback: the workgroups of ALL ranks of a world in ONE grid on one device (blockIdx.y = rank), every
// rank with its own argument block over ordinary device allocations - the world-size-8 instantiations of the product
// bodies above run and rendezvous for real on the single GPU of a test box (eight PROCESSES time-slice the device
// and never make progress together; one grid is co-resident by construction when it is small enough).
// ---------------------------------------------------------------------------------------------
template <int kWs, int kVec>
__global__ __launch_bounds__(kThreads) void ht_loopback_kernel(const HtArgs* all) {
  __shared__ float red[4];
  __shared__ HtArgs a;
  if (threadIdx.x == 0) a = all[blockIdx.y];
  __syncthreads();
  ht_body<kWs, kVec>(a, blockIdx.x, gridDim.x, red);
}
__global__ __launch_bounds__(kThreads) void ll_scatter_loopback_kernel(const LlArgs* all) {
  __shared__ LlArgs a;
  if (threadIdx.x == 0) a = all[blockIdx.y];
  __syncthreads();
  ll_scatter_body(a, blockIdx.x, gridDim.x);
}
template <int kVec>
__global__ __launch_bounds__(kThreads) void ll_reduce_norm_loopback_kernel(const LlArgs* all) {
  __shared__ float red[4];
  __shared__ LlArgs a;
  if (threadIdx.x == 0) a = all[blockIdx.y];
  __syncthreads();
  ll_reduce_norm_body<kVec>(a, blockIdx.x, gridDim.x, red);
}
template <int kVec>
__global__ __launch_bounds__(kThreads) void ll_fused_loopback_kernel(const LlArgs* all) {
  __shared__ float red[4];
  __shared__ LlArgs a;
  if (threadIdx.x == 0) a = all[blockIdx.y];
  __syncthreads();
  ll_scatter_body(a, blockIdx.x, gridDim.x);
  ll_reduce_norm_body<kVec>(a, blockIdx.x, gridDim.x, red);
}
#endif  // HPC_DEV

}  // namespace ar
}  // namespace hpc

using namespace hpc::ar;

namespace {
// development key 10 = n: bounded spins give up after 2^n rounds (tests of the lost-peer path)
long spin_limit() {
  const int n = hpc_dev_tuning_get(10);
  return n > 0 && n < 40 ? 1L << n : kSpinLimit;
}
// one word of pinned, device-mapped host memory shared by every device of the process
int* timeout_word() {
  static int* host = [] {
    int* p = nullptr;
    if (hipHostMalloc(reinterpret_cast<void**>(&p), 64, hipHostMallocPortable | hipHostMallocMapped) != hipSuccess)
      return static_cast<int*>(nullptr);
    *p = 0;
    return p;
  }();
  return host;
}
}  // namespace

extern "C" int hpc_allreduce_timeouts(void) {
  int* p = timeout_word();
  return p ? *reinterpret_cast<volatile int*>(p) : -1;
}

extern "C" int hpc_allreduce_reset_timeouts(void) {
  int* p = timeout_word();
  if (!p) return HPC_ERR_LAUNCH;
  *reinterpret_cast<volatile int*>(p) = 0;
  return HPC_OK;
}

extern "C" int hpc_fuse_allreduce_rmsnorm_high_throughput_grid(int world_size, int num_max_blocks,
                                                               int signal_pad_words) {
  if (world_size < 1 || world_size > kMaxWs || num_max_blocks <= 0 || signal_pad_words <= 0)
    return HPC_ERR_INVALID;
  // The barriers pair block b of this rank with block b of every peer, so the grid must be the same
  // on every rank whatever slice each of them was handed: it is a function of rank-invariant inputs
  // only (num_max_blocks, world size, pad capacity) - never of this rank's row count (the reference
  // launches num_max_blocks blocks for the same reason, high_throughput.cu:135-150).  One row is
  // only 2 * hidden bytes per peer, so the reference's SM-count-sized default leaves most of an
  // MI355X idle: at least two workgroups per CU (development key 11 = n replaces the floor of 512 -
  // the tests that run two ranks on ONE GPU need both ranks' grids co-resident).  Block b posts into
  // words [b * ws, (b + 1) * ws) of a pad.
  // Round 6: TWO workgroups per CU (512).  The kernels hold <= 256 registers and a few bytes of LDS, so two 4-wave workgroups
  // of a rank are resident per CU and one's memory round trip runs under the other's reduction.  Measured at ws = 1, H 8192
  // (profiles/round6_allreduce_ab.txt, us per call inside a replay, floor 128 / 256 / 512): T = 512 13.0 / 9.9 / 8.9, T = 4096
  // 72.9 / 49.9 / 45.3, T = 16384 376 / 218 / 196.  (With the system-scope fences of rounds 1-5 in the barriers every
  // workgroup cost 85 ns of serialised L2 write-back / invalidate and 256 beat 512: 79 against 103 us at T = 4096.)
  // A constant, not the device's CU count: the value must be the same on every rank.
  const int floor_dev = hpc_dev_tuning_get(11);
  const int floor_blocks = floor_dev > 0 ? floor_dev : 512;
  int grid = num_max_blocks > floor_blocks ? num_max_blocks : floor_blocks;
  if (grid > signal_pad_words / world_size) grid = signal_pad_words / world_size;
  return grid > 0 ? grid : HPC_ERR_INVALID;
}

// Words between the flag groups of consecutive blocks in a signal pad: block b uses words [b * stride, b * stride + ws).
// Rounds 1-5 packed them (stride = ws, the reference's layout): the 256 blocks' flags of a rank sat in 1-8 KB of UNCACHED
// memory, behind one or two memory channels.  Now the groups are spread over the whole pad in 64-byte units: stride =
// floor(pad_words / grid) rounded down to a multiple of 16 words, never below ws.  Rank-invariant like the grid.  Measured
// at ws = 1 it is worth little (T = 512: 31.4 -> 29.7 us per call with the old fences in place; the fixed cost was the
// fences, see signal_barrier) - kept because with eight ranks every word is polled across a link and eight times as many
// words share a line; development key 48 = 1 restores the packed layout.
extern "C" int hpc_fuse_allreduce_rmsnorm_high_throughput_signal_stride(int world_size, int grid, int signal_pad_words) {
  if (world_size < 1 || world_size > kMaxWs || grid <= 0 || signal_pad_words < grid * world_size) return HPC_ERR_INVALID;
  if (hpc_dev_tuning_get(48) == 1) return world_size;  // development key 48 = 1: the packed layout of rounds 1-5 (A/B)
  const int spread = (signal_pad_words / grid) & ~15;
  return spread > world_size ? spread : world_size;
}

extern "C" int hpc_fuse_allreduce_rmsnorm_high_throughput_async(
    const void* const* peer_x_ptrs, void* const* peer_out_ptrs, void* const* peer_signal_ptrs,
    const void* residual_ptr, void* out_residual_ptr, const void* weight_ptr, float rms_norm_eps,
    int num_rows, int hidden_size, int rank, int world_size, int num_max_blocks, int signal_pad_words,
    hipStream_t stream) {
  if (int* tw = timeout_word(); tw && *reinterpret_cast<volatile int*>(tw) != 0) return HPC_ERR_TIMEOUT;
  if (!peer_x_ptrs || !peer_out_ptrs || !peer_signal_ptrs || !residual_ptr || !out_residual_ptr || !weight_ptr)
    return HPC_ERR_INVALID;
  if (world_size < 1 || world_size > kMaxWs || rank < 0 || rank >= world_size) return HPC_ERR_UNSUPPORTED;
  if ((hidden_size & 7) || hidden_size <= 0 || hidden_size > kMaxVec * kThreads * 8) return HPC_ERR_UNSUPPORTED;
  if (num_max_blocks <= 0 || num_rows < 0) return HPC_ERR_INVALID;
  const int grid = hpc_fuse_allreduce_rmsnorm_high_throughput_grid(world_size, num_max_blocks, signal_pad_words);
  if (grid <= 0) return HPC_ERR_INVALID;  // pad too small for even one block
  HtArgs a;
  a.timeouts = timeout_word();
  a.spin_limit = spin_limit();
  if (!a.timeouts) return HPC_ERR_LAUNCH;
  if (*reinterpret_cast<volatile int*>(a.timeouts) != 0) return HPC_ERR_TIMEOUT;
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
  a.cas_barrier = hpc_dev_tuning_get(47) == 1;
  a.sig_stride = hpc_fuse_allreduce_rmsnorm_high_throughput_signal_stride(world_size, grid, signal_pad_words);
  if (a.sig_stride < world_size) return HPC_ERR_INVALID;
#define HPC_HT_LAUNCH(WS)                                          \
  if (hidden_size <= 4 * kThreads * 8)                              \
    ht_kernel<WS, 4><<<grid, kThreads, 0, stream>>>(a);             \
  else                                                              \
    ht_kernel<WS, 8><<<grid, kThreads, 0, stream>>>(a)
  switch (hpc_dev_tuning_get(9) == 1 ? 0 : world_size) {  // key 9 = 1: runtime-world-size kernel
    case 1: HPC_HT_LAUNCH(1); break;
    case 2: HPC_HT_LAUNCH(2); break;
    case 4: HPC_HT_LAUNCH(4); break;
    case 8: HPC_HT_LAUNCH(8); break;
    default: HPC_HT_LAUNCH(0); break;
  }
#undef HPC_HT_LAUNCH
  HPC_CHECK_LAUNCH();
  return HPC_OK;
}

// workgroups of the fused low-latency kernel the current device holds at once (per instantiation, cached per device)
static int ll_fused_capacity(bool wide, bool resident) {
  static int cap[16][2], cu_count[16];
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 16) return 0;
  if (cap[dev][wide] == 0) {
    int per_cu = 0, cus = 0;
    hipError_t e = wide ? hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, ll_fused_kernel<8>, kThreads, 0)
                        : hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, ll_fused_kernel<4>, kThreads, 0);
    if (e != hipSuccess || hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess) {
      (void)hipGetLastError();
      return 0;
    }
    cap[dev][wide] = per_cu * cus > 0 ? per_cu * cus : -1;
    cu_count[dev] = cus;
  }
  if (cap[dev][wide] <= 0) return 0;
  return resident || cap[dev][wide] < cu_count[dev] ? cap[dev][wide] : cu_count[dev];
}

extern "C" int hpc_fuse_allreduce_rmsnorm_low_latency_async(
    void* output_ptr, void* residual_out_ptr, const void* input_ptr, const void* data_buffer_ptrs_dev,
    void* local_workspace_ptr, void* buffer_flags_dev, const void* residual_in_ptr,
    const void* weight_ptr, float rms_norm_eps, int num_tokens, int hidden_size, int rank,
    int world_size, int64_t workspace_bytes, hipStream_t stream) {
  return hpc_allreduce_low_latency_async(output_ptr, residual_out_ptr, input_ptr, data_buffer_ptrs_dev, local_workspace_ptr,
                                         buffer_flags_dev, residual_in_ptr, weight_ptr, rms_norm_eps, num_tokens, hidden_size,
                                         rank, world_size, workspace_bytes, 1, 1, stream);
}

extern "C" int hpc_allreduce_low_latency_async(
    void* output_ptr, void* residual_out_ptr, const void* input_ptr, const void* data_buffer_ptrs_dev,
    void* local_workspace_ptr, void* buffer_flags_dev, const void* residual_in_ptr,
    const void* weight_ptr, float rms_norm_eps, int num_tokens, int hidden_size, int rank,
    int world_size, int64_t workspace_bytes, int rmsnorm_fusion, int use_two_shot, hipStream_t stream) {
  if (int* tw = timeout_word(); tw && *reinterpret_cast<volatile int*>(tw) != 0) return HPC_ERR_TIMEOUT;
  if (!output_ptr || !input_ptr || !data_buffer_ptrs_dev || !local_workspace_ptr || !buffer_flags_dev) return HPC_ERR_INVALID;
  if (rmsnorm_fusion && (!residual_out_ptr || !residual_in_ptr || !weight_ptr)) return HPC_ERR_INVALID;
  if (world_size < 1 || world_size > 64 || rank < 0 || rank >= world_size) return HPC_ERR_UNSUPPORTED;
  if ((hidden_size & 7) || hidden_size <= 0 || hidden_size > kMaxVec * kThreads * 8) return HPC_ERR_UNSUPPORTED;
  if (num_tokens <= 0) return HPC_OK;
  const int64_t n_pad = (num_tokens + world_size - 1) / world_size * world_size;
  // a slot holds the scatter + broadcast rows of the two-shot form, or every rank's copy of every row of the one-shot form
  const int64_t need = 3 * (use_two_shot ? 2 * n_pad : static_cast<int64_t>(num_tokens) * world_size) * hidden_size * 2;
  if (workspace_bytes < need) return HPC_ERR_INVALID;
  LlArgs a;
  a.timeouts = timeout_word();
  a.spin_limit = spin_limit();
  if (!a.timeouts) return HPC_ERR_LAUNCH;
  if (*reinterpret_cast<volatile int*>(a.timeouts) != 0) return HPC_ERR_TIMEOUT;
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
  a.one_shot = use_two_shot ? 0 : 1;
  a.fuse_norm = rmsnorm_fusion ? 1 : 0;
  const int grid = num_tokens < 2048 ? num_tokens : 2048;
  const bool wide = hidden_size > 4 * kThreads * 8;
  // one launch up to one workgroup per CU (measured at ws = 1, H 8192: T 8 / 128 9.0 / 11.5 us against 10.0 / 12.5 with two
  // launches, T 512 25.6 against 23.1 - with two workgroups per CU the second's scatter waits behind the first's poll)
  const int key35 = hpc_dev_tuning_get(35);  // development: 1 = always two launches, 2 = one launch whenever resident
  if (key35 != 1 && grid <= ll_fused_capacity(wide, key35 == 2)) {
    if (!wide)
      ll_fused_kernel<4><<<grid, kThreads, 0, stream>>>(a);
    else
      ll_fused_kernel<8><<<grid, kThreads, 0, stream>>>(a);
    HPC_CHECK_LAUNCH();
    return HPC_OK;
  }
  ll_scatter_kernel<<<grid, kThreads, 0, stream>>>(a);
  HPC_CHECK_LAUNCH();
  if (!wide)
    ll_reduce_norm_kernel<4><<<grid, kThreads, 0, stream>>>(a);
  else
    ll_reduce_norm_kernel<8><<<grid, kThreads, 0, stream>>>(a);
  HPC_CHECK_LAUNCH();
  return HPC_OK;
}

#ifdef HPC_DEV
// ---- development loopback entries (tests/test_allreduce.py::test_allreduce_rmsnorm_ws8_loopback) ---------------------
// Tables are rank-major: entry [r * world_size + p] is what rank r would pass as element p of the product entry's
// table; [r] what rank r would pass as the scalar argument.  One launch runs every rank (see the loopback kernels).
namespace {
template <typename T>
T* loopback_args_on_device(const T* host, int n, hipStream_t stream) {
  static void* dev = nullptr;  // big enough for either argument block
  if (!dev && hipMalloc(&dev, 8 * (sizeof(HtArgs) > sizeof(LlArgs) ? sizeof(HtArgs) : sizeof(LlArgs))) != hipSuccess)
    return nullptr;
  if (hipMemcpyAsync(dev, host, n * sizeof(T), hipMemcpyHostToDevice, stream) != hipSuccess) return nullptr;
  return static_cast<T*>(dev);
}
}  // namespace

extern "C" int hpc_dev_allreduce_loopback_ht(const void* const* peer_x, void* const* peer_out, void* const* peer_sig,
                                             const void* const* residual, void* const* out_residual,
                                             const int* num_rows, const void* weight, float eps, int hidden,
                                             int world_size, int num_max_blocks, int pad_words, hipStream_t stream) {
  if (world_size < 1 || world_size > kMaxWs || (hidden & 7) || hidden <= 0 || hidden > kMaxVec * kThreads * 8)
    return HPC_ERR_UNSUPPORTED;
  const int grid = hpc_fuse_allreduce_rmsnorm_high_throughput_grid(world_size, num_max_blocks, pad_words);
  if (grid <= 0) return HPC_ERR_INVALID;
  static HtArgs host[kMaxWs];
  for (int r = 0; r < world_size; ++r) {
    HtArgs& a = host[r];
    a.timeouts = timeout_word();
    a.spin_limit = spin_limit();
    if (!a.timeouts) return HPC_ERR_LAUNCH;
    for (int p = 0; p < kMaxWs; ++p) {
      a.in[p] = p < world_size ? static_cast<const uint16_t*>(peer_x[r * world_size + p]) : nullptr;
      a.out[p] = p < world_size ? static_cast<uint16_t*>(peer_out[r * world_size + p]) : nullptr;
      a.sig[p] = p < world_size ? static_cast<uint32_t*>(peer_sig[r * world_size + p]) : nullptr;
    }
    a.residual = static_cast<const uint16_t*>(residual[r]);
    a.out_residual = static_cast<uint16_t*>(out_residual[r]);
    a.w = static_cast<const uint16_t*>(weight);
    a.eps = eps;
    a.rows = num_rows[r];
    a.hidden = hidden;
    a.rank = r;
    a.ws = world_size;
    a.cas_barrier = hpc_dev_tuning_get(47) == 1;
    a.sig_stride = hpc_fuse_allreduce_rmsnorm_high_throughput_signal_stride(world_size, grid, pad_words);
    if (a.sig_stride < world_size) return HPC_ERR_INVALID;
  }
  const HtArgs* dev = loopback_args_on_device(host, world_size, stream);
  if (!dev) return HPC_ERR_LAUNCH;
  const dim3 g(grid, world_size);
#define HPC_HT_LOOP(WS)                                                  \
  if (hidden <= 4 * kThreads * 8)                                        \
    ht_loopback_kernel<WS, 4><<<g, kThreads, 0, stream>>>(dev);          \
  else                                                                   \
    ht_loopback_kernel<WS, 8><<<g, kThreads, 0, stream>>>(dev)
  switch (hpc_dev_tuning_get(9) == 1 ? 0 : world_size) {
    case 1: HPC_HT_LOOP(1); break;
    case 2: HPC_HT_LOOP(2); break;
    case 4: HPC_HT_LOOP(4); break;
    case 8: HPC_HT_LOOP(8); break;
    default: HPC_HT_LOOP(0); break;
  }
#undef HPC_HT_LOOP
  HPC_CHECK_LAUNCH();
  return HPC_OK;
}

extern "C" int hpc_dev_allreduce_loopback_ll(void* const* output, void* const* residual_out, const void* const* input,
                                             const void* data_buffer_ptrs_dev, void* const* local_workspace,
                                             void* const* buffer_flags, const void* const* residual_in,
                                             const void* weight, float eps, int num_tokens, int hidden, int world_size,
                                             int64_t workspace_bytes, hipStream_t stream) {
  if (world_size < 1 || world_size > kMaxWs || (hidden & 7) || hidden <= 0 || hidden > kMaxVec * kThreads * 8)
    return HPC_ERR_UNSUPPORTED;
  if (num_tokens <= 0) return HPC_OK;
  const int64_t n_pad = (num_tokens + world_size - 1) / world_size * world_size;
  if (workspace_bytes < 3 * 2 * n_pad * hidden * 2) return HPC_ERR_INVALID;
  static LlArgs host[kMaxWs];
  for (int r = 0; r < world_size; ++r) {
    LlArgs& a = host[r];
    a.timeouts = timeout_word();
    a.spin_limit = spin_limit();
    if (!a.timeouts) return HPC_ERR_LAUNCH;
    a.x = static_cast<const uint16_t*>(input[r]);
    a.peers = static_cast<const long*>(data_buffer_ptrs_dev);
    a.local_ws = static_cast<uint8_t*>(local_workspace[r]);
    a.flags = static_cast<uint32_t*>(buffer_flags[r]);
    a.residual = static_cast<const uint16_t*>(residual_in[r]);
    a.residual_out = static_cast<uint16_t*>(residual_out[r]);
    a.y = static_cast<uint16_t*>(output[r]);
    a.w = static_cast<const uint16_t*>(weight);
    a.eps = eps;
    a.rows = num_tokens;
    a.hidden = hidden;
    a.rank = r;
    a.ws = world_size;
    a.one_shot = hpc_dev_tuning_get(52) == 1;   // development key 52 = 1: the one-shot form in the loopback grid
    a.fuse_norm = hpc_dev_tuning_get(53) != 1;  // development key 53 = 1: the all-reduce alone (rmsnorm_fusion = false)
    if (a.one_shot && workspace_bytes < 3l * num_tokens * world_size * hidden * 2) return HPC_ERR_INVALID;
  }
  const LlArgs* dev = loopback_args_on_device(host, world_size, stream);
  if (!dev) return HPC_ERR_LAUNCH;
  const dim3 g(num_tokens < 2048 ? num_tokens : 2048, world_size);
  if (static_cast<int>(g.x * g.y) <= ll_fused_capacity(hidden > 4 * kThreads * 8, true) && hpc_dev_tuning_get(35) != 1) {
    if (hidden <= 4 * kThreads * 8)
      ll_fused_loopback_kernel<4><<<g, kThreads, 0, stream>>>(dev);
    else
      ll_fused_loopback_kernel<8><<<g, kThreads, 0, stream>>>(dev);
    HPC_CHECK_LAUNCH();
    return HPC_OK;
  }
  ll_scatter_loopback_kernel<<<g, kThreads, 0, stream>>>(dev);
  HPC_CHECK_LAUNCH();
  if (hidden <= 4 * kThreads * 8)
    ll_reduce_norm_loopback_kernel<4><<<g, kThreads, 0, stream>>>(dev);
  else
    ll_reduce_norm_loopback_kernel<8><<<g, kThreads, 0, stream>>>(dev);
  HPC_CHECK_LAUNCH();
  return HPC_OK;
}
#endif  // HPC_DEV

