// Paged-KV causal prefill attention, FP8 (e4m3) Q/K/V, bf16 output - gfx950.
//
// Replaces reference attention_with_kvcache_prefill_*_fp8_async (src/attention/prefill/prefill.h,
// kernels src/attention/prefill/sm90/**, entry src/attention/entry.cc:152-262): the chunked-prefill step
// that runs before decode on the same paged cache (q tokens of request b are the LAST seqlen_q[b] of its
// seqlens_kvcache[b] cached tokens; row s attends keys j <= (L_b - Sq_b) + s).
//
// MI355X design (first version): the decode kernel's tile machinery, re-mapped.  Same
// MFMA orientation (64 KV tokens on M, q rows on N: S^T = K Q^T, O^T = V^T P^T with
// v_mfma_f32_16x16x32_fp8_fp8), same full-row K loads transposed through a wave-private LDS tile, same
// private 8x8 V byte transposes, same base-2 online softmax with P quantised as e4m3(256 p) against the
// running max.  What changes: a wave owns 32 q rows (= 32/G consecutive positions x the G q heads of
// one kv head) and finishes alone - no split-KV, no merge; a workgroup = 4 waves = 128 consecutive rows
// that walk the KV tiles together (causal: each wave masks what its rows cannot see) and share every
// K/V tile through a double-buffered LDS stage filled one tile ahead.
#include "hpc_common.h"
#include "hpc_dev.h"
#include "../../include/hpc_amd.h"

namespace hpc {
namespace prefill {

struct Args {
  const void* q;
  const void* kcache;
  const void* vcache;
  const int* block_ids;
  const int* cu_seqlens_q;
  const int* seqlens_kv;
  uint16_t* y;
  const float* qscale;  // [B][Hq][qs_pad]
  const float* kscale;  // [1] or base of the per-token K-scale rows
  const float* vscale;  // [1] or [Hkv]
  int by_head;  // row mapping, see the kernel
  const uint8_t* block_mask;  // null, or [B][Hq][mask_tiles_m][mask_tiles_kv]: 128 x 128 (q pos x kv token) tiles
  int mask_tiles_m, mask_tiles_kv;
  int num_batch, num_head_q, num_head_kv, g_shift, page_shift, max_blocks, qs_pad;
  int ldq, ldy;
  long k_block_stride, k_token_stride, k_head_stride;  // elements (= bytes)
  long v_block_stride, v_token_stride, v_head_stride;
  long ks_block_stride, ks_row_stride, ks_head_stride;  // bytes
  float scale_log2;
};

constexpr int kThreads = 256;
constexpr int kWaves = 4;
constexpr int kNB = 2;                  // 16-row q blocks per wave
constexpr int kRowsPerWave = 16 * kNB;  // 32
constexpr float kNegInf = -__builtin_inff();
constexpr int kKRow = 128 + 16;  // padded LDS row of the K transpose tile
constexpr int kMaskCols = 512;    // block-sparse: mask columns (128-token tiles) cached per head

// ---- loads the compiler does not see (hand-counted vmcnt, as in attention_decode_v2.hip) ---------------------------
__device__ __forceinline__ i32x4 srd_of(const void* base) {  // every word pinned: an "s" operand must be provably uniform
  const uint64_t v = reinterpret_cast<uint64_t>(base);
  return i32x4{__builtin_amdgcn_readfirstlane(static_cast<int>(static_cast<uint32_t>(v))),
               __builtin_amdgcn_readfirstlane(static_cast<int>(static_cast<uint32_t>(v >> 32))), -1, 0x00020000};
}
__device__ __forceinline__ void ld_pair(u32x4& x0, u32x4& x1, int voff, i32x4 rs, int soff1) {
  const int s1 = __builtin_amdgcn_readfirstlane(soff1);
  asm volatile("s_nop 4\n\tbuffer_load_dwordx4 %0, %2, %3, 0 offen\n\tbuffer_load_dwordx4 %1, %2, %3, %4 offen"
               : "=&v"(x0), "=&v"(x1)
               : "v"(voff), "s"(rs), "s"(s1));
}
__device__ __forceinline__ void ld_one(float& x, int voff, i32x4 rs) {
  asm volatile("s_nop 4\n\tbuffer_load_dword %0, %1, %2, 0 offen" : "=&v"(x) : "v"(voff), "s"(rs));
}
// the set's loads have landed once at most kLeft younger loads are in flight
template <int kLeft>
__device__ __forceinline__ void wait_set(u32x4 (&k)[2], u32x4 (&v)[2], float& ks) {
  asm volatile("s_waitcnt vmcnt(%5)" : "+v"(k[0]), "+v"(k[1]), "+v"(v[0]), "+v"(v[1]), "+v"(ks) : "n"(kLeft));
}

template <int kN>
struct IntC {
  static constexpr int value = kN;
};
typedef int v2i32 __attribute__((ext_vector_type(2)));
typedef __attribute__((address_space(3))) v2i32 lds_v2i32;
typedef __attribute__((address_space(3))) uint8_t lds_u8;
// max / sum over the 4 lanes that share a q row (lane, lane ^ 16, lane ^ 32, lane ^ 48): gfx950 row swaps,
// permlane16_swap(x, x) = {[x0 x0 x2 x2], [x1 x1 x3 x3]} by 16-lane rows, permlane32_swap(y, y) = {[y0 y1 y0 y1], [y2 y3 y2 y3]}
__device__ __forceinline__ float row4_max(float x) {
  const auto a = __builtin_amdgcn_permlane16_swap(__float_as_uint(x), __float_as_uint(x), false, false);
  const float y = fmaxf(__uint_as_float(a[0]), __uint_as_float(a[1]));
  const auto b = __builtin_amdgcn_permlane32_swap(__float_as_uint(y), __float_as_uint(y), false, false);
  return fmaxf(__uint_as_float(b[0]), __uint_as_float(b[1]));
}
__device__ __forceinline__ float row4_sum(float x) {
  const auto a = __builtin_amdgcn_permlane16_swap(__float_as_uint(x), __float_as_uint(x), false, false);
  const float y = __uint_as_float(a[0]) + __uint_as_float(a[1]);
  const auto b = __builtin_amdgcn_permlane32_swap(__float_as_uint(y), __float_as_uint(y), false, false);
  return __uint_as_float(b[0]) + __uint_as_float(b[1]);
}

__device__ __forceinline__ long pack64(uint32_t lo, uint32_t hi) {
  return static_cast<long>(static_cast<uint64_t>(lo) | (static_cast<uint64_t>(hi) << 32));
}

// kQuant 1: q per token / per head, k and v per tensor.  kQuant 0: k per token / per head (scales in
// the page tail rows), v per kv head.
//
// K/V staging: every 64-token tile is fetched ONCE per workgroup - wave w loads token block w of K and
// of V (16 rows x 128 B each, full-row loads, one tile ahead, held in 4+4 VGPRs while the current tile
// computes) and drops them into a double-buffered LDS tile; one barrier per tile; all waves then read K
// in MFMA A-operand layout (ds_read_b128) and V^T through ds_read_b64_tr_b8 (round 3: the transposing LDS read
// hands every lane one dim of an 8-token x 16-dim tile = the A operand of O^T += V^T P^T; before, 16 plain reads
// and 48 v_perm byte shuffles per tile did that on the VALU).  The V tile is stored unpadded with its 16-byte chunks
// XOR-swizzled (slot = chunk ^ key(row), key = (row / 2) % 4 | ((row / 16) % 2) * 4): the 16 rows x 2 halves of a
// transpose read's 32 lanes fall on 32 different 8-byte bank pairs.
template <int kQuant, bool kSparse>
__global__ __launch_bounds__(kThreads, 2) void prefill_fp8_kernel(const Args a) {
  // A STAGE = two 64-token tiles: one barrier, one round of fetch bookkeeping per 128 tokens (round 3).
  __shared__ float s_l[kWaves][16];
  __shared__ __attribute__((aligned(16))) uint8_t s_k[2][128 * kKRow];
  __shared__ __attribute__((aligned(1024))) uint8_t s_v[2][128 * 128];
  __shared__ __attribute__((aligned(16))) float s_ks[2][128];  // per-token K scales of the stage (kQuant 0)
  // the epilogue's row-major tile lives in the K stage (idle by then; every wave is past the loop's last barrier)
  static_assert(sizeof(float) * kWaves * 16 * (128 + 4) <= 2 * 128 * kKRow, "s_o aliases s_k");
  float (*s_o)[16][128 + 4] = reinterpret_cast<float (*)[16][128 + 4]>(&s_k[0][0]);
  // block-sparse: the mask rows of this workgroup's q tile, one per q head of the kv head (<= 64k tokens)
  // (round 3: ONE word per mask column, bit gq = q head gq of this kv head attends the column - a tile costs one
  // broadcast LDS read instead of G + 2 byte reads; 16 bits: the launcher accepts up to 16 q heads per kv head)
  __shared__ uint16_t s_mask[kSparse ? kMaskCols : 16];

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int n = lane & 15, g = lane >> 4;
  const int h = blockIdx.y, b = blockIdx.z;
  const int G = 1 << a.g_shift;
  const int q0 = as_const(a.cu_seqlens_q)[b];
  const int Sq = as_const(a.cu_seqlens_q)[b + 1] - q0;
  const int L = as_const(a.seqlens_kv)[b];
  // Row mapping of the workgroup's 128 (position, q head) rows.  G <= 8: every 16-row MFMA block is ONE
  // q head x 16 consecutive positions (block-sparse masks are per head: a whole block can then skip a
  // tile); G = 16: 8 positions x 16 heads, head fastest.
  constexpr int kWgRows = kWaves * kRowsPerWave;
  // q tiles are handed out last-first: a late tile walks the most KV (causal), so the longest workgroups start first
  // and the grid's tail is made of the short ones
  const int wg_pos0 = ((gridDim.x - 1 - blockIdx.x) * kWgRows) >> a.g_shift;  // positions per workgroup = 128 / G
  if (wg_pos0 >= Sq) return;  // whole workgroup past the request (uniform exit)
  const bool by_head = a.by_head != 0;
  const int blk_per_head = by_head ? (kWgRows >> a.g_shift) >> 4 : 0;  // 16-position blocks per head
  auto map_row = [&](int r, int& pos, int& hl) {  // r in [0, 128) -> position, local q head
    if (by_head) {
      const int blk = r >> 4;
      hl = blk / blk_per_head;
      pos = wg_pos0 + (blk % blk_per_head) * 16 + (r & 15);
    } else {
      hl = r & (G - 1);
      pos = wg_pos0 + (r >> a.g_shift);
    }
  };
  const int row0 = wave * kRowsPerWave;  // first row of this wave inside the workgroup
  int pos_first = 0x7fffffff;  // earliest position among this wave's rows (first row of each 16-row block)
#pragma unroll
  for (int nb = 0; nb < kNB; ++nb) {
    int p0, hl0;
    map_row(row0 + nb * 16, p0, hl0);
    pos_first = min(pos_first, p0);
  }
  const int past = L - Sq;  // cached tokens before the first q token
  // the workgroup walks the tiles its LAST row can see; a wave's own rows mask what they cannot
  const int wg_pos_last = min(Sq - 1, wg_pos0 + (kWgRows >> a.g_shift) - 1);
  const int num_seqkv = past + wg_pos_last + 1;
  const int ntile = (num_seqkv + 63) >> 6;
  const int ntile_full = max(past + pos_first + 1, 0) >> 6;  // tiles visible to every row of this wave
  const int page_mask = (1 << a.page_shift) - 1;
  const uint8_t* qbase = static_cast<const uint8_t*>(a.q);
  const uint8_t* kbase = static_cast<const uint8_t*>(a.kcache);
  const uint8_t* vbase = static_cast<const uint8_t*>(a.vcache);

  // ---- Q fragments (B operand of S^T = K Q^T): lane (n, g) holds 16-byte chunks g and g+4 of row n ----
  u32x4 qf[kNB][2];
  float row_scale[kNB];
  int row_lim[kNB];  // last visible key of this lane's q row (-1: row does not exist)
#pragma unroll
  for (int nb = 0; nb < kNB; ++nb) {
    int pos, hl;
    map_row(row0 + nb * 16 + n, pos, hl);
    const int hq = (h << a.g_shift) + hl;
    const bool ok = pos < Sq;
    row_lim[nb] = ok ? past + pos : -1;
    const long qoff = static_cast<long>(q0 + pos) * a.ldq + hq * 128;
#pragma unroll
    for (int c = 0; c < 2; ++c) {
      qf[nb][c] = u32x4{0u, 0u, 0u, 0u};
      if (ok) qf[nb][c] = ld16(qbase + qoff + (g + 4 * c) * 16);
    }
    float qs = ok ? a.qscale[(static_cast<long>(b) * a.num_head_q + hq) * a.qs_pad + pos] : 0.f;
    if constexpr (kQuant == 1) qs *= a.kscale[0];
    row_scale[nb] = a.scale_log2 * qs;
  }
  // O = (sum_j e4m3(256 p_j) v_j) / (sum_j 256 p_j) * vscale: the two 256 (row sum units, P units) cancel
  const float out_scale = kQuant == 1 ? a.vscale[0] : a.vscale[h];

  // ---- staging role: token block `wave` of each tile, lane -> (row lane/8 (+8), 16-byte chunk lane%8) ----
  const cint_ptr bid_row = as_const(a.block_ids) + static_cast<long>(b) * a.max_blocks;
  const int last_blk16 = (num_seqkv - 1) >> 4;
  const int st_chunk = lane & 7, st_rsub = lane >> 3;
  const int k_voff = st_rsub * static_cast<int>(a.k_token_stride) + st_chunk * 16;
  const int v_voff = st_rsub * static_cast<int>(a.v_token_stride) + st_chunk * 16;
  const int k_ld_bytes = 8 * static_cast<int>(a.k_token_stride), v_ld_bytes = 8 * static_cast<int>(a.v_token_stride);
  // Two register sets = the two tiles of the NEXT stage, fetched at the top of a stage and written to the other LDS
  // buffer at its end.  (Fetching two 64-token tiles ahead with one barrier per tile measured the same as one ahead: the
  // loop is not latency-bound.)
  const uint8_t* kbase_h = kbase + h * a.k_head_stride;
  const uint8_t* vbase_h = vbase + h * a.v_head_stride;
  u32x4 kst[2][2], vst[2][2];
  float ksst[2] = {0.f, 0.f};
  auto fetch = [&](int t, auto set_c) {  // unconditional: tiles past the end re-read the last block (never used)
    constexpr int kSet = decltype(set_c)::value;
    int blk = t * 4 + wave;
    blk = blk < last_blk16 ? blk : last_blk16;
    const int gtok = blk << 4;
    const int pid = __builtin_amdgcn_readfirstlane(bid_row[gtok >> a.page_shift]);
    const int inpage = gtok & page_mask;
    // unsigned 32 x 32 -> 64 products (the launcher refuses block strides of 4 GB and more): as signed 64-bit
    // arithmetic the four descriptor bases of a stage cost ~90 scalar instructions
    const uint32_t pidu = static_cast<uint32_t>(pid), inp = static_cast<uint32_t>(inpage);
    const i32x4 rk = srd_of(kbase_h + static_cast<uint64_t>(pidu) * static_cast<uint32_t>(a.k_block_stride) +
                            inp * static_cast<uint32_t>(a.k_token_stride));
    const i32x4 rv = srd_of(vbase_h + static_cast<uint64_t>(pidu) * static_cast<uint32_t>(a.v_block_stride) +
                            inp * static_cast<uint32_t>(a.v_token_stride));
    // Loads the compiler does not see (as in attention_decode_v2.hip): with compiler-visible loads hipcc drains the
    // whole queue - the set fetched at the top of this tile included - in front of every stash.  A fetch is kPer loads
    // in a fixed order; the stash of the older set waits with vmcnt(kPer).
    ld_pair(kst[kSet][0], kst[kSet][1], k_voff, rk, k_ld_bytes);
    ld_pair(vst[kSet][0], vst[kSet][1], v_voff, rv, v_ld_bytes);
    if constexpr (kQuant == 0) {
      // scales of tokens inpage .. inpage+15: tail row (tok >> 5), float (tok & 31); lanes 0..15 fetch one each
      const uint8_t* sp = reinterpret_cast<const uint8_t*>(a.kscale) + pid * a.ks_block_stride +
                          (inpage >> 5) * a.ks_row_stride + h * a.ks_head_stride + (inpage & 31) * 4;
      ld_one(ksst[kSet], (lane & 15) * 4, srd_of(sp));
    }
  };
  auto landed = [&](auto set_c) {  // everything fetched so far has landed (the sets are written to LDS together)
    constexpr int kSet = decltype(set_c)::value;
    wait_set<0>(kst[kSet], vst[kSet], ksst[kSet]);
  };
  auto stash = [&](auto set_c, int buf) {  // tile kSet of a stage -> rows kSet * 64 .. + 63 of LDS buffer buf
    constexpr int kSet = decltype(set_c)::value;
#pragma unroll
    for (int c = 0; c < 2; ++c) {
      const int row = kSet * 64 + wave * 16 + c * 8 + st_rsub;
      *reinterpret_cast<u32x4*>(&s_k[buf][row * kKRow + st_chunk * 16]) = kst[kSet][c];
      const int key = ((row >> 1) & 3) | (((row >> 4) & 1) << 2);
      *reinterpret_cast<u32x4*>(&s_v[buf][row * 128 + ((st_chunk ^ key) << 4)]) = vst[kSet][c];
    }
    if constexpr (kQuant == 0) {
      if (lane < 16) s_ks[buf][kSet * 64 + wave * 16 + lane] = ksst[kSet];
    }
  };

  // transposing V reads: this lane's row of the 8 x 16 tile and its (swizzled) byte address in buffer 0, ks 0, dims 0-15
  const int tr_j = (lane & 15) >> 1;
  const int tr_row = 16 * (tr_j >> 2) + 4 * g + (tr_j & 3);
  const int tr_key = ((tr_row >> 1) & 3) | (((tr_row >> 4) & 1) << 2);
  const uint32_t vt_base = static_cast<uint32_t>(reinterpret_cast<uintptr_t>((lds_u8*)&s_v[0][0])) + tr_row * 128 + (tr_key << 4) +
                           (lane & 1) * 8;
  f32x4 o[kNB][8];
  float m_run[kNB], l_run[kNB];
#pragma unroll
  for (int nb = 0; nb < kNB; ++nb) {
#pragma unroll
    for (int jj = 0; jj < 8; ++jj) o[nb][jj] = f32x4{0.f, 0.f, 0.f, 0.f};
    m_run[nb] = kNegInf;
    l_run[nb] = 0.f;
  }

  // block-sparse: a 64-token tile t belongs to mask column t / 2.  A row computes the tile only if its
  // (head, q tile) bit is set; the workgroup's rows all sit in one 128-position q tile and share the
  // G heads of this kv head, so "no row needs tile t" is the same in every wave: such tiles are neither
  // fetched nor computed.
  int row_hl[kNB];  // local q head of this lane's row in block nb
#pragma unroll
  for (int nb = 0; nb < kNB; ++nb) {
    int p_, hl_;
    map_row(row0 + nb * 16 + n, p_, hl_);
    row_hl[nb] = hl_;
  }
  if constexpr (kSparse) {
    const int mask_tm = min(wg_pos0 >> 7, a.mask_tiles_m - 1);
    const int cols = min(a.mask_tiles_kv, kMaskCols);
    for (int col = tid; col < cols; col += kThreads) {
      uint32_t bits = 0;
      for (int gq = 0; gq < G; ++gq)
        bits |= (a.block_mask[((static_cast<long>(b) * a.num_head_q + (h << a.g_shift) + gq) * a.mask_tiles_m + mask_tm) *
                                  a.mask_tiles_kv + col] != 0 ? 1u : 0u) << gq;
      s_mask[col] = static_cast<uint16_t>(bits);
    }
    __syncthreads();
  }
  auto tile_bits = [&](int t, bool (&bit)[kNB]) {
    if constexpr (!kSparse) {
#pragma unroll
      for (int nb = 0; nb < kNB; ++nb) bit[nb] = true;
      return true;
    } else {
      const int col = min(t, min(a.mask_tiles_kv, kMaskCols) - 1);  // t: stage = mask column
      const uint32_t bits = s_mask[col];
#pragma unroll
      for (int nb = 0; nb < kNB; ++nb) bit[nb] = ((bits >> row_hl[nb]) & 1u) != 0;
      // fetch / skip must be one decision for the whole workgroup: any of the G heads of this kv head
      return __builtin_amdgcn_readfirstlane(bits) != 0;
    }
  };
  bool scales_nonneg = true;
#pragma unroll
  for (int nb = 0; nb < kNB; ++nb) scales_nonneg &= __ballot(row_scale[nb] < 0.f) == 0;
  bool bit_cur[kNB];
  bool need_cur = true;
  fetch(0, IntC<0>{});
  fetch(1, IntC<1>{});  // (fetches are unconditional: past the end they re-read the last block)
  landed(IntC<0>{});
  landed(IntC<1>{});
  stash(IntC<0>{}, 0);
  stash(IntC<1>{}, 0);
  __syncthreads();
  // kFast (compile time): per-row scale only and nothing to mask - the softmax never forms s * rs.  As a run-time
  // condition inside one body hipcc if-converts the two paths: every tile then pays the general path's 16 multiplies,
  // compares and selects per 16-row block on top of the fast one (150 of the tile's 370 VALU instructions).
  // One body per STAGE (128 tokens = tiles t, t + 1): eight S^T blocks per 16-row block, ONE softmax update, and P V with
  // the K = 128 MFMA - lane (n, g) of the B operand holds its own 32 probabilities (byte 4 v + r of its 8 registers =
  // token 16 v + 4 g + r of the stage), lane (dim, g) of the A operand the same 32 tokens of V^T from four transposing
  // reads; the two sides agree on which token sits in which byte, which is all a dot product needs.  Half the matrix
  // pipe time of the K = 32 form for P V, 32 instead of 80 MFMA instructions per stage.  A second tile past the end
  // (odd tile count) holds a re-read of the last block: its tokens lie beyond every row's limit and the stage is never
  // a fast one.
  auto body = [&](int t, int buf, auto fast_c) {
    constexpr bool kFast = decltype(fast_c)::value != 0;
    const uint8_t* kt = s_k[buf];
    // O^T += V^T P^T: four transposing reads per 16 dims (lane (i, g): row j = i / 2 of read u is token
    // 32 u + 16 (j / 4) + 4 g + j % 4 = byte 8 u + j of the lane's 32 - the byte order of pf -, 8-byte half i % 2)
    auto vt_operand = [&](int jj) {
      v2i32 vtr[4];
#pragma unroll
      for (int u = 0; u < 4; ++u)
        vtr[u] = __builtin_amdgcn_ds_read_tr8_b64_v2i32(
            reinterpret_cast<lds_v2i32*>(static_cast<uint32_t>((vt_base ^ (jj << 4)) + buf * (128 * 128) + u * (32 * 128))));
      return i32x8{vtr[0][0], vtr[0][1], vtr[1][0], vtr[1][1], vtr[2][0], vtr[2][1], vtr[3][0], vtr[3][1]};
    };
    auto pv_block = [&](int nb, const i32x8& p) {
#pragma unroll
      for (int jj = 0; jj < 8; ++jj)
        o[nb][jj] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(vt_operand(jj), p, o[nb][jj], 0, 0, 0, 0, 0, 0);
    };
    bool nb_on[kNB];  // block-sparse: does this 16-row block (one head when G <= 8) attend the stage at all?
#pragma unroll
    for (int nb = 0; nb < kNB; ++nb) nb_on[nb] = !kSparse || __ballot(bit_cur[nb] && row_lim[nb] >= 0) != 0;
    if (!need_cur) return;

    // ---- S^T = K Q^T: the whole head dim in ONE v_mfma_f32_16x16x128_f8f6f4 per 16 x 16 block; lane (n, g) supplies
    // chunks g and g + 4 of its row on both sides ---------------------------------------------------------------
    f32x4 s[kNB][8];
#pragma unroll
    for (int tb = 0; tb < 8; ++tb) {
      u32x4 ka[2];
#pragma unroll
      for (int c = 0; c < 2; ++c)
        ka[c] = *reinterpret_cast<const u32x4*>(kt + (tb * 16 + n) * kKRow + (g + 4 * c) * 16);
      const i32x8 kv8 = {static_cast<int>(ka[0][0]), static_cast<int>(ka[0][1]), static_cast<int>(ka[0][2]),
                         static_cast<int>(ka[0][3]), static_cast<int>(ka[1][0]), static_cast<int>(ka[1][1]),
                         static_cast<int>(ka[1][2]), static_cast<int>(ka[1][3])};
#pragma unroll
      for (int nb = 0; nb < kNB; ++nb) {
        // (also for a block the mask switches off: its softmax is skipped and its P stays zero - a branch around
        // every MFMA costs the stages that ARE computed more than the idle MFMAs cost the ones that are not)
        const i32x8 qv8 = {static_cast<int>(qf[nb][0][0]), static_cast<int>(qf[nb][0][1]),
                           static_cast<int>(qf[nb][0][2]), static_cast<int>(qf[nb][0][3]),
                           static_cast<int>(qf[nb][1][0]), static_cast<int>(qf[nb][1][1]),
                           static_cast<int>(qf[nb][1][2]), static_cast<int>(qf[nb][1][3])};
        s[nb][tb] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(kv8, qv8, f32x4{0.f, 0.f, 0.f, 0.f}, 0, 0, 0, 0, 0, 0);
      }
    }

    // ---- online softmax, base 2.  p is produced as 256 p (the +8 rides in the exponent): it feeds the
    // e4m3 pack directly and the row sum is kept in the same units (undone once in the epilogue). -------
    i32x8 pf[kNB];
    bool all_bits = true;
#pragma unroll
    for (int nb = 0; nb < kNB; ++nb) all_bits &= bit_cur[nb] || row_lim[nb] < 0;
    // head-major blocks share head and q tile: sparsity switches whole blocks (nb_on), never single rows
    const bool masked = !kFast && (t + 1 >= ntile_full || (kSparse && !by_head && __ballot(!all_bits) != 0));
#pragma unroll
    for (int nb = 0; nb < kNB; ++nb) {
      pf[nb] = i32x8{0, 0, 0, 0, 0, 0, 0, 0};
      if (!nb_on[nb]) continue;  // nothing of this stage is visible to the block: m, l, O stay as they are
      const float rs = row_scale[nb];
      const int lim = bit_cur[nb] ? row_lim[nb] : -1;
      float mt = kNegInf;
      if constexpr (kFast) {
        // per-row scale only, nothing to mask: never form s * rs (a negative scale - not what a quantiser produces -
        // takes the general body: one test per wave, hoisted)
#pragma unroll
        for (int tb = 0; tb < 8; ++tb)
#pragma unroll
          for (int r = 0; r < 4; ++r) mt = fmaxf(mt, s[nb][tb][r]);
        mt *= rs;
      } else {
#pragma unroll
        for (int tb = 0; tb < 8; ++tb) {
          f32x4 kscl = f32x4{1.f, 1.f, 1.f, 1.f};
          if constexpr (kQuant == 0) kscl = *reinterpret_cast<const f32x4*>(&s_ks[buf][tb * 16 + g * 4]);
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            float x = s[nb][tb][r] * rs;
            if constexpr (kQuant == 0) x *= kscl[r];
            if (masked) {
              const int tok = t * 64 + tb * 16 + g * 4 + r;
              x = tok <= lim ? x : kNegInf;
            }
            s[nb][tb][r] = x;
            mt = fmaxf(mt, x);
          }
        }
      }
      mt = row4_max(mt);  // the 4 lanes of a q row, by row-swap instructions (no LDS round trip)
      const float m_new = fmaxf(m_run[nb], mt);
      const float m_use = m_new == kNegInf ? 0.f : m_new;
      const float bias = 8.0f - m_use;  // exp2(x - m + 8) = 256 p
      float psum = 0.f;
      const float mul = kFast ? rs : 1.0f;  // one form for both bodies: the general one left s * rs (masked) in s
#pragma unroll
      for (int tb = 0; tb < 8; ++tb) {
        float p[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          p[r] = __builtin_amdgcn_exp2f(fmaf(s[nb][tb][r], mul, bias));
          psum += p[r];
        }
        // p <= 2^8 < 448 by construction (x <= m): no clamp in front of the conversion
        int w = __builtin_amdgcn_cvt_pk_fp8_f32(p[0], p[1], 0, false);
        w = __builtin_amdgcn_cvt_pk_fp8_f32(p[2], p[3], w, true);
        pf[nb][tb] = w;
      }
      // rescale only when some row's maximum moved (past the first stages of a long row it rarely does)
      if (__builtin_amdgcn_ballot_w64(m_new != m_run[nb]) != 0) {
        const float alpha = __builtin_amdgcn_exp2f(m_run[nb] - m_use);
        l_run[nb] *= alpha;
#pragma unroll
        for (int jj = 0; jj < 8; ++jj) o[nb][jj] *= alpha;
        m_run[nb] = m_new;
      }
      l_run[nb] += psum;

      if constexpr (kSparse) pv_block(nb, pf[nb]);  // block-sparse: right here, and not at all for a block that is off
    }
    // dense: V^T operands shared by the two blocks (848 us against 888 us with one pass per block; the block-sparse
    // form is the other way round, 799 against 839 us at skip 0.5: half its blocks need no pass)
    if constexpr (!kSparse) {
#pragma unroll
      for (int jj = 0; jj < 8; ++jj) {
        const i32x8 va = vt_operand(jj);
#pragma unroll
        for (int nb = 0; nb < kNB; ++nb)
          o[nb][jj] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(va, pf[nb], o[nb][jj], 0, 0, 0, 0, 0, 0);
      }
    }
  };
  // tiles every row of this wave sees in full, scales of a real quantiser (>= 0), no per-row mask bits: the fast body
  const int n_fast = (kQuant == 1 && scales_nonneg && (!kSparse || by_head)) ? ntile_full : 0;
  auto stage = [&](int t, auto fast_c) {  // tiles t, t + 1
    const int buf = (t >> 1) & 1;
    fetch(t + 2, IntC<0>{});  // the next stage (both sets went to LDS at the end of the previous one)
    fetch(t + 3, IntC<1>{});
    need_cur = tile_bits(t >> 1, bit_cur);
    body(t, buf, fast_c);
    landed(IntC<0>{});
    landed(IntC<1>{});
    stash(IntC<0>{}, buf ^ 1);
    stash(IntC<1>{}, buf ^ 1);
    __syncthreads();
  };
  // two loops (one loop that picks the body per tile keeps four bodies live and spills): the stages whose two tiles
  // are both fast, then the rest on the general body
  const int t_fast = min(n_fast & ~1, ntile & ~1);
  for (int t = 0; t < t_fast; t += 2) stage(t, IntC<1>{});
  for (int t = t_fast; t < ntile; t += 2) stage(t, IntC<0>{});
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // the fetches past the end own their registers until they retire

  // ---- finish: row-major re-read through the wave's LDS tile, scale, bf16 store ---------------------------
#pragma unroll
  for (int nb = 0; nb < kNB; ++nb) {
    const float l = row4_sum(l_run[nb]);
    if (g == 0) s_l[wave][n] = l;
#pragma unroll
    for (int jj = 0; jj < 8; ++jj)  // o[nb][jj][r] = dim 16 jj + 4 g + r of q row n
      *reinterpret_cast<f32x4*>(&s_o[wave][n][jj * 16 + g * 4]) = o[nb][jj];
#pragma unroll 1
    for (int it = 0; it < 4; ++it) {
      const int row16 = it * 4 + (lane >> 4), c8 = lane & 15;
      int pos, hl;
      map_row(row0 + nb * 16 + row16, pos, hl);
      const f32x4 x0 = *reinterpret_cast<const f32x4*>(&s_o[wave][row16][c8 * 8]);
      const f32x4 x1 = *reinterpret_cast<const f32x4*>(&s_o[wave][row16][c8 * 8 + 4]);
      const float L2 = s_l[wave][row16];
      const float inv = (L2 > 0.f ? 1.0f / L2 : 0.f) * out_scale;
      if (pos < Sq) {
        u32x4 pk;
        pk[0] = pack_bf16x2(x0[0] * inv, x0[1] * inv);
        pk[1] = pack_bf16x2(x0[2] * inv, x0[3] * inv);
        pk[2] = pack_bf16x2(x1[0] * inv, x1[1] * inv);
        pk[3] = pack_bf16x2(x1[2] * inv, x1[3] * inv);
        st16(a.y + static_cast<long>(q0 + pos) * a.ldy + ((h << a.g_shift) + hl) * 128 + c8 * 8, pk);
      }
    }
  }
}

}  // namespace prefill
}  // namespace hpc

// reference: attention_with_kvcache_prefill_{qpertoken_perhead_kvpertensor,qkpertoken_perhead_vperhead}_fp8_async
// (src/attention/prefill/prefill.h; argument meaning kept; TMA scratch dropped).  quant_type 1 / 0 as in
// hpc_attention_decode_fp8_async; kscale strides (bytes) only for quant_type 0.
static int prefill_fp8_launch(const void* block_mask_ptr, int mask_tiles_m, int mask_tiles_kv,
    void* y_ptr, const void* q_ptr, const void* kcache_ptr, const void* vcache_ptr, const void* qscale_ptr,
    const void* kscale_ptr, const void* vscale_ptr, const void* cu_seqlens_q_ptr, const void* block_ids_ptr,
    const void* seqlens_kvcache_ptr, int quant_type, int num_batch, int max_seqlens_q, int max_seqlens_q_pad,
    int num_dim_qk, int num_dim_v, int num_head_q, int num_head_kv, int block_size, int num_seq_max_blocks,
    int ldY, int ldQ, int64_t kcache_block_stride, int64_t kcache_token_stride, int64_t kcache_head_stride,
    int64_t vcache_block_stride, int64_t vcache_token_stride, int64_t vcache_head_stride,
    int64_t kscale_block_stride_bytes, int64_t kscale_row_stride_bytes, int64_t kscale_head_stride_bytes,
    hipStream_t stream) {
  using namespace hpc::prefill;
  if (!y_ptr || !q_ptr || !kcache_ptr || !vcache_ptr || !qscale_ptr || !kscale_ptr || !vscale_ptr ||
      !cu_seqlens_q_ptr || !block_ids_ptr || !seqlens_kvcache_ptr)
    return HPC_ERR_INVALID;
  if (quant_type != 0 && quant_type != 1) return HPC_ERR_INVALID;
  if (num_batch <= 0 || max_seqlens_q <= 0) return num_batch < 0 || max_seqlens_q < 0 ? HPC_ERR_INVALID : HPC_OK;
  if (num_dim_qk != 128 || num_dim_v != 128) return HPC_ERR_UNSUPPORTED;
  if (block_size != 16 && block_size != 32 && block_size != 64) return HPC_ERR_UNSUPPORTED;
  if (quant_type == 0 && block_size < 32) return HPC_ERR_UNSUPPORTED;
  if (num_head_kv <= 0 || num_head_q % num_head_kv) return HPC_ERR_INVALID;
  const int group = num_head_q / num_head_kv;
  if (group != 1 && group != 2 && group != 4 && group != 8 && group != 16) return HPC_ERR_UNSUPPORTED;
  if ((ldQ & 15) || (ldY & 7) || (kcache_token_stride & 15) || (vcache_token_stride & 7) ||
      (kcache_head_stride & 15) || (vcache_head_stride & 7) || (kcache_block_stride & 15) ||
      (vcache_block_stride & 7))
    return HPC_ERR_UNSUPPORTED;
  Args a{};
  a.q = q_ptr;
  a.kcache = kcache_ptr;
  a.vcache = vcache_ptr;
  a.block_ids = static_cast<const int*>(block_ids_ptr);
  a.cu_seqlens_q = static_cast<const int*>(cu_seqlens_q_ptr);
  a.seqlens_kv = static_cast<const int*>(seqlens_kvcache_ptr);
  a.y = static_cast<uint16_t*>(y_ptr);
  a.qscale = static_cast<const float*>(qscale_ptr);
  a.kscale = static_cast<const float*>(kscale_ptr);
  a.vscale = static_cast<const float*>(vscale_ptr);
  a.block_mask = static_cast<const uint8_t*>(block_mask_ptr);
  // head-major 16-row blocks let a block skip masked-out tiles (block-sparse); dense attention measured ~8 %
  // faster with every wave holding all G heads of a few positions (key 7 overrides: 1 = by head, 2 = by position)
  a.by_head = group <= 8 && (hpc_dev_tuning_get(7) ? hpc_dev_tuning_get(7) == 1 : block_mask_ptr != nullptr);
  a.mask_tiles_m = mask_tiles_m;
  a.mask_tiles_kv = mask_tiles_kv;
  if (block_mask_ptr && (mask_tiles_m <= 0 || mask_tiles_kv <= 0)) return HPC_ERR_INVALID;
  if (block_mask_ptr && mask_tiles_kv > kMaskCols) return HPC_ERR_UNSUPPORTED;  // > 64k tokens of mask columns
  a.num_batch = num_batch;
  a.num_head_q = num_head_q;
  a.num_head_kv = num_head_kv;
  a.g_shift = group == 1 ? 0 : (group == 2 ? 1 : (group == 4 ? 2 : (group == 8 ? 3 : 4)));
  a.page_shift = block_size == 64 ? 6 : (block_size == 32 ? 5 : 4);
  a.max_blocks = num_seq_max_blocks;
  a.qs_pad = max_seqlens_q_pad;
  a.ldq = ldQ;
  a.ldy = ldY;
  a.k_block_stride = kcache_block_stride;
  a.k_token_stride = kcache_token_stride;
  a.k_head_stride = kcache_head_stride;
  a.v_block_stride = vcache_block_stride;
  a.v_token_stride = vcache_token_stride;
  a.v_head_stride = vcache_head_stride;
  a.ks_block_stride = kscale_block_stride_bytes;
  a.ks_row_stride = kscale_row_stride_bytes;
  a.ks_head_stride = kscale_head_stride_bytes;
  a.scale_log2 = 1.4426950408889634f / 11.313708498984761f;  // log2(e) / sqrt(128)
  // the kernel forms page offsets as unsigned 32 x 32 -> 64 products
  if (kcache_block_stride <= 0 || vcache_block_stride <= 0 || kcache_token_stride <= 0 || vcache_token_stride <= 0 ||
      kcache_block_stride >= (1ll << 32) || vcache_block_stride >= (1ll << 32) ||
      kcache_token_stride * block_size >= (1ll << 32) || vcache_token_stride * block_size >= (1ll << 32))
    return HPC_ERR_UNSUPPORTED;
  const long rows = static_cast<long>(max_seqlens_q) * group;
  dim3 grid(static_cast<unsigned>((rows + kWaves * kRowsPerWave - 1) / (kWaves * kRowsPerWave)), num_head_kv, num_batch);
  if (grid.z > 65535 || grid.y > 65535) return HPC_ERR_UNSUPPORTED;
  // default (temporal) cache policy: K/V tiles are re-read by every q tile of the request from L2
  if (block_mask_ptr) {
    if (quant_type == 1) prefill_fp8_kernel<1, true><<<grid, kThreads, 0, stream>>>(a);
    else prefill_fp8_kernel<0, true><<<grid, kThreads, 0, stream>>>(a);
  } else {
    if (quant_type == 1) prefill_fp8_kernel<1, false><<<grid, kThreads, 0, stream>>>(a);
    else prefill_fp8_kernel<0, false><<<grid, kThreads, 0, stream>>>(a);
  }
  HPC_CHECK_LAUNCH();
  return HPC_OK;
}

extern "C" int hpc_attention_with_kvcache_prefill_fp8_async(
    void* y_ptr, const void* q_ptr, const void* kcache_ptr, const void* vcache_ptr, const void* qscale_ptr,
    const void* kscale_ptr, const void* vscale_ptr, const void* cu_seqlens_q_ptr, const void* block_ids_ptr,
    const void* seqlens_kvcache_ptr, int quant_type, int num_batch, int max_seqlens_q, int max_seqlens_q_pad,
    int num_dim_qk, int num_dim_v, int num_head_q, int num_head_kv, int block_size, int num_seq_max_blocks,
    int ldY, int ldQ, int64_t kcache_block_stride, int64_t kcache_token_stride, int64_t kcache_head_stride,
    int64_t vcache_block_stride, int64_t vcache_token_stride, int64_t vcache_head_stride,
    int64_t kscale_block_stride_bytes, int64_t kscale_row_stride_bytes, int64_t kscale_head_stride_bytes,
    hipStream_t stream) {
  return prefill_fp8_launch(nullptr, 0, 0, y_ptr, q_ptr, kcache_ptr, vcache_ptr, qscale_ptr, kscale_ptr, vscale_ptr,
                            cu_seqlens_q_ptr, block_ids_ptr, seqlens_kvcache_ptr, quant_type, num_batch,
                            max_seqlens_q, max_seqlens_q_pad, num_dim_qk, num_dim_v, num_head_q, num_head_kv,
                            block_size, num_seq_max_blocks, ldY, ldQ, kcache_block_stride, kcache_token_stride,
                            kcache_head_stride, vcache_block_stride, vcache_token_stride, vcache_head_stride,
                            kscale_block_stride_bytes, kscale_row_stride_bytes, kscale_head_stride_bytes, stream);
}

// reference: attention_with_kvcache_blocksparse_prefill_fp8 (src/attention/entry.cc:264-409): the same
// kernel family with an optional uint8 block mask [B, Hq, ceil(max_seqlens_q / 128), mask_tiles_kv] over
// 128 (q positions) x 128 (kv tokens) tiles; null mask = dense.
extern "C" int hpc_attention_with_kvcache_blocksparse_prefill_fp8_async(
    void* y_ptr, const void* q_ptr, const void* kcache_ptr, const void* vcache_ptr, const void* qscale_ptr,
    const void* kscale_ptr, const void* vscale_ptr, const void* cu_seqlens_q_ptr, const void* block_ids_ptr,
    const void* seqlens_kvcache_ptr, const void* block_mask_ptr, int mask_tiles_m, int mask_tiles_kv,
    int quant_type, int num_batch, int max_seqlens_q, int max_seqlens_q_pad, int num_dim_qk, int num_dim_v,
    int num_head_q, int num_head_kv, int block_size, int num_seq_max_blocks, int ldY, int ldQ,
    int64_t kcache_block_stride, int64_t kcache_token_stride, int64_t kcache_head_stride,
    int64_t vcache_block_stride, int64_t vcache_token_stride, int64_t vcache_head_stride,
    int64_t kscale_block_stride_bytes, int64_t kscale_row_stride_bytes, int64_t kscale_head_stride_bytes,
    hipStream_t stream) {
  if (128 % block_size) return HPC_ERR_UNSUPPORTED;
  return prefill_fp8_launch(block_mask_ptr, mask_tiles_m, mask_tiles_kv, y_ptr, q_ptr, kcache_ptr, vcache_ptr,
                            qscale_ptr, kscale_ptr, vscale_ptr, cu_seqlens_q_ptr, block_ids_ptr, seqlens_kvcache_ptr,
                            quant_type, num_batch, max_seqlens_q, max_seqlens_q_pad, num_dim_qk, num_dim_v,
                            num_head_q, num_head_kv, block_size, num_seq_max_blocks, ldY, ldQ, kcache_block_stride,
                            kcache_token_stride, kcache_head_stride, vcache_block_stride, vcache_token_stride,
                            vcache_head_stride, kscale_block_stride_bytes, kscale_row_stride_bytes,
                            kscale_head_stride_bytes, stream);
}
