// Causal prefill attention in bf16: paged KV cache (attention_with_kvcache_prefill_bf16) and contiguous
// varlen K/V (attention_prefill_bf16) - gfx950.
//
// Replaces reference attention_prefill_bf16_async / attention_with_kvcache_prefill_bf16_async
// (src/attention/prefill/prefill.h, kernels src/attention/prefill/sm90/**, entry src/attention/entry.cc:15-150).
//
// Same structure as the FP8 prefill kernel (attention_prefill.hip): 64 KV tokens on the MFMA M axis, q rows
// on N (S^T = K Q^T, O^T = V^T P^T with v_mfma_f32_16x16x32_bf16), a wave owns 32 (position, q head) rows
// and finishes alone, a workgroup = 4 waves = 128 consecutive rows that walk the KV tiles together; every
// 64-token K/V tile (256-byte rows) is fetched once per workgroup - wave w loads token block w, full rows,
// one tile ahead - into a double-buffered LDS stage, one barrier per tile.  P is rounded to bf16 before
// P V like the decode kernel; V^T operands come out of LDS through the transposing read ds_read_b64_tr_b16
// (round 5; rounds 3-4 built them from 16-byte row pieces with v_perm_b32).  The
// epilogue reuses the staging LDS (aliased), so two workgroups fit a CU.
// Contiguous form: K/V rows of request b are rows cu_seqlens_q[b] .. of [total_seq, Hkv, 128] tensors
// (block_ids == null), every q token attends the keys up to itself.
#include "hpc_common.h"
#include "hpc_dev.h"
#include "../../include/hpc_amd.h"

namespace hpc {
namespace prefill16 {

struct Args {
  const void* q;
  const void* k;
  const void* v;
  const int* block_ids;  // null: contiguous K/V
  const int* cu_seqlens_q;
  const int* seqlens_kv;  // null with contiguous K/V (L = Sq)
  uint16_t* y;
  int num_batch, num_head_q, num_head_kv, g_shift, page_shift, max_blocks;
  int ldq, ldy;
  long k_block_stride, k_token_stride, k_head_stride;  // elements
  long v_block_stride, v_token_stride, v_head_stride;
  float scale_log2;
};

constexpr int kThreads = 256;
constexpr int kWaves = 4;
constexpr int kNB = 2;
constexpr int kRowsPerWave = 16 * kNB;
constexpr float kNegInf = -__builtin_inff();
constexpr int kKRow = 256 + 16;             // padded LDS row (bytes)
constexpr int kTileBytes = 64 * kKRow;      // one K or V tile
constexpr int kVTileBytes = 64 * 256;       // one V tile: unpadded 256-byte rows, 16-byte chunks XOR-swizzled (transposing reads)
constexpr int kStageBytes = 4 * kTileBytes;  // K, V x 2 buffers (the padded V form of rounds 3-4 is the larger one)
constexpr int kEpiBytes = kWaves * 16 * (128 + 4) * 4 + kWaves * 16 * 4;

// max / sum over the 4 lanes that share a q row (lane, lane ^ 16, lane ^ 32, lane ^ 48) with the gfx950 row swaps
// (no LDS round trip): permlane16_swap(x, x) = {[x0 x0 x2 x2], [x1 x1 x3 x3]}, permlane32_swap(y, y) = {[y0 y1 y0 y1], [y2 y3 y2 y3]}
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
template <int kN>
struct IntC {
  static constexpr int value = kN;
};

typedef short v4i16 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) v4i16 lds_v4i16;
typedef __attribute__((address_space(3))) uint8_t lds_u8;

union Frag16 {
  u32x4 u;
  bf16x8 b;
};

// kTr: V^T operands through the transposing LDS read (round 5); false: the v_perm_b32 form of rounds 3-4 (development key 46 = 1)
template <bool kTr>
__global__ __launch_bounds__(kThreads, 2) void prefill_bf16_kernel(const Args a) {
  constexpr int kVStride = kTr ? 256 : kKRow, kVBuf = kTr ? kVTileBytes : kTileBytes;
  __shared__ __attribute__((aligned(16))) uint8_t s_raw[kStageBytes > kEpiBytes ? kStageBytes : kEpiBytes];
  uint8_t* s_k = s_raw;                   // [2][64 * kKRow]
  uint8_t* s_v = s_raw + 2 * kTileBytes;  // [2][64 * 256], chunk c of token row t in slot c ^ (2 * (t & 7))

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int n = lane & 15, g = lane >> 4;
  const int h = blockIdx.y, b = blockIdx.z;
  const int G = 1 << a.g_shift;
  const int q0 = as_const(a.cu_seqlens_q)[b];
  const int Sq = as_const(a.cu_seqlens_q)[b + 1] - q0;
  const int L = a.seqlens_kv ? as_const(a.seqlens_kv)[b] : Sq;
  constexpr int kWgRows = kWaves * kRowsPerWave;
  // q tiles are handed out last-first: a late tile walks the most KV (causal), so the longest workgroups start first
  // and the grid's tail is made of the short ones
  const int wg_pos0 = ((gridDim.x - 1 - blockIdx.x) * kWgRows) >> a.g_shift;
  if (wg_pos0 >= Sq) return;
  const int row0 = wave * kRowsPerWave;  // rows of the workgroup: position-major, q head fastest
  const int pos_first = wg_pos0 + (row0 >> a.g_shift);
  const int past = L - Sq;
  const int wg_pos_last = min(Sq - 1, wg_pos0 + (kWgRows >> a.g_shift) - 1);
  const int num_seqkv = past + wg_pos_last + 1;
  const int ntile = (num_seqkv + 63) >> 6;
  const int ntile_full = max(past + pos_first + 1, 0) >> 6;
  const int page_mask = (1 << a.page_shift) - 1;
  const uint8_t* qbase = static_cast<const uint8_t*>(a.q);
  const uint8_t* kbase = static_cast<const uint8_t*>(a.k);
  const uint8_t* vbase = static_cast<const uint8_t*>(a.v);

  // ---- Q fragments: lane (n, g) holds dims (4j + g) * 8 .. + 7 of row n for k-step j ---------------------
  u32x4 qf[kNB][4];
  int row_lim[kNB];
#pragma unroll
  for (int nb = 0; nb < kNB; ++nb) {
    const int row = row0 + nb * 16 + n;
    const int pos = wg_pos0 + (row >> a.g_shift);
    const int hq = (h << a.g_shift) + (row & (G - 1));
    const bool ok = pos < Sq;
    row_lim[nb] = ok ? past + pos : -1;
    const long qoff = (static_cast<long>(q0 + pos) * a.ldq + hq * 128) * 2;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      qf[nb][j] = u32x4{0u, 0u, 0u, 0u};
      if (ok) qf[nb][j] = ld16(qbase + qoff + (4 * j + g) * 16);
    }
  }

  // ---- staging role: token block `wave` of each tile, lane -> (row lane/16 (+4), 16-byte chunk lane%16) --
  const int last_blk16 = (num_seqkv - 1) >> 4;
  const int st_chunk = lane & 15, st_rsub = lane >> 4;
  const int k_voff = (st_rsub * static_cast<int>(a.k_token_stride)) * 2 + st_chunk * 16;
  const int v_voff = (st_rsub * static_cast<int>(a.v_token_stride)) * 2 + st_chunk * 16;
  const int k_ld_bytes = 4 * static_cast<int>(a.k_token_stride) * 2, v_ld_bytes = 4 * static_cast<int>(a.v_token_stride) * 2;
  // K and V of the next tile are fetched and parked separately (K before Q K^T / stored after the
  // softmax, V after Q K^T / stored after P V): 16 staging registers live at a time instead of 32
  u32x4 kst[4], vst[4];
  auto tile_base = [&](int t, long& kb, long& vb, unsigned& k_lim, unsigned& v_lim) {
    int blk = t * 4 + wave;
    blk = blk < last_blk16 ? blk : last_blk16;
    const int gtok = blk << 4;
    if (a.block_ids) {
      const int pid = __builtin_amdgcn_readfirstlane(
          as_const(a.block_ids)[static_cast<long>(b) * a.max_blocks + (gtok >> a.page_shift)]);
      const int inpage = gtok & page_mask;
      // unsigned 32 x 32 -> 64 products (the launcher refuses larger strides): a quarter of the scalar instructions
      // of the signed 64-bit form
      const uint32_t pidu = static_cast<uint32_t>(pid), inp = static_cast<uint32_t>(inpage);
      kb = static_cast<long>(static_cast<uint64_t>(pidu) * static_cast<uint32_t>(a.k_block_stride) +
                             inp * static_cast<uint32_t>(a.k_token_stride));
      vb = static_cast<long>(static_cast<uint64_t>(pidu) * static_cast<uint32_t>(a.v_block_stride) +
                             inp * static_cast<uint32_t>(a.v_token_stride));
      k_lim = v_lim = 0xffffffffu;
    } else {
      // contiguous K/V: the last 16-token block of a request may run past its end - bound the read by
      // the rows that exist (reads beyond return zero); pages are always whole
      kb = static_cast<long>(q0 + gtok) * a.k_token_stride;
      vb = static_cast<long>(q0 + gtok) * a.v_token_stride;
      const unsigned rows = static_cast<unsigned>(min(16, L - gtok));
      k_lim = rows * static_cast<unsigned>(a.k_token_stride) * 2u;
      v_lim = rows * static_cast<unsigned>(a.v_token_stride) * 2u;
    }
  };
  // (the page lookup and the two row offsets of a tile are computed once, by fetch_k; fetch_v of the same tile - issued a
  //  section later - takes the V half from there: round 5, ~35 scalar instructions per tile less)
  long nx_vb = 0;
  unsigned nx_vl = 0;
  auto fetch_k = [&](int t) {
    long kb;
    unsigned kl;
    tile_base(t, kb, nx_vb, kl, nx_vl);
    const auto rk = make_rsrc(kbase + (kb + h * a.k_head_stride) * 2, kl);
#pragma unroll
    for (int c = 0; c < 4; ++c) kst[c] = buf_ld16<0>(rk, k_voff, c * k_ld_bytes);
  };
  auto fetch_v = [&]() {  // the tile fetch_k was last called for
    const auto rv = make_rsrc(vbase + (nx_vb + h * a.v_head_stride) * 2, nx_vl);
#pragma unroll
    for (int c = 0; c < 4; ++c) vst[c] = buf_ld16<0>(rv, v_voff, c * v_ld_bytes);
  };
  auto stash_k = [&](int buf) {
#pragma unroll
    for (int c = 0; c < 4; ++c)
      *reinterpret_cast<u32x4*>(s_k + buf * kTileBytes + (wave * 16 + c * 4 + st_rsub) * kKRow + st_chunk * 16) = kst[c];
  };
  // V stage: row t = 256 bytes, its 16-byte chunk c in slot c ^ (2 * (t & 7)): the 32 lanes of a transposing read (8 token rows
  // x 2 chunks x 2 halves) fall on 32 different 8-byte bank pairs, the 8 lanes of a 16-byte store group on 8 different slots
  const int vkey_e = (2 * st_rsub) * 16, vkey_o = (8 + 2 * st_rsub) * 16;  // rows c * 4 + st_rsub: (t & 7) = 4 (c & 1) + st_rsub
  auto stash_v = [&](int buf) {
#pragma unroll
    for (int c = 0; c < 4; ++c)
      *reinterpret_cast<u32x4*>(s_v + buf * kVBuf + (wave * 16 + c * 4 + st_rsub) * kVStride +
                                (kTr ? ((st_chunk * 16) ^ ((c & 1) ? vkey_o : vkey_e)) : st_chunk * 16)) = vst[c];
  };

  // transposing reads: lane -> token row 4g + j (j = (lane % 16) / 4) of a 16-token block, bytes 8 (lane % 4) .. + 7 of the 32 bytes
  // of a 16-dim block: chunk 2 u + (lane % 4) / 2 (swizzled: ^ 2 * (row & 7)), half lane % 2
  const int vtr_j = (lane & 15) >> 2, vtr_c = lane & 3;
  const int vtr_row = (4 * g + vtr_j) * 256 + (vtr_c >> 1) * 16 + (vtr_c & 1) * 8;
  const int vtr_key = (2 * ((4 * g + vtr_j) & 7)) * 16;
  f32x4 o[kNB][8];
  float m_run[kNB], l_run[kNB];
#pragma unroll
  for (int nb = 0; nb < kNB; ++nb) {
#pragma unroll
    for (int jj = 0; jj < 8; ++jj) o[nb][jj] = f32x4{0.f, 0.f, 0.f, 0.f};
    m_run[nb] = kNegInf;
    l_run[nb] = 0.f;
  }

  fetch_k(0);
  fetch_v();
  stash_k(0);
  stash_v(0);
  __syncthreads();
  // kMasked (compile time): tiles some row of the wave does not see in full.  As a run-time condition hipcc
  // if-converts the masking into every tile (round 3, as in attention_prefill.hip).
  auto tile = [&](int t, auto masked_c) {
    constexpr bool masked = decltype(masked_c)::value != 0;
    const int buf = t & 1;
    const bool more = t + 1 < ntile;
    if (more) fetch_k(t + 1);
    const uint8_t* kt = s_k + buf * kTileBytes;
    const uint8_t* vt = s_v + buf * kVBuf;

    // ---- S^T = K Q^T --------------------------------------------------------------------------------
    f32x4 s[kNB][4];
#pragma unroll
    for (int tb = 0; tb < 4; ++tb) {
      Frag16 ka[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) ka[j].u = *reinterpret_cast<const u32x4*>(kt + (tb * 16 + n) * kKRow + (4 * j + g) * 16);
#pragma unroll
      for (int nb = 0; nb < kNB; ++nb) {
        f32x4 acc = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          Frag16 qa;
          qa.u = qf[nb][j];
          acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ka[j].b, qa.b, acc, 0, 0, 0);
        }
        s[nb][tb] = acc;
      }
    }

    if (more) fetch_v();

    // ---- online softmax, base 2 ---------------------------------------------------------------------
    uint32_t pf[kNB][2][4];
#pragma unroll
    for (int nb = 0; nb < kNB; ++nb) {
      float mt = kNegInf;
#pragma unroll
      for (int tb = 0; tb < 4; ++tb)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          float x = s[nb][tb][r];
          if (masked) {
            const int tok = t * 64 + tb * 16 + g * 4 + r;
            x = tok <= row_lim[nb] ? x : kNegInf;
            s[nb][tb][r] = x;
          }
          mt = fmaxf(mt, x);
        }
      mt *= a.scale_log2;  // scale > 0: max commutes with it
      mt = row4_max(mt);
      const float m_new = fmaxf(m_run[nb], mt);
      const float m_use = m_new == kNegInf ? 0.f : m_new;
      float psum = 0.f;
#pragma unroll
      for (int tb = 0; tb < 4; ++tb) {
        float p[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          p[r] = __builtin_amdgcn_exp2f(fmaf(s[nb][tb][r], a.scale_log2, -m_use));
          psum += p[r];
        }
        pf[nb][tb >> 1][(tb & 1) * 2] = pack_bf16x2(p[0], p[1]);
        pf[nb][tb >> 1][(tb & 1) * 2 + 1] = pack_bf16x2(p[2], p[3]);
      }
      // rescale only when some row's maximum moved (past the first tiles of a long row it rarely does)
      if (__builtin_amdgcn_ballot_w64(m_new != m_run[nb]) != 0) {
        const float alpha = __builtin_amdgcn_exp2f(m_run[nb] - m_use);
        l_run[nb] *= alpha;
#pragma unroll
        for (int jj = 0; jj < 8; ++jj) o[nb][jj] *= alpha;
        m_run[nb] = m_new;
      }
      l_run[nb] += psum;
    }

    if (more) stash_k(buf ^ 1);

    // ---- O^T += V^T P^T: V^T operands straight out of LDS through ds_read_b64_tr_b16 (the gfx950 transposing read: lanes
    // 4j .. 4j+3 of a 16-lane group give the address of 8 bytes of row j of a 4 x 16 tile, lane i gets column i of the four rows).
    // Lane (i, g) of the A operand = dim 16 u + i of tokens 16 tb + 4g + 0..3 for tb = 2 ks, 2 ks + 1: the eight k-slots the B
    // operand (this lane group's probabilities, pf) holds.  Round 5: 32 transposing reads per tile instead of 16 row reads + 64
    // v_perm_b32.
    if constexpr (kTr) {
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) {
#pragma unroll
        for (int u = 0; u < 8; ++u) {
          Frag16 va;
#pragma unroll
          for (int hb = 0; hb < 2; ++hb) {
            const v4i16 tr = __builtin_amdgcn_ds_read_tr16_b64_v4i16(reinterpret_cast<lds_v4i16*>(
                static_cast<uint32_t>(reinterpret_cast<uintptr_t>((lds_u8*)(vt + (2 * ks + hb) * 16 * 256 + vtr_row + ((u * 32) ^ vtr_key))))));
            const u32x2 t2 = __builtin_bit_cast(u32x2, tr);
            va.u[2 * hb] = t2[0];
            va.u[2 * hb + 1] = t2[1];
          }
#pragma unroll
          for (int nb = 0; nb < kNB; ++nb) {
            Frag16 pa;
            pa.u = u32x4{pf[nb][ks][0], pf[nb][ks][1], pf[nb][ks][2], pf[nb][ks][3]};
            o[nb][u] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(va.b, pa.b, o[nb][u], 0, 0, 0);
          }
        }
      }
    } else {  // rounds 3-4: lane (n, g) reads dims 8n .. 8n+7 of tokens 16 tb + 4g + r; output block jj holds dims 8 i + jj
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) {
        u32x4 vf[8];
#pragma unroll
        for (int hb = 0; hb < 2; ++hb)
#pragma unroll
          for (int r = 0; r < 4; ++r)
            vf[hb * 4 + r] = *reinterpret_cast<const u32x4*>(vt + ((2 * ks + hb) * 16 + g * 4 + r) * kKRow + n * 16);
#pragma unroll
        for (int jj = 0; jj < 8; ++jj) {
          Frag16 va;
#pragma unroll
          for (int p2 = 0; p2 < 4; ++p2) {
            const uint32_t lo = vf[2 * p2][jj >> 1], hi = vf[2 * p2 + 1][jj >> 1];
            va.u[p2] = __builtin_amdgcn_perm(hi, lo, (jj & 1) ? 0x07060302u : 0x05040100u);
          }
#pragma unroll
          for (int nb = 0; nb < kNB; ++nb) {
            Frag16 pa;
            pa.u = u32x4{pf[nb][ks][0], pf[nb][ks][1], pf[nb][ks][2], pf[nb][ks][3]};
            o[nb][jj] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(va.b, pa.b, o[nb][jj], 0, 0, 0);
          }
        }
      }
    }
    if (more) stash_v(buf ^ 1);
    __syncthreads();
  };
  const int n_plain = min(max(ntile_full, 0), ntile);  // (two loops: one loop choosing per tile spills 556 bytes)
  for (int t = 0; t < n_plain; ++t) tile(t, IntC<0>{});
  for (int t = n_plain; t < ntile; ++t) tile(t, IntC<1>{});

  // ---- finish (the staging LDS is free now: reuse it for the row-major re-read) ------------------------------
  float (*s_o)[16][128 + 4] = reinterpret_cast<float (*)[16][128 + 4]>(s_raw);
  float (*s_l)[16] = reinterpret_cast<float (*)[16]>(s_raw + kWaves * 16 * (128 + 4) * 4);
#pragma unroll
  for (int nb = 0; nb < kNB; ++nb) {
    const float l = row4_sum(l_run[nb]);
    if (g == 0) s_l[wave][n] = l;
#pragma unroll
    for (int jj = 0; jj < 8; ++jj)
#pragma unroll
      for (int r = 0; r < 4; ++r) s_o[wave][n][kTr ? 16 * jj + g * 4 + r : 8 * (g * 4 + r) + jj] = o[nb][jj][r];
#pragma unroll 1
    for (int it = 0; it < 4; ++it) {
      const int row16 = it * 4 + (lane >> 4), c8 = lane & 15;
      const int row = row0 + nb * 16 + row16;
      const int pos = wg_pos0 + (row >> a.g_shift);
      const f32x4 x0 = *reinterpret_cast<const f32x4*>(&s_o[wave][row16][c8 * 8]);
      const f32x4 x1 = *reinterpret_cast<const f32x4*>(&s_o[wave][row16][c8 * 8 + 4]);
      const float L2 = s_l[wave][row16];
      const float inv = L2 > 0.f ? 1.0f / L2 : 0.f;
      if (pos < Sq) {
        u32x4 pk;
        pk[0] = pack_bf16x2(x0[0] * inv, x0[1] * inv);
        pk[1] = pack_bf16x2(x0[2] * inv, x0[3] * inv);
        pk[2] = pack_bf16x2(x1[0] * inv, x1[1] * inv);
        pk[3] = pack_bf16x2(x1[2] * inv, x1[3] * inv);
        st16(a.y + static_cast<long>(q0 + pos) * a.ldy + ((h << a.g_shift) + (row & (G - 1))) * 128 + c8 * 8, pk);
      }
    }
  }
}

}  // namespace prefill16
}  // namespace hpc

namespace {
int launch_prefill_bf16(hpc::prefill16::Args& a, int max_seqlens_q, int num_head_q, int num_head_kv, int num_batch,
                        hipStream_t stream) {
  using namespace hpc::prefill16;
  if (num_head_kv <= 0 || num_head_q % num_head_kv) return HPC_ERR_INVALID;
  const int group = num_head_q / num_head_kv;
  if (group != 1 && group != 2 && group != 4 && group != 8 && group != 16) return HPC_ERR_UNSUPPORTED;
  if ((a.ldq & 7) || (a.ldy & 7) || (a.k_token_stride & 7) || (a.v_token_stride & 7) || (a.k_head_stride & 7) ||
      (a.v_head_stride & 7) || (a.k_block_stride & 7) || (a.v_block_stride & 7))
    return HPC_ERR_UNSUPPORTED;  // 16-byte vector accesses
  a.num_batch = num_batch;
  a.num_head_q = num_head_q;
  a.num_head_kv = num_head_kv;
  a.g_shift = group == 1 ? 0 : (group == 2 ? 1 : (group == 4 ? 2 : (group == 8 ? 3 : 4)));
  a.scale_log2 = 1.4426950408889634f / 11.313708498984761f;  // log2(e) / sqrt(128)
  const long rows = static_cast<long>(max_seqlens_q) * group;
  dim3 grid(static_cast<unsigned>((rows + kWaves * kRowsPerWave - 1) / (kWaves * kRowsPerWave)), num_head_kv, num_batch);
  if (grid.z > 65535 || grid.y > 65535) return HPC_ERR_UNSUPPORTED;
  if (kHpcDevBuild && hpc_dev_tuning_get(46) == 1)  // development key 46 = 1: V^T operands built with v_perm_b32 (rounds 3-4)
    prefill_bf16_kernel<false><<<grid, kThreads, 0, stream>>>(a);
  else
    prefill_bf16_kernel<true><<<grid, kThreads, 0, stream>>>(a);
  HPC_CHECK_LAUNCH();
  return HPC_OK;
}
}  // namespace

// reference: attention_with_kvcache_prefill_bf16_async (src/attention/prefill/prefill.h; entry
// src/attention/entry.cc:83-150).  seqlens_kvcache = cached tokens INCLUDING the q tokens (the reference
// tests' model: q row s attends keys j <= L - Sq + s).
extern "C" int hpc_attention_with_kvcache_prefill_bf16_async(
    void* y_ptr, const void* q_ptr, const void* kcache_ptr, const void* vcache_ptr, const void* cu_seqlens_q_ptr,
    const void* block_ids_ptr, const void* seqlens_kvcache_ptr, int num_batch, int max_seqlens_q, int num_dim_qk,
    int num_dim_v, int num_head_q, int num_head_kv, int block_size, int num_seq_max_blocks, int ldY, int ldQ,
    int64_t kcache_block_stride, int64_t kcache_token_stride, int64_t kcache_head_stride,
    int64_t vcache_block_stride, int64_t vcache_token_stride, int64_t vcache_head_stride, hipStream_t stream) {
  if (!y_ptr || !q_ptr || !kcache_ptr || !vcache_ptr || !cu_seqlens_q_ptr || !block_ids_ptr || !seqlens_kvcache_ptr)
    return HPC_ERR_INVALID;
  if (num_batch <= 0 || max_seqlens_q <= 0) return num_batch < 0 || max_seqlens_q < 0 ? HPC_ERR_INVALID : HPC_OK;
  if (num_dim_qk != 128 || num_dim_v != 128) return HPC_ERR_UNSUPPORTED;
  if (block_size != 16 && block_size != 32 && block_size != 64) return HPC_ERR_UNSUPPORTED;
  hpc::prefill16::Args a{};
  a.q = q_ptr;
  a.k = kcache_ptr;
  a.v = vcache_ptr;
  a.block_ids = static_cast<const int*>(block_ids_ptr);
  a.cu_seqlens_q = static_cast<const int*>(cu_seqlens_q_ptr);
  a.seqlens_kv = static_cast<const int*>(seqlens_kvcache_ptr);
  a.y = static_cast<uint16_t*>(y_ptr);
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
  // the kernel forms page offsets (in elements) as unsigned 32 x 32 -> 64 products
  if (kcache_block_stride <= 0 || vcache_block_stride <= 0 || kcache_token_stride <= 0 || vcache_token_stride <= 0 ||
      kcache_block_stride >= (1ll << 31) || vcache_block_stride >= (1ll << 31) ||
      kcache_token_stride * block_size >= (1ll << 31) || vcache_token_stride * block_size >= (1ll << 31))
    return HPC_ERR_UNSUPPORTED;
  return launch_prefill_bf16(a, max_seqlens_q, num_head_q, num_head_kv, num_batch, stream);
}

// reference: attention_prefill_bf16_async (src/attention/prefill/prefill.h; entry src/attention/entry.cc:15-81):
// q [total, Hq, 128], k / v [total, Hkv, 128] (row strides ldK / ldV elements), causal inside each request.
extern "C" int hpc_attention_prefill_bf16_async(void* y_ptr, const void* q_ptr, const void* k_ptr, const void* v_ptr,
                                                const void* cu_seqlens_q_ptr, int num_batch, int max_seqlens_q,
                                                int num_dim_qk, int num_dim_v, int num_head_q, int num_head_kv,
                                                int ldY, int ldQ, int ldK, int ldV, hipStream_t stream) {
  if (!y_ptr || !q_ptr || !k_ptr || !v_ptr || !cu_seqlens_q_ptr) return HPC_ERR_INVALID;
  if (num_batch <= 0 || max_seqlens_q <= 0) return num_batch < 0 || max_seqlens_q < 0 ? HPC_ERR_INVALID : HPC_OK;
  if (num_dim_qk != 128 || num_dim_v != 128) return HPC_ERR_UNSUPPORTED;
  hpc::prefill16::Args a{};
  a.q = q_ptr;
  a.k = k_ptr;
  a.v = v_ptr;
  a.block_ids = nullptr;
  a.cu_seqlens_q = static_cast<const int*>(cu_seqlens_q_ptr);
  a.seqlens_kv = nullptr;
  a.y = static_cast<uint16_t*>(y_ptr);
  a.page_shift = 6;
  a.max_blocks = 0;
  a.ldq = ldQ;
  a.ldy = ldY;
  a.k_block_stride = 0;
  a.k_token_stride = ldK;
  a.k_head_stride = 128;
  a.v_block_stride = 0;
  a.v_token_stride = ldV;
  a.v_head_stride = 128;
  return launch_prefill_bf16(a, max_seqlens_q, num_head_q, num_head_kv, num_batch, stream);
}
