// FP8 paged decode attention for NHD pages, second generation: a wave fetches the rows of TWO adjacent kv
// heads with one instruction, stages a whole wave-iteration through LDS (where the hardware transposes V), keeps
// the next one in flight in registers and plans its own work in closed form (the dynamic tile schedule of
// assign_task.hip restated on the head-PAIR axis).
//
// Measurements that shaped it (tools/probes/probe_pair.hip, probe_tr; rocprofv3 SQ counters, tools/pmc_decode.py):
//  * on NHD pages [page][token][head][128 B] a kv head's fp8 row is 128 bytes at a 1 KB stride.  A wave that asks
//    for 128-byte pieces streams at 0.75 of 8 TB/s in a pure read probe (0.58-0.66 in the first-generation kernel,
//    attention_decode.hip, which stays for HND pages, bf16, per-token K scales and odd head counts); asking for the
//    256 bytes of a head pair IN ONE INSTRUCTION streams at 0.82-0.84 (512 bytes: 0.88).
//  * the chip needs >= 8 waves per CU issuing loads: 4 waves per CU with 32 KB each in flight top out at 0.75.
//  * a wave issues at most one instruction every 4 cycles: a first attempt with the loads held in registers
//    (v_permlane16_swap to separate the heads, private v_perm_b32 byte transposes for V^T, ~800 instructions per
//    16 KB, 400 registers -> one wave per SIMD) spent 52 % of its cycles issuing and 24 % on dependency stalls.
// So: ~350 instructions per 16 KB, <= 256 registers (two waves per SIMD), 16 KB per wave in flight.
//
// Structure (replaces, for this case, reference src/attention/decode/sm90/dynamic/smallm_fp8_*_dim128_*.cu(h)):
//  * wave-iteration ("WI") = 32 tokens x 2 heads = 8 KB of K + 8 KB of V: 16 x buffer_load_dwordx4 whose lanes
//    cover 4 token rows x 256 B (+ 6 small loads for Q / q scales, real only for a wave's first WI of a task).
//    The loads are inline asm with hand-counted vmcnt (hipcc drained the queue once per WI).
//  * at the top of a WI the landed registers are written to the wave's private 16 KB LDS stage
//    ([token][256 B], 16-byte chunks XOR-swizzled with the token index) and the registers are immediately
//    re-used for the NEXT WI's loads - across task boundaries too - so 16 KB per wave stay in flight while the
//    MFMAs run.  K leaves LDS as MFMA A operands with ds_read_b128 (one v_mfma_f32_16x16x128_f8f6f4 per
//    16-token block and head); V leaves it through ds_read_b64_tr_b8, the gfx950 transpose read: every lane gives
//    the address of 8 bytes of one token row, the 16 lanes of a group get back one COLUMN (dim) of the 8 x 16
//    tile each = the V^T operand of O^T += V^T P^T with no VALU work at all.
//  * the plan is computed by every wave in SGPRs from the request lengths: requests laid end to end on a tile
//    axis per head pair, range r owns tiles [r * per, (r + 1) * per) (workgroup = range x pair), found with one ballot over a wave-wide
//    prefix sum - the closed form of csrc/assign_task.hip (reference assign_task.cu:362-492) on a different
//    axis, so the caller's task map is not needed here (the first-generation kernel consumes it).
//  * the 4 waves of a workgroup share a task (WIs w, w+4, ...), merge through LDS once per task (the idle
//    stage regions double as the merge buffer) and write bf16 y.  A request cut by a range boundary leaves
//    an fp32 partial + base-2 LSE per chunk (write-through stores) and takes a ticket on the request's arrival
//    counter; the chunk that arrives LAST merges all of them in the same launch (the reference's static path
//    does the same, static_splitk_kernels.cuh:362-377) - no second kernel, no launch boundary.  The counters
//    live in the call's scratch, are tagged with a per-launch epoch (stale contents read as zero arrivals) and
//    are left zero.  Development key 17 = 2 keeps the merge in a second kernel (measured equal: 148.6 vs 148.1 us
//    on the C3 mix, same box).  Two workgroups per CU.
//  * fp8 numerics as in the first generation / the reference kernels.
#include <atomic>
#include <type_traits>
#include <utility>

#include "hpc_common.h"
#include "hpc_dev.h"
#include "../../include/hpc_amd.h"
#include "attention_decode_v2.h"

namespace hpc {
namespace decode2_old {
using hpc::decode2::Args;

constexpr int kThreads = 256;
constexpr int kWaves = 4;
constexpr int kHP = 2;          // heads per workgroup pass
constexpr int kTok = 32;        // tokens per wave-iteration
constexpr float kNegInf = -__builtin_inff();
constexpr int kRow = kHP * 128;                       // LDS row of a token: head 0 | head 1
constexpr int kVOff = kTok * kRow;                    // V half of a wave's stage
constexpr int kWaveLds = 2 * kTok * kRow;             // 16 KB: K + V of one WI; the merge buffer s_o[head][16][128] aliases it
typedef int v2i32 __attribute__((ext_vector_type(2)));
typedef __attribute__((address_space(3))) v2i32 lds_v2i32;

__device__ __forceinline__ long pack64(uint32_t lo, uint32_t hi) {
  return static_cast<long>(static_cast<uint64_t>(lo) | (static_cast<uint64_t>(hi) << 32));
}
__device__ __forceinline__ int sgpr(int v) { return __builtin_amdgcn_readfirstlane(v); }

// compile-time loop over ring slots: f(std::integral_constant<int, 0>{}), f(<1>), ...
template <typename F, int... S>
__device__ __forceinline__ void static_for_impl(F& f, std::integer_sequence<int, S...>) {
  (f(std::integral_constant<int, S>{}), ...);
}
template <int N, typename F>
__device__ __forceinline__ void static_for(F&& f) {
  static_for_impl(f, std::make_integer_sequence<int, N>{});
}

// One wave-iteration as the load side planned it; everything here is wave-uniform (SGPRs).
struct Stage {
  int flags;      // bit 0: valid (inside this workgroup's range), 1: this wave's last WI of the task, 2: its first
  int bp;         // request | head pair << 16
  int tok0;       // first token of the WI inside the request
  int tok_end;    // end of the task's token range (<= total tokens of the request)
  int ltot;       // total tokens of the request (incl. the num_seq_q new ones)
  int req0;       // position of the request's first tile on the pair axis (chunk bookkeeping)
  int tiles;      // tiles of the request
};

// ---- loads the compiler does not see ---------------------------------------------------------------------------
// hipcc's waitcnt insertion drained the whole queue once per wave-iteration in every compiler-visible form of this
// pipeline (vmcnt(0) at the V wait whatever the prefetch depth), so the K / V / Q loads are inline asm and their
// completion is counted by hand: loads retire in order, a WI's loads are issued in the fixed order
// Q (6) | K block 0 (4) | K block 1 (4) | V block 0 (4) | V block 1 (4), and a consumer waits with
// vmcnt(number of loads issued after the ones it needs).  Stores and other compiler-visible memory operations in
// the queue only make these waits conservative, never unsafe.  Destinations are "v" registers: the kernel stays
// well below 256 registers, so hipcc has no reason to shuffle a not-yet-landed destination through an AGPR (it did
// at 400 registers; tools: audit the .s for compiler instructions touching a destination between load and wait).
typedef int i32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ i32x4 srd(const void* base, unsigned num_records) {
  const uint64_t v = reinterpret_cast<uint64_t>(base);
  return i32x4{sgpr(static_cast<int>(v)), sgpr(static_cast<int>(v >> 32)), sgpr(static_cast<int>(num_records)), 0x00020000};
}
// every statement opens with s_nop 4: its SGPR operands may have been written by v_readfirstlane just before
// (the descriptor is re-pinned word by word: an "s" operand must be provably wave-uniform or hipcc hands the
// assembler a VGPR tuple)
__device__ __forceinline__ i32x4 pin(i32x4 r) { return i32x4{sgpr(r[0]), sgpr(r[1]), sgpr(r[2]), sgpr(r[3])}; }
template <int kAux>
__device__ __forceinline__ void ld_x4x4(u32x4 (&k)[4], int voff, i32x4 rs_in, int s1, int s2, int s3) {
  const i32x4 rs = pin(rs_in);
  if constexpr (kAux == 2)
    asm volatile("s_nop 4\n\tbuffer_load_dwordx4 %0, %4, %5, 0 offen nt\n\tbuffer_load_dwordx4 %1, %4, %5, %6 offen nt\n\t"
                 "buffer_load_dwordx4 %2, %4, %5, %7 offen nt\n\tbuffer_load_dwordx4 %3, %4, %5, %8 offen nt"
                 : "=&v"(k[0]), "=&v"(k[1]), "=&v"(k[2]), "=&v"(k[3])
                 : "v"(voff), "s"(rs), "s"(s1), "s"(s2), "s"(s3));
  else
    asm volatile("s_nop 4\n\tbuffer_load_dwordx4 %0, %4, %5, 0 offen\n\tbuffer_load_dwordx4 %1, %4, %5, %6 offen\n\t"
                 "buffer_load_dwordx4 %2, %4, %5, %7 offen\n\tbuffer_load_dwordx4 %3, %4, %5, %8 offen"
                 : "=&v"(k[0]), "=&v"(k[1]), "=&v"(k[2]), "=&v"(k[3])
                 : "v"(voff), "s"(rs), "s"(s1), "s"(s2), "s"(s3));
}
// Q fragments (2 x 16 B, chunks g and g + 4 of the row) + the row's q scale, for one head
__device__ __forceinline__ void ld_q3(u32x4 (&q)[2], uint32_t& sc, int voff_q, i32x4 rq_in, int voff_s, i32x4 rsc_in) {
  const i32x4 rq = pin(rq_in), rsc = pin(rsc_in);
  asm volatile("s_nop 4\n\tbuffer_load_dwordx4 %0, %3, %4, 0 offen\n\tbuffer_load_dwordx4 %1, %3, %4, 0 offen offset:64\n\t"
               "buffer_load_dword %2, %5, %6, 0 offen"
               : "=&v"(q[0]), "=&v"(q[1]), "=&v"(sc)
               : "v"(voff_q), "s"(rq), "v"(voff_s), "s"(rsc));
}
__device__ __forceinline__ void ld_x4(u32x4& d, int voff, i32x4 rs) {
  asm volatile("s_nop 4\n\tbuffer_load_dwordx4 %0, %1, %2, 0 offen" : "=&v"(d) : "v"(voff), "s"(rs));
}
// "wait until at most N loads are outstanding", tied to the registers it makes valid
template <int N>
__device__ __forceinline__ void wait_x4x4(u32x4 (&k)[4]) {
  asm volatile("s_waitcnt vmcnt(%4)" : "+v"(k[0]), "+v"(k[1]), "+v"(k[2]), "+v"(k[3]) : "n"(N < 63 ? N : 63));
}
template <int N>
__device__ __forceinline__ void wait_q(u32x4 (&q0)[2], u32x4 (&q1)[2], uint32_t& s0, uint32_t& s1) {
  asm volatile("s_waitcnt vmcnt(%6)" : "+v"(q0[0]), "+v"(q0[1]), "+v"(q1[0]), "+v"(q1[1]), "+v"(s0), "+v"(s1) : "n"(N < 63 ? N : 63));
}
template <int N>
__device__ __forceinline__ void wait_x4(u32x4& x0, u32x4& x1, u32x4& x2, u32x4& x3) {
  asm volatile("s_waitcnt vmcnt(%4)" : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3) : "n"(N < 63 ? N : 63));
}

typedef __attribute__((address_space(3))) u32x4 lds_u32x4;
typedef __attribute__((address_space(3))) uint8_t lds_u8;
// max over the 4 lanes that share a q row (lane, lane ^ 16, lane ^ 32, lane ^ 48) with the gfx950 row-swap
// instructions instead of ds_bpermute (an LDS round trip on the critical path of every WI):
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
typedef const float __attribute__((address_space(4))) * cfloat_ptr;
__device__ __forceinline__ cfloat_ptr as_constf(const float* p) { return (cfloat_ptr)(reinterpret_cast<uintptr_t>(p)); }

template <int kAux>
__global__ __launch_bounds__(kThreads, 2) void decode2_kernel(const Args a) {
  __shared__ __attribute__((aligned(1024))) uint8_t s_wave[kWaves][kWaveLds];  // stage addresses are (base) ^ (bits 4-7)
  __shared__ float s_m[kHP][kWaves][16];
  __shared__ float s_l[kHP][kWaves][16];
  __shared__ int s_ticket;
  // loads of one WI, in issue order: Q (6) | K (8) | V (8)
  constexpr int kNV = 8;

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = sgpr(tid >> 6);
  const int n = lane & 15;  // MFMA N index: q row
  const int g = lane >> 4;  // MFMA k-slot group / C row group
  const int wg = blockIdx.x, nwg = gridDim.x;
  const int B = a.num_batch, Sq = a.num_seq_q;
  const int G = 1 << a.g_shift;
  const int rows_valid = Sq << a.g_shift;
  const int page_mask = (1 << a.page_shift) - 1;
  const cint_ptr lens = as_const(a.lens);
  const int add_new = a.new_kv_included ? 0 : Sq;
  const uint8_t* qbase = static_cast<const uint8_t*>(a.q);
  const uint8_t* kbase = static_cast<const uint8_t*>(a.kcache);
  const uint8_t* vbase = static_cast<const uint8_t*>(a.vcache);
  const auto part_rs = make_rsrc(a.part_o);
  const auto lse_rs = make_rsrc(a.part_lse);
  auto ltot_of = [&](int b) __attribute__((always_inline)) { const int l = lens[b] + add_new; return l > 0 ? l : 0; };
  auto tiles_of = [&](int b) __attribute__((always_inline)) { return (ltot_of(b) + 63) >> 6; };

  // ---- plan: this workgroup's range, its first (request, tile) ------------------------------------------------
  // Requests are laid end to end on a COST axis: tiles_b tiles of 64 tokens preceded by kOvh units of overhead
  // per non-empty request (a task boundary costs a merge through LDS, two barriers, idle pipeline slots and up to
  // three phantom WIs: about kOvh tiles' worth of time), so a range that holds many short requests gets fewer
  // tiles.  Range r owns the cost positions [r * per, (r + 1) * per); tile t of request b sits at position
  // cost_start(b) + kOvh + t.  lane l sums the costs of requests [l * kpl, (l + 1) * kpl); an inclusive wave scan
  // gives the prefix at every chunk start; the request that holds a position is found with one ballot + a walk
  // over at most kpl requests.
  constexpr int kOvh = 2;  // measured: 1..8 within noise on the mixed workload, 0 (no balancing) 3 % slower
  auto cost_of = [&](int b) __attribute__((always_inline)) { const int t = tiles_of(b); return t > 0 ? t + kOvh : 0; };
  const int kpl = (B + 63) >> 6;
  int chunk_sum = 0;
  for (int i = 0; i < kpl; ++i) {
    const int b = lane * kpl + i;
    if (b < B) {
      const int l = a.lens[b] + add_new;
      const int t = ((l > 0 ? l : 0) + 63) >> 6;
      chunk_sum += t > 0 ? t + kOvh : 0;
    }
  }
  int incl = chunk_sum;
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) {
    const int t = __shfl_up(incl, o, 64);
    if (lane >= o) incl += t;
  }
  const int chunk_start = incl - chunk_sum;
  const int Th = sgpr(__shfl(incl, 63, 64));       // total cost per head pair
  const int npair = a.num_head_kv / kHP;
  if (Th == 0) return;
  // Workgroup -> (range r, pair p) with the pair index MINOR: workgroups r * npair .. r * npair + npair - 1 stream
  // the npair 256-byte slices of the same token rows at about the same time (identical work, started together).
  // The grid is a multiple of npair (launcher).
  const int nrange = nwg / npair;
  const int rng = wg / npair;
  // a range is never smaller than min_range_cost: a batch with little work (one long request among a few short
  // ones) runs on fewer workgroups instead of being cut into one-tile chunks that the last arriver of the long
  // request has to merge one by one (15 x 64 + 1 x 16k tokens: 87-99 us with 128 ranges per pair, 47 us with a floor of 8)
  const int per_even = (Th + nrange - 1) / nrange;
  const int per = per_even > a.min_range_cost ? per_even : a.min_range_cost;
  const long g_begin = static_cast<long>(rng) * per;
  const long g_end = g_begin + per < Th ? g_begin + per : Th;
  if (g_begin >= g_end) return;

  // ---- load cursor (SGPRs) ---------------------------------------------------------------------------
  // The range is converted to real tiles once: it starts at tile c_rt0 of request c_b and ends in front of tile
  // e_rt of request e_b (a boundary inside a request's overhead units is a boundary at that request's first tile).
  auto locate = [&](int x, int& b_out, int& rt_out, int& cc_out) __attribute__((always_inline)) {
    if (x >= Th) {
      b_out = B;
      rt_out = 0;
      cc_out = Th;
      return;
    }
    const uint64_t le = __ballot(chunk_start <= x);
    const int cl = 63 - __builtin_clzll(le);  // last lane whose chunk starts at or before x
    int pos = sgpr(__shfl(chunk_start, cl, 64));
    int b = cl * kpl;
    while (true) {  // position x lies in this chunk (empty requests cost nothing and are stepped over)
      const int t = cost_of(b);
      if (x < pos + t) break;
      pos += t;
      ++b;
    }
    b_out = b;
    rt_out = x - pos > kOvh ? x - pos - kOvh : 0;
    cc_out = pos;
  };
  int c_p = wg % npair, c_b, c_rt0, c_cc, e_b, e_rt, e_cc;
  locate(static_cast<int>(g_begin), c_b, c_rt0, c_cc);
  locate(static_cast<int>(g_end), e_b, e_rt, e_cc);
  if (c_b == e_b && c_rt0 >= e_rt) return;  // the whole range lies inside one request's overhead units
  int c_wi, c_nwi, c_ltot, c_tiles, c_tok_end, c_valid = 1;
  auto open_task = [&]() __attribute__((always_inline)) {  // c_b, c_rt0, c_cc set: derive the rest
    c_ltot = ltot_of(c_b);
    c_tiles = (c_ltot + 63) >> 6;
    const int end_t = c_b == e_b ? e_rt : c_tiles;
    c_tok_end = end_t * 64 < c_ltot ? end_t * 64 : c_ltot;
    c_nwi = (c_tok_end - c_rt0 * 64 + kTok - 1) / kTok;
    c_wi = wave;
  };
  open_task();
  auto advance = [&]() __attribute__((always_inline)) {  // to this wave's next WI (every wave visits every task at least once)
    c_wi += kWaves;
    const int lim = c_nwi > wave + 1 ? c_nwi : wave + 1;
    if (c_wi < lim) return;
    if (c_b == e_b) {  // that was the range's last task
      c_valid = 0;
      return;
    }
    c_cc += c_tiles + kOvh;
    do {
      ++c_b;
    } while (c_b < e_b && tiles_of(c_b) == 0);
    if (c_b == e_b && e_rt == 0) {
      c_valid = 0;
      return;
    }
    c_rt0 = 0;
    open_task();
  };
  auto snapshot = [&](Stage& st) __attribute__((always_inline)) {
    const int lim = c_nwi > wave + 1 ? c_nwi : wave + 1;
    st.flags = c_valid | ((c_wi + kWaves >= lim) ? 2 : 0) | (c_wi == wave ? 4 : 0);
    st.bp = c_b | (c_p << 16);
    st.tok0 = c_rt0 * 64 + c_wi * kTok;
    st.tok_end = c_tok_end;
    st.ltot = c_ltot;
    st.req0 = c_cc;
    st.tiles = c_tiles;
  };

  // ---- the registers the next WI lands in ------------------------------------------------------------------------
  u32x4 kr[2][4];    // [16-token block][4 token rows x 256 B per instruction]
  u32x4 vr[2][4];
  uint32_t qsc[kHP];
  Stage nx;          // descriptor of the WI those registers belong to
  const int kts = static_cast<int>(a.k_token_stride), vts = static_cast<int>(a.v_token_stride);
  const int k_voff = (lane >> 4) * kts + (lane & 15) * 16;  // row lane/16, 16-byte chunk lane%16 of the 256 B
  const int v_voff = (lane >> 4) * vts + (lane & 15) * 16;
  const int k_voff16 = k_voff + 16 * kts, v_voff16 = v_voff + 16 * vts;  // block 1 of a WI inside the same page
  const int ks1 = sgpr(4 * kts), ks2 = sgpr(8 * kts), ks3 = sgpr(12 * kts);
  const int vs1 = sgpr(4 * vts), vs2 = sgpr(8 * vts), vs3 = sgpr(12 * vts);
  // SGPR economy matters (a wave issues at most one instruction every 4 cycles, scalar ones included): with 32- or
  // 64-token pages a WI lies inside one page, so there is ONE page lookup and one K / one V base per WI, block 1 is
  // block 0 + 16 token rows (a second lane offset) and only its num_records word differs; 16-token pages take the
  // general two-lookup path.
  // The cursor runs one WI ahead of the loads: `nn` is the WI that will be issued next and its page ids are
  // requested (s_load through the scalar cache) a whole iteration before issue() needs them - a page-table
  // lookup in front of every batch of loads was ~0.5 us of exposed latency per WI.
  Stage nn;
  int nn_pid0 = 0, nn_pid1 = 0;
  auto prefetch_pages = [&]() __attribute__((always_inline)) {
    snapshot(nn);
    const bool valid = nn.flags & 1;
    const cint_ptr bid_row = as_const(a.block_ids) + static_cast<long>(nn.bp & 0xffff) * a.max_blocks;
    const bool on0 = valid && nn.tok0 < nn.tok_end, on1 = valid && nn.tok0 + 16 < nn.tok_end;
    nn_pid0 = on0 ? bid_row[nn.tok0 >> a.page_shift] : 0;
    nn_pid1 = (a.page_shift < 5 && on1) ? bid_row[(nn.tok0 + 16) >> a.page_shift] : 0;
    if (c_valid) advance();
  };
  const bool mem = a.dev_nomem == 0;  // development key 15 = 1: K / V loads fetch nothing (compute-only timing)
  auto issue = [&]() __attribute__((always_inline)) {
    nx = nn;
    const Stage& d = nx;
    const int dp = d.bp >> 16;
    const bool valid = d.flags & 1;
    const int pid0 = sgpr(nn_pid0), pid1 = sgpr(nn_pid1);
    const int pair_off = dp * kRow;
    const bool on0 = valid && d.tok0 < d.tok_end, on1 = valid && d.tok0 + 16 < d.tok_end;
    const int tk0 = on0 ? d.tok0 : 0, tk1 = on1 ? d.tok0 + 16 : 0;
    const i32x4 k0 = srd(kbase + pid0 * a.k_block_stride + static_cast<long>(tk0 & page_mask) * kts + pair_off,
                         on0 && mem ? 0xffffffffu : 0u);
    const i32x4 v0 = srd(vbase + pid0 * a.v_block_stride + static_cast<long>(tk0 & page_mask) * vts + pair_off,
                         on0 && mem ? 0xffffffffu : 0u);
    if (a.page_shift >= 5) {  // a WI lies inside one page: block 1 = block 0 + 16 token rows, own num_records
      const int nrec1 = sgpr(on1 && mem ? -1 : 0);
      const i32x4 k1 = i32x4{sgpr(k0[0]), sgpr(k0[1]), nrec1, 0x00020000};
      const i32x4 v1 = i32x4{sgpr(v0[0]), sgpr(v0[1]), nrec1, 0x00020000};
      ld_x4x4<kAux>(kr[0], k_voff, k0, ks1, ks2, ks3);
      ld_x4x4<kAux>(kr[1], k_voff16, k1, ks1, ks2, ks3);
      ld_x4x4<kAux>(vr[0], v_voff, v0, vs1, vs2, vs3);
      ld_x4x4<kAux>(vr[1], v_voff16, v1, vs1, vs2, vs3);
    } else {  // 16-token pages: block 1 lives in its own page
      const i32x4 k1 = srd(kbase + pid1 * a.k_block_stride + static_cast<long>(tk1 & page_mask) * kts + pair_off,
                           on1 && mem ? 0xffffffffu : 0u);
      const i32x4 v1 = srd(vbase + pid1 * a.v_block_stride + static_cast<long>(tk1 & page_mask) * vts + pair_off,
                           on1 && mem ? 0xffffffffu : 0u);
      ld_x4x4<kAux>(kr[0], k_voff, k0, ks1, ks2, ks3);
      ld_x4x4<kAux>(kr[1], k_voff, k1, ks1, ks2, ks3);
      ld_x4x4<kAux>(vr[0], v_voff, v0, vs1, vs2, vs3);
      ld_x4x4<kAux>(vr[1], v_voff, v1, vs1, vs2, vs3);
    }
    prefetch_pages();  // the WI after this one: its page ids are on their way while this one computes
  };

  // ---- LDS stage addressing (loop-invariant per lane) ----------------------------------------------------------
  // stage image [token 0..31][256 B]; the 16-byte chunk c of token t sits in slot c ^ key(t), key(t) =
  // (t & 15) ^ ((t >> 4) << 3): the 16 rows of a ds_read_b128 / the 8 rows of a transpose read hit 16 / 8
  // different slots.  All addresses are (loop-invariant base) ^ (compile-time constant).
  uint8_t* my_lds = s_wave[wave];
  float* my_so = reinterpret_cast<float*>(my_lds);
  const uint32_t lds0 = static_cast<uint32_t>(reinterpret_cast<uintptr_t>((lds_u8*)my_lds));  // LDS byte address of the stage
  // writes: lane (r4 = lane / 16, c = lane % 16) of instruction (tb, q) holds chunk c of token tb * 16 + q * 4 + r4
  const uint32_t w0_inv = lds0 + (lane >> 4) * kRow + (((lane & 15) ^ (lane >> 4)) * 16);
  // K reads: lane (n, g), token tb * 16 + n, chunk hh * 8 + g + 4 c
  const uint32_t r0_inv = lds0 + n * kRow + ((g ^ n) * 16);
  // V transpose reads: lane (i = lane % 16, g): row j = i / 2 of the 8 x 16 tile is token 16 (j / 4) + 4 g + (j % 4)
  // (the k-slot order of P: slot hb * 4 + r <-> token 16 hb + 4 g + r), 8-byte half i % 2, chunk hh * 8 + jj
  const int tj = (lane & 15) >> 1;
  const int ttok = 16 * (tj >> 2) + 4 * g + (tj & 3);
  const uint32_t t0_inv = lds0 + kVOff + ttok * kRow + ((((ttok & 15) ^ ((ttok >> 4) << 3))) * 16) + (lane & 1) * 8;

  // ---- per-task state ----------------------------------------------------------------------------------------
  u32x4 qf[kHP][2];      // fp8 Q fragments: 16-byte chunks g and g + 4 of row n
  float row_scale[kHP];  // qscale * kscale / sqrt(d) * log2(e)
  float out_scale[kHP];  // vscale (l_run carries the factor 256 of P~)
  f32x4 o[kHP][8];       // O^T: o[hh][jj][r] = dim jj * 16 + 4 g + r of q row n
  float m_run[kHP], l_run[kHP];
  // Q fragments + q scales of the task (b, p), straight into qf / qsc.  Called when the PREVIOUS task has been
  // finished (qf is dead then) for the task of the WI that is already in flight, so these 6 loads are the
  // youngest in the queue: the K / V waits of that WI only get more conservative, and its last wait (vmcnt(0))
  // covers them.  One descriptor for the pair - the 2 G q heads of two adjacent kv heads are contiguous - bounded to
  // the request's Sq rows: lanes of the rows past rows_valid read zeros.
  auto load_q = [&](int db, int dp) __attribute__((always_inline)) {
    const int q_voff = (n >> a.g_shift) * a.ldq + (n & (G - 1)) * 128 + g * 16;
    const int s_voff = ((n >> a.g_shift) * a.qscale_stride + (n & (G - 1))) * 4;
    const i32x4 rq = srd(qbase + static_cast<long>(db) * Sq * a.ldq + ((dp * kHP) << a.g_shift) * 128,
                         static_cast<unsigned>((Sq - 1) * a.ldq + kHP * G * 128));
    const i32x4 rsq = srd(a.qscale + static_cast<long>(db) * Sq * a.qscale_stride + ((dp * kHP) << a.g_shift),
                          static_cast<unsigned>(((Sq - 1) * a.qscale_stride + kHP * G) * 4));
    ld_q3(qf[0], qsc[0], q_voff, rq, s_voff, rsq);
    ld_q3(qf[1], qsc[1], q_voff + G * 128, rq, s_voff + G * 4, rsq);
  };
  auto reset_state = [&]() __attribute__((always_inline)) {
#pragma unroll
    for (int hh = 0; hh < kHP; ++hh) {
#pragma unroll
      for (int jj = 0; jj < 8; ++jj) o[hh][jj] = f32x4{0.f, 0.f, 0.f, 0.f};
      m_run[hh] = kNegInf;
      l_run[hh] = 0.f;
    }
  };

  // ---- end of a task: merge the 4 waves and emit (the stage regions are idle: they double as s_o) ---------------
  auto finish_task = [&](const Stage& d) __attribute__((always_inline)) {
    const int db = d.bp & 0xffff, dp = d.bp >> 16;
    // tile t of the request sits at cost position req0 + kOvh + t
    const int first_rng = (d.req0 + kOvh) / per;
    const int nchunks = (d.req0 + kOvh + d.tiles - 1) / per - first_rng + 1;
    const int ichunk = rng - first_rng;
#pragma unroll
    for (int hh = 0; hh < kHP; ++hh) {
      const float l = row4_sum(l_run[hh]);
      if (g == 0) {
        s_m[hh][wave][n] = m_run[hh];
        s_l[hh][wave][n] = l;
      }
#pragma unroll
      for (int jj = 0; jj < 8; ++jj)
        *reinterpret_cast<f32x4*>(&my_so[(hh * 16 + n) * 128 + jj * 16 + g * 4]) = o[hh][jj];
    }
    __syncthreads();
    {
      const int row16 = tid >> 4, c8 = tid & 15;
#pragma unroll
      for (int hh = 0; hh < kHP; ++hh) {
        float mw[kWaves], M = kNegInf;
#pragma unroll
        for (int w = 0; w < kWaves; ++w) {
          mw[w] = s_m[hh][w][row16];
          M = fmaxf(M, mw[w]);
        }
        const float Mu = M == kNegInf ? 0.f : M;
        float L = 0.f, acc[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) acc[i] = 0.f;
#pragma unroll
        for (int w = 0; w < kWaves; ++w) {
          const float wgt = __builtin_amdgcn_exp2f(mw[w] - Mu);
          L += wgt * s_l[hh][w][row16];
          const float* so = reinterpret_cast<const float*>(s_wave[w]) + (hh * 16 + row16) * 128 + c8 * 8;
          const f32x4 x0 = *reinterpret_cast<const f32x4*>(so);
          const f32x4 x1 = *reinterpret_cast<const f32x4*>(so + 4);
#pragma unroll
          for (int i = 0; i < 4; ++i) {
            acc[i] = fmaf(wgt, x0[i], acc[i]);
            acc[4 + i] = fmaf(wgt, x1[i], acc[4 + i]);
          }
        }
        const float inv = (L > 0.f ? 1.0f / L : 0.f) * out_scale[hh];
#pragma unroll
        for (int i = 0; i < 8; ++i) acc[i] *= inv;
        if (row16 < rows_valid) {
          const int h = dp * kHP + hh;
          if (nchunks == 1) {
            const int rs = row16 >> a.g_shift;
            uint16_t* dst = a.y + (static_cast<long>(db) * Sq + rs) * a.ldy + ((h << a.g_shift) + (row16 & (G - 1))) * 128 +
                            c8 * 8;
            u32x4 pk;
#pragma unroll
            for (int i = 0; i < 4; ++i) pk[i] = pack_bf16x2(acc[2 * i], acc[2 * i + 1]);
            st16(dst, pk);
          } else {
            // fp32 partial + base-2 LSE of this chunk, written THROUGH to memory (sc1): the workgroup that
            // arrives last at the request reads them with sc1 loads - per-XCD L2s are not coherent, and
            // write-through stores + a drained counter are the cheap valid hand-off (no cache-wide fences)
            const long slot = (static_cast<long>(wg) * 2 + (ichunk == 0 ? 1 : 0)) * kHP + hh;
            const int off = static_cast<int>(((slot * 16 + row16) * 128 + c8 * 8) * 4);
            __builtin_amdgcn_raw_buffer_store_b128(u32x4{__float_as_uint(acc[0]), __float_as_uint(acc[1]), __float_as_uint(acc[2]),
                                                         __float_as_uint(acc[3])}, part_rs, off, 0, 16);
            __builtin_amdgcn_raw_buffer_store_b128(u32x4{__float_as_uint(acc[4]), __float_as_uint(acc[5]), __float_as_uint(acc[6]),
                                                         __float_as_uint(acc[7])}, part_rs, off + 16, 0, 16);
            if (c8 == 0)
              __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(L > 0.f ? M + __builtin_amdgcn_logf(L) - 8.0f : kNegInf),
                                                    lse_rs, static_cast<int>((slot * 16 + row16) * 4), 0, 16);
          }
        }
      }
    }
    if (!a.in_kernel_combine) {
      if (tid == 0 && ichunk == 0) {  // chunk table for the combine kernel (every request: 1 = nothing to merge)
        int* e = a.arrive + static_cast<long>(npair) * B + 2 * (static_cast<long>(dp) * B + db);
        e[0] = nchunks;
        e[1] = first_rng;
      }
    }
    if (nchunks > 1 && a.in_kernel_combine) {
      // ---- split request: take a ticket; the last chunk to arrive merges all of them (reference: the last
      // CTA of a request reduces, static_splitk_kernels.cuh:362-377; combine math: splitk_combine_kernels.cuh)
      int* cnt = a.arrive + static_cast<long>(dp) * B + db;
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // my partial stores have reached memory
      __syncthreads();
      if (tid == 0) {
        // arrival count tagged with this launch's epoch: whatever the word held before (an aborted launch, a buffer
        // that was never cleared) reads as "no arrivals yet"; the last arriver leaves 0 behind, which no epoch matches
        int cur = __hip_atomic_load(cnt, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT), next;
        do {
          next = (cur >> 16) == a.epoch ? cur + 1 : ((a.epoch << 16) | 1);
        } while (!__hip_atomic_compare_exchange_strong(cnt, &cur, next, __ATOMIC_RELAXED, __ATOMIC_RELAXED,
                                                       __HIP_MEMORY_SCOPE_AGENT));
        s_ticket = next & 0xffff;
      }
      __syncthreads();
      if (s_ticket == nchunks) {
        const int row16 = tid >> 4, c8 = tid & 15;
        if (row16 < rows_valid) {
#pragma unroll 1
          for (int hh = 0; hh < kHP; ++hh) {
            auto slot_of = [&](int c) __attribute__((always_inline)) {
              return ((static_cast<long>(first_rng + c) * npair + dp) * 2 + (c == 0 ? 1 : 0)) * kHP + hh;
            };
            float M = kNegInf;
            for (int c0 = 0; c0 < nchunks; c0 += 8) {
              float l8[8];
#pragma unroll
              for (int u = 0; u < 8; ++u) {
                const int c = c0 + u < nchunks ? c0 + u : nchunks - 1;
                l8[u] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(lse_rs, static_cast<int>((slot_of(c) * 16 + row16) * 4), 0, 16));
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
              u32x4 x0[4], x1[4];
#pragma unroll
              for (int u = 0; u < 4; ++u) {
                const int c = c0 + u < nchunks ? c0 + u : nchunks - 1;
                const long slot = slot_of(c);
                l4[u] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(lse_rs, static_cast<int>((slot * 16 + row16) * 4), 0, 16));
                const int off = static_cast<int>(((slot * 16 + row16) * 128 + c8 * 8) * 4);
                x0[u] = __builtin_amdgcn_raw_buffer_load_b128(part_rs, off, 0, 16);
                x1[u] = __builtin_amdgcn_raw_buffer_load_b128(part_rs, off + 16, 0, 16);
              }
#pragma unroll
              for (int u = 0; u < 4; ++u) {
                const float wgt = c0 + u < nchunks ? __builtin_amdgcn_exp2f(l4[u] - Mu) : 0.f;
                W += wgt;
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                  acc[i] = fmaf(wgt, __uint_as_float(x0[u][i]), acc[i]);
                  acc[4 + i] = fmaf(wgt, __uint_as_float(x1[u][i]), acc[4 + i]);
                }
              }
            }
            const float inv = W > 0.f ? 1.0f / W : 0.f;
            const int rs = row16 >> a.g_shift;
            const int h = dp * kHP + hh;
            uint16_t* dst = a.y + (static_cast<long>(db) * Sq + rs) * a.ldy + ((h << a.g_shift) + (row16 & (G - 1))) * 128 + c8 * 8;
            u32x4 pk;
#pragma unroll
            for (int i = 0; i < 4; ++i) pk[i] = pack_bf16x2(acc[2 * i] * inv, acc[2 * i + 1] * inv);
            st16(dst, pk);
          }
        }
        if (tid == 0) __hip_atomic_store(cnt, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);  // ready for the next call
      }
    }
    __syncthreads();
    reset_state();
    if (nx.flags & 1) load_q(nx.bp & 0xffff, nx.bp >> 16);  // nx: the WI in flight = the next task's first WI
  };

  // ---- prologue ------------------------------------------------------------------------------------------------
  reset_state();
#pragma unroll
  for (int hh = 0; hh < kHP; ++hh) {
    qf[hh][0] = qf[hh][1] = u32x4{0u, 0u, 0u, 0u};
    row_scale[hh] = out_scale[hh] = 0.f;
    qsc[hh] = 0u;
  }
  prefetch_pages();
  issue();
  load_q(nx.bp & 0xffff, nx.bp >> 16);

  // ---- main loop: one wave-iteration per trip -------------------------------------------------------------
  while (true) {
    const Stage d = nx;  // the WI whose loads are landing
    if (!(d.flags & 1)) break;
    // keep the ~32 LDS addresses of a WI out of loop-invariant registers: they are one XOR away from these three
    // bases, and 32 pinned VGPRs were the difference between 2 waves per SIMD and spilling
    uint32_t w0 = w0_inv, r0 = r0_inv, t0 = t0_inv;
    asm volatile("" : "+v"(w0), "+v"(r0), "+v"(t0));
    // registers -> the wave's LDS stage (token rows of 256 B, chunks swizzled)
    wait_x4x4<kNV + 4>(kr[0]);
    wait_x4x4<kNV>(kr[1]);
#pragma unroll
    for (int tb = 0; tb < 2; ++tb)
#pragma unroll
      for (int q = 0; q < 4; ++q)
        *reinterpret_cast<lds_u32x4*>(static_cast<uint32_t>((w0 ^ (((q << 2) ^ (tb << 3)) * 16)) + (tb * 16 + q * 4) * kRow)) = kr[tb][q];
    wait_x4x4<4>(vr[0]);
    wait_x4x4<0>(vr[1]);
#pragma unroll
    for (int tb = 0; tb < 2; ++tb)
#pragma unroll
      for (int q = 0; q < 4; ++q)
        *reinterpret_cast<lds_u32x4*>(static_cast<uint32_t>((w0 ^ (((q << 2) ^ (tb << 3)) * 16)) + kVOff + (tb * 16 + q * 4) * kRow)) = vr[tb][q];
    // first WI of a task: its Q loads were the youngest in the queue, the vmcnt(0) above covered them
    // (rows past rows_valid came back as zeros from the bounded descriptors)
    if (d.flags & 4) {
      wait_q<0>(qf[0], qf[1], qsc[0], qsc[1]);
      const float kmul = as_constf(a.kscale)[0];
#pragma unroll
      for (int hh = 0; hh < kHP; ++hh) {
        row_scale[hh] = a.scale_log2 * __uint_as_float(qsc[hh]) * kmul;
        out_scale[hh] = as_constf(a.vscale)[0];  // the 1/256 of the reference formula cancels: l = 256 sum p
      }
    }
    // the registers are free again: the next WI (of this or the next task) goes in flight now
    issue();

    // S^T = K Q^T: one K = 128 MFMA per 16-token block and head (lane (n, g) supplies chunks g and g + 4 of its
    // row on both sides - a dot product does not care which lane slot a dim sits in)
    f32x4 sacc[kHP][2];
#pragma unroll
    for (int tb = 0; tb < 2; ++tb)
#pragma unroll
      for (int hh = 0; hh < kHP; ++hh) {
        const u32x4 k0 = *reinterpret_cast<const lds_u32x4*>(static_cast<uint32_t>((r0 ^ (((hh << 3) ^ (tb << 3)) * 16)) + tb * 16 * kRow));
        const u32x4 k1 = *reinterpret_cast<const lds_u32x4*>(static_cast<uint32_t>((r0 ^ (((hh << 3) ^ (tb << 3) ^ 4) * 16)) + tb * 16 * kRow));
        const i32x8 kv8 = {static_cast<int>(k0[0]), static_cast<int>(k0[1]), static_cast<int>(k0[2]), static_cast<int>(k0[3]),
                           static_cast<int>(k1[0]), static_cast<int>(k1[1]), static_cast<int>(k1[2]), static_cast<int>(k1[3])};
        const i32x8 qv8 = {static_cast<int>(qf[hh][0][0]), static_cast<int>(qf[hh][0][1]), static_cast<int>(qf[hh][0][2]),
                           static_cast<int>(qf[hh][0][3]), static_cast<int>(qf[hh][1][0]), static_cast<int>(qf[hh][1][1]),
                           static_cast<int>(qf[hh][1][2]), static_cast<int>(qf[hh][1][3])};
        sacc[hh][tb] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(kv8, qv8, f32x4{0.f, 0.f, 0.f, 0.f}, 0, 0, 0, 0, 0, 0);
      }

    // online softmax in base 2, per head; lane (n, g) holds tokens 16 tb + 4 g + r of q row n.
    // P~ = e4m3(256 p) is computed as exp2(x - m + 8) (<= 256 < 448: no clamp needed) and the row sum is kept in
    // the same units (l = 256 sum p; the 1/256 of the reference formula is folded into the final scale).
    const bool masked = d.tok0 + kTok > d.ltot - Sq + 1 || d.tok0 + kTok > d.tok_end;
    uint32_t pf[kHP][2];
#pragma unroll
    for (int hh = 0; hh < kHP; ++hh) {
#pragma unroll
      for (int tb = 0; tb < 2; ++tb)
#pragma unroll
        for (int r = 0; r < 4; ++r) sacc[hh][tb][r] *= row_scale[hh];
      if (masked) {  // wave-uniform: only the WIs that hold a request's last tokens
        const int sq_row = n >> a.g_shift;
        const int lim = (d.tok_end - 1 < d.ltot - Sq + sq_row) ? d.tok_end - 1 : d.ltot - Sq + sq_row;
#pragma unroll
        for (int tb = 0; tb < 2; ++tb)
#pragma unroll
          for (int r = 0; r < 4; ++r)
            sacc[hh][tb][r] = (d.tok0 + tb * 16 + g * 4 + r) <= lim ? sacc[hh][tb][r] : kNegInf;
      }
      float mt = __builtin_fmaxf(__builtin_fmaxf(sacc[hh][0][0], sacc[hh][0][1]), sacc[hh][0][2]);
      mt = __builtin_fmaxf(__builtin_fmaxf(mt, sacc[hh][0][3]), sacc[hh][1][0]);
      mt = __builtin_fmaxf(__builtin_fmaxf(mt, sacc[hh][1][1]), sacc[hh][1][2]);
      mt = __builtin_fmaxf(mt, sacc[hh][1][3]);
      mt = row4_max(mt);
      const float m_new = fmaxf(m_run[hh], mt);
      const float m_use = m_new == kNegInf ? 0.f : m_new;
      const float m8 = m_use - 8.0f;
      float psum = 0.f;
#pragma unroll
      for (int tb = 0; tb < 2; ++tb) {
        float pr[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          pr[r] = __builtin_amdgcn_exp2f(sacc[hh][tb][r] - m8);
          psum += pr[r];
        }
        int w = __builtin_amdgcn_cvt_pk_fp8_f32(pr[0], pr[1], 0, false);
        w = __builtin_amdgcn_cvt_pk_fp8_f32(pr[2], pr[3], w, true);
        pf[hh][tb] = static_cast<uint32_t>(w);
      }
      // rescale only when some row's maximum moved (after the first few WIs of a long request it rarely does)
      if (__builtin_amdgcn_ballot_w64(m_new != m_run[hh]) != 0) {
        const float alpha = __builtin_amdgcn_exp2f(m_run[hh] - m_use);
        l_run[hh] *= alpha;
#pragma unroll
        for (int jj = 0; jj < 8; ++jj) o[hh][jj] *= alpha;
        m_run[hh] = m_new;
      }
      l_run[hh] += psum;
    }

    // O^T += V^T P^T: the transpose read hands every lane one dim (column) of an 8-token x 16-dim tile.
    constexpr int kTrBatch = 8;
    // Reads go out eight at a time ahead of their MFMAs (hipcc, left alone, reuses one register pair and
    // serialises read -> lgkmcnt(0) -> MFMA sixteen times: sixteen exposed LDS round trips per WI).
#pragma unroll
    for (int hh = 0; hh < kHP; ++hh)
#pragma unroll
      for (int j4 = 0; j4 < 8; j4 += kTrBatch) {
        v2i32 vt[kTrBatch];
#pragma unroll
        for (int u = 0; u < kTrBatch; ++u)
          vt[u] = __builtin_amdgcn_ds_read_tr8_b64_v2i32(
              reinterpret_cast<lds_v2i32*>(static_cast<uint32_t>(t0 ^ (((hh << 3) | (j4 + u)) * 16))));
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int u = 0; u < kTrBatch; ++u)
          o[hh][j4 + u] = __builtin_amdgcn_mfma_f32_16x16x32_fp8_fp8(
              pack64(static_cast<uint32_t>(vt[u][0]), static_cast<uint32_t>(vt[u][1])), pack64(pf[hh][0], pf[hh][1]),
              o[hh][j4 + u], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
      }
    if (d.flags & 2) finish_task(d);
  }
  // the loads issued for the WI past the end were no-ops, but they own the registers until they retire
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}

// ---- split-KV combine: y = sum_c 2^(lse_c - max) O_c / sum_c 2^(lse_c - max) ------------------------------
// (reference splitk_combine_kernels.cuh:140-322).  One workgroup per (kv head, request): chunk c of a
// request lives in workgroup first + c, slot 1 for c == 0 and slot 0 otherwise.
__global__ __launch_bounds__(kThreads) void decode2_combine_kernel(const Args a) {
  const int hh = blockIdx.x % kHP;
  const int pb = blockIdx.x / kHP;
  const int p = pb / a.num_batch, b = pb % a.num_batch;
  const int ltot = as_const(a.lens)[b] + (a.new_kv_included ? 0 : a.num_seq_q);
  if (ltot <= 0) return;
  const int* table = a.arrive + (a.num_head_kv / kHP) * a.num_batch;
  const int nchunks = as_const(table)[2 * pb];
  if (nchunks <= 1) return;
  const int fw = as_const(table)[2 * pb + 1];
  const int tid = threadIdx.x;
  const int row = tid >> 4, c8 = tid & 15;
  const int rows_valid = a.num_seq_q << a.g_shift;
  if (row >= rows_valid) return;
  const int G = 1 << a.g_shift;
  const int npair = a.num_head_kv / kHP;
  auto slot_of = [&](int c) __attribute__((always_inline)) { return ((static_cast<long>(fw + c) * npair + p) * 2 + (c == 0 ? 1 : 0)) * kHP + hh; };
  float M = kNegInf;
  for (int c0 = 0; c0 < nchunks; c0 += 8) {
    float l8[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const int c = c0 + u < nchunks ? c0 + u : nchunks - 1;
      l8[u] = a.part_lse[slot_of(c) * 16 + row];
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
      l4[u] = a.part_lse[slot * 16 + row];
      const float* po = a.part_o + (slot * 16 + row) * 128 + c8 * 8;
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
  const int h = p * kHP + hh;
  uint16_t* dst = a.y + (static_cast<long>(b) * a.num_seq_q + rs) * a.ldy + ((h << a.g_shift) + (row & (G - 1))) * 128 +
                  c8 * 8;
  u32x4 pk;
#pragma unroll
  for (int i = 0; i < 4; ++i) pk[i] = pack_bf16x2(acc[2 * i] * inv, acc[2 * i + 1] * inv);
  st16(dst, pk);
}

int64_t workspace_bytes(int num_wg, int num_batch, int num_head_kv) {
  const int64_t part_o = static_cast<int64_t>(num_wg) * 2 * kHP * 16 * 128 * 4;
  const int64_t part_lse = static_cast<int64_t>(num_wg) * 2 * kHP * 16 * 4;
  const int64_t arrive = (static_cast<int64_t>(num_batch) * (num_head_kv / kHP + 1) * 12 + 15) / 16 * 16;  // counters + chunk table
  return part_o + part_lse + arrive;
}

bool eligible(const Args& a, int num_head_q, int block_size, int64_t k_head_stride, int64_t v_head_stride) {
  const int group = num_head_q / a.num_head_kv;
  return a.lens != nullptr && (a.num_head_kv % kHP) == 0 && a.num_seq_q * group <= 16 && k_head_stride == 128 &&
         v_head_stride == 128 && a.num_batch <= 64 * 16 && (block_size == 64 || block_size == 32 || block_size == 16) &&
         (a.k_token_stride % 16) == 0 && (a.v_token_stride % 8) == 0 && (a.k_block_stride % 16) == 0 &&
         (a.v_block_stride % 8) == 0;
}

int launch(Args a, void* workspace, int num_wg, int quant_type, hipStream_t stream) {
  char* ws = static_cast<char*>(workspace);
  a.part_o = reinterpret_cast<float*>(ws);
  ws += static_cast<int64_t>(num_wg) * 2 * kHP * 16 * 128 * 4;
  a.part_lse = reinterpret_cast<float*>(ws);
  ws += static_cast<int64_t>(num_wg) * 2 * kHP * 16 * 4;
  a.arrive = reinterpret_cast<int*>(ws);
  static std::atomic<int> epoch{0};
  a.epoch = (epoch.fetch_add(1, std::memory_order_relaxed) % 32767) + 1;  // 1 .. 32767, frozen inside a captured graph
  if (quant_type != 1) return HPC_ERR_UNSUPPORTED;  // per-token K scales: first-generation kernel
  if (hpc_dev_tuning_get(0) == 1)
    decode2_kernel<0><<<num_wg, kThreads, 0, stream>>>(a);
  else
    decode2_kernel<2><<<num_wg, kThreads, 0, stream>>>(a);
  HPC_CHECK_LAUNCH();
  if (!a.in_kernel_combine) {
    decode2_combine_kernel<<<a.num_batch * a.num_head_kv, kThreads, 0, stream>>>(a);
    HPC_CHECK_LAUNCH();
  }
  return HPC_OK;
}

}  // namespace decode2_old
}  // namespace hpc
