// Grouped FP8 GEMM, 256 x 256 tile, two staggered wave groups, two MFMA sections per k-tile - gfx950.
//
// Same contract as group_gemm_tiled256.hip (reference src/group_gemm/kernels.cuh:215-892).  The
// 256 x 128 ring kernel stages 48 KB per k-step for 16 MFMAs per wave and its waves spend as long
// issuing LDS-DMA pieces and operand reads as the matrix pipe spends multiplying.  This form halves the
// staging per MFMA and takes most of what is left off the matrix pipe's critical path:
//   * tile = 256 weight rows x 256 tokens x 128 k-bytes; 8 waves = 2 groups (weight-row halves) x 4
//     (64-token strips); a wave owns 128 x 64 outputs = 8 x 4 MFMA blocks, one
//     v_mfma_f32_16x16x128_f8f6f4 per block and k-tile (32 per k-tile and wave, 2048 matrix-pipe cycles per
//     SIMD), 24 ds_read_b128 and 8-9 DMA pieces (the 256 x 128 kernel: 16 MFMAs, 16-20 reads, 6-7 pieces);
//   * a wave's k-tile is two SECTIONS of 16 MFMAs (rows 0-63 / rows 64-127 of its half, all 64 tokens):
//     [operand reads | DMA pieces | counted vmcnt + lgkmcnt(0) | barrier | 16 MFMAs at raised priority |
//     barrier].  Waves 0-3 and waves 4-7 - the two waves of every SIMD - run ONE BARRIER apart, so while one
//     wave of a SIMD multiplies the other loads: the matrix pipe sees back-to-back MFMA sections, and
//     whatever a load section costs is free as long as it is shorter than 16 MFMAs (512 cycles);
//   * a DMA piece costs its wave 60-100 cycles of issue (address path).  Load section Y (8 reads) takes
//     four pieces, load section X (16 reads) two, and three go between the MFMAs of section Y (measured:
//     all nine between MFMAs 1.86 / 2.00 PFLOP/s on the two GEMMs of the MoE, this split 1.86 / 2.07,
//     four sections of 8 MFMAs per k-tile 1.78 / 1.91; with no DMA at all the loop runs at 2.5 / 2.8);
//   * staging is in four 16 KB units per k-tile, cut by the section that reads them: U0 = weight rows
//     0-63 of either group's half, U3 = rows 64-127, U1 / U2 = the first / second 32 tokens of every strip.
//     Two buffers; a unit is refilled behind the barrier after its last read (operand reads retire BEFORE a
//     section's barrier) and lands two or more sections before its first read; the waits are the constants
//     vmcnt(9|8) and vmcnt(6);
//   * the blockwise rescale is software-pipelined by hand: the four FMAs of block n follow the MFMA of
//     block n+2, and the last two blocks' are CARRIED into the wave's next MFMA section (behind its first
//     MFMAs).  Round 5: a load section holds no VALU work at all - with the partner wave of the SIMD
//     multiplying at raised priority, every VALU instruction of a load section waits for an issue slot
//     between the partner's MFMAs (s_memtime stamps: profiles/round5_moe_p8_section_profile.txt);
//   * three bodies share the launch (round 5): the full tile, a half tile for a group's last <= 128 rows
//     (half the token strips, the same sections) and a tail body for its last <= 64 rows (p8_tail_body: no
//     shared weight staging - each wave streams its own 32 weight rows through a private 3-stage ring, the
//     <= 64 token rows arrive in chunks of 3 k-slabs, one barrier per 3 k-tiles);
//   * LDS image, source-side XOR swizzle and the K = 128 MFMA operand convention as in
//     group_gemm_tiled256.hip; 2 x 64 KB + scales = 131 KB of LDS, one workgroup per CU;
//   * epilogue: neighbouring row blocks are exchanged between lane quarters (v_permlane16_swap) so that a
//     lane stores 16 contiguous bytes of an output row instead of 8.
#include "hpc_common.h"
#include "hpc_dev.h"
#include "../../include/hpc_amd.h"
#include "group_gemm.h"

namespace hpc {
namespace ggemm {
namespace {

constexpr int kThreads = 512;
constexpr int kBN = 256, kBM = 256, kBK = 128;
constexpr int kUnit = 128 * kBK;       // 16 KB: 128 rows of 128 k-bytes
// LDS image of the full / half-tile bodies (round 6): the two k-tile buffers INTERLEAVED per unit -
//   [U0 b0 | U0 b1 | U3 b0 | U3 b1 | U1 b0 | U1 b1 | U2 b0 | U2 b1]   (U0 / U3 weights early / late, U1 / U2 tokens early / late)
// so every weight operand read is ONE base register + a 16-bit immediate (< 64 KB) and every token operand read another:
// rounds 2-5 laid the buffers end to end (the second one past the offset field of a ds_read) and paid a second set of base
// registers - four VGPRs the ride-along rows (kExt) need.
constexpr int kAOff = 0;               // weight units: unit u (0 = U0, 1 = U3), buffer p at kAOff + (2 u + p) * kUnit
constexpr int kBOff = 4 * kUnit;       // token units: unit u (0 = U1, 1 = U2), buffer p at kBOff + (2 u + p) * kUnit
constexpr int kXsOff = 8 * kUnit;      // 2 x 1 KB of token scales + 1 KB nobody reads (idle waves' scale DMA)
// ride-along rows (kExt): 2 buffers x 16 token rows x 128 B, then 2 x 256 B of their scales (lanes 0-15 of a 64-lane piece)
constexpr int kExtOff = kXsOff + 3 * 1024;
constexpr int kExtXsOff = kExtOff + 2 * 2048;
constexpr int kLdsMain = kExtXsOff + 2 * 256;
// the HALF-tile body keeps THREE k-tiles in flight (round 6): it has no late-token unit, so three buffers of [U0 | U3 | U1] =
// 144 KB fit - [U0 b0 b1 b2 | U3 b0 b1 b2 | U1 b0 b1 b2], then 3 x 1 KB of token scales + 1 KB nobody reads
constexpr int kHalfBOff = 6 * kUnit;
constexpr int kHalfXsOff = 9 * kUnit;
constexpr int kLdsHalf = kHalfXsOff + 4 * 1024;
// the tail body's layout (p8_tail_body): per-wave weight rings, two chunk buffers of token slabs, their scales
constexpr int kTW = 3;                               // stages of a wave's weight ring = k-tiles of a token chunk
constexpr int kTWOff = 0;                            // [8 waves][kTW][32 rows x 128 B]
constexpr int kTXOff = kTWOff + 8 * kTW * 4096;      // [2 chunk buffers][kTW slabs][64 token rows x 128 B]
constexpr int kTSOff = kTXOff + 2 * kTW * 8192;      // [2][kTW][64 floats]
constexpr int kTDummy = kTSOff + 2 * kTW * 256;      // 256 B nobody reads (idle waves' scale DMA)
constexpr int kLdsTail = kTDummy + 256;
// the register-streamed tail body (p8_tail_body_r): no weights in LDS at all - two chunk buffers of token slabs and their scales
constexpr int kTR = 6;                               // k-slabs of a token chunk = register stages of the weight stream
constexpr int kRXOff = 0;                            // [2 chunk buffers][kTR slabs][64 token rows x 128 B]
constexpr int kRSOff = kRXOff + 2 * kTR * 8192;      // [2][kTR][64 floats]
constexpr int kRDummy = kRSOff + 2 * kTR * 256;
static_assert(kRDummy + 256 <= kLdsTail, "the register-streamed tail body fits the tail body's LDS");
constexpr int kLds = (kLdsTail > kLdsMain ? kLdsTail : kLdsMain) > kLdsHalf ? (kLdsTail > kLdsMain ? kLdsTail : kLdsMain) : kLdsHalf;
static_assert(kLds <= 160 * 1024, "one workgroup per CU");

typedef __attribute__((address_space(3))) void lds_void;

template <int kN>
struct IntC {
  static constexpr int value = kN;
};

// The work item of this workgroup.  Items are dealt to the XCDs in contiguous eighths (workgroup `lin` -> XCD lin % 8,
// position lin / 8 of that XCD's share), so that the token tiles sharing a weight tile meet in one L2.  The 256-token
// tile counts come from the scan of ceil(len / 128) the callers already have (`cut`): c = cut[g + 1] - cut[g] 128-row
// tiles = c / 2 FULL 256-token tiles + (c odd) one tail tile of <= 128 rows, which runs the half-tile body.
//   order 0 (round 2-4): group -> weight tile -> token tile, tail tiles in place;
//   order 1 (round 5, development): all full tiles first (group -> weight tile -> token tile), then all tail tiles (group ->
//     weight tile).  A tail tile costs ~0.6 of a full one, and the dispatcher hands workgroups out in index order: with
//     the cheap items last the end of the launch is filled evenly (the down GEMM of the MoE has only ~9.4 items per CU).
//     Measured slower (the launcher says by how much): the tail tiles no longer meet their weight tile in L2.
// Ride-along rows (round 6, `ext`): a group that ends in a SHORT tail - f full tiles and then t <= 16 f rows - has no tail
// item at all: full tile mt carries rows [256 f + 16 mt, + 16) of the group as a 17th token block (p8_body<kExt>): one more
// MFMA per wave and section instead of a tail tile that re-streams the 256 x K weight tile for a dozen rows (at 512 +- 22
// routed rows per expert a tail tile cost ~0.6 of a full tile and the two GEMMs of the MoE ran 15-19 % behind exactly 512
// rows per expert).  Longer tails keep their half-tile / tail-body item.
// One round of lane-parallel loads per 64 groups; with <= 64 groups everything comes from one round.
constexpr int kExtRows = 16;  // ride-along rows per full tile
struct Item {
  int e, mt, wt, m_cnt, m0;  // group, 256-token tile of the group, weight tile, rows of the group, its first row
  int ext0, ext_cnt;         // ride-along rows of this tile: group rows [ext0, ext0 + ext_cnt), ext_cnt = 0: none
  bool valid;
};
// Inclusive wave scan on the DPP path (round 6: six v_add with a DPP source instead of six ds_bpermute round trips + selects -
// every workgroup of the kernel runs this lookup before its first request for data): Hillis-Steele within the rows of 16
// (row_shr 1, 2, 4, 8: lanes without a source add 0), then lane 15 of row 0 / 2 into row 1 / 3 (row_bcast:15, rows 0xA) and
// lane 31 into rows 2 and 3 (row_bcast:31, rows 0xC).
__device__ __forceinline__ int wave_incl_scan(int v, int lane, bool old_path = false) {
  if (kHpcDevBuild && old_path) {  // development key 43 = 1: the ds_bpermute form of rounds 2-5
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
      const int u = __shfl_up(v, o, 64);
      v += lane >= o ? u : 0;
    }
    return v;
  }
  v += __builtin_amdgcn_update_dpp(0, v, 0x111, 0xF, 0xF, false);  // row_shr:1
  v += __builtin_amdgcn_update_dpp(0, v, 0x112, 0xF, 0xF, false);  // row_shr:2
  v += __builtin_amdgcn_update_dpp(0, v, 0x114, 0xF, 0xF, false);  // row_shr:4
  v += __builtin_amdgcn_update_dpp(0, v, 0x118, 0xF, 0xF, false);  // row_shr:8
  v += __builtin_amdgcn_update_dpp(0, v, 0x142, 0xA, 0xF, false);  // row_bcast:15 -> rows 1, 3
  v += __builtin_amdgcn_update_dpp(0, v, 0x143, 0xC, 0xF, false);  // row_bcast:31 -> rows 2, 3
  return v;
}
__device__ __forceinline__ int lane_of(int v, int l, bool old_path = false) {  // l wave-uniform
  return (kHpcDevBuild && old_path) ? __shfl(v, l, 64) : __builtin_amdgcn_readlane(v, l);
}
__device__ __forceinline__ Item locate_item(cint_ptr cut, const int* seqlens, const int* cu_seqlens, int num_group,
                                            int nt, int lane, int lin, int order, int ext_on, bool old_path = false) {
  Item it = {0, 0, 0, 0, 0, 0, 0, false};
  const int x = lin & 7, j = lin >> 3;
  // round 0 of every per-group quantity is loaded once, side by side
  const bool on0 = lane < num_group;
  const int c_first = on0 ? cut[lane + 1] - cut[lane] : 0;
  const int len_first = on0 ? seqlens[lane] : 0, row_first = on0 ? cu_seqlens[lane] : 0;
  // this lane's group of round g0: full tiles f, tail tile h (0 / 1) - a short tail rides along with the full tiles (h = 0)
  auto split = [&](int g0, int& f, int& h) {
    const bool on = g0 + lane < num_group;
    const int c = g0 == 0 ? c_first : (on ? cut[g0 + lane + 1] - cut[g0 + lane] : 0);
    f = c >> 1;
    h = c & 1;
    if (ext_on && h && f > 0) {
      const int len = g0 == 0 ? len_first : as_const(seqlens)[g0 + lane];
      if (len - (f << 8) <= kExtRows * f) h = 0;
    }
  };
  // pass 1: totals
  int tot_f = 0, tot_h = 0;
  for (int g0 = 0; g0 < num_group; g0 += 64) {
    int f, h;
    split(g0, f, h);
    const int both = lane_of(wave_incl_scan(f | (h << 20), lane, old_path), 63, old_path);  // full tiles of 64 groups < 2^20
    tot_f += both & 0xfffff;
    tot_h += both >> 20;
  }
  int idx, kind;  // kind 0: every 256-token tile of a group (order 0), 1: full tiles only, 2: tail tiles only
  if (order == 0) {
    const int total = (tot_f + tot_h) * nt, chunk = (total + 7) >> 3;
    idx = x * chunk + j;
    kind = 0;
    if (j >= chunk || idx >= total) return it;
  } else {
    const int total_f = tot_f * nt, total_h = tot_h * nt;
    const int chunk_f = (total_f + 7) >> 3, chunk_h = (total_h + 7) >> 3;
    if (j < chunk_f) {
      idx = x * chunk_f + j;
      kind = 1;
      if (idx >= total_f) return it;
    } else {
      if (j - chunk_f >= chunk_h) return it;
      idx = x * chunk_h + j - chunk_f;
      kind = 2;
      if (idx >= total_h) return it;
    }
  }
  // pass 2: the group that holds item idx
  int base = 0;
  for (int g0 = 0; g0 < num_group; g0 += 64) {
    int f, h;
    split(g0, f, h);
    const int t = kind == 0 ? f + h : (kind == 1 ? f : h);
    const int inc = wave_incl_scan(t, lane, old_path);
    const unsigned long long hit = __ballot(g0 + lane < num_group && idx < (base + inc) * nt);
    if (hit) {
      const int l = __builtin_ctzll(hit);
      const int tl = lane_of(t, l, old_path), fl = lane_of(f, l, old_path), hl = lane_of(h, l, old_path);
      const int rem = idx - (base + lane_of(inc, l, old_path) - tl) * nt;
      it.e = g0 + l;
      if (kind == 2) {
        it.wt = rem;
        it.mt = fl;
      } else {
        it.wt = rem / tl;
        it.mt = rem % tl;
      }
      if (g0 == 0) {
        it.m_cnt = lane_of(len_first, l, old_path);
        it.m0 = lane_of(row_first, l, old_path);
      } else {
        it.m_cnt = as_const(seqlens)[it.e];
        it.m0 = as_const(cu_seqlens)[it.e];
      }
      // ride-along rows: the group has no tail item (hl == 0) but rows past its full tiles
      if (ext_on && hl == 0 && it.mt < fl) {
        const int r0 = (fl << 8) + kExtRows * it.mt;
        const int left = it.m_cnt - r0;
        it.ext0 = r0;
        it.ext_cnt = left > kExtRows ? kExtRows : (left > 0 ? left : 0);
      }
      it.valid = true;
      return it;
    }
    base += lane_of(inc, 63, old_path);
  }
  return it;
}

// ---- variants of the k-loop (development key 22 selects one in the development build; the product is CfgProduct) -------
// Where a wave issues its DMA pieces of a k-tile: slot 0 = in the load section, slot n + 1 = behind MFMA n of the MMA
// section that follows it.  X half of k-tile T: U3(T+1) q0 q1 (free since the barrier that ended the other group's load
// section Y of T-1).  Y half: U0(T+2) q0 q1, U1(T+2) q0 q1, U2(T+2) q0 q1 (full body only), scales(T+2) - their units are
// read for the last time in the other group's load section X of T, which ends one barrier before this wave's load section Y.
// The two waits follow from the slots: the wait that ends load section X of T must see U3(T) landed - everything but the
// Y-half pieces of T-1 and the pieces of this load section; the wait that ends load section Y must see all of k-tile T+1's
// Y-half pieces landed - everything but the X-half pieces of T and the pieces of this load section.
struct CfgProduct {
  // the rescale of a section's last kDist blocks rides under the next section's first MFMAs instead of sitting in the load
  // section behind the barrier (round 5, measured: gate-up / down GEMM of the MoE 3507 / 1671 -> 3088 / 1548 us routed,
  // 2.02 / 2.13 -> 2.21 / 2.32 PFLOP/s at 512 rows per expert, bit-identical results - a load section's VALU work loses
  // every issue slot to the MMA wave of its SIMD)
  static constexpr bool kCarry = true;
  static constexpr int kDist = 2;          // the rescale of block n follows the MFMA of block n + kDist
  static constexpr int kEarly = 0;         // the barrier that ends an MMA section sits in front of its last kEarly MFMAs
  static constexpr bool kDmaFirst = false; // load sections issue their DMA pieces before their operand reads (measured: -4 %)
  static constexpr bool kPrio = true;      // s_setprio 1 around the MFMAs
  static constexpr bool kProf = false;     // s_memtime log of the section boundaries (development)
  static constexpr int fx(int i) { constexpr int t[2] = {0, 0}; return t[i]; }
  static constexpr int fy(int i) { constexpr int t[7] = {0, 0, 0, 0, 4, 9, 13}; return t[i]; }
  static constexpr int hx(int i) { constexpr int t[2] = {0, 0}; return t[i]; }
  static constexpr int hy(int i) { constexpr int t[7] = {0, 0, 0, 0, -1, -1, 7}; return t[i]; }
};
#ifdef HPC_DEV
struct CfgProf : CfgProduct { static constexpr bool kProf = true; };
struct CfgRound4 : CfgProduct { static constexpr bool kCarry = false; };  // the round-4 loop (tails behind the barrier)
struct CfgRound4Prof : CfgRound4 { static constexpr bool kProf = true; };
struct CfgNoPrio : CfgProduct { static constexpr bool kPrio = false; };
// (measured and removed, profiles/round5_moe_ggemm_ab.txt: rescale distance 3, the barrier in front of a section's last 2 / 4
//  MFMAs, DMA pieces before the operand reads, DMA slot schedules with no piece in a load section / in an MMA section -
//  all within noise of or behind the product loop once the carried tails were in)
#endif

// kNoDma (development key 18 = 1, timing only - results are wrong): no DMA inside the k-loop
// kAct: the gate-up GEMM of the fused MoE - a tile is 128 gate rows (wave group 0) + the 128 up rows of the same
// columns (group 1); the epilogue applies SiLU(gate) * up and the 128-block quantisation instead of storing y
// Blockwise rescale (kHasXs): the ARITHMETIC OF THE REFERENCE KERNEL (src/group_gemm/kernels.cuh:808-834) - every MFMA
// starts from zero (K = 128 = one scale block), `f = xs[token, kb] * ws[n / 128, kb]` is one fp32 multiply and the
// block's fp32 partial is folded into the running sum with ONE fused multiply-add per element, k blocks in order.
// What remains between this and a CPU restatement of that arithmetic is the matrix pipe's own rounding of a 128-term
// block sum (not correctly rounded even when the sum is representable: ~1e-4 of the bf16 outputs move by one ulp,
// tests/test_fuse_moe_blockwise.py::test_group_gemm_blockwise_is_the_reference_kernel_arithmetic).  Round 3 shipped
// a cheaper form (running sums kept in units of the current block's scale, tot' = tot / f_T: +2-6 %): it rounds
// differently from the reference kernel and clamped tiny scales - removed in round 4, parity first.
// kHalf: the HALF-TILE body for a group's last token tile when it holds <= 128 rows (at 512 +- 22 routed rows per
// expert half the groups end in a third 256-row tile for a dozen rows).  The rows sit in the EARLY 32 slots of every
// strip - tile row r -> slot (r / 32) * 64 + r % 32 - so the late-token unit (U2) is never fetched, its operand reads
// and the MFMAs of token blocks 2-3 do not exist: 16 instead of 32 MFMAs, 20 instead of 24 operand reads and 6-7
// instead of 8-9 DMA pieces per wave and k-tile, and in the fused activation epilogue group 0 finishes everything.
// The weight units are what they are: a tail tile still re-streams its 256 weight rows.
// kExt: the FULL body with 16 ride-along rows (locate_item: a group's short tail dealt to its full tiles) - a 17th token
// block of the tile, rows [ext0, ext0 + ext_cnt) of the group:
//   * staging: 2 KB per k-tile (16 rows x 128 B) + 16 scales, fetched in the SCALE slot of the DMA schedule by the waves whose
//     scale piece fetches nothing (waves 4-7: the four strips' scales are waves 0-3's) - waves 4 / 5 the two 8-row pieces,
//     wave 6 the scales: not one VMEM instruction more per wave, the counted waits are untouched;
//   * MFMAs: the block x the tile's 256 weight rows = 16 per k-tile, two per wave: wave (group, strip s) multiplies row
//     blocks s (rows 0-63 of its half, unit U0) and 4 + s (rows 64-127, U3) - BOTH behind the 16 MFMAs of section X (by then
//     U3 of this k-tile has landed for the whole group: every wave waited for its pieces in front of the barrier that opened
//     the section), so the ride-along buffers are read in section X only and refilled in section Y (slot 13), with barriers
//     in between for either group.  Section X is 18 MFMAs, section Y 16; the two groups alternate, a k-tile is 4 x 17;
//   * registers: the operands are read JUST IN TIME inside section X into the registers of row blocks that are finished
//     (the token block + row block s behind MFMA 7, row block 4 + s behind MFMA 11) - the section starts at the kernel's
//     register peak; what is new at the peak is 8 accumulators, 3 scale values and 5 read bases;
//   * the rescale pipeline is the section's own (block n follows MFMA n + 2; section X's last two blocks - the ride-along
//     ones - are folded under section Y's first MFMAs): the arithmetic per output element is that of every other body, so a
//     row's results do not depend on whether it rode along or went through the tail body (tests/test_fuse_moe_blockwise.py).
template <class Cfg, bool kHasXs, bool kNoDma, bool kAct, bool kHalf, bool kKTail, bool kExt = false>
__device__ __forceinline__ void p8_body(const Args& a, uint8_t* s_mem, int e, int mt0, int n0, int m_cnt, int m0, int ext0 = 0,
                                        int ext_cnt = 0) {
  static_assert(!kExt || (!kHalf && !kNoDma && !kKTail), "ride-along rows: full body only");
  // per-tensor scales + ride-along rows: there is no scale piece to ride in, so the body gets the seventh DMA slot anyway
  // (waves 4 / 5 fetch the rows, the others an empty piece: equal counts), and the block accumulates in the matrix pipe
  constexpr bool kXsSlot = kHasXs || kExt;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int r16 = lane & 15, g4 = lane >> 4;
  const int wn = wave >> 2, wm = wave & 3;  // group (weight-row half) / 64-token strip
  const int K = a.K, KB = a.KB;
  constexpr int kJ = kHalf ? 2 : 4;  // token blocks per strip
  // kNB buffers = k-tiles in flight: k-tile T -> buffer T % kNB.  The full body has room for two; the half-tile body (no late
  // token unit) for THREE: at 128 rows per group (configs[3] at T = 1024) every tile streams its weight tile from memory, and
  // with two k-tiles in flight a CU's stream was bound by the round trip - 4.25 TB/s of weights whatever the body computed
  // (the half body took 63 us per tile against 72 for a full one with twice the MFMAs)
  constexpr int kNB = kHalf ? 3 : 2;
  constexpr int kBOffL = kHalf ? kHalfBOff : kBOff;
  constexpr int kXsOffL = kHalf ? kHalfXsOff : kXsOff;
  // tile-local token slot (strip * 64 + 16 j + r) -> row of the tile
  auto row_of_slot = [](int slot) { return kHalf ? (slot >> 6) * 32 + (slot & 31) : slot; };

  // ---- DMA roles: two 8-row pieces of every unit per wave --------------------------------------------------
  // piece pc = wave * 2 + q covers unit rows pc*8 .. +8; lane -> row lane/8, LDS position lane%8 <- source
  // chunk (lane%8) ^ (lane/8).  Unit rows: weights  ur -> n = (ur/64)*128 + ur%64 (+64 for U3);
  // tokens ur -> slot (ur/32)*64 + ur%32 (+32 for U2).
  const int p_row = lane >> 3, p_chunk = (lane & 7) ^ (lane >> 3);
  const int inter = a.N >> 1;      // kAct only
  const int col0 = n0 >> 1;        // kAct only: first activation column of the tile
  const uint8_t* wsrc = a.w + (static_cast<long>(e) * a.N + (kAct ? 0 : n0)) * K;
  const unsigned w_bytes = static_cast<unsigned>(kAct ? a.N : kBN) * static_cast<unsigned>(K);
  const int late_w = 64 * K;  // byte distance of the U3 rows from the U0 rows
  unsigned w_voff[2];
#pragma unroll
  for (int q = 0; q < 2; ++q) {
    const int ur = (wave * 2 + q) * 8 + p_row;
    const int wrow = kAct ? (ur >> 6) * inter + col0 + (ur & 63) : (ur >> 6) * 128 + (ur & 63);
    w_voff[q] = static_cast<unsigned>(wrow) * static_cast<unsigned>(K) + p_chunk * 16;
  }
  // k-tile T -> buffer T % kNB (`par`, a compile-time value).  Past the end: empty descriptors (nothing fetched,
  // zeros written, still counted by vmcnt).  One piece (q = 0, 1) per call.
  auto dma_w = [&](int T, bool on, auto par, auto late, int q) {
    constexpr int kP = decltype(par)::value, kLate = decltype(late)::value;
    const int koff = T * kBK;
    const auto rw = make_rsrc(wsrc, on ? w_bytes : 0u);
    // K % 128 == 64 (per-tensor scales only; kKTail): the chunks past K get an out-of-range offset and land as zeros.
    // Its own instantiation: the per-lane compare + select per DMA piece is VALU work in the load sections, which
    // loses every issue slot to the MMA wave of its SIMD (round 4 paid it on every per-tensor call: matrix pipe busy
    // 0.85 -> 0.72)
    const bool k_ok = !kKTail || koff + p_chunk * 16 < K;
    uint8_t* base = s_mem + kAOff + (kNB * kLate + kP) * kUnit;
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rw, (lds_void*)(base + (wave * 2 + q) * 1024), 16,
                                             k_ok ? w_voff[q] : 0xffffff00u, koff + (kLate ? late_w : 0), 0, 0);
  };
  // the weight pieces of k-tile 0 (two buffers: and the early ones of k-tile 1) do not depend on the token rows: they go out
  // before the row-index loads
  dma_w(0, true, IntC<0>{}, IntC<0>{}, 0);
  dma_w(0, true, IntC<0>{}, IntC<0>{}, 1);
  dma_w(0, true, IntC<0>{}, IntC<1>{}, 0);
  dma_w(0, true, IntC<0>{}, IntC<1>{}, 1);
  if constexpr (kNB == 2) {
    dma_w(1, 1 < KB, IntC<1>{}, IntC<0>{}, 0);
    dma_w(1, 1 < KB, IntC<1>{}, IntC<0>{}, 1);
  }

  unsigned x_voff[2][2];
#pragma unroll
  for (int q = 0; q < 2; ++q) {
    const int ur = (wave * 2 + q) * 8 + p_row;
#pragma unroll
    for (int u = 0; u < (kHalf ? 1 : 2); ++u) {
      // unit row ur of the early (u = 0) / late (u = 1) token unit = tile slot (ur / 32) * 64 + ur % 32 + 32 u
      const int slot = mt0 + (kHalf ? ur : (ur >> 5) * 64 + (ur & 31) + u * 32);
      const int sc = slot < m_cnt ? slot : m_cnt - 1;
      const int xrow = a.row_index ? a.row_index[m0 + sc] : m0 + sc;
      x_voff[u][q] = static_cast<unsigned>(xrow) * static_cast<unsigned>(K) + p_chunk * 16;
    }
  }
  unsigned xs_voff = 0;
  if constexpr (kHasXs) {
    // LDS position `lane` of the strip's 256 B holds the scale of token (lane & 3) * 16 + (lane >> 2): the four scales a lane
    // multiplies with - token blocks 0-3, row r16 - are 16 contiguous bytes, ONE ds_read_b128 at an immediate offset (rounds
    // 2-5 kept them in token order, 64 B apart: two ds_read2_b32 and, for the second buffer, a VALU address add in a load
    // section).  (half tile: the positions of token blocks 2-3 hold scales nobody reads)
    const int slot = mt0 + row_of_slot((wave & 3) * 64 + (lane & 3) * 16 + (lane >> 2));
    const int sc = slot < m_cnt ? slot : m_cnt - 1;
    const long col0 = a.col_base ? static_cast<long>(as_const(a.col_base)[e]) * a.tile_m : 0;
    const long term = a.col_base ? col0 + sc : static_cast<long>(a.row_index ? a.row_index[m0 + sc] : m0 + sc);
    xs_voff = static_cast<unsigned>(term * a.xs_row_stride * 4);
  }
  if constexpr (kExt) {
    // waves 4-7 fetch no strip scales: their scale slot carries the ride-along rows (waves 4 / 5: 8 rows each, the DMA
    // role of a token piece) and the rows' scales (wave 6, lanes 0-15); rows past ext_cnt repeat the last one (not stored)
    if (wave == 4 || wave == 5) {
      const int er = (wave - 4) * 8 + p_row;
      const int tok = ext0 + (er < ext_cnt ? er : ext_cnt - 1);
      const int xrow = a.row_index ? a.row_index[m0 + tok] : m0 + tok;
      xs_voff = static_cast<unsigned>(xrow) * static_cast<unsigned>(K) + p_chunk * 16;
    } else if (kHasXs && wave == 6) {
      const int er = lane >> 2;  // position 4 r16 = byte 16 r16 of the 256 B: where xs_rd + ext_sx points
      const int tok = ext0 + (er < ext_cnt ? er : ext_cnt - 1);
      const long col0 = a.col_base ? static_cast<long>(as_const(a.col_base)[e]) * a.tile_m : 0;
      const long term = a.col_base ? col0 + tok : static_cast<long>(a.row_index ? a.row_index[m0 + tok] : m0 + tok);
      xs_voff = static_cast<unsigned>(term * a.xs_row_stride * 4);
    }
  }
  const int xs_kb_bytes = static_cast<int>(a.xs_kb_stride * 4);
  auto dma_x = [&](int T, bool on, auto par, auto late, int q) {
    constexpr int kP = decltype(par)::value, kLate = decltype(late)::value;
    const int koff = T * kBK;
    const auto rx = make_rsrc(a.x, on ? a.x_bytes : 0u);
    const bool k_ok = !kKTail || koff + p_chunk * 16 < K;
    uint8_t* base = s_mem + kBOffL + (kNB * kLate + kP) * kUnit;
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rx, (lds_void*)(base + (wave * 2 + q) * 1024), 16,
                                             k_ok ? x_voff[kLate][q] : 0xffffff00u, koff, 0, 0);
  };
  const unsigned xs_nrec = __builtin_amdgcn_readfirstlane(kHasXs && (wave < 4 || (kExt && wave == 6)) ? 0xffffffffu : 0u);
  auto dma_xs = [&](int T, bool on, auto par) {
    constexpr int kP = decltype(par)::value;
    if constexpr (kExt) {
      if (wave == 4 || wave == 5) {  // (wave-uniform: a scalar branch) this wave's 8 ride-along rows of k-tile T
        const auto rx = make_rsrc(a.x, on ? a.x_bytes : 0u);
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rx, (lds_void*)(s_mem + kExtOff + kP * 2048 + (wave - 4) * 1024), 16, xs_voff,
                                                 T * kBK, 0, 0);
      } else {
        const auto rs = make_rsrc(kHasXs ? static_cast<const void*>(a.xs) : static_cast<const void*>(a.x), on ? xs_nrec : 0u);
        uint8_t* dst = s_mem + (wave < 4 ? kXsOff + kP * 1024 + wave * 256 : (wave == 6 ? kExtXsOff + kP * 256 : kXsOff + 2048));
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lds_void*)dst, 4, xs_voff, T * xs_kb_bytes, 0, 0);
      }
    } else if constexpr (kHasXs) {
      // all eight waves issue it (the vmcnt arithmetic needs equal counts); waves 4-7 fetch nothing
      // (the wave's share of the select is hoisted out of the loop: round 4 evaluated `on && wave < 4` through
      // v_cndmask + v_readfirstlane per k-tile - VALU work in a load section, which loses every issue slot to the MMA
      // wave of its SIMD; what is left is a scalar select on `on`)
      const auto rs = make_rsrc(a.xs, on ? xs_nrec : 0u);
      uint8_t* dst = s_mem + kXsOffL + (wave < 4 ? kP * 1024 + wave * 256 : kNB * 1024);
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lds_void*)dst, 4, xs_voff, T * xs_kb_bytes, 0, 0);
    }
  };

  const cint_ptr ws_row = as_const(reinterpret_cast<const int*>(a.ws)) + static_cast<long>(e) * a.ws_group_stride +
                          ((kAct ? wn * inter + col0 : n0 + wn * 128) >> 7) * a.ws_ntile_stride;

  f32x4 tot[8][4];
#pragma unroll
  for (int i = 0; i < 8; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) tot[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  // operand read offsets inside a unit (swizzled): row r, chunk c -> r*128 + ((c ^ (r & 7)) << 4); the rows of a
  // wave's blocks are 16 apart, so (r & 7) does not depend on the block and block offsets are immediates - and so are the
  // unit and the buffer (the interleaved image above): one base register per operand side and chunk half.  (Any address
  // arithmetic at a read would be VALU work in a load section, which loses every issue slot to the MMA wave of its SIMD.)
  int a_off[2], b_off[2];
#pragma unroll
  for (int c = 0; c < 2; ++c) {
    a_off[c] = kAOff + (wn * 64 + r16) * kBK + (((g4 + 4 * c) ^ (r16 & 7)) << 4);
    b_off[c] = kBOffL + (wm * 32 + r16) * kBK + (((g4 + 4 * c) ^ (r16 & 7)) << 4);
    asm volatile("" : "+v"(a_off[c]), "+v"(b_off[c]));  // opaque: hipcc would otherwise fold the immediates back into copies
  }
  // (three buffers: the late weight unit's third buffer ends past the 16-bit offset field - its own base, the half body has the registers)
  int a_off_late[2] = {a_off[0], a_off[1]};
  if constexpr (kNB == 3) {
#pragma unroll
    for (int c = 0; c < 2; ++c) {
      a_off_late[c] = a_off[c] + kNB * kUnit;
      asm volatile("" : "+v"(a_off_late[c]));
    }
  }
  int xs_rd = kXsOffL + wm * 256 + r16 * 16;  // this lane's four token scales (token blocks 0-3) in scale buffer 0 (opaque base: as above)
  asm volatile("" : "+v"(xs_rd));
  // ride-along block: row block wm of either weight unit, the 16 rows' operand bytes and scales (read bases: opaque as above)
  f32x4 tot_ext[2] = {f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}};  // rows 16 wm .. / 64 + 16 wm .. of the wave's half
  // their read addresses are derived from a_off[0] / xs_rd INSIDE the MMA section (wave-uniform distances; chunk half c = 1
  // is the address with bit 6 flipped): a handful of VALU instructions where the wave owns the issue slots, no register held
  const int ext_sa = wm * 2048;                           // row block wm of a weight unit
  const int ext_sb = kExtOff - kAOff - wn * 64 * kBK;     // a_off[0] -> row r16 of the ride-along buffer, same chunk slot
  const int ext_sx = kExtXsOff - kXsOff - wm * 256;       // xs_rd -> the ride-along row's scale
  auto read_a = [&](auto par, int late, u32x4 (&af)[4][2]) {  // late: 0 = U0, 1 = U3 of the buffer
    constexpr int kP = decltype(par)::value;
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int c = 0; c < 2; ++c)
        af[i][c] = kNB == 3 && late ? *reinterpret_cast<const u32x4*>(s_mem + a_off_late[c] + kP * kUnit + i * 2048)
                                    : *reinterpret_cast<const u32x4*>(s_mem + a_off[c] + (kNB * late + kP) * kUnit + i * 2048);
  };
  auto read_b = [&](auto par, int late, u32x4 (&bf)[2][2]) {
    constexpr int kP = decltype(par)::value;
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int c = 0; c < 2; ++c)
        bf[j][c] = *reinterpret_cast<const u32x4*>(s_mem + b_off[c] + (kNB * late + kP) * kUnit + j * 2048);
  };

  // One section: 4 row blocks x all 4 token blocks (16 MFMAs), written as the software pipeline it has to be:
  // the rescale of block n follows the MFMA of block n + 2 (its result is ready by then without wait states: with
  // one MFMA in between hipcc pads 7-11 idle cycles per MFMA and the matrix pipe is busy 52-55 % of the time),
  // `hook(n)` may issue DMA pieces behind them, and the order is pinned.  Blockwise form: fp32 partial of the
  // 128-k block, rescaled into the running sum (reference kernels.cuh:473-476); the rescale of the LAST TWO blocks
  // is the caller's (`tail`), behind the barrier that ends the section - otherwise that barrier would wait for the
  // last MFMA's result and the other wave of the SIMD would start its section ~40 cycles late.
  // Cfg::kCarry: the last two blocks' rescale is not done in the load section that follows but rides under the first two
  // MFMAs of the NEXT section (`pend`; `f` still holds the scale products of the k-tile they belong to): the load sections carry no
  // FMA, a section's head needs no hazard padding (its first FMAs read results that are a whole load section old), and
  // the new k-tile's four scale products (`pre`) sit behind MFMA 0 instead of in front of it.
  constexpr int kN = 4 * kJ;       // MFMAs per section (kExt: section X has two more - the ride-along block x row blocks wm, 4 + wm)
  constexpr int kD = Cfg::kDist;   // the rescale of block n follows the MFMA of block n + kD
  f32x4 pend[kD];
  float f_ext = 0.f;  // kExt: the ride-along block's scale product (formed in section X, used by the folds under section Y's head)
#pragma unroll
  for (int t = 0; t < kD; ++t) pend[t] = f32x4{0.f, 0.f, 0.f, 0.f};
  // block b of the section whose row half starts at row block sec_i0: its partial folded into the running sum
  // (b < kN: row block b / kJ, token block b % kJ; b >= kN: the ride-along block on row block wm of unit b - kN)
  auto fold = [&](int sec_i0, int b, const f32x4& p, const float (&ff)[4], float ffe) {
    if (b < kN) {
      const int pi = b / kJ, pj = b % kJ;
#pragma unroll
      for (int r = 0; r < 4; ++r) tot[sec_i0 + pi][pj][r] = fmaf(p[r], ff[pj], tot[sec_i0 + pi][pj][r]);
      // kExt: the scale slot's DMA is a wave-uniform BRANCH (dma_xs), the k-loop is several basic blocks, and hipcc sinks
      // the folds - pure arithmetic whose results are next read a k-tile later - down to the last block: 64 partials stay
      // live and the schedule is gone.  An empty asm that "reads and writes" the sum keeps every fold where it is written.
      if constexpr (kExt) asm volatile("" : "+v"(tot[sec_i0 + pi][pj]));
    } else {
#pragma unroll
      for (int r = 0; r < 4; ++r) tot_ext[b - kN][r] = fmaf(p[r], ffe, tot_ext[b - kN][r]);
      asm volatile("" : "+v"(tot_ext[b - kN]));
    }
  };
  auto blocks_of = [](int i0) { return kN + (kExt && i0 == 0 ? 2 : 0); };  // section X carries the ride-along MFMAs
  auto section = [&](auto i0c, auto par, const u32x4 (&af)[4][2], const u32x4 (&be)[2][2], const u32x4 (&bl)[2][2],
                     float (&f)[4], float wsk, auto&& pre, auto&& hook, auto&& early_leave) {
    constexpr int i0 = decltype(i0c)::value, kP = decltype(par)::value;
    constexpr int kNs = blocks_of(i0), kNp = blocks_of(4 - i0);  // blocks of this section / of the one before it
    u32x4 a_ext[2][2], b_ext[2];  // kExt: row block wm of U0 / U3 and the ride-along rows, read just in time (see above)
    float xs_ext = 1.f;
    f32x4 pv[kD];  // pv[0] = the newest partial
#pragma unroll
    for (int t = 0; t < kD; ++t) pv[t] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int n = 0; n < kNs; ++n) {
      if constexpr (kExt && i0 == 0) {
        if (n == 8) {  // row blocks 0-1 are finished: their registers take the ride-along rows and row block wm of U0
#pragma unroll
          for (int c = 0; c < 2; ++c) {
            b_ext[c] = *reinterpret_cast<const u32x4*>(s_mem + ((a_off[0] + ext_sb) ^ (c * 64)) + kP * 2048);
            a_ext[0][c] = *reinterpret_cast<const u32x4*>(s_mem + ((a_off[0] + ext_sa) ^ (c * 64)) + kP * kUnit);
          }
          if constexpr (kHasXs) xs_ext = *reinterpret_cast<const float*>(s_mem + xs_rd + ext_sx + kP * 256);
          __builtin_amdgcn_sched_barrier(0);
        }
        if (n == 12) {  // row block 2 is finished: row block wm of U3
#pragma unroll
          for (int c = 0; c < 2; ++c)
            a_ext[1][c] = *reinterpret_cast<const u32x4*>(s_mem + ((a_off[0] + ext_sa) ^ (c * 64)) + (2 + kP) * kUnit);
          __builtin_amdgcn_sched_barrier(0);
        }
        if (n == kN) {
          __builtin_amdgcn_s_waitcnt(0xC07F);  // lgkmcnt(0): the just-in-time reads (nothing else is outstanding there)
          if constexpr (kHasXs) f_ext = wsk * xs_ext;
          __builtin_amdgcn_sched_barrier(0);
        }
      }
      const int i = n < kN ? n / kJ : 0, j = n < kN ? n % kJ : 0;
      const u32x4(&bf)[2] = n >= kN ? b_ext : (j < 2 ? be[j] : bl[j - 2]);
      const u32x4(&ar)[2] = n >= kN ? a_ext[n >= kN ? n - kN : 0] : af[i];
      const i32x8 av = {static_cast<int>(ar[0][0]), static_cast<int>(ar[0][1]), static_cast<int>(ar[0][2]),
                        static_cast<int>(ar[0][3]), static_cast<int>(ar[1][0]), static_cast<int>(ar[1][1]),
                        static_cast<int>(ar[1][2]), static_cast<int>(ar[1][3])};
      const i32x8 bv = {static_cast<int>(bf[0][0]), static_cast<int>(bf[0][1]), static_cast<int>(bf[0][2]),
                        static_cast<int>(bf[0][3]), static_cast<int>(bf[1][0]), static_cast<int>(bf[1][1]),
                        static_cast<int>(bf[1][2]), static_cast<int>(bf[1][3])};
      if constexpr (kHasXs) {
        const f32x4 part = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(av, bv, f32x4{0.f, 0.f, 0.f, 0.f}, 0, 0, 0, 0,
                                                                            0, 0);
        __builtin_amdgcn_sched_barrier(0);  // MFMA n first, then the rescale of block n - kD
        if (n >= kD) {
          fold(i0, n - kD, pv[kD - 1], f, f_ext);
        } else if constexpr (Cfg::kCarry) {
          // block kNp - kD + n of the PREVIOUS section (the other row half: 4 - i0), under the scales of its k-tile
          // (`f` still holds that k-tile's products: the new k-tile's - `pre` - are formed behind the LAST carried fold, so
          // the two sets never live side by side: four registers less at the kernel's register peak than rounds 5's `fpend`)
          fold(4 - i0, kNp - kD + n, pend[n], f, f_ext);
          if (n == kD - 1) pre();
        }
#pragma unroll
        for (int t = kD - 1; t > 0; --t) pv[t] = pv[t - 1];
        pv[0] = part;
      } else {
        // per-tensor: one scale per group: accumulate straight into the running sum, scale once in the epilogue
        // (the reference scales every k-tile: same value up to fp32 rounding)
        if (n < kN) {
          tot[i0 + i][j] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(av, bv, tot[i0 + i][j], 0, 0, 0, 0, 0, 0);
          // (kExt: the scale slot's DMA is a wave-uniform branch and the loop several basic blocks - keep every MFMA where it
          //  is written, like the folds of the blockwise form)
          if constexpr (kExt) asm volatile("" : "+v"(tot[i0 + i][j]));
        } else {
          tot_ext[n >= kN ? n - kN : 0] =
              __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(av, bv, tot_ext[n >= kN ? n - kN : 0], 0, 0, 0, 0, 0, 0);
          asm volatile("" : "+v"(tot_ext[n >= kN ? n - kN : 0]));
        }
      }
      __builtin_amdgcn_sched_barrier(0);
      hook(n);
      __builtin_amdgcn_sched_barrier(0);
      if (n == kNs - 1 - Cfg::kEarly) early_leave();
    }
    if constexpr (kHasXs) {  // the last kD partials: block kNs - kD + t <- pv[kD - 1 - t]
#pragma unroll
      for (int t = 0; t < kD; ++t) pend[t] = pv[kD - 1 - t];
    }
  };
  // without Cfg::kCarry the pending blocks are folded right behind the barrier that ends the section (round 2-4 form)
  auto apply_tail = [&](int i0, const float (&f)[4]) {
    if constexpr (kHasXs && !Cfg::kCarry) {
#pragma unroll
      for (int t = 0; t < kD; ++t) fold(i0, blocks_of(i0) - kD + t, pend[t], f, f_ext);
    }
  };

  // ---- DMA schedule (see the Cfg structs): slots of this body's pieces and the two waits that follow from them ------
  auto sx = [](int i) { return kHalf ? Cfg::hx(i) : Cfg::fx(i); };
  auto sy = [](int i) { return kHalf ? Cfg::hy(i) : Cfg::fy(i); };
  auto y_exists = [](int i) { return i < 4 || (i < 6 ? !kHalf : kXsSlot); };
  constexpr int kPiecesY = (kHalf ? 4 : 6) + (kXsSlot ? 1 : 0);
  int n_xload = 0, n_yload = 0;
#pragma unroll
  for (int i = 0; i < 2; ++i) n_xload += sx(i) == 0;
#pragma unroll
  for (int i = 0; i < 7; ++i) n_yload += y_exists(i) && sy(i) == 0;
  // s_waitcnt immediates: vmcnt = n (split 4 + 2 bits), expcnt untouched, lgkmcnt 0 (or untouched: | 0x0F00)
  // (every further buffer puts one more k-tile's pieces - the Y half's and the X half's two - between a piece and its wait: the
  //  stream a wave issues is [U0 U1 (U2) scales U3] per k-tile, and loads retire in order)
  const int fly_x = kPiecesY + n_xload + (kNB - 2) * (kPiecesY + 2), fly_y = 2 + n_yload + (kNB - 2) * (kPiecesY + 2);  // compile-time values after inlining

  // ---- development: s_memtime log of the section boundaries (Cfg::kProf) -------------------------------------------
  // A = first instruction behind the barrier that opens an MMA section, B = behind its last MFMA, C = behind the barrier
  // that closes it, D = end of the load section's issue (in front of its wait).  The stamps are read where the load
  // section's own lgkmcnt(0) has retired them; lane 4 e + {0, 1, 2, 3} of `plog` = (A, B, C of the section before, D of
  // this one) for the 16 sections from k-tile 4 on.
  unsigned long long ts_a = 0, ts_b = 0, ts_c = 0, ts_d = 0;
  int plog = 0, sec = 0;
  auto stamp = [&](unsigned long long& t) {
    if constexpr (Cfg::kProf) asm volatile("s_memtime %0" : "=s"(t));
  };
  auto log4 = [&]() {
    if constexpr (Cfg::kProf) {
      __builtin_amdgcn_sched_barrier(0);
      const int e = sec - 8;
      const int d = lane - e * 4;  // (e outside 0 .. 15: no lane matches)
      plog = d == 0 ? static_cast<int>(ts_a) : plog;
      plog = d == 1 ? static_cast<int>(ts_b) : plog;
      plog = d == 2 ? static_cast<int>(ts_c) : plog;
      plog = d == 3 ? static_cast<int>(ts_d) : plog;
      ++sec;
      __builtin_amdgcn_sched_barrier(0);
    }
  };

  auto wait_fly = [&](int n) {  // vmcnt(n), lgkmcnt(0): n is a compile-time value after inlining
    switch (n) {
#define HPC_P8_WAIT(N) case N: __builtin_amdgcn_s_waitcnt(0x0070 | ((N) & 15) | (((N) >> 4) << 14)); break;
      HPC_P8_WAIT(2) HPC_P8_WAIT(3) HPC_P8_WAIT(4) HPC_P8_WAIT(5) HPC_P8_WAIT(6) HPC_P8_WAIT(7) HPC_P8_WAIT(8) HPC_P8_WAIT(9)
      HPC_P8_WAIT(10) HPC_P8_WAIT(11) HPC_P8_WAIT(12) HPC_P8_WAIT(13) HPC_P8_WAIT(14) HPC_P8_WAIT(15) HPC_P8_WAIT(16) HPC_P8_WAIT(17)
#undef HPC_P8_WAIT
      default: __builtin_amdgcn_s_waitcnt(0x0070); break;  // vmcnt(0)
    }
  };
  auto enter_mma = [&](int fly) {  // end of a load section
    __builtin_amdgcn_sched_barrier(0);
    stamp(ts_d);
    // this wave's pieces of the unit(s) read behind the NEXT barrier have landed; its operand reads have retired
    if constexpr (kNoDma)
      __builtin_amdgcn_s_waitcnt(0xC07F);
    else
      wait_fly(fly);
    log4();
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
    if constexpr (Cfg::kPrio) __builtin_amdgcn_s_setprio(1);
    stamp(ts_a);
  };
  auto leave_mma = [&]() {  // (Cfg::kEarly > 0: called in front of the section's last kEarly MFMAs)
    if constexpr (Cfg::kPrio && Cfg::kEarly == 0) __builtin_amdgcn_s_setprio(0);
    __builtin_amdgcn_sched_barrier(0);
    stamp(ts_b);
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
    stamp(ts_c);
  };

  // ---- prologue: the token pieces of k-tiles 0 and 1 ---------------------------------------------------------
  // Steady-state issue order per k-tile T (product schedule): load section Y: U0(T+2), U1(T+2); MFMAs of Y: U2(T+2),
  // scales(T+2); load section X of T+1: U3(T+2).  Every target's last reads (by either group) retired before the barrier
  // that precedes the issue.  Waits: end of load section X leaves fly_x pieces in flight - U3(T), read behind
  // the next barrier but one, has landed; end of load section Y leaves fly_y - all of k-tile T+1 up to its scales
  // has landed.  Here the weight pieces went first, so the first wait leaves k-tile 1's token pieces.
  if constexpr (kNB == 2) {
    dma_x(0, true, IntC<0>{}, IntC<0>{}, 0);
    dma_x(0, true, IntC<0>{}, IntC<0>{}, 1);
    if constexpr (!kHalf) {
      dma_x(0, true, IntC<0>{}, IntC<1>{}, 0);
      dma_x(0, true, IntC<0>{}, IntC<1>{}, 1);
    }
    dma_xs(0, true, IntC<0>{});
    dma_x(1, 1 < KB, IntC<1>{}, IntC<0>{}, 0);
    dma_x(1, 1 < KB, IntC<1>{}, IntC<0>{}, 1);
    if constexpr (!kHalf) {
      dma_x(1, 1 < KB, IntC<1>{}, IntC<1>{}, 0);
      dma_x(1, 1 < KB, IntC<1>{}, IntC<1>{}, 1);
    }
    dma_xs(1, 1 < KB, IntC<1>{});
    constexpr int kFly0 = (kXsSlot ? 5 : 4) - (kHalf ? 2 : 0);
    __builtin_amdgcn_s_waitcnt(0x0F70 | kFly0);  // k-tile 0 (and the weight pieces of k-tile 1) have landed
  } else {
    // three buffers (half body): the rest of k-tiles 0 .. 2 in the loop's own order - [U0 U1 scales U3] per k-tile; U3 of
    // k-tile 2 is load section X's of k-tile 0 - so that the loop's waits hold from the first k-tile on (k-tile 0's weight
    // pieces are older than the loop would have made them: its waits only get more conservative)
    dma_x(0, true, IntC<0>{}, IntC<0>{}, 0);
    dma_x(0, true, IntC<0>{}, IntC<0>{}, 1);
    dma_xs(0, true, IntC<0>{});
    dma_w(1, 1 < KB, IntC<1>{}, IntC<0>{}, 0);
    dma_w(1, 1 < KB, IntC<1>{}, IntC<0>{}, 1);
    dma_x(1, 1 < KB, IntC<1>{}, IntC<0>{}, 0);
    dma_x(1, 1 < KB, IntC<1>{}, IntC<0>{}, 1);
    dma_xs(1, 1 < KB, IntC<1>{});
    dma_w(1, 1 < KB, IntC<1>{}, IntC<1>{}, 0);
    dma_w(1, 1 < KB, IntC<1>{}, IntC<1>{}, 1);
    dma_w(2, 2 < KB, IntC<2>{}, IntC<0>{}, 0);
    dma_w(2, 2 < KB, IntC<2>{}, IntC<0>{}, 1);
    dma_x(2, 2 < KB, IntC<2>{}, IntC<0>{}, 0);
    dma_x(2, 2 < KB, IntC<2>{}, IntC<0>{}, 1);
    dma_xs(2, 2 < KB, IntC<2>{});
    constexpr int kFly0 = 2 * (4 + (kXsSlot ? 1 : 0)) + 2;  // everything younger than k-tile 0's scales
    __builtin_amdgcn_s_waitcnt(0x0F70 | kFly0);  // k-tile 0 has landed
  }
  __builtin_amdgcn_s_barrier();
  if (wn == 1) __builtin_amdgcn_s_barrier();  // the second group runs one barrier behind from here on
  __builtin_amdgcn_sched_barrier(0);

  u32x4 a_frag[4][2], b_early[2][2], b_late[2][2];
  float f[4] = {1.f, 1.f, 1.f, 1.f};
  auto k_tile = [&](int T, auto par) {
    constexpr int kP = decltype(par)::value;
    // (kNB buffers: "T + 1" / "T + 2" of the two-buffer schedule are T + kNB - 1 / T + kNB)
    const bool on1 = T + kNB - 1 < KB, on2 = T + kNB < KB;
    auto x_piece = [&](int i) { dma_w(T + kNB - 1, on1, IntC<(kP + kNB - 1) % kNB>{}, IntC<1>{}, i); };  // U3(T + 1), piece q = i
    auto y_piece = [&](int i) {  // U0(T + 2) q0 q1, U1(T + 2) q0 q1, U2(T + 2) q0 q1, scales(T + 2)
      if (i < 2) {
        dma_w(T + kNB, on2, IntC<kP>{}, IntC<0>{}, i);
      } else if (i < 4) {
        dma_x(T + kNB, on2, IntC<kP>{}, IntC<0>{}, i - 2);
      } else if (i < 6) {
        if constexpr (!kHalf) dma_x(T + kNB, on2, IntC<kP>{}, IntC<1>{}, i - 4);
      } else {
        dma_xs(T + kNB, on2, IntC<kP>{});
      }
    };
    auto issue_x = [&](int slot) {
      if constexpr (!kNoDma) {
#pragma unroll
        for (int i = 0; i < 2; ++i)
          if (sx(i) == slot) x_piece(i);
      }
    };
    auto issue_y = [&](int slot) {
      if constexpr (!kNoDma) {
#pragma unroll
        for (int i = 0; i < 7; ++i)
          if (y_exists(i) && sy(i) == slot) y_piece(i);
      }
    };
    // ---- section X: rows 0-63 of the wave's half ------------------------------------------------------------
    if constexpr (Cfg::kDmaFirst) {
      issue_x(0);
      __builtin_amdgcn_sched_barrier(0);
    }
    read_b(par, 0, b_early);
    if constexpr (!kHalf) read_b(par, 1, b_late);
    read_a(par, 0, a_frag);
    float xsv[4] = {1.f, 1.f, 1.f, 1.f}, wsk = 1.f;
    if constexpr (kHasXs) {
      wsk = __int_as_float(ws_row[T * a.ws_kb_stride]);
      const f32x4 x4 = *reinterpret_cast<const f32x4*>(s_mem + xs_rd + kP * 1024);
#pragma unroll
      for (int j = 0; j < kJ; ++j) xsv[j] = x4[j];
    }
    if constexpr (!Cfg::kDmaFirst) issue_x(0);
    enter_mma(fly_x);
    auto make_f = [&]() {
#pragma unroll
      for (int j = 0; j < 4; ++j) f[j] = wsk * xsv[j];
    };
    if constexpr (!(Cfg::kCarry && kHasXs)) make_f();
    section(IntC<0>{}, par, a_frag, b_early, b_late, f, wsk, make_f, [&](int n) { issue_x(n + 1); }, leave_mma);
    if constexpr (Cfg::kPrio && Cfg::kEarly > 0) __builtin_amdgcn_s_setprio(0);
    apply_tail(0, f);
    // ---- section Y: rows 64-127 ----------------------------------------------------------------------------
    if constexpr (Cfg::kDmaFirst) {
      issue_y(0);
      __builtin_amdgcn_sched_barrier(0);
    }
    read_a(par, 1, a_frag);
    if constexpr (!Cfg::kDmaFirst) issue_y(0);
    enter_mma(fly_y);
    section(IntC<4>{}, par, a_frag, b_early, b_late, f, wsk, [] {}, [&](int n) { issue_y(n + 1); }, leave_mma);
    if constexpr (Cfg::kPrio && Cfg::kEarly > 0) __builtin_amdgcn_s_setprio(0);
    apply_tail(4, f);
  };
  for (int kb = 0; kb < KB; kb += kNB) {
    k_tile(kb, IntC<0>{});
    if (kb + 1 < KB) k_tile(kb + 1, IntC<1>{});
    if constexpr (kNB == 3)
      if (kb + 2 < KB) k_tile(kb + 2, IntC<2>{});
  }
  if constexpr (Cfg::kCarry && kHasXs) {  // the last section's pending blocks (row half 4 .. 7)
#pragma unroll
    for (int t = 0; t < kD; ++t) fold(4, kN - kD + t, pend[t], f, f_ext);
  }
  if constexpr (Cfg::kProf) {
    if (a.prof && blockIdx.x < 16)
      reinterpret_cast<int*>(a.prof)[(blockIdx.x * 8 + wave) * 64 + lane] = plog;
  }
  if (wn == 0) __builtin_amdgcn_s_barrier();  // even out the barrier count
  __builtin_amdgcn_s_waitcnt(0x0F70);         // drain the (empty) tail DMAs before the workgroup's LDS is released

  if constexpr (!kHasXs) {
    const float gs = __int_as_float(ws_row[0]);  // per-tensor form: strides are zero, one scale per group
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j) tot[i][j] *= gs;
    if constexpr (kExt) {
      tot_ext[0] *= gs;
      tot_ext[1] *= gs;
    }
  }
  if constexpr (kAct) {
    // ---- fused activation epilogue ------------------------------------------------------------------------------
    // Wave (group 0, strip s) holds the gate values and wave (group 1, strip s) the up values of the same 128
    // columns x 64 tokens, in the same lane <-> (column, token) mapping.  They swap halves through LDS (a
    // lane-linear 8-byte slot per block, bf16-rounded like the GEMM output the separate kernel would read): group
    // 0 finishes token blocks 0-1, group 1 token blocks 2-3.  a = silu(g) * u in fp32, abs-max over the tile's 128
    // columns per token (32 values in the lane, 4 lanes per token), scale = amax / 448, q = e4m3(a / (scale + 1e-8)) -
    // the arithmetic of act_mul_blockwise_quant_kernel (reference src/activation/activation.cu:282-355), bit for bit.
    __builtin_amdgcn_s_barrier();  // every wave is past its last LDS read and its last (empty) DMA has landed
    uint32_t* xch = reinterpret_cast<uint32_t*>(s_mem) + (wm * 16 * 64 + lane) * 2;  // [strip][i * 2 + jj][lane] of 8 B
    // (half tile: only token blocks 0-1 exist - group 1 sends its up values, group 0 finishes, nothing else)
    auto send = [&](auto j0c) {  // the two token blocks the OTHER group finishes
      constexpr int j0 = decltype(j0c)::value;
#pragma unroll
      for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int jj = 0; jj < 2; ++jj)
          *reinterpret_cast<u32x2*>(xch + (i * 2 + jj) * 128) =
              u32x2{pack_bf16x2(tot[i][j0 + jj][0], tot[i][j0 + jj][1]), pack_bf16x2(tot[i][j0 + jj][2], tot[i][j0 + jj][3])};
    };
    auto finish = [&](auto j0c, auto mine_is_gate) {
      constexpr int j0 = decltype(j0c)::value;
      constexpr bool kGate = decltype(mine_is_gate)::value;
#pragma unroll
      for (int jj = 0; jj < 2; ++jj) {
        const int j = j0 + jj;
        const int slot = mt0 + row_of_slot(wm * 64 + j * 16 + r16);
        if constexpr (!kHasXs) {
          // per-tensor API: a = silu(g) * u (bf16-rounded factors and product when use_bf16_mul), times one scale -
          // the arithmetic of act_mul_quant_kernel (csrc/fuse_moe.hip), value for value
          const float sc = a.act_mul_scale[0];
          uint8_t* orow = a.act_out + static_cast<long>(m0 + slot) * inter + col0 + g4 * 4;
#pragma unroll
          for (int i = 0; i < 8; ++i) {
            const u32x2 ov = *reinterpret_cast<const u32x2*>(xch + (i * 2 + jj) * 128);
            const uint32_t m01 = pack_bf16x2(tot[i][j][0], tot[i][j][1]), m23 = pack_bf16x2(tot[i][j][2], tot[i][j][3]);
            const float mine[4] = {bf16lo_to_f32(m01), bf16hi_to_f32(m01), bf16lo_to_f32(m23), bf16hi_to_f32(m23)};
            const float oth[4] = {bf16lo_to_f32(ov[0]), bf16hi_to_f32(ov[0]), bf16lo_to_f32(ov[1]), bf16hi_to_f32(ov[1])};
            float v[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) {
              const float g = kGate ? mine[r] : oth[r], u = kGate ? oth[r] : mine[r];
              float sv = g / (1.0f + __expf(-g));
              if (a.use_bf16_mul)
                sv = bf16_to_f32(f32_to_bf16(bf16_to_f32(f32_to_bf16(sv)) * u));
              else
                sv *= u;
              v[r] = sv * sc;
            }
            if (slot < m_cnt) *reinterpret_cast<uint32_t*>(orow + i * 16) = quant_4xe4m3(v[0], v[1], v[2], v[3]);
          }
          continue;
        }
        float amax = 0.f;
#pragma unroll
        for (int i = 0; i < 8; ++i) {
          const u32x2 ov = *reinterpret_cast<const u32x2*>(xch + (i * 2 + jj) * 128);
          const uint32_t m01 = pack_bf16x2(tot[i][j][0], tot[i][j][1]), m23 = pack_bf16x2(tot[i][j][2], tot[i][j][3]);
          const float mine[4] = {bf16lo_to_f32(m01), bf16hi_to_f32(m01), bf16lo_to_f32(m23), bf16hi_to_f32(m23)};
          const float oth[4] = {bf16lo_to_f32(ov[0]), bf16hi_to_f32(ov[0]), bf16lo_to_f32(ov[1]), bf16hi_to_f32(ov[1])};
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const float g = kGate ? mine[r] : oth[r], u = kGate ? oth[r] : mine[r];
            const float v = g / (1.0f + __expf(-g)) * u;
            tot[i][j][r] = v;
            amax = fmaxf(amax, fabsf(v));
          }
        }
        // the token's other 96 columns sit in lanes r16 + 16, + 32, + 48
        amax = fmaxf(amax, __shfl_xor(amax, 16, 64));
        amax = fmaxf(amax, __shfl_xor(amax, 32, 64));
        const float scale = amax / 448.0f;
        const float inv = 1.0f / (scale + 1e-8f);
        if (slot < m_cnt) {
          uint8_t* orow = a.act_out + static_cast<long>(m0 + slot) * inter + col0 + g4 * 4;
#pragma unroll
          for (int i = 0; i < 8; ++i)
            *reinterpret_cast<uint32_t*>(orow + i * 16) =
                quant_4xe4m3(tot[i][j][0] * inv, tot[i][j][1] * inv, tot[i][j][2] * inv, tot[i][j][3] * inv);
          if (g4 == 0) a.act_scale[static_cast<long>(m0 + slot) * (inter >> 7) + (col0 >> 7)] = scale;
        }
      }
    };
    // group 0 sends its gate values of token blocks 2-3 into the first 32 KB, group 1 its up values of blocks 0-1
    // into the second; each then reads the other's region
    // ride-along block: the up wave hands its two blocks (columns 16 wm .. and 64 + 16 wm .. of the tile's 128) to the gate
    // wave of the same strip index through the (now idle) ride-along row buffers; the token's abs-max over the 128 columns
    // is folded over the four gate waves through the scale buffers - the tail body's epilogue (tail_finish), value for value
    uint32_t* ext_xch = reinterpret_cast<uint32_t*>(s_mem + kExtOff) + (wm * 2 * 64 + lane) * 2;  // [wm][unit][lane] of 8 B
    float* ext_red = reinterpret_cast<float*>(s_mem + kExtXsOff);                                 // [wm][16 tokens]
    if (wn == 0) {
      if constexpr (!kHalf) send(IntC<2>{});
    } else {
      xch += 4 * 16 * 64 * 2;
      send(IntC<0>{});
      if constexpr (kExt) {
#pragma unroll
        for (int u = 0; u < 2; ++u)
          *reinterpret_cast<u32x2*>(ext_xch + u * 128) =
              u32x2{pack_bf16x2(tot_ext[u][0], tot_ext[u][1]), pack_bf16x2(tot_ext[u][2], tot_ext[u][3])};
      }
    }
    __syncthreads();
    if constexpr (kExt && !kHasXs) {
      // per-tensor API: a = silu(g) * u (bf16-rounded factors and product when use_bf16_mul), times one scale - `finish` above,
      // value for value; no abs-max, so no second barrier
      if (wn == 0 && r16 < ext_cnt) {
        const float sc = a.act_mul_scale[0];
        uint8_t* orow = a.act_out + static_cast<long>(m0 + ext0 + r16) * inter + col0 + wm * 16 + g4 * 4;
#pragma unroll
        for (int u = 0; u < 2; ++u) {
          const u32x2 ov = *reinterpret_cast<const u32x2*>(ext_xch + u * 128);
          const uint32_t m01 = pack_bf16x2(tot_ext[u][0], tot_ext[u][1]), m23 = pack_bf16x2(tot_ext[u][2], tot_ext[u][3]);
          const float gv[4] = {bf16lo_to_f32(m01), bf16hi_to_f32(m01), bf16lo_to_f32(m23), bf16hi_to_f32(m23)};
          const float uv[4] = {bf16lo_to_f32(ov[0]), bf16hi_to_f32(ov[0]), bf16lo_to_f32(ov[1]), bf16hi_to_f32(ov[1])};
          float v[4];
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            float sv = gv[r] / (1.0f + __expf(-gv[r]));
            if (a.use_bf16_mul)
              sv = bf16_to_f32(f32_to_bf16(bf16_to_f32(f32_to_bf16(sv)) * uv[r]));
            else
              sv *= uv[r];
            v[r] = sv * sc;
          }
          *reinterpret_cast<uint32_t*>(orow + u * 64) = quant_4xe4m3(v[0], v[1], v[2], v[3]);
        }
      }
    }
    if constexpr (kExt && kHasXs) {
      if (wn == 0) {
        float amax = 0.f;
#pragma unroll
        for (int u = 0; u < 2; ++u) {
          const u32x2 ov = *reinterpret_cast<const u32x2*>(ext_xch + u * 128);
          const uint32_t m01 = pack_bf16x2(tot_ext[u][0], tot_ext[u][1]), m23 = pack_bf16x2(tot_ext[u][2], tot_ext[u][3]);
          const float gv[4] = {bf16lo_to_f32(m01), bf16hi_to_f32(m01), bf16lo_to_f32(m23), bf16hi_to_f32(m23)};
          const float uv[4] = {bf16lo_to_f32(ov[0]), bf16hi_to_f32(ov[0]), bf16lo_to_f32(ov[1]), bf16hi_to_f32(ov[1])};
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const float v = gv[r] / (1.0f + __expf(-gv[r])) * uv[r];
            tot_ext[u][r] = v;
            amax = fmaxf(amax, fabsf(v));
          }
        }
        amax = fmaxf(amax, __shfl_xor(amax, 16, 64));
        amax = fmaxf(amax, __shfl_xor(amax, 32, 64));
        if (g4 == 0) ext_red[wm * 16 + r16] = amax;
      }
      __syncthreads();
      if (wn == 0 && r16 < ext_cnt) {
        const float amax = fmaxf(fmaxf(ext_red[r16], ext_red[16 + r16]), fmaxf(ext_red[32 + r16], ext_red[48 + r16]));
        const float scale = amax / 448.0f;
        const float inv = 1.0f / (scale + 1e-8f);
        const long row = static_cast<long>(m0 + ext0 + r16);
        uint8_t* orow = a.act_out + row * inter + col0 + wm * 16 + g4 * 4;
#pragma unroll
        for (int u = 0; u < 2; ++u)
          *reinterpret_cast<uint32_t*>(orow + u * 64) =
              quant_4xe4m3(tot_ext[u][0] * inv, tot_ext[u][1] * inv, tot_ext[u][2] * inv, tot_ext[u][3] * inv);
        if (wm == 0 && g4 == 0) a.act_scale[row * (inter >> 7) + (col0 >> 7)] = scale;
      }
    }
    if (wn == 0) {
      xch += 4 * 16 * 64 * 2;
      finish(IntC<0>{}, IntC<1>{});
    } else if constexpr (!kHalf) {
      xch -= 4 * 16 * 64 * 2;
      finish(IntC<2>{}, IntC<0>{});
    }
    return;
  }
  // ---- epilogue ---------------------------------------------------------------------------------------------
  // A lane holds rows n = wn*128 + i*16 + g4*4 + r of token slot wm*64 + j*16 + r16: 8 bytes of the token's output
  // row per block.  v_permlane16_swap on the blocks i, i+1 hands lane quarter q the 8 rows [(q&2)*4, +8) of block
  // i + (q&1): 16 contiguous bytes per store, half as many stores.
#pragma unroll
  for (int j = 0; j < kJ; ++j) {
    const int slot = mt0 + row_of_slot(wm * 64 + j * 16 + r16);
    uint16_t* yrow = a.y + static_cast<long>(m0 + slot) * a.N + n0 + wn * 128 + (g4 & 1) * 16 + (g4 >> 1) * 8;
#pragma unroll
    for (int i = 0; i < 8; i += 2) {
      const uint32_t a0 = pack_bf16x2(tot[i][j][0], tot[i][j][1]), a1 = pack_bf16x2(tot[i][j][2], tot[i][j][3]);
      const uint32_t b0 = pack_bf16x2(tot[i + 1][j][0], tot[i + 1][j][1]), b1 = pack_bf16x2(tot[i + 1][j][2], tot[i + 1][j][3]);
      const auto s0 = __builtin_amdgcn_permlane16_swap(a0, b0, false, false);
      const auto s1 = __builtin_amdgcn_permlane16_swap(a1, b1, false, false);
      if (slot < m_cnt) *reinterpret_cast<u32x4*>(yrow + i * 16) = u32x4{s0[0], s1[0], s0[1], s1[1]};
    }
  }
  if constexpr (kExt) {  // the ride-along block: rows 16 wm + 4 g4 .. + 3 and 64 + the same of the wave's half, token ext0 + r16
    if (r16 < ext_cnt) {
      uint16_t* yrow = a.y + static_cast<long>(m0 + ext0 + r16) * a.N + n0 + wn * 128 + wm * 16 + g4 * 4;
#pragma unroll
      for (int u = 0; u < 2; ++u)
        *reinterpret_cast<u32x2*>(yrow + u * 64) =
            u32x2{pack_bf16x2(tot_ext[u][0], tot_ext[u][1]), pack_bf16x2(tot_ext[u][2], tot_ext[u][3])};
    }
  }
}

// epilogue of the tail body: wave w holds weight rows 32 w ... + 31 (2 row blocks) x 64 token slots (4 blocks)
template <bool kHasXs, bool kAct>
__device__ __forceinline__ void tail_finish(const Args& a, uint8_t* s_mem, f32x4 (&tot)[2][4], cint_ptr ws_row, int mt0, int n0,
                                            int m_cnt, int m0) {
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int r16 = lane & 15, g4 = lane >> 4;
  const int inter = a.N >> 1;  // kAct only
  const int col0 = n0 >> 1;    // kAct only
  if constexpr (!kHasXs) {
    const float gs = __int_as_float(ws_row[0]);  // per-tensor form: strides are zero, one scale per group
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j) tot[i][j] *= gs;
  }
  if constexpr (kAct) {
    // ---- fused activation epilogue: waves 4-7 hand their up values (bf16-rounded like the GEMM output the separate kernel
    // would read) to the gate wave of the same columns; blockwise form: the 128-column abs-max of a token is the maximum
    // over the four gate waves' 32 columns, through LDS.  Arithmetic of the full body's epilogue, value for value.
    __syncthreads();  // every wave is past its last LDS read; its last (empty) DMA has landed
    uint32_t* xch = reinterpret_cast<uint32_t*>(s_mem) + (((wave & 3) * 8) * 64 + lane) * 2;  // [gate wave][i * 4 + j][lane] of 8 B
    float* red = reinterpret_cast<float*>(s_mem + 16384);                                      // [gate wave][64 tokens]
    if (wave >= 4) {
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j)
          *reinterpret_cast<u32x2*>(xch + (i * 4 + j) * 128) =
              u32x2{pack_bf16x2(tot[i][j][0], tot[i][j][1]), pack_bf16x2(tot[i][j][2], tot[i][j][3])};
    }
    __syncthreads();
    if (wave < 4) {
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        float amax = 0.f;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
          const u32x2 ov = *reinterpret_cast<const u32x2*>(xch + (i * 4 + j) * 128);
          const uint32_t m01 = pack_bf16x2(tot[i][j][0], tot[i][j][1]), m23 = pack_bf16x2(tot[i][j][2], tot[i][j][3]);
          const float gv[4] = {bf16lo_to_f32(m01), bf16hi_to_f32(m01), bf16lo_to_f32(m23), bf16hi_to_f32(m23)};
          const float uv[4] = {bf16lo_to_f32(ov[0]), bf16hi_to_f32(ov[0]), bf16lo_to_f32(ov[1]), bf16hi_to_f32(ov[1])};
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const float g = gv[r], u = uv[r];
            float v;
            if constexpr (kHasXs) {
              v = g / (1.0f + __expf(-g)) * u;
            } else {
              float sv = g / (1.0f + __expf(-g));
              if (a.use_bf16_mul)
                sv = bf16_to_f32(f32_to_bf16(bf16_to_f32(f32_to_bf16(sv)) * u));
              else
                sv *= u;
              v = sv * a.act_mul_scale[0];
            }
            tot[i][j][r] = v;
            amax = fmaxf(amax, fabsf(v));
          }
        }
        if constexpr (kHasXs) {
          amax = fmaxf(amax, __shfl_xor(amax, 16, 64));
          amax = fmaxf(amax, __shfl_xor(amax, 32, 64));
          if (g4 == 0) red[wave * 64 + j * 16 + r16] = amax;
        }
      }
    }
    __syncthreads();
    if (wave < 4) {
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int slot = mt0 + j * 16 + r16;
        float inv = 1.0f, scale = 0.f;
        if constexpr (kHasXs) {
          const int t = j * 16 + r16;
          const float amax = fmaxf(fmaxf(red[t], red[64 + t]), fmaxf(red[128 + t], red[192 + t]));
          scale = amax / 448.0f;
          inv = 1.0f / (scale + 1e-8f);
        }
        if (slot < m_cnt) {
          uint8_t* orow = a.act_out + static_cast<long>(m0 + slot) * inter + col0 + wave * 32 + g4 * 4;
#pragma unroll
          for (int i = 0; i < 2; ++i)
            *reinterpret_cast<uint32_t*>(orow + i * 16) =
                quant_4xe4m3(tot[i][j][0] * inv, tot[i][j][1] * inv, tot[i][j][2] * inv, tot[i][j][3] * inv);
          if constexpr (kHasXs) {
            if (wave == 0 && g4 == 0) a.act_scale[static_cast<long>(m0 + slot) * (inter >> 7) + (col0 >> 7)] = scale;
          }
        }
      }
    }
    return;
  }
  // ---- plain epilogue: 16-byte stores through v_permlane16_swap (as in the full body) ---------------------------------
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int slot = mt0 + j * 16 + r16;
    uint16_t* yrow = a.y + static_cast<long>(m0 + slot) * a.N + n0 + wave * 32 + (g4 & 1) * 16 + (g4 >> 1) * 8;
    const uint32_t a0 = pack_bf16x2(tot[0][j][0], tot[0][j][1]), a1 = pack_bf16x2(tot[0][j][2], tot[0][j][3]);
    const uint32_t b0 = pack_bf16x2(tot[1][j][0], tot[1][j][1]), b1 = pack_bf16x2(tot[1][j][2], tot[1][j][3]);
    const auto s0 = __builtin_amdgcn_permlane16_swap(a0, b0, false, false);
    const auto s1 = __builtin_amdgcn_permlane16_swap(a1, b1, false, false);
    if (slot < m_cnt) *reinterpret_cast<u32x4*>(yrow) = u32x4{s0[0], s1[0], s0[1], s1[1]};
  }
}

// ---- the TAIL body: a group's last token tile when it holds <= 64 rows ------------------------------------------------
// At 512 +- 22 routed rows per expert half the groups end in a third token tile of a dozen rows.  The half-tile body
// above serves it with the staggered-section machinery of a full tile - four barriers per k-tile, 48 KB of staging - and
// costs ~0.62 of a full tile for 2-20 % of its rows (the routed GEMMs of the MoE take 15-19 % longer than at exactly 512
// rows per expert).  What such a tile needs is its 256 weight rows streamed past 64 tokens: DMA-bound work, 8 MFMAs per
// wave and k-tile.  This body is built for that:
//   * wave w owns weight rows 32 w .. 32 w + 31 of the tile (gate-up form: waves 0-3 gate columns, 4-7 the up rows of
//     the same columns) x all 64 tokens: 2 x 4 blocks, 32 accumulator registers;
//   * a wave fetches exactly the weight rows it consumes, into a PRIVATE three-stage LDS ring (4 pieces per k-tile, two
//     k-tiles ahead): no barrier guards the weight stream, only the wave's own counted vmcnt;
//   * the 64 token rows are shared: chunks of kTW = 3 k-slabs (24 KB) in two buffers, one DMA piece per wave and slab,
//     refilled a whole chunk ahead - ONE barrier per three k-tiles, so the two waves of a SIMD drift apart and cover each
//     other's LDS latency;
//   * same operand conventions, swizzle and arithmetic as the full body (one FMA per k block, the last two blocks of a
//     k-tile folded under the first MFMAs of the next): results are bit-identical to the half-tile body's
//     (tests/test_fuse_moe_blockwise.py::test_group_gemm_tail_body_is_bit_identical, development key 21 = 2);
//   * VMEM order per k-tile T (q = T % 3): [W(T+2) x 4] and, behind the chunk barrier at q = 0, [X chunk T/3 + 1 x 3,
//     scales x 1]; the wait in front of k-tile T leaves 4 (q = 0) or 8 + 3 + (scales) pieces in flight.
template <bool kHasXs, bool kAct, bool kKTail, bool kNt>
__device__ __forceinline__ void p8_tail_body(const Args& a, uint8_t* s_mem, int e, int mt0, int n0, int m_cnt, int m0) {
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int r16 = lane & 15, g4 = lane >> 4;
  const int K = a.K, KB = a.KB;
  const int inter = a.N >> 1;  // kAct only
  const int col0 = n0 >> 1;    // kAct only: first activation column of the tile
  const int p_row = lane >> 3, p_chunk = (lane & 7) ^ (lane >> 3);

  // ---- DMA roles -------------------------------------------------------------------------------------------------
  const int wrow0 = kAct ? (wave >> 2) * inter + col0 + (wave & 3) * 32 : n0 + wave * 32;  // first weight row (in the group)
  const uint8_t* wsrc = a.w + static_cast<long>(e) * a.N * K;
  const unsigned w_bytes = static_cast<unsigned>(a.N) * static_cast<unsigned>(K);
  unsigned w_voff[4];
#pragma unroll
  for (int q = 0; q < 4; ++q) w_voff[q] = static_cast<unsigned>(wrow0 + q * 8 + p_row) * static_cast<unsigned>(K) + p_chunk * 16;
  auto dma_w = [&](int T, bool on, auto stage, int q) {  // piece q of k-tile T's 32 rows -> stage
    constexpr int kS = decltype(stage)::value;
    const int koff = T * kBK;
    const auto rw = make_rsrc(wsrc, on ? w_bytes : 0u);
    const bool k_ok = !kKTail || koff + p_chunk * 16 < K;
    uint8_t* dst = s_mem + kTWOff + (wave * kTW + kS) * 4096 + q * 1024;
    // kNt: the group's ONLY token tile - nobody else reads these weight rows: non-temporal (streamed once, read by one CU)
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rw, (lds_void*)dst, 16, k_ok ? w_voff[q] : 0xffffff00u, koff, 0, kNt ? 2 : 0);
  };
  // the weight pieces of k-tiles 0 and 1 do not depend on the token rows: they could go first, but the wait arithmetic of
  // the loop wants the token chunk OLDER than them - the row-index load below is one L2 round trip
  unsigned x_voff;
  {
    const int slot = mt0 + wave * 8 + p_row;
    const int sc = slot < m_cnt ? slot : m_cnt - 1;
    const int xrow = a.row_index ? a.row_index[m0 + sc] : m0 + sc;
    x_voff = static_cast<unsigned>(xrow) * static_cast<unsigned>(K) + p_chunk * 16;
  }
  auto dma_x = [&](int T, auto buf, auto slab) {  // this wave's 8 token rows of k-tile T -> chunk buffer, slab
    constexpr int kP = decltype(buf)::value, kT = decltype(slab)::value;
    const int koff = T * kBK;
    const auto rx = make_rsrc(a.x, T < KB ? a.x_bytes : 0u);
    const bool k_ok = !kKTail || koff + p_chunk * 16 < K;
    uint8_t* dst = s_mem + kTXOff + (kP * kTW + kT) * 8192 + wave * 1024;
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rx, (lds_void*)dst, 16, k_ok ? x_voff : 0xffffff00u, koff, 0, 0);
  };
  unsigned xs_voff = 0;
  if constexpr (kHasXs) {
    const int slot = mt0 + lane;
    const int sc = slot < m_cnt ? slot : m_cnt - 1;
    const long cb = a.col_base ? static_cast<long>(as_const(a.col_base)[e]) * a.tile_m : 0;
    const long term = a.col_base ? cb + sc : static_cast<long>(a.row_index ? a.row_index[m0 + sc] : m0 + sc);
    xs_voff = static_cast<unsigned>(term * a.xs_row_stride * 4);
  }
  const int xs_kb_bytes = static_cast<int>(a.xs_kb_stride * 4);
  const unsigned xs_nrec = __builtin_amdgcn_readfirstlane(wave < kTW ? 0xffffffffu : 0u);
  auto dma_xs = [&](int chunk, auto buf) {  // every wave issues one piece per chunk (equal vmcnt counts); wave t < kTW fetches slab t's scales
    constexpr int kP = decltype(buf)::value;
    if constexpr (kHasXs) {
      const int T = chunk * kTW + wave;
      const auto rs = make_rsrc(a.xs, T < KB ? xs_nrec : 0u);
      uint8_t* dst = s_mem + (wave < kTW ? kTSOff + (kP * kTW + wave) * 256 : kTDummy);
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lds_void*)dst, 4, xs_voff, T * xs_kb_bytes, 0, 0);
    }
  };
  const cint_ptr ws_row = as_const(reinterpret_cast<const int*>(a.ws)) + static_cast<long>(e) * a.ws_group_stride +
                          ((kAct ? (wave >> 2) * inter + col0 : n0 + wave * 32) >> 7) * a.ws_ntile_stride;

  f32x4 tot[2][4];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) tot[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  // operand read bases (opaque: no v_add per read): row r, chunk c -> r * 128 + ((c ^ (r & 7)) << 4)
  int a_rd[2], b_rd[2], xs_rd;
#pragma unroll
  for (int c = 0; c < 2; ++c) {
    a_rd[c] = kTWOff + wave * kTW * 4096 + r16 * kBK + (((g4 + 4 * c) ^ (r16 & 7)) << 4);  // + stage * 4096 + i * 2048
    b_rd[c] = kTXOff + r16 * kBK + (((g4 + 4 * c) ^ (r16 & 7)) << 4);                      // + (buffer * kTW + slab) * 8192 + j * 2048
    asm volatile("" : "+v"(a_rd[c]), "+v"(b_rd[c]));
  }
  xs_rd = kTSOff + r16 * 4;  // + (buffer * kTW + slab) * 256 + j * 64
  asm volatile("" : "+v"(xs_rd));

  // ---- prologue: token chunk 0 (+ its scales), then the weight pieces of k-tiles 0 and 1 -------------------------------
  dma_x(0, IntC<0>{}, IntC<0>{});
  dma_x(1, IntC<0>{}, IntC<1>{});
  dma_x(2, IntC<0>{}, IntC<2>{});
  dma_xs(0, IntC<0>{});
#pragma unroll
  for (int q = 0; q < 4; ++q) dma_w(0, true, IntC<0>{}, q);
#pragma unroll
  for (int q = 0; q < 4; ++q) dma_w(1, 1 < KB, IntC<1>{}, q);
  __builtin_amdgcn_sched_barrier(0);

  constexpr int kNx = kTW + (kHasXs ? 1 : 0);  // pieces of a token chunk per wave
  f32x4 pend[2] = {f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}};
  float fpend[4] = {0.f, 0.f, 0.f, 0.f};
  auto k_tile = [&](int T, auto qc, auto pc) {
    constexpr int kQ = decltype(qc)::value, kP = decltype(pc)::value;  // weight stage = slab of the chunk; chunk buffer
    const bool on2 = T + 2 < KB;
    __builtin_amdgcn_sched_barrier(0);
    if constexpr (kQ == 0) {
      // W(T) landed (everything older too: this chunk's token slabs); all waves' slabs landed and the other buffer is free
      __builtin_amdgcn_s_waitcnt(0x0F70 | 4);
      __builtin_amdgcn_s_barrier();
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int q = 0; q < 4; ++q) dma_w(T + 2, on2, IntC<2>{}, q);
      dma_x(T + 3, IntC<1 - kP>{}, IntC<0>{});
      dma_x(T + 4, IntC<1 - kP>{}, IntC<1>{});
      dma_x(T + 5, IntC<1 - kP>{}, IntC<2>{});
      dma_xs(T / kTW + 1, IntC<1 - kP>{});
    } else {
#pragma unroll
      for (int q = 0; q < 4; ++q) dma_w(T + 2, on2, IntC<(kQ + 2) % kTW>{}, q);
      __builtin_amdgcn_sched_barrier(0);
      constexpr int kFly = 8 + kNx;
      __builtin_amdgcn_s_waitcnt(0x0F70 | (kFly & 15) | ((kFly >> 4) << 14));
    }
    __builtin_amdgcn_sched_barrier(0);
    u32x4 af[2][2], bf[4][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int c = 0; c < 2; ++c) af[i][c] = *reinterpret_cast<const u32x4*>(s_mem + a_rd[c] + kQ * 4096 + i * 2048);
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int c = 0; c < 2; ++c) bf[j][c] = *reinterpret_cast<const u32x4*>(s_mem + b_rd[c] + (kP * kTW + kQ) * 8192 + j * 2048);
    float f[4] = {1.f, 1.f, 1.f, 1.f};
    if constexpr (kHasXs) {
      const float wsk = __int_as_float(ws_row[T * a.ws_kb_stride]);
      float xsv[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) xsv[j] = *reinterpret_cast<const float*>(s_mem + xs_rd + (kP * kTW + kQ) * 256 + j * 64);
      __builtin_amdgcn_s_waitcnt(0xC07F);  // lgkmcnt(0)
#pragma unroll
      for (int j = 0; j < 4; ++j) f[j] = wsk * xsv[j];
    } else {
      __builtin_amdgcn_s_waitcnt(0xC07F);
    }
    __builtin_amdgcn_sched_barrier(0);
    f32x4 pv[2] = {f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}};
#pragma unroll
    for (int n = 0; n < 8; ++n) {
      const int i = n >> 2, j = n & 3;
      const i32x8 av = {static_cast<int>(af[i][0][0]), static_cast<int>(af[i][0][1]), static_cast<int>(af[i][0][2]),
                        static_cast<int>(af[i][0][3]), static_cast<int>(af[i][1][0]), static_cast<int>(af[i][1][1]),
                        static_cast<int>(af[i][1][2]), static_cast<int>(af[i][1][3])};
      const i32x8 bv = {static_cast<int>(bf[j][0][0]), static_cast<int>(bf[j][0][1]), static_cast<int>(bf[j][0][2]),
                        static_cast<int>(bf[j][0][3]), static_cast<int>(bf[j][1][0]), static_cast<int>(bf[j][1][1]),
                        static_cast<int>(bf[j][1][2]), static_cast<int>(bf[j][1][3])};
      if constexpr (kHasXs) {
        const f32x4 part = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(av, bv, f32x4{0.f, 0.f, 0.f, 0.f}, 0, 0, 0, 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
        if (n >= 2) {
          const int pi = (n - 2) >> 2, pj = (n - 2) & 3;
#pragma unroll
          for (int r = 0; r < 4; ++r) tot[pi][pj][r] = fmaf(pv[1][r], f[pj], tot[pi][pj][r]);
        } else {  // blocks 6, 7 of the previous k-tile, under its scales
#pragma unroll
          for (int r = 0; r < 4; ++r) tot[1][2 + n][r] = fmaf(pend[n][r], fpend[2 + n], tot[1][2 + n][r]);
        }
        pv[1] = pv[0];
        pv[0] = part;
      } else {
        tot[i][j] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(av, bv, tot[i][j], 0, 0, 0, 0, 0, 0);
      }
      __builtin_amdgcn_sched_barrier(0);
    }
    if constexpr (kHasXs) {
      pend[0] = pv[1];
      pend[1] = pv[0];
#pragma unroll
      for (int j = 0; j < 4; ++j) fpend[j] = f[j];
    }
  };
  for (int kb = 0; kb < KB; kb += 2 * kTW) {
    k_tile(kb, IntC<0>{}, IntC<0>{});
    if (kb + 1 < KB) k_tile(kb + 1, IntC<1>{}, IntC<0>{});
    if (kb + 2 < KB) k_tile(kb + 2, IntC<2>{}, IntC<0>{});
    if (kb + 3 < KB) k_tile(kb + 3, IntC<0>{}, IntC<1>{});
    if (kb + 4 < KB) k_tile(kb + 4, IntC<1>{}, IntC<1>{});
    if (kb + 5 < KB) k_tile(kb + 5, IntC<2>{}, IntC<1>{});
  }
  if constexpr (kHasXs) {
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
      for (int r = 0; r < 4; ++r) tot[1][2 + t][r] = fmaf(pend[t][r], fpend[2 + t], tot[1][2 + t][r]);
  }
  __builtin_amdgcn_s_waitcnt(0x0F70);  // drain the (empty) tail DMAs before the workgroup's LDS is reused / released

  tail_finish<kHasXs, kAct>(a, s_mem, tot, ws_row, mt0, n0, m_cnt, m0);
}

// DEVELOPMENT VARIANT (key 26 = 1; not in the shipped library): the tail body with the weights streamed THROUGH REGISTERS.
// Question it answers: is a tail tile - ~32 us for the 32 k-tiles of the MoE's gate-up GEMM, 3 times what its 8 MFMAs per
// wave and k-tile need - slow because the LDS-ring body above keeps too little in flight (two k-tiles of weights, the next
// three k-tiles of tokens)?  Nothing in a tail tile shares weights between waves, so here a lane loads exactly its MFMA A
// operand - chunks g and g + 4 of row r of a 16-row block, two 16-byte loads - into one of SIX register stages, five
// k-tiles ahead (20 KB per wave, 160 KB per CU in flight instead of 64), and the LDS the rings leave free holds token
// chunks of SIX k-slabs, fetched a whole chunk ahead, one barrier per six k-tiles.  Bit-identical results.
// Answer (profiles/round5_moe_tail_body_ab.txt): no.  Routed sizes of the MoE 2 955 / 1 463 us against 2 922-2 946 /
// 1 455-1 461 with the rings; 320 rows per expert 2 245 / 1 068 against 2 268 / 1 140; 64 rows per expert (every tile a
// tail, the whole chip streaming) 1 350-1 400 / 673-690 against 1 105 / 574: the rings' LDS-DMA moves full 128-byte lines,
// the operand-shaped loads 64-byte halves.  Either way a CU pulls ~45 GB/s here: what is in flight per CU is capped below
// what either body asks for (~64 KB at ~1.4 us), so a tail tile costs its bytes - 1 MB of weights nobody shares with
// it + 256 KB of tokens = ~28 us - whatever the depth of the software pipeline.  Tails are a bandwidth-per-CU cost.
//   * loads are inline asm and counted by hand; VMEM order per k-tile T (s = T % 6): [s = 0, behind the wait and the
//     barrier: token chunk T / 6 + 1 (6 slabs + scales)] then [W(T + 5) x 4].  The wait in front of k-tile T leaves in
//     flight: W(T + 1 ... T + 4) = 16, plus the chunk issued in k-tile T - s when s = 1 ... 4 (the chunk a k-tile opens
//     is older than its own W: no extra wait at s = 0).
template <bool kNt>
__device__ __forceinline__ void ld_w2(u32x4& c0, u32x4& c1, unsigned voff, unsigned voff1, i32x4 rs, int soff) {
  // chunk g at voff, chunk g + 4 at voff1 (= voff + 64, or out of range for the half k-tile of K % 128 == 64)
  if constexpr (kNt)
    asm volatile("s_nop 4\n\tbuffer_load_dwordx4 %0, %2, %4, %5 offen nt\n\tbuffer_load_dwordx4 %1, %3, %4, %5 offen nt"
                 : "=&v"(c0), "=&v"(c1) : "v"(voff), "v"(voff1), "s"(rs), "s"(soff));
  else
    asm volatile("s_nop 4\n\tbuffer_load_dwordx4 %0, %2, %4, %5 offen\n\tbuffer_load_dwordx4 %1, %3, %4, %5 offen"
                 : "=&v"(c0), "=&v"(c1) : "v"(voff), "v"(voff1), "s"(rs), "s"(soff));
}
template <int N>
__device__ __forceinline__ void wait_w(u32x4 (&w)[2][2]) {  // at most N VMEM operations outstanding; makes this stage valid
  asm volatile("s_waitcnt vmcnt(%4)" : "+v"(w[0][0]), "+v"(w[0][1]), "+v"(w[1][0]), "+v"(w[1][1]) : "n"(N));
}

template <bool kHasXs, bool kAct, bool kKTail, bool kNt>
__device__ __forceinline__ void p8_tail_body_r(const Args& a, uint8_t* s_mem, int e, int mt0, int n0, int m_cnt, int m0) {
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int r16 = lane & 15, g4 = lane >> 4;
  const int K = a.K, KB = a.KB;
  const int inter = a.N >> 1;  // kAct only
  const int col0 = n0 >> 1;    // kAct only: first activation column of the tile
  const int p_row = lane >> 3, p_chunk = (lane & 7) ^ (lane >> 3);

  // ---- weights: this lane's operand bytes of row block i: row wrow0 + 16 i + r16, chunks g4 and g4 + 4 -----------------
  const int wrow0 = kAct ? (wave >> 2) * inter + col0 + (wave & 3) * 32 : n0 + wave * 32;  // first weight row (in the group)
  const uint8_t* wsrc = a.w + static_cast<long>(e) * a.N * K;
  const unsigned w_bytes = static_cast<unsigned>(a.N) * static_cast<unsigned>(K);
  const uint64_t wbase = reinterpret_cast<uint64_t>(wsrc);
  const int w_lo = __builtin_amdgcn_readfirstlane(static_cast<int>(static_cast<uint32_t>(wbase)));
  const int w_hi = __builtin_amdgcn_readfirstlane(static_cast<int>(static_cast<uint32_t>(wbase >> 32)));
  unsigned w_voff[2];
#pragma unroll
  for (int i = 0; i < 2; ++i) w_voff[i] = static_cast<unsigned>(wrow0 + i * 16 + r16) * static_cast<unsigned>(K) + g4 * 16;
  u32x4 wr[6][2][2];  // [stage][row block][chunk g4 / g4 + 4]
  auto ld_w = [&](int T, auto stage) {
    constexpr int kS = decltype(stage)::value;
    const int koff = T * kBK;
    const i32x4 rs = i32x4{w_lo, w_hi, __builtin_amdgcn_readfirstlane(T < KB ? static_cast<int>(w_bytes) : 0), 0x00020000};
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      unsigned v1 = w_voff[i] + 64;
      if constexpr (kKTail) v1 = koff + 64 + g4 * 16 < K ? v1 : 0xffffff00u;
      ld_w2<kNt>(wr[kS][i][0], wr[kS][i][1], w_voff[i], v1, rs, koff);
    }
  };

  // ---- tokens (LDS-DMA, as in the LDS-ring tail body) ---------------------------------------------------------------------
  unsigned x_voff;
  {
    const int slot = mt0 + wave * 8 + p_row;
    const int sc = slot < m_cnt ? slot : m_cnt - 1;
    const int xrow = a.row_index ? a.row_index[m0 + sc] : m0 + sc;
    x_voff = static_cast<unsigned>(xrow) * static_cast<unsigned>(K) + p_chunk * 16;
  }
  auto dma_x = [&](int T, auto buf, auto slab) {  // this wave's 8 token rows of k-tile T -> chunk buffer, slab
    constexpr int kP = decltype(buf)::value, kT = decltype(slab)::value;
    const int koff = T * kBK;
    const auto rx = make_rsrc(a.x, T < KB ? a.x_bytes : 0u);
    const bool k_ok = !kKTail || koff + p_chunk * 16 < K;
    uint8_t* dst = s_mem + kRXOff + (kP * kTR + kT) * 8192 + wave * 1024;
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rx, (lds_void*)dst, 16, k_ok ? x_voff : 0xffffff00u, koff, 0, 0);
  };
  unsigned xs_voff = 0;
  if constexpr (kHasXs) {
    const int slot = mt0 + lane;
    const int sc = slot < m_cnt ? slot : m_cnt - 1;
    const long cb = a.col_base ? static_cast<long>(as_const(a.col_base)[e]) * a.tile_m : 0;
    const long term = a.col_base ? cb + sc : static_cast<long>(a.row_index ? a.row_index[m0 + sc] : m0 + sc);
    xs_voff = static_cast<unsigned>(term * a.xs_row_stride * 4);
  }
  const int xs_kb_bytes = static_cast<int>(a.xs_kb_stride * 4);
  const unsigned xs_nrec = __builtin_amdgcn_readfirstlane(wave < kTR ? 0xffffffffu : 0u);
  auto dma_xs = [&](int chunk, auto buf) {  // every wave issues one piece per chunk (equal vmcnt counts); wave t < kTR fetches slab t's scales
    constexpr int kP = decltype(buf)::value;
    if constexpr (kHasXs) {
      const int T = chunk * kTR + wave;
      const auto rs = make_rsrc(a.xs, T < KB ? xs_nrec : 0u);
      uint8_t* dst = s_mem + (wave < kTR ? kRSOff + (kP * kTR + wave) * 256 : kRDummy);
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lds_void*)dst, 4, xs_voff, T * xs_kb_bytes, 0, 0);
    }
  };
  const cint_ptr ws_row = as_const(reinterpret_cast<const int*>(a.ws)) + static_cast<long>(e) * a.ws_group_stride +
                          ((kAct ? (wave >> 2) * inter + col0 : n0 + wave * 32) >> 7) * a.ws_ntile_stride;

  f32x4 tot[2][4];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) tot[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  int b_rd[2], xs_rd;
#pragma unroll
  for (int c = 0; c < 2; ++c) {
    b_rd[c] = kRXOff + r16 * kBK + (((g4 + 4 * c) ^ (r16 & 7)) << 4);  // + (buffer * kTR + slab) * 8192 + j * 2048
    asm volatile("" : "+v"(b_rd[c]));
  }
  xs_rd = kRSOff + r16 * 4;  // + (buffer * kTR + slab) * 256 + j * 64
  asm volatile("" : "+v"(xs_rd));

  // ---- prologue, in the order the loop would have issued it: chunk 0 | W(0) ... W(4) -------------------------------------
  constexpr int kNx = kTR + (kHasXs ? 1 : 0);  // pieces of a token chunk per wave
  dma_x(0, IntC<0>{}, IntC<0>{});
  dma_x(1, IntC<0>{}, IntC<1>{});
  dma_x(2, IntC<0>{}, IntC<2>{});
  dma_x(3, IntC<0>{}, IntC<3>{});
  dma_x(4, IntC<0>{}, IntC<4>{});
  dma_x(5, IntC<0>{}, IntC<5>{});
  dma_xs(0, IntC<0>{});
  ld_w(0, IntC<0>{});
  ld_w(1, IntC<1>{});
  ld_w(2, IntC<2>{});
  ld_w(3, IntC<3>{});
  ld_w(4, IntC<4>{});
  __builtin_amdgcn_sched_barrier(0);

  f32x4 pend[2] = {f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}};
  float fpend[4] = {0.f, 0.f, 0.f, 0.f};
  auto k_tile = [&](int T, auto sc_, auto pc) {
    constexpr int kS = decltype(sc_)::value, kP = decltype(pc)::value;  // register stage = slab of the chunk; chunk buffer
    constexpr int kQ = kS;
    __builtin_amdgcn_sched_barrier(0);
    if constexpr (kS == 0) {
      wait_w<16>(wr[kS]);                 // W(T) landed - and this chunk's slabs (mine), issued six k-tiles ago
      __builtin_amdgcn_s_barrier();       // everybody's slabs landed; everybody is done with the other buffer
      __builtin_amdgcn_sched_barrier(0);
      dma_x(T + 6, IntC<1 - kP>{}, IntC<0>{});
      dma_x(T + 7, IntC<1 - kP>{}, IntC<1>{});
      dma_x(T + 8, IntC<1 - kP>{}, IntC<2>{});
      dma_x(T + 9, IntC<1 - kP>{}, IntC<3>{});
      dma_x(T + 10, IntC<1 - kP>{}, IntC<4>{});
      dma_x(T + 11, IntC<1 - kP>{}, IntC<5>{});
      dma_xs(T / kTR + 1, IntC<1 - kP>{});
    } else if constexpr (kS <= 4) {
      wait_w<16 + kNx>(wr[kS]);           // the chunk issued in k-tile T - s is younger than W(T)
    } else {
      wait_w<16>(wr[kS]);
    }
    __builtin_amdgcn_sched_barrier(0);
    ld_w(T + 5, IntC<(kS + 5) % 6>{});
    __builtin_amdgcn_sched_barrier(0);
    u32x4 bf[4][2];
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int c = 0; c < 2; ++c) bf[j][c] = *reinterpret_cast<const u32x4*>(s_mem + b_rd[c] + (kP * kTR + kQ) * 8192 + j * 2048);
    float f[4] = {1.f, 1.f, 1.f, 1.f};
    if constexpr (kHasXs) {
      const float wsk = __int_as_float(ws_row[T * a.ws_kb_stride]);
      float xsv[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) xsv[j] = *reinterpret_cast<const float*>(s_mem + xs_rd + (kP * kTR + kQ) * 256 + j * 64);
      __builtin_amdgcn_s_waitcnt(0xC07F);  // lgkmcnt(0)
#pragma unroll
      for (int j = 0; j < 4; ++j) f[j] = wsk * xsv[j];
    } else {
      __builtin_amdgcn_s_waitcnt(0xC07F);
    }
    __builtin_amdgcn_sched_barrier(0);
    f32x4 pv[2] = {f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}};
#pragma unroll
    for (int n = 0; n < 8; ++n) {
      const int i = n >> 2, j = n & 3;
      const i32x8 av = {static_cast<int>(wr[kS][i][0][0]), static_cast<int>(wr[kS][i][0][1]), static_cast<int>(wr[kS][i][0][2]),
                        static_cast<int>(wr[kS][i][0][3]), static_cast<int>(wr[kS][i][1][0]), static_cast<int>(wr[kS][i][1][1]),
                        static_cast<int>(wr[kS][i][1][2]), static_cast<int>(wr[kS][i][1][3])};
      const i32x8 bv = {static_cast<int>(bf[j][0][0]), static_cast<int>(bf[j][0][1]), static_cast<int>(bf[j][0][2]),
                        static_cast<int>(bf[j][0][3]), static_cast<int>(bf[j][1][0]), static_cast<int>(bf[j][1][1]),
                        static_cast<int>(bf[j][1][2]), static_cast<int>(bf[j][1][3])};
      if constexpr (kHasXs) {
        const f32x4 part = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(av, bv, f32x4{0.f, 0.f, 0.f, 0.f}, 0, 0, 0, 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
        if (n >= 2) {
          const int pi = (n - 2) >> 2, pj = (n - 2) & 3;
#pragma unroll
          for (int r = 0; r < 4; ++r) tot[pi][pj][r] = fmaf(pv[1][r], f[pj], tot[pi][pj][r]);
        } else {  // blocks 6, 7 of the previous k-tile, under its scales
#pragma unroll
          for (int r = 0; r < 4; ++r) tot[1][2 + n][r] = fmaf(pend[n][r], fpend[2 + n], tot[1][2 + n][r]);
        }
        pv[1] = pv[0];
        pv[0] = part;
      } else {
        tot[i][j] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(av, bv, tot[i][j], 0, 0, 0, 0, 0, 0);
      }
      __builtin_amdgcn_sched_barrier(0);
    }
    if constexpr (kHasXs) {
      pend[0] = pv[1];
      pend[1] = pv[0];
#pragma unroll
      for (int j = 0; j < 4; ++j) fpend[j] = f[j];
    }
  };
  for (int kb = 0; kb < KB; kb += 2 * kTR) {
    k_tile(kb, IntC<0>{}, IntC<0>{});
    if (kb + 1 < KB) k_tile(kb + 1, IntC<1>{}, IntC<0>{});
    if (kb + 2 < KB) k_tile(kb + 2, IntC<2>{}, IntC<0>{});
    if (kb + 3 < KB) k_tile(kb + 3, IntC<3>{}, IntC<0>{});
    if (kb + 4 < KB) k_tile(kb + 4, IntC<4>{}, IntC<0>{});
    if (kb + 5 < KB) k_tile(kb + 5, IntC<5>{}, IntC<0>{});
    if (kb + 6 < KB) k_tile(kb + 6, IntC<0>{}, IntC<1>{});
    if (kb + 7 < KB) k_tile(kb + 7, IntC<1>{}, IntC<1>{});
    if (kb + 8 < KB) k_tile(kb + 8, IntC<2>{}, IntC<1>{});
    if (kb + 9 < KB) k_tile(kb + 9, IntC<3>{}, IntC<1>{});
    if (kb + 10 < KB) k_tile(kb + 10, IntC<4>{}, IntC<1>{});
    if (kb + 11 < KB) k_tile(kb + 11, IntC<5>{}, IntC<1>{});
  }
  if constexpr (kHasXs) {
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
      for (int r = 0; r < 4; ++r) tot[1][2 + t][r] = fmaf(pend[t][r], fpend[2 + t], tot[1][2 + t][r]);
  }
  // drain: the (empty) loads past the last k-tile still write their stage registers and the chunk buffers
#pragma unroll
  for (int st = 0; st < 6; ++st) wait_w<0>(wr[st]);

  tail_finish<kHasXs, kAct>(a, s_mem, tot, ws_row, mt0, n0, m_cnt, m0);
}

template <class Cfg, bool kHasXs, bool kNoDma = false, bool kAct = false, bool kKTail = false>
__global__ __launch_bounds__(kThreads, 1) void gemm_fp8_p8_kernel(const Args a, const int* __restrict__ cu_tiles,
                                                                  int num_group) {
  __shared__ __attribute__((aligned(1024))) uint8_t s_mem[kLds];
  const int nt = a.N / kBN;  // kAct: N = 2 * inter, tile tn = columns [tn * 128, +128) of gate and of up
  const Item it = locate_item(as_const(cu_tiles), a.seqlens, a.cu_seqlens, num_group, nt, threadIdx.x & 63, blockIdx.x,
                              a.item_order, !kNoDma && !kKTail ? a.ext_rows : 0, a.item_scan_old != 0);
  if (!it.valid) return;
  const int e = __builtin_amdgcn_readfirstlane(it.e);
  const int m_cnt = __builtin_amdgcn_readfirstlane(it.m_cnt);
  const int m0 = __builtin_amdgcn_readfirstlane(it.m0);
  const int mt0 = __builtin_amdgcn_readfirstlane(it.mt) * kBM;
  const int n0 = __builtin_amdgcn_readfirstlane(it.wt) * kBN;
  // a group's last token tile: <= 64 rows the tail body, <= 128 rows the half-tile body (development key 21 = 1: neither,
  // 2: no tail body)
  // (development key 26 = 1: the register-streamed tail body instead of the LDS-ring one - development build only)
  const bool tail = m_cnt - mt0 <= 64 && a.no_half_tile == 0 && !kNoDma;
  const bool single = m_cnt <= 64 && a.nt_single;
  if (kHpcDevBuild && tail && a.tail_regs && single)
    p8_tail_body_r<kHasXs, kAct, kKTail, true>(a, s_mem, e, mt0, n0, m_cnt, m0);
  else if (kHpcDevBuild && tail && a.tail_regs)
    p8_tail_body_r<kHasXs, kAct, kKTail, false>(a, s_mem, e, mt0, n0, m_cnt, m0);
  else if (tail && single)
    p8_tail_body<kHasXs, kAct, kKTail, true>(a, s_mem, e, mt0, n0, m_cnt, m0);
  else if (tail)
    p8_tail_body<kHasXs, kAct, kKTail, false>(a, s_mem, e, mt0, n0, m_cnt, m0);
  else if (m_cnt - mt0 <= 128 && a.no_half_tile != 1)
    p8_body<Cfg, kHasXs, kNoDma, kAct, true, kKTail>(a, s_mem, e, mt0, n0, m_cnt, m0);
  else if constexpr (!kNoDma && !kKTail) {
    // a full tile of a group whose short tail rides along (locate_item) carries up to 16 of those rows as a 17th token block
    const int ext_cnt = __builtin_amdgcn_readfirstlane(it.ext_cnt);
    if (ext_cnt > 0)
      p8_body<Cfg, kHasXs, kNoDma, kAct, false, kKTail, true>(a, s_mem, e, mt0, n0, m_cnt, m0,
                                                              __builtin_amdgcn_readfirstlane(it.ext0), ext_cnt);
    else
      p8_body<Cfg, kHasXs, kNoDma, kAct, false, kKTail>(a, s_mem, e, mt0, n0, m_cnt, m0);
  } else
    p8_body<Cfg, kHasXs, kNoDma, kAct, false, kKTail>(a, s_mem, e, mt0, n0, m_cnt, m0);
}

#ifdef HPC_DEV
// development key 22: variant of the k-loop for the blockwise kernels (A/B runs, tools/tune_ggemm.py; tools/prof_p8.py)
template <class Cfg>
void launch_blockwise_variant(const Args& a, const int* cu_tiles, int num_group, dim3 grid, hipStream_t stream) {
  if (a.act_out)
    gemm_fp8_p8_kernel<Cfg, true, false, true><<<grid, kThreads, 0, stream>>>(a, cu_tiles, num_group);
  else
    gemm_fp8_p8_kernel<Cfg, true><<<grid, kThreads, 0, stream>>>(a, cu_tiles, num_group);
}
#endif

}  // namespace
}  // namespace ggemm
}  // namespace hpc

#ifdef HPC_DEV
static void* g_p8_prof = nullptr;  // device buffer [16 workgroups][8 waves][64] int32 (tools/prof_p8.py), null = off
extern "C" int hpc_dev_p8_prof_buffer(void* p) {
  g_p8_prof = p;
  return 0;
}
#endif

int hpc_ggemm_launch_p8(const hpc::ggemm::Args& a_in, const int* cu_tiles, int num_group, int m, int n,
                        hipStream_t stream) {
  using namespace hpc::ggemm;
  Args a = a_in;
  a.no_half_tile = hpc_dev_tuning_get(21);  // development: 1 = full body only, 2 = no tail body
  // a group's ONLY (<= 64-row) token tile streams its weights non-temporally (development key 24 = 1: default policy).
  // (A four-stage form of the weight rings with a single-slab token ring - 96 instead of 64 KB of weights in flight per CU -
  // was built, bit-identical, and measured no faster: T = 256 1 515-1 563 against 1 505-1 512 us, profiles/
  // round5_moe_kernel_choice.txt; the stream is not bound by the bytes in flight at that point.  Removed.)
  a.nt_single = hpc_dev_tuning_get(24) != 1;
  a.tail_regs = hpc_dev_tuning_get(26) == 1;
  a.item_scan_old = hpc_dev_tuning_get(43) == 1;
  // a group's short tail (<= 16 rows per full tile it has) rides along with its full tiles instead of running as a tail
  // item (blockwise scales; development key 49 = 1: tail items for every tail, the dispatch of round 5)
  a.ext_rows = hpc_dev_tuning_get(49) != 1;
  if (n % kBN || a.K < kBK) return HPC_ERR_UNSUPPORTED;
  const long max_tiles = m / kBM + num_group;  // upper bound of sum_g ceil(len_g / 256)
  const long items = max_tiles * (n / kBN) + 16;  // + 16: the per-XCD chunks of the full and of the tail tiles round up
  // tail tiles stay next to their full siblings (order 0).  Order 1 - all full tiles first, tail tiles last, which evens
  // out the end of a launch (the down GEMM of the MoE has ~9.4 items per CU) - measured SLOWER on the same box: gate-up /
  // down GEMM 3493 / 1624 us against 3225 / 1583 us: a tail tile that cannot meet its weight tile in L2 streams it from
  // memory and its DMA pieces land late (development key 23 = 1 selects order 1; profiles/round5_moe_ggemm_ab.txt)
  a.item_order = hpc_dev_tuning_get(23) == 1 ? 1 : 0;
  if (items > 0x7fffffffl) return HPC_ERR_UNSUPPORTED;
  dim3 grid(static_cast<unsigned>(items));
#ifdef HPC_DEV
  a.prof = g_p8_prof;
  if (a.has_xs && hpc_dev_tuning_get(22) > 0) {
    switch (hpc_dev_tuning_get(22)) {
      case 1: launch_blockwise_variant<CfgProf>(a, cu_tiles, num_group, grid, stream); break;
      case 2: launch_blockwise_variant<CfgRound4>(a, cu_tiles, num_group, grid, stream); break;
      case 3: launch_blockwise_variant<CfgRound4Prof>(a, cu_tiles, num_group, grid, stream); break;
      case 4: launch_blockwise_variant<CfgNoPrio>(a, cu_tiles, num_group, grid, stream); break;
      default: return HPC_ERR_INVALID;
    }
    HPC_CHECK_LAUNCH();
    return HPC_OK;
  }
#endif
  using P = CfgProduct;
  if (a.has_xs && a.act_out)
    gemm_fp8_p8_kernel<P, true, false, true><<<grid, kThreads, 0, stream>>>(a, cu_tiles, num_group);
  else if (a.act_out && a.K % kBK)
    gemm_fp8_p8_kernel<P, false, false, true, true><<<grid, kThreads, 0, stream>>>(a, cu_tiles, num_group);
  else if (a.act_out)
    gemm_fp8_p8_kernel<P, false, false, true><<<grid, kThreads, 0, stream>>>(a, cu_tiles, num_group);
  else if (kHpcDevBuild && a.has_xs && hpc_dev_tuning_get(18) == 1)
    gemm_fp8_p8_kernel<P, true, true><<<grid, kThreads, 0, stream>>>(a, cu_tiles, num_group);
  else if (a.has_xs)
    gemm_fp8_p8_kernel<P, true><<<grid, kThreads, 0, stream>>>(a, cu_tiles, num_group);
  else if (a.K % kBK)  // per-tensor scales, K % 128 == 64: the k-tail instantiation
    gemm_fp8_p8_kernel<P, false, false, false, true><<<grid, kThreads, 0, stream>>>(a, cu_tiles, num_group);
  else
    gemm_fp8_p8_kernel<P, false><<<grid, kThreads, 0, stream>>>(a, cu_tiles, num_group);
  HPC_CHECK_LAUNCH();
  return HPC_OK;
}
