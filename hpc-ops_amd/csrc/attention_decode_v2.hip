// FP8 paged decode attention for NHD pages, second generation: a wave fetches the rows of TWO adjacent kv heads
// (256 contiguous bytes per token) with one instruction, stages a whole wave-iteration through LDS (where the
// hardware transposes V), keeps the next one in flight in registers and plans its own work in closed form (the
// dynamic tile schedule of assign_task.hip restated on the head-pair axis).
//
// Measurements that shaped it (tools/probes/probe_pair.hip, probe_tr; rocprofv3 SQ counters, tools/pmc_decode.py):
//  * on NHD pages [page][token][head][128 B] a kv head's fp8 row is 128 bytes at a 1 KB stride.  A wave that asks
//    for 128-byte pieces streams at 0.75 of 8 TB/s in a pure read probe (0.58-0.66 in the first-generation kernel,
//    attention_decode.hip, which stays for HND pages, bf16 and per-token K scales); asking for the
//    256 bytes of a head pair IN ONE INSTRUCTION streams at 0.82-0.84 (512 bytes: 0.88).
//  * the chip needs >= 8 waves per CU issuing loads: 4 waves per CU with 32 KB each in flight top out at 0.75.
//  * a wave issues at most one instruction every ~4 cycles, scalar ones included.  The round-2 form of this kernel
//    spent ~300 of its ~650 instructions per wave-iteration on scalar bookkeeping (a generic cursor that re-derived
//    the whole description of a wave-iteration three times per trip, 64-bit address arithmetic for four
//    descriptors, 69 SGPRs spilled to VGPR lanes); this form carries a wave-iteration through the pipeline as two
//    SGPRs (first token, flag bits) + its page ids, keeps per-task facts in a three-entry queue that is touched
//    only at task boundaries, and folds everything loop-invariant (in-page row offset, head offset) into the lane
//    offsets once.
//
// Structure (replaces, for this case, reference src/attention/decode/sm90/dynamic/smallm_fp8_*_dim128_*.cu(h)):
//  * wave-iteration ("WI") = 32 rows of 256 B of K + the same of V (32 tokens x 2 heads):
//    16 x buffer_load_dwordx4 whose lanes cover 4 rows x 256 B (+ 6 small loads for
//    Q / q scales, real only for a wave's first WI of a task).  The loads are inline asm with hand-counted vmcnt
//    (hipcc drained the queue once per WI).
//  * at the top of a WI the landed registers are written to the wave's private 16 KB LDS stage
//    ([row][256 B], 16-byte chunks XOR-swizzled with the row index) and the registers are immediately
//    re-used for the NEXT WI's loads - across task boundaries too - so 16 KB per wave stay in flight while the
//    MFMAs run.  K leaves LDS as MFMA A operands with ds_read_b128 (one v_mfma_f32_16x16x128_f8f6f4 per
//    16-row block and 128-byte half); V leaves it through ds_read_b64_tr_b8, the gfx950 transpose read: every lane
//    gives the address of 8 bytes of one row, the 16 lanes of a group get back one COLUMN (dim) of the 8 x 16
//    tile each = the V^T operand of O^T += V^T P^T with no VALU work at all.
//  * the plan is computed by every wave in SGPRs from the request lengths: requests laid end to end on a tile
//    axis per head pair, range r owns tiles [r * per, (r + 1) * per) (workgroup = range x pair), found with one
//    ballot over a wave-wide prefix sum - the closed form of csrc/assign_task.hip (reference
//    assign_task.cu:362-492) on a different axis, so the caller's task map is not needed here (the
//    first-generation kernel consumes it).
//  * the 4 waves of a workgroup share a task (WIs w, w+4, ...), merge through LDS once per task (the idle
//    stage regions double as the merge buffer) and write bf16 y.  A request cut by a range boundary leaves
//    an fp32 partial + base-2 LSE per chunk (write-through stores) and takes a ticket on the request's arrival
//    counter - at the END of its workgroup's range, once for the (at most two) split tasks of the range; the chunk
//    that arrives LAST merges all of them in the same launch (the reference's static path does the same,
//    static_splitk_kernels.cuh:362-377) - no second kernel, no launch boundary.  That merge is
//    spread over the workgroup: wave w folds chunks w, w+4, ... into a partial in the same (max, sum, O) form the
//    per-task merge uses, and the per-task combine code finishes it.  The counters live at the start of the
//    call's scratch: a fixed 64 KB region that must be zero on first use and is left zero by every call (one
//    atomic add per arrival; round 2 tagged them with a launch epoch instead of requiring zeros, which cost a
//    second round trip per split task and could misread stale words).  Two workgroups per CU.
//  * a CU's two workgroups stream DIFFERENT slices of the token rows: workgroups from #CUs on serve head pair p ^ mask, the
//    slice across byte-address bit 9 from their CU mate's (round 5: +4-6 % on every fp8 shape with >= 4 pairs, +1-2.5 % bf16;
//    profiles/round5_decode_pair_map_ab.txt).
//  * fp8 numerics as in the first generation / the reference kernels.
#include <atomic>
#include <type_traits>
#include <utility>

#include "hpc_common.h"
#include "hpc_dev.h"
#include "../../include/hpc_amd.h"
#include "attention_decode_v2.h"

namespace hpc {
namespace decode2 {

#ifdef HPC_DEV
__device__ int g_ticket_overruns;  // see ticket_overruns() in attention_decode_v2.h
#endif
constexpr int kThreads = 256;
constexpr int kWaves = 4;
constexpr float kNegInf = -__builtin_inff();
constexpr int kRow = 256;                             // LDS row: head 0 | head 1 of a token, or token 2r | token 2r + 1
constexpr int kRows = 32;                             // rows per wave-iteration
constexpr int kVOff = kRows * kRow;                   // V half of a wave's stage
constexpr int kWaveLds = 2 * kRows * kRow;            // 16 KB: K + V of one WI; the merge buffer s_o[2][16][128] aliases it
typedef int v2i32 __attribute__((ext_vector_type(2)));
typedef __attribute__((address_space(3))) v2i32 lds_v2i32;
typedef short v4i16 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) v4i16 lds_v4i16;

__device__ __forceinline__ long pack64(uint32_t lo, uint32_t hi) {
  return static_cast<long>(static_cast<uint64_t>(lo) | (static_cast<uint64_t>(hi) << 32));
}
__device__ __forceinline__ int sgpr(int v) { return __builtin_amdgcn_readfirstlane(v); }

// ---- loads the compiler does not see ---------------------------------------------------------------------------
// hipcc's waitcnt insertion drained the whole queue once per wave-iteration in every compiler-visible form of this
// pipeline (vmcnt(0) at the V wait whatever the prefetch depth), so the K / V / Q loads are inline asm and their
// completion is counted by hand: loads retire in order, a WI's loads are issued in the fixed order
// K block 0 (4) | K block 1 (4) | V block 0 (4) | V block 1 (4) [| Q of the next task], and a consumer waits with
// vmcnt(number of loads issued after the ones it needs).  Stores and other compiler-visible memory operations in
// the queue only make these waits conservative, never unsafe.  Destinations are "v" registers: the kernel stays
// well below 256 registers, so hipcc has no reason to shuffle a not-yet-landed destination through an AGPR.
__device__ __forceinline__ i32x4 srd(uint32_t lo, uint32_t hi, int num_records) {
  return i32x4{sgpr(static_cast<int>(lo)), sgpr(static_cast<int>(hi)), sgpr(num_records), 0x00020000};
}
__device__ __forceinline__ i32x4 srd_of(const void* base, unsigned num_records) {
  const uint64_t v = reinterpret_cast<uint64_t>(base);
  return srd(static_cast<uint32_t>(v), static_cast<uint32_t>(v >> 32), static_cast<int>(num_records));
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
// bf16 Q fragments of one head: the four 16-byte chunks g, g + 4, g + 8, g + 12 of the 256-byte row (k-step j = chunk 4 j + g)
__device__ __forceinline__ void ld_q4(u32x4 (&q)[4], int voff_q, i32x4 rq_in) {
  const i32x4 rq = pin(rq_in);
  asm volatile("s_nop 4\n\tbuffer_load_dwordx4 %0, %4, %5, 0 offen\n\tbuffer_load_dwordx4 %1, %4, %5, 0 offen offset:64\n\t"
               "buffer_load_dwordx4 %2, %4, %5, 0 offen offset:128\n\tbuffer_load_dwordx4 %3, %4, %5, 0 offen offset:192"
               : "=&v"(q[0]), "=&v"(q[1]), "=&v"(q[2]), "=&v"(q[3])
               : "v"(voff_q), "s"(rq));
}
// kKtok: the 64 K scales of a wave-iteration (one dword per lane)
__device__ __forceinline__ void ld_ks(uint32_t& sc, int voff, i32x4 rs_in) {
  const i32x4 rs = pin(rs_in);
  asm volatile("s_nop 4\n\tbuffer_load_dword %0, %1, %2, 0 offen" : "=&v"(sc) : "v"(voff), "s"(rs));
}
template <int N>
__device__ __forceinline__ void wait_ks(uint32_t& sc) {
  asm volatile("s_waitcnt vmcnt(%1)" : "+v"(sc) : "n"(N));
}
template <int N>
__device__ __forceinline__ void wait_ks2(uint32_t& sc0, uint32_t& sc1) {
  asm volatile("s_waitcnt vmcnt(%2)" : "+v"(sc0), "+v"(sc1) : "n"(N));
}
template <int N>
__device__ __forceinline__ void wait_q8(u32x4 (&q0)[4], u32x4 (&q1)[4]) {
  asm volatile("s_waitcnt vmcnt(%8)"
               : "+v"(q0[0]), "+v"(q0[1]), "+v"(q0[2]), "+v"(q0[3]), "+v"(q1[0]), "+v"(q1[1]), "+v"(q1[2]), "+v"(q1[3])
               : "n"(N < 63 ? N : 63));
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

// flag bits of a wave-iteration
constexpr int kFValid = 1;    // inside this workgroup's range
constexpr int kFLast = 2;     // this wave's last WI of the task
constexpr int kFFirst = 4;    // this wave's first WI of the task
constexpr int kFOn0 = 8;      // rows 0..15 hold at least one token of the task
constexpr int kFOn1 = 16;     // rows 16..31 do
constexpr int kFMasked = 32;  // some token of the WI is invisible to some q row (request end / task end)

// kBf16: bf16 K / V / Q (BASELINE configs[1]).  A head pair's row is then 512 B; a wave-iteration is 16 tokens (the
//        same 8 KB of K + 8 KB of V in flight per wave), QK^T = four v_mfma_f32_16x16x32_bf16 per head, P is rounded to
//        bf16 (reference numerics: softmax in fp32, P.to(bf16) before P V), V^T comes out of LDS through
//        ds_read_b64_tr_b16 (lanes 4j .. 4j+3 of a 16-lane group supply row j of a 4 x 16 tile, lane i gets column i:
//        tools/probes/probe_tr16.hip) into v_mfma_f32_16x16x16_bf16; no scales.  All byte strides in Args.
// kQuad: fp8, FOUR adjacent kv heads per workgroup (512 contiguous bytes per token and load row, the shape that streams
//        best on NHD pages) for calls with <= 8 q rows per kv head (Sq * G <= 8: the graded decode shapes).  The 16
//        columns of an MFMA tile then hold the q rows of TWO kv heads - columns 0..7 head 2p, 8..15 head 2p + 1:
//          S^T: one K = 128 MFMA per head against the pair's packed Q^T; lane (n, g) keeps the result of the head its
//               column belongs to (4 selects) -> ONE softmax per pair instead of one per head;
//          O^T: the K = 32 fp8 MFMA takes k-slots 0..15 from head 2p's 16 tokens and 16..31 from head 2p + 1's:
//               A = [V_2p^T | V_2p+1^T] (two heads' transpose reads), B = P^T with the other head's columns zeroed
//               (v_permlane32_swap puts a column's 16 probabilities in the lane groups of its head, one v_and masks).
//        Same MFMA count per byte as the pair form, half the accumulators (2 tiles x 32 registers for 4 heads), half
//        the exponentials.  A wave-iteration is 16 tokens like the bf16 form (same 8 KB of K + 8 KB of V).
// kProf: development build that accumulates s_memtime deltas per wave (tools/prof_decode.py reads them)
// kKtok: fp8 with PER-TOKEN K scales and per-head V scales (quant_type 0; reference tests/test_attention_decode_qkpertoken_
//        perhead_vperhead_fp8.py).  The scales of a page live in its tail rows - row tok / 32, per head 32 floats = 128 bytes at
//        the head's place in the row - so the 32 tokens x 2 heads of a wave-iteration are ONE contiguous 256-byte piece: one
//        buffer_load_dword per wave-iteration (lane = head * 32 + token), issued between the K and the V loads, staged
//        through 256 bytes of LDS per wave, read back as the four tokens of a lane's score rows.  Scores are scaled
//        (s * qscale / sqrt(d) * log2 e) * kscale[token] - the first-generation kernel's order.
// kHnd:  fp8 head pairs on HND pages [page][head][token][128 B] (round 6, development key 55 = 1; the product keeps the first-generation kernel there).  A kv head's
//        tokens are contiguous there, so a load instruction fetches 8 tokens x 128 B of ONE head (1 KB contiguous - the widest
//        piece the probe knows, and no workgroup fixes byte-address bits 8-9) and a 16-row block is two instructions per head.
//        Only the lane -> (row, chunk) map of the loads and of the stage writes differs: the LDS image, and with it everything
//        downstream, is the NHD form's.
// kSolo: fp8, ONE kv head per workgroup with up to 32 q rows (round 6: speculative steps with num_seq_q * group in 17 ... 32 - the
//        first-generation kernel's two-block form ran those at 0.49 of 8 TB/s on the C3 mix, one workgroup per CU with 506
//        registers).  The workgroup's two "heads" hh = 0, 1 are the q-row halves [16 hh, 16 hh + 16) of the kv head: both read the
//        SAME K / V.  A stage row is one token's 128 bytes, a wave-iteration 64 tokens (the same 8 KB of K + 8 KB of V in flight),
//        a load instruction 8 tokens x 128 B; S^T = one K = 128 MFMA per 16-token block and half, the softmax runs over 64
//        tokens, O^T += V^T P^T takes two K = 32 steps per 16 dims whose V^T operands (transposing reads) serve both halves.
//        The head's rows sit at k / v_head_stride * head, tokens at k / v_token_stride: NHD and HND pages alike.
template <int kAux, bool kBf16 = false, bool kProf = false, bool kQuad = false, bool kKtok = false, bool kHnd = false, bool kSolo = false>
__global__ __launch_bounds__(kThreads, 2) void decode2_kernel(const Args a) {
  static_assert(!kSolo || (!kQuad && !kHnd), "one head per workgroup: fp8 (both quant types) or bf16");
  static_assert(!(kSolo && kBf16 && kKtok), "bf16 has no scales");
  static_assert(!(kBf16 && kQuad), "the quad form is fp8");
  static_assert(!kKtok || (!kBf16 && !kQuad), "per-token K scales: the fp8 head-pair form");
  static_assert(!kHnd || (!kBf16 && !kQuad && !kKtok), "HND pages: the fp8 head-pair form with per-tensor scales");
  __shared__ float s_ks[kKtok ? kWaves : 1][64];
  constexpr bool kWide = (kBf16 && !kSolo) || kQuad;  // 512-byte stage rows, 16-token wave-iterations
  // kSolo + kBf16 (round 6): one kv head's 256-byte rows, 32 tokens per wave-iteration - the fp8 pair form's stage geometry
  // ([32 rows][256 B], four rows per load instruction, the same K image) with the bf16 form's arithmetic; the V image has its own
  // key (slot = chunk ^ 2 (t % 4): the 4 rows x 2 chunks x 2 halves of a transposing b16 read fall on 16 different 8-byte places)
  constexpr bool kBSolo = kBf16 && kSolo;
  __shared__ __attribute__((aligned(1024))) uint8_t s_wave[kWaves][kWaveLds];  // stage addresses are (base) ^ (bits 4-7)
  __shared__ float s_m[2][kWaves][16];
  __shared__ float s_l[2][kWaves][16];
  __shared__ int s_ticket[2];
  constexpr int kH = kSolo ? 1 : kQuad ? 4 : 2;  // kv heads per workgroup
  constexpr int kW = kBSolo ? 32 : kSolo ? 64 : kWide ? 16 : 32;        // tokens (= stage rows) per wave-iteration
  constexpr int kRowB = kBSolo ? 256 : kSolo ? 128 : kWide ? 512 : 256;  // bytes of a stage row: the workgroup's heads of a token
  constexpr int kRpi = kBSolo ? 4 : (kHnd || kSolo) ? 8 : kWide ? 2 : 4;  // rows per load instruction (64 lanes x 16 B = 1 KB; HND / fp8 solo: 8 tokens of one head)
  constexpr int kCpr = 64 / kRpi;            // 16-byte chunks per row

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = sgpr(tid >> 6);
  const int n = lane & 15;  // MFMA N index: q row
  const int g = lane >> 4;  // MFMA k-slot group / C row group
  const int wg = blockIdx.x, nwg = gridDim.x;
  const int B = a.num_batch, Sq = a.num_seq_q;
  const int G = 1 << a.g_shift;
  const int rows_valid = Sq << a.g_shift;
  // tile row (= MFMA column) r16 -> does it hold a q row?  quad: rows 0..7 head 2p, 8..15 head 2p + 1
  // (solo: tile hh holds q rows 16 hh ... 16 hh + 15 of the one kv head)
  auto row_ok = [&](int r16, int hh = 0) __attribute__((always_inline)) { return (kQuad ? (r16 & 7) : kSolo ? 16 * hh + r16 : r16) < rows_valid; };
  const int page_mask = (1 << a.page_shift) - 1;
  const cint_ptr lens = as_const(a.lens);
  const int mpl_tiles = as_const(a.task_map)[6] >> 6;  // scalar load, in flight beside the length loads of the plan
  const int add_new = a.new_kv_included ? 0 : Sq;
  const uint8_t* qbase = static_cast<const uint8_t*>(a.q);
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
  // inclusive wave scan on the DPP path (round 6: six v_add with a DPP source instead of six ds_bpermute round trips in front of the
  // first page-id load): Hillis-Steele within the rows of 16, then lane 15 of rows 0 / 2 into rows 1 / 3, lane 31 into rows 2, 3
  int incl = chunk_sum;
  incl += __builtin_amdgcn_update_dpp(0, incl, 0x111, 0xF, 0xF, false);  // row_shr:1
  incl += __builtin_amdgcn_update_dpp(0, incl, 0x112, 0xF, 0xF, false);  // row_shr:2
  incl += __builtin_amdgcn_update_dpp(0, incl, 0x114, 0xF, 0xF, false);  // row_shr:4
  incl += __builtin_amdgcn_update_dpp(0, incl, 0x118, 0xF, 0xF, false);  // row_shr:8
  incl += __builtin_amdgcn_update_dpp(0, incl, 0x142, 0xA, 0xF, false);  // row_bcast:15 -> rows 1, 3
  incl += __builtin_amdgcn_update_dpp(0, incl, 0x143, 0xC, 0xF, false);  // row_bcast:31 -> rows 2, 3
  const int chunk_start = incl - chunk_sum;
  const int Th = __builtin_amdgcn_readlane(incl, 63);  // total cost per head pair
  const int npair = a.num_head_kv / kH;
  if (Th == 0) return;
  // Workgroup -> (range r, pair p) with the pair index MINOR: workgroups r * npair .. r * npair + npair - 1 stream
  // the npair 256-byte slices of the same token rows at about the same time (identical work, started together).
  // The grid is a multiple of npair (launcher).
  // The slices do not stream equally fast (measured per wave, tools/prof_decode.py dump: with 8 kv heads the slice at
  // byte 256 of every 1 KB token row takes ~20 % longer per wave-iteration than those at 0 and 512, the one at 768 ~8 %,
  // on every XCD), and the kernel ends with the slowest slice.  a.pair_wgs (when set: 4 slices) gives slice p its own
  // number of ranges; the extra workgroups of the slow slices are the last ones of the grid.
  int nrange = nwg / npair;
  int rng = wg / npair;
  int pr = wg % npair;  // this workgroup's head pair
  // A CU's two workgroups on DIFFERENT slices.  Measured (profiles/round5_decode_pair_map_ab.txt): when the second workgroup of
  // every CU (the dispatcher hands out workgroups in index order, one per CU, then the second ones) streams the slice whose
  // byte offset differs in address bit 9 from the first one's, the whole launch runs 4-6 % faster (C3 mix 141 -> 133-137 us,
  // uniform 8k 180 -> 175 us, every box); mates on the same slice (today's kernel until round 5), slices that differ in bit 8, or
  // an XCD that serves all four slices are all slower.  Every workgroup that streams alone on a slice runs as fast as any
  // other (development key 37), so this is not a property of the memory channels behind a slice.  Two more facts of the same
  // sweep that this mapping relies on: pair index bit 0 (= slice address bit 8 for fp8 pairs) is the PARITY of the XCD a
  // workgroup runs on (workgroups go round the 8 XCDs in index order) - every other assignment of slice bits to XCD index bits
  // is 3 % slower (development key 38) -, and the even XCDs stream bit-8 = 0 addresses faster than anything else streams (call 28).
  if (a.xcd_map != 0 && npair == 4 && (nwg & 7) == 0) {  // development: which XCD (= wg % 8) streams which slice
    const int e = (a.xcd_map >> (3 * (wg & 7))) & 7;
    pr = e & 3;
    rng = (wg >> 3) * 2 + (e >> 2);
  }
  if (a.mate_from > 0 && wg >= a.mate_from) pr ^= a.pair_xor;
  int lwg = wg;         // logical workgroup index: partial slots are addressed by (slice offset + range)
  int lwg0 = pr * nrange;  // first logical index of this slice
  if (a.pair_wgs[0] > 0 && npair == 4) {
    const int n0 = a.pair_wgs[0], n1 = a.pair_wgs[1], n2 = a.pair_wgs[2], n3 = a.pair_wgs[3];
    const int nmin = min(min(n0, n1), min(n2, n3));
    if (wg >= nmin * 4) {
      int e = wg - nmin * 4;
      pr = 0;
      rng = nmin + e;
      if (e >= n0 - nmin) { e -= n0 - nmin; pr = 1; rng = nmin + e;
        if (e >= n1 - nmin) { e -= n1 - nmin; pr = 2; rng = nmin + e;
          if (e >= n2 - nmin) { e -= n2 - nmin; pr = 3; rng = nmin + e; } } }
    }
    nrange = pr == 0 ? n0 : pr == 1 ? n1 : pr == 2 ? n2 : n3;
    lwg0 = pr == 0 ? 0 : pr == 1 ? n0 : pr == 2 ? n0 + n1 : n0 + n1 + n2;
  }
  lwg = lwg0 + rng;
  // a range is never smaller than min_range_cost: a batch with little work (one long request among a few short
  // ones) runs on fewer workgroups instead of being cut into one-tile chunks that the last arriver of the long
  // request has to merge one by one (15 x 64 + 1 x 16k tokens: 87-99 us with 128 ranges per pair, 47 us with a floor of 8)
  // Two sizes of range.  The first half of the grid is the first workgroup on every CU (the dispatcher hands out
  // workgroups in index order, one per CU, then the second ones ~4 us later: tools/prof_decode.py dump, 256 CUs x 2),
  // and a CU's first workgroup streams 13-19 % faster than its second for the whole kernel (the older waves win the
  // load-issue arbitration).  With a.big_pct > 100 the ranges of the first-half workgroups (range index < nbig) are
  // that much longer than the others: range r starts at r * per_b (r < nbig) or nbig * per_b + (r - nbig) * per_s.
  const int nbig = a.big_pct > 100 ? (nwg / 2) / npair : 0;
  const long denom = static_cast<long>(nbig) * a.big_pct + static_cast<long>(nrange - nbig) * 100;
  const int per_s_even = static_cast<int>((static_cast<long>(Th) * 100 + denom - 1) / denom);
  // ... and never smaller than the caller's min_process_len (tokens one workgroup processes at least: the scheduler
  // call's argument, which it records in header int 6 of the task map - sched_task_info.h).  The map's bins and split
  // decisions are not consumed on this path; its lower bound on the split granularity is.
  const int per_floor = a.min_range_cost > mpl_tiles ? a.min_range_cost : mpl_tiles;
  const int per_s = per_s_even > per_floor ? per_s_even : per_floor;
  const int per_b = nbig > 0 ? (per_s * a.big_pct + 99) / 100 : per_s;
  const int big_span = nbig * per_b;
  auto range_of = [&](int pos) __attribute__((always_inline)) { return pos < big_span ? pos / per_b : nbig + (pos - big_span) / per_s; };
  const long g_begin = rng < nbig ? static_cast<long>(rng) * per_b : big_span + static_cast<long>(rng - nbig) * per_s;
  const int per = rng < nbig ? per_b : per_s;
  const long g_end = g_begin + per < Th ? g_begin + per : Th;
  if (g_begin >= g_end) return;

  // The range is converted to real tiles once: it starts at tile c_rt0 of request c_b and ends in front of tile
  // e_rt of request e_b (a boundary inside a request's overhead units is a boundary at that request's first tile).
  // Round 6: when the FLOOR sizes the ranges (per_s_even <= per_floor: the launch has fewer tiles than it could spread - 64 x 512
  // tokens, a handful of short requests beside one long one) a request of <= kSnap tiles is never split: a boundary inside it
  // moves to its nearer end.  Such a launch is a chain of round trips, not a stream - a short request cut in two costs two task
  // ends, a partial, a ticket and a merge for a few wave-iterations of work, and the workgroups the imbalance would hurt are
  // idle anyway (reference benchmark case uniform_512, 8 / 64 heads: 32.4 us -> see profiles/round6_decode_ab.txt, call 10).
  // Both neighbours of a boundary compute the same snap (same function, same x); finish_task knows such a request has one chunk.
  constexpr int kSnap = 16;
  const bool snap = per_s_even <= per_floor && !(kHpcDevBuild && a.dev_nosnap);
  auto locate = [&](int x, int& b_out, int& rt_out, int& cc_out) __attribute__((always_inline)) {
    if (x >= Th) {
      b_out = B;
      rt_out = 0;
      cc_out = Th;
      return;
    }
    const uint64_t le = __ballot(chunk_start <= x);
    const int cl = 63 - __builtin_clzll(le);  // last lane whose chunk starts at or before x
    int pos = __builtin_amdgcn_readlane(chunk_start, cl);  // cl is wave-uniform (from the ballot)
    int b = cl * kpl;
    while (true) {  // position x lies in this chunk (empty requests cost nothing and are stepped over)
      const int t = cost_of(b);
      if (x < pos + t) break;
      pos += t;
      ++b;
    }
    if (snap && x > pos) {
      // underloaded launch (`snap`, below): a boundary inside a short request moves to the nearer end of that request
      const int t = cost_of(b);
      if (t - kOvh <= kSnap) {
        if (2 * (x - pos) >= t) {  // to its end = the start of the next non-empty request
          pos += t;
          ++b;
          while (b < B && cost_of(b) == 0) ++b;
          if (b >= B) {
            b_out = B;
            rt_out = 0;
            cc_out = Th;
            return;
          }
        }
        b_out = b;
        rt_out = 0;
        cc_out = pos;
        return;
      }
    }
    b_out = b;
    rt_out = x - pos > kOvh ? x - pos - kOvh : 0;
    cc_out = pos;
  };
  int c_b, c_rt0, c_cc, e_b, e_rt, e_cc;
  locate(static_cast<int>(g_begin), c_b, c_rt0, c_cc);
  locate(static_cast<int>(g_end), e_b, e_rt, e_cc);
  if (c_b == e_b && c_rt0 >= e_rt) return;  // the whole range lies inside one request's overhead units

  // ---- task queue: facts of the tasks whose WIs are in the pipeline (consumed / in flight / page ids requested) ---
  // pushed when the cursor opens a task, popped when the consumer finishes it; entry 0 = the task being consumed.
  int q0_b = 0, q0_ltot = 0, q0_end = 0, q0_cc = 0;
  int q1_b = 0, q1_ltot = 0, q1_end = 0, q1_cc = 0;
  int q2_b = 0, q2_ltot = 0, q2_end = 0, q2_cc = 0;
  int qn = 0;
  // ---- cursor (SGPRs): the next WI of this wave whose page ids will be requested ------------------------------
  // A wave's WIs of a task start at token (first tile of the task) * 64 + wave * kW and advance by 4 * kW; every
  // wave visits every task at least once (a phantom WI with no rows when the task is shorter than that), so that
  // the four waves meet at the task's merge.  The cursor advances lazily - at the start of the next step() -
  // so that at most three tasks are ever open (queue capacity).
  int c_tok = 0, c_tok_end = 0, c_ltot = 0, c_tiles = 0, c_first = 1, c_valid = 1, c_pending = 0, c_last = 0;
  cint_ptr c_row = as_const(a.block_ids);
  auto open_task = [&]() __attribute__((always_inline)) {  // c_b, c_rt0, c_cc set: derive the rest
    c_ltot = ltot_of(c_b);
    c_tiles = (c_ltot + 63) >> 6;
    const int end_t = c_b == e_b ? e_rt : c_tiles;
    c_tok_end = end_t * 64 < c_ltot ? end_t * 64 : c_ltot;
    c_tok = c_rt0 * 64 + wave * kW;
    c_first = 1;
    c_row = as_const(a.block_ids) + static_cast<long>(c_b) * a.max_blocks;
    // (selects on the VALUES: an if / else chain over the three entries becomes a store through a selected
    // pointer, which sends the whole queue to scratch memory)
    const bool e0 = qn == 0, e1 = qn == 1, e2 = qn >= 2;
    q0_b = e0 ? c_b : q0_b, q0_ltot = e0 ? c_ltot : q0_ltot, q0_end = e0 ? c_tok_end : q0_end, q0_cc = e0 ? c_cc : q0_cc;
    q1_b = e1 ? c_b : q1_b, q1_ltot = e1 ? c_ltot : q1_ltot, q1_end = e1 ? c_tok_end : q1_end, q1_cc = e1 ? c_cc : q1_cc;
    q2_b = e2 ? c_b : q2_b, q2_ltot = e2 ? c_ltot : q2_ltot, q2_end = e2 ? c_tok_end : q2_end, q2_cc = e2 ? c_cc : q2_cc;
    ++qn;
  };
  auto next_task = [&]() __attribute__((always_inline)) {
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
  // description of a WI as it travels down the pipeline: first token, flags, page ids of its two 16-row blocks
  const bool blk1_same_page = (1 << a.page_shift) >= kW;
  auto step = [&](int& tok, int& fl, int& pid0, int& pid1) __attribute__((always_inline)) {
    if (c_pending && c_valid) {
      if (c_last) {
        next_task();
      } else {
        c_tok += kWaves * kW;
        c_first = 0;
      }
    }
    c_pending = 1;
    c_last = c_tok + kWaves * kW >= c_tok_end;
    const bool on0 = c_valid && c_tok < c_tok_end, on1 = c_valid && c_tok + kW / 2 < c_tok_end;
    const bool masked = c_tok + kW > c_ltot - Sq + 1 || c_tok + kW > c_tok_end;
    fl = (c_valid ? kFValid : 0) | (c_last ? kFLast : 0) | (c_first ? kFFirst : 0) | (on0 ? kFOn0 : 0) | (on1 ? kFOn1 : 0) |
         (masked ? kFMasked : 0);
    tok = c_tok;
    pid0 = on0 ? c_row[c_tok >> a.page_shift] : 0;
    pid1 = (!blk1_same_page && on1) ? c_row[(c_tok + kW / 2) >> a.page_shift] : 0;
  };

  // ---- the registers the next WI lands in, lane offsets, descriptors ------------------------------------------------
  u32x4 kr[2][4];    // [16-row block][4 rows x 256 B per instruction]
  u32x4 vr[2][4];
  uint32_t qsc[2];
  // A row is 256 contiguous bytes: the two heads of a token (row stride = token stride) or two tokens of the one
  // head (row stride = 2 x 128 B).  Everything loop-invariant goes into the lane offset: the row a lane fetches
  // within an instruction (lane / 16), its 16-byte chunk (lane % 16), and the in-page row of the wave's WIs - a
  // wave's WIs start at multiples of 4 * kW tokens past a 64-token boundary, so tok0 % page is the same for all of
  // them (pages of 16 / 32 / 64 tokens), for block 0 and for block 1 (which may sit in the next page).
  const uint32_t k_rs = static_cast<uint32_t>(a.k_token_stride), v_rs = static_cast<uint32_t>(a.v_token_stride);
  const int in0 = (wave * kW) & page_mask;                                               // in-page row of block 0
  const int in1 = blk1_same_page ? in0 + kW / 2 : ((wave * kW + kW / 2) & page_mask);    // ... of block 1 (the second half of the rows)
  const int lane_off = (lane % kCpr) * 16;
  const int k_voff0 = static_cast<int>((in0 + lane / kCpr) * k_rs) + lane_off, k_voff1 = static_cast<int>((in1 + lane / kCpr) * k_rs) + lane_off;
  const int v_voff0 = static_cast<int>((in0 + lane / kCpr) * v_rs) + lane_off, v_voff1 = static_cast<int>((in1 + lane / kCpr) * v_rs) + lane_off;
  // the four instructions of a 16-row block: rows 0-3 | 4-7 | 8-11 | 12-15 of the pair's rows; HND: (tokens 0-7 | 8-15) of the
  // first head, then of the second one (a head stride further)
  const int ks1 = sgpr(kRpi * k_rs), ks2 = sgpr(kHnd ? static_cast<uint32_t>(a.k_head_stride) : 2 * kRpi * k_rs);
  const int ks3 = sgpr(kHnd ? static_cast<uint32_t>(a.k_head_stride) + kRpi * k_rs : 3 * kRpi * k_rs);
  const int vs1 = sgpr(kRpi * v_rs), vs2 = sgpr(kHnd ? static_cast<uint32_t>(a.v_head_stride) : 2 * kRpi * v_rs);
  const int vs3 = sgpr(kHnd ? static_cast<uint32_t>(a.v_head_stride) + kRpi * v_rs : 3 * kRpi * v_rs);
  const int mem_slice = a.dev_slice > 0 ? a.dev_slice - 1 : pr;
  const uint64_t kbase_h = reinterpret_cast<uint64_t>(a.kcache) + static_cast<uint64_t>(mem_slice) * (kSolo ? static_cast<uint64_t>(a.k_head_stride) : kHnd ? 2 * static_cast<uint64_t>(a.k_head_stride) : kRowB);
  const uint64_t vbase_h = reinterpret_cast<uint64_t>(a.vcache) + static_cast<uint64_t>(mem_slice) * (kSolo ? static_cast<uint64_t>(a.v_head_stride) : kHnd ? 2 * static_cast<uint64_t>(a.v_head_stride) : kRowB);
  const uint32_t kbs = static_cast<uint32_t>(a.k_block_stride), vbs = static_cast<uint32_t>(a.v_block_stride);  // < 4 GB (eligible())
  const bool mem = a.dev_nomem == 0;  // development key 15 = 1: K / V loads fetch nothing (compute-only timing)
  i32x4 dk0, dk1, dv0, dv1;  // descriptors of the WI being issued
  uint32_t ksr = 0;           // kKtok: the WI's 64 K scales in flight (lane = head * 32 + token)
  uint32_t ksr1 = 0;          // kKtok + kSolo: tokens 32 ... 63 of the one head (ksr: tokens 0 ... 31; lanes l and l + 32 hold the same)
  i32x4 dks = i32x4{0, 0, 0, 0x00020000};
  i32x4 dks1 = i32x4{0, 0, 0, 0x00020000};
  // K-scale tail row of the wave's WIs: they start at multiples of 32 tokens, so one row (tok / 32 within the page) and
  // one page hold a WI's scales (pages of 32 / 64 tokens: eligible()); the pair's two heads are 2 x 128 contiguous bytes
  // (one head per workgroup: the 64 tokens of a wave-iteration are two 128-byte pieces - tail rows in0 / 32 and in0 / 32 + 1 of the
  // page, or row in1 / 32 of the next page when a page holds 32 tokens)
  const uint64_t ksbase_h = reinterpret_cast<uint64_t>(a.kscale) + static_cast<uint64_t>(mem_slice) * (kSolo ? 1 : 2) * a.ks_head_stride +
                            static_cast<uint64_t>(in0 >> 5) * a.ks_row_stride;
  const uint64_t ksbase1_h = reinterpret_cast<uint64_t>(a.kscale) + static_cast<uint64_t>(mem_slice) * a.ks_head_stride +
                             static_cast<uint64_t>(blk1_same_page ? (in0 >> 5) + 1 : in1 >> 5) * a.ks_row_stride;
  const uint32_t ksbs = static_cast<uint32_t>(a.ks_block_stride);
  auto make_descs = [&](int fl, int pid0, int pid1) __attribute__((always_inline)) {
    const int nrec0 = (fl & kFOn0) && mem ? -1 : 0, nrec1 = (fl & kFOn1) && mem ? -1 : 0;
    const uint64_t ka = kbase_h + static_cast<uint64_t>(static_cast<uint32_t>(pid0)) * kbs;
    const uint64_t va = vbase_h + static_cast<uint64_t>(static_cast<uint32_t>(pid0)) * vbs;
    dk0 = srd(static_cast<uint32_t>(ka), static_cast<uint32_t>(ka >> 32), nrec0);
    dv0 = srd(static_cast<uint32_t>(va), static_cast<uint32_t>(va >> 32), nrec0);
    if (blk1_same_page) {
      dk1 = i32x4{dk0[0], dk0[1], sgpr(nrec1), 0x00020000};
      dv1 = i32x4{dv0[0], dv0[1], sgpr(nrec1), 0x00020000};
    } else {
      const uint64_t kb = kbase_h + static_cast<uint64_t>(static_cast<uint32_t>(pid1)) * kbs;
      const uint64_t vb = vbase_h + static_cast<uint64_t>(static_cast<uint32_t>(pid1)) * vbs;
      dk1 = srd(static_cast<uint32_t>(kb), static_cast<uint32_t>(kb >> 32), nrec1);
      dv1 = srd(static_cast<uint32_t>(vb), static_cast<uint32_t>(vb >> 32), nrec1);
    }
    if constexpr (kKtok) {
      const uint64_t sa = ksbase_h + static_cast<uint64_t>(static_cast<uint32_t>(pid0)) * ksbs;
      dks = srd(static_cast<uint32_t>(sa), static_cast<uint32_t>(sa >> 32), (fl & kFOn0) ? (kSolo ? 128 : 256) : 0);
      if constexpr (kSolo) {
        const uint64_t sb = ksbase1_h + static_cast<uint64_t>(static_cast<uint32_t>(blk1_same_page ? pid0 : pid1)) * ksbs;
        dks1 = srd(static_cast<uint32_t>(sb), static_cast<uint32_t>(sb >> 32), (fl & kFOn1) ? 128 : 0);
      }
    }
  };
  auto issue_k0 = [&]() __attribute__((always_inline)) { ld_x4x4<kAux>(kr[0], k_voff0, dk0, ks1, ks2, ks3); };
  auto issue_k1 = [&]() __attribute__((always_inline)) { ld_x4x4<kAux>(kr[1], k_voff1, dk1, ks1, ks2, ks3); };
  auto issue_v0 = [&]() __attribute__((always_inline)) { ld_x4x4<kAux>(vr[0], v_voff0, dv0, vs1, vs2, vs3); };
  auto issue_v1 = [&]() __attribute__((always_inline)) { ld_x4x4<kAux>(vr[1], v_voff1, dv1, vs1, vs2, vs3); };
  auto issue_ks = [&]() __attribute__((always_inline)) {  // (between K and V: the V waits do not change)
    if constexpr (kKtok && kSolo) {
      ld_ks(ksr, (lane & 31) * 4, dks);
      ld_ks(ksr1, (lane & 31) * 4, dks1);
    } else if constexpr (kKtok) {
      ld_ks(ksr, lane * 4, dks);
    }
  };

  // ---- LDS stage addressing (loop-invariant per lane) ----------------------------------------------------------
  // stage image [row 0..31][256 B]; the 16-byte chunk c of row t sits in slot c ^ key(t), key(t) =
  // (t & 15) ^ ((t >> 4) << 3): the 16 rows of a ds_read_b128 / the 8 rows of a transpose read hit 16 / 8
  // different slots.  All addresses are (loop-invariant base) ^ (compile-time constant).
  uint8_t* my_lds = s_wave[wave];
  float* my_so = reinterpret_cast<float*>(my_lds);
  const uint32_t lds0 = static_cast<uint32_t>(reinterpret_cast<uintptr_t>((lds_u8*)my_lds));  // LDS byte address of the stage
  // fp8: writes: lane (r4 = lane / 16, c = lane % 16) of instruction (tb, q) holds chunk c of row tb * 16 + q * 4 + r4
  //      K reads: lane (n, g), row tb * 16 + n, chunk hh * 8 + g + 4 c
  //      V transpose reads: lane (i = lane % 16, g): row j = i / 2 of the 8 x 16 tile is stage row 16 (j / 4) + 4 g + (j % 4)
  //      (the k-slot order of P: slot hb * 4 + r <-> row 16 hb + 4 g + r), 8-byte half i % 2, chunk hh * 8 + jj
  // bf16 (16 rows of 512 B = 32 chunks; K stage key(t) = t, V stage key(t) = rotl4(t) = 2 t % 16 | t / 8 - the 32 lanes of
  //      a transpose read are 8 rows x 2 chunks x 2 halves and must fall on 32 different 8-byte bank pairs):
  //      writes: lane (r2 = lane / 32, c = lane % 32) of instruction (h, q) holds chunk c of row h * 8 + q * 2 + r2
  //      K reads: lane (n, g), row n, chunk hh * 16 + j * 4 + g of k-step j
  //      V transpose reads: lane (i, g): row 4 g + i / 4, 8-byte piece i % 4 of the 32 bytes of dims jj * 16 .. + 15
  const int tj = (lane & 15) >> 1;
  const int ttok = 16 * (tj >> 2) + 4 * g + (tj & 3);
  const int btok = 4 * g + ((lane & 15) >> 2);                      // bf16 transpose reads: stage row of this lane
  const int bkey = ((btok << 1) & 15) | (btok >> 3);
  // quad (16 rows of 512 B, K and V stage key(t) = t): writes as bf16's K stage; K reads: lane (n, g), row n, chunks
  //      hh * 8 + g and hh * 8 + g + 4; V transpose reads: lane (i, g) of the pair-p MFMA reads head 2p + g / 2, row j = i / 2
  //      of its 8 x 16 tile = token 4 (g % 2) + j % 4 + 8 (j / 4) (the k-slot order permlane32_swap gives P), half i % 2 -
  //      the 32 lanes of a pass cover all 16 tokens of one chunk: 16 different slots x 2 halves
  const int qtok = 4 * (g & 1) + (tj & 3) + 8 * (tj >> 2);
  // HND writes: lane (r8 = lane / 8, c = lane % 8) of instruction (tb, q) holds chunk (q / 2) * 8 + c of row tb * 16 + (q % 2) * 8 + r8
  // solo (64 rows of 128 B = 8 chunks; key(t) = (t & 7) ^ ((t >> 4) & 1) * 4 - the 8 rows of a transpose read, tokens 4 g + j and
  //      16 + 4 g + j, fall on 8 different slots): writes: lane (r8 = lane / 8, c = lane % 8) of instruction (tb, q) holds chunk c of
  //      row tb * 32 + q * 8 + r8; K reads: lane (n, g), row tb * 16 + n, chunks g and g + 4; V transpose reads: k-step s, lane
  //      (i, g): stage row 32 s + 16 (j / 4) + 4 g + j % 4 with j = i / 2, 8-byte half i % 2, chunk jj
  const uint32_t w0_inv = (kSolo && !kBf16) ? lds0 + (lane >> 3) * kRowB + (((lane & 7) ^ (lane >> 3)) * 16)
                          : kHnd ? lds0 + (lane >> 3) * kRowB + (((lane & 7) ^ (lane >> 3)) * 16)
                          : kWide ? lds0 + (lane >> 5) * kRowB + (((lane & 31) ^ (lane >> 5)) * 16)
                                  : lds0 + (lane >> 4) * kRowB + (((lane & 15) ^ (lane >> 4)) * 16);
  const uint32_t w1_inv = kBSolo ? lds0 + kVOff + (lane >> 4) * kRowB + (((lane & 15) ^ ((lane >> 4) << 1)) * 16)   // bf16, one head: V stage
                                 : lds0 + kVOff + (lane >> 5) * kRowB + (((lane & 31) ^ ((lane >> 5) << 1)) * 16);  // bf16 V stage
  const uint32_t r0_inv = (kSolo && !kBf16) ? lds0 + n * kRowB + ((g ^ (n & 7)) * 16) : lds0 + n * kRowB + ((g ^ n) * 16);
  // bf16, one head: lane (i, g) of a transposing b16 read: row 4 g + i / 4 (+ 16 tb), chunk 2 jj + (i % 4) / 2, half i % 2
  const uint32_t t0_inv = kBSolo ? lds0 + kVOff + btok * kRowB + (((((lane & 3) >> 1) ^ ((btok & 3) << 1))) * 16) + (lane & 1) * 8
                          : kBf16 ? lds0 + kVOff + btok * kRowB + (((((lane & 3) >> 1) ^ bkey)) * 16) + (lane & 1) * 8
                          : kSolo ? lds0 + kVOff + ttok * kRowB + ((((ttok & 7) ^ (((ttok >> 4) & 1) << 2))) * 16) + (lane & 1) * 8
                          : kQuad ? lds0 + kVOff + qtok * kRowB + ((((g >> 1) << 3) ^ qtok) * 16) + (lane & 1) * 8
                                  : lds0 + kVOff + ttok * kRowB + ((((ttok & 15) ^ ((ttok >> 4) << 3))) * 16) + (lane & 1) * 8;
  const uint32_t p_keep = (n >> 3) == (g >> 1) ? 0xffffffffu : 0u;  // quad: this lane group carries k-slots of the column's head

  // ---- per-task state ----------------------------------------------------------------------------------------
  u32x4 qf[2][kBf16 ? 4 : 2];  // Q fragments of row n: fp8 16-byte chunks g and g + 4; bf16 chunks g, g + 4, g + 8, g + 12
  float row_scale[2];   // qscale * kscale / sqrt(d) * log2(e)
  float out_scale;       // vscale (l_run carries the factor 256 of P~)
  float out_scale_h[2] = {0.f, 0.f};  // kKtok: per head
  f32x4 o[2][8];         // O^T: o[hh][jj][r] = dim jj * 16 + 4 g + r of q row n
  float m_run[2], l_run[2];
  // Q fragments + q scales of the task's request, straight into qf / qsc.  Called when the PREVIOUS task has been
  // finished (qf is dead then) for the task of the WI that is already in flight, so these loads are the
  // youngest in the queue: the K / V waits of that WI only get more conservative.  One descriptor for the
  // workgroup's q heads - the kH * G q heads of its kv heads are contiguous - bounded to
  // the request's Sq rows: lanes of the rows past rows_valid read zeros.
  auto load_q = [&](int db) __attribute__((always_inline)) {
    if constexpr (kBSolo) {  // tile hh = q rows 16 hh + n of the kv head (a.ldq in bytes)
      const i32x4 rq = srd_of(qbase + static_cast<long>(db) * Sq * a.ldq + (pr << a.g_shift) * 256,
                              static_cast<unsigned>((Sq - 1) * a.ldq + G * 256));
#pragma unroll
      for (int hh = 0; hh < 2; ++hh) {
        const int r = 16 * hh + n;
        ld_q4(qf[hh], (r >> a.g_shift) * a.ldq + (r & (G - 1)) * 256 + g * 16, rq);
      }
      return;
    }
    if constexpr (kBf16) {  // a.ldq in bytes; one descriptor for the pair's 2 G q heads (256 B each)
      const int q_voff = (n >> a.g_shift) * a.ldq + (n & (G - 1)) * 256 + g * 16;
      const i32x4 rq = srd_of(qbase + static_cast<long>(db) * Sq * a.ldq + ((pr * kH) << a.g_shift) * 256,
                              static_cast<unsigned>((Sq - 1) * a.ldq + kH * G * 256));
      ld_q4(qf[0], q_voff, rq);
      ld_q4(qf[1], q_voff + G * 256, rq);
      return;
    }
    if constexpr (kQuad) {
      // column n of pair p's tile = q row n % 8 of kv head 2p + n / 8; columns past the call's q rows read nothing
      const int r8 = n & 7, hl = n >> 3;
      const bool okc = r8 < rows_valid;
      const int q_voff = okc ? (r8 >> a.g_shift) * a.ldq + ((hl << a.g_shift) + (r8 & (G - 1))) * 128 + g * 16 : -256;
      const int s_voff = okc ? ((r8 >> a.g_shift) * a.qscale_stride + (hl << a.g_shift) + (r8 & (G - 1))) * 4 : -256;
      const i32x4 rq = srd_of(qbase + static_cast<long>(db) * Sq * a.ldq + ((pr * kH) << a.g_shift) * 128,
                              static_cast<unsigned>((Sq - 1) * a.ldq + kH * G * 128));
      const i32x4 rsq = srd_of(a.qscale + static_cast<long>(db) * Sq * a.qscale_stride + ((pr * kH) << a.g_shift),
                               static_cast<unsigned>(((Sq - 1) * a.qscale_stride + kH * G) * 4));
      ld_q3(reinterpret_cast<u32x4(&)[2]>(qf[0]), qsc[0], q_voff, rq, s_voff, rsq);
      ld_q3(reinterpret_cast<u32x4(&)[2]>(qf[1]), qsc[1], okc ? q_voff + 2 * G * 128 : -256, rq, okc ? s_voff + 2 * G * 4 : -256, rsq);
      return;
    }
    if constexpr (kSolo) {  // tile hh = q rows 16 hh + n of the kv head; rows past the call's Sq * G read zeros (bounded descriptors)
      const i32x4 rq = srd_of(qbase + static_cast<long>(db) * Sq * a.ldq + (pr << a.g_shift) * 128,
                              static_cast<unsigned>((Sq - 1) * a.ldq + G * 128));
      const i32x4 rsq = srd_of(a.qscale + static_cast<long>(db) * Sq * a.qscale_stride + (pr << a.g_shift),
                               static_cast<unsigned>(((Sq - 1) * a.qscale_stride + G) * 4));
#pragma unroll
      for (int hh = 0; hh < 2; ++hh) {
        const int r = 16 * hh + n;
        ld_q3(reinterpret_cast<u32x4(&)[2]>(qf[hh]), qsc[hh], (r >> a.g_shift) * a.ldq + (r & (G - 1)) * 128 + g * 16, rq,
              ((r >> a.g_shift) * a.qscale_stride + (r & (G - 1))) * 4, rsq);
      }
      return;
    }
    const int q_voff = (n >> a.g_shift) * a.ldq + (n & (G - 1)) * 128 + g * 16;
    const int s_voff = ((n >> a.g_shift) * a.qscale_stride + (n & (G - 1))) * 4;
    const i32x4 rq = srd_of(qbase + static_cast<long>(db) * Sq * a.ldq + ((pr * kH) << a.g_shift) * 128,
                            static_cast<unsigned>((Sq - 1) * a.ldq + kH * G * 128));
    const i32x4 rsq = srd_of(a.qscale + static_cast<long>(db) * Sq * a.qscale_stride + ((pr * kH) << a.g_shift),
                             static_cast<unsigned>(((Sq - 1) * a.qscale_stride + kH * G) * 4));
    if constexpr (!kBf16) {
      ld_q3(reinterpret_cast<u32x4(&)[2]>(qf[0]), qsc[0], q_voff, rq, s_voff, rsq);
      ld_q3(reinterpret_cast<u32x4(&)[2]>(qf[1]), qsc[1], q_voff + G * 128, rq, s_voff + G * 4, rsq);
    }
  };
  auto reset_state = [&]() __attribute__((always_inline)) {
#pragma unroll
    for (int hh = 0; hh < 2; ++hh)
#pragma unroll
      for (int jj = 0; jj < 8; ++jj) o[hh][jj] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int hh = 0; hh < 2; ++hh) {
      m_run[hh] = kNegInf;
      l_run[hh] = 0.f;
    }
  };

  // ---- the pipeline: d (landing / consumed) <- p1 (in flight) <- p2 (page ids requested) ----------------------------
  int p1_tok, p1_fl, p2_tok, p2_fl, p2_pid0, p2_pid1;

  // Combine the four waves' (max, sum, O) partials in LDS into per-thread results: thread (row16, c8) gets 8 dims
  // of one q row, for head hh.  Returns the row's base-2 log-sum-exp ingredients through M and L.
  auto combine4 = [&](int hh, int row16, int c8, float (&acc)[8], float& M, float& L) __attribute__((always_inline)) {
    float mw[kWaves];
    M = kNegInf;
#pragma unroll
    for (int w = 0; w < kWaves; ++w) {
      mw[w] = s_m[hh][w][row16];
      M = fmaxf(M, mw[w]);
    }
    const float Mu = M == kNegInf ? 0.f : M;
    L = 0.f;
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
  };
  auto store_y = [&](int db, int hh, int row16, int c8, const float (&acc)[8], float inv) __attribute__((always_inline)) {
    const int qrow = kQuad ? (row16 & 7) : kSolo ? 16 * hh + row16 : row16;  // quad: tile hh = heads 2 hh, 2 hh + 1 of the workgroup
    const int rs = qrow >> a.g_shift;
    const int h = kQuad ? pr * kH + hh * 2 + (row16 >> 3) : kSolo ? pr : pr * kH + hh;
    uint16_t* dst = a.y + (static_cast<long>(db) * Sq + rs) * a.ldy + ((h << a.g_shift) + (qrow & (G - 1))) * 128 + c8 * 8;
    u32x4 pk;
#pragma unroll
    for (int i = 0; i < 4; ++i) pk[i] = pack_bf16x2(acc[2 * i] * inv, acc[2 * i + 1] * inv);
    st16(dst, pk);
  };

  // ---- split requests: arrival tickets and merges, deferred to the end of the workgroup's range --------------------
  // (reference: the last CTA of a request reduces, static_splitk_kernels.cuh:362-377; combine math:
  // splitk_combine_kernels.cuh.)  One atomic add per arrival: the counter region is zero on first use - the contract of
  // hpc_attention_decode_workspace_zero_bytes() - and the last arriver puts the zero back.
  int np = 0, p0_b = 0, p0_first = 0, p0_n = 0, p1_b = 0, p1_first = 0, p1_n = 0;
  auto merge_request = [&](int db, int first_rng, int nchunks) __attribute__((always_inline)) {
    int* cnt = a.arrive + static_cast<long>(pr) * B + db;
    // Wave w folds chunks w, w + 4, ... of every (head, row) into (max lse, sum of weights, weighted O) - the
    // form the per-task merge takes - two chunks x four (head, row) units in flight per lane; then the
    // per-task combine finishes.  lane (r4 = lane / 16, c8 = lane % 16): rows r4 + 4 i, dims c8 * 8 .. + 8.
    const int r4 = lane >> 4, c8l = lane & 15;
    auto slot_of = [&](int c, int hh) __attribute__((always_inline)) {
      return (static_cast<long>(lwg0 + first_rng + c) * 2 + (c == 0 ? 1 : 0)) * 2 + hh;
    };
#pragma unroll 1
    for (int ig = 0; ig < 2; ++ig) {     // row groups {r4, r4 + 4} and {r4 + 8, r4 + 12}
      if (!kQuad && ig * 8 >= rows_valid) break;   // wave-uniform
      float um[2][2], ul[2][2], ua[2][2][8];
#pragma unroll
      for (int hh = 0; hh < 2; ++hh)
#pragma unroll
        for (int i = 0; i < 2; ++i) {
          um[hh][i] = kNegInf;
          ul[hh][i] = 0.f;
#pragma unroll
          for (int e = 0; e < 8; ++e) ua[hh][i][e] = 0.f;
        }
      for (int c0 = wave; c0 < nchunks; c0 += 2 * kWaves) {
        float lv[2][2][2];
        u32x4 x0[2][2][2], x1[2][2][2];
#pragma unroll
        for (int u = 0; u < 2; ++u) {
          // (round 6: a second chunk that does not exist is neither loaded nor folded - rounds 3-5 loaded the first one again
          // and gave it weight zero; most split requests have two or three chunks, and a duplicate of a written-through partial
          // is one more trip past L2 at the very end of the launch.  Development key 58 = 1: the duplicate loads.)
          const int c = c0 + u * kWaves < nchunks ? c0 + u * kWaves : c0;
          if (u > 0 && c0 + u * kWaves >= nchunks && !(kHpcDevBuild && a.dev_merge_dup)) continue;  // wave-uniform
#pragma unroll
          for (int hh = 0; hh < 2; ++hh) {
            const long slot = slot_of(c, hh);
#pragma unroll
            for (int i = 0; i < 2; ++i) {
              const int row = ig * 8 + i * 4 + r4;
              lv[u][hh][i] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(lse_rs, static_cast<int>((slot * 16 + row) * 4), 0, 16));
              const int off = static_cast<int>(((slot * 16 + row) * 128 + c8l * 8) * 4);
              x0[u][hh][i] = __builtin_amdgcn_raw_buffer_load_b128(part_rs, off, 0, 16);
              x1[u][hh][i] = __builtin_amdgcn_raw_buffer_load_b128(part_rs, off + 16, 0, 16);
            }
          }
        }
#pragma unroll
        for (int u = 0; u < 2; ++u) {
          const bool real = c0 + u * kWaves < nchunks;
          if (!real && !(kHpcDevBuild && a.dev_merge_dup)) continue;  // wave-uniform
#pragma unroll
          for (int hh = 0; hh < 2; ++hh)
#pragma unroll
            for (int i = 0; i < 2; ++i) {
              const float lse = real ? lv[u][hh][i] : kNegInf;
              const float mn = fmaxf(um[hh][i], lse);
              const float mu = mn == kNegInf ? 0.f : mn;
              const float sc_old = __builtin_amdgcn_exp2f(um[hh][i] - mu), wgt = __builtin_amdgcn_exp2f(lse - mu);
              um[hh][i] = mn;
              ul[hh][i] = ul[hh][i] * sc_old + wgt;
#pragma unroll
              for (int e = 0; e < 4; ++e) {
                ua[hh][i][e] = fmaf(wgt, __uint_as_float(x0[u][hh][i][e]), ua[hh][i][e] * sc_old);
                ua[hh][i][4 + e] = fmaf(wgt, __uint_as_float(x1[u][hh][i][e]), ua[hh][i][4 + e] * sc_old);
              }
            }
        }
      }
#pragma unroll
      for (int hh = 0; hh < 2; ++hh)
#pragma unroll
        for (int i = 0; i < 2; ++i) {
          const int row = ig * 8 + i * 4 + r4;
          if (c8l == 0) {
            s_m[hh][wave][row] = um[hh][i];
            s_l[hh][wave][row] = ul[hh][i];
          }
          float* so = my_so + (hh * 16 + row) * 128 + c8l * 8;
          *reinterpret_cast<f32x4*>(so) = f32x4{ua[hh][i][0], ua[hh][i][1], ua[hh][i][2], ua[hh][i][3]};
          *reinterpret_cast<f32x4*>(so + 4) = f32x4{ua[hh][i][4], ua[hh][i][5], ua[hh][i][6], ua[hh][i][7]};
        }
    }
    __syncthreads();
    const int row16 = tid >> 4, c8 = tid & 15;
#pragma unroll
    for (int hh = 0; hh < 2; ++hh) {
      if (row_ok(row16, hh)) {
        float acc[8], M, L;
        combine4(hh, row16, c8, acc, M, L);
        store_y(db, hh, row16, c8, acc, L > 0.f ? 1.0f / L : 0.f);
      }
    }
    if (tid == 0) __hip_atomic_store(cnt, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);  // ready for the next call
    __syncthreads();  // the merge buffers are free again
  };
  auto flush_pending = [&]() __attribute__((always_inline)) {
    if (np == 0) return;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // my partial stores have reached memory
    __syncthreads();
    if (tid == 0) {
      s_ticket[0] = __hip_atomic_fetch_add(a.arrive + static_cast<long>(pr) * B + p0_b, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) + 1;
      if (np > 1)
        s_ticket[1] = __hip_atomic_fetch_add(a.arrive + static_cast<long>(pr) * B + p1_b, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) + 1;
    }
    __syncthreads();
    const int t0 = s_ticket[0], t1 = s_ticket[1];
#ifdef HPC_DEV
    if (tid == 0 && (t0 > p0_n || (np > 1 && t1 > p1_n))) atomicAdd(&g_ticket_overruns, 1);  // a counter was not zero on entry
#endif
    if (t0 == p0_n) merge_request(p0_b, p0_first, p0_n);
    if (np > 1 && t1 == p1_n) merge_request(p1_b, p1_first, p1_n);
    np = 0;
  };

  // ---- end of a task: merge the 4 waves and emit (the stage regions are idle: they double as s_o) ---------------
  auto finish_task = [&]() __attribute__((always_inline)) {
    const int db = q0_b;
    const int d_tiles = (q0_ltot + 63) >> 6;
    // tile t of the request sits at cost position req0 + kOvh + t
    const bool whole = snap && d_tiles <= kSnap;  // never split (see locate)
    const int first_rng = whole ? rng : range_of(q0_cc + kOvh);
    const int nchunks = whole ? 1 : range_of(q0_cc + kOvh + d_tiles - 1) - first_rng + 1;
    const int ichunk = rng - first_rng;
#pragma unroll
    for (int hh = 0; hh < 2; ++hh) {
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
      for (int hh = 0; hh < 2; ++hh) {
        float acc[8], M, L;
        combine4(hh, row16, c8, acc, M, L);
        const float inv = (L > 0.f ? 1.0f / L : 0.f) * (kKtok ? out_scale_h[hh] : out_scale);
        if (row_ok(row16, hh)) {
          if (nchunks == 1) {
            store_y(db, hh, row16, c8, acc, inv);
          } else {
            // fp32 partial + base-2 LSE of this chunk, written THROUGH to memory (sc1): the workgroup that
            // arrives last at the request reads them with sc1 loads - per-XCD L2s are not coherent, and
            // write-through stores + a drained counter are the cheap valid hand-off (no cache-wide fences)
            const long slot = (static_cast<long>(lwg) * 2 + (ichunk == 0 ? 1 : 0)) * 2 + hh;
            const int off = static_cast<int>(((slot * 16 + row16) * 128 + c8 * 8) * 4);
            __builtin_amdgcn_raw_buffer_store_b128(u32x4{__float_as_uint(acc[0] * inv), __float_as_uint(acc[1] * inv), __float_as_uint(acc[2] * inv),
                                                         __float_as_uint(acc[3] * inv)}, part_rs, off, 0, 16);
            __builtin_amdgcn_raw_buffer_store_b128(u32x4{__float_as_uint(acc[4] * inv), __float_as_uint(acc[5] * inv), __float_as_uint(acc[6] * inv),
                                                         __float_as_uint(acc[7] * inv)}, part_rs, off + 16, 0, 16);
            if (c8 == 0)
              __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(L > 0.f ? M + __builtin_amdgcn_logf(L) - (kBf16 ? 0.0f : 8.0f) : kNegInf),
                                                    lse_rs, static_cast<int>((slot * 16 + row16) * 4), 0, 16);
          }
        }
      }
    }
    if (nchunks > 1) {
      // ---- split request: this chunk's partial is on its way to memory; the arrival ticket - and, for the chunk that
      // arrives last, the merge - wait until this workgroup has walked its whole range (round 3: taken on the spot, the
      // ticket cost every split task a drain of the load queue, two barriers and an atomic round trip with the memory
      // pipeline idle; a range holds at most two split tasks - its first and its last)
      // (a range is an interval of the cost axis: only the request that crosses its start and the one that crosses its
      // end are split - or one request that crosses both)
      p1_b = np == 1 ? db : p1_b, p1_first = np == 1 ? first_rng : p1_first, p1_n = np == 1 ? nchunks : p1_n;
      p0_b = np == 0 ? db : p0_b, p0_first = np == 0 ? first_rng : p0_first, p0_n = np == 0 ? nchunks : p0_n;
      np = np < 2 ? np + 1 : 2;
    }
    __syncthreads();
    reset_state();
    q0_b = q1_b, q0_ltot = q1_ltot, q0_end = q1_end, q0_cc = q1_cc;
    q1_b = q2_b, q1_ltot = q2_ltot, q1_end = q2_end, q1_cc = q2_cc;
    --qn;
    if (p1_fl & kFValid) load_q(q0_b);  // p1: the WI in flight = the next task's first WI
  };

  // ---- prologue ------------------------------------------------------------------------------------------------
  reset_state();
#pragma unroll
  for (int hh = 0; hh < 2; ++hh) {
#pragma unroll
    for (int c = 0; c < (kBf16 ? 4 : 2); ++c) qf[hh][c] = u32x4{0u, 0u, 0u, 0u};
    qsc[hh] = 0u;
  }
#pragma unroll
  for (int hh = 0; hh < 2; ++hh) row_scale[hh] = 0.f;
  out_scale = 0.f;
  open_task();
  step(p2_tok, p2_fl, p2_pid0, p2_pid1);
  make_descs(p2_fl, sgpr(p2_pid0), sgpr(p2_pid1));
  issue_k0();
  issue_k1();
  issue_ks();
  issue_v0();
  issue_v1();
  p1_tok = p2_tok;
  p1_fl = p2_fl;
  load_q(q0_b);
  step(p2_tok, p2_fl, p2_pid0, p2_pid1);

  // ---- main loop: one wave-iteration per trip -------------------------------------------------------------
  uint64_t pf_wk = 0, pf_wr = 0, pf_wv = 0, pf_is = 0, pf_cp = 0, pf_fin = 0, pf_n = 0, pf_t0 = 0, pf_r0 = 0, pf_last = 0;
  auto now = [&]() __attribute__((always_inline)) -> uint64_t { return kProf ? __builtin_amdgcn_s_memtime() : 0; };
  if constexpr (kProf) {
    pf_t0 = now();
    pf_r0 = __builtin_amdgcn_s_memrealtime();
  }
  while (true) {
    const int d_tok = p1_tok, d_fl = p1_fl;  // the WI whose loads are landing
    if (!(d_fl & kFValid)) break;
    pf_last = now();
    // keep the ~32 LDS addresses of a WI out of loop-invariant registers: they are one XOR away from these three
    // bases, and 32 pinned VGPRs were the difference between 2 waves per SIMD and spilling
    uint32_t w0 = w0_inv, w1 = w1_inv, r0 = r0_inv, t0 = t0_inv;
    asm volatile("" : "+v"(w0), "+v"(w1), "+v"(r0), "+v"(t0));
    // the next WI's descriptors (its page ids were requested a whole iteration ago)
    make_descs(p2_fl, sgpr(p2_pid0), sgpr(p2_pid1));
    p1_tok = p2_tok;
    p1_fl = p2_fl;
    auto write_k = [&](int tb) __attribute__((always_inline)) {
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        if constexpr (kSolo && !kBf16)
          *reinterpret_cast<lds_u32x4*>(static_cast<uint32_t>((w0 ^ (((q >> 1) << 2) * 16)) + (tb * 32 + q * 8) * kRowB)) = kr[tb][q];
        else if constexpr (kHnd)
          *reinterpret_cast<lds_u32x4*>(static_cast<uint32_t>((w0 ^ ((((q >> 1) << 3) ^ ((q & 1) << 3) ^ (tb << 3)) * 16)) + (tb * 16 + (q & 1) * 8) * kRowB)) = kr[tb][q];
        else if constexpr (kWide)
          *reinterpret_cast<lds_u32x4*>(static_cast<uint32_t>((w0 ^ (((q << 1) ^ (tb << 3)) * 16)) + (tb * 8 + q * 2) * kRowB)) = kr[tb][q];
        else
          *reinterpret_cast<lds_u32x4*>(static_cast<uint32_t>((w0 ^ (((q << 2) ^ (tb << 3)) * 16)) + (tb * 16 + q * 4) * kRowB)) = kr[tb][q];
      }
    };
    auto write_v = [&](int tb) __attribute__((always_inline)) {
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        if constexpr (kBSolo)
          *reinterpret_cast<lds_u32x4*>(static_cast<uint32_t>(w1 + (tb * 16 + q * 4) * kRowB)) = vr[tb][q];
        else if constexpr (kSolo)
          *reinterpret_cast<lds_u32x4*>(static_cast<uint32_t>((w0 ^ (((q >> 1) << 2) * 16)) + kVOff + (tb * 32 + q * 8) * kRowB)) = vr[tb][q];
        else if constexpr (kHnd)
          *reinterpret_cast<lds_u32x4*>(static_cast<uint32_t>((w0 ^ ((((q >> 1) << 3) ^ ((q & 1) << 3) ^ (tb << 3)) * 16)) + kVOff + (tb * 16 + (q & 1) * 8) * kRowB)) = vr[tb][q];
        else if constexpr (kBf16)
          *reinterpret_cast<lds_u32x4*>(static_cast<uint32_t>((w1 ^ (((q << 2) ^ tb) * 16)) + (tb * 8 + q * 2) * kRowB)) = vr[tb][q];
        else if constexpr (kQuad)
          *reinterpret_cast<lds_u32x4*>(static_cast<uint32_t>((w0 ^ (((q << 1) ^ (tb << 3)) * 16)) + kVOff + (tb * 8 + q * 2) * kRowB)) = vr[tb][q];
        else
          *reinterpret_cast<lds_u32x4*>(static_cast<uint32_t>((w0 ^ (((q << 2) ^ (tb << 3)) * 16)) + kVOff + (tb * 16 + q * 4) * kRowB)) = vr[tb][q];
      }
    };
    auto q_ready = [&]() __attribute__((always_inline)) {
      // first WI of a task: its Q loads are older than everything issued in this trip
      // (rows past rows_valid came back as zeros from the bounded descriptors)
      if (d_fl & kFFirst) {
        if constexpr (kBf16) {
          wait_q8<0>(qf[0], qf[1]);
          row_scale[0] = row_scale[1] = a.scale_log2;
          out_scale = 1.0f;
        } else {
          wait_q<0>(reinterpret_cast<u32x4(&)[2]>(qf[0]), reinterpret_cast<u32x4(&)[2]>(qf[1]), qsc[0], qsc[1]);
          const float kmul = kKtok ? 1.0f : as_constf(a.kscale)[0];
#pragma unroll
          for (int hh = 0; hh < 2; ++hh) row_scale[hh] = a.scale_log2 * __uint_as_float(qsc[hh]) * kmul;
          out_scale = as_constf(a.vscale)[0];  // the 1/256 of the reference formula cancels: l = 256 sum p
          if constexpr (kKtok) {  // per-head V scales
            out_scale_h[0] = as_constf(a.vscale)[kSolo ? pr : pr * 2];
            out_scale_h[1] = as_constf(a.vscale)[kSolo ? pr : pr * 2 + 1];
          }
        }
      }
    };
    // registers -> the wave's LDS stage (rows of 256 B, chunks swizzled); a register set is free again as soon as
    // it has been written out: the next WI (of this or the next task) goes in flight
    // (kKtok: the scale load sits between the K and the V loads - one more load younger than K)
    constexpr int kNks = kKtok ? (kSolo ? 2 : 1) : 0;  // scale loads between the K and the V loads
    wait_x4x4<12 + kNks>(kr[0]);
    wait_x4x4<8 + kNks>(kr[1]);
    if constexpr (kProf) { const uint64_t t = now(); pf_wk += t - pf_last; pf_last = t; }
    write_k(0);
    write_k(1);
    if constexpr (kKtok && kSolo) {
      wait_ks2<8>(ksr, ksr1);
      s_ks[wave][lane] = __uint_as_float(lane < 32 ? ksr : ksr1);  // index = token of the wave-iteration
    } else if constexpr (kKtok) {
      wait_ks<8>(ksr);
      s_ks[wave][lane] = __uint_as_float(ksr);
    }
    if constexpr (kProf) { const uint64_t t = now(); pf_wr += t - pf_last; pf_last = t; }
    wait_x4x4<4>(vr[0]);
    wait_x4x4<0>(vr[1]);
    if constexpr (kProf) { const uint64_t t = now(); pf_wv += t - pf_last; pf_last = t; }
    write_v(0);
    write_v(1);
    if constexpr (kProf) { const uint64_t t = now(); pf_wr += t - pf_last; pf_last = t; }
    // (round 5: s_setprio 2 around this load issue, around the whole memory phase of a wave-iteration, or s_setprio 1
    //  around its compute phase change nothing - C3 mix 140.1-141.0 us against 140.3-141.9 us, uniform 8k 185.3-185.6
    //  against 180.2-185.5 us, profiles/round5_decode_ab.txt: the waves wait for the memory pipeline, not for issue slots)
    if constexpr (kHpcDevBuild) {  // development key 39: (in front of the load issue) throttle the workgroups of the even head pairs (the fast slices)
      if (a.dev_sleep > 0 && !(pr & 1))
        for (int i = 0; i < a.dev_sleep; ++i) __builtin_amdgcn_s_sleep(1);
    }
    issue_k0();
    issue_k1();
    issue_ks();
    issue_v0();
    issue_v1();
    q_ready();
    step(p2_tok, p2_fl, p2_pid0, p2_pid1);  // the WI after the one just issued: its page ids are on their way while this one computes
    if constexpr (kProf) { const uint64_t t = now(); pf_is += t - pf_last; pf_last = t; }

    if constexpr (kBSolo) {
      // S^T = K Q^T: four K = 32 MFMAs per 16-token block and q-row half; a block's K fragments serve both halves
      f32x4 sacc[2][2];
#pragma unroll
      for (int tb = 0; tb < 2; ++tb) {
        f32x4 acc[2] = {f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}};
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const u32x4 kk = *reinterpret_cast<const lds_u32x4*>(static_cast<uint32_t>((r0 ^ (((j << 2) ^ (tb << 3)) * 16)) + tb * 16 * kRowB));
#pragma unroll
          for (int hh = 0; hh < 2; ++hh)
            acc[hh] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, kk), __builtin_bit_cast(bf16x8, qf[hh][j]), acc[hh], 0, 0, 0);
        }
        sacc[0][tb] = acc[0];
        sacc[1][tb] = acc[1];
      }
      v4i16 pf[2][2];
#pragma unroll
      for (int hh = 0; hh < 2; ++hh)
#pragma unroll
        for (int tb = 0; tb < 2; ++tb)
#pragma unroll
          for (int r = 0; r < 4; ++r) sacc[hh][tb][r] *= row_scale[hh];
      if (d_fl & kFMasked) {  // wave-uniform: only the WIs that hold a request's last tokens
#pragma unroll
        for (int hh = 0; hh < 2; ++hh) {
          const int sq_row = (16 * hh + n) >> a.g_shift;
          const int lim = (q0_end - 1 < q0_ltot - Sq + sq_row) ? q0_end - 1 : q0_ltot - Sq + sq_row;
#pragma unroll
          for (int tb = 0; tb < 2; ++tb)
#pragma unroll
            for (int r = 0; r < 4; ++r) sacc[hh][tb][r] = (d_tok + tb * 16 + g * 4 + r) <= lim ? sacc[hh][tb][r] : kNegInf;
        }
      }
#pragma unroll
      for (int hh = 0; hh < 2; ++hh) {
        float mt = kNegInf;
#pragma unroll
        for (int tb = 0; tb < 2; ++tb)
#pragma unroll
          for (int r = 0; r < 4; ++r) mt = __builtin_fmaxf(mt, sacc[hh][tb][r]);
        mt = row4_max(mt);
        const float m_new = fmaxf(m_run[hh], mt);
        const float m_use = m_new == kNegInf ? 0.f : m_new;
        float psum = 0.f;
#pragma unroll
        for (int tb = 0; tb < 2; ++tb) {
          float prb[4];
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            prb[r] = __builtin_amdgcn_exp2f(sacc[hh][tb][r] - m_use);
            psum += prb[r];
          }
          pf[hh][tb] = __builtin_bit_cast(v4i16, u32x2{pack_bf16x2(prb[0], prb[1]), pack_bf16x2(prb[2], prb[3])});
        }
        if (__builtin_amdgcn_ballot_w64(m_new != m_run[hh]) != 0) {
          const float alpha = __builtin_amdgcn_exp2f(m_run[hh] - m_use);
          l_run[hh] *= alpha;
#pragma unroll
          for (int jj = 0; jj < 8; ++jj) o[hh][jj] *= alpha;
          m_run[hh] = m_new;
        }
        l_run[hh] += psum;
      }
      // O^T += V^T P^T: per 16-token block eight transposing reads (16 dims each) that feed both halves
#pragma unroll
      for (int tb = 0; tb < 2; ++tb) {
        v4i16 vt[8];
#pragma unroll
        for (int u = 0; u < 8; ++u)
          vt[u] = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
              reinterpret_cast<lds_v4i16*>(static_cast<uint32_t>((t0 ^ ((u << 1) * 16)) + tb * 16 * kRowB)));
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int hh = 0; hh < 2; ++hh)
#pragma unroll
          for (int u = 0; u < 8; ++u)
            o[hh][u] = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(vt[u], pf[hh][tb], o[hh][u], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
      }
    } else if constexpr (kBf16) {
      // S^T = K Q^T: four K = 32 MFMAs per head (lane (n, g): 8 dims of row n per k-step on both sides)
      f32x4 sacc[2];
#pragma unroll
      for (int hh = 0; hh < 2; ++hh) {
        f32x4 acc = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const u32x4 kk = *reinterpret_cast<const lds_u32x4*>(static_cast<uint32_t>(r0 ^ (((hh << 4) ^ (j << 2)) * 16)));
          acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, kk), __builtin_bit_cast(bf16x8, qf[hh][j]), acc, 0, 0, 0);
        }
        sacc[hh] = acc;
      }
      // online softmax in base 2, per head; lane (n, g) holds tokens 4 g + r of q row n; P is rounded to bf16 for P V,
      // the row sum uses the unrounded p (reference tests/test_attention_decode_bf16.py:15-59, kernels alike)
      v4i16 pf[2];
#pragma unroll
      for (int hh = 0; hh < 2; ++hh) {
#pragma unroll
        for (int r = 0; r < 4; ++r) sacc[hh][r] *= row_scale[hh];
      }
      if (d_fl & kFMasked) {  // wave-uniform: only the WIs that hold a request's last tokens
        const int sq_row = n >> a.g_shift;
        const int lim = (q0_end - 1 < q0_ltot - Sq + sq_row) ? q0_end - 1 : q0_ltot - Sq + sq_row;
#pragma unroll
        for (int hh = 0; hh < 2; ++hh)
#pragma unroll
          for (int r = 0; r < 4; ++r) sacc[hh][r] = (d_tok + g * 4 + r) <= lim ? sacc[hh][r] : kNegInf;
      }
#pragma unroll
      for (int hh = 0; hh < 2; ++hh) {
        float mt = __builtin_fmaxf(__builtin_fmaxf(sacc[hh][0], sacc[hh][1]), __builtin_fmaxf(sacc[hh][2], sacc[hh][3]));
        mt = row4_max(mt);
        const float m_new = fmaxf(m_run[hh], mt);
        const float m_use = m_new == kNegInf ? 0.f : m_new;
        float prb[4], psum = 0.f;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          prb[r] = __builtin_amdgcn_exp2f(sacc[hh][r] - m_use);
          psum += prb[r];
        }
        const uint32_t p01 = pack_bf16x2(prb[0], prb[1]), p23 = pack_bf16x2(prb[2], prb[3]);
        pf[hh] = __builtin_bit_cast(v4i16, u32x2{p01, p23});
        if (__builtin_amdgcn_ballot_w64(m_new != m_run[hh]) != 0) {
          const float alpha = __builtin_amdgcn_exp2f(m_run[hh] - m_use);
          l_run[hh] *= alpha;
#pragma unroll
          for (int jj = 0; jj < 8; ++jj) o[hh][jj] *= alpha;
          m_run[hh] = m_new;
        }
        l_run[hh] += psum;
      }
      // O^T += V^T P^T: one transpose read per 16 dims (4 tokens x 16 dims per 16-lane group) and one K = 16 MFMA
#pragma unroll
      for (int hh = 0; hh < 2; ++hh) {
        v4i16 vt[8];
#pragma unroll
        for (int u = 0; u < 8; ++u)
          vt[u] = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
              reinterpret_cast<lds_v4i16*>(static_cast<uint32_t>(t0 ^ (((hh << 4) ^ (u << 1)) * 16))));
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int u = 0; u < 8; ++u)
          o[hh][u] = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(vt[u], pf[hh], o[hh][u], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
      }
    } else if constexpr (kQuad) {
      // S^T = K Q^T: one K = 128 MFMA per head (16 tokens x 128 dims) against the pair's packed Q^T (columns 0..7 the
      // q rows of head 2p, 8..15 those of head 2p + 1); a lane keeps the product of the head its column belongs to
      f32x4 sacc[2];
      const bool hi_col = n >= 8;
#pragma unroll
      for (int p = 0; p < 2; ++p) {
        f32x4 sh[2];
        const u32x4 q0 = qf[p][0], q1 = qf[p][1];
        const i32x8 qv8 = {static_cast<int>(q0[0]), static_cast<int>(q0[1]), static_cast<int>(q0[2]), static_cast<int>(q0[3]),
                           static_cast<int>(q1[0]), static_cast<int>(q1[1]), static_cast<int>(q1[2]), static_cast<int>(q1[3])};
#pragma unroll
        for (int e = 0; e < 2; ++e) {
          const int hh = p * 2 + e;
          const u32x4 k0 = *reinterpret_cast<const lds_u32x4*>(static_cast<uint32_t>(r0 ^ ((hh << 3) * 16)));
          const u32x4 k1 = *reinterpret_cast<const lds_u32x4*>(static_cast<uint32_t>(r0 ^ (((hh << 3) ^ 4) * 16)));
          const i32x8 kv8 = {static_cast<int>(k0[0]), static_cast<int>(k0[1]), static_cast<int>(k0[2]), static_cast<int>(k0[3]),
                             static_cast<int>(k1[0]), static_cast<int>(k1[1]), static_cast<int>(k1[2]), static_cast<int>(k1[3])};
          sh[e] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(kv8, qv8, f32x4{0.f, 0.f, 0.f, 0.f}, 0, 0, 0, 0, 0, 0);
        }
        const float rsc = row_scale[p];
#pragma unroll
        for (int r = 0; r < 4; ++r) sacc[p][r] = (hi_col ? sh[1][r] : sh[0][r]) * rsc;
      }
      if (d_fl & kFMasked) {  // wave-uniform: only the WIs that hold a request's last tokens
        const int sq_row = (n & 7) >> a.g_shift;
        const int lim = (q0_end - 1 < q0_ltot - Sq + sq_row) ? q0_end - 1 : q0_ltot - Sq + sq_row;
#pragma unroll
        for (int p = 0; p < 2; ++p)
#pragma unroll
          for (int r = 0; r < 4; ++r) sacc[p][r] = (d_tok + g * 4 + r) <= lim ? sacc[p][r] : kNegInf;
      }
      // online softmax in base 2, ONE per head pair: lane (n, g) holds tokens 4 g + r of column n.  P~ = e4m3(256 p) as
      // in the pair form.  The probabilities of a column then move to the lane groups that carry its head's k-slots:
      // permlane32_swap(w, w) = {w of lane group g % 2, w of lane group g % 2 + 2} = k-slots (tokens) 4 (g % 2) + r and
      // 4 (g % 2) + 8 + r; lane groups 0, 1 feed head 2p's slots (columns 8..15 zero there), groups 2, 3 head 2p + 1's.
      uint32_t pf[2][2];
#pragma unroll
      for (int p = 0; p < 2; ++p) {
        float mt = __builtin_fmaxf(__builtin_fmaxf(sacc[p][0], sacc[p][1]), __builtin_fmaxf(sacc[p][2], sacc[p][3]));
        mt = row4_max(mt);
        const float m_new = fmaxf(m_run[p], mt);
        const float m_use = m_new == kNegInf ? 0.f : m_new;
        const float m8 = m_use - 8.0f;
        float prb[4], psum = 0.f;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          prb[r] = __builtin_amdgcn_exp2f(sacc[p][r] - m8);
          psum += prb[r];
        }
        int w = __builtin_amdgcn_cvt_pk_fp8_f32(prb[0], prb[1], 0, false);
        w = __builtin_amdgcn_cvt_pk_fp8_f32(prb[2], prb[3], w, true);
        const auto sw = __builtin_amdgcn_permlane32_swap(static_cast<uint32_t>(w), static_cast<uint32_t>(w), false, false);
        pf[p][0] = sw[0] & p_keep;
        pf[p][1] = sw[1] & p_keep;
        if (__builtin_amdgcn_ballot_w64(m_new != m_run[p]) != 0) {
          const float alpha = __builtin_amdgcn_exp2f(m_run[p] - m_use);
          l_run[p] *= alpha;
#pragma unroll
          for (int jj = 0; jj < 8; ++jj) o[p][jj] *= alpha;
          m_run[p] = m_new;
        }
        l_run[p] += psum;
      }
      // O^T += [V_2p^T | V_2p+1^T] P^T: lane groups 0, 1 transpose-read head 2p's 16 tokens, groups 2, 3 head 2p + 1's
#pragma unroll
      for (int p = 0; p < 2; ++p) {
        v2i32 vt[8];
#pragma unroll
        for (int u = 0; u < 8; ++u)
          vt[u] = __builtin_amdgcn_ds_read_tr8_b64_v2i32(
              reinterpret_cast<lds_v2i32*>(static_cast<uint32_t>(t0 ^ (((p << 4) | u) * 16))));
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int u = 0; u < 8; ++u)
          o[p][u] = __builtin_amdgcn_mfma_f32_16x16x32_fp8_fp8(
              pack64(static_cast<uint32_t>(vt[u][0]), static_cast<uint32_t>(vt[u][1])), pack64(pf[p][0], pf[p][1]), o[p][u], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
      }
    } else if constexpr (kSolo && !kBf16) {
      // S^T = K Q^T: one K = 128 MFMA per 16-token block and q-row half; the K fragments of a block serve both halves
      f32x4 sacc[2][4];
#pragma unroll
      for (int tb = 0; tb < 4; ++tb) {
        const u32x4 k0 = *reinterpret_cast<const lds_u32x4*>(static_cast<uint32_t>((r0 ^ (((tb & 1) << 2) * 16)) + tb * 16 * kRowB));
        const u32x4 k1 = *reinterpret_cast<const lds_u32x4*>(static_cast<uint32_t>((r0 ^ ((((tb & 1) << 2) ^ 4) * 16)) + tb * 16 * kRowB));
        const i32x8 kv8 = {static_cast<int>(k0[0]), static_cast<int>(k0[1]), static_cast<int>(k0[2]), static_cast<int>(k0[3]),
                           static_cast<int>(k1[0]), static_cast<int>(k1[1]), static_cast<int>(k1[2]), static_cast<int>(k1[3])};
#pragma unroll
        for (int hh = 0; hh < 2; ++hh) {
          const u32x4 q0 = qf[hh][0], q1 = qf[hh][1];
          const i32x8 qv8 = {static_cast<int>(q0[0]), static_cast<int>(q0[1]), static_cast<int>(q0[2]), static_cast<int>(q0[3]),
                             static_cast<int>(q1[0]), static_cast<int>(q1[1]), static_cast<int>(q1[2]), static_cast<int>(q1[3])};
          sacc[hh][tb] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(kv8, qv8, f32x4{0.f, 0.f, 0.f, 0.f}, 0, 0, 0, 0, 0, 0);
        }
      }
      // online softmax in base 2 over the 64 tokens; lane (n, g) holds tokens 16 tb + 4 g + r of q row 16 hh + n.  P~ = e4m3(256 p)
      // as in the pair form.
      uint32_t pf[2][4];
#pragma unroll
      for (int hh = 0; hh < 2; ++hh) {
        const float rsc = row_scale[hh];
#pragma unroll
        for (int tb = 0; tb < 4; ++tb) {
          f32x4 kt = f32x4{1.f, 1.f, 1.f, 1.f};
          if constexpr (kKtok) kt = *reinterpret_cast<const f32x4*>(&s_ks[wave][tb * 16 + g * 4]);
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            sacc[hh][tb][r] *= rsc;
            if constexpr (kKtok) sacc[hh][tb][r] *= kt[r];
          }
        }
      }
      if (d_fl & kFMasked) {  // wave-uniform: only the WIs that hold a request's last tokens
#pragma unroll
        for (int hh = 0; hh < 2; ++hh) {
          const int sq_row = (16 * hh + n) >> a.g_shift;
          const int lim = (q0_end - 1 < q0_ltot - Sq + sq_row) ? q0_end - 1 : q0_ltot - Sq + sq_row;
#pragma unroll
          for (int tb = 0; tb < 4; ++tb)
#pragma unroll
            for (int r = 0; r < 4; ++r) sacc[hh][tb][r] = (d_tok + tb * 16 + g * 4 + r) <= lim ? sacc[hh][tb][r] : kNegInf;
        }
      }
#pragma unroll
      for (int hh = 0; hh < 2; ++hh) {
        float mt = kNegInf;
#pragma unroll
        for (int tb = 0; tb < 4; ++tb)
#pragma unroll
          for (int r = 0; r < 4; ++r) mt = __builtin_fmaxf(mt, sacc[hh][tb][r]);
        mt = row4_max(mt);
        const float m_new = fmaxf(m_run[hh], mt);
        const float m_use = m_new == kNegInf ? 0.f : m_new;
        const float m8 = m_use - 8.0f;
        float psum = 0.f;
#pragma unroll
        for (int tb = 0; tb < 4; ++tb) {
          float prb[4];
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            prb[r] = __builtin_amdgcn_exp2f(sacc[hh][tb][r] - m8);
            psum += prb[r];
          }
          int w = __builtin_amdgcn_cvt_pk_fp8_f32(prb[0], prb[1], 0, false);
          w = __builtin_amdgcn_cvt_pk_fp8_f32(prb[2], prb[3], w, true);
          pf[hh][tb] = static_cast<uint32_t>(w);
        }
        if (__builtin_amdgcn_ballot_w64(m_new != m_run[hh]) != 0) {
          const float alpha = __builtin_amdgcn_exp2f(m_run[hh] - m_use);
          l_run[hh] *= alpha;
#pragma unroll
          for (int jj = 0; jj < 8; ++jj) o[hh][jj] *= alpha;
          m_run[hh] = m_new;
        }
        l_run[hh] += psum;
      }
      // O^T += V^T P^T: two K = 32 steps (tokens 32 s ... 32 s + 31); a step's eight transposing reads feed both halves
#pragma unroll
      for (int st = 0; st < 2; ++st) {
        v2i32 vt[8];
#pragma unroll
        for (int u = 0; u < 8; ++u)
          vt[u] = __builtin_amdgcn_ds_read_tr8_b64_v2i32(
              reinterpret_cast<lds_v2i32*>(static_cast<uint32_t>((t0 ^ (u * 16)) + st * 32 * kRowB)));
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int hh = 0; hh < 2; ++hh)
#pragma unroll
          for (int u = 0; u < 8; ++u)
            o[hh][u] = __builtin_amdgcn_mfma_f32_16x16x32_fp8_fp8(
                pack64(static_cast<uint32_t>(vt[u][0]), static_cast<uint32_t>(vt[u][1])), pack64(pf[hh][2 * st], pf[hh][2 * st + 1]),
                o[hh][u], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
      }
    } else {
      // S^T = K Q^T: one K = 128 MFMA per 16-row block and 128-byte half (lane (n, g) supplies chunks g and g + 4 of its
      // row on both sides - a dot product does not care which lane slot a dim sits in)
      f32x4 sacc[2][2];
  #pragma unroll
      for (int tb = 0; tb < 2; ++tb)
  #pragma unroll
        for (int hh = 0; hh < 2; ++hh) {
          const u32x4 k0 = *reinterpret_cast<const lds_u32x4*>(static_cast<uint32_t>((r0 ^ (((hh << 3) ^ (tb << 3)) * 16)) + tb * 16 * kRowB));
          const u32x4 k1 = *reinterpret_cast<const lds_u32x4*>(static_cast<uint32_t>((r0 ^ (((hh << 3) ^ (tb << 3) ^ 4) * 16)) + tb * 16 * kRowB));
          const u32x4 q0 = qf[hh][0], q1 = qf[hh][1];
          const i32x8 kv8 = {static_cast<int>(k0[0]), static_cast<int>(k0[1]), static_cast<int>(k0[2]), static_cast<int>(k0[3]),
                             static_cast<int>(k1[0]), static_cast<int>(k1[1]), static_cast<int>(k1[2]), static_cast<int>(k1[3])};
          const i32x8 qv8 = {static_cast<int>(q0[0]), static_cast<int>(q0[1]), static_cast<int>(q0[2]), static_cast<int>(q0[3]),
                             static_cast<int>(q1[0]), static_cast<int>(q1[1]), static_cast<int>(q1[2]), static_cast<int>(q1[3])};
          sacc[hh][tb] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(kv8, qv8, f32x4{0.f, 0.f, 0.f, 0.f}, 0, 0, 0, 0, 0, 0);
        }

      // online softmax in base 2; lane (n, g) holds rows 16 tb + 4 g + r of q row n.
      // P~ = e4m3(256 p) is computed as exp2(x - m + 8) (<= 256 < 448: no clamp needed) and the row sum is kept in
      // the same units (l = 256 sum p; the 1/256 of the reference formula is folded into the final scale).
      uint32_t pf[2][2];
  #pragma unroll
      for (int hh = 0; hh < 2; ++hh) {
        const float rsc = row_scale[hh];
  #pragma unroll
        for (int tb = 0; tb < 2; ++tb) {
          f32x4 kt = f32x4{1.f, 1.f, 1.f, 1.f};
          if constexpr (kKtok) kt = *reinterpret_cast<const f32x4*>(&s_ks[wave][hh * 32 + tb * 16 + g * 4]);
  #pragma unroll
          for (int r = 0; r < 4; ++r) {
            sacc[hh][tb][r] *= rsc;
            if constexpr (kKtok) sacc[hh][tb][r] *= kt[r];
          }
        }
      }
      if (d_fl & kFMasked) {  // wave-uniform: only the WIs that hold a request's last tokens
        const int sq_row = n >> a.g_shift;
        const int lim = (q0_end - 1 < q0_ltot - Sq + sq_row) ? q0_end - 1 : q0_ltot - Sq + sq_row;
  #pragma unroll
        for (int hh = 0; hh < 2; ++hh)
  #pragma unroll
          for (int tb = 0; tb < 2; ++tb)
  #pragma unroll
            for (int r = 0; r < 4; ++r) {
              const int row = tb * 16 + g * 4 + r;
              const int tk = d_tok + row;
              sacc[hh][tb][r] = tk <= lim ? sacc[hh][tb][r] : kNegInf;
            }
      }
  #pragma unroll
      for (int hh = 0; hh < 2; ++hh) {
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
          float prb[4];
  #pragma unroll
          for (int r = 0; r < 4; ++r) {
            prb[r] = __builtin_amdgcn_exp2f(sacc[hh][tb][r] - m8);
            psum += prb[r];
          }
          int w = __builtin_amdgcn_cvt_pk_fp8_f32(prb[0], prb[1], 0, false);
          w = __builtin_amdgcn_cvt_pk_fp8_f32(prb[2], prb[3], w, true);
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

      // O^T += V^T P^T: the transpose read hands every lane one dim (column) of an 8-row x 16-dim tile.
      constexpr int kTrBatch = 8;
      // Reads go out eight at a time ahead of their MFMAs (hipcc, left alone, reuses one register pair and
      // serialises read -> lgkmcnt(0) -> MFMA sixteen times: sixteen exposed LDS round trips per WI).
  #pragma unroll
      for (int hh = 0; hh < 2; ++hh)
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
    }
    if constexpr (kProf) {
      asm volatile("s_nop 0" ::"v"(o[0][0]), "v"(o[1][7]));  // the MFMAs have been issued (not retired)
      const uint64_t t = now();
      pf_cp += t - pf_last;
      pf_last = t;
      ++pf_n;
    }
    if (d_fl & kFLast) {
      finish_task();
      if constexpr (kProf) { const uint64_t t = now(); pf_fin += t - pf_last; pf_last = t; }
    }
  }
  if constexpr (kProf) {
    if (lane == 0 && a.prof) {
      const uint64_t t1 = now(), r1 = __builtin_amdgcn_s_memrealtime();
      uint64_t* dst = reinterpret_cast<uint64_t*>(a.prof) + (static_cast<long>(wg) * kWaves + wave) * 12;
      dst[0] = pf_r0, dst[1] = r1, dst[2] = t1 - pf_t0, dst[3] = pf_wk, dst[4] = pf_wv, dst[5] = pf_wr, dst[6] = pf_is, dst[7] = pf_cp,
      dst[8] = pf_fin, dst[9] = pf_n, dst[10] = __builtin_amdgcn_s_getreg((4 << 0) | (0 << 6) | (31 << 11)) /* HW_ID */,
      dst[11] = (static_cast<uint64_t>(__builtin_amdgcn_s_getreg((20 << 0) | (0 << 6) | (31 << 11)) /* XCC_ID */) << 32) |
                (static_cast<uint64_t>(rng) << 8) | static_cast<uint64_t>(pr);
    }
  }
  flush_pending();
  // the loads issued for the WI past the end were no-ops, but they own the registers until they retire
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}

// Scratch layout of this kernel inside the call's workspace: the arrival counters sit in the first kCounterBytes of
// the WHOLE workspace (fixed place and size whatever the shapes of the call, so a call with other shapes can never
// land them on stale partials); partial slots follow the first-generation kernel's region.
int64_t workspace_bytes(int num_wg) {
  const int64_t part_o = static_cast<int64_t>(num_wg) * 2 * 2 * 16 * 128 * 4;
  const int64_t part_lse = static_cast<int64_t>(num_wg) * 2 * 2 * 16 * 4;
  return part_o + part_lse;
}

int mode_of(Args& a, int num_head_q, int block_size, int64_t k_head_stride, int64_t v_head_stride) {
  if (a.num_head_kv <= 0 || num_head_q % a.num_head_kv) return 0;
  const int group = num_head_q / a.num_head_kv;
  const int head_bytes = a.bf16 ? 256 : 128;  // strides in BYTES: adjacent kv heads of a token must be contiguous (NHD pages)
  // ... or (fp8, per-tensor scales) a head's tokens: HND pages - development key 55 = 1 only.  Measured (profiles/round6_decode_ab.txt,
  // call 3): against the first-generation kernel the HND form wins on length mixes (C3 mix 137.3 vs 141.2 us, 32 x 128 + 32 x 4k
  // 57.8 vs 60.6) and loses where the task map gives every workgroup one whole (request, head) and this kernel's plan cuts every
  // request in two (uniform 8k: 181.7 us = 0.74 against 164-178 us = 0.75-0.82) - and 1 KB contiguous pieces stream no faster
  // through this pipeline than the NHD form's 256-byte slices do (0.74 vs 0.76 on uniform 8k): the access pattern is not what
  // bounds it.  HND pages stay on the first generation.
  a.hnd = 0;
  a.k_head_stride = k_head_stride;
  a.v_head_stride = v_head_stride;
  // One kv head per workgroup with up to 32 q rows (kSolo, mode 3): speculative steps with 17 ... 32 q rows per kv head (round 6).
  // Any head / token strides (NHD and HND pages), any head count, pages of 32 / 64 tokens.  Measured against the first generation's
  // two-block form (profiles/round6_decode_ab.txt, call 8; C3 lengths, us): num_seq_q 3, 8 / 64 heads NHD mix 195.6 -> 163.8,
  // uniform 8k 220 -> 196; HND 193 -> 150 / 218 -> 180; num_seq_q 4: 4 / 32 heads 112 -> 87.5 / 112 -> 95, 1 / 8 heads 49.5 -> 41.4.
  // Development key 60: 1 = never (rounds 1-5), 2 = also every other fp8 call with per-tensor scales that the pair form does not
  // take (<= 16 q rows on HND pages or with an odd head count) - there the first generation stays ahead (one kv head, 8 q rows:
  // 31 against 38-40 us; HND mix 138 / 139, 32 x 128 + 32 x 4k 63 against 71 us).
  {
    const int rows = a.num_seq_q * group, k60 = hpc_dev_tuning_get(60);
    const bool pair_case = (a.num_head_kv % 2) == 0 && rows <= 16 && k_head_stride == head_bytes && v_head_stride == head_bytes &&
                           (!a.ktok || a.ks_head_stride == 128);
    const bool ks_ok = !a.ktok || (a.ks_block_stride > 0 && a.ks_block_stride < (1ll << 32) && (a.ks_row_stride % 4) == 0 &&
                                   (a.ks_head_stride % 4) == 0);
    const bool shape_ok = ks_ok && a.lens != nullptr && rows <= 32 && (block_size == 64 || block_size == 32 || (a.bf16 && block_size == 16)) &&
                          (a.k_token_stride % 16) == 0 && (a.v_token_stride % 16) == 0 && (k_head_stride % 16) == 0 &&
                          (v_head_stride % 16) == 0 && (a.k_block_stride % 16) == 0 && (a.v_block_stride % 16) == 0 &&
                          a.k_block_stride > 0 && a.v_block_stride > 0 && a.k_block_stride < (1ll << 32) &&
                          a.v_block_stride < (1ll << 32) && a.k_token_stride * 32 < (1ll << 31) && a.v_token_stride * 32 < (1ll << 31) &&
                          a.num_batch <= 64 * 16 && static_cast<int64_t>(a.num_batch) * a.num_head_kv * 4 <= kCounterBytes;
    const bool wanted = k60 == 3 ? true : k60 == 2 ? !pair_case : (k60 == 0 && rows > 16);  // 3: every eligible call (A/B against the pair forms)
    if (shape_ok && wanted) return 3;
  }
  const bool hnd = !a.bf16 && !a.ktok && a.k_token_stride == 128 && a.v_token_stride == 128 && k_head_stride >= 128 * block_size &&
                   v_head_stride >= 128 * block_size && (k_head_stride % 16) == 0 && (v_head_stride % 16) == 0 &&
                   k_head_stride < (1ll << 28) && v_head_stride < (1ll << 28) && a.num_head_kv > 1 && hpc_dev_tuning_get(55) == 1;
  if (hnd) {
    a.hnd = 1;
    k_head_stride = v_head_stride = head_bytes;  // the checks below are the NHD form's
  }
  const bool ok = a.lens != nullptr && (a.num_head_kv % 2) == 0 && a.num_seq_q * group <= 16 && k_head_stride == head_bytes &&
                  v_head_stride == head_bytes && (block_size == 64 || block_size == 32 || block_size == 16) &&
                  (a.k_token_stride % 16) == 0 && (a.v_token_stride % 16) == 0 && (a.k_block_stride % 16) == 0 &&
                  (a.v_block_stride % 16) == 0 && a.k_block_stride > 0 && a.v_block_stride > 0 &&
                  a.k_block_stride < (1ll << 32) && a.v_block_stride < (1ll << 32) && a.num_batch <= 64 * 16 &&
                  static_cast<int64_t>(a.num_batch) * (a.num_head_kv / 2) * 4 <= kCounterBytes;
  if (!ok) {
    a.hnd = 0;
    return 0;
  }
  // per-token K scales (quant_type 0): a wave-iteration's 32 tokens x 2 heads of scales must be one contiguous 256-byte piece
  // of a page's tail row - pages of 32 / 64 tokens, 32 floats per head and row, adjacent heads 128 bytes apart
  if (a.ktok && (a.bf16 || block_size < 32 || a.ks_head_stride != 128 || a.ks_block_stride <= 0 || a.ks_block_stride >= (1ll << 32) ||
                 (a.ks_row_stride % 4) != 0))
    return 0;
  if (a.ktok) return 1;
  // fp8 with <= 8 q rows per kv head and a multiple of 4 kv heads can run four heads per workgroup (kQuad).  Measured
  // 3-5 % SLOWER than head pairs on the graded shapes (uniform 8k 188.8 vs 183.0 us, C3 mix 145.4 vs 138.6 us, same box,
  // profiles/round3_decode_fp8_forms_ab.txt): the wider rows do not pay in the kernel although they do in a pure streaming
  // probe - the waves sit in the load issue either way (tools/prof_decode.py: 52-57 % of a wave's cycles).  So head
  // pairs stay the default and development key 29 = 2 selects the four-head form (kept: tested, half the softmax work).
  const bool quad = !a.bf16 && !a.hnd && (a.num_head_kv % 4) == 0 && a.num_seq_q * group <= 8 && hpc_dev_tuning_get(29) == 2;
  return quad ? 2 : 1;
}

#ifdef HPC_DEV
int ticket_overruns(bool reset) {
  int v = 0;
  if (hipMemcpyFromSymbol(&v, HIP_SYMBOL(g_ticket_overruns), sizeof(int)) != hipSuccess) return -1;
  const int zero = 0;
  if (reset && hipMemcpyToSymbol(HIP_SYMBOL(g_ticket_overruns), &zero, sizeof(int)) != hipSuccess) return -1;
  return v;
}
#endif

int launch(Args a, void* counters, void* partials, int num_wg, int mode, hipStream_t stream) {
  char* ws = static_cast<char*>(partials);
  a.part_o = reinterpret_cast<float*>(ws);
  ws += static_cast<int64_t>(num_wg) * 2 * 2 * 16 * 128 * 4;
  a.part_lse = reinterpret_cast<float*>(ws);
  a.arrive = static_cast<int*>(counters);
  const bool temporal = hpc_dev_tuning_get(0) == 1;
  if (mode == 3) {  // one kv head per workgroup
    a.dev_slice = a.xcd_map = a.dev_sleep = 0;
    a.pair_xor = a.mate_from = 0;
    a.pair_wgs[0] = a.pair_wgs[1] = a.pair_wgs[2] = a.pair_wgs[3] = 0;
    a.big_pct = 100;
    {  // the pair form's CU-mate rule (a CU's two workgroups on slices across an address bit) makes no difference here - masks 1, 2,
       // 4, 6 on 8 heads: C3 mix 160.6-164.3 us with and without, call 8 - so it is off; development key 36 = mask + 1 switches it on
      int dev = 0;
      const int cus = hipGetDevice(&dev) == hipSuccess ? hpc_get_cu_count(dev) : 0;
      const int k36 = hpc_dev_tuning_get(36);
      const int mask = k36 > 0 ? k36 - 1 : 0;
      const int np = a.num_head_kv;
      if (cus > 0 && num_wg > cus && (np & (np - 1)) == 0 && mask > 0 && mask < np && cus % np == 0) {
        a.pair_xor = mask;
        a.mate_from = cus;
      }
    }
    if (a.bf16)
      decode2_kernel<2, true, false, false, false, false, true><<<num_wg, kThreads, 0, stream>>>(a);
    else if (a.ktok)
      decode2_kernel<2, false, false, false, true, false, true><<<num_wg, kThreads, 0, stream>>>(a);
    else if (kHpcDevBuild && a.prof)
      decode2_kernel<2, false, true, false, false, false, true><<<num_wg, kThreads, 0, stream>>>(a);
    else
      decode2_kernel<2, false, false, false, false, false, true><<<num_wg, kThreads, 0, stream>>>(a);
    if (hipGetLastError() != hipSuccess) {
      (void)hipMemsetAsync(counters, 0, kCounterBytes, stream);
      (void)hipGetLastError();
      return HPC_ERR_LAUNCH;
    }
    return HPC_OK;
  }
  a.dev_slice = hpc_dev_tuning_get(37);
  a.xcd_map = hpc_dev_tuning_get(38);
  a.dev_sleep = hpc_dev_tuning_get(39);
  a.dev_merge_dup = hpc_dev_tuning_get(58) == 1;
  a.dev_nosnap = hpc_dev_tuning_get(61) == 1;
  // the second workgroup of every CU on the slice across address bit 9 (see the kernel): slices of 256 B (fp8 pairs) -> pair
  // index bit 1, of 512 B (bf16 pairs, fp8 quads) -> bit 0.  Measured per shape (profiles/round5_decode_pair_map_ab.txt):
  // fp8 8 / 64 heads +4-6 %, 16 / 128 heads +4 % (bit 8: +2 %, bit 10: 0), bf16 8 / 64 +1-2.5 % (bit 10: 0); with two pairs
  // (4 kv heads, fp8) only bit 8 exists: +2 % on the length mix, +-1 % on uniform lengths - taken.  Needs a power-of-two pair
  // count that holds the bit, more than one workgroup per CU, whole rows of pairs in front of the second workgroups, equal
  // shares.  Development key 36 = mask + 1 overrides the mask (1 = off: both workgroups of a CU on the same slice).
  {
    int dev = 0;
    const int npair = a.num_head_kv / (mode == 2 ? 4 : 2);
    const int cus = hipGetDevice(&dev) == hipSuccess ? hpc_get_cu_count(dev) : 0;
    const int k36 = hpc_dev_tuning_get(36);
    int mask = (a.bf16 || mode == 2 || npair == 2) ? 1 : 2;
    if (k36 > 0) mask = k36 - 1;
    // (HND pages: a workgroup walks whole 1 KB pieces of a head - no slice fixes address bits 8-9, the rule has nothing to separate)
    const bool ok = cus > 0 && num_wg > cus && (npair & (npair - 1)) == 0 && mask > 0 && mask < npair && cus % npair == 0 &&
                    (!a.hnd || k36 > 0);
    a.pair_xor = ok ? mask : 0;
    a.mate_from = ok ? cus : 0;
  }
  // Unequal shares for the four 256-byte slices of an 8-kv-head fp8 row (see the kernel): slice 1 gets d1 more
  // workgroups than the even share, slice 3 d3, slices 0 and 2 give them up.  Off by default: measured +3 % on the C3 mix
  // at d1 = 12 of 128 and nothing on uniform 8k (where the even split puts exactly one request half on every
  // workgroup); development keys 30 / 31 set the deltas (value - 100, per 128 workgroups of a slice).
  a.pair_wgs[0] = a.pair_wgs[1] = a.pair_wgs[2] = a.pair_wgs[3] = 0;
  if (!a.bf16 && mode == 1 && a.num_head_kv == 8 && num_wg >= 256 && num_wg % 4 == 0) {
    const int k30 = hpc_dev_tuning_get(30), k31 = hpc_dev_tuning_get(31);
    const int even = num_wg / 4;
    int d1 = (k30 != 0 ? k30 - 100 : 0) * even / 128, d3 = (k31 != 0 ? k31 - 100 : 0) * even / 128;
    if (d1 != 0 || d3 != 0) {
      const int give = d1 + d3, g0 = give / 2, g2 = give - g0;
      a.pair_wgs[0] = even - g0, a.pair_wgs[1] = even + d1, a.pair_wgs[2] = even - g2, a.pair_wgs[3] = even + d3;
    }
  }
  if (a.pair_wgs[0] > 0) a.pair_xor = a.mate_from = 0;  // unequal shares (development) re-map the tail of the grid themselves (both together: measured worse)
  // longer ranges for the first workgroup of every CU (see the kernel): only when the grid is exactly two workgroups
  // per CU, so that "first half of the grid" means "first on its CU".  Development key 32: percentage (0 = off).
  // Measured: no gain at 108 / 115 / 122 % (C3 mix 139.7 / 139.0 / 138.4 us vs 139.3 us) - when a CU's first workgroup
  // ends early its second one speeds up, and the chip's aggregate rate does not change: the spread of finish times
  // (first halves 119-143 us, second halves 142-169 us on uniform 8k) is not where the time goes.  Off.
  {
    int dev = 0;
    const int k32 = hpc_dev_tuning_get(32);
    a.big_pct = 100;
    if (k32 > 100 && k32 <= 200 && hipGetDevice(&dev) == hipSuccess && num_wg == 2 * hpc_get_cu_count(dev) &&
        (num_wg / 2) % (a.num_head_kv / (mode == 2 ? 4 : 2)) == 0)
      a.big_pct = k32;
  }
  if (a.ktok) {
    decode2_kernel<2, false, false, false, true><<<num_wg, kThreads, 0, stream>>>(a);
  } else if (a.hnd) {
    if (kHpcDevBuild && a.prof)
      decode2_kernel<2, false, true, false, false, true><<<num_wg, kThreads, 0, stream>>>(a);
    else
      decode2_kernel<2, false, false, false, false, true><<<num_wg, kThreads, 0, stream>>>(a);
  } else if (a.bf16) {
    if (temporal)
      decode2_kernel<0, true><<<num_wg, kThreads, 0, stream>>>(a);
    else
      decode2_kernel<2, true><<<num_wg, kThreads, 0, stream>>>(a);
  } else if (kHpcDevBuild && mode == 2) {
    if (a.prof)  // development: per-wave s_memtime sums (hpc_dev_decode_prof_buffer)
      decode2_kernel<2, false, true, true><<<num_wg, kThreads, 0, stream>>>(a);
    else if (temporal)
      decode2_kernel<0, false, false, true><<<num_wg, kThreads, 0, stream>>>(a);
    else
      decode2_kernel<2, false, false, true><<<num_wg, kThreads, 0, stream>>>(a);
  } else if (kHpcDevBuild && a.prof) {  // development: per-wave s_memtime sums (hpc_dev_decode_prof_buffer)
    decode2_kernel<2, false, true><<<num_wg, kThreads, 0, stream>>>(a);
  } else if (temporal) {
    decode2_kernel<0><<<num_wg, kThreads, 0, stream>>>(a);
  } else {
    decode2_kernel<2><<<num_wg, kThreads, 0, stream>>>(a);
  }
  if (hipGetLastError() != hipSuccess) {
    // the contract of the counter region is "zero on entry, zero on exit": a launch that was refused leaves it as it
    // found it, but the caller cannot tell a refused launch from one that died half way - restore the invariant in
    // stream order before reporting (best effort: the stream may be beyond repair)
    (void)hipMemsetAsync(counters, 0, kCounterBytes, stream);
    (void)hipGetLastError();
    return HPC_ERR_LAUNCH;
  }
  return HPC_OK;
}

}  // namespace decode2
}  // namespace hpc
