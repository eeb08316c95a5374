// Interface between the decode-attention C entry (attention_decode.hip) and the second-generation
// FP8 / NHD kernel (attention_decode_v2.hip).  Internal header.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace hpc {
namespace decode2 {

struct Args {
  const void* q;
  const void* kcache;
  const void* vcache;
  const int* block_ids;
  const int* lens;       // num_seq_kvcache [B]
  const int* task_map;   // the scheduler's task map: only header int 6 (min_process_len) is read - the plan is in-kernel
  uint16_t* y;
  float* part_o;         // [workgroups][2][2 heads][16][128]
  float* part_lse;       // [workgroups][2][2 heads][16]
  int* arrive;           // [pairs * B] arrival counters of split requests (zero before the call, left zero)
  const float* qscale;   // [B * Sq, qscale_stride]
  const float* kscale;   // [1] or the K-scale tail rows of the cache
  const float* vscale;   // [1] or [Hkv]
  int num_batch, num_seq_q, num_head_kv, g_shift, page_shift, max_blocks;
  int ldq, ldy, qscale_stride, new_kv_included;
  int min_range_cost;  // smallest range of the in-kernel plan, in cost units (64-token tiles + 2 per request)
  int bf16;       // 1: bf16 q / K / V (no scales; ldq and every stride in BYTES), 0: fp8 e4m3
  int ktok = 0;   // fp8: 1 = per-token K scales in the pages' tail rows (kscale + ks_* strides) and per-head V scales (quant_type 0)
  int pair_wgs[4];  // fp8, 4 head pairs: workgroups (= ranges) per pair, [0] = 0: equal shares
  int big_pct;      // > 100: ranges of the first half of the grid are this many percent of the others' length
  int dev_nomem;  // development key 15 = 1: K / V loads fetch nothing (compute-only timing; results are wrong)
  int pair_xor;   // workgroups from mate_from on (the SECOND workgroup of every CU) serve head pair p ^ pair_xor: a CU's two
  int mate_from;  // workgroups stream slices of the token rows that differ in byte-address bit 9 (see the kernel); 0 = off
  int xcd_map;    // development key 38: eight 3-bit entries, workgroup 8 j + x serves pair (e & 3) of range 2 j + (e >> 2), e = entry x
  int dev_sleep;  // development key 39: workgroups of even head pairs sleep this many x 64 clocks per wave-iteration
  int dev_merge_dup = 0;  // development key 58 = 1: the last arriver loads (and folds with weight zero) a duplicate for a missing second chunk
  int dev_nosnap = 0;     // development key 61 = 1: short requests of an underloaded launch may be split (rounds 2-5)
  int dev_slice;  // development key 37 = s + 1: every workgroup streams slice s of the token rows (timing only; results are wrong)
  long k_block_stride, k_token_stride;  // bytes
  long v_block_stride, v_token_stride;
  long ks_block_stride, ks_row_stride, ks_head_stride;  // bytes
  int hnd = 0;    // fp8: 1 = HND pages [page][head][token][128 B] (k / v_head_stride below; token strides are 128)
  long k_head_stride = 0, v_head_stride = 0;  // bytes (read by the HND form only)
  float scale_log2;
  void* prof;  // development: per-wave timing sums [workgroups][4][12] uint64 (null = off)
};

// Arrival counters of split requests: a fixed region at the very start of a call's workspace.  It must be zero
// the first time a workspace is used (the kernel leaves it zero); its place and size do not depend on the call.
constexpr int64_t kCounterBytes = 64 * 1024;
int64_t workspace_bytes(int num_wg);  // partial slots (2 per workgroup x 2 heads), after the first-generation region
// 3: one kv head per workgroup (fp8, per-tensor scales, 17 ... 32 q rows per kv head, any page layout, pages of 32 / 64 tokens);
// 0: not served here; 1: served (NHD pages with adjacent heads contiguous - 128 B apart for fp8, 256 B for bf16 -, or, fp8
// with per-tensor scales and development key 55 = 1, HND pages with a head's tokens contiguous (a.hnd is set then); an even number of kv heads,
// <= 16 q rows per kv head, <= 1024 requests).
int mode_of(Args& a, int num_head_q, int block_size, int64_t k_head_stride, int64_t v_head_stride);
int launch(Args a, void* counters, void* partials, int num_wg, int mode, hipStream_t stream);
#ifdef HPC_DEV
// development build: arrivals that drew a ticket ABOVE their request's chunk count since the last reset - the signature of
// an arrival counter that was not zero on entry (the zero-bytes contract of the workspace was broken: the last arriver is
// never recognised and the request's rows of y stay unwritten).  Synchronises the device.
int ticket_overruns(bool reset);
#endif

}  // namespace decode2
}  // namespace hpc
