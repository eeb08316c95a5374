// Decode-attention dynamic tile scheduler: task-map "wire format" shared by the scheduler
// (assign_task.hip) and its consumers (decode attention + split-KV combine kernels).
//
// Same layout as the reference (src/attention/decode/sched_task_info.h:18-31 and the workspace
// math in hpc/attention.py:540-582), so a task map is interchangeable at the byte level:
//   int32 rows of 12 ("records", 48 bytes):
//   row 0                      header: [0]=tiles_per_bin+1  [1]=num_bins  [2]=num_head_kv
//                                      [3]=max_batch  [4]=scheduler bytes  [5]=max chunks of any (h,b)
//                                      [6]=min_process_len of the scheduler call (ours; the reference leaves it 0):
//                                          the head-pair decode kernels (attention_decode_v2.hip) plan in closed
//                                          form inside the launch and read it so that the caller's lower bound on
//                                          the KV length one workgroup processes holds on that path too
//   rows 1 + bin*(T+1) + i     task i of bin `bin` (T = tiles_per_bin); list ends at ihead_kv<0
//   then num_chunks[h*B + b]   (padded to a multiple of 12 ints using max_batch*Hkv)
//   then finish flags          pad12(num_bins) ints   (unused by this implementation, kept zero)
//   then tasks-per-bin         pad12(num_bins) ints
//
// What differs from the reference is the MI355X choice of bins: num_bins = CUs * kCtaPerCu[Sq-1]
// (reference: SMs * kCtaPerSmMap, sched_task_info.h:35-36); consumers read it from header[1].
#pragma once
#include <stdint.h>

namespace hpc {
namespace sched {

constexpr int kTaskStride = 12;  // ints per record
constexpr int kTileN = 64;       // KV tokens per scheduling tile (reference sm90 value)
constexpr int kMaxSeqQ = 5;

// workgroups ("bins") per CU by num_seq_q (1..5): 256-thread workgroups, <=256 VGPRs each.
__host__ __device__ inline int cta_per_cu(int num_seq_q) {
  return num_seq_q <= 2 ? 2 : 1;
}

// MI355X choice (the reference plans every batch on all of its CTAs): a batch with little work is planned on FEWER bins,
// so that a bin holds at least kMinTilesPerBin tiles - never fewer than kMinBins bins, never more than the launch has.  With
// one tile per bin a 16k-token request among a few short ones is cut into 256 chunks whose fp32 partials the last
// arriver then merges one after the other (reference benchmark case skewed_extreme, one kv head: 31 us; 64 uniform
// 512-token requests would be cut into 8 chunks each instead of running whole).  The device scheduler and the CPU
// entry of the op apply it (header int 1 records the count they used; consumers read it from there); the C function
// hpc_assign_attention_decode_task_sync plans on exactly the bin count its caller passes, like the reference's.
constexpr int kMinTilesPerBin = 8;
constexpr int kMinBins = 64;
__host__ __device__ inline int effective_bins(long grand_tiles, int num_bins) {
  long want = (grand_tiles + kMinTilesPerBin - 1) / kMinTilesPerBin;
  if (want < kMinBins) want = kMinBins;
  return want < num_bins ? static_cast<int>(want) : num_bins;
}

struct alignas(16) TaskInfo {
  int ihead_kv, ibatch, ichunk, iseq_start;
  int num_seqkv, num_seqkvcache, num_tile_kv, num_tile_full;
  int is_casual_chunk, pad[3];
};
static_assert(sizeof(TaskInfo) == 48, "TaskInfo must be 48 bytes");

__host__ __device__ inline int pad12(int n) { return (n + kTaskStride - 1) / kTaskStride * kTaskStride; }

// int offsets inside the task map
__host__ __device__ inline long chunk_table_off(int tiles_per_bin, int num_bins) {
  return static_cast<long>(kTaskStride) * (static_cast<long>(tiles_per_bin + 1) * num_bins + 1);
}

}  // namespace sched
}  // namespace hpc
