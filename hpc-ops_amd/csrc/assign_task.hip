// Dynamic split-KV tile scheduler for decode attention (host + gfx950 device implementations).
//
// Replaces reference src/attention/decode/assign_task.cu:
//   assign_attention_decode_task_kernel (:41-329)  -> assign_task_kernel below
//   assign_attention_decode_task_sync   (:362-492) -> hpc_assign_attention_decode_task_sync
// and the packing of the CPU entry (src/attention/entry.cc:727-778).
//
// The reference GPU kernel replays every earlier bin serially in thread 0 of each CTA and
// hand-shakes neighbouring CTAs with spin flags.  Here the schedule is restated in closed form:
// requests are laid end to end in (head, batch) order on a global tile axis, bin i owns tiles
// [i*T, (i+1)*T) (T = tiles per bin), so a bin is found with one binary search over the per-batch
// prefix sum and every bin is planned independently - one lane per bin, no inter-workgroup
// communication, no spinning.  The "Q rows overflow into the previous task" fix-up
// (assign_task.cu:451-466, :250-262) becomes a pure look-ahead at the next task.
// The SAME __host__ __device__ planner runs on CPU and GPU, so both paths are byte-identical by
// construction; the result layout is the reference's (csrc/sched_task_info.h).
#include <hip/hip_runtime.h>

#include <vector>

#include "../../include/hpc_amd.h"
#include "hpc_common.h"
#include "hpc_dev.h"
#include "sched_task_info.h"

namespace hpc {
namespace sched {

struct Plan {
  const int* seqkv;  // [B] total KV tokens per request (incl. the new ones)
  const int* tiles;  // [B] ceil(seqkv / tilen)
  const int* cum;    // [B] exclusive prefix sum of tiles
  int total;         // tiles per head = sum(tiles)
  int num_batch, num_head_kv, num_seq_q, tilen, per, num_bins;
};

__host__ __device__ inline int imin(int a, int b) { return a < b ? a : b; }
__host__ __device__ inline int imax(int a, int b) { return a > b ? a : b; }

// largest b with cum[b] <= r and tiles[b] > 0 covering tile r (0 <= r < total)
__host__ __device__ inline int find_batch(const Plan& p, int r) {
  int lo = 0, hi = p.num_batch;  // first index with cum > r
  while (lo < hi) {
    const int mid = (lo + hi) >> 1;
    if (p.cum[mid] <= r) lo = mid + 1; else hi = mid;
  }
  return lo - 1;
}

// negative Q-row overflow of the task that FOLLOWS a task ending at (h, b, next_start_tile) whose
// bin still has `bucket_left` tiles of room; 0 when there is none.
__host__ __device__ inline int next_task_overflow(const Plan& p, int h, int b, int next_start_tile,
                                                  int bucket_left) {
  int left = p.tiles[b] - next_start_tile;
  int start = next_start_tile;
  if (left <= 0) {  // the follower is chunk 0 of the next non-empty request
    start = 0;
    do {
      if (++b >= p.num_batch) {
        b = 0;
        if (++h >= p.num_head_kv) return 0;
      }
    } while (p.tiles[b] <= 0);
    left = p.tiles[b];
  }
  const int room = bucket_left > 0 ? bucket_left : p.per;
  const int add = imin(left, room);
  if (add != left) return 0;  // follower is not a last chunk
  const int seqkv = imin(add * p.tilen, p.seqkv[b] - start * p.tilen);
  return imin(seqkv - p.num_seq_q, 0);
}

// Plans bin `ibin`: writes its task records and returns the number of tasks.  The terminators of
// the unused slots are written by the caller (host: plan_bin; device: 8 lanes per bin).
__host__ __device__ inline int plan_bin_tasks(const Plan& p, int ibin, int* bin_ptr) {
  const long grand = static_cast<long>(p.total) * p.num_head_kv;
  long g = static_cast<long>(ibin) * p.per;
  const long end = g + p.per < grand ? g + p.per : grand;
  int itask = 0;
  if (g < end) {
    int h = static_cast<int>(g / p.total);
    int b = find_batch(p, static_cast<int>(g % p.total));
    while (g < end) {
      const long g0 = static_cast<long>(h) * p.total + p.cum[b];  // first tile of (h, b)
      const int start_tile = static_cast<int>(g - g0);
      const int left = p.tiles[b] - start_tile;
      const int add = imin(left, static_cast<int>(end - g));
      TaskInfo t;
      t.ihead_kv = h;
      t.ibatch = b;
      t.ichunk = ibin - static_cast<int>(g0 / p.per);
      t.iseq_start = start_tile * p.tilen;
      t.num_seqkv = imin(add * p.tilen, p.seqkv[b] - t.iseq_start);
      t.num_seqkvcache = t.num_seqkv;
      t.num_tile_kv = (t.num_seqkv + p.tilen - 1) / p.tilen;
      t.is_casual_chunk = 0;
      t.pad[0] = t.pad[1] = t.pad[2] = 0;
      if (add == left) {  // last chunk of (h, b): the Sq new tokens are causal
        t.is_casual_chunk = 1;
        t.num_seqkvcache -= p.num_seq_q;
      }
      g += add;
      const int ov = next_task_overflow(p, h, b, start_tile + add, static_cast<int>(end - g));
      if (ov < 0) {
        t.is_casual_chunk = 1;
        t.num_seqkvcache += ov;
      }
      t.num_tile_full = imax(t.num_seqkvcache / p.tilen, 0);
      int* dst = bin_ptr + itask * kTaskStride;
      const int* src = reinterpret_cast<const int*>(&t);
      for (int i = 0; i < kTaskStride; ++i) dst[i] = src[i];
      ++itask;
      if (add == left) {  // advance to the next non-empty request
        do {
          if (++b >= p.num_batch) {
            b = 0;
            ++h;
          }
        } while (h < p.num_head_kv && p.tiles[b] <= 0);
      }
    }
  }
  return itask;
}

__host__ __device__ inline int plan_bin(const Plan& p, int ibin, int* bin_ptr) {
  const int itask = plan_bin_tasks(p, ibin, bin_ptr);
  for (int slot = itask; slot <= p.per; ++slot) {  // terminators in every unused slot
    bin_ptr[slot * kTaskStride] = -1;
    bin_ptr[slot * kTaskStride + 1] = -1;
  }
  return itask;
}

// number of chunks request (h, b) is cut into
__host__ __device__ inline int chunks_of(const Plan& p, int h, int b) {
  if (p.tiles[b] <= 0) return 0;
  const long g0 = static_cast<long>(h) * p.total + p.cum[b];
  return static_cast<int>((g0 + p.tiles[b] - 1) / p.per - g0 / p.per) + 1;
}

namespace {

constexpr int kMaxBatch = 4096;
constexpr int kThreads = 64;      // one wave per workgroup
constexpr int kLanesPerBin = 8;   // lane 0 of each group plans the bin, all 8 write its terminators
constexpr int kBinsPerWg = kThreads / kLanesPerBin;

__global__ __launch_bounds__(kThreads) void assign_task_kernel(
    int* __restrict__ task_map, const int* __restrict__ num_seq_kvcache, int num_batch,
    int num_head_kv, int num_seq_q, int new_kv_included, int min_process_len, int max_bins) {
  __shared__ int s_seqkv[kMaxBatch];
  __shared__ int s_tiles[kMaxBatch];
  __shared__ int s_cum[kMaxBatch];
  const int lane = threadIdx.x;

  // per-batch tile counts + exclusive scan (each lane owns a contiguous segment)
  const int seg = (num_batch + kThreads - 1) / kThreads;
  const int b0 = lane * seg, b1 = imin(b0 + seg, num_batch);
  int local = 0;
  for (int b = b0; b < b1; ++b) {
    const int n = num_seq_kvcache[b] + (new_kv_included ? 0 : num_seq_q);
    const int t = (n + kTileN - 1) / kTileN;
    s_seqkv[b] = n;
    s_tiles[b] = t;
    local += t;
  }
  int incl = local;
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) {
    const int v = __shfl_up(incl, o, 64);
    if (lane >= o) incl += v;
  }
  const int total = __shfl(incl, 63, 64);
  int run = incl - local;
  for (int b = b0; b < b1; ++b) {
    s_cum[b] = run;
    run += s_tiles[b];
  }
  __syncthreads();

  Plan p;
  p.seqkv = s_seqkv;
  p.tiles = s_tiles;
  p.cum = s_cum;
  p.total = total;
  p.num_batch = num_batch;
  p.num_head_kv = num_head_kv;
  p.num_seq_q = num_seq_q;
  p.tilen = kTileN;
  const long grand = static_cast<long>(total) * num_head_kv;
  const int num_bins = effective_bins(grand, max_bins);  // the launch is sized for max_bins; a small batch uses fewer (sched_task_info.h)
  p.num_bins = num_bins;
  p.per = imax(static_cast<int>((grand + num_bins - 1) / num_bins), min_process_len / kTileN);

  const int max_batch = task_map[3];
  int* chunk_tab = task_map + chunk_table_off(p.per, num_bins);
  int* tasks_per_bin = chunk_tab + pad12(max_batch * num_head_kv) + pad12(num_bins);

  // The layout demands a terminator in every unused slot (tiles_per_bin + 1 records per bin, 48 B
  // apart): spread that over 8 lanes per bin and 8 bins per workgroup so the scattered stores of
  // all bins proceed in parallel across the chip.
  const int ibin = blockIdx.x * kBinsPerWg + lane / kLanesPerBin;
  const int sub = lane % kLanesPerBin;
  int* bin_ptr = task_map + static_cast<long>(kTaskStride) * (1 + static_cast<long>(ibin) * (p.per + 1));
  int itask = 0;
  if (ibin < num_bins && sub == 0) {
    itask = plan_bin_tasks(p, ibin, bin_ptr);
    tasks_per_bin[ibin] = itask;
  }
  itask = __shfl(itask, lane - sub, 64);
  if (ibin < num_bins) {
    for (int slot = itask + sub; slot <= p.per; slot += kLanesPerBin) {
      bin_ptr[slot * kTaskStride] = -1;
      bin_ptr[slot * kTaskStride + 1] = -1;
    }
  }

  if (blockIdx.x == 0) {  // header + chunk table (closed form, no atomics)
    int mx = 0;
    const int n_used = num_head_kv * num_batch;
    const int n_all = imax(max_batch * num_head_kv, n_used);
    for (int i = lane; i < n_all; i += kThreads) {
      int c = 0;
      if (i < n_used) c = chunks_of(p, i / num_batch, i % num_batch);
      chunk_tab[i] = c;
      mx = imax(mx, c);
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) mx = imax(mx, __shfl_xor(mx, o, 64));
    if (lane == 0) {
      task_map[0] = p.per + 1;
      task_map[1] = num_bins;
      task_map[5] = mx;
      task_map[6] = min_process_len;  // ours: the head-pair decode kernels plan in-kernel and honour it (sched_task_info.h)
    }
  }
}

}  // namespace
}  // namespace sched
}  // namespace hpc

using namespace hpc::sched;

extern "C" int hpc_attention_decode_num_bins(int num_seq_q, int device_id) {
  if (num_seq_q < 1 || num_seq_q > kMaxSeqQ) return HPC_ERR_INVALID;
  const int cus = hpc_get_cu_count(device_id);
  if (cus <= 0) return HPC_ERR_LAUNCH;
  const int dev_bins = hpc_dev_tuning_get(34);  // development: bin count override (A/B of the bin size on small problems)
  if (dev_bins > 0 && dev_bins <= 4 * cus) return dev_bins;
  return cus * cta_per_cu(num_seq_q);
}

extern "C" int hpc_attention_decode_tile_n(void) { return kTileN; }

extern "C" int hpc_attention_decode_effective_bins(const int* num_seq_kvcache, int num_batch, int num_head_kv, int num_seq_q,
                                                   int new_kv_included, int max_bins) {
  if (!num_seq_kvcache || num_batch < 0 || num_head_kv <= 0 || max_bins <= 0) return HPC_ERR_INVALID;
  long total = 0;
  for (int b = 0; b < num_batch; ++b) {
    const int n = num_seq_kvcache[b] + (new_kv_included ? 0 : num_seq_q);
    total += (n + kTileN - 1) / kTileN;
  }
  return effective_bins(total * num_head_kv, max_bins);
}

extern "C" int hpc_assign_attention_decode_task_rows(const int* num_seq_kvcache, int num_total_ctas,
                                                     int num_batch, int num_head_kv, int num_seq_q,
                                                     int new_kv_included, int min_process_len) {
  if (!num_seq_kvcache || num_total_ctas <= 0 || num_batch < 0 || num_head_kv <= 0) return HPC_ERR_INVALID;
  long total = 0;
  for (int b = 0; b < num_batch; ++b) {
    const int n = num_seq_kvcache[b] + (new_kv_included ? 0 : num_seq_q);
    total += (n + kTileN - 1) / kTileN;
  }
  const long grand = total * num_head_kv;
  const int per = imax(static_cast<int>((grand + num_total_ctas - 1) / num_total_ctas),
                       min_process_len / kTileN);
  return 1 + num_total_ctas * (per + 1) + (num_head_kv * num_batch * 4 + 47) / 48;
}

extern "C" int hpc_assign_attention_decode_task_sync(const int* num_seq_kvcache, int num_total_ctas,
                                                     int num_batch, int num_head_kv, int num_seq_q,
                                                     int new_kv_included, int min_process_len,
                                                     int* task_map, int task_map_rows) {
  const int rows = hpc_assign_attention_decode_task_rows(num_seq_kvcache, num_total_ctas, num_batch,
                                                         num_head_kv, num_seq_q, new_kv_included,
                                                         min_process_len);
  if (rows < 0) return rows;
  if (!task_map || task_map_rows < rows) return HPC_ERR_INVALID;
  if (num_seq_q < 1 || num_seq_q > kMaxSeqQ) return HPC_ERR_INVALID;
  std::vector<int> seqkv(num_batch), tiles(num_batch), cum(num_batch);
  int total = 0;
  for (int b = 0; b < num_batch; ++b) {
    seqkv[b] = num_seq_kvcache[b] + (new_kv_included ? 0 : num_seq_q);
    tiles[b] = (seqkv[b] + kTileN - 1) / kTileN;
    cum[b] = total;
    total += tiles[b];
  }
  Plan p;
  p.seqkv = seqkv.data();
  p.tiles = tiles.data();
  p.cum = cum.data();
  p.total = total;
  p.num_batch = num_batch;
  p.num_head_kv = num_head_kv;
  p.num_seq_q = num_seq_q;
  p.tilen = kTileN;
  p.num_bins = num_total_ctas;
  const long grand = static_cast<long>(total) * num_head_kv;
  p.per = imax(static_cast<int>((grand + num_total_ctas - 1) / num_total_ctas),
               min_process_len / kTileN);
  for (long i = 0; i < static_cast<long>(rows) * kTaskStride; ++i) task_map[i] = 0;
  for (int ibin = 0; ibin < num_total_ctas; ++ibin)
    plan_bin(p, ibin, task_map + static_cast<long>(kTaskStride) * (1 + static_cast<long>(ibin) * (p.per + 1)));
  int* chunk_tab = task_map + chunk_table_off(p.per, num_total_ctas);
  int mx = 0;
  for (int h = 0; h < num_head_kv; ++h)
    for (int b = 0; b < num_batch; ++b) {
      const int c = chunks_of(p, h, b);
      chunk_tab[h * num_batch + b] = c;
      mx = imax(mx, c);
    }
  task_map[0] = p.per + 1;
  task_map[1] = num_total_ctas;
  task_map[5] = mx;
  task_map[6] = min_process_len;
  return rows;
}

extern "C" int hpc_assign_attention_decode_task_async(int* task_map, const int* num_seq_kvcache,
                                                      int num_total_ctas, int num_batch,
                                                      int num_head_kv, int num_seq_q,
                                                      int new_kv_included, int min_process_len,
                                                      hipStream_t stream) {
  if (!task_map || !num_seq_kvcache) return HPC_ERR_INVALID;
  if (num_batch <= 0 || num_batch > kMaxBatch) return HPC_ERR_UNSUPPORTED;
  if (num_seq_q < 1 || num_seq_q > kMaxSeqQ || num_head_kv <= 0 || num_total_ctas <= 0)
    return HPC_ERR_INVALID;
  const int grid = (num_total_ctas + kBinsPerWg - 1) / kBinsPerWg;
  assign_task_kernel<<<grid, kThreads, 0, stream>>>(task_map, num_seq_kvcache, num_batch,
                                                    num_head_kv, num_seq_q, new_kv_included ? 1 : 0,
                                                    min_process_len, num_total_ctas);
  HPC_CHECK_LAUNCH();
  return HPC_OK;
}
