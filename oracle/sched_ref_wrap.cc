// sched_ref_wrap.cc — TEST INFRASTRUCTURE.  extern "C" shim around the REFERENCE's own
// assign_attention_decode_task_sync (compiled from /root/reference by oracle/Makefile into
// oracle/_ref/libsched_ref.so) and a restatement of the packing done by the reference CPU entry
// (src/attention/entry.cc:758-775) so both checkers return the same host task-map image.
#include <cstring>
#include <utility>
#include <vector>

#include "src/attention/decode/sched_task_info.h"

namespace hpc { namespace attention { namespace decode {
std::pair<std::vector<dynamic::TaskScheduleInfo>, std::vector<int>>
assign_attention_decode_task_sync(const int* num_seq_kvcache, int num_total_ctas, int num_batch,
                                  int num_head_kv, int num_seq_q, int tilen, bool new_kv_included,
                                  int min_process_len);
}}}

extern "C" int sched_ref_task_map(const int* num_seq_kvcache, int num_total_ctas, int num_batch,
                                  int num_head_kv, int num_seq_q, int tilen, int new_kv_included,
                                  int min_process_len, int* map, int map_rows) {
  auto pr = hpc::attention::decode::assign_attention_decode_task_sync(
      num_seq_kvcache, num_total_ctas, num_batch, num_head_kv, num_seq_q, tilen,
      new_kv_included != 0, min_process_len);
  auto& tasks = pr.first;
  auto& num_chunks = pr.second;
  const int per1 = num_chunks[num_head_kv * num_batch];
  const int num_task = static_cast<int>(tasks.size());
  const int rows = 1 + num_task + (num_head_kv * num_batch * 4 + 47) / 48;
  if (rows > map_rows) return -rows;
  std::memset(map, 0, sizeof(int) * 12 * rows);
  map[0] = per1;
  map[1] = num_total_ctas;
  std::memcpy(map + 12, tasks.data(), 48 * static_cast<size_t>(num_task));
  std::memcpy(map + 12 * (num_task + 1), num_chunks.data(), 4 * num_head_kv * num_batch);
  int mx = 0;
  for (int r = 0; r < num_head_kv * num_batch; ++r) mx = num_chunks[r] > mx ? num_chunks[r] : mx;
  map[5] = mx;
  return rows;
}
