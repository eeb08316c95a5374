/* sched_oracle.c — TEST INFRASTRUCTURE (see oracle/__init__.py). NOT PRODUCT.
 *
 * CPU restatement, in plain C, of the reference's decode-attention dynamic tile scheduler:
 *   assign_attention_decode_task_sync        reference src/attention/decode/assign_task.cu:362-492
 *   task-map packing of the CPU entry        reference src/attention/entry.cc:727-778
 *   TaskScheduleInfo (12 x int32 = 48 bytes) reference src/attention/decode/sched_task_info.h:18-31
 *
 * Pinning: oracle/Makefile compiles the reference's own function (extracted where it lies under
 * /root/reference) into oracle/_ref/libsched_ref.so; tests/golden/make_golden.py checks this file
 * against it and records golden task maps under tests/golden/ (replayed by
 * tests/test_sched_oracle.py without the reference).
 *
 * Known difference kept on purpose: the reference copies an uninitialised `pad[3]` into each task
 * (TaskScheduleInfo task_info; is not value-initialised); this restatement writes zeros there, and
 * every comparison masks ints 9..11 of each record.
 */
#include <stdlib.h>
#include <string.h>

enum { kStride = 12 }; /* ints per task record */

typedef struct {
  int ihead_kv, ibatch, ichunk, iseq_start;
  int num_seqkv, num_seqkvcache, num_tile_kv, num_tile_full;
  int is_casual_chunk, pad[3];
} task_t;

static int imax(int a, int b) { return a > b ? a : b; }
static int imin(int a, int b) { return a < b ? a : b; }

/* Number of tiles per bin for these inputs (reference assign_task.cu:369-382). */
int sched_oracle_tiles_per_bin(const int* num_seq_kvcache, int num_total_ctas, int num_batch,
                               int num_head_kv, int num_seq_q, int tilen, int new_kv_included,
                               int min_process_len) {
  long total = 0;
  for (int b = 0; b < num_batch; ++b) {
    int n = new_kv_included ? num_seq_kvcache[b] : num_seq_kvcache[b] + num_seq_q;
    total += (n + tilen - 1) / tilen;
  }
  total *= num_head_kv;
  return imax((int)((total + num_total_ctas - 1) / num_total_ctas), min_process_len / tilen);
}

/* Fills `tasks` (num_total_ctas * (tiles_per_bin + 1) records, caller-zeroed) and `num_chunks`
 * (num_head_kv * num_batch ints, caller-zeroed).  Returns tiles_per_bin.
 * Line-by-line restatement of reference assign_task.cu:362-492, quirks included (last_cta is only
 * advanced when a bin closes, :473). */
int sched_oracle_assign(const int* num_seq_kvcache, int num_total_ctas, int num_batch,
                        int num_head_kv, int num_seq_q, int tilen, int new_kv_included,
                        int min_process_len, int* tasks_out, int* num_chunks) {
  task_t* tasks = (task_t*)tasks_out;
  int* num_seqkvs = (int*)calloc(num_batch > 0 ? num_batch : 1, sizeof(int));
  int* num_tiles = (int*)calloc(num_batch > 0 ? num_batch : 1, sizeof(int));
  const int nhb = imax(num_batch * num_head_kv, 1);
  int* start_tiles = (int*)calloc(nhb, sizeof(int));
  int* chunks_in_progress = (int*)calloc(nhb, sizeof(int));
  int* num_tiles_left = (int*)calloc(nhb, sizeof(int));

  int total_tiles_per_head = 0;
  for (int b = 0; b < num_batch; ++b) {
    int n = new_kv_included ? num_seq_kvcache[b] : num_seq_kvcache[b] + num_seq_q;
    num_seqkvs[b] = n;
    num_tiles[b] = (n + tilen - 1) / tilen;
    total_tiles_per_head += num_tiles[b];
  }
  const int total_tiles_all_heads = total_tiles_per_head * num_head_kv;
  const int per = imax((total_tiles_all_heads + num_total_ctas - 1) / num_total_ctas,
                       min_process_len / tilen);
  for (int h = 0; h < num_head_kv; ++h)
    for (int b = 0; b < num_batch; ++b) num_tiles_left[h * num_batch + b] = num_tiles[b];

  int ihead_kv = 0, ibatch = 0, last_cta = 0, last_task = 0;
  for (int icta = 0; icta < num_total_ctas; ++icta) {
    int bucket = per, itask = 0;
    task_t* bin = tasks + (long)icta * (per + 1);
    while (bucket > 0 && ihead_kv < num_head_kv) {
      const int idx = ihead_kv * num_batch + ibatch;
      const int num_tile = num_tiles_left[idx];
      if (num_tile <= 0) { /* skip (head, batch) with no tiles */
        if (++ibatch >= num_batch) {
          ibatch = 0;
          if (++ihead_kv >= num_head_kv) break;
        }
        continue;
      }
      int add_tiles = imin(num_tile, bucket);
      const int num_seqkv = num_seqkvs[ibatch];
      if (chunks_in_progress[idx] == num_total_ctas - 1) add_tiles = num_tile;

      task_t t;
      memset(&t, 0, sizeof(t));
      t.ihead_kv = ihead_kv;
      t.ibatch = ibatch;
      t.ichunk = chunks_in_progress[idx];
      t.iseq_start = start_tiles[idx] * tilen;
      t.num_seqkv = imin(add_tiles * tilen, num_seqkv - t.iseq_start);
      t.num_seqkvcache = t.num_seqkv;
      t.num_tile_kv = (t.num_seqkv + tilen - 1) / tilen;
      t.num_tile_full = t.num_seqkvcache / tilen;
      t.is_casual_chunk = 0;
      bin[itask] = t;

      itask++;
      chunks_in_progress[idx]++;
      start_tiles[idx] += add_tiles;
      num_tiles_left[idx] -= add_tiles;
      bucket -= add_tiles;

      if (num_tiles_left[idx] <= 0) { /* last chunk of (head, batch) */
        task_t* cur = &bin[itask - 1];
        cur->is_casual_chunk = 1;
        cur->num_seqkvcache -= num_seq_q;
        cur->num_tile_full = imax(cur->num_seqkvcache / tilen, 0);
        num_chunks[ihead_kv * num_batch + ibatch] = chunks_in_progress[idx];
        if (cur->num_seqkvcache < 0) { /* Q tokens overflow into the previous task */
          task_t* prev = &tasks[(long)last_cta * (per + 1) + last_task];
          prev->is_casual_chunk = 1;
          prev->num_seqkvcache += cur->num_seqkvcache;
          prev->num_tile_full = imax(prev->num_seqkvcache / tilen, 0);
        }
        if (++ibatch >= num_batch) {
          ibatch = 0;
          ihead_kv++;
        }
      }
      last_task = itask - 1;
    }
    last_cta = icta;
    for (int slot = itask; slot <= per; ++slot) { /* terminators in every unused slot */
      bin[slot].ihead_kv = -1;
      bin[slot].ibatch = -1;
    }
  }
  free(num_seqkvs);
  free(num_tiles);
  free(start_tiles);
  free(chunks_in_progress);
  free(num_tiles_left);
  return per;
}

/* Rows (of 48 bytes) of the tensor returned by the reference CPU entry (entry.cc:758-760). */
int sched_oracle_map_rows(int tiles_per_bin, int num_total_ctas, int num_batch, int num_head_kv) {
  const int num_task = num_total_ctas * (tiles_per_bin + 1);
  const int chunk_bytes = num_head_kv * num_batch * 4;
  return 1 + num_task + (chunk_bytes + 47) / 48;
}

/* Packs the host task map exactly like assign_attention_decode_task_cpu_entry
 * (reference entry.cc:758-775): row 0 header ([0]=tiles_per_bin+1, [1]=bins, [5]=max chunks),
 * then the task records, then num_chunks[h*B+b].  `map` must hold sched_oracle_map_rows()*12 ints
 * and be zeroed by the caller.  Returns the number of rows. */
int sched_oracle_task_map(const int* num_seq_kvcache, int num_total_ctas, int num_batch,
                          int num_head_kv, int num_seq_q, int tilen, int new_kv_included,
                          int min_process_len, int* map) {
  const int per = sched_oracle_tiles_per_bin(num_seq_kvcache, num_total_ctas, num_batch,
                                             num_head_kv, num_seq_q, tilen, new_kv_included,
                                             min_process_len);
  const int num_task = num_total_ctas * (per + 1);
  int* num_chunks = (int*)calloc(imax(num_head_kv * num_batch, 1), sizeof(int));
  sched_oracle_assign(num_seq_kvcache, num_total_ctas, num_batch, num_head_kv, num_seq_q, tilen,
                      new_kv_included, min_process_len, map + kStride, num_chunks);
  map[0] = per + 1;
  map[1] = num_total_ctas;
  memcpy(map + (long)kStride * (num_task + 1), num_chunks, sizeof(int) * num_head_kv * num_batch);
  int mx = 0;
  for (int r = 0; r < num_head_kv * num_batch; ++r) mx = imax(mx, num_chunks[r]);
  map[5] = mx;
  free(num_chunks);
  return sched_oracle_map_rows(per, num_total_ctas, num_batch, num_head_kv);
}
