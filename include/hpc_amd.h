/* hpc_amd.h — C-ABI of libhpc_amd.so, the MI355X (gfx950) drop-in for the decode-step hot path
 * of Tencent/hpc-ops.
 *
 * Every entry point replaces one `*_async` host launcher of the reference (cited per function)
 * and keeps its argument meaning: raw device pointers, int dims, int64 strides for the KV cache,
 * the stream last.  No torch types cross this boundary.
 *
 * Return convention (replaces the reference's `bool running` / cudaError_t mix,
 * src/attention/entry.cc:722, src/allreduce/entry.cc:187-189):
 *    0  launched on `stream` (asynchronous, never synchronises)
 *   -1  HPC_ERR_UNSUPPORTED  shape / tile configuration not supported
 *   -2  HPC_ERR_INVALID      bad argument (null pointer, inconsistent sizes)
 *   -3  HPC_ERR_LAUNCH       the HIP runtime refused the launch
 *   -4  HPC_ERR_TIMEOUT      (fused all-reduce only) an earlier call on this process gave up waiting
 *                            for a peer; the symmetric buffers / signal pads are in an undefined state
 * The Python shim turns every non-zero code into RuntimeError("... launch failed!").
 */
#ifndef HPC_AMD_H_
#define HPC_AMD_H_

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* hipStream_t without dragging hip headers into FFI users. */
typedef struct ihipStream_t* hpc_stream_t;

/* ---- library identity (reference: src/C/version.cc, src/C/built_json.cu:17-43) ---------- */
const char* hpc_version(void);
const char* hpc_built_json(void);
/* Number of compute units of device `device_id` (reference get_sm_count(), src/utils/utils.cc:15-27;
 * here per device, not a device-0 cache). */
int hpc_get_cu_count(int device_id);

/* ---- RMSNorm (+fp8 quant) ------------------------------------------------------------------
 * reference: fused_rmsnorm_with_scale_async, src/normalization/fused_rmsnorm_with_scale.h:19-22
 *            kernel src/normalization/fused_rmsnorm_with_scale.cu:14-137.
 * input bf16 [batch, hidden], weight bf16 [hidden], scale f32 [1] (or [2] when is_moe).
 * output_fp8 e4m3 [batch, hidden] = y / scale[0];  is_moe: output_fp32 = y,
 * output_fp8_scale2 = y / scale[1].  hidden % 8 == 0, hidden <= 16384. */
int hpc_fused_rmsnorm_with_scale_async(const void* input, const void* weight, void* output_fp8,
                                       void* output_fp32, void* output_fp8_scale2,
                                       const void* scale, float eps, int batch_size,
                                       int hidden_state, int is_moe, hpc_stream_t stream);

/* ---- decode attention: dynamic split-KV tile scheduler ----------------------------------------
 * reference: assign_attention_decode_task_{sync,async}, src/attention/decode/decode.h:37-46,
 *            src/attention/decode/assign_task.cu:333-492, CPU entry packing src/attention/entry.cc:727-778,
 *            record/bins constants src/attention/decode/sched_task_info.h:18-36.
 * Task-map layout (int32 rows of 12) is the reference's; see hpc-ops_amd/csrc/sched_task_info.h.
 *
 * hpc_attention_decode_num_bins: number of scheduler bins (= decode workgroups) on `device_id`
 *   for num_seq_q in 1..5 (reference: kCtaPerSmMap[sm][num_seq_q-1] * get_sm_count()).
 * hpc_attention_decode_tile_n:   KV tokens per scheduling tile (64).
 * hpc_assign_attention_decode_task_rows / _sync: host scheduler. `_rows` returns the number of
 *   48-byte rows the host image needs (1 + bins*(tiles_per_bin+1) + ceil(Hkv*B*4/48)); `_sync`
 *   fills `task_map` (>= rows*12 ints) like the reference CPU entry and returns rows.
 * hpc_assign_attention_decode_task_async: device scheduler; `task_map` is the workspace of
 *   hpc.get_attention_decode_task_workspace (header ints 2..4 pre-filled by the allocator),
 *   num_seq_kvcache is a device pointer.  `num_total_ctas` is the UPPER bound the decode launch is sized for: a
 *   batch with little work is planned on fewer bins (hpc_attention_decode_effective_bins: at least 8 tiles per bin,
 *   at least 64 bins) and header int 1 of the map records the count used - consumers read it from there.  Byte-
 *   identical to the host scheduler run with that count.
 * hpc_attention_decode_effective_bins: that count from host lengths (what the CPU entry of the op passes to _sync). */
int hpc_attention_decode_num_bins(int num_seq_q, int device_id);
int hpc_attention_decode_effective_bins(const int* num_seq_kvcache, int num_batch, int num_head_kv, int num_seq_q,
                                        int new_kv_included, int max_bins);
int hpc_attention_decode_tile_n(void);
int hpc_assign_attention_decode_task_rows(const int* num_seq_kvcache, int num_total_ctas,
                                          int num_batch, int num_head_kv, int num_seq_q,
                                          int new_kv_included, int min_process_len);
int hpc_assign_attention_decode_task_sync(const int* num_seq_kvcache, int num_total_ctas,
                                          int num_batch, int num_head_kv, int num_seq_q,
                                          int new_kv_included, int min_process_len, int* task_map,
                                          int task_map_rows);
int hpc_assign_attention_decode_task_async(int* task_map, const int* num_seq_kvcache,
                                           int num_total_ctas, int num_batch, int num_head_kv,
                                           int num_seq_q, int new_kv_included, int min_process_len,
                                           hpc_stream_t stream);

/* ---- decode attention (paged KV, D = 128, GQA group 4 or 8) ----------------------------------------
 * reference: attention_decode_bf16_async / attention_decode_fp8_async,
 *            src/attention/decode/decode.h:17-35 (kernels under src/attention/decode/sm90/).
 * q [B*Sq, Hq, 128] (row stride ldQ elements), kcache/vcache logical [blocks, block_size, Hkv, 128]
 * with element strides (block, token, head), block_ids int32 [B, num_seq_max_blocks],
 * y bf16 [B*Sq, Hq, 128] (row stride ldY).  `task_map_ptr` is a task map produced by the
 * scheduler above for the same num_seq_kvcache / num_seq_q / new_kv_included.  `num_bins` is the UPPER bound the
 * launch and the workspace are sized with - hpc_attention_decode_num_bins(num_seq_q, device), the value the torch
 * binding passes - NOT the map's header[1]: the scheduler plans a batch with little work on fewer bins
 * (hpc_attention_decode_effective_bins) and records the count it used in header[1] <= num_bins; the kernels read it
 * from there.  A caller that passed header[1] as num_bins would cap the head-pair kernel's grid at that count instead of
 * two workgroups per CU.
 * `workspace` holds the arrival counters of split requests and the fp32 split-KV partials:
 * hpc_attention_decode_workspace_bytes bytes.  Its first hpc_attention_decode_workspace_zero_bytes() bytes (the
 * counters: a fixed place and size whatever the shapes of the call) must be ZERO the first time the buffer is
 * used - like the reference's split_flag, which its entry allocates zeroed (src/attention/entry.cc:690-694) - and
 * every call leaves them zero again, so one buffer per stream can be reused call after call, with any sequence
 * of shapes, and inside a captured hipGraph; the rest needs no initialisation (the reference allocates
 * lse/split_out per call, src/attention/entry.cc:492-499; here it is 2 slots per workgroup instead of splitk
 * slots per request).  The zero-bytes contract applies to EVERY decode path (since round 5 the first-generation
 * kernels take their tickets there too): a counter that is not zero on entry - a caller-owned buffer that was never
 * cleared, or one left behind by a launch that faulted - means the last arriver of a split request is never
 * recognised and that request's rows of y are not written.  One buffer must not be used by two calls that may run
 * concurrently (two streams: two buffers).  The split-KV combine runs inside the launch on both kernel generations:
 * the chunk of a request that arrives LAST merges all of them (reference: static_splitk_kernels.cuh:362-377); there
 * is no second kernel.
 * num_seq_kvcache_ptr (device int32 [num_batch]) / new_kv_included as in the reference launchers (decode.h:17-35): with
 * them, NHD pages (adjacent kv heads contiguous) with an even head count and <= 16 q rows per kv head take the
 * second-generation kernel (attention_decode_v2.hip), which plans its ranges in closed form from the lengths and takes
 * from the task map only header int 6 (the scheduler call's min_process_len: the lower bound of its split granularity);
 * NULL lengths = the task map's bins drive the first-generation kernel. */
int64_t hpc_attention_decode_workspace_bytes(int num_bins, int num_batch, int num_head_kv,
                                             int num_seq_q, int heads_per_group);
int64_t hpc_attention_decode_workspace_zero_bytes(void);
/* 0 when `stream` is not being captured into a hipGraph, else the unique non-zero id of the capture (< 0: error).  A
 * host that caches the decode workspace per stream must not carry a buffer whose zero-fill was only RECORDED in one
 * capture over to another capture or to eager calls: key the cache on (stream, capture id). */
long long hpc_stream_capture_id(hpc_stream_t stream);
int hpc_attention_decode_bf16_async(void* y_ptr, void* workspace, const int* task_map_ptr,
                                    const void* q_ptr, const void* kcache_ptr,
                                    const void* vcache_ptr, const int* block_ids_ptr,
                                    const int* num_seq_kvcache_ptr, int new_kv_included, int num_bins,
                                    int num_batch, int num_seq_q, int num_head_q, int num_head_kv,
                                    int num_dim_qk, int num_dim_v, int block_size,
                                    int num_seq_max_blocks, int ldY, int ldQ,
                                    int64_t kcache_block_stride, int64_t kcache_token_stride,
                                    int64_t kcache_head_stride, int64_t vcache_block_stride,
                                    int64_t vcache_token_stride, int64_t vcache_head_stride,
                                    hpc_stream_t stream);
/* FP8 (OCP e4m3fn) variant.  q e4m3 [B*Sq, Hq, 128]; qscale f32 [B*Sq, qscale_pad_stride] (row
 * b*Sq+s, column q head); quant_type 1 (QPERTOKEN_PERHEAD_KPERTENSOR_VPERTENSOR): kscale f32[1],
 * vscale f32[1]; quant_type 0 (QPERTOKEN_PERHEAD_KPERTOKEN_PERHEAD_VPERHEAD): kscale_ptr = base of
 * the K-scale rows that follow the block_size token rows of every K page (row r of head h holds
 * the fp32 scales of tokens 32r..32r+31 as 128 raw bytes; byte strides given explicitly - the
 * reference reads them through a TMA view, sm90/dynamic/...qkpertoken..._dynamic.cu:84-91),
 * vscale f32[Hkv].  Numerics: P~ = e4m3(256 * 2^(s - running max)), O = sum(P~ V)/sum(p) * vscale/256.
 * num_seq_q <= 4.  block_size 16/32/64 for quant_type 1, 32/64 for quant_type 0.
 * num_seq_kvcache_ptr (device int32 [num_batch]) / new_kv_included as in the reference launcher
 * (decode.h:27-35): with them, NHD pages (adjacent kv heads 128 bytes apart), <= 16 q rows per kv head and
 * an even kv head count take the second-generation kernel (attention_decode_v2.hip: 256 contiguous bytes per
 * row and load - two heads of a token - deep prefetch, the schedule planned in-kernel from the lengths in the
 * closed form of the scheduler above: of the task map that path consumes header int 6 = the scheduler call's
 * min_process_len - a workgroup's range is never shorter than that many KV tokens - while the map's bins and
 * split decisions do not apply there; since round 6 per-token K scales on NHD pages run there too, and calls with
 * 17 ... 32 q rows per kv head - speculative steps, num_seq_q 3 / 4 at 8 q heads per kv head - run the same pipeline with ONE
 * kv head per workgroup, on NHD and HND pages, both quant types, pages of 32 / 64 tokens); with <= 16 q rows an odd kv head
 * count - incl. a single kv head - and HND pages run the first-generation kernel from the task map;
 * num_seq_kvcache_ptr may be NULL, then the task map drives the first-generation kernel as for bf16. */
int hpc_attention_decode_fp8_async(void* y_ptr, void* workspace, const int* task_map_ptr,
                                   const void* q_ptr, const void* kcache_ptr,
                                   const void* vcache_ptr, const int* block_ids_ptr,
                                   const int* num_seq_kvcache_ptr, const float* qscale_ptr,
                                   const void* kscale_ptr, const float* vscale_ptr,
                                   int new_kv_included, int quant_type, int num_bins,
                                   int num_batch, int num_seq_q, int num_head_q, int num_head_kv,
                                   int num_dim_qk, int num_dim_v, int block_size,
                                   int num_seq_max_blocks, int qscale_pad_stride, int ldY, int ldQ,
                                   int64_t kcache_block_stride, int64_t kcache_token_stride,
                                   int64_t kcache_head_stride, int64_t vcache_block_stride,
                                   int64_t vcache_token_stride, int64_t vcache_head_stride,
                                   int64_t kscale_block_stride, int64_t kscale_row_stride,
                                   int64_t kscale_head_stride, hpc_stream_t stream);

/* ---- grouped FP8 GEMM with 128-block scales -------------------------------------------------------
 * reference: group_gemm_blockwise_fp8_async, src/group_gemm/group_gemm.h:12-29
 *            (kernel src/group_gemm/kernels.cuh:532-892; scatter-A variant cp_async/group_gemm_fp8_scatter.cu).
 * Y[m, n] = bf16( sum_kb (sum_{k in kb} X[row(m),k] W[g,n,k]) * xs(m,kb) * ws[g, n/128, kb] ) for the
 * seqlens[g] rows starting at cu_seqlens[g].  x e4m3 [rows, k]; w e4m3 [G, n, k]; ws f32
 * [G, n/128, num_block_k_pad4]; y bf16 [m, n].  n, k multiples of 128.
 * row_index (nullable): x / xs row of output row m (gather-free MoE), else row(m) = m.
 * xs addressing in floats: xs[term * xscale_row_stride + kb * xscale_kb_stride] with
 *   term = row(m)                       when col_base == NULL  (row-major [rows, k/128]: strides k/128, 1)
 *   term = col_base[g]*tile_m + slot    when col_base != NULL  (reference layout [k/128, m_pad],
 *                                        col_base = cu_tiles: strides 1, m_pad).
 * cu_tiles128_ptr (nullable): exclusive scan of ceil(seqlens/128), [G+1] (hpc_moe_tiles_async with
 * tile_m = 128); when given and groups are large (> 32 rows on average) the MFMA-bound 128x128-tile
 * kernel is used instead of the weight-streaming one. */
int hpc_group_gemm_blockwise_fp8_async(void* y_ptr, const void* x_ptr, const void* w_ptr,
                                       const void* seqlens_ptr, const void* cu_seqlens_ptr,
                                       const void* xscale_ptr, const void* wscale_ptr,
                                       const void* row_index_ptr, const void* col_base_ptr,
                                       int num_group, int m, int n, int k, int num_block_k_pad4,
                                       int tile_m, int64_t xscale_row_stride,
                                       int64_t xscale_kb_stride, const void* cu_tiles128_ptr,
                                       hpc_stream_t stream);

/* ---- fused MoE pieces ---------------------------------------------------------------------------
 * reference: src/fuse_moe/fuse_moe.h:15-62 (count_and_gather_async / blockwise_count_and_gather_async,
 *            reduce_async, fuse_moe_blockwise_async), src/activation/activation.h:15-53.
 * hpc_moe_count_and_slot_async: seqlens[e], cu_seqlens[E+1], tiles[e] = ceil(seqlens/tile_m),
 *   cu_tiles[E+1], topk_pos[T,k] (stable arrival order, -1 for experts outside
 *   [rank_ep*E, (rank_ep+1)*E)), row_index[pos] = token.  All int32 device buffers.
 * hpc_moe_tiles_async: tiles / cu_tiles from seqlens only.
 * hpc_moe_gather_blockwise_async: x_gathered[pos] = x[token]; xscale_t[kb][cu_tiles[e]*tile_m+slot].
 * hpc_act_mul_and_blockwise_quant_async: gate_up bf16 [rows, 2*I] -> out e4m3 [rows, I],
 *   out_scale[r*scale_row_stride + jb*scale_block_stride] (r = row_to_col[row] or row); rows =
 *   min(*num_rows_ptr, max_rows) when num_rows_ptr != NULL (device int).
 * hpc_moe_reduce_async: y[t] = bf16(sum_j topk_scale[t,j]*x[topk_pos[t,j]] + shared[t]). */
int hpc_moe_count_and_slot_async(const void* topk_ids, int num_tokens, int num_topk, int num_expert,
                                 int rank_ep, int tile_m, void* seqlens, void* cu_seqlens,
                                 void* tiles, void* cu_tiles, void* topk_pos, void* row_index,
                                 hpc_stream_t stream);
int hpc_moe_tiles_async(const void* seqlens, int num_group, int tile_m, void* tiles, void* cu_tiles,
                        hpc_stream_t stream);
int hpc_moe_gather_blockwise_async(const void* x, const void* x_scale, const void* topk_ids,
                                   const void* topk_pos, const void* cu_seqlens,
                                   const void* cu_tiles, int num_tokens, int num_topk,
                                   int num_expert, int rank_ep, int hidden, int tile_m, int m_pad,
                                   void* x_gathered, void* xscale_t, hpc_stream_t stream);
int hpc_act_mul_and_blockwise_quant_async(void* out_ptr, void* out_scale_ptr, const void* gate_up_ptr,
                                          const void* num_rows_ptr, int max_rows,
                                          int intermediate_size, int64_t scale_row_stride,
                                          int64_t scale_block_stride, const void* row_to_col_ptr,
                                          hpc_stream_t stream);
int hpc_moe_reduce_async(void* y_ptr, const void* x_ptr, const void* topk_pos_ptr,
                         const void* topk_scale_ptr, const void* shared_output_ptr, int num_tokens,
                         int num_topk, int hidden_size, hpc_stream_t stream);
/* Whole blockwise pipeline (count/slot -> gate_up GEMM -> SiLU*up + quant -> down GEMM -> reduce);
 * `workspace` >= hpc_fuse_moe_blockwise_workspace_bytes(...) bytes, uninitialised.
 * intermediate_size2 = gate_up_weight.size(1) = 2*I (as in the reference entry, fuse_moe/entry.cc:520). */
int64_t hpc_fuse_moe_blockwise_workspace_bytes(int num_tokens, int num_topk, int hidden_size,
                                               int intermediate_size2, int num_expert);
int hpc_fuse_moe_blockwise_async(void* y_ptr, void* workspace, const void* x_ptr,
                                 const void* x_scale_ptr, const void* gate_up_weight_ptr,
                                 const void* gate_up_weight_scale_ptr, const void* down_weight_ptr,
                                 const void* down_weight_scale_ptr, const void* topk_ids_ptr,
                                 const void* topk_scale_ptr, const void* shared_output_ptr,
                                 int num_tokens, int hidden_size, int intermediate_size2,
                                 int num_topk, int num_expert_total, int num_expert,
                                 int gate_up_ws_pad4, int down_ws_pad4, int rank_ep,
                                 hpc_stream_t stream);

/* Per-tensor FP8 MoE pieces (reference src/fuse_moe/fuse_moe.h:15-40 fuse_moe_async, src/group_gemm/
 * group_gemm.h group_gemm_fp8_async, src/activation/activation.h act_mul_and_quant_async /
 * scaled_fp8_quant_async, cp.async pipeline src/fuse_moe/cp_async/fuse_moe.cu:16-63):
 * Y = bf16((X W^T) * y_scale[g]); n % 64 == 0, k % 64 == 0; x_rows = rows of x (bounds the loads when
 * row_index is used).  act: out = e4m3(silu(gate) * up * scale[0]) with the bf16-rounded multiply of
 * the reference when use_bf16_mul.  The fused per-tensor pipeline uses the same workspace size as the
 * blockwise one (hpc_fuse_moe_blockwise_workspace_bytes with intermediate_size2 rounded up to 256). */
int hpc_group_gemm_pertensor_fp8_async(void* y_ptr, const void* x_ptr, const void* w_ptr,
                                       const void* seqlens_ptr, const void* cu_seqlens_ptr,
                                       const void* yscale_ptr, const void* row_index_ptr, int num_group,
                                       int m, int x_rows, int n, int k, const void* cu_tiles128_ptr,
                                       hpc_stream_t stream);
int hpc_act_mul_and_quant_async(void* out_ptr, const void* gate_up_ptr, const void* scale_ptr,
                                const void* num_rows_ptr, int max_rows, int intermediate_size,
                                int use_bf16_mul, hpc_stream_t stream);
/* out[i] = e4m3(float(in[i]) * (1.0f / scale[0])), saturating, any numel > 0 (reference scaled_fp8_quant_async,
 * src/activation/activation.h + activation.cu:461-505, entry src/activation/entry.cc:158-200).  The reference
 * overloads on the input type; here in_dtype says it: 0 = bf16, 1 = fp16, 2 = fp32. */
int hpc_scaled_fp8_quant_async(void* out_ptr, const void* in_ptr, const void* scale_ptr, int64_t numel,
                               int in_dtype, hpc_stream_t stream);
int hpc_moe_gather_rows_async(const void* x, const void* topk_pos, int num_tokens, int num_topk,
                              int hidden, void* x_gathered, hpc_stream_t stream);
int hpc_fuse_moe_pertensor_async(void* y_ptr, void* workspace, const void* x_ptr,
                                 const void* gate_up_weight_ptr, const void* down_weight_ptr,
                                 const void* gate_up_scale_ptr, const void* down_scale_ptr,
                                 const void* act_and_mul_scale_ptr, const void* topk_ids_ptr,
                                 const void* topk_scale_ptr, const void* shared_output_ptr,
                                 int num_tokens, int hidden_size, int intermediate_size2, int num_topk,
                                 int num_expert, int rank_ep, int use_bf16_mul, hpc_stream_t stream);

/* ---- RoPE + optional QK RMSNorm + paged KV-cache write (producer of the decode-attention inputs) ----
 * reference: rope_norm_store_kv_async / rope_norm_store_kv_fp8_async, src/rope/rope.cu:790-950
 *            (kernels :99, :420), entry src/rope/entry.cc:14-240.
 * qkv bf16 [rows, (Hq + 2*Hkv) * 128]; cos_sin f32 [max_pos, 128] (cos | sin halves, neox pairing);
 * num_seqlen_per_req int32 [num_req] (total length incl. the new tokens), q_index int32 [num_req+1],
 * kvcache_indices int32 [num_req, max_blocks_per_req]; caches [blocks, block_size, Hkv, 128]
 * contiguous inside a block, block stride in elements; out_k / out_v (nullable) bypass the cache
 * ([rows, Hkv, 128]).  qk_norm_policy 0 none / 1 rope then RMSNorm / 2 RMSNorm then rope (eps 1e-6,
 * f32 weights [128]).  The tail of each request's last page and (fp8) its split_k_flag row are zeroed.
 * fp8: q/k/v e4m3; quant_policy 1 = dynamic per-token per-head q scale amax/upper_max written to
 * q_scale (decode [rows, Hq]; prefill [num_req, Hq, max_seqlens_pad]), 2 = static q_scale_inv[0];
 * k, v divided by k_scale[0], v_scale[0].  head dims must be 128. */
int hpc_rope_norm_store_kv_async(void* out_q, void* kcache, void* vcache, void* out_k, void* out_v,
                                 const void* qkv, const void* cos_sin, const void* num_seqlen_per_req,
                                 const void* q_index, const void* kvcache_indices,
                                 const void* q_norm_weight, const void* k_norm_weight,
                                 int64_t kcache_block_stride, int64_t vcache_block_stride, int num_req,
                                 int max_blocks_per_req, int kv_block_size, int num_rows,
                                 int num_q_heads, int num_kv_heads, int qk_head_dim, int v_head_dim,
                                 int is_prefill, int qk_norm_policy, hpc_stream_t stream);
int hpc_rope_norm_store_kv_fp8_async(void* out_q, void* kcache, void* vcache, void* out_k, void* out_v,
                                     void* split_k_flag, void* q_scale, const void* qkv,
                                     const void* cos_sin, const void* num_seqlen_per_req,
                                     const void* q_index, const void* kvcache_indices,
                                     const void* q_norm_weight, const void* k_norm_weight,
                                     const void* k_scale, const void* v_scale, const void* q_scale_inv,
                                     float upper_max, int max_seqlens_pad, int64_t kcache_block_stride,
                                     int64_t vcache_block_stride, int num_req, int max_blocks_per_req,
                                     int kv_block_size, int num_rows, int num_q_heads, int num_kv_heads,
                                     int qk_head_dim, int v_head_dim, int is_prefill, int qk_norm_policy,
                                     int quant_policy, hpc_stream_t stream);

/* ---- router GEMM: y = x (w_high + scale * w_low)^T, bf16 operands, fp32 accumulate ----
 * reference: gemm_bf16xfp32_async, src/gemm/gemm.h:13-17 (kernel src/gemm/sm90/gemm_bf16xfp32.cu:88-410,
 * config src/gemm/sm90/entry.cc:23-84).  x [m,k] bf16, w_high / w_low [n,k] bf16, y [m,n] bf16 or fp32.
 * n % 64 == 0, k % 64 == 0.  splits = hpc_gemm_bf16xfp32_splits(m, n, k, use_splitk); when > 1 the
 * caller provides splitk_y (splits*m*n fp32) and zeroed int32 arrival counters split_flag: for
 * m <= 256 a flat [ceil(m/tm), flag_ld = n/16] array (tm = 16 / 32 / 64 for m <= 16 / 32 / 256), for
 * larger m a [ceil(m/64), flag_ld >= n/64] grid (the tile kernel counts on its first ceil(m/128) rows); the counters are zero
 * again when the call retires. */
int hpc_gemm_bf16xfp32_splits(int m, int n, int k, int use_splitk);
int hpc_gemm_bf16xfp32_async(void* y, void* splitk_y, void* split_flag, const void* x, const void* w_high,
                             const void* w_low, int m, int n, int k, float scale, int use_fp32_output,
                             int splits, int flag_ld, hpc_stream_t stream);

/* ---- fused softmax + top-k router (the step between the router GEMM and fuse_moe*) ---------------------
 * No reference kernel exists (hpc/gemm.py:16-61 stops at the GEMM; BASELINE north_star names the op); the
 * semantics are pinned to stable PyTorch: ids = the first `topk` entries of a STABLE descending sort of the
 * fp32 logits (ties -> smaller expert id; bit-exact), p = softmax(logits) in fp32,
 * topk_scale = p[ids] (renormalize = 0) or p[ids] / sum(p[ids]) (renormalize = 1), best expert first.
 * logits f32 [num_tokens, ld_logits >= num_expert], 16-byte aligned rows; num_expert % 4 == 0, <= 1024;
 * topk <= 64.  topk_ids int32 [num_tokens, topk], topk_scale f32 [num_tokens, topk] - the tensors
 * hpc_fuse_moe_*_async take. */
int hpc_topk_router_async(int* topk_ids, float* topk_scale, const float* logits, int num_tokens, int num_expert,
                          int64_t ld_logits, int topk, int renormalize, hpc_stream_t stream);

/* bf16 causal prefill: paged KV cache, and contiguous varlen K/V [total_seq, Hkv, 128] (row strides ldK / ldV
 * elements; every q token attends the keys of its request up to itself).
 * reference: attention_with_kvcache_prefill_bf16_async / attention_prefill_bf16_async (src/attention/prefill/prefill.h,
 * entries src/attention/entry.cc:83-150 and :15-81).  seqlens_kvcache counts the request's q tokens too. */
int hpc_attention_with_kvcache_prefill_bf16_async(
    void* y, const void* q, const void* kcache, const void* vcache, const void* cu_seqlens_q, const void* block_ids,
    const void* seqlens_kvcache, int num_batch, int max_seqlens_q, int num_dim_qk, int num_dim_v, int num_head_q,
    int num_head_kv, int block_size, int num_seq_max_blocks, int ldY, int ldQ, int64_t kcache_block_stride,
    int64_t kcache_token_stride, int64_t kcache_head_stride, int64_t vcache_block_stride,
    int64_t vcache_token_stride, int64_t vcache_head_stride, hpc_stream_t stream);
int hpc_attention_prefill_bf16_async(void* y, const void* q, const void* k, const void* v, const void* cu_seqlens_q,
                                     int num_batch, int max_seqlens_q, int num_dim_qk, int num_dim_v, int num_head_q,
                                     int num_head_kv, int ldY, int ldQ, int ldK, int ldV, hpc_stream_t stream);

/* Block-sparse form (reference attention_with_kvcache_blocksparse_prefill_fp8, src/attention/entry.cc:264-409):
 * block_mask uint8 [B, Hq, ceil(max_seqlens_q/128) = mask_tiles_m, mask_tiles_kv] over 128 (q positions of the
 * request) x 128 (kv tokens) tiles, non-zero = attend; null = dense.  128 % block_size == 0. */
int hpc_attention_with_kvcache_blocksparse_prefill_fp8_async(
    void* y, const void* q, const void* kcache, const void* vcache, const void* qscale, const void* kscale,
    const void* vscale, const void* cu_seqlens_q, const void* block_ids, const void* seqlens_kvcache,
    const void* block_mask, int mask_tiles_m, int mask_tiles_kv, int quant_type, int num_batch, int max_seqlens_q,
    int max_seqlens_q_pad, int num_dim_qk, int num_dim_v, int num_head_q, int num_head_kv, int block_size,
    int num_seq_max_blocks, int ldY, int ldQ, int64_t kcache_block_stride, int64_t kcache_token_stride,
    int64_t kcache_head_stride, int64_t vcache_block_stride, int64_t vcache_token_stride,
    int64_t vcache_head_stride, int64_t kscale_block_stride_bytes, int64_t kscale_row_stride_bytes,
    int64_t kscale_head_stride_bytes, hpc_stream_t stream);

/* Masked (DeepEP-layout) activation + quant: gate_up bf16 [num_expert * rows_per_expert, 2 I]; only the first
 * num_per_expert[e] rows of every expert are computed (others untouched).  Per-tensor form: e4m3(silu(g) u scale[0]);
 * blockwise form: per-128 scale = amax / 448 to out_scale [rows, I/128], q = e4m3(a / (scale + 1e-8)).
 * reference: masked_act_mul_and_quant_async / masked_act_mul_and_blockwise_quant_async (src/activation/activation.h,
 * entry src/activation/entry.cc:50-110). */
int hpc_masked_act_mul_and_quant_async(void* out, const void* gate_up, const void* scale, const void* num_per_expert,
                                       int num_total_tokens, int intermediate_size, int num_tokens_per_expert,
                                       hpc_stream_t stream);
int hpc_masked_act_mul_and_blockwise_quant_async(void* out, void* out_scale, const void* gate_up,
                                                 const void* num_per_expert, int num_total_tokens,
                                                 int intermediate_size, int num_tokens_per_expert,
                                                 hpc_stream_t stream);

/* x_scale [rows, n = K/128] -> transposed, tile-padded, compact [n, m] layout that
 * hpc_group_gemm_blockwise_fp8_async reads (DeepEP-format inputs).
 * reference: reformat_x_scale_async, src/group_gemm/group_gemm.h:27-29 (entry src/group_gemm/entry.cc:170-222). */
int hpc_reformat_x_scale_async(void* output, const void* x_scale, const void* seqlens, const void* cu_seqlens,
                               int num_group, int m, int n, int tilem, hpc_stream_t stream);

/* ---- paged-KV causal prefill attention, FP8 ----
 * reference: attention_with_kvcache_prefill_{qpertoken_perhead_kvpertensor,qkpertoken_perhead_vperhead}_fp8_async
 *            (src/attention/prefill/prefill.h, entry src/attention/entry.cc:152-262).
 * q e4m3 [total_q, Hq, 128]; caches as in hpc_attention_decode_fp8_async (strides in elements); qscale f32
 * [B, Hq, max_seqlens_q_pad]; cu_seqlens_q int32 [B+1]; seqlens_kvcache int32 [B] = cached tokens INCLUDING
 * the request's q tokens (q row s attends keys j <= L_b - Sq_b + s); y bf16 [total_q, Hq, 128].
 * quant_type 1: kscale/vscale f32 [1]; 0: kscale = K-scale tail rows of the cache (strides in bytes),
 * vscale f32 [Hkv].  head dims 128, page size 16/32/64, Hq/Hkv in {1,2,4,8,16}. */
int hpc_attention_with_kvcache_prefill_fp8_async(
    void* y, const void* q, const void* kcache, const void* vcache, const void* qscale, const void* kscale,
    const void* vscale, const void* cu_seqlens_q, const void* block_ids, const void* seqlens_kvcache,
    int quant_type, int num_batch, int max_seqlens_q, int max_seqlens_q_pad, int num_dim_qk, int num_dim_v,
    int num_head_q, int num_head_kv, int block_size, int num_seq_max_blocks, int ldY, int ldQ,
    int64_t kcache_block_stride, int64_t kcache_token_stride, int64_t kcache_head_stride,
    int64_t vcache_block_stride, int64_t vcache_token_stride, int64_t vcache_head_stride,
    int64_t kscale_block_stride_bytes, int64_t kscale_row_stride_bytes, int64_t kscale_head_stride_bytes,
    hpc_stream_t stream);

/* ---- fused sampler (end of the decode step) ----
 * reference: fused_sampler_async / fused_sampler_temperature_async, src/sampler/sampler.h:17-45
 *            (kernels src/sampler/fused_sampler.cu, fused_sampler_temperature.cu; entry src/sampler/entry.cc).
 * logits [B, V] fp32 (dtype 0) or bf16 (1), inner stride 1, row stride in elements; V % 8 == 0, V < 2^20.
 * Pipeline: repetition penalty (bit mask rows selected by slot_id) -> temperature -> [softmax over V,
 * policy 1] -> top-k (k <= max_topk in {32, 64}; 0 / out of range = max_topk) -> [softmax over the top-k,
 * policy 2] -> top-p -> Gumbel-max (gumbel_noise [B, V] fp32, or Philox noise from rng_seed when null) ->
 * token_ids int32 [B] and the sampled token's bit OR-ed into its penalty row.  Ties resolve towards the
 * smaller token id.  workspace: hpc_fused_sampler_workspace_bytes (full sampler) or
 * batch_size * hpc_sampler_segments(V) * 8 bytes (temperature path). */
int hpc_sampler_segments(int vocab_size);
int64_t hpc_fused_sampler_workspace_bytes(int batch_size, int vocab_size, int max_topk);
int hpc_fused_sampler_async(void* token_ids, void* workspace, const void* logits, int logits_dtype,
                            void* penalty_mask, int64_t penalty_mask_row_bytes, const void* slot_id,
                            const void* repetition_penalty, float repetition_penalty_val,
                            const void* temperature, float temperature_val, int softmax_policy,
                            const void* topk, int topk_int_bytes, int topk_val, const void* topp,
                            float topp_val, const void* gumbel_noise, int batch_size, int vocab_size,
                            int64_t logits_row_stride, int max_topk, uint64_t rng_seed, hpc_stream_t stream);
int hpc_fused_sampler_temperature_async(void* token_ids, void* workspace, const void* logits,
                                        int logits_dtype, int64_t logits_row_stride, const void* temperature,
                                        float temperature_val, const void* gumbel_noise,
                                        const void* draft_token_ids, int batch_size, int vocab_size,
                                        uint64_t rng_seed, hpc_stream_t stream);

/* ---- communicator: socket rendezvous + symmetric device buffers (HIP IPC over xGMI) ---------------
 * reference: src/communicator/{communicator,channel,listener,connector,protocol}.cc (rank-0 star over
 *            an abstract unix socket "unix://name" / bare name, or "tcp://ip:port"),
 *            multicast_communicator.cc:21-153 (CreateTensorSync), entry.cc:18-90 (torch class).
 * hpc_comm_create returns a handle > 0 (negative = error); device_id < 0 = no HIP device (host-only
 * rendezvous, used by the CPU tests).  hpc_comm_create_tensor_sync is collective: allocates nbytes of
 * uncached device memory on every rank, exchanges IPC handles, ptrs_out[r] = rank r's buffer mapped
 * in this process (there is no multicast object on xGMI).  hpc_comm_lookup_peers translates an
 * address inside a local symmetric buffer into the same offset of every rank's buffer (returns the
 * world size or -1); hpc_comm_region_bytes_left returns the bytes from `ptr` to the end of the local
 * symmetric buffer that contains it (-1 if none) - the entries use it to learn a signal pad's capacity. */
int hpc_comm_create(int rank, int world_size, int device_id, const char* name);
int hpc_comm_destroy(int handle);
int hpc_comm_barrier(int handle);
int hpc_comm_allgather(int handle, const void* in, int64_t nbytes, void* out);
int hpc_comm_info(int handle, int* rank, int* world_size, int* device_id);
int hpc_comm_create_tensor_sync(int handle, int64_t nbytes, void** ptrs_out);
int hpc_comm_lookup_peers(const void* ptr, void** peer_ptrs, int* rank_out);
int64_t hpc_comm_region_bytes_left(const void* ptr);

/* ---- fused AllReduce + residual + RMSNorm (bf16) over peer memory ------------------------------------
 * reference: fuse_allreduce_rmsnorm_high_throughput_async, src/allreduce/
 *            fuse_allreduce_rmsnorm_high_throughput.h:11-17 (kernel .cu:15-99), and
 *            fuse_allreduce_rmsnorm_low_latency_async, fuse_allreduce_rmsnorm_low_latency.h:29-49,503-504.
 * R = bf16(sum_r x_r + residual); out = bf16(float(R) * rsqrt(mean(R^2) + eps) * w).
 * High throughput: this rank owns `num_rows` token rows (slices may differ between ranks); peer_x_ptrs[p] /
 *   peer_out_ptrs[p] address those rows inside rank p's symmetric input / output buffers,
 *   peer_signal_ptrs[p] rank p's signal pad of `signal_pad_words` zero-initialised uint32 (block b uses
 *   words [b * stride, b * stride + world_size), stride = hpc_fuse_allreduce_rmsnorm_high_throughput_signal_stride(
 *   world_size, grid, signal_pad_words): the blocks' flag groups are spread over the whole pad in 64-byte units - packed
 *   together they sit behind one channel of the uncached memory and every barrier access serialises there).
 *   The barriers pair block b of every rank, so the grid is a
 *   function of rank-invariant inputs only: hpc_fuse_allreduce_rmsnorm_high_throughput_grid(world_size,
 *   num_max_blocks, signal_pad_words) = min(max(num_max_blocks, 512), signal_pad_words / world_size); all
 *   ranks must pass the same num_max_blocks and pad size.  -2 when the pad cannot hold one block.
 *   hidden <= 16384, world_size <= 8.  (The reference supports H in {4096,5120,7168}.)
 * Low latency (Lamport, token t owned by rank t % world_size): data_buffer_ptrs_dev = device int64
 *   table of the ranks' workspace bases, workspace = 3 slots x 2 stages, pre-filled with 0x80000000
 *   words by the caller, buffer_flags_dev = 9 x uint32 {cur, dirty, bytes per slot, 0, bytes to clear
 *   x4, arrive} advanced on the device.
 * Every spin on a peer is bounded (~seconds).  A spin that gives up bumps a counter in pinned host memory:
 *   hpc_allreduce_timeouts reads it without synchronising any stream (0 in a healthy run), and from then on
 *   both entries return HPC_ERR_TIMEOUT instead of launching - after a lost peer the Lamport slots / signal
 *   pads hold leftovers, so the handles must be re-created; hpc_allreduce_reset_timeouts re-arms the entries. */
int hpc_fuse_allreduce_rmsnorm_high_throughput_grid(int world_size, int num_max_blocks, int signal_pad_words);
int hpc_fuse_allreduce_rmsnorm_high_throughput_signal_stride(int world_size, int grid, int signal_pad_words);
int hpc_fuse_allreduce_rmsnorm_high_throughput_async(
    const void* const* peer_x_ptrs, void* const* peer_out_ptrs, void* const* peer_signal_ptrs,
    const void* residual_ptr, void* out_residual_ptr, const void* weight_ptr, float rms_norm_eps,
    int num_rows, int hidden_size, int rank, int world_size, int num_max_blocks, int signal_pad_words,
    hpc_stream_t stream);
int hpc_fuse_allreduce_rmsnorm_low_latency_async(
    void* output_ptr, void* residual_out_ptr, const void* input_ptr, const void* data_buffer_ptrs_dev,
    void* local_workspace_ptr, void* buffer_flags_dev, const void* residual_in_ptr,
    const void* weight_ptr, float rms_norm_eps, int num_tokens, int hidden_size, int rank,
    int world_size, int64_t workspace_bytes, hpc_stream_t stream);
/* The same entry with the two mode flags of the reference op (src/allreduce/entry.cc:84: `rmsnorm_fusion`, `use_two_shot`):
 *   rmsnorm_fusion = 0: output = the reduced rows (bf16), no residual / norm (residual and weight pointers may be null);
 *   use_two_shot   = 0: the one-shot Lamport form - every rank pushes its rows to every rank and reduces them itself (the
 *   reference accepts the flag and runs its two-shot kernel either way; results here are bit-identical between the forms).
 *   One-shot workspace: 3 x num_tokens x world_size x hidden x 2 bytes (-2 when workspace_bytes is smaller). */
int hpc_allreduce_low_latency_async(
    void* output_ptr, void* residual_out_ptr, const void* input_ptr, const void* data_buffer_ptrs_dev,
    void* local_workspace_ptr, void* buffer_flags_dev, const void* residual_in_ptr,
    const void* weight_ptr, float rms_norm_eps, int num_tokens, int hidden_size, int rank,
    int world_size, int64_t workspace_bytes, int rmsnorm_fusion, int use_two_shot, hpc_stream_t stream);
int hpc_allreduce_timeouts(void);
int hpc_allreduce_reset_timeouts(void);

#ifdef __cplusplus
}
#endif
#endif /* HPC_AMD_H_ */
