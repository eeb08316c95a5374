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

#ifdef __cplusplus
}
#endif
#endif /* HPC_AMD_H_ */
