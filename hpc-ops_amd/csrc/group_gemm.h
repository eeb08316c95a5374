// Shared argument block of the grouped FP8 GEMM kernels (streaming form: group_gemm_blockwise.hip,
// tiled form: group_gemm_tiled.hip).
#pragma once
#include <stdint.h>

namespace hpc {
namespace ggemm {

struct Args {
  const uint8_t* x;       // [rows, K] e4m3
  const uint8_t* w;       // [G, N, K] e4m3
  const float* xs;        // activation scales, see xs_row_stride / xs_kb_stride
  const float* ws;        // [G, N/128, ws_ld]
  uint16_t* y;            // [M, N] bf16
  const int* seqlens;     // [G]
  const int* cu_seqlens;  // [G]   first row of group g in y (and in x when row_index == null)
  const int* row_index;   // null, or [M] -> row of x / xs for output row m (gather-free MoE)
  const int* col_base;    // null, or [G] (cu_tiles): transposed xs, column = col_base[g]*tile_m + slot
  int N, K, KB, tile_m;
  int ws_group_stride, ws_ntile_stride, ws_kb_stride;  // floats; per-tensor scales: (1, 0, 0)
  int has_xs;                                           // 0: no activation scales (factor 1)
  unsigned x_bytes;                                     // bytes of x (bounds the activation loads)
  long xs_row_stride, xs_kb_stride;                     // in floats
  // fused SiLU(gate) * up + 128-block quantisation in the epilogue of the gate-up GEMM (256 x 256 tile kernel only;
  // N = 2 * inter, a tile = 128 gate rows + the 128 up rows of the same columns): act_out e4m3 [M, inter],
  // act_scale f32 [M, inter / 128]; y is not written
  uint8_t* act_out = nullptr;
  float* act_scale = nullptr;
  // per-tensor form of the same epilogue (act_scale == nullptr): out = e4m3(silu(gate) * up * act_mul_scale[0]), with
  // the bf16-rounded multiply of the reference when use_bf16_mul (reference src/activation/activation.cu:19-75)
  const float* act_mul_scale = nullptr;
  int use_bf16_mul = 0;
  int no_half_tile = 0;  // development (key 21): 1 = the 256 x 256 kernel runs its full body only, 2 = no tail body (<= 64 rows)
  int nt_single = 1;     // 256 x 256 kernel, tail body: non-temporal weight loads for a group's ONLY (<= 64-row) token tile
  int tail_regs = 0;     // development (key 26): 1 = the register-streamed tail body instead of the LDS-ring one
  int item_scan_old = 0; // development (key 43): 1 = the item lookup's scans / lane reads through ds_bpermute (rounds 2-5) instead of DPP + v_readlane
  int ext_rows = 0;      // 256 x 256 kernel: 1 = a group's short tail rides along with its full tiles (group_gemm_p8.hip::locate_item)
  int item_order = 0;    // 256 x 256 kernel: 0 = tail tiles in place, 1 = full tiles first, tail tiles last (group_gemm_p8.hip::locate_item)
  void* prof = nullptr;  // development: s_memtime log of the 256 x 256 kernel's section boundaries (hpc_dev_p8_prof_buffer)
};

}  // namespace ggemm
}  // namespace hpc

// tiled (MFMA-bound) form for large groups; `cu_tiles` = exclusive scan of ceil(seqlens / 128).
int hpc_ggemm_launch_tiled(const hpc::ggemm::Args& a, const int* cu_tiles, int num_group, int m, int n,
                           hipStream_t stream);
int hpc_ggemm_launch_tiled256(const hpc::ggemm::Args& a, const int* cu_tiles, int num_group, int m, int n,
                              hipStream_t stream);
// 256 x 256 tile, staggered wave groups (group_gemm_p8.hip); derives its 256-token tiles from the same scan
int hpc_ggemm_launch_p8(const hpc::ggemm::Args& a, const int* cu_tiles, int num_group, int m, int n,
                        hipStream_t stream);
// would launch_stream_gemm pick the 256 x 256 kernel for this problem? (the fused MoE asks before it fuses)
bool hpc_ggemm_p8_selected(int num_group, int m, int n, int k, const void* cu_tiles128);
