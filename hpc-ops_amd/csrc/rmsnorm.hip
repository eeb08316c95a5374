// fused_rmsnorm_with_scale for gfx950: bf16 rows -> RMSNorm * w -> e4m3 (/scale), optional fp32
// and second-scale e4m3 outputs ("MoE" mode).
//
// Replaces reference src/normalization/fused_rmsnorm_with_scale.cu:14-225 (kernel + launcher).
// HBM-bound: one pass, 16-byte loads, row held in registers, wave64 shuffle
// reduce + one LDS exchange between the waves that share a row.
#include "hpc_common.h"
#include "../../include/hpc_amd.h"

namespace hpc {
namespace {

// kTPR threads cooperate on one row (kTPR in {64,128,256}); kNV 16-byte vectors per thread.
template <int kTPR, int kNV, bool kMoe>
__global__ __launch_bounds__(256) void rmsnorm_scale_kernel(
    const uint16_t* __restrict__ x, const uint16_t* __restrict__ w, uint8_t* __restrict__ out_fp8,
    float* __restrict__ out_f32, uint8_t* __restrict__ out_fp8_2, const float* __restrict__ scale,
    float eps, int rows, int hidden) {
  constexpr int kRowsPerBlock = 256 / kTPR;
  constexpr int kWavesPerRow = kTPR / kWave;
  __shared__ float red[4];

  const int tid = threadIdx.x;
  const int row_in_block = tid / kTPR;
  const int t = tid % kTPR;
  const int64_t row = static_cast<int64_t>(blockIdx.x) * kRowsPerBlock + row_in_block;
  const bool row_ok = row < rows;
  const int nvec = hidden >> 3;

  u32x4 xv[kNV];
  float ss = 0.f;
#pragma unroll
  for (int i = 0; i < kNV; ++i) {
    const int v = t + i * kTPR;
    xv[i] = u32x4{0u, 0u, 0u, 0u};
    if (row_ok && v < nvec) xv[i] = ld16_nt(x + row * hidden + v * 8);
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const float a = bf16lo_to_f32(xv[i][j]), b = bf16hi_to_f32(xv[i][j]);
      ss = fmaf(a, a, ss);
      ss = fmaf(b, b, ss);
    }
  }
  ss = wave_sum(ss);
  if constexpr (kWavesPerRow > 1) {
    const int wave = tid >> 6;
    if ((tid & 63) == 0) red[wave] = ss;
    __syncthreads();
    const int w0 = (wave / kWavesPerRow) * kWavesPerRow;
    ss = 0.f;
#pragma unroll
    for (int k = 0; k < kWavesPerRow; ++k) ss += red[w0 + k];
  }
  const float rms = rsqrtf(ss / static_cast<float>(hidden) + eps);
  const float inv0 = 1.0f / scale[0];
  const float inv1 = kMoe ? 1.0f / scale[1] : 1.0f;
  if (!row_ok) return;

#pragma unroll
  for (int i = 0; i < kNV; ++i) {
    const int v = t + i * kTPR;
    if (v >= nvec) continue;
    const u32x4 wv = ld16(w + v * 8);
    float y[8];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      y[2 * j] = bf16lo_to_f32(xv[i][j]) * rms * bf16lo_to_f32(wv[j]);
      y[2 * j + 1] = bf16hi_to_f32(xv[i][j]) * rms * bf16hi_to_f32(wv[j]);
    }
    const int64_t o = row * hidden + v * 8;
    u32x2 q;
    q[0] = quant_4xe4m3(y[0] * inv0, y[1] * inv0, y[2] * inv0, y[3] * inv0);
    q[1] = quant_4xe4m3(y[4] * inv0, y[5] * inv0, y[6] * inv0, y[7] * inv0);
    *reinterpret_cast<u32x2*>(out_fp8 + o) = q;
    if constexpr (kMoe) {
      *reinterpret_cast<f32x4*>(out_f32 + o) = f32x4{y[0], y[1], y[2], y[3]};
      *reinterpret_cast<f32x4*>(out_f32 + o + 4) = f32x4{y[4], y[5], y[6], y[7]};
      q[0] = quant_4xe4m3(y[0] * inv1, y[1] * inv1, y[2] * inv1, y[3] * inv1);
      q[1] = quant_4xe4m3(y[4] * inv1, y[5] * inv1, y[6] * inv1, y[7] * inv1);
      *reinterpret_cast<u32x2*>(out_fp8_2 + o) = q;
    }
  }
}

template <int kTPR, int kNV>
int launch(const void* x, const void* w, void* o8, void* of, void* o8b, const void* scale,
           float eps, int rows, int hidden, bool moe, hipStream_t stream) {
  constexpr int kRowsPerBlock = 256 / kTPR;
  const int grid = (rows + kRowsPerBlock - 1) / kRowsPerBlock;
  if (moe) {
    rmsnorm_scale_kernel<kTPR, kNV, true><<<grid, 256, 0, stream>>>(
        (const uint16_t*)x, (const uint16_t*)w, (uint8_t*)o8, (float*)of, (uint8_t*)o8b,
        (const float*)scale, eps, rows, hidden);
  } else {
    rmsnorm_scale_kernel<kTPR, kNV, false><<<grid, 256, 0, stream>>>(
        (const uint16_t*)x, (const uint16_t*)w, (uint8_t*)o8, nullptr, nullptr, (const float*)scale,
        eps, rows, hidden);
  }
  HPC_CHECK_LAUNCH();
  return HPC_OK;
}

}  // namespace
}  // namespace hpc

extern "C" int hpc_fused_rmsnorm_with_scale_async(const void* input, const void* weight,
                                                  void* output_fp8, void* output_fp32,
                                                  void* output_fp8_scale2, const void* scale,
                                                  float eps, int batch_size, int hidden_state,
                                                  int is_moe, hipStream_t stream) {
  using namespace hpc;
  if (batch_size <= 0) return HPC_OK;
  if (hidden_state <= 0 || (hidden_state & 7)) return HPC_ERR_UNSUPPORTED;
  if (is_moe && (!output_fp32 || !output_fp8_scale2)) return HPC_ERR_INVALID;
  const int nvec = hidden_state / 8;
  const bool moe = is_moe != 0;
#define HPC_RMS_CASE(TPR, NV)                                                                  \
  return launch<TPR, NV>(input, weight, output_fp8, output_fp32, output_fp8_scale2, scale, eps, \
                         batch_size, hidden_state, moe, stream)
  if (nvec <= 64) HPC_RMS_CASE(64, 1);
  if (nvec <= 128) HPC_RMS_CASE(128, 1);
  if (nvec <= 256) HPC_RMS_CASE(256, 1);
  if (nvec <= 512) HPC_RMS_CASE(256, 2);
  if (nvec <= 768) HPC_RMS_CASE(256, 3);
  if (nvec <= 1024) HPC_RMS_CASE(256, 4);
  if (nvec <= 2048) HPC_RMS_CASE(256, 8);
#undef HPC_RMS_CASE
  return HPC_ERR_UNSUPPORTED;
}
