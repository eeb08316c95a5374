// Matrix-pipe ceiling probe (development tool, tools/moe_clock.py): what does the chip SUSTAIN on v_mfma_f32_16x16x128_f8f6f4 - the
// instruction the grouped GEMMs issue - with nothing else going on?  Every wave loops over 32 back-to-back MFMAs on register
// operands (eight independent accumulator tiles: no dependent issue), no memory, no LDS, no VALU work; 2 waves per SIMD like
// the GEMM kernels.  `mode` 0: operands = pseudo-random bytes (every multiplier input toggles, like real fp8 weights and
// activations), 1: all-zero operands (the same instruction stream, idle multipliers).  The spec peak (5 PF dense) is this loop
// at 2.4 GHz; the difference between the two modes is what the power limit costs.
#include <hip/hip_runtime.h>
#include <stdint.h>
typedef int i32x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

__global__ __launch_bounds__(512, 1) void mfma_probe_kernel(float* out, int iters, int mode) {
  uint32_t s = (blockIdx.x * 512u + threadIdx.x) * 2654435761u + 12345u;
  auto rnd = [&]() {
    s = s * 1664525u + 1013904223u;
    // e4m3 bytes with exponents in the normal range (no NaN: 0x7f / 0xff), sign and mantissa random
    uint32_t v = s ^ (s >> 13);
    v &= 0xbfbfbfbfu;  // clear exponent msb of every byte: |x| < 2
    return mode == 1 ? 0 : static_cast<int>(v);
  };
  i32x8 a[4], b[2];
  for (int i = 0; i < 4; ++i)
    for (int j = 0; j < 8; ++j) a[i][j] = rnd();
  for (int i = 0; i < 2; ++i)
    for (int j = 0; j < 8; ++j) b[i][j] = rnd();
  f32x4 acc[8];
  for (int i = 0; i < 8; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int r = 0; r < 4; ++r)
#pragma unroll
      for (int i = 0; i < 8; ++i)
        acc[i] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(a[i & 3], b[i >> 2], acc[i], 0, 0, 0, 0, 0, 0);
    // keep the accumulators bounded without VALU work in the loop: nothing (fp32 sums of 128 products < 4 stay finite for
    // any iteration count the host asks for: |acc| grows by < 512 per MFMA)
  }
  float t = 0.f;
  for (int i = 0; i < 8; ++i) t += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
  if (t == 123.456f) out[0] = t;  // never true in practice: keeps the loop alive
}

// flops of one launch: workgroups x 8 waves x iters x 32 MFMAs x 2 * 16 * 16 * 128
extern "C" int mfma_probe_launch(void* out, int workgroups, int iters, int mode, void* stream) {
  mfma_probe_kernel<<<workgroups, 512, 0, static_cast<hipStream_t>(stream)>>>(static_cast<float*>(out), iters, mode);
  return hipGetLastError() == hipSuccess ? 0 : -1;
}
