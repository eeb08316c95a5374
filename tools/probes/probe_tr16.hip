// Development probe: what does ds_read_b64_tr_b16 (gfx950) return?  Hypothesis (by analogy with the 8-bit form that
// attention_decode_v2.hip uses): within a 16-lane group, lane i supplies the address of 4 consecutive 16-bit elements;
// lanes 4j .. 4j+3 together supply ROW j (16 elements) of a 4 x 16 tile; lane i receives COLUMN i of it (4 elements,
// rows 0..3).  The probe fills LDS with element value = row * 64 + col, uses exactly that addressing and prints what
// every lane of the first wave got.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
typedef short v4i16 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) v4i16 lds_v4i16;
__global__ void k(uint16_t* out) {
  __shared__ uint16_t tile[64 * 64];  // [row][64 cols]
  for (int i = threadIdx.x; i < 64 * 64; i += 64) tile[i] = static_cast<uint16_t>(i);  // value = row * 64 + col
  __syncthreads();
  const int lane = threadIdx.x, i = lane & 15, g = lane >> 4;
  // group g reads rows 4g .. 4g+3 (row = 4g + i / 4), columns (i % 4) * 4 .. + 3
  const uint16_t* p = &tile[(4 * g + (i >> 2)) * 64 + (i & 3) * 4];
  const v4i16 v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_v4i16*)(reinterpret_cast<uintptr_t>(p)));
  for (int e = 0; e < 4; ++e) out[lane * 4 + e] = static_cast<uint16_t>(v[e]);
}
int main() {
  uint16_t* d;
  hipMalloc(&d, 64 * 4 * 2);
  k<<<1, 64>>>(d);
  uint16_t h[256];
  hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
  int ok = 1;
  for (int lane = 0; lane < 64; ++lane) {
    const int i = lane & 15, g = lane >> 4;
    printf("lane %2d:", lane);
    for (int e = 0; e < 4; ++e) {
      const int row = h[lane * 4 + e] / 64, col = h[lane * 4 + e] % 64;
      printf(" (r%d,c%d)", row, col);
      if (row != 4 * g + e || col != i) ok = 0;
    }
    printf("\n");
  }
  printf("hypothesis (lane i of group g gets column i of rows 4g..4g+3): %s\n", ok ? "CONFIRMED" : "WRONG");
  return 0;
}
