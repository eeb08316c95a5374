// Hardware-semantics probe (development tool, not product): prints the lane mapping of
// ds_read_b64_tr_b8 / ds_read_b64_tr_b16 and checks the 16x16x32 bf16/fp8 MFMA operand layout
// against a host matmul.  Run on the GPU box:  tools/probes/bin/probe_isa
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <math.h>
typedef __attribute__((ext_vector_type(2))) int v2i;
typedef __attribute__((ext_vector_type(4))) short v4s;
typedef __attribute__((ext_vector_type(4))) float f4;
typedef __attribute__((ext_vector_type(8))) __bf16 bf8;

__global__ void k_tr(unsigned* out8, unsigned* out16) {
  __shared__ __attribute__((aligned(16))) unsigned char lds[2048];
  for (int i = threadIdx.x; i < 2048; i += 64) lds[i] = (unsigned char)(i & 255);
  __syncthreads();
  // tr8: lane piece = bytes [8*lane, 8*lane+8) -> values 8*lane..8*lane+7 (mod 256)
  v2i r = __builtin_amdgcn_ds_read_tr8_b64_v2i32(
      (__attribute__((address_space(3))) v2i*)(lds + threadIdx.x * 8));
  out8[threadIdx.x * 2] = r.x;
  out8[threadIdx.x * 2 + 1] = r.y;
  __syncthreads();
  unsigned short* l16 = (unsigned short*)lds;
  for (int i = threadIdx.x; i < 1024; i += 64) l16[i] = (unsigned short)i;
  __syncthreads();
  v4s q = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
      (__attribute__((address_space(3))) v4s*)(lds + threadIdx.x * 8));
  out16[threadIdx.x * 2] = (unsigned short)q.x | ((unsigned)(unsigned short)q.y << 16);
  out16[threadIdx.x * 2 + 1] = (unsigned short)q.z | ((unsigned)(unsigned short)q.w << 16);
}

// C[16x16] = A[16x32] * B^T where Bt[16x32] (both row-major, k contiguous)
__global__ void k_mfma_bf16(const float* A, const float* Bt, float* C) {
  int l = threadIdx.x;
  bf8 a, b;
  for (int j = 0; j < 8; ++j) {
    a[j] = (__bf16)A[(l & 15) * 32 + (l >> 4) * 8 + j];
    b[j] = (__bf16)Bt[(l & 15) * 32 + (l >> 4) * 8 + j];
  }
  f4 acc = {0, 0, 0, 0};
  acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, acc, 0, 0, 0);
  for (int r = 0; r < 4; ++r) C[((l >> 4) * 4 + r) * 16 + (l & 15)] = acc[r];
}
__global__ void k_mfma_fp8(const float* A, const float* Bt, float* C) {
  int l = threadIdx.x;
  float av[8], bv[8];
  for (int j = 0; j < 8; ++j) {
    av[j] = A[(l & 15) * 32 + (l >> 4) * 8 + j];
    bv[j] = Bt[(l & 15) * 32 + (l >> 4) * 8 + j];
  }
  int a0 = __builtin_amdgcn_cvt_pk_fp8_f32(av[0], av[1], 0, false);
  a0 = __builtin_amdgcn_cvt_pk_fp8_f32(av[2], av[3], a0, true);
  int a1 = __builtin_amdgcn_cvt_pk_fp8_f32(av[4], av[5], 0, false);
  a1 = __builtin_amdgcn_cvt_pk_fp8_f32(av[6], av[7], a1, true);
  int b0 = __builtin_amdgcn_cvt_pk_fp8_f32(bv[0], bv[1], 0, false);
  b0 = __builtin_amdgcn_cvt_pk_fp8_f32(bv[2], bv[3], b0, true);
  int b1 = __builtin_amdgcn_cvt_pk_fp8_f32(bv[4], bv[5], 0, false);
  b1 = __builtin_amdgcn_cvt_pk_fp8_f32(bv[6], bv[7], b1, true);
  long a = ((long)(unsigned)a0) | ((long)a1 << 32);
  long b = ((long)(unsigned)b0) | ((long)b1 << 32);
  f4 acc = {0, 0, 0, 0};
  acc = __builtin_amdgcn_mfma_f32_16x16x32_fp8_fp8(a, b, acc, 0, 0, 0);
  for (int r = 0; r < 4; ++r) C[((l >> 4) * 4 + r) * 16 + (l & 15)] = acc[r];
}
// fp8 saturation / rounding probe
__global__ void k_cvt(const float* in, unsigned* out, int n) {
  int i = threadIdx.x;
  if (i < n) out[i] = __builtin_amdgcn_cvt_pk_fp8_f32(in[i], 0.f, 0, false) & 0xff;
}

int main() {
  unsigned *d8, *d16, h8[128], h16[128];
  hipMalloc(&d8, 512); hipMalloc(&d16, 512);
  k_tr<<<1, 64>>>(d8, d16);
  hipMemcpy(h8, d8, 512, hipMemcpyDeviceToHost);
  hipMemcpy(h16, d16, 512, hipMemcpyDeviceToHost);
  printf("TR8 lane: 8 source byte-indices (value = 8*srclane+byte)\n");
  for (int l = 0; l < 64; ++l) {
    printf("L%02d:", l);
    for (int j = 0; j < 8; ++j) printf(" %4u", (h8[l * 2 + j / 4] >> (8 * (j & 3))) & 255);
    printf("\n");
  }
  printf("TR16 lane: 4 source u16 indices (value = 4*srclane+elem)\n");
  for (int l = 0; l < 64; ++l) {
    printf("L%02d:", l);
    for (int j = 0; j < 4; ++j) printf(" %4u", (h16[l * 2 + j / 2] >> (16 * (j & 1))) & 0xffff);
    printf("\n");
  }
  // mfma check
  float hA[512], hB[512], hC[256], ref[256], *dA, *dB, *dC;
  srand(1);
  for (int i = 0; i < 512; ++i) { hA[i] = (rand() % 9 - 4) * 0.25f; hB[i] = (rand() % 7 - 3) * 0.5f; }
  for (int m = 0; m < 16; ++m) for (int n = 0; n < 16; ++n) { float s = 0; for (int k = 0; k < 32; ++k) s += hA[m*32+k]*hB[n*32+k]; ref[m*16+n] = s; }
  hipMalloc(&dA, 2048); hipMalloc(&dB, 2048); hipMalloc(&dC, 1024);
  hipMemcpy(dA, hA, 2048, hipMemcpyHostToDevice); hipMemcpy(dB, hB, 2048, hipMemcpyHostToDevice);
  k_mfma_bf16<<<1, 64>>>(dA, dB, dC); hipMemcpy(hC, dC, 1024, hipMemcpyDeviceToHost);
  double e = 0; for (int i = 0; i < 256; ++i) e = fmax(e, fabs(hC[i]-ref[i]));
  printf("MFMA bf16 16x16x32 layout max err %g\n", e);
  k_mfma_fp8<<<1, 64>>>(dA, dB, dC); hipMemcpy(hC, dC, 1024, hipMemcpyDeviceToHost);
  e = 0; for (int i = 0; i < 256; ++i) e = fmax(e, fabs(hC[i]-ref[i]));
  printf("MFMA fp8 16x16x32 layout max err %g\n", e);
  float cin[16] = {448.f, 449.f, 464.f, 480.f, 500.f, 1e6f, -1e6f, INFINITY, NAN, 0.0009765625f, 0.001953125f, 0.0015f, 17.f, 18.f, 19.f, -0.f};
  unsigned cout_[16]; float* dcin; unsigned* dcout;
  hipMalloc(&dcin, 64); hipMalloc(&dcout, 64);
  hipMemcpy(dcin, cin, 64, hipMemcpyHostToDevice);
  k_cvt<<<1, 64>>>(dcin, dcout, 16); hipMemcpy(cout_, dcout, 64, hipMemcpyDeviceToHost);
  printf("CVT fp8:"); for (int i = 0; i < 16; ++i) printf(" %g->0x%02x", cin[i], cout_[i]); printf("\n");
  hipDeviceProp_t p; hipGetDeviceProperties(&p, 0);
  printf("CUs %d clock %d kHz name %s arch %s\n", p.multiProcessorCount, p.clockRate, p.name, p.gcnArchName);
  return 0;
}
