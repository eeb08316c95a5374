// Shared device helpers for the gfx950 (MI355X / CDNA4) kernels of hpc-ops_amd.
// Wave64 only. No CUDA compatibility layer: this header is HIP-for-CDNA4 code.
//
// Replaces the role of the reference's src/utils/utils.cuh (vec_t / load / store /
// warp reductions / fp8 conversions) with CDNA4 idioms: 64-lane DPP/shuffle
// reductions, 16-byte vector global accesses, OCP e4m3 hardware conversions.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

namespace hpc {

constexpr int kWave = 64;

typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(8))) short s16x8;
typedef __attribute__((ext_vector_type(4))) short s16x4;
typedef __attribute__((ext_vector_type(2))) int i32x2;
typedef __attribute__((ext_vector_type(4))) int i32x4;
typedef __attribute__((ext_vector_type(8))) int i32x8;
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;
typedef __attribute__((ext_vector_type(2))) unsigned int u32x2;

// ---- bf16 <-> f32 (bit tricks; RNE on the way down, NaN preserved) -------------------
__device__ __forceinline__ float bf16_to_f32(uint16_t h) {
  return __uint_as_float(static_cast<uint32_t>(h) << 16);
}
__device__ __forceinline__ float bf16lo_to_f32(uint32_t w) { return __uint_as_float(w << 16); }
__device__ __forceinline__ float bf16hi_to_f32(uint32_t w) {
  return __uint_as_float(w & 0xffff0000u);
}
__device__ __forceinline__ uint16_t f32_to_bf16(float f) {
  uint32_t u = __float_as_uint(f);
  if ((u & 0x7fffffffu) > 0x7f800000u) return static_cast<uint16_t>((u >> 16) | 0x40u);  // NaN
  u += 0x7fffu + ((u >> 16) & 1u);
  return static_cast<uint16_t>(u >> 16);
}
// gfx950 has the conversion in hardware (v_cvt_pk_bf16_f32: RNE, NaN stays NaN): one instruction per pair
typedef __attribute__((ext_vector_type(2))) float f32x2;
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2;
__device__ __forceinline__ uint32_t pack_bf16x2(float lo, float hi) {
  const bf16x2 b = __builtin_convertvector(f32x2{lo, hi}, bf16x2);
  return __builtin_bit_cast(uint32_t, b);
}

// ---- OCP e4m3fn conversions (gfx950 hardware cvt; saturating to +-448 like the reference's
// __nv_fp8 casts, SURVEY 7 "hard parts") ------------------------------------------------
// Two clamps.  clamp_e4m3: for values that cannot be NaN by construction (softmax probabilities):
// fmaxf(NaN, -448) returns -448, so a NaN would come out as -448.  clamp_e4m3_nan: for quantisers of
// caller data (activations, q/k/v, RMSNorm output) - NaN stays NaN (e.g. 0 * inf when a row's amax is
// 0) like the reference's saturating __nv_fp8_e4m3 cast and torch's .to(float8_e4m3fn).
__device__ __forceinline__ float clamp_e4m3(float x) {
  return __builtin_fminf(__builtin_fmaxf(x, -448.0f), 448.0f);
}
__device__ __forceinline__ float clamp_e4m3_nan(float x) {
  const float c = __builtin_fminf(__builtin_fmaxf(x, -448.0f), 448.0f);
  return x != x ? x : c;
}
// A NaN leaves as the byte 0x7f, like the reference's cast (cvt.rn.satfinite.e4m3x2.f32) and torch's: on gfx950
// v_cvt_pk_fp8_f32 was observed to turn a (positive, quiet) NaN into 0xff - the other NaN encoding of e4m3fn.
__device__ __forceinline__ uint32_t e4m3_nan_byte(uint32_t packed, float x, int shift) {
  return x != x ? (packed & ~(0xffu << shift)) | (0x7fu << shift) : packed;
}
// packs (a, b) into the low 16 bits of the result.
__device__ __forceinline__ uint32_t cvt_pk_e4m3(float a, float b) {
  const uint32_t r = static_cast<uint32_t>(
                         __builtin_amdgcn_cvt_pk_fp8_f32(clamp_e4m3_nan(a), clamp_e4m3_nan(b), 0, false)) &
                     0xffffu;
  return e4m3_nan_byte(e4m3_nan_byte(r, a, 0), b, 8);
}
// probabilities (never NaN): the cheap clamp
__device__ __forceinline__ uint32_t cvt_4xe4m3(float a, float b, float c, float d) {
  int r = __builtin_amdgcn_cvt_pk_fp8_f32(clamp_e4m3(a), clamp_e4m3(b), 0, false);
  r = __builtin_amdgcn_cvt_pk_fp8_f32(clamp_e4m3(c), clamp_e4m3(d), r, true);
  return static_cast<uint32_t>(r);
}
// caller data: NaN-preserving
__device__ __forceinline__ uint32_t quant_4xe4m3(float a, float b, float c, float d) {
  int r = __builtin_amdgcn_cvt_pk_fp8_f32(clamp_e4m3_nan(a), clamp_e4m3_nan(b), 0, false);
  r = __builtin_amdgcn_cvt_pk_fp8_f32(clamp_e4m3_nan(c), clamp_e4m3_nan(d), r, true);
  return e4m3_nan_byte(e4m3_nan_byte(e4m3_nan_byte(e4m3_nan_byte(static_cast<uint32_t>(r), a, 0), b, 8), c, 16), d, 24);
}
// byte `sel` (0..3) of w -> f32
template <int kSel>
__device__ __forceinline__ float e4m3_to_f32(uint32_t w) {
  return __builtin_amdgcn_cvt_f32_fp8(static_cast<int>(w), kSel);
}

// ---- wave64 reductions -----------------------------------------------------------------
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
  return v;
}

// ---- 16-byte global accesses -------------------------------------------------------------
__device__ __forceinline__ u32x4 ld16(const void* p) { return *reinterpret_cast<const u32x4*>(p); }
__device__ __forceinline__ void st16(void* p, u32x4 v) { *reinterpret_cast<u32x4*>(p) = v; }
__device__ __forceinline__ u32x4 ld16_nt(const void* p) {
  return __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(p));
}

// ---- wave-uniform reads of read-only tables (task map, page table) ----------------------------------
// Reading through the constant address space tells hipcc the data is invariant, so a uniform
// address becomes an s_load (lgkmcnt) instead of a global_load whose vmcnt wait would drain the
// KV stream that is in flight.
typedef const int __attribute__((address_space(4))) * cint_ptr;
__device__ __forceinline__ cint_ptr as_const(const int* p) {
  return (cint_ptr)(reinterpret_cast<uintptr_t>(p));
}

// ---- buffer (SRD) loads: wave-uniform 64-bit base in SGPRs + 32-bit lane offset -------------------
// The base must be provably wave-uniform (built from kernargs / readfirstlane values), otherwise
// hipcc wraps every access in a waterfall loop.
typedef __amdgpu_buffer_rsrc_t rsrc_t;
// num_records = 0 makes every access out of range: loads return 0 and fetch nothing.
__device__ __forceinline__ rsrc_t make_rsrc(const void* base, unsigned num_records = 0xffffffffu) {
  return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(base), 0, num_records, 0x00020000);
}
// Pin a pointer that IS wave-uniform (but that hipcc's divergence analysis cannot prove uniform) into
// SGPRs; without it a descriptor built from the pointer costs a waterfall loop per access.
__device__ __forceinline__ const uint8_t* uniform_ptr(const void* p) {
  const uint64_t v = reinterpret_cast<uint64_t>(p);
  const uint32_t lo = __builtin_amdgcn_readfirstlane(static_cast<uint32_t>(v));
  const uint32_t hi = __builtin_amdgcn_readfirstlane(static_cast<uint32_t>(v >> 32));
  return reinterpret_cast<const uint8_t*>(static_cast<uint64_t>(lo) | (static_cast<uint64_t>(hi) << 32));
}
// voff: per-lane byte offset (may carry a compile-time constant), soff: wave-uniform byte offset
// kAux: cache policy bits of the buffer instruction (0 = default, 2 = nt "streaming, read once")
template <int kAux = 0>
__device__ __forceinline__ u32x4 buf_ld16(rsrc_t rs, int voff, int soff) {
  return __builtin_amdgcn_raw_buffer_load_b128(rs, voff, soff, kAux);
}
template <int kAux = 0>
__device__ __forceinline__ u32x2 buf_ld8(rsrc_t rs, int voff, int soff) {
  return __builtin_amdgcn_raw_buffer_load_b64(rs, voff, soff, kAux);
}

}  // namespace hpc

// Error convention of the C-ABI (include/hpc_amd.h): 0 = launched, negative = refused.
#define HPC_OK 0
#define HPC_ERR_UNSUPPORTED (-1)
#define HPC_ERR_INVALID (-2)
#define HPC_ERR_LAUNCH (-3)
#define HPC_ERR_TIMEOUT (-4)

#define HPC_CHECK_LAUNCH()                               \
  do {                                                   \
    if (hipGetLastError() != hipSuccess) return HPC_ERR_LAUNCH; \
  } while (0)
