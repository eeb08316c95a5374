// Hazard probe (development tool, tools/probe_lds_addr_war.py): a VALU write to the ADDRESS register of a DS read that was
// issued a few instructions earlier - while the other wave of the SIMD runs MFMAs + VALU at raised priority.
//
// Found in round 6 in csrc/group_gemm_p8.hip (ride-along body, hipcc's allocation of the build without -fno-slp-vectorize):
//     ds_read2_b32 v[208:209], v196 offset1:16
//     ds_read2_b32 v[244:245], v196 offset0:32 offset1:48
//     buffer_load_dwordx4 ... lds ; s_load_dword ; buffer_load_dwordx4 ... lds ; s_cmp
//     v_mov_b32 v196, v197            <- the address register is dead for the compiler: reused for a scale
//     s_waitcnt vmcnt(9) lgkmcnt(0) ; s_barrier ; ... v_fma_f32 v182, v190, v196, v182
// and the FMA saw a wrong v196 in lanes 48-63, on some calls, in the load section of a wave whose SIMD partner was inside
// its MMA section (s_setprio 1, back-to-back MFMAs with four FMAs each).  This probe replays the pair with a register of
// known content and reports, per lane quarter, how often (a) the register does not hold what the v_mov wrote and (b) the
// DS read returned something other than the LDS word at the ORIGINAL address.
#include <hip/hip_runtime.h>
#include <stdint.h>

#define WAIT_0 ""
#define WAIT_1 "s_nop 0\n\t"
#define WAIT_2 "s_nop 1\n\t"
#define WAIT_4 "s_nop 3\n\t"
#define WAIT_8 "s_nop 7\n\t"
#define WAIT_16 "s_nop 7\n\ts_nop 7\n\t"
#define FILL_0 ""
#define FILL_4 "ds_read_b128 v[100:103], v90\n\tds_read_b128 v[104:107], v90 offset:2048\n\t" \
               "ds_read_b128 v[108:111], v90 offset:4096\n\tds_read_b128 v[112:115], v90 offset:6144\n\t"
#define FILL_16 FILL_4 FILL_4 FILL_4 FILL_4
#define CLOB "v90", "v91", "v92", "v93", "v94", "v95", "v96", "v97", "v100", "v101", "v102", "v103", "v104", "v105", "v106", \
             "v107", "v108", "v109", "v110", "v111", "v112", "v113", "v114", "v115"

// v91 = address (lane * 4), v92 = the value the v_mov writes, results: v[94:95] <- LDS, v96 <- v91 after the write
#define EXPERIMENT(NAME, FILL, WAIT)                                                                                   \
  __device__ __forceinline__ void NAME(int iters, int lane_addr, int pattern, int* bad_reg, int* bad_data, int expect0,  \
                                       int expect1) {                                                                     \
    for (int it = 0; it < iters; ++it) {                                                                                  \
      int got_reg, d0, d1;                                                                                                \
      asm volatile("v_mov_b32 v91, %3\n\tv_mov_b32 v92, %4\n\tv_mov_b32 v90, %3\n\ts_nop 4\n\t" FILL                     \
                   "ds_read2_b32 v[94:95], v91 offset1:16\n\t" WAIT "v_mov_b32 v91, v92\n\t"                              \
                   "s_waitcnt lgkmcnt(0)\n\ts_nop 7\n\ts_nop 7\n\t"                                                        \
                   "v_mov_b32 %0, v91\n\tv_mov_b32 %1, v94\n\tv_mov_b32 %2, v95"                                          \
                   : "=v"(got_reg), "=v"(d0), "=v"(d1) : "v"(lane_addr), "v"(pattern) : CLOB);                            \
      *bad_reg += got_reg != pattern;                                                                                     \
      *bad_data += (d0 != expect0) + (d1 != expect1);                                                                     \
    }                                                                                                                     \
  }
#define GRID(F) EXPERIMENT(exp_f##F##_w0, FILL_##F, WAIT_0) EXPERIMENT(exp_f##F##_w1, FILL_##F, WAIT_1) \
                EXPERIMENT(exp_f##F##_w2, FILL_##F, WAIT_2) EXPERIMENT(exp_f##F##_w4, FILL_##F, WAIT_4) \
                EXPERIMENT(exp_f##F##_w8, FILL_##F, WAIT_8) EXPERIMENT(exp_f##F##_w16, FILL_##F, WAIT_16)
GRID(0) GRID(4) GRID(16)

typedef int i32x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

// out: [18 experiments][4 lane quarters][2: register, data] ints, zeroed by the caller.  partner = 1: waves 4-7 (the second
// wave of every SIMD) run MFMAs + FMAs at raised priority while waves 0-3 run the experiments; 0: they idle
__global__ __launch_bounds__(512, 1) void lds_war_kernel(int* out, int iters, int partner, int* stop_flag) {
  __shared__ int s_words[16384];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  for (int i = tid; i < 16384; i += 512) s_words[i] = i * 2654435761u;
  __shared__ int s_done;
  if (tid == 0) s_done = 0;
  __syncthreads();
  if (wave >= 4) {
    if (!partner) return;
    i32x8 a, b;
    for (int j = 0; j < 8; ++j) {
      a[j] = (tid * 2654435761u + j * 40503u) & 0xbfbfbfbfu;
      b[j] = (tid * 1664525u + j * 22695477u) & 0xbfbfbfbfu;
    }
    f32x4 acc[4] = {f32x4{0, 0, 0, 0}, f32x4{0, 0, 0, 0}, f32x4{0, 0, 0, 0}, f32x4{0, 0, 0, 0}};
    f32x4 tot[4] = {f32x4{0, 0, 0, 0}, f32x4{0, 0, 0, 0}, f32x4{0, 0, 0, 0}, f32x4{0, 0, 0, 0}};
    __builtin_amdgcn_s_setprio(1);
    while (__hip_atomic_load(&s_done, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) < 4) {
#pragma unroll
      for (int r = 0; r < 8; ++r)
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          acc[i] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(a, b, f32x4{0, 0, 0, 0}, 0, 0, 0, 0, 0, 0);
#pragma unroll
          for (int c = 0; c < 4; ++c) tot[(i + 2) & 3][c] = fmaf(acc[(i + 2) & 3][c], 1.0001f, tot[(i + 2) & 3][c]);
        }
    }
    __builtin_amdgcn_s_setprio(0);
    if (tot[0][0] + tot[1][1] + tot[2][2] + tot[3][3] == 123.456f) out[0] = 1;
    return;
  }
  const int lane_addr = static_cast<int>(reinterpret_cast<uintptr_t>(s_words) & 0xffff) + lane * 4 + wave * 1024;
  const int base = (lane_addr - static_cast<int>(reinterpret_cast<uintptr_t>(s_words) & 0xffff)) >> 2;
  const int e0 = static_cast<int>(base * 2654435761u), e1 = static_cast<int>((base + 16) * 2654435761u);
  const int pattern = 0x3f9d70a4 + lane;
  int col = 0;
  auto run = [&](auto fn) {
    int br = 0, bd = 0;
    fn(iters, lane_addr, pattern, &br, &bd, e0, e1);
    atomicAdd(out + (col * 4 + (lane >> 4)) * 2, br);
    atomicAdd(out + (col * 4 + (lane >> 4)) * 2 + 1, bd);
    ++col;
  };
#define RUN(F) run([](auto... x) { exp_f##F##_w0(x...); }); run([](auto... x) { exp_f##F##_w1(x...); }); \
               run([](auto... x) { exp_f##F##_w2(x...); }); run([](auto... x) { exp_f##F##_w4(x...); }); \
               run([](auto... x) { exp_f##F##_w8(x...); }); run([](auto... x) { exp_f##F##_w16(x...); });
  RUN(0) RUN(4) RUN(16)
  if (lane == 0) __hip_atomic_fetch_add(&s_done, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}

extern "C" int lds_war_launch(void* out, int workgroups, int iters, int partner, void* stream) {
  lds_war_kernel<<<workgroups, 512, 0, static_cast<hipStream_t>(stream)>>>(static_cast<int*>(out), iters, partner, nullptr);
  return hipGetLastError() == hipSuccess ? 0 : -1;
}
