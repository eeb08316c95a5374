// Hazard probe (development tool, tools/probe_mfma_war.py): does a VALU write to a SOURCE register of a
// v_mfma_f32_16x16x128_f8f6f4 that was issued just before it corrupt that MFMA's operand?
//
// Found in round 6 (csrc/group_gemm_p8.hip, ride-along body built without -fno-slp-vectorize): hipcc placed `v_mov_b32 v10, ...`
// directly behind `v_mfma_f32_16x16x128_f8f6f4 v[190:193], v[34:41], v[10:17], 0` (v[10:17] dead for the compiler after that
// MFMA) and the block that MFMA computes came out wrong, nondeterministically, in a loop that keeps the matrix pipe saturated.
// hipcc (ROCm 7.2) inserts no wait state there, and the same two instructions in a kernel whose matrix pipe is mostly idle
// (the attention kernels have dozens) never showed a wrong result.  This probe isolates the pair:
//   [one MFMA of backlog] [G unrelated VALU instructions] [the MFMA under test: acc = A x B] [D wait states] [v_mov into
//   register K of A or B] [long pause] [restore the register]
// and compares acc with the same sequence without the write.  One wave per SIMD or two (the GEMM's occupancy).
#include <hip/hip_runtime.h>
#include <stdint.h>

// fixed registers: A = v[64:71], B = v[72:79], junk = v80, saved = v81, backlog accumulators v[84:99], result v[100:103]
#define CLOB "v64", "v65", "v66", "v67", "v68", "v69", "v70", "v71", "v72", "v73", "v74", "v75", "v76", "v77", "v78", "v79", \
             "v80", "v81", "v82", "v83", "v84", "v85", "v86", "v87", "v88", "v89", "v90", "v91", "v92", "v93", "v94", "v95", "v96", "v97", \
             "v98", "v99", "v100", "v101", "v102", "v103"

#define BACKLOG_0 ""
#define BACKLOG_1 "v_mfma_f32_16x16x128_f8f6f4 v[84:87], v[64:71], v[72:79], 0\n\t"
#define BACKLOG_2 BACKLOG_1 "v_mfma_f32_16x16x128_f8f6f4 v[88:91], v[64:71], v[72:79], 0\n\t"
#define BACKLOG_4 BACKLOG_2 "v_mfma_f32_16x16x128_f8f6f4 v[92:95], v[64:71], v[72:79], 0\n\t" \
                            "v_mfma_f32_16x16x128_f8f6f4 v[96:99], v[64:71], v[72:79], 0\n\t"
// G unrelated VALU instructions between the backlog MFMA and the MFMA under test (the GEMM loop has 4-6: the rescale FMAs):
// where in the previous MFMA's execution the one under test is issued
#define GAP_0 ""
#define GAP_2 "v_fma_f32 v82, v82, v82, v82\n\tv_fma_f32 v83, v83, v83, v83\n\t"
#define GAP_4 GAP_2 GAP_2
#define GAP_6 GAP_4 GAP_2
#define GAP_8 GAP_4 GAP_4
#define GAP_10 GAP_8 GAP_2
#define WAIT_0 ""
#define WAIT_1 "s_nop 0\n\t"
#define WAIT_2 "s_nop 1\n\t"
#define WAIT_3 "s_nop 2\n\t"
#define WAIT_4 "s_nop 3\n\t"
#define WAIT_6 "s_nop 5\n\t"
#define WAIT_8 "s_nop 7\n\t"
#define WAIT_12 "s_nop 7\n\ts_nop 3\n\t"
#define WAIT_16 "s_nop 7\n\ts_nop 7\n\t"
#define PAUSE "s_nop 15\n\ts_nop 15\n\ts_nop 15\n\ts_nop 15\n\ts_nop 15\n\ts_nop 15\n\t"

// one experiment: returns the number of result registers (of this lane) that differ from the clean run, summed over `iters`
#define EXPERIMENT(NAME, BACKLOG, WAIT, REG)                                                                        \
  __device__ __forceinline__ int NAME(int iters, bool with_write) {                                                  \
    int bad = 0;                                                                                                     \
    for (int it = 0; it < iters; ++it) {                                                                             \
      float r0, r1, r2, r3, c0, c1, c2, c3;                                                                          \
      /* clean */                                                                                                    \
      asm volatile(PAUSE BACKLOG "v_mfma_f32_16x16x128_f8f6f4 v[100:103], v[64:71], v[72:79], 0\n\t" PAUSE          \
                   "v_mov_b32 %0, v100\n\tv_mov_b32 %1, v101\n\tv_mov_b32 %2, v102\n\tv_mov_b32 %3, v103"           \
                   : "=v"(c0), "=v"(c1), "=v"(c2), "=v"(c3)::CLOB);                                                  \
      if (with_write)                                                                                                \
        asm volatile("v_mov_b32 v81, " REG "\n\t" PAUSE BACKLOG                                                      \
                     "v_mfma_f32_16x16x128_f8f6f4 v[100:103], v[64:71], v[72:79], 0\n\t" WAIT                        \
                     "v_mov_b32 " REG ", v80\n\t" PAUSE "v_mov_b32 " REG ", v81\n\t"                                 \
                     "v_mov_b32 %0, v100\n\tv_mov_b32 %1, v101\n\tv_mov_b32 %2, v102\n\tv_mov_b32 %3, v103"         \
                     : "=v"(r0), "=v"(r1), "=v"(r2), "=v"(r3)::CLOB);                                                \
      else                                                                                                           \
        asm volatile(PAUSE BACKLOG "v_mfma_f32_16x16x128_f8f6f4 v[100:103], v[64:71], v[72:79], 0\n\t" WAIT PAUSE   \
                     "v_mov_b32 %0, v100\n\tv_mov_b32 %1, v101\n\tv_mov_b32 %2, v102\n\tv_mov_b32 %3, v103"         \
                     : "=v"(r0), "=v"(r1), "=v"(r2), "=v"(r3)::CLOB);                                                \
      bad += (__float_as_int(r0) != __float_as_int(c0)) + (__float_as_int(r1) != __float_as_int(c1)) +              \
             (__float_as_int(r2) != __float_as_int(c2)) + (__float_as_int(r3) != __float_as_int(c3));               \
    }                                                                                                                \
    return bad;                                                                                                      \
  }

// gap x wait x register, one MFMA of backlog in front of the gap: B registers v72 (first), v73; A register v64
#define GRID_REG(G, W, TAG, REG)  EXPERIMENT(exp_g##G##_w##W##_##TAG, BACKLOG_1 GAP_##G, WAIT_##W, REG)
#define GRID_W(G, W) GRID_REG(G, W, b0, "v72") GRID_REG(G, W, b1, "v73") GRID_REG(G, W, a0, "v64")
#define GRID_G(G) GRID_W(G, 0) GRID_W(G, 1) GRID_W(G, 2) GRID_W(G, 4)
GRID_G(0) GRID_G(2) GRID_G(4) GRID_G(6) GRID_G(8) GRID_G(10)

__global__ __launch_bounds__(512) void war_probe_kernel(int* out, int iters, int with_write) {
  // operands: pseudo-random e4m3 bytes (|x| < 2, no NaN); junk: a float bit pattern like a scale
  uint32_t s = (blockIdx.x * 512u + threadIdx.x) * 2654435761u + 12345u;
  auto rnd = [&]() {
    s = s * 1664525u + 1013904223u;
    return static_cast<int>((s ^ (s >> 13)) & 0xbfbfbfbfu);
  };
  int v[16];
  for (int i = 0; i < 16; ++i) v[i] = rnd();
  asm volatile("v_mov_b32 v64, %0\n\tv_mov_b32 v65, %1\n\tv_mov_b32 v66, %2\n\tv_mov_b32 v67, %3\n\t"
               "v_mov_b32 v68, %4\n\tv_mov_b32 v69, %5\n\tv_mov_b32 v70, %6\n\tv_mov_b32 v71, %7"
               ::"v"(v[0]), "v"(v[1]), "v"(v[2]), "v"(v[3]), "v"(v[4]), "v"(v[5]), "v"(v[6]), "v"(v[7]) : CLOB);
  asm volatile("v_mov_b32 v72, %0\n\tv_mov_b32 v73, %1\n\tv_mov_b32 v74, %2\n\tv_mov_b32 v75, %3\n\t"
               "v_mov_b32 v76, %4\n\tv_mov_b32 v77, %5\n\tv_mov_b32 v78, %6\n\tv_mov_b32 v79, %7\n\tv_mov_b32 v80, 0x3f9d70a4"
               ::"v"(v[8]), "v"(v[9]), "v"(v[10]), "v"(v[11]), "v"(v[12]), "v"(v[13]), "v"(v[14]), "v"(v[15]) : CLOB);
  const bool ww = with_write != 0;
  int col = 0;
  auto put = [&](int bad) {
    atomicAdd(out + col, bad);
    ++col;
  };
#define RUN_REG(G, W, TAG) put(exp_g##G##_w##W##_##TAG(iters, ww));
#define RUN_W(G, W) RUN_REG(G, W, b0) RUN_REG(G, W, b1) RUN_REG(G, W, a0)
#define RUN_G(G) RUN_W(G, 0) RUN_W(G, 1) RUN_W(G, 2) RUN_W(G, 4)
  RUN_G(0) RUN_G(2) RUN_G(4) RUN_G(6) RUN_G(8) RUN_G(10)
}

// out: 6 gaps x 4 waits x 3 registers ints, zeroed by the caller
extern "C" int war_probe_launch(void* out, int workgroups, int threads, int iters, int with_write, void* stream) {
  war_probe_kernel<<<workgroups, threads, 0, static_cast<hipStream_t>(stream)>>>(static_cast<int*>(out), iters, with_write);
  return hipGetLastError() == hipSuccess ? 0 : -1;
}
