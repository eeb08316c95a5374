// Development tuning registers - INTERNAL, not part of the C-ABI of include/hpc_amd.h, and NOT IN THE SHIPPED LIBRARY.
//
// 64 small integers that select kernel variants inside the launchers for A/B measurements and for
// the parity tests that pin one variant (tests marked `dev`, tools/).  They exist only in the DEVELOPMENT build
// (-DHPC_DEV -> hpc/libhpc_amd_dev.so, loaded by `HPC_AMD_DEV=1`): there storage is a set of relaxed atomics and the
// environment variable HPC_AMD_TUNING="key=value,key=value" seeds them once at library load.  In the production
// build (hpc/libhpc_amd.so) hpc_dev_tuning_get is the constant 0: every variant branch folds away at compile
// time, no setter is exported, no environment variable is read - the timing-only variants that give wrong results
// (15, 18) and the all-reduce knobs that must agree on every rank (9, 10, 11) cannot be switched on by accident.
//   key 0  decode KV load cache policy (1 = temporal instead of nt)
//   key 1  streaming grouped GEMM: forced tokens-per-pass / waves variant
//   key 3  grouped GEMM tiled mode (0 auto, 1 never, 2 always 256x128 when possible, 3 always 128x128,
//          4 always 256x256 when possible)
//   key 5  decode: 1 = never run one task per wave ("solo")
//   key 6  256x128 tiled GEMM: 2 = 32-token narrow form
//   key 7  block-sparse prefill row mapping (1 head-major, 2 position-major)
//   key 9  fused all-reduce (high throughput): 1 = runtime-world-size kernel at any world size
//   key 10 fused all-reduce: bounded spins give up after 2^value rounds (default 2^22)
//   key 11 fused all-reduce (high throughput): minimum grid (default 512 = two workgroups per CU)
//   key 12 decode fp8: 1 = never the head-pair kernel (attention_decode_v2.hip)
//   key 14 decode fp8 v2: workgroup count override
//   key 15 decode fp8 v2: 1 = no KV loads (compute-only timing)
//   key 28 decode bf16: 1 = never the head-pair kernel (attention_decode_v2.hip, bf16 form)
//   key 29 decode fp8: 2 = the four-head form of the head-pair kernel (<= 8 q rows per kv head, kv heads % 4 == 0)
//   key 30 / 31 decode fp8, 8 kv heads: extra workgroups (value - 100 per 128) for the head pair at byte 256 / 768 of a token row
//   key 32 decode fp8: ranges of the first half of the grid in percent of the others' (0 = equal)
//   key 18 256x256 grouped GEMM: 1 = no DMA in the k-loop (timing only - wrong results)
//   key 19 fused MoE: 1 = the activation as a separate kernel instead of the gate-up GEMM's epilogue
//   key 20 decode v2: minimum cost of a range of the in-kernel plan (default 8)
//   key 21 256x256 grouped GEMM: 1 = never the half-tile / tail body for a group's last token tile, 2 = no tail body (<= 64 rows)
//   key 22 256x256 grouped GEMM, blockwise: variant of the k-loop (group_gemm_p8.hip: 1 section profile, 2 the round-4 loop
//          (tails behind the barrier), 3 its profile, 4 no s_setprio)
//   key 26 256x256 grouped GEMM: 1 = the register-streamed tail body (weights global -> MFMA operand registers, six stages)
//          instead of the LDS-ring one (measured: no faster - a tail tile costs its bytes at the per-CU bandwidth)
//   key 25 grouped GEMM kernel choice: 1 = the 256 x 256 kernel only from 192 rows per group on (rounds 2-4)
//   key 24 256x256 grouped GEMM, tail body: 1 = default cache policy instead of non-temporal weight loads for single-tile groups
//   key 23 256x256 grouped GEMM: 1 = all full tiles first, tail tiles last (measured slower than tails in place)
//   key 33 decode, first generation: 1 = split requests merged by decode_combine_kernel (second launch) instead of the last arriver
//   key 34 decode scheduler: bin count override (<= 4 per CU)
//   key 35 low-latency all-reduce: 1 = always two launches, 2 = one launch whenever the grid is resident at once
//          (default: one launch up to one workgroup per CU)
//   key 36 decode v2: head pair of the SECOND workgroup of every CU = pair ^ (value - 1) (0 = the product's rule: the slice across
//          byte-address bit 9; 1 = off, both workgroups of a CU on the same slice)
//   key 37 decode v2: value = s + 1: every workgroup streams slice s of the token rows (timing only - wrong results)
//   key 40 router GEMM: 1 = the 64 x 64 kernel at every m > 256 (rounds 1-4) instead of the LDS-staged tile kernel (which serves every m > 256)
//   key 41 router GEMM tile kernel: bit 0 = no weight loads, bit 1 = no activation loads (timing only - wrong results)
//   key 45 router GEMM: cap on the split count above m = 256 (default 8, clamped to the 16 planes the reduce sums)
//   key 46 bf16 prefill: 1 = V^T operands built with v_perm_b32 (rounds 3-4) instead of the transposing LDS read
//   key 47 fused all-reduce (high throughput): 1 = signal barriers of rounds 1-5 (compare-and-swap loops, system-scope release / acquire fences) instead of
//          poll + store on relaxed accesses behind vmcnt(0)
//   key 48 fused all-reduce (high throughput): 1 = signal flags packed at the start of the pad (rounds 1-5) instead of spread over it
//   key 49 grouped GEMM 256 x 256 kernel: 1 = no ride-along rows (every tail of a group runs as its own tail / half-tile item: rounds 2-5)
//   key 52 low-latency all-reduce, loopback harness: 1 = the one-shot form; key 53: 1 = the all-reduce alone (no residual / norm)
//   key 54 fp8 decode, quant_type 0 (per-token K scales): 1 = the first-generation kernel (rounds 1-5) instead of the head-pair kernel
//   key 55 fp8 decode on HND pages (per-tensor scales): 1 = the head-pair kernel's HND form (1 KB pieces of one head per load) instead of the first-generation kernel
//   key 56 streaming grouped GEMM (16 / 32 tokens per pass): 1 = the stage loop of rounds 1-5 (refill at the end of a stage, scalar scale loads
//          waited for on the spot) instead of the re-ordered one (gemm_blockwise_stream2_kernel); 2 = the re-ordered loop on chains of four
//          K = 32 MFMAs (bit-identical to 1) instead of one K = 128 MFMA per k-block (bit-identical to the 256 x 256 kernel)
//   key 58 decode v2: 1 = the last arriver of a split request loads its first chunk again when the trip's second chunk does not exist (rounds 3-5)
//   key 60 fp8 decode, one kv head per workgroup with <= 32 q rows (attention_decode_v2.hip, kSolo): 0 = for 17 ... 32 q rows per kv head,
//          1 = never (the first generation's two-block form), 2 = also for <= 16 q rows on HND pages / with odd kv-head counts,
//          3 = every eligible call (A/B against the pair forms); bf16 calls follow the same key
//   key 61 decode v2: 1 = range boundaries of an underloaded launch are not moved to the ends of short requests (rounds 2-5)
//   key 43 256x256 grouped GEMM: 1 = the work-item lookup's wave scans and lane reads through ds_bpermute (rounds 2-5) instead of DPP adds + v_readlane
//   others: see the launchers that read them
#pragma once

#ifdef HPC_DEV
constexpr bool kHpcDevBuild = true;
extern "C" int hpc_dev_tuning_set(int key, int value);
extern "C" int hpc_dev_tuning_get(int key);
#else
constexpr bool kHpcDevBuild = false;  // launch sites of development-only instantiations test it: they are not emitted
static constexpr int hpc_dev_tuning_get(int) { return 0; }
#endif
