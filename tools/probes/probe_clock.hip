// Clock probe (development tool, tools/moe_clock.py): eight single-wave workgroups - one per XCD, 16 registers, no LDS, so they
// fit BESIDE a resident one-workgroup-per-CU kernel - sample (s_memtime, s_memrealtime) every `period` reference ticks for
// `samples` rounds.  s_memtime counts shader-core clocks, s_memrealtime the constant reference clock (100 MHz; the host
// calibrates it against HIP events), so the ratio of their increments is the core clock the XCD ran at in that window.
#include <hip/hip_runtime.h>
#include <stdint.h>

__global__ __launch_bounds__(64) void clock_probe_kernel(uint64_t* out, int samples, int period) {
  if (threadIdx.x != 0) return;
  uint64_t* o = out + static_cast<long>(blockIdx.x) * samples * 2;
  uint64_t next = __builtin_amdgcn_s_memrealtime();
  for (int i = 0; i < samples; ++i) {
    uint64_t r;
    do {
      __builtin_amdgcn_s_sleep(8);
      r = __builtin_amdgcn_s_memrealtime();
    } while (r < next);
    o[2 * i] = __builtin_amdgcn_s_memtime();
    o[2 * i + 1] = r;
    next = r + period;
  }
}

extern "C" int clock_probe_launch(void* out, int workgroups, int samples, int period, void* stream) {
  clock_probe_kernel<<<workgroups, 64, 0, static_cast<hipStream_t>(stream)>>>(static_cast<uint64_t*>(out), samples, period);
  return hipGetLastError() == hipSuccess ? 0 : -1;
}
