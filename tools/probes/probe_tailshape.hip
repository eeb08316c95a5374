// Shape probe (development tool): what does ONE workgroup per CU cost a stream of 1-3 MB items?  The weights of the MoE's two GEMMs
// ([E][N][K] fp8) are cut into items of 256 rows x K (the 256 x 256 kernel's tail items); a workgroup takes one item and exits.
// Every wave owns rows of the item and walks K in 128-byte k-tiles through a private LDS ring filled by LDS-DMA loads
// (buffer_load ... lds, 1 KB per instruction), waiting for k-tile T with the next D - 1 in flight; nothing else happens.
//   A: 8 waves x 32 rows, ring of 3 (2 k-tiles = 8 KB per wave in flight), 1 workgroup per CU   - the tail body's shape
//   B: 8 waves x 32 rows, ring of 4 (3 in flight), 1 workgroup per CU                            - the narrow tail body's
//   C: 8 waves x 32 rows, ring of 2 (1 in flight), 2 workgroups per CU (64 KB of LDS each)
//   D: 4 waves x 32 rows (items of 128 rows), ring of 3, 2 workgroups per CU x ... 48 KB each -> 3 per CU
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef __attribute__((address_space(3))) void lds_void;

template <int WAVES, int RING, int WGS>
__global__ __launch_bounds__(64 * WAVES, WGS) void k(const char* __restrict__ w, long rows, int K, unsigned* out) {
  __shared__ __attribute__((aligned(1024))) unsigned char s_ring[WAVES][RING][4096];
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const long row0 = ((long)blockIdx.x * WAVES + wave) * 32;
  if (row0 >= rows) return;
  const int KB = K / 128;
  const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)(w + row0 * K), 0, 32u * (unsigned)K, 0x00020000);
  const unsigned voff = (unsigned)(lane >> 3) * (unsigned)K + (lane & 7) * 16;
  auto dma = [&](int T) {
    const int st = T % RING;
#pragma unroll
    for (int q = 0; q < 4; ++q)
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lds_void*)&s_ring[wave][st][q * 1024], 16, voff + q * 8 * K, T * 128, 0, 2);
  };
  unsigned acc = 0;
  for (int T = 0; T < RING - 1 && T < KB; ++T) dma(T);
  for (int T = 0; T < KB; ++T) {
    if (T + RING - 1 < KB) {
      dma(T + RING - 1);
      if constexpr (RING == 2) __builtin_amdgcn_s_waitcnt(0x0F70 | 4);
      if constexpr (RING == 3) __builtin_amdgcn_s_waitcnt(0x0F70 | 8);
      if constexpr (RING == 4) __builtin_amdgcn_s_waitcnt(0x0F70 | 12);
    } else {
      __builtin_amdgcn_s_waitcnt(0x0F70);
    }
    acc += *(const unsigned*)&s_ring[wave][T % RING][lane * 64];
  }
  if (acc == 0x12345678u) out[0] = acc;
}

int main() {
  char* w; unsigned* out;
  const long bytes_max = 64L * 22016 * 4096;
  hipMalloc(&w, bytes_max); hipMalloc(&out, 64);
  hipMemset(w, 1, bytes_max);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  auto run = [&](const char* name, double bytes, auto launch) {
    for (int i = 0; i < 2; ++i) launch();
    hipDeviceSynchronize();
    float best = 1e9;
    for (int rep = 0; rep < 3; ++rep) {
      hipEventRecord(e0);
      for (int i = 0; i < 4; ++i) launch();
      hipEventRecord(e1); hipEventSynchronize(e1);
      float ms; hipEventElapsedTime(&ms, e0, e1);
      best = ms < best ? ms : best;
    }
    printf("%-64s %8.1f us  %8.1f GB/s  %.3f of 8 TB/s\n", name, best * 250, bytes / (best / 4 * 1e-3) / 1e9, bytes / (best / 4 * 1e-3) / 8e12);
  };
  struct { const char* nm; long rows; int K; } shapes[2] = {{"gate-up 64 x 22016 x 4096", 64L * 22016, 4096}, {"down 64 x 4096 x 11008", 64L * 4096, 11008}};
  for (auto& s : shapes) {
    const double bytes = (double)s.rows * s.K;
    char nm[160];
#define RUN(WAVES, RING, WGS, what) snprintf(nm, sizeof nm, "%s: %s", s.nm, what); \
    run(nm, bytes, [&] { k<WAVES, RING, WGS><<<(int)(s.rows / (32 * WAVES)), 64 * WAVES>>>(w, s.rows, s.K, out); });
    RUN(8, 3, 1, "A 8 waves, ring 3, 1 WG/CU (tail body)")
    RUN(8, 4, 1, "B 8 waves, ring 4, 1 WG/CU (narrow)")
    RUN(8, 2, 2, "C 8 waves, ring 2, 2 WG/CU")
    RUN(4, 3, 3, "D 4 waves, ring 3, 3 WG/CU")
    RUN(4, 4, 2, "E 4 waves, ring 4, 2 WG/CU")
    RUN(4, 5, 2, "F 4 waves, ring 5, 2 WG/CU")
    RUN(8, 3, 1, "A again")
  }
  return 0;
}
