// Access-pattern probe (development tool): NHD paged-KV streaming, one head per workgroup (tiles
// interleaved over the 4 waves) vs 4 adjacent heads per workgroup (one head per wave, lockstep).
// Each "tile" = 64 token rows x 256 B per head, token stride 2 KB (8 heads), pages of 64 tokens in
// random order.  Reads only; sums to defeat DCE.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
#include <algorithm>
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;

// mode 0: WG = (head h, request b); wave w takes tiles w, w+4, ...
// mode 1: WG = (head group hg of 4 heads, request b); wave w = head hg*4+w, all tiles
template <int MODE, int INFLIGHT, int KSTYLE = 1, int NT = 0>
__global__ __launch_bounds__(256) void k(const char* __restrict__ kv, const int* __restrict__ pages,
                                         int tiles_per_req, int nreq, unsigned* out) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  int h, b, t0, tstep;
  if (MODE == 0) { h = blockIdx.x / nreq; b = blockIdx.x % nreq; t0 = wave; tstep = 4; }
  else { h = (blockIdx.x / nreq) * 4 + wave; b = blockIdx.x % nreq; t0 = 0; tstep = 1; }
  const int* pg = pages + (long)b * tiles_per_req;
  unsigned acc = 0;
  // per tile: K-like 16 instr of 16 rows x 64 B, V-like 16 instr of 4 rows x 256 B (two caches)
  const int kr = lane & 15, kc = lane >> 4;   // K: row, 16-B chunk within 64 B
  const int vr = lane >> 4, vc = lane & 15;   // V: row, chunk within 256 B
  const long cache_bytes = (long)gridDim.y;   // unused
  (void)cache_bytes;
  for (int t = t0; t < tiles_per_req; t += tstep * INFLIGHT) {
    u32x4 r[INFLIGHT][32];
#pragma unroll
    for (int f = 0; f < INFLIGHT; ++f) {
      const int tt = t + f * tstep;
      const long base = tt < tiles_per_req ? (long)pg[tt] * (64 * 2048 * 2) : -1;
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        const int tb = i >> 2, j = i & 3;
        { const char* p = KSTYLE ? kv + base + (long)(tb * 16 + kr) * 2048 + h * 256 + j * 64 + kc * 16 : kv + base + (long)(i * 4 + vr) * 2048 + h * 256 + vc * 16;
        r[f][i] = base < 0 ? u32x4{0,0,0,0} : (NT ? __builtin_nontemporal_load((const u32x4*)p) : *(const u32x4*)p); }
      }
#pragma unroll
      for (int i = 0; i < 16; ++i)
        { const char* p = kv + base + 64 * 2048 + (long)(i * 4 + vr) * 2048 + h * 256 + vc * 16;
        r[f][16 + i] = base < 0 ? u32x4{0,0,0,0} : (NT ? __builtin_nontemporal_load((const u32x4*)p) : *(const u32x4*)p); }
    }
#pragma unroll
    for (int f = 0; f < INFLIGHT; ++f)
#pragma unroll
      for (int i = 0; i < 32; ++i) acc += r[f][i][0] ^ r[f][i][3];
  }
  if (acc == 0x12345678u) out[0] = acc;
}

int main() {
  const int nreq = 64, tiles = 128, heads = 8;
  const int npages = nreq * tiles + 100;
  const long page_bytes = 64L * 2048 * 2;  // K page + V page adjacent
  char* kv; int* pages; unsigned* out;
  hipMalloc(&kv, npages * page_bytes); hipMalloc(&pages, nreq * tiles * 4); hipMalloc(&out, 64);
  hipMemset(kv, 1, npages * page_bytes);
  std::vector<int> perm(npages); for (int i = 0; i < npages; ++i) perm[i] = i;
  srand(1); std::random_shuffle(perm.begin(), perm.end());
  hipMemcpy(pages, perm.data(), nreq * tiles * 4, hipMemcpyHostToDevice);
  const double bytes = (double)nreq * heads * tiles * 64 * 512;
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  auto run = [&](const char* name, auto launch) {
    for (int i = 0; i < 3; ++i) launch();
    hipDeviceSynchronize();
    hipEventRecord(e0);
    for (int i = 0; i < 10; ++i) launch();
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    printf("%-40s %8.1f us  %8.1f GB/s\n", name, ms * 100, bytes / (ms / 10 * 1e-3) / 1e9);
  };
  run("mode0 head/WG tiles over waves, 1 tile", [&] { k<0, 1><<<heads * nreq, 256>>>(kv, pages, tiles, nreq, out); });
  run("mode1 4 heads/WG lockstep, 1 tile", [&] { k<1, 1><<<heads / 4 * nreq, 256>>>(kv, pages, tiles, nreq, out); });
  run("mode1 4 heads/WG lockstep, 2 tiles", [&] { k<1, 2><<<heads / 4 * nreq, 256>>>(kv, pages, tiles, nreq, out); });
  run("mode0 nt", [&] { k<0, 1, 1, 1><<<heads * nreq, 256>>>(kv, pages, tiles, nreq, out); });
  run("mode0 V-style K loads", [&] { k<0, 1, 0, 0><<<heads * nreq, 256>>>(kv, pages, tiles, nreq, out); });
  run("mode0 V-style K loads nt", [&] { k<0, 1, 0, 1><<<heads * nreq, 256>>>(kv, pages, tiles, nreq, out); });
  run("mode0 2 tiles in flight nt", [&] { k<0, 2, 1, 1><<<heads * nreq, 256>>>(kv, pages, tiles, nreq, out); });
  run("mode0 again", [&] { k<0, 1><<<heads * nreq, 256>>>(kv, pages, tiles, nreq, out); });
  return 0;
}
