// Library identity + device queries.
// Replaces reference src/C/version.cc, src/C/built_json.cu:17-43 and get_sm_count()
// (src/utils/utils.cc:15-41; the reference caches device 0 only, this one is per device).
#include <hip/hip_runtime.h>
#include <hip/hip_version.h>

#include <mutex>
#include <sstream>
#include <string>

#include "../../include/hpc_amd.h"

#ifndef HPC_VERSION_STR
#define HPC_VERSION_STR "unknown"
#endif
#ifndef HPC_GIT_HASH_STR
#define HPC_GIT_HASH_STR "unknown"
#endif

extern "C" const char* hpc_version(void) { return HPC_VERSION_STR; }

extern "C" const char* hpc_built_json(void) {
  static std::string json = [] {
    std::ostringstream oss;
    oss << "{\n";
    oss << " \"version\": \"" << HPC_VERSION_STR << "\",\n";
    oss << " \"git-hash\": \"" << HPC_GIT_HASH_STR << "\",\n";
    oss << " \"hipcc\": \"hip-" << HIP_VERSION_MAJOR << "." << HIP_VERSION_MINOR << "."
        << HIP_VERSION_PATCH << "\",\n";
    oss << " \"clang\": \"" << __clang_major__ << "." << __clang_minor__ << "."
        << __clang_patchlevel__ << "\",\n";
    oss << " \"offload-arch\": \"gfx950\",\n";
    oss << " \"stdc++\": \"" << __cplusplus << "\",\n";
    oss << " \"built-date\": \"" << __DATE__ << "\",\n";
    oss << " \"built-time\": \"" << __TIME__ << "\",\n";
    oss << " \"_C\": \"" << __FILE__ << "\"\n";
    oss << "}\n";
    return oss.str();
  }();
  return json.c_str();
}

extern "C" int hpc_get_cu_count(int device_id) {
  static std::mutex mu;
  static int cache[64];
  std::lock_guard<std::mutex> lock(mu);
  if (device_id < 0) {
    if (hipGetDevice(&device_id) != hipSuccess) return -1;
  }
  if (device_id >= 64) return -1;
  if (cache[device_id] == 0) {
    int n = 0;
    if (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, device_id) != hipSuccess)
      return -1;
    cache[device_id] = n;
  }
  return cache[device_id];
}

// 0 when `stream` is not being captured into a hipGraph, else the (non-zero) unique id of the capture; < 0 on error.
// The hosts key their per-stream decode scratch on it: a buffer first used inside a capture has its zero-fill recorded
// as a node of THAT graph only, so it must not be reused by a later capture or by eager calls.
extern "C" long long hpc_stream_capture_id(hipStream_t stream) {
  hipStreamCaptureStatus st = hipStreamCaptureStatusNone;
  unsigned long long id = 0;
  if (hipStreamGetCaptureInfo(stream, &st, &id) != hipSuccess) return -1;
  if (st != hipStreamCaptureStatusActive) return 0;
  return static_cast<long long>(id ? id : 1);
}

#ifdef HPC_DEV
// Development tuning registers: see csrc/hpc_dev.h (internal; not in include/hpc_amd.h).
#include <atomic>
#include <cstdio>
#include <cstdlib>
#include <cstring>

#include "hpc_dev.h"

namespace {
struct Tuning {
  static constexpr int kKeys = 64;
  std::atomic<int> v[kKeys];
  Tuning() {
    for (auto& x : v) x.store(0, std::memory_order_relaxed);
    bool seeded = false;
    const char* env = std::getenv("HPC_AMD_TUNING");  // "key=value,key=value"
    while (env && *env) {
      char* end = nullptr;
      const long k = std::strtol(env, &end, 10);
      if (end == env || *end != '=') break;
      env = end + 1;
      const long val = std::strtol(env, &end, 10);
      if (end == env) break;
      if (k >= 0 && k < kKeys) {
        v[k].store(static_cast<int>(val), std::memory_order_relaxed);
        seeded = seeded || val != 0;
      }
      env = (*end == ',') ? end + 1 : end;
      if (*end != ',') break;
    }
    // Not silent: some registers select timing-only kernel variants whose RESULTS ARE WRONG (15, 18) and the
    // all-reduce ones (9, 10, 11) must agree on every rank.
    if (seeded)
      std::fprintf(stderr, "[hpc_amd] WARNING: development tuning registers set from HPC_AMD_TUNING=\"%s\" - kernel variants "
                           "for A/B measurements are active; some give wrong results by design, the all-reduce ones must "
                           "match on every rank.  Unset the variable in production.\n", std::getenv("HPC_AMD_TUNING"));
  }
};
Tuning& tuning() {
  static Tuning t;
  return t;
}
}  // namespace

extern "C" int hpc_dev_tuning_set(int key, int value) {
  if (key < 0 || key >= Tuning::kKeys) return -2;
  tuning().v[key].store(value, std::memory_order_relaxed);
  return 0;
}
extern "C" int hpc_dev_tuning_get(int key) {
  return (key < 0 || key >= Tuning::kKeys) ? 0 : tuning().v[key].load(std::memory_order_relaxed);
}
#endif  // HPC_DEV
