// Shared by the C++ host translation units (csrc/torch_*.cpp -> hpc/_hpc_torch.so): stream lookup, raw pointers,
// the launch-failed convention of the reference (TORCH_CHECK(running, "<op> launch failed!"), src/attention/entry.cc:722).
#pragma once
#include <ATen/ATen.h>
#include <c10/hip/HIPStream.h>
#include <torch/library.h>

#include <map>
#include <mutex>
#include <string>
#include <tuple>
#include <vector>

#include "hpc_amd.h"

namespace hpc_torch {

inline hpc_stream_t stream_of(const at::Tensor& t) {
  // reference: at::cuda::getCurrentCUDAStream(tensor.device) - the current stream of the tensor's device
  return reinterpret_cast<hpc_stream_t>(c10::hip::getCurrentHIPStream(t.device().index()).stream());
}

inline const char* err_text(int code) {
  switch (code) {
    case -1: return "unsupported configuration";
    case -2: return "invalid argument";
    case -3: return "HIP launch error";
    case -4: return "an earlier fused all-reduce timed out waiting for a peer";
    default: return "error";
  }
}
#define HPC_LAUNCH_CHECK(rc, what) TORCH_CHECK((rc) == 0, what, " launch failed! (", ::hpc_torch::err_text(rc), ")")

inline void* ptr(const at::Tensor& t) { return t.data_ptr(); }
inline void* ptr(const c10::optional<at::Tensor>& t) { return t.has_value() ? t->data_ptr() : nullptr; }

inline void cuda_contig(const at::Tensor& t, const char* name) {
  TORCH_CHECK(t.is_cuda(), name, " tensor must be cuda");
  TORCH_CHECK(t.is_contiguous(), name, " tensor must be contiguous");
}
inline int i32(int64_t v) { return static_cast<int>(v); }

// ---- zero-once scratch --------------------------------------------------------------------------------------------
// Some kernels keep arrival counters in caller scratch that must be ZERO the first time a buffer is used and that
// every call leaves zero again (decode: the counters of split requests; router GEMM: its split-K tickets).  Instead of
// an allocation + a zero-fill launch per call, such scratch is one buffer per (purpose, device, stream, hipGraph
// capture), allocated once with its first `zero_bytes` bytes zeroed.  Calls on one stream are ordered, so sharing the
// buffer between them is safe; different streams get different buffers.  A buffer first used while a hipGraph is being
// captured has its zero-fill only RECORDED (a node of that graph): the key carries the capture id, so such a buffer
// serves the calls of that capture and nothing else, the entries of earlier (ended) captures on the stream are dropped on
// the next call (their memory stays with their graph's pool), and a buffer cached by eager calls is not used inside a capture - every graph
// owns its buffer and its zero node.
enum ScratchPurpose { kScratchDecode = 0, kScratchGemmFlags = 1 };
using ScratchKey = std::tuple<int, int, void*, long long>;
inline std::mutex& scratch_mutex() {
  static std::mutex mu;
  return mu;
}
inline std::map<ScratchKey, at::Tensor>& scratch_cache() {
  // never destroyed: tensors must not be released during static destruction, after the allocator is gone
  static auto& cache = *new std::map<ScratchKey, at::Tensor>();
  return cache;
}
inline at::Tensor cached_scratch(int purpose, const at::Tensor& like, int64_t nbytes, int64_t zero_bytes) {
  const auto hip_stream = stream_of(like);
  void* stream = static_cast<void*>(hip_stream);
  const long long cap = hpc_stream_capture_id(hip_stream);
  TORCH_CHECK(cap >= 0, "hipStreamGetCaptureInfo failed");
  const int dev = static_cast<int>(like.device().index());
  const ScratchKey key{purpose, dev, stream, cap};
  std::lock_guard<std::mutex> lock(scratch_mutex());
  auto& cache = scratch_cache();
  // Entries of captures that have ENDED on this stream are dropped on every call, not only on a miss: a
  // stream records one capture at a time, so every entry of this (purpose, device, stream) whose capture id is neither
  // 0 (eager) nor the current one belongs to a finished capture - its tensor lives in that graph's private pool and the
  // cache must not keep the pool alive after the graph is destroyed.
  for (auto o = cache.begin(); o != cache.end();)
    o = (std::get<0>(o->first) == purpose && std::get<1>(o->first) == dev && std::get<2>(o->first) == stream &&
         std::get<3>(o->first) != 0 && std::get<3>(o->first) != cap)
            ? cache.erase(o)
            : std::next(o);
  auto it = cache.find(key);
  if (it == cache.end() || it->second.numel() < nbytes) {
    at::Tensor ws = at::empty({nbytes}, like.options().dtype(at::kByte));
    if (zero_bytes > 0) ws.narrow(0, 0, std::min(zero_bytes, nbytes)).zero_();
    it = cache.insert_or_assign(key, ws).first;
  }
  return it->second;
}
// Drops EVERY cached scratch buffer of every purpose: the decode workspaces AND the router GEMM's split-K ticket
// buffers (hpc.release_decode_workspaces documents both).
inline void release_cached_scratch() {
  std::lock_guard<std::mutex> lock(scratch_mutex());
  scratch_cache().clear();
}
inline std::vector<at::Tensor> list_cached_scratch(int purpose) {
  std::lock_guard<std::mutex> lock(scratch_mutex());
  std::vector<at::Tensor> out;
  for (auto& kv : scratch_cache())
    if (std::get<0>(kv.first) == purpose) out.push_back(kv.second);
  return out;
}

}  // namespace hpc_torch
