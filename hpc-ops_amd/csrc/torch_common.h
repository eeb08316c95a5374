// Shared by the C++ host translation units (csrc/torch_*.cpp -> hpc/_hpc_torch.so): stream lookup, raw pointers,
// the launch-failed convention of the reference (TORCH_CHECK(running, "<op> launch failed!"), src/attention/entry.cc:722).
#pragma once
#include <ATen/ATen.h>
#include <c10/hip/HIPStream.h>
#include <torch/library.h>

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

}  // namespace hpc_torch
