// common.hpp -- shared host-side helpers of the LibTorch operator surface.
//
// Mirrors the role of the reference's src/common.hpp:10-11 (Slc, CUDAFloat) and
// src/common_cuda.hpp; launch geometry lives behind the C ABI (include/f2nerf_hip.h), so nothing of
// it is visible here.  Every kernel call goes through f2n::check(), which turns a non-zero ABI status
// into a c10::Error -- there is no CPU fallback anywhere in this library.
#pragma once

#include <c10/hip/HIPStream.h>
#include <torch/torch.h>

#include <string>

#include "f2nerf_hip.h"

using Slc = torch::indexing::Slice;

namespace f2n
{

using Tensor = torch::Tensor;

// The reference hard-wires kCUDA (src/common.hpp:11); on a ROCm build of LibTorch kCUDA is the HIP
// device.  Host-logic tests construct the modules on CPU, so the device is taken from a module
// option instead, and ops refuse CPU tensors.
inline torch::TensorOptions float_on(const torch::Device & d)
{
  return torch::TensorOptions().dtype(torch::kFloat32).device(d);
}
inline torch::TensorOptions int_on(const torch::Device & d)
{
  return torch::TensorOptions().dtype(torch::kInt32).device(d);
}

inline torch::Device default_device()
{
  return torch::cuda::is_available() ? torch::Device(torch::kCUDA) : torch::Device(torch::kCPU);
}

inline void * current_stream(const Tensor & t)
{
  return (void *)c10::hip::getCurrentHIPStream(t.device().index()).stream();
}

inline void check(int status, const char * what)
{
  TORCH_CHECK(status == F2N_OK, what, " failed: ", f2n_status_string(status), " (", status, ")");
}

// Contiguous f32 / i32 tensor on the GPU, or an error naming the argument.
inline Tensor dev_f32(const Tensor & t, const char * name)
{
  TORCH_CHECK(t.defined(), name, " is undefined");
  TORCH_CHECK(
    t.is_cuda(), name,
    " must live on the GPU: the F2-NeRF hot path has no CPU implementation in this library");
  TORCH_CHECK(t.scalar_type() == torch::kFloat32, name, " must be float32");
  return t.contiguous();
}
inline Tensor dev_i32(const Tensor & t, const char * name)
{
  TORCH_CHECK(t.defined(), name, " is undefined");
  TORCH_CHECK(
    t.is_cuda(), name,
    " must live on the GPU: the F2-NeRF hot path has no CPU implementation in this library");
  TORCH_CHECK(t.scalar_type() == torch::kInt32, name, " must be int32");
  return t.contiguous();
}

inline const float * fptr(const Tensor & t) { return t.defined() ? t.data_ptr<float>() : nullptr; }
inline float * fptr_mut(Tensor & t) { return t.defined() ? t.data_ptr<float>() : nullptr; }
inline const int32_t * iptr(const Tensor & t)
{
  return t.defined() ? t.data_ptr<int32_t>() : nullptr;
}

}  // namespace f2n
