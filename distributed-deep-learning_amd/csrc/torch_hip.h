// PyTorch-ROCm glue shared by the binding files: the current HIP stream and a device guard.
// (On ROCm builds PyTorch exposes its HIP streams/guards under the "MasqueradingAsCUDA" names.)
#pragma once
#include <torch/extension.h>
#include <ATen/hip/impl/HIPGuardImplMasqueradingAsCUDA.h>
#include <ATen/hip/impl/HIPStreamMasqueradingAsCUDA.h>
#include <hip/hip_runtime.h>

namespace ddl {
inline hipStream_t cur_stream() { return c10::hip::getCurrentHIPStreamMasqueradingAsCUDA().stream(); }
using DeviceGuard = c10::hip::HIPGuardMasqueradingAsCUDA;
}  // namespace ddl

#define DDL_CHECK_DEV(x) TORCH_CHECK((x).is_cuda(), #x " must be a GPU tensor")
#define DDL_CHECK_CONTIG(x) TORCH_CHECK((x).is_contiguous(), #x " must be contiguous")
#define DDL_CHECK_F32(x) TORCH_CHECK((x).scalar_type() == at::kFloat, #x " must be float32")
