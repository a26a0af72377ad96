// pybind11 / ATen face of the gfx950 kernels: shape checks on the host (a faulting kernel can
// reset the whole node), then a launch on PyTorch's current HIP stream.  No allocation happens
// inside the launchers themselves; outputs are allocated here through the caching allocator,
// which is graph-capture safe.
#include "torch_hip.h"
#include "ddl_kernels.h"

namespace {

using ddl::cur_stream;
#define CHECK_DEV DDL_CHECK_DEV
#define CHECK_CONTIG DDL_CHECK_CONTIG
#define CHECK_F32 DDL_CHECK_F32

at::Tensor normalize_u8(const at::Tensor& x, bool nhwc) {
  CHECK_DEV(x); CHECK_CONTIG(x);
  TORCH_CHECK(x.scalar_type() == at::kByte && x.dim() == 4, "normalize_u8: uint8 [B,C,H,W] expected");
  ddl::DeviceGuard guard(x.device());
  const int B = x.size(0), C = x.size(1), H = x.size(2), W = x.size(3);
  auto out = nhwc ? at::empty({B, H, W, C}, x.options().dtype(at::kFloat))
                  : at::empty({B, C, H, W}, x.options().dtype(at::kFloat));
  ddl::launch_normalize_u8(x.data_ptr<uint8_t>(), out.data_ptr<float>(), B, C, H, W, nhwc, cur_stream());
  return out;
}

std::vector<at::Tensor> ce_fwd_bwd(const at::Tensor& logits, const at::Tensor& targets) {
  CHECK_DEV(logits); CHECK_CONTIG(logits); CHECK_F32(logits);
  CHECK_DEV(targets); CHECK_CONTIG(targets);
  TORCH_CHECK(logits.dim() == 2 && targets.dim() == 1 && targets.size(0) == logits.size(0),
              "ce_fwd_bwd: logits [B,C], targets [B]");
  TORCH_CHECK(targets.scalar_type() == at::kLong, "targets must be int64");
  ddl::DeviceGuard guard(logits.device());
  const int B = logits.size(0), C = logits.size(1);
  auto loss = at::empty({}, logits.options());
  auto dlogits = at::empty_like(logits);
  auto preds = at::empty({B}, targets.options());
  ddl::launch_ce_fwd_bwd(logits.data_ptr<float>(), targets.data_ptr<int64_t>(), loss.data_ptr<float>(),
                         dlogits.data_ptr<float>(), preds.data_ptr<int64_t>(), B, C, cur_stream());
  return {loss, dlogits, preds};
}

// dst_a <- src_a, dst_b <- src_b in one launch (the captured step's static inputs); contiguous same-shape pairs,
// the first one a multiple of 16 bytes at 16-byte aligned addresses
void copy_pair(at::Tensor dst_a, const at::Tensor& src_a, at::Tensor dst_b, const at::Tensor& src_b) {
  for (const at::Tensor* t : std::initializer_list<const at::Tensor*>{&dst_a, &src_a, &dst_b, &src_b}) {
    CHECK_DEV(*t); CHECK_CONTIG(*t);
  }
  TORCH_CHECK(dst_a.scalar_type() == src_a.scalar_type() && dst_a.sizes() == src_a.sizes() &&
                  dst_b.scalar_type() == src_b.scalar_type() && dst_b.sizes() == src_b.sizes(), "copy_pair: shape / dtype");
  TORCH_CHECK(src_a.device() == dst_a.device() && src_b.device() == dst_a.device() && dst_b.device() == dst_a.device(),
              "copy_pair: all four tensors must live on one device (the launch runs under dst_a's device guard)");
  const int64_t na = dst_a.numel() * dst_a.element_size(), nb = dst_b.numel() * dst_b.element_size();
  TORCH_CHECK(na % 16 == 0 && reinterpret_cast<uintptr_t>(dst_a.data_ptr()) % 16 == 0 &&
                  reinterpret_cast<uintptr_t>(src_a.data_ptr()) % 16 == 0, "copy_pair: 16-byte granularity for the first pair");
  ddl::DeviceGuard guard(dst_a.device());
  ddl::launch_copy_pair(src_a.data_ptr(), dst_a.data_ptr(), na, src_b.data_ptr(), dst_b.data_ptr(), nb, cur_stream());
}

void adam_step(at::Tensor p, at::Tensor g, at::Tensor m, at::Tensor v, at::Tensor step_t, at::Tensor aux,
               double lr, double b1, double b2, double eps, double wd, bool zero_grad,
               std::vector<int64_t> starts, std::vector<int64_t> ends) {
  for (const at::Tensor* t : std::initializer_list<const at::Tensor*>{&p, &g, &m, &v, &step_t, &aux}) {
    CHECK_DEV(*t); CHECK_CONTIG(*t); CHECK_F32(*t);
  }
  const int64_t n = p.numel();
  TORCH_CHECK(g.numel() == n && m.numel() == n && v.numel() == n, "adam_step: arena size mismatch");
  TORCH_CHECK(step_t.numel() >= 1 && aux.numel() >= 2, "adam_step: step_t[1], aux[2] expected");
  TORCH_CHECK(starts.size() == ends.size(), "adam_step: ranges");
  TORCH_CHECK((reinterpret_cast<uintptr_t>(p.data_ptr()) % 16) == 0, "arena must be 16-byte aligned");
  ddl::DeviceGuard guard(p.device());
  hipStream_t s = cur_stream();
  ddl::launch_adam_inc_step(step_t.data_ptr<float>(), aux.data_ptr<float>(), b1, b2, s);
  for (size_t i = 0; i < starts.size(); ++i) {
    TORCH_CHECK(0 <= starts[i] && starts[i] <= ends[i] && ends[i] <= n, "adam_step: range out of bounds");
    TORCH_CHECK(starts[i] % 4 == 0 && (ends[i] - starts[i]) % 4 == 0, "adam_step: ranges must be float4 aligned");
    ddl::launch_adam_flat(p.data_ptr<float>(), g.data_ptr<float>(), m.data_ptr<float>(), v.data_ptr<float>(),
                          aux.data_ptr<float>(), starts[i], ends[i], lr, b1, b2, eps, wd, zero_grad, s);
  }
}

}  // namespace

void register_conv_ops(pybind11::module_& m);   // conv_bindings.cpp
void register_stem_ops(pybind11::module_& m);   // stem_bindings.cpp
void register_reducer(pybind11::module_& m);    // reducer.cpp
void register_p2p(pybind11::module_& m);        // p2p.cpp
void register_io(pybind11::module_& m);         // png_loader.cpp
void register_rccl(pybind11::module_& m);       // rccl_comm.cpp

// fence-test helpers (tests/test_gpu_fences.py): busy-wait on / scale a tensor on the current stream
void test_spin(double micros, int64_t device_index) {
  TORCH_CHECK(micros >= 0 && micros <= 2e6, "test_spin: at most 2 s");
  ddl::DeviceGuard guard(c10::Device(c10::kCUDA, (c10::DeviceIndex)device_index));
  ddl::launch_spin(micros, ddl::cur_stream());
}

void test_scale(at::Tensor t, double a) {
  DDL_CHECK_DEV(t); DDL_CHECK_CONTIG(t); DDL_CHECK_F32(t);
  ddl::DeviceGuard guard(t.device());
  ddl::launch_scale(t.data_ptr<float>(), t.numel(), (float)a, ddl::cur_stream());
}

PYBIND11_MODULE(TORCH_EXTENSION_NAME, m) {
  m.doc() = "hand-written gfx950 (MI355X) kernels of the ddl framework";
  m.def("normalize_u8", &normalize_u8, "uint8 NCHW -> fp32/255 (NHWC or NCHW)");
  m.def("ce_fwd_bwd", &ce_fwd_bwd, "fused softmax+CE loss, dlogits, argmax");
  m.def("adam_step", &adam_step, "flat fused Adam (+zero_grad)");
  m.def("copy_pair", &copy_pair, "two device-to-device copies in one launch");
  register_rccl(m);          // first: RcclComm is a default-argument type of the reducer / P2P engine bindings
  register_conv_ops(m);
  register_stem_ops(m);
  register_reducer(m);
  register_p2p(m);
  register_io(m);
  m.def("test_spin", &test_spin, "busy-wait `micros` on the current stream (bounded; fence tests)");
  m.def("test_scale", &test_scale, "t *= a on the current stream (fence tests)");
}
