// Bindings of the stem (conv0 / norm0 / relu0 / pool0) and classifier kernels.  Each binding runs
// a whole sub-pipeline of launches on the current stream (graph-capturable: allocations go through
// the caching allocator, nothing synchronises).  All geometry checks happen here on the host.
#include "torch_hip.h"
#include "ddl_kernels.h"

namespace {

using ddl::cur_stream;

void check_vec(const at::Tensor& t, const char* name, int64_t n) {
  TORCH_CHECK(t.is_cuda() && t.scalar_type() == at::kFloat && t.stride(-1) == 1 && t.numel() >= n, name,
              ": contiguous float32 GPU vector of >= ", n, " expected");
  TORCH_CHECK((reinterpret_cast<uintptr_t>(t.data_ptr()) % 16) == 0, name, ": 16-byte alignment required");
}

struct StemGeom {
  int64_t B, H, W, H0, W0, H1, W1;
};

StemGeom stem_geom(int64_t B, int64_t H, int64_t W) {
  StemGeom g{};
  g.B = B; g.H = H; g.W = W;
  g.H0 = (H + 6 - 7) / 2 + 1; g.W0 = (W + 6 - 7) / 2 + 1;
  g.H1 = (g.H0 + 2 - 3) / 2 + 1; g.W1 = (g.W0 + 2 - 3) / 2 + 1;
  return g;
}

// forward: returns {pooled [B,H1,W1,64], xp [B,H,W,4], y0 [B,H0,W0,64], stats [3,64] (mean,var,rstd),
//                   ss [2,64] (scale, shift), idx uint8 [B,H1,W1,64] (train only, else empty)}
std::vector<at::Tensor> stem_fwd(const at::Tensor& x, const at::Tensor& w0, const at::Tensor& gamma,
                                 const at::Tensor& beta, at::Tensor run_mean, at::Tensor run_var, bool train,
                                 double momentum, double eps) {
  TORCH_CHECK(x.is_cuda() && x.dim() == 4 && x.size(1) == 3 && x.is_contiguous(),
              "stem: contiguous NCHW [B,3,H,W] GPU image batch expected");
  TORCH_CHECK(x.scalar_type() == at::kByte || x.scalar_type() == at::kFloat, "stem: uint8 or float32 images");
  TORCH_CHECK(w0.is_cuda() && w0.scalar_type() == at::kFloat && w0.is_contiguous() && w0.numel() == 64 * 3 * 7 * 7,
              "conv0 weight must be a contiguous float32 [64,3,7,7]");
  check_vec(gamma, "norm0.weight", 64); check_vec(beta, "norm0.bias", 64);
  check_vec(run_mean, "norm0.running_mean", 64); check_vec(run_var, "norm0.running_var", 64);
  ddl::DeviceGuard guard(x.device());
  const StemGeom g = stem_geom(x.size(0), x.size(2), x.size(3));
  TORCH_CHECK(g.H >= 7 && g.W >= 7 && g.H1 >= 1 && g.W1 >= 1, "image too small for the stem");
  const int64_t M0 = g.B * g.H0 * g.W0;
  TORCH_CHECK(M0 > 0 && M0 < (1LL << 31), "stem: batch too large (pixel rows are 32-bit)");
  auto f32 = x.options().dtype(at::kFloat);
  hipStream_t s = cur_stream();
  auto xp = at::empty({g.B, g.H, g.W, 4}, f32);
  ddl::launch_stem_pack(x.data_ptr(), x.scalar_type() == at::kByte, xp.data_ptr<float>(), (int)g.B, (int)g.H,
                        (int)g.W, s);
  auto wp = at::empty({64, 224}, f32);
  ddl::launch_stem_pack_weight(w0.data_ptr<float>(), wp.data_ptr<float>(), s);
  auto y0 = at::empty({g.B, g.H0, g.W0, 64}, f32);
  at::Tensor part;
  float* pp = nullptr;
  const int G = ddl::stem_conv_grid(M0);
  if (train) {
    part = at::empty({G, 2, 64}, f32);
    pp = part.data_ptr<float>();
  }
  ddl::launch_stem_conv_fwd(xp.data_ptr<float>(), wp.data_ptr<float>(), y0.data_ptr<float>(), pp, M0, (int)g.H,
                            (int)g.W, (int)g.H0, (int)g.W0, s);
  auto stats = at::empty({3, 64}, f32);
  auto ss = at::empty({2, 64}, f32);
  float* st = stats.data_ptr<float>();
  ddl::launch_bn_prep(pp, train ? G : 0, 64, 0, train ? 64 : 0, (double)M0, st, st + 64, st + 128,
                      gamma.data_ptr<float>(), beta.data_ptr<float>(), run_mean.data_ptr<float>(),
                      run_var.data_ptr<float>(), ss.data_ptr<float>(), ss.data_ptr<float>() + 64, 64,
                      (float)momentum, (float)eps, train, nullptr, 0, 0, 0, s);
  auto pooled = at::empty({g.B, g.H1, g.W1, 64}, f32);
  at::Tensor idx;
  uint8_t* ip = nullptr;
  if (train) {
    idx = at::empty({g.B, g.H1, g.W1, 64}, x.options().dtype(at::kByte));
    ip = idx.data_ptr<uint8_t>();
  } else {
    idx = at::empty({0}, x.options().dtype(at::kByte));
  }
  ddl::launch_stem_pool_fwd(y0.data_ptr<float>(), ss.data_ptr<float>(), ss.data_ptr<float>() + 64,
                            pooled.data_ptr<float>(), ip, (int)g.B, (int)g.H0, (int)g.W0, (int)g.H1, (int)g.W1, s);
  return {pooled, xp, y0, stats, ss, idx};
}

// backward: accumulates into the arena gradient views of conv0.weight / norm0.weight / norm0.bias
void stem_bwd(const at::Tensor& dp, const at::Tensor& xp, const at::Tensor& y0, const at::Tensor& stats,
              const at::Tensor& ss, const at::Tensor& idx, at::Tensor gw0, at::Tensor ggamma, at::Tensor gbeta) {
  TORCH_CHECK(xp.is_cuda() && xp.dim() == 4 && xp.size(3) == 4 && xp.is_contiguous() &&
                  xp.scalar_type() == at::kFloat, "xp: [B,H,W,4] float32 expected");
  ddl::DeviceGuard guard(xp.device());
  const StemGeom g = stem_geom(xp.size(0), xp.size(1), xp.size(2));
  const int64_t M0 = g.B * g.H0 * g.W0;
  auto chk = [&](const at::Tensor& t, const char* name, int64_t h, int64_t w, at::ScalarType ty) {
    TORCH_CHECK(t.is_cuda() && t.scalar_type() == ty && t.is_contiguous() && t.dim() == 4 && t.size(0) == g.B &&
                    t.size(1) == h && t.size(2) == w && t.size(3) == 64, name, ": bad shape / dtype / layout");
    TORCH_CHECK((reinterpret_cast<uintptr_t>(t.data_ptr()) % 16) == 0, name, ": 16-byte alignment required");
  };
  chk(dp, "dp", g.H1, g.W1, at::kFloat);
  chk(idx, "idx", g.H1, g.W1, at::kByte);
  chk(y0, "y0", g.H0, g.W0, at::kFloat);
  TORCH_CHECK(stats.is_cuda() && stats.is_contiguous() && stats.numel() == 3 * 64 && ss.is_cuda() &&
                  ss.is_contiguous() && ss.numel() == 2 * 64, "stats [3,64] / ss [2,64]");
  TORCH_CHECK(gw0.is_cuda() && gw0.scalar_type() == at::kFloat && gw0.is_contiguous() && gw0.numel() == 64 * 147,
              "conv0 weight gradient must be contiguous [64,3,7,7]");
  check_vec(ggamma, "norm0.weight.grad", 64); check_vec(gbeta, "norm0.bias.grad", 64);
  hipStream_t s = cur_stream();
  auto f32 = xp.options();
  auto g0 = at::empty({M0, 64}, f32);
  const int G = ddl::stem_pool_bwd_grid(M0);
  auto part = at::empty({G, 2, 64}, f32);
  const float* ssp = ss.data_ptr<float>();
  ddl::launch_stem_pool_bwd(y0.data_ptr<float>(), ssp, ssp + 64, dp.data_ptr<float>(), idx.data_ptr<uint8_t>(),
                            g0.data_ptr<float>(), part.data_ptr<float>(), (int)g.B, (int)g.H0, (int)g.W0,
                            (int)g.H1, (int)g.W1, s);
  auto k = at::empty({3, 64}, f32);
  float* kp = k.data_ptr<float>();
  const float* st = stats.data_ptr<float>();
  ddl::launch_bn_bwd_prep(part.data_ptr<float>(), G, 64, (double)M0, st, st + 128, ssp, ggamma.data_ptr<float>(),
                          gbeta.data_ptr<float>(), kp, kp + 64, kp + 128, 64, 0, s);
  const int S = ddl::stem_wgrad_slabs((int)g.B, (int)g.H0, (int)g.W0);
  auto slabs = at::empty({S, 64, 224}, f32);
  ddl::launch_stem_wgrad(g0.data_ptr<float>(), y0.data_ptr<float>(), kp, kp + 64, kp + 128, xp.data_ptr<float>(),
                         slabs.data_ptr<float>(), gw0.data_ptr<float>(), (int)g.B, (int)g.H, (int)g.W, (int)g.H0,
                         (int)g.W0, s);
}

at::Tensor linear_fwd(const at::Tensor& x, const at::Tensor& w, const at::Tensor& bias) {
  TORCH_CHECK(x.is_cuda() && x.scalar_type() == at::kFloat && x.is_contiguous() && x.dim() == 2, "x: [B,K] float32");
  TORCH_CHECK(w.is_cuda() && w.scalar_type() == at::kFloat && w.is_contiguous() && w.dim() == 2 &&
                  w.size(1) == x.size(1), "w: [N,K] float32");
  TORCH_CHECK(bias.is_cuda() && bias.scalar_type() == at::kFloat && bias.numel() == w.size(0), "bias: [N]");
  ddl::DeviceGuard guard(x.device());
  auto out = at::empty({x.size(0), w.size(0)}, x.options());
  ddl::launch_linear_fwd(x.data_ptr<float>(), w.data_ptr<float>(), bias.data_ptr<float>(), out.data_ptr<float>(),
                         (int)x.size(0), (int)x.size(1), (int)w.size(0), cur_stream());
  return out;
}

// returns dx; dW and db are accumulated in place
at::Tensor linear_bwd(const at::Tensor& dy, const at::Tensor& x, const at::Tensor& w, at::Tensor dW, at::Tensor db) {
  TORCH_CHECK(x.is_cuda() && x.scalar_type() == at::kFloat && x.is_contiguous() && x.dim() == 2, "x: [B,K] float32");
  TORCH_CHECK(w.is_cuda() && w.scalar_type() == at::kFloat && w.is_contiguous() && w.dim() == 2 &&
                  w.size(1) == x.size(1), "w: [N,K] float32");
  TORCH_CHECK(dy.is_cuda() && dy.scalar_type() == at::kFloat && dy.is_contiguous() && dy.dim() == 2 &&
                  dy.size(0) == x.size(0) && dy.size(1) == w.size(0), "dy: [B,N] float32");
  TORCH_CHECK(dW.is_cuda() && dW.scalar_type() == at::kFloat && dW.is_contiguous() && dW.numel() == w.numel() &&
                  db.is_cuda() && db.scalar_type() == at::kFloat && db.is_contiguous() && db.numel() == w.size(0),
              "dW / db: gradient views of w / bias");
  ddl::DeviceGuard guard(x.device());
  auto dx = at::empty_like(x);
  ddl::launch_linear_bwd(dy.data_ptr<float>(), x.data_ptr<float>(), w.data_ptr<float>(), dx.data_ptr<float>(),
                         dW.data_ptr<float>(), db.data_ptr<float>(), (int)x.size(0), (int)x.size(1), (int)w.size(0),
                         cur_stream());
  return dx;
}

}  // namespace

void register_stem_ops(pybind11::module_& m) {
  m.def("stem_fwd", &stem_fwd, "conv0 7x7/2 (MFMA) + norm0 + ReLU + maxpool 3x3/2");
  m.def("stem_bwd", &stem_bwd, "stem backward: pool/ReLU/BN masks + conv0 weight gradient");
  m.def("linear_fwd", &linear_fwd, "classifier forward");
  m.def("linear_bwd", &linear_bwd, "classifier backward (dx returned, dW/db accumulated)");
}
