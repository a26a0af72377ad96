// Bindings of the convolution / batch-norm engine kernels.  Every operand that walks over pixels
// is passed as a 2-D row-strided view [rows, channels] (stride(1) == 1), e.g. a channel slice of a
// dense block's NHWC buffer; leading dimensions come from stride(0).  All shape / bounds checks
// happen here, on the host, before anything is launched.
#include "torch_hip.h"
#include "ddl_kernels.h"

#include <cstdlib>

namespace {

using OptT = c10::optional<at::Tensor>;
using ddl::cur_stream;

void check_rows(const at::Tensor& t, const char* name, int64_t rows, int64_t ch) {
  TORCH_CHECK(t.is_cuda() && t.scalar_type() == at::kFloat, name, ": float32 GPU tensor expected");
  TORCH_CHECK(t.dim() == 2 && t.stride(1) == 1, name, ": 2-D view with unit channel stride expected");
  TORCH_CHECK(t.size(0) >= rows && t.size(1) == ch, name, ": expected [>=", rows, ", ", ch, "], got [",
              t.size(0), ", ", t.size(1), "]");
  TORCH_CHECK(t.stride(0) % 4 == 0 && (reinterpret_cast<uintptr_t>(t.data_ptr()) % 16) == 0, name,
              ": rows must be 16-byte aligned (float4 loads)");
  TORCH_CHECK(ch % 4 == 0, name, ": channel count must be a multiple of 4");
}

const float* vec_ptr(const OptT& v, const char* name, int64_t n) {
  if (!v.has_value()) return nullptr;
  const at::Tensor& t = *v;
  TORCH_CHECK(t.is_cuda() && t.scalar_type() == at::kFloat && t.dim() == 1 && t.stride(0) == 1 &&
                  t.size(0) >= n, name, ": contiguous float32 vector of >= ", n, " expected");
  TORCH_CHECK((reinterpret_cast<uintptr_t>(t.data_ptr()) % 16) == 0, name, ": 16-byte alignment required");
  return t.data_ptr<float>();
}

ddl::RowSrc make_src(const at::Tensor& x0, const OptT& x1, const OptT& s0, const OptT& s1, const OptT& s2,
                     bool relu, int64_t rows, int64_t ch, const char* name, bool fused_affine = false) {
  ddl::RowSrc r{};
  check_rows(x0, name, rows, ch);
  r.X0 = x0.data_ptr<float>();
  r.ld0 = (int)x0.stride(0);
  r.s0 = vec_ptr(s0, "s0", ch);
  r.s2 = vec_ptr(s2, "s2", ch);
  TORCH_CHECK((r.s0 == nullptr) == (r.s2 == nullptr), name, ": s0 and s2 come together");
  if (x1.has_value()) {
    check_rows(*x1, name, rows, ch);
    r.X1 = x1->data_ptr<float>();
    r.ld1 = (int)x1->stride(0);
    r.s1 = vec_ptr(s1, "s1", ch);
    TORCH_CHECK(fused_affine ? (r.s0 == nullptr && r.s1 == nullptr) : (r.s0 != nullptr && r.s1 != nullptr), name,
                ": a second source needs s0, s1, s2 (or fused coefficients instead of them)");
  }
  r.relu = relu ? 1 : 0;
  // extents for the buffer descriptors (32-bit offsets, hardware bounds check)
  const int64_t b0 = ((rows - 1) * (int64_t)r.ld0 + ch) * 4;
  const int64_t b1 = r.X1 ? ((rows - 1) * (int64_t)r.ld1 + ch) * 4 : 0;
  r.bytes0 = (long)b0;      // any size: the kernels rebase their descriptors per workgroup (device_utils.h)
  r.bytes1 = (long)b1;
  return r;
}

void check_acc(const at::Tensor& t, int64_t n, const char* name) {
  TORCH_CHECK(t.is_cuda() && t.scalar_type() == at::kLong && t.dim() == 3 && t.size(0) == ddl::kStatReplicas &&
                  t.size(1) == 2 && t.size(2) >= n && t.stride(2) == 1 && t.stride(0) == 2 * t.stride(1), name,
              ": [", ddl::kStatReplicas, ", 2, >=", n, "] int64 view expected");
}

float* fvec(const at::Tensor& t, int64_t n, const char* name) {
  TORCH_CHECK(t.is_cuda() && t.scalar_type() == at::kFloat && t.dim() == 1 && t.stride(0) == 1 && t.size(0) >= n, name,
              ": contiguous float32 vector of >= ", n, " expected");
  return t.data_ptr<float>();
}

#ifdef DDL_PHASE_PROBE
long long* g_phase_probe = nullptr;      // bench/phase_probe.py: [2][16] int64 device buffer, or null
int64_t g_phase_probe_n = 0;             // its length: the weight-gradient probe also logs (start, end) of 2048 workgroups
void set_phase_probe(const OptT& t) {
  if (!t.has_value()) { g_phase_probe = nullptr; g_phase_probe_n = 0; return; }
  TORCH_CHECK(t->is_cuda() && t->scalar_type() == at::kLong && t->is_contiguous() && t->numel() >= 32, "probe: int64[32]");
  g_phase_probe = reinterpret_cast<long long*>(t->data_ptr<int64_t>());
  g_phase_probe_n = t->numel();
}
#define DDL_SET_PROBE(p) (p).dbg = g_phase_probe
#else
#define DDL_SET_PROBE(p) do {} while (0)
#endif

// fused BN-backward coefficients (FusedK): accumulators + the BN's forward statistics
ddl::FusedK make_fk(const OptT& acc, const OptT& mean, const OptT& rstd, const OptT& scale, const OptT& dgamma,
                    const OptT& dbeta, double count, int64_t n) {
  ddl::FusedK f{};
  if (!acc.has_value()) return f;
  TORCH_CHECK(mean.has_value() && rstd.has_value() && scale.has_value() && count > 0, "fused k: mean, rstd, scale, count");
  check_acc(*acc, n, "k_acc");
  f.acc = reinterpret_cast<const long long*>(acc->data_ptr<int64_t>());
  f.acc_ld = (int)acc->stride(1);
  f.mean = fvec(*mean, n, "k_mean"); f.rstd = fvec(*rstd, n, "k_rstd"); f.scale = fvec(*scale, n, "k_scale");
  f.inv_count = 1.0 / count;
  if (dgamma.has_value()) {
    TORCH_CHECK(dbeta.has_value(), "dgamma and dbeta come together");
    f.dgamma = fvec(*dgamma, n, "dgamma"); f.dbeta = fvec(*dbeta, n, "dbeta");
  }
  return f;
}

// out (op)= prologue(x0, x1) (*) weight.  Returns the per-tile statistics table [G,2,N] (or empty).
at::Tensor conv_gemm(const at::Tensor& x0, const OptT& x1, const OptT& s0, const OptT& s1, const OptT& s2,
                     bool relu, int64_t M, int64_t H, int64_t W, int64_t taps, bool flip,
                     const at::Tensor& weight, bool b_kmajor, at::Tensor out, bool ep_mask, bool ep_accum,
                     const OptT& aux, const OptT& e0, const OptT& e1, int64_t stats_mode, int64_t force_bm, int64_t prec,
                     const OptT& acc_out, const OptT& k_acc, const OptT& k_mean, const OptT& k_rstd,
                     const OptT& k_scale, const OptT& k_dgamma, const OptT& k_dbeta, double k_count,
                     const OptT& tail_counters, int64_t tail_mode, double tail_count, const OptT& tail_mean,
                     const OptT& tail_rstd, const OptT& tail_scale, const OptT& tail_dgamma, const OptT& tail_dbeta,
                     const OptT& tail_k0, const OptT& tail_k1, const OptT& tail_k2) {
  ddl::DeviceGuard guard(x0.device());
  const int64_t C = x0.size(1), N = out.size(1);
  TORCH_CHECK(taps == 1 || taps == 9, "taps must be 1 or 9");
  TORCH_CHECK(M > 0 && M < (1LL << 31), "bad M");
  TORCH_CHECK(C % 32 == 0 && N % 4 == 0, "conv_gemm: C must be a multiple of 32 and N of 4");
  if (taps == 9) {
    TORCH_CHECK(H > 0 && W > 0 && M % (H * W) == 0, "M must be a whole number of HxW images");
  }
  ddl::ConvGemmParams p{};
  p.a = make_src(x0, x1, s0, s1, s2, relu, M, C, "conv_gemm input", k_acc.has_value());
  p.fk = make_fk(k_acc, k_mean, k_rstd, k_scale, k_dgamma, k_dbeta, k_count, C);
  p.M = (int)M; p.C = (int)C; p.N = (int)N; p.H = (int)H; p.W = (int)W; p.taps = (int)taps; p.flip = flip ? 1 : 0;
  TORCH_CHECK(weight.is_cuda() && weight.scalar_type() == at::kFloat && weight.dim() == 2 &&
                  weight.stride(1) == 1 && weight.stride(0) % 4 == 0 &&
                  (reinterpret_cast<uintptr_t>(weight.data_ptr()) % 16) == 0, "weight: 2-D aligned float32 expected");
  if (!b_kmajor) {
    TORCH_CHECK(weight.size(0) == N && weight.size(1) == taps * C, "weight must be [N, taps*C]");
  } else {
    TORCH_CHECK(weight.size(0) == C && weight.size(1) == taps * N, "k-major weight must be [C, taps*N]");
  }
  p.Wt = weight.data_ptr<float>(); p.ldb = (int)weight.stride(0); p.b_kmajor = b_kmajor ? 1 : 0;
  check_rows(out, "out", M, N);
  p.out = out.data_ptr<float>(); p.ldo = (int)out.stride(0);
  p.ep_mask = ep_mask ? 1 : 0; p.ep_accum = ep_accum ? 1 : 0;
  if (aux.has_value()) { check_rows(*aux, "aux", M, N); p.aux = aux->data_ptr<float>(); p.ldaux = (int)aux->stride(0); }
  p.e0 = vec_ptr(e0, "e0", N); p.e1 = vec_ptr(e1, "e1", N);
  if (ep_mask) { TORCH_CHECK(p.aux && p.e0 && p.e1, "ep_mask needs aux, e0, e1"); }
  if (ep_accum) { TORCH_CHECK(p.e0, "ep_accum needs e0"); }
  TORCH_CHECK(stats_mode >= 0 && stats_mode <= 2, "stats_mode");
  if (stats_mode == 2) { TORCH_CHECK(p.aux, "stats_mode 2 needs aux"); }
  p.stats = (int)stats_mode;
  TORCH_CHECK(prec == 0 || prec == 1 || prec == 3, "prec: 0 (fp32 MFMA), 1 (bf16), 3 (fp32 as three bf16 pieces)");
  p.bf16 = (int)prec;
  // operation kind: the combinations DenseNet uses (conv_gemm.hip)
  const bool two = x1.has_value();
  if (!two && !b_kmajor && !ep_mask && !ep_accum && stats_mode != 2) p.kind = 0;
  else if (!two && b_kmajor && !ep_mask && !ep_accum && stats_mode != 2) p.kind = 1;
  else if (two && b_kmajor && ep_mask && !ep_accum && stats_mode != 1) p.kind = 2;
  else if (two && b_kmajor && ep_mask && ep_accum && stats_mode != 1) p.kind = 3;
  else TORCH_CHECK(false, "conv_gemm: unsupported operand / epilogue combination");
  const int64_t cch = C / 32;
  TORCH_CHECK(taps * cch <= 320, "conv_gemm: K too large for the sub-slab index arithmetic");
  p.inv_cc = (int)((65536 + cch - 1) / cch);
  const int64_t wb = b_kmajor ? ((C - 1) * (int64_t)p.ldb + taps * N) * 4 : ((N - 1) * (int64_t)p.ldb + taps * C) * 4;
  const int64_t ob = ((M - 1) * (int64_t)p.ldo + N) * 4;
  const int64_t xb = p.aux ? ((M - 1) * (int64_t)p.ldaux + N) * 4 : 0;
  TORCH_CHECK(wb < (1LL << 31), "conv_gemm: the weight tensor must span < 2 GiB");
  p.wbytes = (unsigned)wb; p.obytes = (long)ob; p.auxbytes = (long)xb;
  if (p.fk.acc != nullptr) { TORCH_CHECK(p.kind == 3 && C <= ddl::kMaxAffineC, "fused k: 1x1 data gradient only"); }
  if (p.kind == 3) { TORCH_CHECK(C <= ddl::kMaxAffineC, "conv_gemm: dgrad1x1 stages its coefficients for <= 1024 channels"); }
  if (p.kind == 0 && p.a.s0 != nullptr) { TORCH_CHECK(C <= ddl::kMaxAffineC, "conv_gemm: BN prologue for <= 1024 channels"); }
  at::Tensor stats;
  if (acc_out.has_value()) {            // statistics into fixed-point accumulators instead of a per-tile table
    TORCH_CHECK(stats_mode != 0 && (p.kind == 0 || p.kind == 2 || p.kind == 3), "acc_out: forward / data-gradient statistics only");
    check_acc(*acc_out, N, "acc_out");
    p.bn.acc_out = reinterpret_cast<long long*>(acc_out->data_ptr<int64_t>());
    p.bn.acc_out_ld = (int)acc_out->stride(1);
  } else if (stats_mode) {
    const int G = ddl::conv_gemm_grid_x(p, (int)force_bm);
    stats = at::empty({G, 2, N}, x0.options());
    p.stats_out = stats.data_ptr<float>();
  }
  if (tail_counters.has_value()) {      // BN-backward bookkeeping in the kernel's tail (TailPrep, conv_params.h)
    TORCH_CHECK((p.kind == 2 || p.kind == 3) && acc_out.has_value() && stats_mode == 2,
                "tail prep: data-gradient kinds with accumulator statistics only");
    TORCH_CHECK(tail_mode == 0 || tail_mode == 1, "tail_mode");
    const at::Tensor& ct = *tail_counters;
    TORCH_CHECK(ct.is_cuda() && ct.scalar_type() == at::kInt && ct.is_contiguous() && ct.numel() >= (N + 31) / 32,
                "tail_counters: contiguous int32 tensor with one ticket per 32 output columns");
    TORCH_CHECK(tail_count > 0 && tail_mean.has_value() && tail_rstd.has_value() && tail_scale.has_value() &&
                    tail_dgamma.has_value() && tail_dbeta.has_value() && tail_k1.has_value() && tail_k2.has_value(),
                "tail prep: count, mean, rstd, scale, dgamma, dbeta, k1, k2");
    p.tail.counters = reinterpret_cast<unsigned*>(ct.data_ptr<int>());
    p.tail.mode = (int)tail_mode;
    static const int tail_fence = [] { const char* e = ddl_knob("DDL_TAIL_FENCE"); return (e && e[0] == '1') ? 1 : 0; }();
    p.tail.fence = tail_fence;
    p.tail.inv_count = 1.0 / tail_count;
    p.tail.mean = fvec(*tail_mean, N, "tail_mean"); p.tail.rstd = fvec(*tail_rstd, N, "tail_rstd");
    p.tail.scale = fvec(*tail_scale, N, "tail_scale");
    p.tail.dgamma = fvec(*tail_dgamma, N, "tail_dgamma"); p.tail.dbeta = fvec(*tail_dbeta, N, "tail_dbeta");
    p.tail.k1 = fvec(*tail_k1, N, "tail_k1"); p.tail.k2 = fvec(*tail_k2, N, "tail_k2");
    if (tail_mode == 0) { TORCH_CHECK(tail_k0.has_value(), "tail mode 0 needs k0"); p.tail.k0 = fvec(*tail_k0, N, "tail_k0"); }
  }
  DDL_SET_PROBE(p);
  ddl::launch_conv_gemm(p, (int)force_bm, cur_stream());
  return stats;
}

// Forward conv of a dense layer with fused batch-norm statistics (training): the BN(+ReLU) prologue's
// scale / shift come from the accumulators `acc_in` of the input channels (channels below `new0`: from the
// mean / var / rstd tables an earlier call filled), the epilogue adds the output's column sums to `acc_out`.
// Accumulators are [R, 2, ld] int64 views (sum, sum of squares; fixed point 2^28; R = kStatReplicas copies);
// `acc_out` may be a column slice of a wider table.
void conv_fwd_fused(const at::Tensor& x, const at::Tensor& acc_in, int64_t new0, const at::Tensor& gamma,
                    const at::Tensor& beta, const OptT& run_mean, const OptT& run_var, at::Tensor scale_out,
                    at::Tensor shift_out, at::Tensor mean, at::Tensor var, at::Tensor rstd, double count,
                    double momentum, double eps, int64_t M, int64_t H, int64_t W, int64_t taps,
                    const at::Tensor& weight, at::Tensor out, at::Tensor acc_out, int64_t force_bm, int64_t prec) {
  ddl::DeviceGuard guard(x.device());
  const int64_t C = x.size(1), N = out.size(1);
  TORCH_CHECK(taps == 1 || taps == 9, "taps must be 1 or 9");
  TORCH_CHECK(M > 0 && M < (1LL << 31) && count > 0, "bad M / count");
  TORCH_CHECK(C % 32 == 0 && C <= ddl::kMaxAffineC && N % 4 == 0, "conv_fwd_fused: C % 32 == 0, C <= 1024, N % 4 == 0");
  TORCH_CHECK(new0 >= 0 && new0 <= C, "new0 out of range");
  if (taps == 9) { TORCH_CHECK(H > 0 && W > 0 && M % (H * W) == 0, "M must be a whole number of HxW images"); }
  ddl::ConvGemmParams p{};
  p.a = make_src(x, c10::nullopt, c10::nullopt, c10::nullopt, c10::nullopt, true, M, C, "conv_fwd_fused input");
  p.M = (int)M; p.C = (int)C; p.N = (int)N; p.H = (int)H; p.W = (int)W; p.taps = (int)taps;
  TORCH_CHECK(weight.is_cuda() && weight.scalar_type() == at::kFloat && weight.dim() == 2 && weight.stride(1) == 1 &&
                  weight.stride(0) % 4 == 0 && (reinterpret_cast<uintptr_t>(weight.data_ptr()) % 16) == 0 &&
                  weight.size(0) == N && weight.size(1) == taps * C, "weight must be an aligned [N, taps*C] float32 tensor");
  p.Wt = weight.data_ptr<float>(); p.ldb = (int)weight.stride(0);
  check_rows(out, "out", M, N);
  p.out = out.data_ptr<float>(); p.ldo = (int)out.stride(0);
  check_acc(acc_in, C, "acc_in");
  check_acc(acc_out, N, "acc_out");
  p.bn.acc = reinterpret_cast<const long long*>(acc_in.data_ptr<int64_t>());
  p.bn.acc_ld = (int)acc_in.stride(1);
  p.bn.new0 = (int)new0;
  p.bn.acc_out = reinterpret_cast<long long*>(acc_out.data_ptr<int64_t>());
  p.bn.acc_out_ld = (int)acc_out.stride(1);
  p.bn.gamma = vec_ptr(gamma, "gamma", C);
  p.bn.beta = vec_ptr(beta, "beta", C);
  auto out_vec = [&](const at::Tensor& t, const char* name) {
    TORCH_CHECK(t.is_cuda() && t.scalar_type() == at::kFloat && t.dim() == 1 && t.stride(0) == 1 && t.size(0) >= C, name,
                ": contiguous float32 vector of >= ", C, " expected");
    return t.data_ptr<float>();
  };
  p.bn.scale_out = out_vec(scale_out, "scale_out");
  p.bn.shift_out = out_vec(shift_out, "shift_out");
  p.bn.mean = out_vec(mean, "mean");
  p.bn.var = out_vec(var, "var");
  p.bn.rstd = out_vec(rstd, "rstd");
  if (run_mean.has_value()) {
    TORCH_CHECK(run_var.has_value(), "run_mean and run_var come together");
    p.bn.run_mean = out_vec(*run_mean, "run_mean");
    p.bn.run_var = out_vec(*run_var, "run_var");
  }
  p.bn.inv_count = 1.0 / (count * ddl::kStatScale);
  p.bn.unbias = count > 1.0 ? (float)(count / (count - 1.0)) : 1.f;
  p.bn.momentum = (float)momentum;
  p.bn.eps = (float)eps;
  p.stats = 3;
  p.kind = 0;
  TORCH_CHECK(prec == 0 || prec == 1 || prec == 3, "prec: 0 (fp32 MFMA), 1 (bf16), 3 (fp32 as three bf16 pieces)");
  p.bf16 = (int)prec;
  const int64_t cch = C / 32;
  TORCH_CHECK(taps * cch <= 320, "conv_fwd_fused: K too large for the sub-slab index arithmetic");
  p.inv_cc = (int)((65536 + cch - 1) / cch);
  const int64_t wb = ((N - 1) * (int64_t)p.ldb + taps * C) * 4, ob = ((M - 1) * (int64_t)p.ldo + N) * 4;
  TORCH_CHECK(wb < (1LL << 31), "conv_fwd_fused: the weight tensor must span < 2 GiB");
  p.wbytes = (unsigned)wb; p.obytes = (long)ob;
  DDL_SET_PROBE(p);
  ddl::launch_conv_gemm(p, (int)force_bm, cur_stream());
}

void bn_acc_finalize(const at::Tensor& acc, int64_t c0, int64_t n, double count, double eps, at::Tensor chan_mean,
                     at::Tensor chan_var, at::Tensor chan_rstd) {
  ddl::DeviceGuard guard(acc.device());
  check_acc(acc, c0 + n, "acc");
  TORCH_CHECK(c0 >= 0 && n > 0 && count > 0, "bn_acc_finalize: bad range / count");
  for (const at::Tensor* t : std::initializer_list<const at::Tensor*>{&chan_mean, &chan_var, &chan_rstd})
    TORCH_CHECK(t->is_cuda() && t->scalar_type() == at::kFloat && t->is_contiguous() && t->numel() >= c0 + n,
                "channel stat table too small");
  ddl::launch_bn_acc_finalize(reinterpret_cast<const long long*>(acc.data_ptr<int64_t>()), (int)acc.stride(1), (int)c0,
                              (int)n, count, (float)eps, chan_mean.data_ptr<float>(), chan_var.data_ptr<float>(),
                              chan_rstd.data_ptr<float>(), cur_stream());
}

// Deferred slab reductions.  A weight-gradient kernel leaves S partial slabs; `dW += sum of slabs` of MANY weight
// gradients (all layers of a dense block) runs as one launch when the queue is flushed, on the stream that is
// current at that moment (the weight-gradient side stream).  The queue keeps the slab tensors alive until then.
class SlabQueue {
 public:
  void push(at::Tensor slabs, at::Tensor dW, int64_t S, int64_t total4) {
    entries_.push_back({std::move(slabs), std::move(dW), S, total4});
  }
  void flush() {
    size_t i = 0;
    while (i < entries_.size()) {
      ddl::SlabReduceParams p{};
      for (; i < entries_.size() && p.n < ddl::kMaxSlabEntries; ++i, ++p.n) {
        const Entry& e = entries_[i];
        p.e[p.n].slabs = e.slabs.data_ptr<float>();
        p.e[p.n].dW = e.dW.data_ptr<float>();
        p.e[p.n].S = (int)e.S;
        p.e[p.n].total4 = (long)e.total4;
      }
      ddl::DeviceGuard guard(entries_[0].dW.device());
      ddl::launch_slab_reduce_multi(p, cur_stream());
    }
    entries_.clear();
  }
  int64_t size() const { return (int64_t)entries_.size(); }

 private:
  struct Entry { at::Tensor slabs, dW; int64_t S, total4; };
  std::vector<Entry> entries_;
};

// dW[N, taps*C] += dO^T * rows(A).  With a `queue` the slab reduction is deferred to queue.flush().
void conv_wgrad(const at::Tensor& d0, const OptT& d1, const OptT& ds0, const OptT& ds1, const OptT& ds2,
                const at::Tensor& a0, const OptT& as0, const OptT& as2, bool a_relu, int64_t M, int64_t H,
                int64_t W, int64_t taps, at::Tensor dW, int64_t min_rows, const OptT& k_acc, const OptT& k_mean,
                const OptT& k_rstd, const OptT& k_scale, double k_count, SlabQueue* queue, int64_t force_sub,
                int64_t prec) {
  ddl::DeviceGuard guard(d0.device());
  const int64_t N = d0.size(1), C = a0.size(1);
  TORCH_CHECK(taps == 1 || taps == 9, "taps must be 1 or 9");
  TORCH_CHECK(prec == 0 || prec == 1, "conv_wgrad: prec 0 (fp32 MFMA) or 1 (bf16 MFMA inputs)");
  TORCH_CHECK(M > 0 && M < (1LL << 31), "bad M");
  ddl::WgradParams p{};
  p.d = make_src(d0, d1, ds0, ds1, ds2, false, M, N, "wgrad dO", k_acc.has_value());
  p.fk = make_fk(k_acc, k_mean, k_rstd, k_scale, c10::nullopt, c10::nullopt, k_count, N);
  if (p.fk.acc != nullptr) { TORCH_CHECK(taps == 1 && d1.has_value(), "fused k: 1x1 weight gradient with two dO sources"); }
  p.a = make_src(a0, c10::nullopt, as0, c10::nullopt, as2, a_relu, M, C, "wgrad activation");
  p.M = (int)M; p.C = (int)C; p.N = (int)N; p.H = (int)H; p.W = (int)W; p.taps = (int)taps; p.bf16 = (int)prec;
  if (taps == 9) {
    TORCH_CHECK(N <= 32 && N % 4 == 0, "3x3 wgrad: N <= 32 output channels");
    TORCH_CHECK(H > 0 && W > 0 && M % (H * W) == 0, "M must be a whole number of HxW images");
  }
  TORCH_CHECK(dW.is_cuda() && dW.scalar_type() == at::kFloat && dW.is_contiguous() &&
                  dW.numel() == N * taps * C && (reinterpret_cast<uintptr_t>(dW.data_ptr()) % 16) == 0,
              "dW must be a contiguous aligned [N, taps*C] float32 tensor");
  int other;
  if (taps == 1) {
    ddl::wgrad1x1_tiling((int)M, (int)C, (int)N, (int)force_sub, p.sub, p.ctiles);
    TORCH_CHECK(p.sub >= 1 && p.sub <= 4 && p.sub != 2 && (int64_t)p.ctiles * p.sub * 32 >= C, "wgrad tiling does not cover C");
    other = (int)(p.ctiles * ((N + 127) / 128));
  } else {
    other = (int)(3 * ((C + 127) / 128));
  }
  // resident workgroup slots: 1x1 with 128-column tiles 2 per CU (64 KB LDS), 32-column tiles 3-4 per CU (40 KB, registers)
  static const int target1 = [] { const char* e = ddl_knob("DDL_WGRAD_TARGET1"); return e ? atoi(e) : 512; }();
  p.rows_per_wg = ddl::wgrad_rows_per_wg((int)M, other, (int)min_rows, taps == 1 ? (p.sub == 1 ? target1 : 512) : 768);
  p.S = (int)((M + p.rows_per_wg - 1) / p.rows_per_wg);
  auto slabs = at::empty({p.S, N, taps * C}, d0.options());
  p.slabs = slabs.data_ptr<float>();
  DDL_SET_PROBE(p);
#ifdef DDL_PHASE_PROBE
  if (g_phase_probe_n < 32 + 3 * 2048) p.dbg = nullptr;
  static const int wko = [] { const char* e = ddl_knob("DDL_WGRAD_KO"); return e ? atoi(e) : 0; }();
  p.ko = wko;
#endif
  ddl::launch_wgrad(p, cur_stream());
  const int64_t total4 = N * taps * C / 4;
  if (queue != nullptr) {
    queue->push(slabs, dW, p.S, total4);
  } else {
    ddl::SlabReduceParams r{};
    r.n = 1;
    r.e[0].slabs = p.slabs; r.e[0].dW = dW.data_ptr<float>(); r.e[0].S = p.S; r.e[0].total4 = (long)total4;
    ddl::launch_slab_reduce_multi(r, cur_stream());
  }
}

// Weight gradients of up to 24 dense layers in ONE launch (conv_wgrad.hip: wgrad_group_kernel).  Entry i:
//   dW[i][N, taps*C_i] += (ds0[i]*d0[i] + ds1[i]*d1[i] + ds2[i])^T  (*)  relu(as0[i]*a0[i] + as2[i])
// All entries share M, N, H, W, taps and the row strides of their operands.  The slab reduction goes to `queue`.
void conv_wgrad_group(int64_t taps, std::vector<at::Tensor> d0, std::vector<at::Tensor> d1, std::vector<at::Tensor> ds0,
                      std::vector<at::Tensor> ds1, std::vector<at::Tensor> ds2, std::vector<at::Tensor> a0,
                      std::vector<at::Tensor> as0, std::vector<at::Tensor> as2, std::vector<at::Tensor> dW, int64_t M,
                      int64_t H, int64_t W, SlabQueue* queue, int64_t prec) {
  const size_t n = d0.size();
  TORCH_CHECK(prec == 0 || prec == 1, "conv_wgrad_group: prec 0 (fp32 MFMA) or 1 (bf16 MFMA inputs)");
  TORCH_CHECK(n >= 1 && n <= (size_t)ddl::kMaxWgradGroup, "conv_wgrad_group: 1..", ddl::kMaxWgradGroup, " layers");
  TORCH_CHECK(d1.size() == n && ds0.size() == n && ds1.size() == n && ds2.size() == n && a0.size() == n &&
                  as0.size() == n && as2.size() == n && dW.size() == n, "conv_wgrad_group: list lengths differ");
  TORCH_CHECK(queue != nullptr, "conv_wgrad_group: a SlabQueue is required");
  TORCH_CHECK(taps == 1 || taps == 9, "taps must be 1 or 9");
  TORCH_CHECK(M > 0 && M < (1LL << 31), "bad M");
  ddl::DeviceGuard guard(d0[0].device());
  const int64_t N = d0[0].size(1);
  if (taps == 9) {
    TORCH_CHECK(N <= 32 && N % 4 == 0, "3x3 wgrad: N <= 32 output channels");
    TORCH_CHECK(H > 0 && W > 0 && M % (H * W) == 0, "M must be a whole number of HxW images");
  }
  ddl::WgradGroupParams g{};
  g.n = (int)n; g.M = (int)M; g.N = (int)N; g.H = (int)H; g.W = (int)W; g.taps = (int)taps; g.a_relu = 1; g.bf16 = (int)prec;
  int64_t tiles = 0, slab_elems = 0;
  for (size_t i = 0; i < n; ++i) {
    const int64_t C = a0[i].size(1);
    TORCH_CHECK(d0[i].size(1) == N && d1[i].size(1) == N, "conv_wgrad_group: dO channel counts differ");
    const ddl::RowSrc d = make_src(d0[i], d1[i], ds0[i], ds1[i], ds2[i], false, M, N, "wgrad group dO");
    const ddl::RowSrc a = make_src(a0[i], c10::nullopt, as0[i], c10::nullopt, as2[i], true, M, C, "wgrad group activation");
    if (i == 0) { g.ld_d0 = d.ld0; g.ld_d1 = d.ld1; g.ld_a0 = a.ld0; }
    TORCH_CHECK(d.ld0 == g.ld_d0 && d.ld1 == g.ld_d1 && a.ld0 == g.ld_a0, "conv_wgrad_group: entries must share row strides");
    TORCH_CHECK(taps == 1 || C == 128, "3x3 wgrad group: 128 input channels");
    TORCH_CHECK(dW[i].is_cuda() && dW[i].scalar_type() == at::kFloat && dW[i].is_contiguous() &&
                    dW[i].numel() == N * taps * C && (reinterpret_cast<uintptr_t>(dW[i].data_ptr()) % 16) == 0,
                "dW must be a contiguous aligned [N, taps*C] float32 tensor");
    ddl::WgradGroupEntry& e = g.e[i];
    e.d0 = d.X0; e.d1 = d.X1; e.ds0 = d.s0; e.ds1 = d.s1; e.ds2 = d.s2;
    e.a0 = a.X0; e.as0 = a.s0; e.as2 = a.s2;
    e.C = (int)C;
    e.ctiles = (int)((C + 127) / 128);
    tiles += taps == 1 ? e.ctiles * ((N + 127) / 128) : 3 * ((C + 127) / 128);
    slab_elems += N * taps * C;
  }
  // pixel chunking: about `target` workgroups in all (two rounds of the resident slots), every chunk >= 4 steps
  static const int target = [] { const char* e = ddl_knob("DDL_WGRAD_GROUP_WGS"); return e ? atoi(e) : 768; }();
  int64_t chunks = (target + tiles - 1) / tiles;
  int64_t rows = (M + chunks - 1) / chunks;
  if (rows < 128) rows = 128;
  rows = (rows + 31) / 32 * 32;
  g.rows_per_wg = (int)rows;
  g.S = (int)((M + rows - 1) / rows);
  auto slabs = at::empty({(int64_t)g.S * slab_elems}, d0[0].options());
  int64_t off = 0;
  for (size_t i = 0; i < n; ++i) {
    const int64_t per = N * taps * a0[i].size(1);
    g.e[i].slabs = slabs.data_ptr<float>() + off;
    queue->push(slabs.narrow(0, off, (int64_t)g.S * per), dW[i], g.S, per / 4);
    off += (int64_t)g.S * per;
  }
  ddl::launch_wgrad_group(g, cur_stream());
}

void bn_prep(const OptT& partials, int64_t new0, int64_t n_new, double count, at::Tensor chan_mean,
             at::Tensor chan_var, at::Tensor chan_rstd, const OptT& gamma, const OptT& beta, const OptT& run_mean,
             const OptT& run_var, const OptT& scale, const OptT& shift, int64_t C, double momentum, double eps,
             bool train, const OptT& acc, int64_t acc_c0, int64_t acc_n) {
  ddl::DeviceGuard guard(chan_mean.device());
  const long long* accp = nullptr; int acc_ld = 0;
  if (acc.has_value()) {     // channels [acc_c0, acc_c0 + acc_n): statistics from fixed-point accumulators (FusedBn)
    TORCH_CHECK(train && acc_c0 >= 0 && acc_n > 0 && acc_c0 + acc_n <= C, "bn_prep: accumulator channel range");
    check_acc(*acc, acc_c0 + acc_n, "acc");
    accp = reinterpret_cast<const long long*>(acc->data_ptr<int64_t>());
    acc_ld = (int)acc->stride(1);
  }
  int G = 0, ldp = 0;
  const float* pp = nullptr;
  if (partials.has_value()) {
    const at::Tensor& t = *partials;
    TORCH_CHECK(t.is_cuda() && t.scalar_type() == at::kFloat && t.is_contiguous() && t.dim() == 3 && t.size(1) == 2,
                "partials: [G,2,n] expected");
    TORCH_CHECK(t.size(2) >= n_new && n_new % 32 == 0 && new0 % 32 == 0, "new channel range must be 32-aligned");
    G = (int)t.size(0); ldp = (int)t.size(2); pp = t.data_ptr<float>();
  }
  TORCH_CHECK(C % 32 == 0, "bn_prep: C must be a multiple of 32");
  for (const at::Tensor* t : std::initializer_list<const at::Tensor*>{&chan_mean, &chan_var, &chan_rstd})
    TORCH_CHECK(t->is_cuda() && t->scalar_type() == at::kFloat && t->is_contiguous() && t->numel() >= C &&
                    t->numel() >= new0 + n_new, "channel stat table too small");
  const float* g = vec_ptr(gamma, "gamma", C);
  TORCH_CHECK(!train || count > 0, "count");
  float* rm = run_mean.has_value() ? run_mean->data_ptr<float>() : nullptr;
  float* rv = run_var.has_value() ? run_var->data_ptr<float>() : nullptr;
  if (run_mean.has_value()) { TORCH_CHECK(run_mean->numel() >= C && run_var.has_value() && run_var->numel() >= C, "running stats"); }
  if (!train && g != nullptr) { TORCH_CHECK(rm && rv, "eval mode needs running stats"); }
  float* sc = scale.has_value() ? scale->data_ptr<float>() : nullptr;
  float* sh = shift.has_value() ? shift->data_ptr<float>() : nullptr;
  if (g != nullptr) { TORCH_CHECK(sc && sh && scale->numel() >= C && shift->numel() >= C && vec_ptr(beta, "beta", C), "scale/shift/beta"); }
  ddl::launch_bn_prep(pp, G, ldp, (int)new0, (int)n_new, count, chan_mean.data_ptr<float>(),
                      chan_var.data_ptr<float>(), chan_rstd.data_ptr<float>(), g, vec_ptr(beta, "beta", g ? C : 0), rm, rv,
                      sc, sh, (int)C, (float)momentum, (float)eps, train, accp, acc_ld, (int)acc_c0, (int)acc_n,
                      cur_stream());
}

void bn_bwd_prep(const at::Tensor& partials, double count, const at::Tensor& chan_mean, const at::Tensor& chan_rstd,
                 const at::Tensor& scale, at::Tensor dgamma, at::Tensor dbeta, const OptT& k0, at::Tensor k1,
                 at::Tensor k2, int64_t C, int64_t mode) {
  ddl::DeviceGuard guard(partials.device());
  TORCH_CHECK(partials.is_cuda() && partials.scalar_type() == at::kFloat && partials.is_contiguous() &&
                  partials.dim() == 3 && partials.size(1) == 2 && partials.size(2) >= C, "partials: [G,2,>=C]");
  TORCH_CHECK(C % 32 == 0, "C must be a multiple of 32");
  for (const at::Tensor* t : std::initializer_list<const at::Tensor*>{&chan_mean, &chan_rstd, &scale, &dgamma, &dbeta, &k1, &k2})
    TORCH_CHECK(t->is_cuda() && t->scalar_type() == at::kFloat && t->stride(-1) == 1 && t->numel() >= C, "vector too small");
  float* k0p = nullptr;
  if (mode == 0) { TORCH_CHECK(k0.has_value() && k0->numel() >= C, "mode 0 needs k0"); k0p = k0->data_ptr<float>(); }
  ddl::launch_bn_bwd_prep(partials.data_ptr<float>(), (int)partials.size(0), (int)partials.size(2), count,
                          chan_mean.data_ptr<float>(), chan_rstd.data_ptr<float>(), scale.data_ptr<float>(),
                          dgamma.data_ptr<float>(), dbeta.data_ptr<float>(), k0p, k1.data_ptr<float>(),
                          k2.data_ptr<float>(), (int)C, (int)mode, cur_stream());
}

// pixel glue ops (see bn_kernels.hip rows_op); returns the statistics table or an empty tensor
at::Tensor rows_op(int64_t mode, int64_t M, const at::Tensor& in, const OptT& x, const OptT& sc, const OptT& sh,
                   at::Tensor out, int64_t H, int64_t W, bool want_stats, const OptT& acc_out) {
  ddl::DeviceGuard guard(in.device());
  const int64_t C = out.size(1);
  TORCH_CHECK(mode >= 0 && mode <= 4, "rows_op mode");
  TORCH_CHECK(C % 4 == 0 && C / 4 <= 256, "rows_op: C must be a multiple of 4, <= 1024");
  check_rows(out, "out", M, C);
  int64_t in_rows = M;
  if (mode == 1) { TORCH_CHECK(H % 2 == 0 && W % 2 == 0 && M % ((H / 2) * (W / 2)) == 0, "pool fwd geometry"); in_rows = 0; }
  if (mode == 2) { TORCH_CHECK(H % 2 == 0 && W % 2 == 0 && M % (H * W) == 0, "pool bwd geometry"); in_rows = M / 4; }
  if (mode == 3) { TORCH_CHECK(H > 0 && W > 0 && M % (H * W) == 0, "gap bwd geometry"); in_rows = M / (H * W); }
  const float* inp = nullptr; int ldi = 0;
  if (mode != 1) { check_rows(in, "in", in_rows, C); inp = in.data_ptr<float>(); ldi = (int)in.stride(0); }
  const float* xp = nullptr; int ldx = 0;
  if (mode != 0) {
    TORCH_CHECK(x.has_value(), "rows_op: x required");
    check_rows(*x, "x", mode == 1 ? M * 4 : M, C);
    xp = x->data_ptr<float>(); ldx = (int)x->stride(0);
    TORCH_CHECK(sc.has_value() && sh.has_value(), "rows_op: sc/sh required");
  }
  const float* scp = vec_ptr(sc, "sc", C);
  const float* shp = vec_ptr(sh, "sh", C);
  at::Tensor stats;
  float* sp = nullptr;
  if (want_stats) {
    TORCH_CHECK(mode == 0 || mode == 2 || mode == 3, "stats only for copy / backward modes");
    stats = at::empty({ddl::rows_op_grid(M, (int)C), 2, C}, out.options());
    sp = stats.data_ptr<float>();
  }
  long long* accp = nullptr; int acc_ld = 0;
  if (acc_out.has_value()) {      // copy mode: channel sums into fixed-point accumulators instead of a table
    TORCH_CHECK(mode == 0 && !want_stats, "rows_op: acc_out is the copy mode's alternative to the statistics table");
    check_acc(*acc_out, C, "acc_out");
    accp = reinterpret_cast<long long*>(acc_out->data_ptr<int64_t>());
    acc_ld = (int)acc_out->stride(1);
  }
  ddl::launch_rows_op((int)mode, M, (int)C, inp, ldi, xp, ldx, scp, shp, out.data_ptr<float>(), (int)out.stride(0),
                      (int)H, (int)W, sp, accp, acc_ld, cur_stream());
  return stats;
}

at::Tensor bnrelu_gap_fwd(const at::Tensor& x, int64_t B, int64_t HW, const at::Tensor& sc, const at::Tensor& sh) {
  ddl::DeviceGuard guard(x.device());
  const int64_t C = x.size(1);
  check_rows(x, "x", B * HW, C);
  TORCH_CHECK(sc.numel() >= C && sh.numel() >= C && sc.is_cuda() && sh.is_cuda(), "sc/sh");
  auto pooled = at::empty({B, C}, x.options().memory_format(at::MemoryFormat::Contiguous));
  ddl::launch_bnrelu_gap_fwd(x.data_ptr<float>(), (int)x.stride(0), (int)B, (int)HW, sc.data_ptr<float>(),
                             sh.data_ptr<float>(), pooled.data_ptr<float>(), (int)C, cur_stream());
  return pooled;
}

}  // namespace

void register_conv_ops(pybind11::module_& m) {
  m.def("conv_gemm", &conv_gemm, "implicit-GEMM conv (1x1 / 3x3) with fused prologue + epilogue",
        pybind11::arg("x0"), pybind11::arg("x1"), pybind11::arg("s0"), pybind11::arg("s1"), pybind11::arg("s2"),
        pybind11::arg("relu"), pybind11::arg("M"), pybind11::arg("H"), pybind11::arg("W"), pybind11::arg("taps"),
        pybind11::arg("flip"), pybind11::arg("weight"), pybind11::arg("b_kmajor"), pybind11::arg("out"),
        pybind11::arg("ep_mask"), pybind11::arg("ep_accum"), pybind11::arg("aux"), pybind11::arg("e0"),
        pybind11::arg("e1"), pybind11::arg("stats_mode"), pybind11::arg("force_bm"), pybind11::arg("prec") = 0,
        pybind11::arg("acc_out") = c10::nullopt, pybind11::arg("k_acc") = c10::nullopt,
        pybind11::arg("k_mean") = c10::nullopt, pybind11::arg("k_rstd") = c10::nullopt,
        pybind11::arg("k_scale") = c10::nullopt, pybind11::arg("k_dgamma") = c10::nullopt,
        pybind11::arg("k_dbeta") = c10::nullopt, pybind11::arg("k_count") = 0.0,
        pybind11::arg("tail_counters") = c10::nullopt, pybind11::arg("tail_mode") = 0, pybind11::arg("tail_count") = 0.0,
        pybind11::arg("tail_mean") = c10::nullopt, pybind11::arg("tail_rstd") = c10::nullopt,
        pybind11::arg("tail_scale") = c10::nullopt, pybind11::arg("tail_dgamma") = c10::nullopt,
        pybind11::arg("tail_dbeta") = c10::nullopt, pybind11::arg("tail_k0") = c10::nullopt,
        pybind11::arg("tail_k1") = c10::nullopt, pybind11::arg("tail_k2") = c10::nullopt);
#ifdef DDL_PHASE_PROBE
  m.def("set_phase_probe", &set_phase_probe, "device buffer for the kernels' phase timestamps (probe builds only)");
#endif
  m.def("conv_fwd_fused", &conv_fwd_fused, "forward conv with fused batch-norm statistics (accumulator in / out)");
  m.attr("STAT_REPLICAS") = pybind11::int_(ddl::kStatReplicas);
  m.def("bn_acc_finalize", &bn_acc_finalize, "accumulators -> channel mean / var / rstd");
  m.def("conv_wgrad", &conv_wgrad, "conv weight gradient (split-M slabs + deterministic reduce)",
        pybind11::arg("d0"), pybind11::arg("d1"), pybind11::arg("ds0"), pybind11::arg("ds1"), pybind11::arg("ds2"),
        pybind11::arg("a0"), pybind11::arg("as0"), pybind11::arg("as2"), pybind11::arg("a_relu"), pybind11::arg("M"),
        pybind11::arg("H"), pybind11::arg("W"), pybind11::arg("taps"), pybind11::arg("dW"), pybind11::arg("min_rows"),
        pybind11::arg("k_acc") = c10::nullopt, pybind11::arg("k_mean") = c10::nullopt,
        pybind11::arg("k_rstd") = c10::nullopt, pybind11::arg("k_scale") = c10::nullopt, pybind11::arg("k_count") = 0.0,
        pybind11::arg("queue") = nullptr, pybind11::arg("force_sub") = 0, pybind11::arg("prec") = 0);
  m.def("conv_wgrad_group", &conv_wgrad_group, "weight gradients of up to 24 dense layers in one launch",
        pybind11::arg("taps"), pybind11::arg("d0"), pybind11::arg("d1"), pybind11::arg("ds0"), pybind11::arg("ds1"),
        pybind11::arg("ds2"), pybind11::arg("a0"), pybind11::arg("as0"), pybind11::arg("as2"), pybind11::arg("dW"),
        pybind11::arg("M"), pybind11::arg("H"), pybind11::arg("W"), pybind11::arg("queue"), pybind11::arg("prec") = 0);
  m.attr("MAX_WGRAD_GROUP") = pybind11::int_(ddl::kMaxWgradGroup);
  pybind11::class_<SlabQueue>(m, "SlabQueue")
      .def(pybind11::init<>())
      .def("flush", &SlabQueue::flush)
      .def("__len__", &SlabQueue::size);
  m.def("bn_prep", &bn_prep, "finalize channel statistics, derive BN scale/shift, update running stats",
        pybind11::arg("partials"), pybind11::arg("new0"), pybind11::arg("n_new"), pybind11::arg("count"),
        pybind11::arg("chan_mean"), pybind11::arg("chan_var"), pybind11::arg("chan_rstd"), pybind11::arg("gamma"),
        pybind11::arg("beta"), pybind11::arg("run_mean"), pybind11::arg("run_var"), pybind11::arg("scale"),
        pybind11::arg("shift"), pybind11::arg("C"), pybind11::arg("momentum"), pybind11::arg("eps"),
        pybind11::arg("train"), pybind11::arg("acc") = c10::nullopt, pybind11::arg("acc_c0") = 0,
        pybind11::arg("acc_n") = 0);
  m.def("bn_bwd_prep", &bn_bwd_prep, "BN backward per-channel coefficients, dgamma/dbeta");
  m.def("rows_op", &rows_op, "pixel glue ops (copy+stats, pool fwd/bwd, gap bwd, deferred-term apply)",
        pybind11::arg("mode"), pybind11::arg("M"), pybind11::arg("in"), pybind11::arg("x"), pybind11::arg("sc"),
        pybind11::arg("sh"), pybind11::arg("out"), pybind11::arg("H"), pybind11::arg("W"), pybind11::arg("want_stats"),
        pybind11::arg("acc_out") = c10::nullopt);
  m.def("bnrelu_gap_fwd", &bnrelu_gap_fwd, "BN+ReLU+global average pool");
}
