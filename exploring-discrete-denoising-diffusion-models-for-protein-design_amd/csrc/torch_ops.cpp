// torch.ops.d3pm_mi355x.* -- the PyTorch-ROCm extension form of the boundary.
//
// A thin shim: every operator checks its tensors, allocates its outputs (and workspace) with torch's
// allocator, and calls the C ABI of libd3pm_hip.so (include/d3pm_hip.h) on torch's current HIP stream.  No
// arithmetic happens here.  The operators carry the reference function they stand for in their schema comment;
// autograd nodes live in the Python facade, which calls these operators.  Status codes of the library become
// the exceptions the reference raises: D3PM_EINVAL -> ValueError, D3PM_EUNSUPPORTED -> NotImplementedError,
// D3PM_EHIP -> RuntimeError.  Every operator makes the device of its first tensor current for the call and checks
// every tensor's element count against the batch it describes before a pointer reaches the library: a mis-shaped
// call raises, it does not read or write out of bounds.
#include <ATen/ATen.h>
#include <ATen/DeviceGuard.h>
#include <c10/hip/HIPStream.h>
#include <torch/library.h>

#include <tuple>
#include <vector>

#include "../../include/d3pm_hip.h"

namespace {

using at::Tensor;
using OptTensor = c10::optional<Tensor>;

void check_rc(int rc) {
  if (rc == D3PM_OK) return;
  const char* msg = d3pm_last_error();
  TORCH_CHECK_VALUE(rc != D3PM_EINVAL, msg);
  TORCH_CHECK_NOT_IMPLEMENTED(rc != D3PM_EUNSUPPORTED, msg);
  TORCH_CHECK(false, "libd3pm_hip: ", msg, " (code ", rc, ")");
}

void* stream() { return (void*)c10::hip::getCurrentHIPStream().stream(); }

void need_device(const Tensor& t, const char* what) {
  TORCH_CHECK(t.is_cuda(), "d3pm_mi355x: ", what, " must live on an MI355X (device='cuda'); there is no CPU path");
  TORCH_CHECK(t.is_contiguous(), "d3pm_mi355x: ", what, " must be contiguous");
}
int tok_bytes(const Tensor& t, const char* what) {
  need_device(t, what);
  if (t.scalar_type() == at::kLong) return 8;
  TORCH_CHECK(t.scalar_type() == at::kInt, "d3pm_mi355x: ", what, " must be int32 or int64");
  return 4;
}
const void* opt_ptr(const OptTensor& t, at::ScalarType st, const char* what) {
  if (!t.has_value() || !t->defined()) return nullptr;
  need_device(*t, what);
  TORCH_CHECK(t->scalar_type() == st, "d3pm_mi355x: ", what, " has the wrong dtype");
  return t->data_ptr();
}
const float* f32(const Tensor& t, const char* what) {
  need_device(t, what);
  TORCH_CHECK(t.scalar_type() == at::kFloat, "d3pm_mi355x: ", what, " must be float32");
  return t.data_ptr<float>();
}

// ---- shape contracts (element counts: the library sees flat [B, npos, ...] buffers)
void need_numel(const Tensor& t, int64_t n, const char* what) {
  TORCH_CHECK_VALUE(t.numel() == n, "d3pm_mi355x: ", what, " has ", t.numel(), " elements, expected ", n);
}
void need_numel(const OptTensor& t, int64_t n, const char* what) {
  if (t.has_value() && t->defined()) need_numel(*t, n, what);
}
void need_same_shape(const Tensor& a, const Tensor& b, const char* what) {
  TORCH_CHECK_VALUE(a.sizes() == b.sizes(), "d3pm_mi355x: ", what, " must have the same shape, got ", a.sizes(), " and ",
                    b.sizes());
}
// rows [B, npos, ld] with ld >= K
int64_t need_rows(const Tensor& rows, int B, long npos, int64_t K, const char* what) {
  TORCH_CHECK_VALUE(rows.dim() >= 1, "d3pm_mi355x: ", what, " needs a trailing class dimension");
  const int64_t ld = rows.size(-1);
  TORCH_CHECK_VALUE(ld >= K, "d3pm_mi355x: ", what, " has rows of ", ld, " < K = ", K, " entries");
  need_numel(rows, (int64_t)B * npos * ld, what);
  return ld;
}
void need_image_tables(at::TensorList tables, int64_t K, int64_t T, bool structured) {
  TORCH_CHECK(tables.size() == 5 || tables.size() == 7,
              "d3pm_mi355x: tables = [qbar_digest, qone_digest, qbar, qoneT, qbarT(, qbar_f16x2, qbarT_f16x2)]");
  TORCH_CHECK_VALUE(K >= 2 && K <= 256 && T >= 1, "d3pm_mi355x: need 2 <= K <= 256 and T >= 1");
  need_numel(tables[0], T * 8 * K, "qbar_digest (float32 [T, 8, K])");
  need_numel(tables[1], T * 8 * K, "qone_digest (float32 [T, 8, K])");
  if (!structured)
    for (int i = 2; i < 5; ++i) need_numel(tables[i], T * K * K, "dense table copy (float32 [T, K, K])");
  if (tables.size() == 7) {
    const bool h0 = tables[5].defined() && tables[5].numel() > 0, h1 = tables[6].defined() && tables[6].numel() > 0;
    TORCH_CHECK_VALUE(h0 == h1, "d3pm_mi355x: the float16 split tables come as a pair (qbar_f16x2, qbarT_f16x2)");
    if (h0) {
      TORCH_CHECK_VALUE(!structured && (K == 64 || K == 128 || K == 256),
                        "d3pm_mi355x: float16 split tables serve dense tables with K = 64, 128 or 256");
      for (int i = 5; i < 7; ++i) {
        need_device(tables[i], "float16 split table");
        TORCH_CHECK(tables[i].scalar_type() == at::kHalf, "d3pm_mi355x: split tables are float16");
        need_numel(tables[i], T * 2 * K * K, "split table (float16 [T, K / 32, K / 16, 2, 64, 8])");
      }
    }
  }
}
void need_protein_tables(const Tensor& qbar_digest, const Tensor& qone_digest, int64_t K, int64_t T) {
  TORCH_CHECK_VALUE(K >= 2 && K <= 256 && T >= 1, "d3pm_mi355x: need 2 <= K <= 256 and T >= 1");
  need_numel(qbar_digest, T * 4 * K, "qbar_digest (float64 [T, 4, K])");
  need_numel(qone_digest, T * 4 * K, "qone_digest (float64 [T, 4, K])");
}
void need_counter(const OptTensor& c, const char* what) {
  if (c.has_value() && c->defined()) need_numel(*c, 1, what);
}
std::vector<int64_t> with_k(const Tensor& tokens, int64_t K) {
  auto s = tokens.sizes().vec();
  s.push_back(K);
  return s;
}

// tables = [qbar_digest, qone_digest, qbar, qoneT, qbarT(, qbar_f16x2, qbarT_f16x2)]; all but the digests may be
// empty tensors (structured tables; no float16 split tables: the float32 matrix instructions serve K = 256)
d3pm_tables_f32 image_tables(at::TensorList tables, int64_t K, int64_t T, bool structured, double eps) {
  need_image_tables(tables, K, T, structured);
  d3pm_tables_f32 tb{};
  tb.K = (int)K; tb.T = (int)T; tb.structured = structured ? 1 : 0; tb.eps = (float)eps;
  auto get = [&](int i, const char* what) -> const float* {
    return tables[i].defined() && tables[i].numel() > 0 ? f32(tables[i], what) : nullptr;
  };
  tb.qbar_digest = get(0, "qbar_digest"); tb.qone_digest = get(1, "qone_digest");
  tb.qbar = get(2, "qbar"); tb.qoneT = get(3, "qoneT"); tb.qbarT = get(4, "qbarT");
  if (tables.size() == 7 && tables[5].defined() && tables[5].numel() > 0) {
    tb.qbar_f16x2 = (const uint16_t*)tables[5].data_ptr();
    tb.qbarT_f16x2 = (const uint16_t*)tables[6].data_ptr();
  }
  return tb;
}
d3pm_tables_f64 protein_tables(const Tensor& qbar_digest, const Tensor& qone_digest, const Tensor& lvb_digest, int64_t K,
                               int64_t T) {
  need_protein_tables(qbar_digest, qone_digest, K, T);
  need_device(qbar_digest, "qbar_digest"); need_device(qone_digest, "qone_digest");
  TORCH_CHECK(qbar_digest.scalar_type() == at::kDouble && qone_digest.scalar_type() == at::kDouble,
              "d3pm_mi355x: protein digests are float64");
  const float* lvb = nullptr;
  if (lvb_digest.defined() && lvb_digest.numel() > 0) {        // empty: no float32 loss digest (K > 32)
    need_numel(lvb_digest, (int64_t)(d3pm_lvb_digest_f32_size((int)T) / sizeof(float)), "lvb_digest");
    lvb = f32(lvb_digest, "lvb_digest");
  }
  return d3pm_tables_f64{(int)K, (int)T, qbar_digest.data_ptr<double>(), qone_digest.data_ptr<double>(), lvb};
}
std::pair<int, long> dims(const Tensor& tokens) {
  const int B = tokens.dim() ? (int)tokens.size(0) : 1;
  return {B, B ? (long)(tokens.numel() / B) : 0};
}

#define D3PM_GUARD(t) const at::DeviceGuard device_guard_((t).device())

// DiscreteDiffusion.q_probs, d3pm/diffusion_discrete.py:250-263
Tensor q_probs(at::TensorList tables, int64_t K, int64_t T, bool structured, double eps, const Tensor& x0, const Tensor& t) {
  D3PM_GUARD(x0);
  auto tb = image_tables(tables, K, T, structured, eps);
  auto [B, npos] = dims(x0);
  need_numel(t, B, "t");
  Tensor out = at::empty(with_k(x0, K), x0.options().dtype(at::kFloat));
  check_rc(d3pm_q_probs(&tb, x0.data_ptr(), tok_bytes(x0, "x0"), t.data_ptr(), tok_bytes(t, "t"), B, npos,
                        out.data_ptr<float>(), stream()));
  return out;
}

// DiscreteDiffusion.q_sample, :266-285
Tensor q_sample(at::TensorList tables, int64_t K, int64_t T, bool structured, double eps, const Tensor& x0, const Tensor& t,
                const OptTensor& noise, int64_t noise_kind, int64_t seed, int64_t offset, const OptTensor& offset_dev) {
  D3PM_GUARD(x0);
  auto tb = image_tables(tables, K, T, structured, eps);
  auto [B, npos] = dims(x0);
  need_numel(t, B, "t");
  need_numel(noise, (int64_t)B * npos * K, "noise");
  need_counter(offset_dev, "offset_dev");
  Tensor out = at::empty(x0.sizes(), x0.options().dtype(at::kLong));
  check_rc(d3pm_q_sample(&tb, x0.data_ptr(), tok_bytes(x0, "x0"), t.data_ptr(), tok_bytes(t, "t"), B, npos,
                         (const float*)opt_ptr(noise, at::kFloat, "noise"), (int)noise_kind, (uint64_t)seed,
                         (uint64_t)offset, (const uint64_t*)opt_ptr(offset_dev, at::kLong, "offset_dev"),
                         out.data_ptr<int64_t>(), stream()));
  return out;
}

// the first two lines of training_losses, :549-556, with device noise: (t int32 [B], x_t int64) in one launch;
// `clear` (a float32 scalar, optional) is zeroed by the launch
std::tuple<Tensor, Tensor> q_sample_draw_t(at::TensorList tables, int64_t K, int64_t T, bool structured, double eps,
                                           const Tensor& x0, int64_t seed, int64_t offset, const OptTensor& offset_dev,
                                           const OptTensor& clear) {
  D3PM_GUARD(x0);
  auto tb = image_tables(tables, K, T, structured, eps);
  auto [B, npos] = dims(x0);
  need_counter(offset_dev, "offset_dev");
  need_counter(clear, "clear");
  Tensor t = at::empty({B}, x0.options().dtype(at::kInt));
  Tensor x_t = at::empty(x0.sizes(), x0.options().dtype(at::kLong));
  check_rc(d3pm_q_sample_draw_t(&tb, x0.data_ptr(), tok_bytes(x0, "x0"), B, npos, (uint64_t)seed, (uint64_t)offset,
                                (const uint64_t*)opt_ptr(offset_dev, at::kLong, "offset_dev"),
                                (float*)opt_ptr(clear, at::kFloat, "clear"), t.data_ptr<int32_t>(),
                                x_t.data_ptr<int64_t>(), stream()));
  return {t, x_t};
}

// DiscreteDiffusion.q_posterior_logits, :312-336 (and p_logits :339-381 with x_start_is_logits)
Tensor q_posterior_logits(at::TensorList tables, int64_t K, int64_t T, bool structured, double eps, const Tensor& x_start,
                          bool x_start_is_logits, const Tensor& x_t, const Tensor& t) {
  D3PM_GUARD(x_t);
  auto tb = image_tables(tables, K, T, structured, eps);
  auto [B, npos] = dims(x_t);
  need_numel(t, B, "t");
  if (x_start_is_logits) need_rows(x_start, B, npos, K, "x_start logits");
  else need_same_shape(x_start, x_t, "x_start and x_t");
  Tensor out = at::empty(with_k(x_t, K), x_t.options().dtype(at::kFloat));
  const void* xs = x_start_is_logits ? (const void*)f32(x_start, "x_start logits") : x_start.data_ptr();
  const int xb = x_start_is_logits ? 8 : tok_bytes(x_start, "x_start");
  check_rc(d3pm_q_posterior_logits(&tb, xs, x_start_is_logits ? 1 : 0, x_start_is_logits ? x_start.size(-1) : K, xb,
                                   x_t.data_ptr(), tok_bytes(x_t, "x_t"), t.data_ptr(), tok_bytes(t, "t"), B, npos,
                                   out.data_ptr<float>(), stream()));
  return out;
}

// gradient of the above with respect to the x_start logits (autograd of :320-336)
Tensor q_posterior_logits_backward(at::TensorList tables, int64_t K, int64_t T, bool structured, double eps,
                                   const Tensor& x_start_logits, const Tensor& t, const Tensor& grad_out) {
  D3PM_GUARD(x_start_logits);
  auto tb = image_tables(tables, K, T, structured, eps);
  TORCH_CHECK_VALUE(x_start_logits.dim() >= 2, "d3pm_mi355x: x_start logits are [B, ..., K]");
  const int B = (int)x_start_logits.size(0);
  const long npos = B ? (long)(x_start_logits.numel() / x_start_logits.size(-1) / B) : 0;
  need_rows(x_start_logits, B, npos, K, "x_start logits");
  need_numel(t, B, "t");
  need_numel(grad_out, (int64_t)B * npos * K, "grad_out");
  Tensor gz = at::empty_like(x_start_logits);
  check_rc(d3pm_q_posterior_logits_backward(&tb, f32(x_start_logits, "x_start logits"), x_start_logits.size(-1),
                                            t.data_ptr(), tok_bytes(t, "t"), B, npos, f32(grad_out, "grad_out"),
                                            gz.data_ptr<float>(), stream()));
  return gz;
}

struct LossShape { int B; long npos; };
LossShape check_loss_args(const Tensor& logits, const Tensor& x0, const Tensor& x_t, const Tensor& t, int64_t K,
                          const OptTensor& grad_out) {
  auto [B, npos] = dims(x0);
  need_same_shape(x0, x_t, "x0 and x_t");
  need_numel(t, B, "t");
  need_rows(logits, B, npos, K, "logits");
  need_numel(grad_out, B, "grad_out");
  return {B, npos};
}

// vb_terms_bpd :462-489 + cross_entropy_start :527-542 + training_losses :558-582, and d/d logits in the same launch
std::tuple<Tensor, Tensor, Tensor, Tensor> hybrid_loss(at::TensorList tables, int64_t K, int64_t T, bool structured,
                                                       double eps, const Tensor& logits, const Tensor& x0,
                                                       const Tensor& x_t, const Tensor& t, int64_t loss_type,
                                                       double hybrid_coeff, double grad_scale, const OptTensor& grad_out,
                                                       bool want_grad) {
  D3PM_GUARD(logits);
  auto tb = image_tables(tables, K, T, structured, eps);
  auto [B, npos] = check_loss_args(logits, x0, x_t, t, K, grad_out);
  auto fo = logits.options().dtype(at::kFloat);
  Tensor loss = at::empty({B}, fo), vb = at::empty({B}, fo), ce = at::empty({B}, fo);
  Tensor dl = want_grad ? at::empty_like(logits) : at::empty({0}, fo);
  const size_t wsb = d3pm_hybrid_loss_workspace(B, npos, (int)K);
  Tensor ws = at::empty({(int64_t)(wsb / 4 + 1)}, fo);
  check_rc(d3pm_hybrid_loss(&tb, f32(logits, "logits"), logits.size(-1), x0.data_ptr(), tok_bytes(x0, "x0"), x_t.data_ptr(),
                            tok_bytes(x_t, "x_t"), t.data_ptr(), tok_bytes(t, "t"), B, npos, (int)loss_type,
                            (float)hybrid_coeff, (float)grad_scale, (const float*)opt_ptr(grad_out, at::kFloat, "grad_out"),
                            vb.data_ptr<float>(), ce.data_ptr<float>(), loss.data_ptr<float>(),
                            want_grad ? dl.data_ptr<float>() : nullptr, nullptr, nullptr, 0, ws.data_ptr(), wsb, stream()));
  return {loss, vb, ce, dl};
}

// the loss launch of a training step: per-example losses and d/d logits only, optionally the batch mean accumulated
// on the device (`loss_mean`, zeroed beforehand) and the Philox counter of q_sample_draw_t advanced by `rng_step`
std::tuple<Tensor, Tensor> hybrid_loss_step(at::TensorList tables, int64_t K, int64_t T, bool structured, double eps,
                                            const Tensor& logits, const Tensor& x0, const Tensor& x_t, const Tensor& t,
                                            int64_t loss_type, double hybrid_coeff, double grad_scale, bool want_grad,
                                            const OptTensor& loss_mean, const OptTensor& rng_counter, int64_t rng_step) {
  D3PM_GUARD(logits);
  auto tb = image_tables(tables, K, T, structured, eps);
  auto [B, npos] = check_loss_args(logits, x0, x_t, t, K, c10::nullopt);
  need_counter(loss_mean, "loss_mean");
  need_counter(rng_counter, "rng_counter");
  auto fo = logits.options().dtype(at::kFloat);
  Tensor loss = at::empty({B}, fo);
  Tensor dl = want_grad ? at::empty_like(logits) : at::empty({0}, fo);
  const size_t wsb = d3pm_hybrid_loss_workspace(B, npos, (int)K);
  Tensor ws = at::empty({(int64_t)(wsb / 4 + 1)}, fo);
  check_rc(d3pm_hybrid_loss(&tb, f32(logits, "logits"), logits.size(-1), x0.data_ptr(), tok_bytes(x0, "x0"), x_t.data_ptr(),
                            tok_bytes(x_t, "x_t"), t.data_ptr(), tok_bytes(t, "t"), B, npos, (int)loss_type,
                            (float)hybrid_coeff, (float)grad_scale, nullptr, nullptr, nullptr, loss.data_ptr<float>(),
                            want_grad ? dl.data_ptr<float>() : nullptr, (float*)opt_ptr(loss_mean, at::kFloat, "loss_mean"),
                            (uint64_t*)opt_ptr(rng_counter, at::kLong, "rng_counter"), (uint64_t)rng_step, ws.data_ptr(),
                            wsb, stream()));
  return {loss, dl};
}

// the same launch carrying the next batch's timestep draw and corruption (training_losses :549-556 of step i + 1):
// returns (losses, dlogits, next_t int32 [B], next_x_t int64)
std::tuple<Tensor, Tensor, Tensor, Tensor> hybrid_loss_then_q_sample(
    at::TensorList tables, int64_t K, int64_t T, bool structured, double eps, const Tensor& logits, const Tensor& x0,
    const Tensor& x_t, const Tensor& t, int64_t loss_type, double hybrid_coeff, double grad_scale, bool want_grad,
    const Tensor& next_x0, int64_t seed, int64_t offset, const OptTensor& offset_dev) {
  D3PM_GUARD(logits);
  auto tb = image_tables(tables, K, T, structured, eps);
  auto [B, npos] = check_loss_args(logits, x0, x_t, t, K, c10::nullopt);
  need_same_shape(next_x0, x0, "next_x0 and x0");
  need_counter(offset_dev, "offset_dev");
  auto fo = logits.options().dtype(at::kFloat);
  Tensor loss = at::empty({B}, fo);
  Tensor dl = want_grad ? at::empty_like(logits) : at::empty({0}, fo);
  Tensor nt = at::empty({B}, x0.options().dtype(at::kInt));
  Tensor nxt = at::empty(x0.sizes(), x0.options().dtype(at::kLong));
  const size_t wsb = d3pm_hybrid_loss_workspace(B, npos, (int)K);
  Tensor ws = at::empty({(int64_t)(wsb / 4 + 1)}, fo);
  check_rc(d3pm_hybrid_loss_then_q_sample(
      &tb, f32(logits, "logits"), logits.size(-1), x0.data_ptr(), tok_bytes(x0, "x0"), x_t.data_ptr(), tok_bytes(x_t, "x_t"),
      t.data_ptr(), tok_bytes(t, "t"), B, npos, (int)loss_type, (float)hybrid_coeff, (float)grad_scale, nullptr, nullptr,
      nullptr, loss.data_ptr<float>(), want_grad ? dl.data_ptr<float>() : nullptr, ws.data_ptr(), wsb, next_x0.data_ptr(),
      tok_bytes(next_x0, "next_x0"), (uint64_t)seed, (uint64_t)offset,
      (const uint64_t*)opt_ptr(offset_dev, at::kLong, "offset_dev"), nt.data_ptr<int32_t>(), nxt.data_ptr<int64_t>(),
      stream()));
  return {loss, dl, nt, nxt};
}

// hybrid_loss into buffers the caller owns, with the library's static scratch instead of a workspace: nothing is
// allocated by the call (callers under HIP graph capture with their own buffers, serving loops)
void hybrid_loss_static_(at::TensorList tables, int64_t K, int64_t T, bool structured, double eps, const Tensor& logits,
                         const Tensor& x0, const Tensor& x_t, const Tensor& t, int64_t loss_type, double hybrid_coeff,
                         double grad_scale, Tensor loss, const OptTensor& dlogits) {
  D3PM_GUARD(logits);
  auto tb = image_tables(tables, K, T, structured, eps);
  auto [B, npos] = check_loss_args(logits, x0, x_t, t, K, c10::nullopt);
  need_numel(loss, B, "loss");
  need_numel(dlogits, logits.numel(), "dlogits");
  check_rc(d3pm_hybrid_loss_static(&tb, f32(logits, "logits"), logits.size(-1), x0.data_ptr(), tok_bytes(x0, "x0"),
                                   x_t.data_ptr(), tok_bytes(x_t, "x_t"), t.data_ptr(), tok_bytes(t, "t"), B, npos,
                                   (int)loss_type, (float)hybrid_coeff, (float)grad_scale, nullptr, nullptr, nullptr,
                                   const_cast<float*>(f32(loss, "loss")), (float*)opt_ptr(dlogits, at::kFloat, "dlogits"),
                                   nullptr, nullptr, 0, stream()));
}

// backward of the loss nodes: g[b,:] *= upstream[b] / baked, in place
void scale_rows_(Tensor g, const Tensor& upstream, double baked) {
  D3PM_GUARD(g);
  const int B = (int)upstream.numel();
  TORCH_CHECK_VALUE(B == 0 || g.numel() % B == 0, "d3pm_mi355x: g has ", g.numel(), " elements, no multiple of the ", B,
                    " upstream gradients");
  check_rc(d3pm_scale_rows(const_cast<float*>(f32(g, "g")), f32(upstream, "upstream"), (float)baked, B,
                           B ? (long)(g.numel() / B) : 0, stream()));
}

// DiscreteDiffusion.p_sample, :384-403
std::tuple<Tensor, Tensor> p_sample(at::TensorList tables, int64_t K, int64_t T, bool structured, double eps,
                                    const Tensor& logits, const Tensor& x_t, const Tensor& t, const OptTensor& noise,
                                    int64_t noise_kind, int64_t seed, int64_t offset, const OptTensor& offset_dev,
                                    bool want_probs) {
  D3PM_GUARD(logits);
  auto tb = image_tables(tables, K, T, structured, eps);
  auto [B, npos] = dims(x_t);
  need_numel(t, B, "t");
  need_rows(logits, B, npos, K, "logits");
  need_numel(noise, (int64_t)B * npos * K, "noise");
  need_counter(offset_dev, "offset_dev");
  TORCH_CHECK_VALUE(!want_probs || logits.size(-1) == K, "d3pm_mi355x: probabilities are returned for rows of exactly K logits");
  Tensor sample = at::empty(x_t.sizes(), x_t.options().dtype(at::kLong));
  Tensor probs = want_probs ? at::empty_like(logits) : at::empty({0}, logits.options());
  check_rc(d3pm_p_sample(&tb, f32(logits, "logits"), logits.size(-1), x_t.data_ptr(), tok_bytes(x_t, "x_t"), t.data_ptr(),
                         tok_bytes(t, "t"), B, npos, (const float*)opt_ptr(noise, at::kFloat, "noise"), (int)noise_kind,
                         (uint64_t)seed, (uint64_t)offset, (const uint64_t*)opt_ptr(offset_dev, at::kLong, "offset_dev"),
                         sample.data_ptr<int64_t>(), want_probs ? probs.data_ptr<float>() : nullptr, stream()));
  return {sample, probs};
}

// DiscreteDiffusion.prior_bpd, :492-524
Tensor prior_bpd(at::TensorList tables, int64_t K, int64_t T, bool structured, double eps, const Tensor& x0, bool absorbing) {
  D3PM_GUARD(x0);
  auto tb = image_tables(tables, K, T, structured, eps);
  auto [B, npos] = dims(x0);
  auto fo = x0.options().dtype(at::kFloat);
  Tensor out = at::empty({B}, fo);
  const size_t wsb = d3pm_hybrid_loss_workspace(B, npos, (int)K);
  Tensor ws = at::empty({(int64_t)(wsb / 4 + 1)}, fo);
  check_rc(d3pm_prior_bpd(&tb, x0.data_ptr(), tok_bytes(x0, "x0"), B, npos, absorbing ? 1 : 0, out.data_ptr<float>(),
                          ws.data_ptr(), wsb, stream()));
  return out;
}

// _get_logits_from_logistic_pars, :288-309 (+ utils.log_min_exp, d3pm/utils.py:15-18)
Tensor logistic_logits(const Tensor& loc, const Tensor& log_scale, const Tensor& centers, double eps) {
  D3PM_GUARD(loc);
  need_same_shape(loc, log_scale, "loc and log_scale");
  const int64_t K = centers.numel();
  TORCH_CHECK_VALUE(K >= 2, "d3pm_mi355x: at least two bin centres");
  Tensor out = at::empty(with_k(loc, K), loc.options().dtype(at::kFloat));
  check_rc(d3pm_logistic_logits(f32(loc, "loc"), f32(log_scale, "log_scale"), f32(centers, "centers"), (long)loc.numel(),
                                (int)K, (float)eps, out.data_ptr<float>(), stream()));
  return out;
}

// its gradient to (loc, log_scale)
std::tuple<Tensor, Tensor> logistic_logits_backward(const Tensor& loc, const Tensor& log_scale, const Tensor& centers,
                                                    const Tensor& dlogits, double eps) {
  D3PM_GUARD(loc);
  need_same_shape(loc, log_scale, "loc and log_scale");
  const int64_t K = centers.numel();
  need_numel(dlogits, loc.numel() * K, "dlogits");
  Tensor dloc = at::empty_like(loc), dls = at::empty_like(log_scale);
  check_rc(d3pm_logistic_logits_backward(f32(loc, "loc"), f32(log_scale, "log_scale"), f32(centers, "centers"),
                                         f32(dlogits, "dlogits"), (long)loc.numel(), (int)K, (float)eps,
                                         dloc.data_ptr<float>(), dls.data_ptr<float>(), stream()));
  return {dloc, dls};
}

// ---- transition tables
const double* f64(const Tensor& t, const char* what) {
  TORCH_CHECK(t.is_cuda(), "d3pm_mi355x: ", what, " must live on an MI355X (device='cuda'); there is no CPU path");
  TORCH_CHECK(t.scalar_type() == at::kDouble, "d3pm_mi355x: ", what, " must be float64");
  return t.data_ptr<double>();
}
void need_table(const Tensor& t, const char* what) {
  TORCH_CHECK_VALUE(t.dim() == 3 && t.size(1) == t.size(2), "d3pm_mi355x: ", what, " must be [T, K, K]");
}

// DiscreteDiffusion.__init__ table construction, :56-99: (one-step Q_t [T,K,K], q_mats [T,K,K]) float64
std::tuple<Tensor, Tensor> build_tables(const Tensor& betas, int64_t kind, int64_t K, int64_t bands) {
  D3PM_GUARD(betas);
  TORCH_CHECK(betas.is_contiguous(), "d3pm_mi355x: betas must be contiguous");
  const int64_t T = betas.numel();
  Tensor one = at::empty({T, K, K}, betas.options().dtype(at::kDouble)), bar = at::empty_like(one);
  check_rc(d3pm_build_tables_f64((int)kind, (int)K, (int)T, (int)bands, f64(betas, "betas"), one.data_ptr<double>(),
                                 bar.data_ptr<double>(), stream()));
  return {one, bar};
}

// Tokenizer.q_uniform_schedule + cumprod_matrix, protein_generation/utils.py:65-83, :7-18: (Q_t, Q_bar) float64
std::tuple<Tensor, Tensor> protein_uniform_tables(const Tensor& off, const Tensor& norm, int64_t K) {
  D3PM_GUARD(off);
  need_same_shape(off, norm, "off and norm");
  const int64_t T = off.numel();
  Tensor one = at::empty({T, K, K}, off.options().dtype(at::kDouble)), bar = at::empty_like(one);
  check_rc(d3pm_protein_uniform_tables_f64((int)K, (int)T, f32(off, "off"), f32(norm, "norm"), one.data_ptr<double>(),
                                           bar.data_ptr<double>(), stream()));
  return {one, bar};
}

// cumprod_matrix, protein_generation/utils.py:7-18, on a (possibly strided) [T,K,K] float64 tensor
Tensor cumprod(const Tensor& mats) {
  D3PM_GUARD(mats);
  need_table(mats, "mats");
  Tensor out = at::empty(mats.sizes(), mats.options().memory_format(at::MemoryFormat::Contiguous));
  check_rc(d3pm_cumprod_f64((int)mats.size(1), (int)mats.size(0), f64(mats, "mats"), mats.stride(0), mats.stride(1),
                            mats.stride(2), out.data_ptr<double>(), stream()));
  return out;
}

// column-structure digests of a (possibly strided) float64 table: (digest, violations int32 [1])
std::tuple<Tensor, Tensor> digest_f32(const Tensor& table, double eps) {
  D3PM_GUARD(table);
  need_table(table, "table");
  const int64_t T = table.size(0), K = table.size(1);
  Tensor dg = at::empty({T, 8, K}, table.options().dtype(at::kFloat));
  Tensor viol = at::zeros({1}, table.options().dtype(at::kInt));
  check_rc(d3pm_digest_f32(f64(table, "table"), table.stride(0), table.stride(1), table.stride(2), (int)T, (int)K,
                           (float)eps, dg.data_ptr<float>(), viol.data_ptr<int>(), stream()));
  return {dg, viol};
}
std::tuple<Tensor, Tensor> digest_f64(const Tensor& table, double rtol) {
  D3PM_GUARD(table);
  need_table(table, "table");
  const int64_t T = table.size(0), K = table.size(1);
  Tensor dg = at::empty({T, 4, K}, table.options().dtype(at::kDouble));
  Tensor viol = at::zeros({1}, table.options().dtype(at::kInt));
  check_rc(d3pm_digest_f64(f64(table, "table"), table.stride(0), table.stride(1), table.stride(2), (int)T, (int)K, rtol,
                           dg.data_ptr<double>(), viol.data_ptr<int>(), stream()));
  return {dg, viol};
}
// per-timestep float32 constants of the training form of the LVB loss, from the two float64 digests
Tensor lvb_digest(const Tensor& qbar_digest, const Tensor& qone_digest, int64_t K, int64_t T) {
  D3PM_GUARD(qbar_digest);
  auto tb = protein_tables(qbar_digest, qone_digest, Tensor(), K, T);
  Tensor out = at::empty({T, (int64_t)(d3pm_lvb_digest_f32_size(1) / (32 * sizeof(float))), 32},
                         qbar_digest.options().dtype(at::kFloat));
  check_rc(d3pm_lvb_digest_f32(&tb, out.data_ptr<float>(), stream()));
  return out;
}

// content fingerprints of several same-device float64 tables in one tensor (int64 [n]): the cache key of callers
// that are handed fresh copies of their tables on every call (notebook cell-12)
Tensor table_fingerprints(at::TensorList tables) {
  TORCH_CHECK(tables.size() >= 1, "d3pm_mi355x: at least one table");
  D3PM_GUARD(tables[0]);
  Tensor out = at::zeros({(int64_t)tables.size()}, tables[0].options().dtype(at::kLong));
  for (size_t i = 0; i < tables.size(); ++i) {
    const Tensor& tb = tables[i];
    need_table(tb, "table");
    TORCH_CHECK(tb.device() == tables[0].device(), "d3pm_mi355x: tables on different devices");
    check_rc(d3pm_table_fingerprint(f64(tb, "table"), tb.stride(0), tb.stride(1), tb.stride(2), (int)tb.size(0),
                                    (int)tb.size(1), (uint64_t*)(out.data_ptr<int64_t>() + i), stream()));
  }
  return out;
}

// the float32 cast of _at / _at_onehot (:218, :242), done once, into a contiguous [T,K,K]
Tensor table_to_f32(const Tensor& table) {
  D3PM_GUARD(table);
  need_table(table, "table");
  Tensor out = at::empty(table.sizes(), table.options().dtype(at::kFloat).memory_format(at::MemoryFormat::Contiguous));
  check_rc(d3pm_table_to_f32(f64(table, "table"), table.stride(0), table.stride(1), table.stride(2), (int)table.size(0),
                             (int)table.size(1), out.data_ptr<float>(), stream()));
  return out;
}

// a contiguous float32 [T,K,K] table of probabilities as two float16 planes (hi, lo of 2^14 x the entry), laid out
// [T, K / 32, K / 16, 2, 64, 8] as the operand image of the float16 matrix instructions: the operand form of
// _at_onehot's matmul (:242-247)
Tensor table_split_f16x2(const Tensor& table) {
  D3PM_GUARD(table);
  TORCH_CHECK_VALUE(table.dim() == 3 && table.size(1) == table.size(2), "d3pm_mi355x: table must be [T, K, K]");
  const int64_t T = table.size(0), K = table.size(1);
  TORCH_CHECK_VALUE(K % 32 == 0, "d3pm_mi355x: the operand image needs K to be a multiple of 32");
  Tensor out = at::empty({T, K / 32, K / 16, 2, 64, 8}, table.options().dtype(at::kHalf));
  check_rc(d3pm_table_split_f16x2(f32(table, "table"), (int)T, (int)K, (uint16_t*)out.data_ptr(), stream()));
  return out;
}

struct ProtShape { int B; long npos; };
ProtShape check_prot_args(const Tensor& pred, const Tensor* x0, const Tensor& x_t, const Tensor& t, int64_t K) {
  auto [B, npos] = dims(x_t);
  if (x0) need_same_shape(*x0, x_t, "x0 and x_t");
  need_numel(t, B, "t");
  need_rows(pred, B, npos, K, "pred");
  return {B, npos};
}

// D3PMLVBLoss.forward / D3PMLVBLossMSA.forward per batch element, protein_generation/losses.py:48-155
std::tuple<Tensor, Tensor> lvb_loss(const Tensor& qbar_digest, const Tensor& qone_digest, const Tensor& lvb_digest, int64_t K, int64_t T,
                                    const Tensor& pred, const Tensor& x0, const Tensor& x_t, const Tensor& t,
                                    const OptTensor& q, int64_t tmax, int64_t msa, double grad_scale, bool want_grad) {
  D3PM_GUARD(pred);
  auto tb = protein_tables(qbar_digest, qone_digest, lvb_digest, K, T);
  auto [B, npos] = check_prot_args(pred, &x0, x_t, t, K);
  need_numel(q, (int64_t)B * npos * K, "q");
  TORCH_CHECK(tok_bytes(x0, "x0") == 8 && tok_bytes(x_t, "x_t") == 8 && tok_bytes(t, "t") == 8,
              "d3pm_mi355x: protein-style tokens and timesteps are int64");
  Tensor per = at::empty({B}, pred.options().dtype(at::kDouble));
  Tensor dp = want_grad ? at::empty_like(pred) : at::empty({0}, pred.options());
  const size_t wsb = d3pm_lvb_loss_workspace(B, npos, (int)K);
  Tensor ws = at::empty({(int64_t)(wsb / 8 + 1)}, pred.options().dtype(at::kDouble));
  check_rc(d3pm_lvb_loss(&tb, f32(pred, "pred"), pred.size(-1), x0.data_ptr<int64_t>(), x_t.data_ptr<int64_t>(),
                         t.data_ptr<int64_t>(), (const double*)opt_ptr(q, at::kDouble, "q"), B, npos, (int)tmax, (int)msa,
                         grad_scale, per.data_ptr<double>(), want_grad ? dp.data_ptr<float>() : nullptr, ws.data_ptr(),
                         wsb, stream()));
  return {per, dp};
}

// the same with the driver's cross-entropy term (Protein Generation Notebook cell-12: (lvb + lambda ce) n_tokens):
// returns (per_seq, ce_seq, dpred)
std::tuple<Tensor, Tensor, Tensor> lvb_hybrid_loss(const Tensor& qbar_digest, const Tensor& qone_digest, const Tensor& lvb_digest, int64_t K, int64_t T,
                                                   const Tensor& pred, const Tensor& x0, const Tensor& x_t, const Tensor& t,
                                                   const OptTensor& q, int64_t tmax, int64_t msa, double grad_scale,
                                                   double ce_grad_scale, bool want_grad) {
  D3PM_GUARD(pred);
  auto tb = protein_tables(qbar_digest, qone_digest, lvb_digest, K, T);
  auto [B, npos] = check_prot_args(pred, &x0, x_t, t, K);
  need_numel(q, (int64_t)B * npos * K, "q");
  TORCH_CHECK(tok_bytes(x0, "x0") == 8 && tok_bytes(x_t, "x_t") == 8 && tok_bytes(t, "t") == 8,
              "d3pm_mi355x: protein-style tokens and timesteps are int64");
  auto dopt = pred.options().dtype(at::kDouble);
  Tensor per = at::empty({B}, dopt), ce = at::empty({B}, dopt);
  Tensor dp = want_grad ? at::empty_like(pred) : at::empty({0}, pred.options());
  const size_t wsb = d3pm_lvb_loss_workspace(B, npos, (int)K);
  Tensor ws = at::empty({(int64_t)(wsb / 8 + 1)}, dopt);
  check_rc(d3pm_lvb_hybrid_loss(&tb, f32(pred, "pred"), pred.size(-1), x0.data_ptr<int64_t>(), x_t.data_ptr<int64_t>(),
                                t.data_ptr<int64_t>(), (const double*)opt_ptr(q, at::kDouble, "q"), B, npos, (int)tmax,
                                (int)msa, grad_scale, ce_grad_scale, per.data_ptr<double>(), ce.data_ptr<double>(),
                                want_grad ? dp.data_ptr<float>() : nullptr, ws.data_ptr(), wsb, stream()));
  return {per, ce, dp};
}

// reverse step of generate_d3pm / generate_msa_d3pm, protein_generation/generate.py:28-40, :70-93
Tensor lvb_reverse_step(const Tensor& qbar_digest, const Tensor& qone_digest, const Tensor& lvb_digest, int64_t K, int64_t T, const Tensor& pred,
                        const Tensor& x_t, const Tensor& t, const OptTensor& expo, int64_t seed, int64_t offset,
                        const OptTensor& offset_dev, double last_class_div, int64_t exclude_last_prefix) {
  D3PM_GUARD(pred);
  auto tb = protein_tables(qbar_digest, qone_digest, lvb_digest, K, T);
  auto [B, npos] = check_prot_args(pred, nullptr, x_t, t, K);
  need_numel(expo, (int64_t)B * npos * K, "expo");
  need_counter(offset_dev, "offset_dev");
  TORCH_CHECK(tok_bytes(x_t, "x_t") == 8 && tok_bytes(t, "t") == 8, "d3pm_mi355x: protein-style tokens are int64");
  Tensor out = at::empty_like(x_t);
  check_rc(d3pm_lvb_reverse_step(&tb, f32(pred, "pred"), pred.size(-1), x_t.data_ptr<int64_t>(), t.data_ptr<int64_t>(), B,
                                 npos, (const double*)opt_ptr(expo, at::kDouble, "expo"), (uint64_t)seed, (uint64_t)offset,
                                 (const uint64_t*)opt_ptr(offset_dev, at::kLong, "offset_dev"), last_class_div,
                                 (long)exclude_last_prefix, out.data_ptr<int64_t>(), stream()));
  return out;
}

// batched sample_transition_matrix as the collaters use it, protein_generation/utils.py:44-52, dataset.py:234-242
std::tuple<Tensor, Tensor> corrupt(const Tensor& qbar_digest, const Tensor& qone_digest, const Tensor& lvb_digest, int64_t K, int64_t T,
                                   const Tensor& x0, const Tensor& t, const OptTensor& expo, int64_t seed, int64_t offset,
                                   bool want_probs) {
  D3PM_GUARD(x0);
  auto tb = protein_tables(qbar_digest, qone_digest, lvb_digest, K, T);
  auto [B, npos] = dims(x0);
  need_numel(t, B, "t");
  need_numel(expo, (int64_t)B * npos * K, "expo");
  TORCH_CHECK(tok_bytes(x0, "x0") == 8 && tok_bytes(t, "t") == 8, "d3pm_mi355x: protein-style tokens are int64");
  Tensor x_t = at::empty_like(x0);
  Tensor probs = want_probs ? at::empty(with_k(x0, K), x0.options().dtype(at::kDouble))
                            : at::empty({0}, x0.options().dtype(at::kDouble));
  check_rc(d3pm_corrupt(&tb, x0.data_ptr<int64_t>(), t.data_ptr<int64_t>(), B, npos,
                        (const double*)opt_ptr(expo, at::kDouble, "expo"), (uint64_t)seed, (uint64_t)offset,
                        x_t.data_ptr<int64_t>(), want_probs ? probs.data_ptr<double>() : nullptr, stream()));
  return {x_t, probs};
}

}  // namespace

#define D3PM_TB "Tensor[] tables, int K, int T, bool structured, float eps, "
#define D3PM_PT "Tensor qbar_digest, Tensor qone_digest, Tensor lvb_digest, int K, int T, "
#define D3PM_LOSS_ARGS "Tensor logits, Tensor x0, Tensor x_t, Tensor t, int loss_type, float hybrid_coeff, float grad_scale, "

TORCH_LIBRARY(d3pm_mi355x, m) {
  m.def("q_probs(" D3PM_TB "Tensor x0, Tensor t) -> Tensor");
  m.def("q_sample(" D3PM_TB "Tensor x0, Tensor t, Tensor? noise, int noise_kind, int seed, int offset, "
        "Tensor? offset_dev) -> Tensor");
  m.def("q_sample_draw_t(" D3PM_TB "Tensor x0, int seed, int offset, Tensor? offset_dev, Tensor(a!)? clear) -> (Tensor, Tensor)");
  m.def("q_posterior_logits(" D3PM_TB "Tensor x_start, bool x_start_is_logits, Tensor x_t, Tensor t) -> Tensor");
  m.def("q_posterior_logits_backward(" D3PM_TB "Tensor x_start_logits, Tensor t, Tensor grad_out) -> Tensor");
  m.def("hybrid_loss(" D3PM_TB D3PM_LOSS_ARGS "Tensor? grad_out, bool want_grad) -> (Tensor, Tensor, Tensor, Tensor)");
  m.def("hybrid_loss_step(" D3PM_TB D3PM_LOSS_ARGS "bool want_grad, Tensor(a!)? loss_mean, Tensor(b!)? rng_counter, "
        "int rng_step) -> (Tensor, Tensor)");
  m.def("hybrid_loss_then_q_sample(" D3PM_TB D3PM_LOSS_ARGS "bool want_grad, Tensor next_x0, int seed, int offset, "
        "Tensor? offset_dev) -> (Tensor, Tensor, Tensor, Tensor)");
  m.def("hybrid_loss_static_(" D3PM_TB D3PM_LOSS_ARGS "Tensor(a!) loss, Tensor(b!)? dlogits) -> ()");
  m.def("scale_rows_(Tensor(a!) g, Tensor upstream, float baked) -> ()");
  m.def("p_sample(" D3PM_TB "Tensor logits, Tensor x_t, Tensor t, Tensor? noise, int noise_kind, int seed, int offset, "
        "Tensor? offset_dev, bool want_probs) -> (Tensor, Tensor)");
  m.def("prior_bpd(" D3PM_TB "Tensor x0, bool absorbing) -> Tensor");
  m.def("logistic_logits(Tensor loc, Tensor log_scale, Tensor centers, float eps) -> Tensor");
  m.def("logistic_logits_backward(Tensor loc, Tensor log_scale, Tensor centers, Tensor dlogits, float eps) -> (Tensor, Tensor)");
  m.def("build_tables(Tensor betas, int kind, int K, int bands) -> (Tensor, Tensor)");
  m.def("protein_uniform_tables(Tensor off, Tensor norm, int K) -> (Tensor, Tensor)");
  m.def("cumprod(Tensor mats) -> Tensor");
  m.def("digest_f32(Tensor table, float eps) -> (Tensor, Tensor)");
  m.def("digest_f64(Tensor table, float rtol) -> (Tensor, Tensor)");
  m.def("table_to_f32(Tensor table) -> Tensor");
  m.def("table_split_f16x2(Tensor table) -> Tensor");
  m.def("table_fingerprints(Tensor[] tables) -> Tensor");
  m.def("lvb_digest(Tensor qbar_digest, Tensor qone_digest, int K, int T) -> Tensor");
  m.def("lvb_loss(" D3PM_PT "Tensor pred, Tensor x0, Tensor x_t, Tensor t, Tensor? q, int tmax, int msa, "
        "float grad_scale, bool want_grad) -> (Tensor, Tensor)");
  m.def("lvb_hybrid_loss(" D3PM_PT "Tensor pred, Tensor x0, Tensor x_t, Tensor t, Tensor? q, int tmax, int msa, "
        "float grad_scale, float ce_grad_scale, bool want_grad) -> (Tensor, Tensor, Tensor)");
  m.def("lvb_reverse_step(" D3PM_PT "Tensor pred, Tensor x_t, Tensor t, Tensor? expo, int seed, int offset, "
        "Tensor? offset_dev, float last_class_div, int exclude_last_prefix) -> Tensor");
  m.def("corrupt(" D3PM_PT "Tensor x0, Tensor t, Tensor? expo, int seed, int offset, bool want_probs) -> (Tensor, Tensor)");
}

TORCH_LIBRARY_IMPL(d3pm_mi355x, CUDA, m) {
  m.impl("q_probs", q_probs);
  m.impl("q_sample", q_sample);
  m.impl("q_sample_draw_t", q_sample_draw_t);
  m.impl("q_posterior_logits", q_posterior_logits);
  m.impl("q_posterior_logits_backward", q_posterior_logits_backward);
  m.impl("hybrid_loss", hybrid_loss);
  m.impl("hybrid_loss_step", hybrid_loss_step);
  m.impl("hybrid_loss_then_q_sample", hybrid_loss_then_q_sample);
  m.impl("hybrid_loss_static_", hybrid_loss_static_);
  m.impl("scale_rows_", scale_rows_);
  m.impl("p_sample", p_sample);
  m.impl("prior_bpd", prior_bpd);
  m.impl("logistic_logits", logistic_logits);
  m.impl("logistic_logits_backward", logistic_logits_backward);
  m.impl("build_tables", build_tables);
  m.impl("protein_uniform_tables", protein_uniform_tables);
  m.impl("cumprod", cumprod);
  m.impl("digest_f32", digest_f32);
  m.impl("digest_f64", digest_f64);
  m.impl("table_to_f32", table_to_f32);
  m.impl("table_split_f16x2", table_split_f16x2);
  m.impl("table_fingerprints", table_fingerprints);
  m.impl("lvb_digest", lvb_digest);
  m.impl("lvb_loss", lvb_loss);
  m.impl("lvb_hybrid_loss", lvb_hybrid_loss);
  m.impl("lvb_reverse_step", lvb_reverse_step);
  m.impl("corrupt", corrupt);
}
