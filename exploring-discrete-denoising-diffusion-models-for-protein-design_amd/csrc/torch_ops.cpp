// torch.ops.d3pm_mi355x.* -- the PyTorch-ROCm extension form of the boundary.
//
// A thin shim: every operator checks its tensors, allocates its outputs (and workspace) with torch's
// allocator, and calls the C ABI of libd3pm_hip.so (include/d3pm_hip.h) on torch's current HIP stream.  No
// arithmetic happens here.  The operators carry the reference function they stand for in their schema comment;
// autograd nodes live in the Python facade, which calls these operators.  Status codes of the library become
// the exceptions the reference raises: D3PM_EINVAL -> ValueError, D3PM_EUNSUPPORTED -> NotImplementedError,
// D3PM_EHIP -> RuntimeError.  The caller makes the tensors' device current (the facade does).
#include <ATen/ATen.h>
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

// tables = [qbar_digest, qone_digest, qbar, qoneT, qbarT]; the last three may be empty tensors (structured tables)
d3pm_tables_f32 image_tables(at::TensorList tables, int64_t K, int64_t T, bool structured, double eps) {
  TORCH_CHECK(tables.size() == 5, "d3pm_mi355x: tables = [qbar_digest, qone_digest, qbar, qoneT, qbarT]");
  d3pm_tables_f32 tb{};
  tb.K = (int)K; tb.T = (int)T; tb.structured = structured ? 1 : 0; tb.eps = (float)eps;
  auto get = [&](int i, const char* what) -> const float* {
    return tables[i].defined() && tables[i].numel() > 0 ? f32(tables[i], what) : nullptr;
  };
  tb.qbar_digest = get(0, "qbar_digest"); tb.qone_digest = get(1, "qone_digest");
  tb.qbar = get(2, "qbar"); tb.qoneT = get(3, "qoneT"); tb.qbarT = get(4, "qbarT");
  return tb;
}
d3pm_tables_f64 protein_tables(const Tensor& qbar_digest, const Tensor& qone_digest, int64_t K, int64_t T) {
  need_device(qbar_digest, "qbar_digest"); need_device(qone_digest, "qone_digest");
  TORCH_CHECK(qbar_digest.scalar_type() == at::kDouble && qone_digest.scalar_type() == at::kDouble,
              "d3pm_mi355x: protein digests are float64");
  return d3pm_tables_f64{(int)K, (int)T, qbar_digest.data_ptr<double>(), qone_digest.data_ptr<double>()};
}
std::pair<int, long> dims(const Tensor& tokens) {
  const int B = tokens.dim() ? (int)tokens.size(0) : 1;
  return {B, B ? (long)(tokens.numel() / B) : 0};
}
std::vector<int64_t> with_k(const Tensor& tokens, int64_t K) {
  auto s = tokens.sizes().vec();
  s.push_back(K);
  return s;
}

// DiscreteDiffusion.q_probs, d3pm/diffusion_discrete.py:250-263
Tensor q_probs(at::TensorList tables, int64_t K, int64_t T, bool structured, double eps, const Tensor& x0, const Tensor& t) {
  auto tb = image_tables(tables, K, T, structured, eps);
  auto [B, npos] = dims(x0);
  Tensor out = at::empty(with_k(x0, K), x0.options().dtype(at::kFloat));
  check_rc(d3pm_q_probs(&tb, x0.data_ptr(), tok_bytes(x0, "x0"), t.data_ptr(), tok_bytes(t, "t"), B, npos,
                        out.data_ptr<float>(), stream()));
  return out;
}

// DiscreteDiffusion.q_sample, :266-285
Tensor q_sample(at::TensorList tables, int64_t K, int64_t T, bool structured, double eps, const Tensor& x0, const Tensor& t,
                const OptTensor& noise, int64_t noise_kind, int64_t seed, int64_t offset, const OptTensor& offset_dev) {
  auto tb = image_tables(tables, K, T, structured, eps);
  auto [B, npos] = dims(x0);
  Tensor out = at::empty(x0.sizes(), x0.options().dtype(at::kLong));
  check_rc(d3pm_q_sample(&tb, x0.data_ptr(), tok_bytes(x0, "x0"), t.data_ptr(), tok_bytes(t, "t"), B, npos,
                         (const float*)opt_ptr(noise, at::kFloat, "noise"), (int)noise_kind, (uint64_t)seed,
                         (uint64_t)offset, (const uint64_t*)opt_ptr(offset_dev, at::kLong, "offset_dev"),
                         out.data_ptr<int64_t>(), stream()));
  return out;
}

// DiscreteDiffusion.q_posterior_logits, :312-336 (and p_logits :339-381 with x_start_is_logits)
Tensor q_posterior_logits(at::TensorList tables, int64_t K, int64_t T, bool structured, double eps, const Tensor& x_start,
                          bool x_start_is_logits, const Tensor& x_t, const Tensor& t) {
  auto tb = image_tables(tables, K, T, structured, eps);
  auto [B, npos] = dims(x_t);
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
  auto tb = image_tables(tables, K, T, structured, eps);
  const int B = (int)x_start_logits.size(0);
  const long npos = B ? (long)(x_start_logits.numel() / x_start_logits.size(-1) / B) : 0;
  Tensor gz = at::empty_like(x_start_logits);
  check_rc(d3pm_q_posterior_logits_backward(&tb, f32(x_start_logits, "x_start logits"), x_start_logits.size(-1),
                                            t.data_ptr(), tok_bytes(t, "t"), B, npos, f32(grad_out, "grad_out"),
                                            gz.data_ptr<float>(), stream()));
  return gz;
}

// vb_terms_bpd :462-489 + cross_entropy_start :527-542 + training_losses :558-582, and d/d logits in the same launch
std::tuple<Tensor, Tensor, Tensor, Tensor> hybrid_loss(at::TensorList tables, int64_t K, int64_t T, bool structured,
                                                       double eps, const Tensor& logits, const Tensor& x0,
                                                       const Tensor& x_t, const Tensor& t, int64_t loss_type,
                                                       double hybrid_coeff, double grad_scale, const OptTensor& grad_out,
                                                       bool want_grad) {
  auto tb = image_tables(tables, K, T, structured, eps);
  auto [B, npos] = dims(x0);
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

// backward of the loss nodes: g[b,:] *= upstream[b] / baked, in place
void scale_rows_(Tensor g, const Tensor& upstream, double baked) {
  const int B = (int)upstream.numel();
  check_rc(d3pm_scale_rows(const_cast<float*>(f32(g, "g")), f32(upstream, "upstream"), (float)baked, B,
                           B ? (long)(g.numel() / B) : 0, stream()));
}

// DiscreteDiffusion.p_sample, :384-403
std::tuple<Tensor, Tensor> p_sample(at::TensorList tables, int64_t K, int64_t T, bool structured, double eps,
                                    const Tensor& logits, const Tensor& x_t, const Tensor& t, const OptTensor& noise,
                                    int64_t noise_kind, int64_t seed, int64_t offset, const OptTensor& offset_dev,
                                    bool want_probs) {
  auto tb = image_tables(tables, K, T, structured, eps);
  auto [B, npos] = dims(x_t);
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

// D3PMLVBLoss.forward / D3PMLVBLossMSA.forward per batch element, protein_generation/losses.py:48-155
std::tuple<Tensor, Tensor> lvb_loss(const Tensor& qbar_digest, const Tensor& qone_digest, int64_t K, int64_t T,
                                    const Tensor& pred, const Tensor& x0, const Tensor& x_t, const Tensor& t,
                                    const OptTensor& q, int64_t tmax, int64_t msa, double grad_scale, bool want_grad) {
  auto tb = protein_tables(qbar_digest, qone_digest, K, T);
  auto [B, npos] = dims(x0);
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
std::tuple<Tensor, Tensor, Tensor> lvb_hybrid_loss(const Tensor& qbar_digest, const Tensor& qone_digest, int64_t K, int64_t T,
                                                   const Tensor& pred, const Tensor& x0, const Tensor& x_t, const Tensor& t,
                                                   const OptTensor& q, int64_t tmax, int64_t msa, double grad_scale,
                                                   double ce_grad_scale, bool want_grad) {
  auto tb = protein_tables(qbar_digest, qone_digest, K, T);
  auto [B, npos] = dims(x0);
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
Tensor lvb_reverse_step(const Tensor& qbar_digest, const Tensor& qone_digest, int64_t K, int64_t T, const Tensor& pred,
                        const Tensor& x_t, const Tensor& t, const OptTensor& expo, int64_t seed, int64_t offset,
                        const OptTensor& offset_dev, double last_class_div, int64_t exclude_last_prefix) {
  auto tb = protein_tables(qbar_digest, qone_digest, K, T);
  auto [B, npos] = dims(x_t);
  TORCH_CHECK(tok_bytes(x_t, "x_t") == 8 && tok_bytes(t, "t") == 8, "d3pm_mi355x: protein-style tokens are int64");
  Tensor out = at::empty_like(x_t);
  check_rc(d3pm_lvb_reverse_step(&tb, f32(pred, "pred"), pred.size(-1), x_t.data_ptr<int64_t>(), t.data_ptr<int64_t>(), B,
                                 npos, (const double*)opt_ptr(expo, at::kDouble, "expo"), (uint64_t)seed, (uint64_t)offset,
                                 (const uint64_t*)opt_ptr(offset_dev, at::kLong, "offset_dev"), last_class_div,
                                 (long)exclude_last_prefix, out.data_ptr<int64_t>(), stream()));
  return out;
}

// batched sample_transition_matrix as the collaters use it, protein_generation/utils.py:44-52, dataset.py:234-242
std::tuple<Tensor, Tensor> corrupt(const Tensor& qbar_digest, const Tensor& qone_digest, int64_t K, int64_t T,
                                   const Tensor& x0, const Tensor& t, const OptTensor& expo, int64_t seed, int64_t offset,
                                   bool want_probs) {
  auto tb = protein_tables(qbar_digest, qone_digest, K, T);
  auto [B, npos] = dims(x0);
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
#define D3PM_PT "Tensor qbar_digest, Tensor qone_digest, int K, int T, "

TORCH_LIBRARY(d3pm_mi355x, m) {
  m.def("q_probs(" D3PM_TB "Tensor x0, Tensor t) -> Tensor");
  m.def("q_sample(" D3PM_TB "Tensor x0, Tensor t, Tensor? noise, int noise_kind, int seed, int offset, "
        "Tensor? offset_dev) -> Tensor");
  m.def("q_posterior_logits(" D3PM_TB "Tensor x_start, bool x_start_is_logits, Tensor x_t, Tensor t) -> Tensor");
  m.def("q_posterior_logits_backward(" D3PM_TB "Tensor x_start_logits, Tensor t, Tensor grad_out) -> Tensor");
  m.def("hybrid_loss(" D3PM_TB "Tensor logits, Tensor x0, Tensor x_t, Tensor t, int loss_type, float hybrid_coeff, "
        "float grad_scale, Tensor? grad_out, bool want_grad) -> (Tensor, Tensor, Tensor, Tensor)");
  m.def("scale_rows_(Tensor(a!) g, Tensor upstream, float baked) -> ()");
  m.def("p_sample(" D3PM_TB "Tensor logits, Tensor x_t, Tensor t, Tensor? noise, int noise_kind, int seed, int offset, "
        "Tensor? offset_dev, bool want_probs) -> (Tensor, Tensor)");
  m.def("prior_bpd(" D3PM_TB "Tensor x0, bool absorbing) -> Tensor");
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
  m.impl("q_posterior_logits", q_posterior_logits);
  m.impl("q_posterior_logits_backward", q_posterior_logits_backward);
  m.impl("hybrid_loss", hybrid_loss);
  m.impl("scale_rows_", scale_rows_);
  m.impl("p_sample", p_sample);
  m.impl("prior_bpd", prior_bpd);
  m.impl("lvb_loss", lvb_loss);
  m.impl("lvb_hybrid_loss", lvb_hybrid_loss);
  m.impl("lvb_reverse_step", lvb_reverse_step);
  m.impl("corrupt", corrupt);
}
