/*
 * d3pm_hip.h - C ABI of libd3pm_hip.so: the D3PM categorical-diffusion core for
 * AMD MI355X (gfx950).
 *
 * The reference project is pure Python/PyTorch and has no FFI of its own: the
 * boundary this library replaces is the Python call surface of its D3PM core.
 * Every entry point below names the reference function(s) whose arithmetic it
 * carries out (paths relative to the reference tree).  All pointers except
 * `d3pm_*_desc` structs, and the structs' own storage, are DEVICE pointers on
 * the current HIP device; `stream` is a hipStream_t passed as void*.  Nothing
 * here synchronises the host with the device, allocates or frees.
 *
 * Token arrays are int32 or int64 (`tok_bytes` = 4 or 8), per-sample
 * timesteps likewise (`t_bytes`).  Logits rows may be wider than K
 * (`ld` = row stride in elements >= K); only the first K entries take part
 * unless stated otherwise.  "npos" is the number of positions per batch
 * element (H*W*C for images, L for sequences, D*L for MSAs); tensors are laid
 * out [B, npos, K] row-major.
 *
 * Accepted ranges.  Tokens: [0, K-1].  Image-style timesteps (d3pm_q_*,
 * d3pm_hybrid_loss, d3pm_p_sample, d3pm_prior_bpd): [0, T-1].  Protein-style
 * timesteps: [0, T-1] for d3pm_corrupt; [1, T-1] for d3pm_lvb_reverse_step and
 * d3pm_lvb_loss, which also takes t == tmax (that branch reads no table).
 * The host cannot see these values without a synchronisation, so the kernels
 * do not report violations: a value outside its range (64-bit values are
 * read as their low 32 bits, signed) is replaced by the nearest valid one
 * before it indexes anything (the reference would raise an
 * index error, or wrap to the last matrix for t - 1 == -1), and the result is
 * that of the clamped input.  No value of a token or timestep makes a kernel
 * read or write outside the tables and tensors it was given.
 *
 * Return value: D3PM_OK or an error code; d3pm_last_error() gives the message
 * of the calling thread's last failure.
 */
#ifndef D3PM_HIP_H
#define D3PM_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define D3PM_OK            0
#define D3PM_EINVAL        1   /* bad argument: the facade raises ValueError/AssertionError */
#define D3PM_EUNSUPPORTED  2   /* valid but not implemented: NotImplementedError */
#define D3PM_EHIP          3   /* HIP runtime failure: RuntimeError */

#define D3PM_KIND_UNIFORM   0
#define D3PM_KIND_GAUSSIAN  1
#define D3PM_KIND_ABSORBING 2

#define D3PM_NOISE_UNIFORM     0  /* U[0,1) float32 [B,npos,K], transformed on device */
#define D3PM_NOISE_GUMBEL      1  /* pre-transformed -log(-log u) float32 [B,npos,K]  */
#define D3PM_NOISE_PHILOX      2  /* no tensor: Philox4x32-10 on device (seed, offset) */
#define D3PM_NOISE_EXPONENTIAL 3  /* Exp(1) float64 [B,npos,K] (protein style)         */

#define D3PM_LOSS_KL      0
#define D3PM_LOSS_CE      1   /* cross_entropy_x_start */
#define D3PM_LOSS_HYBRID  2

const char* d3pm_last_error(void);
int d3pm_abi_version(void);

/* ------------------------------------------------------------------------
 * Transition tables
 * --------------------------------------------------------------------- */

/* One-step matrices Q_t [T,K,K] (row = from, column = to) and cumulative
 * products Qbar_t = Q_0 ... Q_t, float64.
 * Replaces DiscreteDiffusion.__init__ table construction,
 * d3pm/diffusion_discrete.py:56-99 with _get_full_transition_mat :101-119,
 * _get_transition_mat :122-154, _get_gaussian_transition_mat :156-190 and
 * _get_absorbing_transition_mat :192-202.  bands < 0 means "None". */
int d3pm_build_tables_f64(int kind, int K, int T, int bands, const double* betas,
                          double* q_onestep, double* q_mats, void* stream);

/* Protein-style uniform tables: Q_t = off_t * ones + norm_t * I (float32
 * scalars promoted to float64) and their running products.
 * Replaces Tokenizer.q_uniform_schedule + cumprod_matrix,
 * protein_generation/utils.py:65-83 and :7-18.  off[T], norm[T] float32. */
int d3pm_protein_uniform_tables_f64(int K, int T, const float* off, const float* norm,
                                    double* q_onestep, double* q_bar, void* stream);

/* Generic running product of T float64 [K,K] matrices with strides in elements
 * (cumprod_matrix, protein_generation/utils.py:7-18). */
int d3pm_cumprod_f64(int K, int T, const double* mats, long st, long sr, long sc,
                     double* out, void* stream);

/* Column-structured digest of a [T,K,K] float64 table with element strides
 * (st, sr, sc): if, after the float32 cast the reference applies in _at
 * (d3pm/diffusion_discrete.py:218), every column j of every matrix t has one
 * off-diagonal value, the table is  entry(i,j) = i==j ? diag[t][j] : off[t][j].
 * digest is float32 [T][8][K]: diag, off, log(diag+eps), log(off+eps), and four
 * rows for sampling a row of the table with weights p + eps (the Gumbel-max of
 * q_sample :266-285 draws from those): w = off+eps, its inclusive prefix sums,
 * P(stay in class j) as a 32-bit integer threshold (bit pattern), one spare row.
 * *violations (device int, zeroed by the caller) counts entries that break the
 * structure; the dense path must be used when it is non-zero. */
int d3pm_digest_f32(const double* table, long st, long sr, long sc, int T, int K, float eps,
                    float* digest, int* violations, void* stream);

/* Same test in float64 with a relative tolerance; digest float64 [T][4][K]:
 * diag, off, log(diag), log(off) (no eps).  For the protein-style kernels, which compute in float64
 * (protein_generation/losses.py:65). */
int d3pm_digest_f64(const double* table, long st, long sr, long sc, int T, int K, double rtol,
                    double* digest, int* violations, void* stream);

/* 64-bit content fingerprint of a strided [T,K,K] float64 table, added into
 * *out (device, zeroed by the caller): sum over all elements of bit pattern x
 * index-derived odd multiplier, modulo 2^64.  The reference's protein drivers
 * hand Q and Q_bar to every call and upload them anew every step
 * (Protein Generation Notebook.ipynb cell-12: Q.to(device), Q_bar.to(device);
 * losses.py:48, generate.py:8 take them as arguments): a caller that caches
 * digests per table keys them on this, so that a fresh copy of known tables
 * costs one short launch instead of two digest builds. */
int d3pm_table_fingerprint(const double* table, long st, long sr, long sc, int T, int K, uint64_t* out,
                           void* stream);

/* float64 -> float32 copy of a strided [T,K,K] table into a contiguous one
 * (the `a.to(dtype=float32)` of _at/_at_onehot, :218,:242, done once). */
int d3pm_table_to_f32(const double* table, long st, long sr, long sc, int T, int K,
                      float* out, void* stream);

/* A contiguous float32 [T][K][K] table (row = reduction index k, column =
 * output class m of the contraction it serves) of probabilities (entries in
 * [0, 1]) as two float16 planes, hi = float16(2^14 x), lo = float16(2^14 x - hi),
 * round to nearest: 22 significant bits of an entry of 2^-16 and above, an
 * absolute 2^-39 below -- stored as the operand image of
 * v_mfma_f32_16x16x32_f16: float16 [T][K/32][K/16][2][64][8] = reduction block u,
 * output-class tile mt, piece, lane, slot, where lane (kg << 4 | mm) holds for
 * class 16 mt + mm the reduction indices 32 u + 16 (s >> 2) + 4 kg + (s & 3),
 * s = 0..7.  4 T K K bytes; K a multiple of 32.  Serves the `a.to(float32)` +
 * matmul of _at_onehot, d3pm/diffusion_discrete.py:242-247, on the float16 matrix
 * instructions. */
int d3pm_table_split_f16x2(const float* table, int T, int K, uint16_t* out, void* stream);

/* Float32 view of the tables a kernel needs.  `structured` selects the digest
 * path (O(K) per token) or the dense path (gathers + K x K contraction). */
typedef struct d3pm_tables_f32 {
  int K, T;
  int structured;
  float eps;                 /* 1e-6, DiscreteDiffusion.eps (:49) */
  const float* qbar_digest;  /* [T][8][K] of q_mats                       */
  const float* qone_digest;  /* [T][8][K] of the one-step Q_t             */
  const float* qbar;         /* [T][K][K] q_mats, float32 (dense only)    */
  const float* qoneT;        /* [T][K][K] transpose_q_onestep_mats (dense)*/
  const float* qbarT;        /* [T][K][K] q_mats transposed (dense, for the backward contraction) */
  /* Optional (both or neither; dense tables with K = 64, 128 or 256 whose entries are at most 1): qbar / qbarT as
   * d3pm_table_split_f16x2 writes them.  When present the K x K contractions of _at_onehot (:242) in the loss and
   * in p_sample run as float32 emulated on the float16 matrix instructions (two float16 pieces per operand, three
   * products per term: relative error below 2^-22 per product, against the 2^-24 of a float32 multiply-add); when
   * NULL, on the float32 matrix instructions (K = 256) or the vector ALUs. */
  const uint16_t* qbar_f16x2;
  const uint16_t* qbarT_f16x2;
} d3pm_tables_f32;

/* Float64 view for the protein-style kernels (structured only). */
typedef struct d3pm_tables_f64 {
  int K, T;
  const double* qbar_digest; /* [T][4][K] of Q_bar: diag, off, log(diag), log(off) */
  const double* qone_digest; /* [T][4][K] of Q_t   */
  const float* lvb_digest;   /* d3pm_lvb_digest_f32 of the two, or NULL (K > 32, or a
                                caller that does not train): the float32 training form
                                of d3pm_lvb_loss then falls back to the float64 kernel */
} d3pm_tables_f64;

/* Per-timestep constants of the float32 training form of d3pm_lvb_loss, from
 * the two float64 digests of `tb` (its lvb_digest member is not read): for the
 * loss at timestep t the column of Q_t, the off-diagonal of Qbar_{t-1}, the
 * entries Qbar_t[x0, x_t] that losses.py:79 divides by (and their ratio to
 * those of Qbar_{t-1}, losses.py:148), formed in float64 and rounded once.
 * out: d3pm_lvb_digest_f32_size(T) bytes on the device.  K <= 32. */
size_t d3pm_lvb_digest_f32_size(int T);
int d3pm_lvb_digest_f32(const d3pm_tables_f64* tb, float* out, void* stream);

/* ------------------------------------------------------------------------
 * Image-style process (d3pm/diffusion_discrete.py)
 * --------------------------------------------------------------------- */

/* q_probs (:250-263): out[b,p,:] = float32(q_mats)[t_b, x0[b,p], :]. */
int d3pm_q_probs(const d3pm_tables_f32* tb, const void* x0, int tok_bytes, const void* t, int t_bytes,
                 int B, long npos, float* out, void* stream);

/* q_sample (:266-285): x_t = argmax_j(log(q_probs + eps) + gumbel), first
 * maximal index; x_t int64.  noise_kind UNIFORM / GUMBEL / PHILOX.  For PHILOX
 * the stream is (seed, offset + *offset_dev); offset_dev (device, may be NULL)
 * lets a captured HIP graph draw fresh noise on every replay. */
int d3pm_q_sample(const d3pm_tables_f32* tb, const void* x0, int tok_bytes, const void* t, int t_bytes,
                  int B, long npos, const float* noise, int noise_kind, uint64_t seed, uint64_t offset,
                  const uint64_t* offset_dev, int64_t* x_t, void* stream);

/* The first two lines of training_losses (:549-556) with on-device noise: t_b
 * uniform on [0, T) for every batch element (t_out int32 [B], the reference
 * draws torch.randint) and x_t = q_sample(x0, t), one launch for column-
 * structured tables.  Philox stream as for d3pm_q_sample.  clear (may be
 * NULL): one float32 the kernel sets to 0 -- the loss_mean accumulator of the
 * d3pm_hybrid_loss launch that follows in the same step. */
int d3pm_q_sample_draw_t(const d3pm_tables_f32* tb, const void* x0, int tok_bytes, int B, long npos,
                         uint64_t seed, uint64_t offset, const uint64_t* offset_dev, float* clear,
                         int32_t* t_out, int64_t* x_t, void* stream);

/* q_posterior_logits (:312-336).  x_start_is_logits == 0: x_start is a token
 * array; != 0: float32 logits [B,npos,ld].  out float32 [B,npos,K].  With
 * x_start_is_logits this is also p_logits (:339-381) for model_output='logits',
 * model_prediction='x_start' (the where(t==0) of :362 coincides with :336). */
int d3pm_q_posterior_logits(const d3pm_tables_f32* tb, const void* x_start, int x_start_is_logits, long ld,
                            int tok_bytes, const void* x_t, int xt_bytes, const void* t, int t_bytes,
                            int B, long npos, float* out, void* stream);

/* Gradient of d3pm_q_posterior_logits(x_start_is_logits != 0) with respect to
 * the x_start logits: what autograd derives in the reference from
 * q_posterior_logits (:320-336) / p_logits (:362-366) when a caller builds a
 * loss of its own on those logits.  grad_out float32 [B,npos,K] (gradient at
 * the posterior logits), grad_logits float32 [B,npos,ld].  Column-structured
 * tables only (D3PM_EUNSUPPORTED otherwise: the facade contracts dense tables
 * with a library GEMM instead). */
int d3pm_q_posterior_logits_backward(const d3pm_tables_f32* tb, const float* x_start_logits, long ld,
                                     const void* t, int t_bytes, int B, long npos, const float* grad_out,
                                     float* grad_logits, void* stream);

/* Workspace (bytes) d3pm_hybrid_loss needs for B x npos x K. */
size_t d3pm_hybrid_loss_workspace(int B, long npos, int K);

/* vb_terms_bpd (:462-489) + cross_entropy_start (:527-542) + the combination
 * of training_losses (:558-582), and the gradient of
 * sum_b grad_scale*grad_out[b]*loss[b] with respect to the denoiser logits, in
 * one pass (grad_scale = 1/B bakes in the `.mean()` of the training loop).
 *   logits  float32 [B,npos,ld] (the model_fn output, 'logits'/'x_start' mode)
 *   grad_out float32 [B] upstream gradient per sample, or NULL for all ones
 *   vb, ce, loss  float32 [B] (any may be NULL)
 *   dlogits float32 [B,npos,ld] or NULL (forward only); columns >= K get 0
 *   loss_mean (may be NULL): float32 scalar that must hold 0 on entry (the
 *     `clear` argument of d3pm_q_sample_draw_t zeroes it in the step's first
 *     launch); every batch element adds loss[b]/B to it with one device-scope
 *     atomic -- the `.mean()` of the training loop (:582 returns [B]; the
 *     notebook reduces it) with no launch of its own.  The order of the B
 *     additions varies: equal to the mean up to float32 rounding, not bit for
 *     bit from run to run (loss[] itself is summed in a fixed order)
 *   rng_counter (may be NULL): *rng_counter += rng_step, done by the loss kernel;
 *     the device-side Philox counter of d3pm_q_sample(_draw_t), so that a
 *     captured HIP graph needs no separate increment
 *   workspace: d3pm_hybrid_loss_workspace() bytes, contents irrelevant. */
int d3pm_hybrid_loss(const d3pm_tables_f32* tb, const float* logits, long ld,
                     const void* x0, int tok_bytes, const void* x_t, int xt_bytes, const void* t, int t_bytes,
                     int B, long npos, int loss_type, float hybrid_coeff, float grad_scale, const float* grad_out,
                     float* vb, float* ce, float* loss, float* dlogits,
                     float* loss_mean, uint64_t* rng_counter, uint64_t rng_step,
                     void* workspace, size_t workspace_bytes, void* stream);

/* d3pm_hybrid_loss for this step, then the first two lines of training_losses
 * (:549-556) for the NEXT batch -- what d3pm_q_sample_draw_t does, with the same
 * Philox stream (seed, offset + *offset_dev) and the same result -- for a
 * training loop whose batches are staged ahead of the step that consumes them
 * (the reference's loop draws t and x_t at the top of every step, notebook
 * cell-12; here step i's loss launch hands step i+1 its t and x_t).  One launch
 * where the fused kernel takes the shape (column-structured tables, K = 32,
 * 16-byte aligned rows with ld = K, B >= 192, npos <= 1024, a gradient asked
 * for: the corruption runs while the loss's rows are on their way from memory),
 * otherwise the two launches in sequence.  next_x_t / next_t_out must not be
 * this step's x_t / t.  The Philox counter is only read here (offset_dev may be
 * NULL): advance it with the rng_counter of a plain d3pm_hybrid_loss, e.g. the
 * last step of a captured sequence. */
int d3pm_hybrid_loss_then_q_sample(const d3pm_tables_f32* tb, const float* logits, long ld, const void* x0,
                                   int tok_bytes, const void* x_t, int xt_bytes, const void* t, int t_bytes,
                                   int B, long npos, int loss_type, float hybrid_coeff, float grad_scale,
                                   const float* grad_out, float* vb, float* ce, float* loss, float* dlogits,
                                   void* workspace, size_t workspace_bytes, const void* next_x0,
                                   int next_tok_bytes, uint64_t seed, uint64_t offset, const uint64_t* offset_dev,
                                   int32_t* next_t_out, int64_t* next_x_t, void* stream);

/* d3pm_hybrid_loss for callers that cannot allocate device memory or carry a
 * workspace (code under HIP graph capture, tight serving loops, harnesses that
 * forbid hipMalloc).  No counterpart in the reference.  Same arguments, results
 * and kernels as d3pm_hybrid_loss, minus the workspace: the per-workgroup
 * partial sums go to a 128 KiB array in the library's device image (one copy
 * per GPU, held until the process ends).  The call allocates and frees nothing.
 * Limit: 8 bytes per workgroup of the loss launch, i.e. at most 16 384
 * workgroups -- any structured (uniform / absorbing) problem, and dense
 * (Gaussian / banded) problems up to B x chunks <= 16 384; beyond that the call
 * returns D3PM_EINVAL and launches nothing.  All calls on one GPU share the
 * array: calls that overlap on one GPU (two streams, two threads) race on it;
 * calls in stream order do not.  The first call on a device looks the array's
 * address up (hipGetSymbolAddress): make that call outside a graph capture. */
int d3pm_hybrid_loss_static(const d3pm_tables_f32* tb, const float* logits, long ld,
                            const void* x0, int tok_bytes, const void* x_t, int xt_bytes, const void* t, int t_bytes,
                            int B, long npos, int loss_type, float hybrid_coeff, float grad_scale, const float* grad_out,
                            float* vb, float* ce, float* loss, float* dlogits,
                            float* loss_mean, uint64_t* rng_counter, uint64_t rng_step, void* stream);

/* In-place per-sample rescale g[b,:] *= upstream[b]/baked; blocks whose
 * upstream[b] equals `baked` exactly touch nothing.  Backward of the autograd
 * node when the upstream gradient differs from the scale d3pm_hybrid_loss /
 * d3pm_lvb_loss baked in. */
int d3pm_scale_rows(float* g, const float* upstream, float baked, int B, long row_elems, void* stream);

/* p_sample (:384-403): sample = argmax_j(p_logits + [t!=0]*gumbel) int64;
 * probs (may be NULL) = softmax(logits) float32 [B,npos,K].  offset_dev as in
 * d3pm_q_sample. */
int d3pm_p_sample(const d3pm_tables_f32* tb, const float* logits, long ld, const void* x_t, int xt_bytes,
                  const void* t, int t_bytes, int B, long npos, const float* noise, int noise_kind,
                  uint64_t seed, uint64_t offset, const uint64_t* offset_dev, int64_t* sample, float* probs,
                  void* stream);

/* prior_bpd (:492-524): KL(q(x_{T-1}|x0) || stationary)/ln2 averaged over
 * positions; stationary is uniform, or one-hot at K/2 when absorbing != 0.
 * out float32 [B]; workspace as for d3pm_hybrid_loss. */
int d3pm_prior_bpd(const d3pm_tables_f32* tb, const void* x0, int tok_bytes, int B, long npos,
                   int absorbing, float* out, void* workspace, size_t workspace_bytes, void* stream);

/* _get_logits_from_logistic_pars (d3pm/diffusion_discrete.py:288-309) with
 * utils.log_min_exp (d3pm/utils.py:15-18): the model_output == 'logistic_pars'
 * head.  logits[n,k] = hi + log1p(-exp(lo - hi) + eps) with
 * lo/hi = logsigmoid(exp(2 - log_scale[n]) * (centers[k] - loc[n] -/+ 1/(K-1))).
 *   loc, log_scale float32 [N] (N = B*npos); centers float32 [K], the reference's
 *   torch.linspace(-1, 1, K); logits float32 [N,K]. */
int d3pm_logistic_logits(const float* loc, const float* log_scale, const float* centers, long N, int K,
                         float eps, float* logits, void* stream);

/* Gradient of the above (what autograd derives from :288-309 in the
 * reference): dloc[n] = sum_k dlogits[n,k] * d logits[n,k] / d loc[n], likewise
 * dlog_scale; float32 [N]. */
int d3pm_logistic_logits_backward(const float* loc, const float* log_scale, const float* centers,
                                  const float* dlogits, long N, int K, float eps, float* dloc,
                                  float* dlog_scale, void* stream);

/* ------------------------------------------------------------------------
 * Protein-style process (protein_generation/)
 * --------------------------------------------------------------------- */

/* Batched sample_transition_matrix (utils.py:44-52) as the collaters use it
 * (dataset.py:234-242): x_t[b,p] = argmax_j(Qbar[t_b][x0,j] / E_j), the
 * CPU torch.multinomial draw.  expo float64 [B,npos,K] or NULL (Philox).
 * probs (may be NULL) float64 [B,npos,K] = the q_x the collater returns. */
int d3pm_corrupt(const d3pm_tables_f64* tb, const int64_t* x0, const int64_t* t, int B, long npos,
                 const double* expo, uint64_t seed, uint64_t offset, int64_t* x_t, double* probs,
                 void* stream);

/* Workspace (bytes) d3pm_lvb_loss needs for B x npos x K. */
size_t d3pm_lvb_loss_workspace(int B, long npos, int K);

/* D3PMLVBLoss.forward (losses.py:48-93) / D3PMLVBLossMSA.forward (:120-155)
 * per batch element.  msa = 0: single sequences; msa = D > 0: MSAs of depth D
 * (npos = D*L): the posterior denominator comes from Qbar[t-1] (:148) and the
 * t==tmax term is normalised by D, the leading size of q[i] (:130).  Also the gradient of
 * grad_scale * sum_b per_seq[b] with respect to the predictions.
 *   pred float32 [B,npos,ld], ld = n_tokens >= K (t==1 branch uses all ld)
 *   q    float64 [B,npos,K] forward probabilities, read only where t==tmax; may be
 *        NULL, and then an element with t==tmax gets per_seq = NaN (nothing is read)
 *   per_seq float64 [B];  dpred float32 [B,npos,ld] or NULL
 *   workspace: d3pm_lvb_loss_workspace() bytes, contents irrelevant. */
int d3pm_lvb_loss(const d3pm_tables_f64* tb, const float* pred, long ld, const int64_t* x0,
                  const int64_t* x_t, const int64_t* t, const double* q, int B, long npos, int tmax,
                  int msa, double grad_scale, double* per_seq, float* dpred,
                  void* workspace, size_t workspace_bytes, void* stream);

/* The protein-style training loss as the driver forms it (Protein Generation
 * Notebook.ipynb cell-12 / cell-13):
 *   loss = (lvb + lambda * CrossEntropyLoss(outputs.reshape(-1, n), tgt.reshape(-1))) * n_tokens
 * d3pm_lvb_loss plus, in the same pass over the same rows, the cross-entropy of
 * EVERY token whatever its element's timestep: -log_softmax(pred)[x0] over all
 * ld logits (losses.py:52-55 is the same term for an element at t == 1).
 *   ce_seq float64 [B]: mean over the element's positions of that term, so that
 *     the driver's batch mean is sum_b ce_seq[b] / B
 *   dpred: gradient of grad_scale * sum_b per_seq[b] + ce_grad_scale * sum_b ce_seq[b];
 *     the notebook's loss is grad_scale = n_tokens / B, ce_grad_scale = lambda * n_tokens / B
 * Everything else as d3pm_lvb_loss (whose workspace size serves both). */
int d3pm_lvb_hybrid_loss(const d3pm_tables_f64* tb, const float* pred, long ld, const int64_t* x0,
                         const int64_t* x_t, const int64_t* t, const double* q, int B, long npos, int tmax,
                         int msa, double grad_scale, double ce_grad_scale, double* per_seq, double* ce_seq,
                         float* dpred, void* workspace, size_t workspace_bytes, void* stream);

/* Reverse step of generate_d3pm (generate.py:28-40) and of generate_msa_d3pm
 * (generate.py:70-93): x_{t-1} = argmax_j(P_j / E_j), P the normalised p-squared
 * marginal; expo float64 [B,npos,K] holds the Exp(1) draws E, or is NULL for
 * on-device noise: Philox stream (seed, offset + *offset_dev), offset_dev a
 * device counter or NULL (as in d3pm_q_sample; lets a captured HIP graph draw
 * fresh noise on every replay).  t int64 [B].
 *   last_class_div: P[K-1] is divided by this (1 + penalty_value of the MSA
 *     sampler, generate.py:84-86; 1.0 otherwise)
 *   exclude_last_prefix: class K-1 cannot be drawn at positions < this (the
 *     query row of an MSA, generate.py:88; 0 otherwise) */
int d3pm_lvb_reverse_step(const d3pm_tables_f64* tb, const float* pred, long ld, const int64_t* x_t,
                          const int64_t* t, int B, long npos, const double* expo, uint64_t seed,
                          uint64_t offset, const uint64_t* offset_dev, double last_class_div,
                          long exclude_last_prefix, int64_t* out, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* D3PM_HIP_H */
