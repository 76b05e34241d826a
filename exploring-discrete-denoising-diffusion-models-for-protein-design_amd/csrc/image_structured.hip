// Image-style D3PM kernels on column-structured tables
// (entry(i,j) = i==j ? diag[j] : off[j]; uniform and absorbing transitions).
//
// With that structure the K x K contraction of the predicted x0 distribution
// against Qbar_{t-1} collapses to O(K) per token,
//   sum_i pi_i Qbar[i,j] = off_j * (1 - pi_j) + diag_j * pi_j,
// so every kernel here is a streaming pass over [B*npos, K] float32 rows:
// HBM-bound, nothing but the inputs and outputs touches memory.  One workgroup
// works on positions of a single batch element (one timestep), keeps that
// timestep's digest rows in registers, and reduces over K with DPP butterflies
// inside lane groups.
#include "layout.h"
#include "rowimage.h"

#include <cstdlib>
#include <cstring>
#include <mutex>
#include <type_traits>
#include <utility>

namespace d3pm {

constexpr float LN2 = 0.693147180559945309417f;
constexpr int BLOCK = 256;

struct ImgArgs {
  d3pm_tables_f32 tb;
  const void* x0; int x0_bytes;
  const void* xt; int xt_bytes;
  const void* t; int t_bytes;
  int B; long npos;
  int chunks; long chunk_pos;
};

// Digest component c (0 diag, 1 off, 2 log(diag+eps), 3 log(off+eps)) of matrix t.
__device__ __forceinline__ const float* dg(const float* digest, int t, int c, int K) {
  return digest + ((size_t)t * 8 + c) * K;        // c: 0 diag, 1 off, 2 log(diag+eps), 3 log(off+eps), 4-6 sampling rows
}

template <class L>
__device__ __forceinline__ void load_digest(const float* digest, int t, int c, int K, int g, float (&out)[L::V]) {
  load_slots<L>(dg(digest, t, c, K), g, K, 0.0f, out);
}

// ------------------------------------------------------------------ q_probs
template <class L>
__global__ __launch_bounds__(BLOCK) void k_q_probs(ImgArgs a, float* __restrict__ out) {
  const int K = a.tb.K;
  const int b = blockIdx.x / a.chunks, chunk = blockIdx.x % a.chunks;
  const int g = threadIdx.x % L::G, slot = threadIdx.x / L::G;
  const int t = load_index(a.t, b, a.t_bytes, a.tb.T);
  float d[L::V], o[L::V];
  load_digest<L>(a.tb.qbar_digest, t, 0, K, g, d);
  load_digest<L>(a.tb.qbar_digest, t, 1, K, g, o);
  const long p1 = min(a.npos, (chunk + 1) * a.chunk_pos);
  for (long pos = chunk * a.chunk_pos + slot; pos < p1; pos += BLOCK / L::G) {
    const long n = (long)b * a.npos + pos;
    const int x0 = load_index(a.x0, n, a.x0_bytes, a.tb.K);
    float r[L::V];
#pragma unroll
    for (int v = 0; v < L::V; ++v) r[v] = (L::cls(g, v) == x0) ? d[v] : o[v];
    store_slots<L>(out + n * K, g, K, r);
  }
}

// ------------------------------------------------------------------ q_sample
template <class L, int NOISE>
__global__ __launch_bounds__(BLOCK) void k_q_sample(ImgArgs a, const float* __restrict__ noise, uint64_t seed,
                                                    uint64_t offset, const uint64_t* __restrict__ offset_dev,
                                                    int64_t* __restrict__ out) {
  if (NOISE == D3PM_NOISE_PHILOX && offset_dev) offset += *offset_dev;
  const int K = a.tb.K;
  const int b = blockIdx.x / a.chunks, chunk = blockIdx.x % a.chunks;
  const int g = threadIdx.x % L::G, slot = threadIdx.x / L::G;
  const int t = load_index(a.t, b, a.t_bytes, a.tb.T);
  float ld_[L::V], lo_[L::V];
  load_digest<L>(a.tb.qbar_digest, t, 2, K, g, ld_);
  load_digest<L>(a.tb.qbar_digest, t, 3, K, g, lo_);
  const long p1 = min(a.npos, (chunk + 1) * a.chunk_pos);
  for (long pos = chunk * a.chunk_pos + slot; pos < p1; pos += BLOCK / L::G) {
    const long n = (long)b * a.npos + pos;
    const int x0 = load_index(a.x0, n, a.x0_bytes, a.tb.K);
    float u[L::V];
    if constexpr (NOISE == D3PM_NOISE_PHILOX) {
      const U4 r = philox4x32_10((uint64_t)n * L::G + g, offset, seed);
      const uint32_t rr[4] = {r.x, r.y, r.z, r.w};
#pragma unroll
      for (int v = 0; v < L::V; ++v) u[v] = u01(rr[v]);
    } else {
      load_slots<L>(noise + n * K, g, K, 0.5f, u);
    }
    float best = -INFINITY;
    int besti = 0x7fffffff;
#pragma unroll
    for (int v = 0; v < L::V; ++v) {
      const int c = L::cls(g, v);
      const float gum = (NOISE == D3PM_NOISE_GUMBEL) ? u[v]
                        : (NOISE == D3PM_NOISE_PHILOX ? gumbel_from_uniform_fast(u[v]) : gumbel_from_uniform(u[v]));
      const float val = ((c == x0) ? ld_[v] : lo_[v]) + gum;
      const bool ok = c < K;
      // NaN-free inputs: strict > keeps the first maximal class of this lane
      if (ok && (val > best || besti == 0x7fffffff)) { best = val; besti = c; }
    }
    group_argmax<L::G>(best, besti);
    if (g == 0) out[n] = besti;
  }
}

// ------------------------------------------------------------------ q_sample, device noise, structured tables
// The reference samples x_t by Gumbel-max over softmax(log(q + eps)), i.e. from
// weights w_j = q(x_t = j | x0) + eps.  With column structure those weights are
// off_j + eps for every j except x0, where the weight is diag_x0 + eps.  With noise
// drawn on the device the draw has two stages, from the two words of one
// Philox2x32-10 call per token:
//   x_t == x0 with probability (diag_x0 + eps) / (sum_j w_j), decided on a 32-bit
//   integer threshold formed in float64 when the digest is built (a step that moves
//   one token in 10^5 keeps five digits of that probability);
//   otherwise class j != x0 in proportion to off_j + eps: inverse CDF over the
//   prefix sums of (off + eps), kept in LDS per workgroup, with x0's share taken
//   out of the target; binary search, float32.
// One thread per token, 16 bytes of traffic per token.  The per-launch key of the
// token stream is a Philox4x32-10 call on (offset, seed), which the scalar unit
// evaluates.  QsTables is the per-timestep part, copied from rows 4-6 of the digest.
template <int KMAX>
struct QsTablesT {
  float cdf[KMAX + 1];   // cdf[j] = sum_{i<j} (off_i + eps)
  float w[KMAX];         // off_j + eps
  uint32_t stay[KMAX];   // 2^32 * P(x_t == x0 = j)
};
typedef QsTablesT<256> QsTables;

// The per-timestep tables come ready from the digest (rows 4-6, built once at set-up: prefix sums in class order,
// thresholds in float64): entry j is copied by thread / lane j; the caller orders the copy before qs_draw (a workgroup
// barrier, or a wave-level fence for a wave-private copy).
template <class Tables>
__device__ __forceinline__ void qs_copy(Tables& q, const float* __restrict__ digest, int t, int K, int j) {
  if (j == 0) q.cdf[0] = 0.0f;
  if (j < K) {
    q.w[j] = dg(digest, t, 4, K)[j];
    q.cdf[j + 1] = dg(digest, t, 5, K)[j];
    q.stay[j] = __builtin_bit_cast(uint32_t, dg(digest, t, 6, K)[j]);
  }
}

struct QsKey { uint32_t x, y, z; };
__device__ __forceinline__ QsKey qs_key(uint64_t offset, uint64_t seed) {
  // (the offset may have come through a vector load: say that it is uniform, so that the scalar unit does this)
  const uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)offset);
  const uint32_t hi = __builtin_amdgcn_readfirstlane((uint32_t)(offset >> 32));
  const U4 pad = philox4x32_10(((uint64_t)hi << 32) | lo, 0xA4093822299F31D0ull, seed);
  return QsKey{pad.x, pad.y, pad.z};
}

// x_t of token n (global index) whose clean class is x0.
template <class Tables>
__device__ __forceinline__ int qs_draw(const Tables& q, int K, int x0, uint64_t n, QsKey key) {
#pragma clang fp contract(off)      // the same roundings in every kernel this is inlined into
  const U2 r = philox2x32_10((uint32_t)n ^ key.x, (uint32_t)(n >> 32) ^ key.y, key.z);
  if (r.y < q.stay[x0]) return x0;
  const float lo = q.cdf[x0], wx = q.w[x0];
  const float target = u01_open(r.x) * (q.cdf[K] - wx);      // over the other classes
  const float rr = target < lo ? target : target + wx;
  // largest j with cdf[j] <= rr, j in [0, K-1]
  int l = 0, h = K;            // invariant: cdf[l] <= rr < cdf[h] (up to rounding at the ends)
  while (h - l > 1) {
    const int mid = (l + h) >> 1;
    if (q.cdf[mid] <= rr) l = mid; else h = mid;
  }
  // rounding at the splice can land on x0's own slot: move to the side the target came from
  if (l == x0) l = (target < lo) ? (x0 > 0 ? x0 - 1 : (K > 1 ? 1 : 0)) : (x0 + 1 < K ? x0 + 1 : (K > 1 ? K - 2 : 0));
  return l;
}

// With `t_draw` the timestep of each batch element is drawn here as well, uniform on [0, T): one Philox2x32-10 call
// per batch element under the launch's key with its last word inverted (a key of its own: disjoint from the token
// stream), written to t_draw[b] for the denoiser and the loss.
__device__ __forceinline__ int draw_timestep(int b, int T, QsKey key) {
  const U2 r = philox2x32_10((uint32_t)b ^ key.x, key.y ^ 0x9E3779B9u, ~key.z);
  return (int)(((uint64_t)r.x * (uint64_t)T) >> 32);
}
__device__ __forceinline__ int draw_timestep(int b, int T, uint64_t offset, uint64_t seed) {
  return draw_timestep(b, T, qs_key(offset, seed));
}

// qs_draw's search as a count, for a compile-time K: cdf is non-decreasing, so the largest j with cdf[j] <= rr is the
// number of entries cdf[1 .. K-1] that are <= rr -- the same index the bisection finds, from K / 4 wide LDS reads
// issued together instead of log2(K) dependent ones.
template <int KC, class Tables>
__device__ __forceinline__ int qs_draw_counted(const Tables& q, int x0, uint64_t n, QsKey key) {
#pragma clang fp contract(off)      // the same roundings as qs_draw
  const U2 r = philox2x32_10((uint32_t)n ^ key.x, (uint32_t)(n >> 32) ^ key.y, key.z);
  if (r.y < q.stay[x0]) return x0;
  const float lo = q.cdf[x0], wx = q.w[x0];
  const float target = u01_open(r.x) * (q.cdf[KC] - wx);
  const float rr = target < lo ? target : target + wx;
  int l = 0;
#pragma unroll
  for (int j = 1; j < KC; ++j) l += (q.cdf[j] <= rr) ? 1 : 0;
  if (l == x0) l = (target < lo) ? (x0 > 0 ? x0 - 1 : (KC > 1 ? 1 : 0)) : (x0 + 1 < KC ? x0 + 1 : (KC > 1 ? KC - 2 : 0));
  return l;
}

__global__ __launch_bounds__(BLOCK) void k_q_sample_icdf(ImgArgs a, uint64_t seed, uint64_t offset,
                                                         const uint64_t* __restrict__ offset_dev,
                                                         int32_t* __restrict__ t_draw, float* __restrict__ clear,
                                                         int64_t* __restrict__ out) {
  if (clear && blockIdx.x == 0 && threadIdx.x == 0) *clear = 0.0f;
  __shared__ QsTables q;
  const int K = a.tb.K;
  if (offset_dev) offset += *offset_dev;
  const int b = blockIdx.x / a.chunks, chunk = blockIdx.x % a.chunks;
  const long p1 = min(a.npos, (chunk + 1) * a.chunk_pos);
  const long pfirst = chunk * a.chunk_pos + threadIdx.x;
  // the first token's load is in flight while the timestep -> digest -> prefix-sum chain runs
  int x0_next = 0;
  if (pfirst < p1) x0_next = load_index(a.x0, (long)b * a.npos + pfirst, a.x0_bytes, a.tb.K);
  const QsKey key = qs_key(offset, seed);
  int t;
  if (t_draw) {
    t = draw_timestep(b, a.tb.T, key);
    if (chunk == 0 && threadIdx.x == 0) t_draw[b] = t;
  } else {
    t = load_index(a.t, b, a.t_bytes, a.tb.T);
  }
  qs_copy(q, a.tb.qbar_digest, t, K, threadIdx.x);         // K <= 256 = BLOCK
  __syncthreads();
  for (long pos = pfirst; pos < p1; pos += blockDim.x) {
    const long n = (long)b * a.npos + pos;
    const int x0 = x0_next;
    if (pos + blockDim.x < p1) x0_next = load_index(a.x0, n + blockDim.x, a.x0_bytes, a.tb.K);
    out[n] = qs_draw(q, K, x0, (uint64_t)n, key);
  }
}

// ------------------------------------------------------------------ posterior pieces
// Everything a token needs from the digests, for t > 0.
template <class L>
struct Post {
  float bd[L::V], bo[L::V];     // Qbar_{t-1} diag / off per class
  float lbd[L::V], lbo[L::V];   // their log(. + eps)
};
template <class L>
__device__ __forceinline__ void load_post(const d3pm_tables_f32& tb, int t, int g, Post<L>& p) {
  load_digest<L>(tb.qbar_digest, t - 1, 0, tb.K, g, p.bd);
  load_digest<L>(tb.qbar_digest, t - 1, 1, tb.K, g, p.bo);
  load_digest<L>(tb.qbar_digest, t - 1, 2, tb.K, g, p.lbd);
  load_digest<L>(tb.qbar_digest, t - 1, 3, tb.K, g, p.lbo);
}

// softmax of a token's logits over the lane group; returns max and sum(exp).
template <class L>
__device__ __forceinline__ void group_softmax(const float (&z)[L::V], float (&pi)[L::V], float& zmax, float& zsum) {
  float m = z[0];
#pragma unroll
  for (int v = 1; v < L::V; ++v) m = fmaxf(m, z[v]);
  m = group_max<L::G>(m);
  float s = 0;
#pragma unroll
  for (int v = 0; v < L::V; ++v) { pi[v] = expf(z[v] - m); s += pi[v]; }   // exp(-inf) = 0 on masked slots
  s = group_sum<L::G>(s);
  const float inv = 1.0f / s;
#pragma unroll
  for (int v = 0; v < L::V; ++v) pi[v] *= inv;
  zmax = m; zsum = s;
}

// ------------------------------------------------------------------ q_posterior_logits
template <class L, bool FROM_LOGITS>
__global__ __launch_bounds__(BLOCK) void k_posterior_logits(ImgArgs a, const float* __restrict__ xs_logits, long ld,
                                                            float* __restrict__ out) {
  const int K = a.tb.K;
  const float eps = a.tb.eps;
  const int b = blockIdx.x / a.chunks, chunk = blockIdx.x % a.chunks;
  const int g = threadIdx.x % L::G, slot = threadIdx.x / L::G;
  const int t = load_index(a.t, b, a.t_bytes, a.tb.T);
  Post<L> P;
  if (t > 0) load_post<L>(a.tb, t, g, P);
  const float log_one = (float)log((double)(1.0f + eps)), log_zero = (float)log((double)eps);
  const long p1 = min(a.npos, (chunk + 1) * a.chunk_pos);
  for (long pos = chunk * a.chunk_pos + slot; pos < p1; pos += BLOCK / L::G) {
    const long n = (long)b * a.npos + pos;
    const int xt = load_index(a.xt, n, a.xt_bytes, a.tb.K);
    float r[L::V];
    float z[L::V];
    int x0 = -1;
    if constexpr (FROM_LOGITS) load_slots<L>(xs_logits + n * ld, g, K, -INFINITY, z);
    else x0 = load_index(a.x0, n, a.x0_bytes, a.tb.K);
    if (t == 0) {
#pragma unroll
      for (int v = 0; v < L::V; ++v) r[v] = FROM_LOGITS ? z[v] : ((L::cls(g, v) == x0) ? log_one : log_zero);
    } else {
      const float l1d = dg(a.tb.qone_digest, t, 2, K)[xt], l1o = dg(a.tb.qone_digest, t, 3, K)[xt];
      if constexpr (FROM_LOGITS) {
        float pi[L::V], zm, zs;
        group_softmax<L>(z, pi, zm, zs);
#pragma unroll
        for (int v = 0; v < L::V; ++v) {
          const float f2 = fmaf(P.bd[v] - P.bo[v], pi[v], P.bo[v]);
          r[v] = ((L::cls(g, v) == xt) ? l1d : l1o) + logf(f2 + eps);
        }
      } else {
#pragma unroll
        for (int v = 0; v < L::V; ++v) {
          const int c = L::cls(g, v);
          r[v] = ((c == xt) ? l1d : l1o) + ((c == x0) ? P.lbd[v] : P.lbo[v]);
        }
      }
    }
    store_slots<L>(out + n * K, g, K, r);
  }
}

// ------------------------------------------------------------------ q_posterior_logits, gradient to the x_start logits
// out_j = log(a_j + eps) + log(f2_j + eps), f2 = softmax(z) Qbar_{t-1}: with G the gradient arriving at `out`,
//   gf2_j = G_j / (f2_j + eps),  g pi_i = sum_j gf2_j Qbar[i,j] = S + gf2_i (diag_i - off_i),  S = sum_j gf2_j off_j,
//   gz_i = pi_i (g pi_i - <pi, g pi>);   t == 0: out = z, so gz = G.
// (what autograd derives from d3pm/diffusion_discrete.py:320-336 in the reference)
template <class L>
__global__ __launch_bounds__(BLOCK) void k_posterior_logits_backward(ImgArgs a, const float* __restrict__ xs_logits,
                                                                     long ld, const float* __restrict__ gout,
                                                                     float* __restrict__ gz_out) {
  const int K = a.tb.K;
  const float eps = a.tb.eps;
  const int b = blockIdx.x / a.chunks, chunk = blockIdx.x % a.chunks;
  const int g = threadIdx.x % L::G, slot = threadIdx.x / L::G;
  const int t = load_index(a.t, b, a.t_bytes, a.tb.T);
  Post<L> P;
  if (t > 0) load_post<L>(a.tb, t, g, P);
  const long p1 = min(a.npos, (chunk + 1) * a.chunk_pos);
  for (long pos = chunk * a.chunk_pos + slot; pos < p1; pos += BLOCK / L::G) {
    const long n = (long)b * a.npos + pos;
    float z[L::V], G[L::V], r[L::V];
    load_slots<L>(gout + n * K, g, K, 0.0f, G);
    if (t == 0) {
#pragma unroll
      for (int v = 0; v < L::V; ++v) r[v] = G[v];
    } else {
      load_slots<L>(xs_logits + n * ld, g, K, -INFINITY, z);
      float pi[L::V], zm, zs;
      group_softmax<L>(z, pi, zm, zs);
      float S = 0;
#pragma unroll
      for (int v = 0; v < L::V; ++v) {
        const bool ok = L::VEC || L::cls(g, v) < K;
        const float f2 = fmaf(P.bd[v] - P.bo[v], pi[v], P.bo[v]);
        G[v] = ok ? G[v] / (f2 + eps) : 0.0f;                 // gf2
        S = fmaf(G[v], ok ? P.bo[v] : 0.0f, S);
      }
      S = group_sum<L::G>(S);
      float dot = 0;
#pragma unroll
      for (int v = 0; v < L::V; ++v) {
        G[v] = fmaf(G[v], P.bd[v] - P.bo[v], S);              // g pi
        dot = fmaf(pi[v], G[v], dot);
      }
      dot = group_sum<L::G>(dot);
#pragma unroll
      for (int v = 0; v < L::V; ++v) r[v] = pi[v] * (G[v] - dot);
    }
    store_slots<L>(gz_out + n * ld, g, K, r);
    if (ld > K) zero_tail<L>(gz_out + n * ld, g, K, ld);
  }
}

// ------------------------------------------------------------------ fused hybrid loss, forward + backward
// Per token (t > 0), with pi = softmax(z), a = column x_t of Q_t (+eps),
// Qm = Qbar_{t-1}, b = row x0 of Qm (+eps), f2 = pi Qm + eps:
//   p_model ~ a .* f2,   q_true ~ a .* b           (normalised over classes: Zm, Zt)
//   kl = sum_j q_j log(b_j / f2_j) + log(Zm / Zt),   ce = -log_softmax(z)[x0]
// which is the reference's softmax(log(a)+log(.)+eps) / log_softmax difference
// (d3pm/utils.py:54-58) with exp(log(x)) folded to x; logs are taken in base 2
// because the result is wanted in bits.  t == 0: kl is replaced by ce.
//
// Column structure makes a and b constant over classes except at x_t
// (a = a_o + da) and x0 (b = boe + bdo), so every class is first treated as a
// generic one (a_o, boe_j) inside the lanes, and the two special classes are
// corrected per token with scalars:
//   Zt  = a_o (sum_j boe_j + bdo_x0) + da b_xt,          b_xt = boe_xt + [x0==xt] bdo_xt
//   Zm  = a_o sum_j f2_j + da f2_xt
//   kl Zt = a_o S1 - a_o boe_x0 lg(boe_x0/f2_x0) + a_x0 b_x0 lg(b_x0/f2_x0)
//           + [x0!=xt] da boe_xt lg(boe_xt/f2_xt),        S1 = sum_j boe_j lg(boe_j/f2_j)
//   d kl / d f2_j = w a_j (1/Zm - b_j/(Zt f2_j)) = A1 - A2 boe_j/f2_j  (+ d_x0 at x0, + d_xt at xt)
// and the chain through f2 = boe + bdo pi and the softmax of z is
//   g pi_j = R + gf2_j bdo_j,  R = sum_j gf2_j bo_j,  gz_j = pi_j (g pi_j - <pi, g pi>) + w_ce (pi_j - [j==x0]).
// Per class: one exp2, one rcp, one log2 and ~20 other VALU operations; four
// rounds of lane-group reductions per token.  The values at x0 and x_t are read
// again from the row (an L1 hit) instead of being extracted across lanes.
//
// Rows move through a rolling pipeline of PIPE slots per lane group (64 bytes
// per lane in flight), the first of them requested before the
// timestep -> digest dependency chain.
//
// Reduction over positions: lanes by butterfly, waves through LDS in index
// order, then either directly to vb/ce/loss (one workgroup per batch element)
// or to `part` ([B][chunks][2]) for k_loss_finalize; the order is fixed, so the
// result does not depend on scheduling.
constexpr float LOG2E = 1.44269504088896340736f;
// rows in flight per lane group: 64 bytes per lane whatever the layout
template <class L> constexpr int pipe_depth() { return L::V >= 8 ? 2 : 4; }
template <class L> constexpr int loss_max_threads() { return 1024; }
constexpr int MAX_WAVES = 16;

__device__ __forceinline__ float fast_exp2(float x) { return __builtin_amdgcn_exp2f(x); }
__device__ __forceinline__ float fast_log2(float x) { return __builtin_amdgcn_logf(x); }
__device__ __forceinline__ float fast_rcp(float x) { return __builtin_amdgcn_rcpf(x); }

struct LossOut {
  float* vb; float* ce; float* loss;   // [B], any may be null (chunks == 1)
  float* part;                         // [B][chunks][2] (chunks > 1)
  // the caller's Philox counter, advanced by the first workgroup of the loss kernel:
  // every reader of the old value (the corruption kernel) finished before this
  // launch began, the next reader starts after it ends
  unsigned long long* rng_counter; unsigned long long rng_step;
  // batch mean (may be null): every batch element adds loss[b] * inv_B with one device-scope atomic; the caller
  // zeroes it before the launch (d3pm_q_sample_draw_t does, in the step's first launch)
  float* mean; float inv_B;
};

__device__ __forceinline__ void write_losses(const LossOut& o, int b, float svb, float sce, long npos, int loss_type,
                                             float hybrid_coeff) {
  // sums arrive in bits
  const float v = svb / (float)npos, c_ = sce / (float)npos;
  if (o.vb) o.vb[b] = v;
  if (o.ce) o.ce[b] = c_;
  const float l = loss_type == D3PM_LOSS_KL ? v : (loss_type == D3PM_LOSS_CE ? c_ : v + hybrid_coeff * c_);
  if (o.loss) o.loss[b] = l;
  // One float atomic per batch element, performed at the memory side: no fence, no extra launch.  (A last-workgroup
  // reduction needs device-scope release fences, which on a multi-XCD part write back the whole L2: 18.8 us against
  // 10.5; a separate one-workgroup launch costs 3.6 us plus a dependent graph node.)  The order of the B additions
  // varies from run to run, so the mean is reproducible to float32 rounding of a B-term sum, not bit for bit.
  if (o.mean) __hip_atomic_fetch_add(o.mean, l * o.inv_B, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// Debug builds (-DD3PM_TIMELINE) record the 100 MHz wall clock at a few points of
// the first and last wave of every workgroup; d3pm_debug_timeline copies them out.
#ifdef D3PM_TIMELINE
constexpr int TL_POINTS = 9;
__device__ unsigned long long g_timeline[256][16][TL_POINTS];
#define D3PM_TL_DECL unsigned long long tl_[TL_POINTS] = {}
#ifdef D3PM_TL_SHADERCLK
#define D3PM_CLK() __builtin_amdgcn_s_memtime()
#else
#define D3PM_CLK() wall_clock64()
#endif
#define D3PM_TL(i) tl_[i] = D3PM_CLK()
// waits until the register holds its loaded value, then records the clock
#define D3PM_TL_AFTER(i, reg) do { asm volatile("" ::"v"(reg)); tl_[i] = D3PM_CLK(); } while (0)
#define D3PM_TL_FLUSH                                                                                    \
  do {                                                                                                   \
    if (blockIdx.x < 256 && (threadIdx.x & 63) == 0)                                                     \
      for (int i_ = 0; i_ < TL_POINTS; ++i_) g_timeline[blockIdx.x][threadIdx.x >> 6][i_] = tl_[i_];     \
  } while (0)
#else
#define D3PM_TL_DECL do {} while (0)
#define D3PM_TL(i) do {} while (0)
#define D3PM_TL_AFTER(i, reg) do {} while (0)
#define D3PM_TL_FLUSH do {} while (0)
#endif

template <class L, bool GRAD>
__global__ __launch_bounds__(loss_max_threads<L>()) void k_hybrid_loss(
    ImgArgs a, const float* __restrict__ logits, long ld, int loss_type, float hybrid_coeff, float grad_scale,
    const float* __restrict__ grad_out, LossOut out, float* __restrict__ dlogits) {
  __shared__ float red[2 * MAX_WAVES];
  // by class: column of Q_t (diag+eps, off+eps), Qbar_{t-1} (diag-off, off+eps)
  __shared__ float s_ad[256], s_ao[256], s_bdo[256], s_boe[256];
  constexpr int PIPE = pipe_depth<L>();
  const int K = a.tb.K;
  const float eps = a.tb.eps;
  const int b = blockIdx.x / a.chunks, chunk = blockIdx.x % a.chunks;
  const int g = threadIdx.x % L::G, slot = threadIdx.x / L::G;
  const unsigned tpp = blockDim.x / L::G;     // tokens per pass
  // Rows of this batch element through wave-uniform bases and 32-bit byte
  // offsets (the host checks npos * ld < 2^30).  Tokens are non-negative and
  // below 2^31, so the low dword of an int64 token is the token.
  const char* __restrict__ lrow = (const char*)(logits + (size_t)b * a.npos * ld);
  char* __restrict__ drow = GRAD ? (char*)(dlogits + (size_t)b * a.npos * ld) : nullptr;
  const char* __restrict__ x0row = (const char*)a.x0 + (size_t)b * a.npos * a.x0_bytes;
  const char* __restrict__ xtrow = (const char*)a.xt + (size_t)b * a.npos * a.xt_bytes;
  const unsigned x0sh = a.x0_bytes == 8 ? 3 : 2, xtsh = a.xt_bytes == 8 ? 3 : 2;
  const unsigned p1 = (unsigned)min(a.npos, (chunk + 1) * a.chunk_pos);
  const unsigned pbeg = (unsigned)(chunk * a.chunk_pos) + slot;
  const unsigned row_bytes = (unsigned)ld * 4u;

  // Start-up order: the timestep (a scalar load), then the few digest rows that
  // depend on it, then the first rows of logits; vector loads return in issue
  // order, so the digests must not queue behind the row burst.
  if (blockIdx.x == 0 && threadIdx.x == 0 && out.rng_counter) *out.rng_counter += out.rng_step;
  D3PM_TL_DECL;
  D3PM_TL(0);
  const int t = load_index(a.t, b, a.t_bytes, a.tb.T);
  const bool need_post = (t > 0) && loss_type != D3PM_LOSS_CE;
  D3PM_TL_AFTER(1, t);
  float bdo[L::V], boe[L::V], bo[L::V];
  float sum_boe = 0, sum_bo = 0;
  float f_ad = 0, f_ao = 0, f_d = 0, f_o = 0;
  if (need_post) {
    load_digest<L>(a.tb.qbar_digest, t - 1, 0, K, g, bdo);     // diag for now
    load_digest<L>(a.tb.qbar_digest, t - 1, 1, K, g, bo);
    if (threadIdx.x < K) {                   // K <= blockDim.x (checked by the host)
      f_ad = dg(a.tb.qone_digest, t, 0, K)[threadIdx.x];
      f_ao = dg(a.tb.qone_digest, t, 1, K)[threadIdx.x];
      f_d = dg(a.tb.qbar_digest, t - 1, 0, K)[threadIdx.x];
      f_o = dg(a.tb.qbar_digest, t - 1, 1, K)[threadIdx.x];
    }
  }

  float zz[PIPE][L::V];
  int x0s[PIPE], xts[PIPE];
  auto issue_load = [&](int u, unsigned pos) {
    load_slots_off<L>(lrow, pos * row_bytes, g, K, -INFINITY, zz[u]);
    x0s[u] = *reinterpret_cast<const int*>(x0row + (pos << x0sh));
    xts[u] = *reinterpret_cast<const int*>(xtrow + (pos << xtsh));
  };
#pragma unroll
  for (int u = 0; u < PIPE; ++u) {
    const unsigned pos = pbeg + u * tpp;
    if (pos < p1) issue_load(u, pos);
  }
  // The logits of a token's two special classes (x0, x_t) are read again from its
  // row, one stage ahead of use and before the previous stage's stores, so that
  // waiting for them never waits for a store.
  float sp0[PIPE], spt[PIPE];
  auto issue_special = [&](int u, unsigned pos) {
    sp0[u] = *reinterpret_cast<const float*>(lrow + (pos * row_bytes + 4u * (unsigned)x0s[u]));
    spt[u] = *reinterpret_cast<const float*>(lrow + (pos * row_bytes + 4u * (unsigned)xts[u]));
  };

  // per-class constants of Qbar_{t-1} for this lane's slots, and their sums over all classes
  if (need_post) {
    D3PM_TL_AFTER(2, bo[L::V - 1]);
#pragma unroll
    for (int v = 0; v < L::V; ++v) {
      const bool ok = L::VEC || L::cls(g, v) < K;
      bdo[v] = ok ? bdo[v] - bo[v] : 0.0f;
      boe[v] = ok ? bo[v] + eps : 1.0f;        // masked slots: f2 = 1
      bo[v] = ok ? bo[v] : 0.0f;
      sum_boe += ok ? boe[v] : 0.0f;
      sum_bo += bo[v];
    }
    sum_boe = group_sum<L::G>(sum_boe);
    sum_bo = group_sum<L::G>(sum_bo);
    if (threadIdx.x < K) {
      s_ad[threadIdx.x] = f_ad + eps;
      s_ao[threadIdx.x] = f_ao + eps;
      s_bdo[threadIdx.x] = f_d - f_o;
      s_boe[threadIdx.x] = f_o + eps;
    }
    __syncthreads();
  }
  D3PM_TL(3);
  if (pbeg < p1) issue_special(0, pbeg);
  const float gb = grad_scale * ((GRAD && grad_out) ? grad_out[b] : 1.0f);
  const float cnorm = gb / ((float)a.npos * LN2);
  // weights of d(vb)/dz and d(ce)/dz in the requested loss
  const float w_vb = (loss_type == D3PM_LOSS_CE) ? 0.0f : cnorm;
  const float w_ce = (loss_type == D3PM_LOSS_KL) ? 0.0f : (loss_type == D3PM_LOSS_CE ? cnorm : hybrid_coeff * cnorm);
  float acc_vb = 0, acc_ce = 0;          // per-lane partial sums, in bits; lane 0 of a group carries the token
  D3PM_TL_AFTER(4, w_ce);
  for (unsigned pos0 = pbeg; pos0 < p1; pos0 += PIPE * tpp) {
#pragma unroll
    for (int u = 0; u < PIPE; ++u) {
      const unsigned pos = pos0 + u * tpp;
      if (pos >= p1) break;              // whole lane groups leave together
      const int x0 = x0s[u];
      const int xt = xts[u];
      const float z_x0 = sp0[u], z_xt = spt[u];
      float z[L::V];
#pragma unroll
      for (int v = 0; v < L::V; ++v) z[v] = zz[u][v];
      if (u == 0 && pos0 == pbeg) D3PM_TL_AFTER(5, z_xt);
      {                                  // special classes of the next stage, then refill this stage's slot
        const unsigned nxt1 = pos + tpp, nxt = pos + PIPE * tpp;
        if (nxt1 < p1) issue_special((u + 1) % PIPE, nxt1);
        if (nxt < p1) issue_load(u, nxt);
      }
      // ---- reductions 1 and 2: softmax of z
      float m = z[0];
#pragma unroll
      for (int v = 1; v < L::V; ++v) m = fmaxf(m, z[v]);
      m = group_max<L::G>(m);
      const float mneg = -m * LOG2E;
      float pi[L::V], s = 0;
#pragma unroll
      for (int v = 0; v < L::V; ++v) { pi[v] = fast_exp2(fmaf(z[v], LOG2E, mneg)); s += pi[v]; }
      s = group_sum<L::G>(s);
      const float inv = fast_rcp(s);
      const float ce_bits = fmaf(m - z_x0, LOG2E, fast_log2(s));     // -log2 softmax(z)[x0]
      const float tok = (g == 0) ? 1.0f : 0.0f;
      acc_ce = fmaf(tok, ce_bits, acc_ce);
      const int d0 = x0 - L::cls(g, 0), dt = xt - L::cls(g, 0);       // slot of x0 / x_t in this lane, if any
      constexpr int CS = L::VEC ? 1 : L::G;                           // class stride between slots
      float gz[L::V];
      if (!need_post) {
        acc_vb = fmaf(tok, ce_bits, acc_vb);
        if constexpr (GRAD) {
          const float w = (t == 0 ? w_vb : 0.0f) + w_ce;
          const float wi = w * inv;
#pragma unroll
          for (int v = 0; v < L::V; ++v) gz[v] = fmaf(pi[v], wi, (d0 == v * CS) ? -w : 0.0f);
        }
      } else {
        const float a_d = s_ad[xt], a_o = s_ao[xt], da = a_d - a_o;
        const float bdo_x0 = s_bdo[x0], boe_x0 = s_boe[x0], bdo_xt = s_bdo[xt], boe_xt = s_boe[xt];
        const float same = (x0 == xt) ? 1.0f : 0.0f;
        // ---- reduction 3: generic-class sums
        float pr[L::V], Sf = 0, Rp = 0;
#pragma unroll
        for (int v = 0; v < L::V; ++v) {
          pi[v] *= inv;
          const float f2 = fmaf(bdo[v], pi[v], boe[v]);
          pr[v] = boe[v] * fast_rcp(f2);                // boe_j / f2_j
          Sf += (L::VEC || L::cls(g, v) < K) ? f2 : 0.0f;
          if constexpr (GRAD) Rp = fmaf(pr[v], bo[v], Rp);
        }
        Sf = group_sum<L::G>(Sf);
        if constexpr (GRAD) Rp = group_sum<L::G>(Rp);
        // special classes
        const float pi_x0 = fast_exp2(fmaf(z_x0, LOG2E, mneg)) * inv, pi_xt = fast_exp2(fmaf(z_xt, LOG2E, mneg)) * inv;
        const float f2_x0 = fmaf(bdo_x0, pi_x0, boe_x0), f2_xt = fmaf(bdo_xt, pi_xt, boe_xt);
        const float rf_x0 = fast_rcp(f2_x0), rf_xt = fast_rcp(f2_xt);
        const float b_x0 = boe_x0 + bdo_x0, b_xt = fmaf(same, bdo_xt, boe_xt);
        const float a_x0 = fmaf(same, da, a_o);
        const float Zt = fmaf(a_o, sum_boe + bdo_x0, da * b_xt);
        const float Zm = fmaf(a_o, Sf, da * f2_xt);
        const float izt = fast_rcp(Zt), izm = fast_rcp(Zm);
        // kl = sum_j q_j log2(q_j / p_j), q_j / p_j = (b_j / f2_j) (Zm / Zt): every log is of a ratio that
        // is close to one wherever q_j is large, so nothing cancels
        const float zr = Zm * izt;
        const float A1 = w_vb * a_o * izm, A2 = w_vb * a_o * izt;
        // ---- reduction 4: kl sum, <pi, g pi>
        float h[L::V], S1 = 0, D = 0;
#pragma unroll
        for (int v = 0; v < L::V; ++v) {
          S1 = fmaf(boe[v], fast_log2(pr[v] * zr), S1);       // masked slots: boe = 1 but pr * zr != 1
          if constexpr (!L::VEC) S1 = (L::cls(g, v) < K) ? S1 : S1 - fast_log2(pr[v] * zr);
          if constexpr (GRAD) {
            h[v] = fmaf(-A2, pr[v], A1) * bdo[v];
            D = fmaf(pi[v], h[v], D);
          }
        }
        S1 = group_sum<L::G>(S1);
        if constexpr (GRAD) D = group_sum<L::G>(D);
        float kls = a_o * S1;
        kls = fmaf(-a_o * boe_x0, fast_log2(boe_x0 * rf_x0 * zr), kls);
        kls = fmaf(a_x0 * b_x0, fast_log2(b_x0 * rf_x0 * zr), kls);
        kls = fmaf((1.0f - same) * da * boe_xt, fast_log2(boe_xt * rf_xt * zr), kls);
        acc_vb = fmaf(tok, kls * izt, acc_vb);
        if constexpr (GRAD) {
          const float d_x0 = -A2 * bdo_x0 * rf_x0;                       // added to gf2 at class x0
          const float d_xt = w_vb * da * (izm - b_xt * rf_xt * izt);     // added to gf2 at class x_t
          const float bo_x0 = boe_x0 - eps, bo_xt = boe_xt - eps;
          const float R = fmaf(A1, sum_bo, fmaf(-A2, Rp, fmaf(d_x0, bo_x0, d_xt * bo_xt)));
          const float p0 = pi_x0 * d_x0 * bdo_x0, pt = pi_xt * d_xt * bdo_xt;   // pi_j * (extra g pi_j) at x0, x_t
          const float dot = R + D + p0 + pt;
          const float C = R - dot + w_ce;
          // additive patches at the two special classes (one class when x0 == x_t)
          const float add0 = p0 - w_ce + same * pt, addt = (1.0f - same) * pt;
#pragma unroll
          for (int v = 0; v < L::V; ++v) {
            const float patch = (d0 == v * CS) ? add0 : ((dt == v * CS) ? addt : 0.0f);
            gz[v] = fmaf(pi[v], h[v] + C, patch);
          }
        }
      }
      if constexpr (GRAD) {
        store_slots_off<L>(drow, pos * row_bytes, g, K, gz);
        if (!L::VEC && ld > K) zero_tail<L>(reinterpret_cast<float*>(drow + (size_t)pos * row_bytes), g, K, ld);
      }
      if (u == 0 && pos0 == pbeg) D3PM_TL(6);
    }
  }
  D3PM_TL(7);
  // workgroup sums (fixed order: lanes by butterfly, waves in index order)
  acc_vb = wave_sum(acc_vb);
  acc_ce = wave_sum(acc_ce);
  const int wave = threadIdx.x >> 6, nw = blockDim.x >> 6;
  if ((threadIdx.x & 63) == 0) { red[2 * wave] = acc_vb; red[2 * wave + 1] = acc_ce; }
  __syncthreads();
  if (threadIdx.x == 0) {
    float svb = 0, sce = 0;
    for (int w = 0; w < nw; ++w) { svb += red[2 * w]; sce += red[2 * w + 1]; }
    if (a.chunks == 1) {
      write_losses(out, b, svb, sce, a.npos, loss_type, hybrid_coeff);
    } else {
      out.part[((size_t)b * a.chunks + chunk) * 2 + 0] = svb;
      out.part[((size_t)b * a.chunks + chunk) * 2 + 1] = sce;
    }
  }

  D3PM_TL(8);
  D3PM_TL_FLUSH;
}
#ifdef D3PM_TIMELINE
extern "C" int d3pm_debug_timeline(void* host_out) {
  return (int)hipMemcpyFromSymbol(host_out, HIP_SYMBOL(g_timeline), sizeof(g_timeline));
}
#endif

// ------------------------------------------------------------------ fused hybrid loss, one token per lane
// For K <= 32 (rows of at most 128 bytes) a lane takes a whole token: its K classes sit in registers, so there is
// no cross-lane reduction, no per-token scalar work repeated on several lanes, and every per-class constant of the
// timestep is wave-uniform (scalar registers).  Rows are moved between global memory and that layout through a
// wave-private LDS image: coalesced 16-byte pieces in, one padded row per lane out (row stride = row + 16 bytes,
// which spreads a 16-lane read group over all banks), and the same image in reverse for the gradient rows.
// Arithmetic as in k_hybrid_loss with three simplifications that hold for every row-stochastic column-structured
// table: diag_j - off_j is one value for all classes (every row sums to one), hence sum_j f2_j needs no reduction
// (sum_j pi_j = 1), and the term R = sum_j gf2_j off_j, which enters g pi_j and <pi, g pi> alike, drops out of the
// softmax backward.  The softmax stays unnormalised (e_j = 2^(z_j - m)); 1 / sum e is folded into the per-token
// scalars.  The two special classes (x0, x_t) are read from and patched into the lane's own LDS row.
template <int KC> struct TplCfg {
  static constexpr int RB = 4 * KC, PIECES = RB / 16, STRIDE = RB + 16;
  static_assert(KC % 4 == 0 && KC <= 32 && (PIECES % 2) == 0, "row of 2, 4, 6 or 8 16-byte pieces");
};

__device__ __forceinline__ float lane_bcast(float v, int lane) {
  return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), lane));
}
__device__ __forceinline__ void wave_lds_fence() {
  // LDS operations of one wave complete in order; this only stops the compiler from moving them across
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

template <int KC, bool GRAD, int MAXT>
__global__ __launch_bounds__(MAXT) void k_hybrid_loss_tpl(
    ImgArgs a, const float* __restrict__ logits, int loss_type, float hybrid_coeff, float grad_scale,
    const float* __restrict__ grad_out, LossOut out, float* __restrict__ dlogits) {
  using C = TplCfg<KC>;
  extern __shared__ __align__(16) char tpl_rows[];          // [waves][64][STRIDE]
  __shared__ float red[2 * MAX_WAVES];
  // column x_t of Q_t (+eps) and off + eps of Qbar_{t-1}, by class: a copy per wave, so that no wave waits for another
  __shared__ float s_tab[MAX_WAVES][3][32];
  const float eps = a.tb.eps;
  const int b = blockIdx.x / a.chunks, chunk = blockIdx.x % a.chunks;
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const unsigned p1 = (unsigned)min(a.npos, (chunk + 1) * a.chunk_pos);
  const unsigned pbeg = (unsigned)(chunk * a.chunk_pos);
  const char* __restrict__ lrow = (const char*)(logits + (size_t)b * a.npos * KC);
  char* __restrict__ drow = GRAD ? (char*)(dlogits + (size_t)b * a.npos * KC) : nullptr;
  const char* __restrict__ x0row = (const char*)a.x0 + (size_t)b * a.npos * a.x0_bytes;
  const char* __restrict__ xtrow = (const char*)a.xt + (size_t)b * a.npos * a.xt_bytes;
  const unsigned x0sh = a.x0_bytes == 8 ? 3 : 2, xtsh = a.xt_bytes == 8 ? 3 : 2;
  char* const img = tpl_rows + (size_t)wave * 64 * C::STRIDE;
  char* const myrow = img + lane * C::STRIDE;

  if (blockIdx.x == 0 && threadIdx.x == 0 && out.rng_counter) *out.rng_counter += out.rng_step;

  D3PM_TL_DECL;
  D3PM_TL(0);
  // piece h of this lane in the coalesced order: row (of the wave's 64) and byte offset inside the row
  v4f_nt stage[C::PIECES];
  int x0 = 0, xt = 0;
  auto issue = [&](unsigned base) {           // base: first position of this wave's 64 tokens
    const unsigned last = p1 - 1 - base;      // rows past the end re-read the last one (never stored)
    const unsigned mine = min((unsigned)lane, last);
    x0 = *reinterpret_cast<const int*>(x0row + ((base + mine) << x0sh));
    xt = *reinterpret_cast<const int*>(xtrow + ((base + mine) << xtsh));
#pragma unroll
    for (int h = 0; h < C::PIECES; ++h) {
      const unsigned q = h * 64 + lane, r = min(q / C::PIECES, last), j = q % C::PIECES;
      stage[h] = *reinterpret_cast<const v4f_nt*>(lrow + ((base + r) * C::RB + j * 16));
    }
  };
  unsigned base = pbeg + wave * 64;
  // The timestep is requested first: loads retire in issue order, and the digest rows that depend on it should be
  // on their way while the logits rows arrive.
  const int t = load_index(a.t, b, a.t_bytes, a.tb.T);
  if (base < p1) issue(base);
  D3PM_TL_AFTER(1, t);
  const bool need_post = (t > 0) && loss_type != D3PM_LOSS_CE;
  // per-class constants of the timestep, wave-uniform: off_j + eps of Qbar_{t-1} by class, their sum, diag - off
  // (requested here, used after the first rows' softmax: the timestep -> digest chain is the longest latency of
  // the kernel's start, and that softmax does not need it)
  float boe[KC];
  unsigned same_w = 0;          // bit q: classes 4q .. 4q+3 have one off-diagonal weight (all of them for uniform Q)
  float bdo = 0, sum_boe = 0;
  float vd = 0, vo = 0, v_ad = 0, v_ao = 0;
  if (need_post && lane < KC) {
    vd = dg(a.tb.qbar_digest, t - 1, 0, KC)[lane];
    vo = dg(a.tb.qbar_digest, t - 1, 1, KC)[lane];
    v_ad = dg(a.tb.qone_digest, t, 0, KC)[lane];
    v_ao = dg(a.tb.qone_digest, t, 1, KC)[lane];
  }
  const float gb = grad_scale * ((GRAD && grad_out) ? grad_out[b] : 1.0f);
  const float cnorm = gb / ((float)a.npos * LN2);
  const float w_vb = (loss_type == D3PM_LOSS_CE) ? 0.0f : cnorm;
  const float w_ce = (loss_type == D3PM_LOSS_KL) ? 0.0f : (loss_type == D3PM_LOSS_CE ? cnorm : hybrid_coeff * cnorm);
  float acc_vb = 0, acc_ce = 0;
  bool first = true;
  for (;; base += blockDim.x) {
    const bool work = base < p1;              // uniform over the wave
    float e[KC];
    float m = 0, inv = 0, l2s = 0, mneg = 0;
    if (work) {
      // ---- rows into the lane-per-token layout
#pragma unroll
      for (int h = 0; h < C::PIECES; ++h) {
        const unsigned q = h * 64 + lane;
        *reinterpret_cast<v4f_nt*>(img + (q / C::PIECES) * C::STRIDE + (q % C::PIECES) * 16) = stage[h];
      }
      if (first) D3PM_TL(2);
      wave_lds_fence();
#pragma unroll
      for (int j = 0; j < C::PIECES; ++j) {
        const v4f_nt q = *reinterpret_cast<const v4f_nt*>(myrow + 16 * j);
        e[4 * j] = q.x; e[4 * j + 1] = q.y; e[4 * j + 2] = q.z; e[4 * j + 3] = q.w;
      }
      // ---- phase A: softmax statistics (no timestep needed)
      m = e[0];
#pragma unroll
      for (int j = 1; j < KC; ++j) m = fmaxf(m, e[j]);
      mneg = -m * LOG2E;
      float s = 0;
#pragma unroll
      for (int j = 0; j < KC; ++j) { e[j] = fast_exp2(fmaf(e[j], LOG2E, mneg)); s += e[j]; }
      inv = fast_rcp(s);
      l2s = fast_log2(s);
      if (first) D3PM_TL_AFTER(3, l2s);
    }
    if (first) {
      first = false;
      if (need_post) {
        const float vboe = vo + eps;
        if (lane < KC) {
          s_tab[wave][0][lane] = v_ad + eps;
          s_tab[wave][1][lane] = v_ao + eps;
          s_tab[wave][2][lane] = vboe;
        }
        bdo = lane_bcast(vd - vo, 0);
        // every row of Qbar_{t-1} sums to one: diag_i + sum_{j != i} off_j = 1, so sum_j off_j = 1 - (diag - off) --
        // no reduction over the classes
        sum_boe = (1.0f - bdo) + (float)KC * eps;
        int bits[KC];                           // scalar registers: the comparisons below are scalar too
#pragma unroll
        for (int j = 0; j < KC; ++j) {
          bits[j] = __builtin_amdgcn_readlane(__builtin_bit_cast(int, vboe), j);
          boe[j] = __builtin_bit_cast(float, bits[j]);
        }
        unsigned sw = 0;
#pragma unroll
        for (int q = 0; q < KC / 4; ++q) {
          const bool eq = bits[4 * q] == bits[4 * q + 1] && bits[4 * q] == bits[4 * q + 2] && bits[4 * q] == bits[4 * q + 3];
          sw |= (eq ? 1u : 0u) << q;
        }
        same_w = __builtin_amdgcn_readfirstlane(sw);
      }
      wave_lds_fence();                       // this wave's table copy
      D3PM_TL(4);
    }
    if (!work) break;
    // ---- phase B
    x0 = clamp_index(x0, KC);                             // out-of-range tokens are clamped: nothing faults
    xt = clamp_index(xt, KC);
    const float z_x0 = *reinterpret_cast<const float*>(myrow + 4 * x0);
    const float z_xt = *reinterpret_cast<const float*>(myrow + 4 * xt);
    const bool live = base + lane < p1;
    const float ce_bits = fmaf(m - z_x0, LOG2E, l2s);     // -log2 softmax(z)[x0]
    acc_ce += live ? ce_bits : 0.0f;
    // gradient values of the two special classes, formed from per-token scalars and written over the row's
    // generic values after the row has gone to LDS (plain stores in program order: no read-modify-write)
    float g_x0 = 0, g_xt = 0;
    const float e_x0 = fast_exp2(fmaf(z_x0, LOG2E, mneg)), e_xt = fast_exp2(fmaf(z_xt, LOG2E, mneg));
    if (!need_post) {
      acc_vb += live ? ce_bits : 0.0f;
      if constexpr (GRAD) {
        const float w = (t == 0 ? w_vb : 0.0f) + w_ce, wi = w * inv;
#pragma unroll
        for (int j = 0; j < KC; ++j) e[j] *= wi;
        g_x0 = fmaf(e_x0, wi, -w);
        g_xt = (x0 == xt) ? g_x0 : e_xt * wi;
      }
    } else {
      const float a_d = s_tab[wave][0][xt], a_o = s_tab[wave][1][xt], da = a_d - a_o;
      const float boe_x0 = s_tab[wave][2][x0], boe_xt = s_tab[wave][2][xt];
      const float same = (x0 == xt) ? 1.0f : 0.0f;
      const float bdoi = bdo * inv;
      const float pi_x0 = e_x0 * inv, pi_xt = e_xt * inv;
      const float f2_x0 = fmaf(bdo, pi_x0, boe_x0), f2_xt = fmaf(bdo, pi_xt, boe_xt);
      const float rf_x0 = fast_rcp(f2_x0), rf_xt = fast_rcp(f2_xt);
      const float b_x0 = boe_x0 + bdo, b_xt = fmaf(same, bdo, boe_xt);
      const float a_x0 = fmaf(same, da, a_o);
      const float Zt = fmaf(a_o, sum_boe + bdo, da * b_xt);
      const float Zm = fmaf(a_o, sum_boe + bdo, da * f2_xt);        // sum_j f2_j = sum_boe + bdo sum_j pi_j
      const float izt = fast_rcp(Zt), izm = fast_rcp(Zm);
      const float zr = Zm * izt;
      float pr[KC], E = 0, S1 = 0;
#pragma unroll
      for (int q = 0; q < KC / 4; ++q) {
#pragma unroll
        for (int j = 4 * q; j < 4 * q + 4; ++j) {
          const float f2 = fmaf(e[j], bdoi, boe[j]);
          pr[j] = boe[j] * fast_rcp(f2);                    // boe_j / f2_j
          if constexpr (GRAD) E = fmaf(e[j], pr[j], E);
        }
        if ((same_w >> q) & 1) {
          // four classes of one weight w: sum_j w log2 pr_j = w log2 prod_j pr_j -- one logarithm.  Every pr_j lies in
          // [off / (off + diag - off), 1] with off >= eps = 1e-6, so the product stays above 1e-24, inside the
          // float32 range.  (The factor zr of every class is taken out of the sum: log2(zr) sum_j w_j below.  A term
          // w log2 P is bounded by 4 w log2(1 / w), so taking the logarithms apart costs no accuracy that shows.)
          const float P = (pr[4 * q] * pr[4 * q + 1]) * (pr[4 * q + 2] * pr[4 * q + 3]);
          S1 = fmaf(boe[4 * q], fast_log2(P), S1);
        } else {
#pragma unroll
          for (int j = 4 * q; j < 4 * q + 4; ++j) S1 = fmaf(boe[j], fast_log2(pr[j]), S1);
        }
      }
      S1 = fmaf(sum_boe, fast_log2(zr), S1);                // sum_j boe_j log2(pr_j zr)
      float kls = a_o * S1;
      kls = fmaf(-a_o * boe_x0, fast_log2(boe_x0 * rf_x0 * zr), kls);
      kls = fmaf(a_x0 * b_x0, fast_log2(b_x0 * rf_x0 * zr), kls);
      kls = fmaf((1.0f - same) * da * boe_xt, fast_log2(boe_xt * rf_xt * zr), kls);
      acc_vb += live ? kls * izt : 0.0f;
      if constexpr (GRAD) {
        const float A1 = w_vb * a_o * izm, A2 = w_vb * a_o * izt;
        const float d_x0 = -A2 * bdo * rf_x0;                          // added to gf2 at class x0
        const float d_xt = w_vb * da * (izm - b_xt * rf_xt * izt);     // added to gf2 at class x_t
        const float p0 = pi_x0 * d_x0 * bdo, pt = pi_xt * d_xt * bdo;  // pi_j * (extra g pi_j) at x0, x_t
        // <pi, g pi> without R: bdo (A1 sum pi - A2 sum pi_j pr_j) + p0 + pt
        const float D = bdo * (A1 - A2 * inv * E);
        const float Cc = w_ce - (D + p0 + pt);
        const float nA2i = -A2 * bdoi, Ci = inv * fmaf(bdo, A1, Cc);
#pragma unroll
        for (int j = 0; j < KC; ++j) e[j] *= fmaf(nA2i, pr[j], Ci);
        g_x0 = fmaf(e_x0, fmaf(nA2i, boe_x0 * rf_x0, Ci), p0 - w_ce);
        g_xt = fmaf(e_xt, fmaf(nA2i, boe_xt * rf_xt, Ci), pt);
        if (x0 == xt) g_x0 = g_xt = g_x0 + pt;
      }
    }
    D3PM_TL_AFTER(5, e[KC - 1]);
    if constexpr (GRAD) {
      // ---- gradient row back through the LDS image; the two special classes are patched in the lane's own row
#pragma unroll
      for (int j = 0; j < C::PIECES; ++j)
        *reinterpret_cast<v4f_nt*>(myrow + 16 * j) = v4f_nt{e[4 * j], e[4 * j + 1], e[4 * j + 2], e[4 * j + 3]};
      *reinterpret_cast<float*>(myrow + 4 * x0) = g_x0;
      *reinterpret_cast<float*>(myrow + 4 * xt) = g_xt;
      wave_lds_fence();
#ifdef D3PM_TIMELINE
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      D3PM_TL(6);
#endif
      const unsigned nrows = p1 - base;       // rows of this wave that exist
      if (nrows >= 64) {                      // (uniform over the wave) the whole image, straight-line
#pragma unroll
        for (int h = 0; h < C::PIECES; ++h) {
          const unsigned q = h * 64 + lane, r = q / C::PIECES, j = q % C::PIECES;
          const v4f_nt v = *reinterpret_cast<const v4f_nt*>(img + r * C::STRIDE + j * 16);
          __builtin_nontemporal_store(v, reinterpret_cast<v4f_nt*>(drow + ((base + r) * C::RB + j * 16)));
        }
      } else {
#pragma unroll
        for (int h = 0; h < C::PIECES; ++h) {
          const unsigned q = h * 64 + lane, r = q / C::PIECES, j = q % C::PIECES;
          const v4f_nt v = *reinterpret_cast<const v4f_nt*>(img + r * C::STRIDE + j * 16);
          if (r < nrows)
            __builtin_nontemporal_store(v, reinterpret_cast<v4f_nt*>(drow + ((base + r) * C::RB + j * 16)));
        }
      }
      D3PM_TL(7);
    }
    D3PM_TL(8);
    if (base + blockDim.x >= p1) break;       // (uniform over the wave)
    wave_lds_fence();
    issue(base + blockDim.x);
  }
  acc_vb = wave_sum(acc_vb);
  acc_ce = wave_sum(acc_ce);
  const int nw = blockDim.x >> 6;
  if (lane == 0) { red[2 * wave] = acc_vb; red[2 * wave + 1] = acc_ce; }
  __syncthreads();
  if (threadIdx.x == 0) {
    float svb = 0, sce = 0;
    for (int w = 0; w < nw; ++w) { svb += red[2 * w]; sce += red[2 * w + 1]; }
    if (a.chunks == 1) {
      write_losses(out, b, svb, sce, a.npos, loss_type, hybrid_coeff);
    } else {
      out.part[((size_t)b * a.chunks + chunk) * 2 + 0] = svb;
      out.part[((size_t)b * a.chunks + chunk) * 2 + 1] = sce;
    }
  }
  D3PM_TL_FLUSH;
}

// ------------------------------------------------------------------ fused hybrid loss, K = 32, two lanes per token
// The same arithmetic with a token on a pair of lanes (lane 2p + h holds classes 16h .. 16h + 15 of the wave's
// token p): a wave takes 32 tokens, so a batch element of 512 positions is 16 waves -- four per SIMD instead of two,
// which is what the vector ALUs need to issue back to back, and a finer pipeline of load / compute / store phases
// over the chip.  The price: the per-token scalar work runs on both lanes, the per-class constants are per lane
// (vector registers) and four sums per token cross the pair (one DPP operation each).  The LDS image holds the
// wave's 32 rows at a stride of 144 bytes (an odd number of 16-byte slots), a lane's half at byte 64 h of its row.
// NEXT: the launch also draws the timesteps and the noisy tokens of the NEXT batch (what d3pm_q_sample_draw_t
// does, same Philox stream, same result) while its own rows are still on their way from memory: a training loop
// whose batches are staged ahead then needs one launch per step instead of two.  The first wave builds the
// per-timestep tables of the batch element's next timestep; its digest loads go out before its row loads, so they
// do not queue behind them.
struct NextQ {
  const void* x0; int x0_bytes;                // clean tokens of the next batch, [B, npos]
  uint64_t seed, offset; const uint64_t* offset_dev;
  int32_t* t_out; int64_t* x_t;                // [B] and [B, npos]
};

// ONE: the workgroup's threads cover the whole batch element (npos <= blockDim.x / 2, every wave has tokens): the
// pass loop is gone at compile time, and with it the loop-carried copies of the row registers and the zeroed rows
// of the wave-without-work path (48 of 435 vector instructions per wave at npos = 512).
template <bool GRAD, bool NEXT = false, bool ONE = false>
__global__ __launch_bounds__(1024) void k_hybrid_loss_pair32(
    ImgArgs a, const float* __restrict__ logits, int loss_type, float hybrid_coeff, float w_vb0,
    const float* __restrict__ grad_out, LossOut out, float* __restrict__ dlogits, float w_ce0, NextQ nx) {
  // w_vb0 / w_ce0: the weights of the two loss terms in the gradient, grad_scale / (npos ln 2) times 1, 0 or the
  // hybrid coefficient as the loss type says, formed on the host: the kernel runs its set-up once per wave of 32 tokens
  constexpr int KC = 32, HC = 16, RB = 4 * KC, TS = RB + 16, PW = 16;         // PW: waves per workgroup at most
  extern __shared__ __align__(16) char pair_rows[];         // [waves][32 rows][TS]
  __shared__ __align__(8) float red[PW * 4 * 2];            // (vb, ce) of each 16-lane row of each wave
  __shared__ __align__(16) float s_tab[PW][3][KC];          // as in k_hybrid_loss_tpl: a copy per wave
  __shared__ __align__(16) float s_w4[PW][KC / 4];          // (off + eps)^4 of the first class of every group of four
  const float eps = a.tb.eps;
  const int b = blockIdx.x;                                 // one workgroup per batch element
  // (the wave index through the scalar unit: everything derived from it -- the wave's LDS image, its block of rows,
  // the full-wave tests -- is then scalar arithmetic and scalar branches)
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int half = lane & 1, ptok = lane >> 1;
  const unsigned p1 = (unsigned)a.npos;
  const unsigned step = blockDim.x >> 1;                    // tokens per pass of the workgroup
  const char* __restrict__ lrow = (const char*)(logits + (size_t)b * a.npos * KC);
  char* __restrict__ drow = GRAD ? (char*)(dlogits + (size_t)b * a.npos * KC) : nullptr;
  const char* __restrict__ x0row = (const char*)a.x0 + (size_t)b * a.npos * a.x0_bytes;
  const char* __restrict__ xtrow = (const char*)a.xt + (size_t)b * a.npos * a.xt_bytes;
  const unsigned x0sh = a.x0_bytes == 8 ? 3 : 2, xtsh = a.xt_bytes == 8 ? 3 : 2;
  // The wave's image: 32 rows of 128 bytes at a stride of 144 (nine 16-byte slots: the 8 lanes of an access group,
  // four tokens' two halves, 9 p + 4 h + j, fall on eight different slots), a lane's half at byte 64 h of its row.
  char* const img = pair_rows + (size_t)wave * 32 * TS;
  char* const tokrow = img + ptok * TS;                     // class c of this token: tokrow + 4 c
  char* const myhalf = tokrow + 64 * half;
  // piece h of this lane in the wave's block of rows (16 bytes at byte 1024 h + 16 lane of it) belongs to row
  // 8 h + lane / 8, 16-byte piece lane % 8 of it
  char* const piece0 = img + (lane >> 3) * TS + (lane & 7) * 16;
  auto cls_off = [](int c) { return 4 * c; };

  if (blockIdx.x == 0 && threadIdx.x == 0 && out.rng_counter) *out.rng_counter += out.rng_step;

  D3PM_TL_DECL;
  D3PM_TL(0);
  v4f_nt stage[4];
  int x0 = 0, xt = 0;
  // The wave's 32 rows are 4 KB of contiguous memory: piece h of a lane is at byte h * 1024 + 16 * lane of it.
  const unsigned lane16 = lane * 16;
  auto issue = [&](unsigned base) {           // base: first position of this wave's 32 tokens
    const unsigned last = p1 - 1 - base;      // rows past the end re-read the last one (never stored)
    const unsigned mine = min((unsigned)ptok, last);
    x0 = *reinterpret_cast<const int*>(x0row + ((base + mine) << x0sh));
    xt = *reinterpret_cast<const int*>(xtrow + ((base + mine) << xtsh));
    if (last >= 31) {                         // (uniform over the wave) all 32 rows exist
      const char* src = lrow + (size_t)base * RB + lane16;
#pragma unroll
      for (int h = 0; h < 4; ++h) stage[h] = *reinterpret_cast<const v4f_nt*>(src + h * 1024);
    } else {
#pragma unroll
      for (int h = 0; h < 4; ++h) {
        const unsigned q = h * 64 + lane, r = min(q / 8, last), j = q % 8;
        stage[h] = *reinterpret_cast<const v4f_nt*>(lrow + ((base + r) * RB + j * 16));
      }
    }
  };
  unsigned base = wave * 32;
  const int t = load_index(a.t, b, a.t_bytes, a.tb.T);      // requested first: loads retire in issue order
  __shared__ QsTablesT<KC> s_next[NEXT ? PW : 1];
  // The next batch's share of this launch, on the LAST threads of the workgroup (token blockDim.x - 1 - threadIdx.x;
  // the host launches this variant for npos <= blockDim.x).  A wave that holds such tokens does all of it on its own,
  // BEFORE it asks for its rows: token load, timestep draw (scalar unit), the sampling rows of that timestep from the
  // digest into the wave's own LDS copy, draw, store.  Rows are served in wave order and the vector
  // units work through a SIMD's waves in that order too, so these waves' rows would have waited for the arithmetic
  // of the earlier waves anyway; the waves without tokens run exactly the code of the plain kernel, and there is no
  // workgroup barrier.
  if constexpr (NEXT) {
    const unsigned ntok = blockDim.x - 1 - threadIdx.x;
    if (blockDim.x - 64u * (unsigned)(wave + 1) < p1) {       // (uniform over the wave)
      __builtin_amdgcn_s_setprio(3);            // ahead of the other waves' arithmetic: this wave's rows wait on it
      const unsigned nsh = nx.x0_bytes == 8 ? 3 : 2;          // the low dword of a token, either width
      // requested first, used after the tables are in LDS (no branch around the load: the compiler would put the
      // clamp, and with it the wait, inside the branch)
      const int tok = *reinterpret_cast<const int*>((const char*)nx.x0 + (((size_t)b * a.npos + min(ntok, p1 - 1)) << nsh));
      uint64_t noff = nx.offset;
      if (nx.offset_dev) {
        // a scalar load (the counter is written by earlier launches only)
        uint64_t c;
        asm volatile("s_load_dwordx2 %0, %1, 0x0\n\ts_waitcnt lgkmcnt(0)" : "=s"(c) : "s"(nx.offset_dev));
        noff += c;
      }
      const QsKey key = qs_key(noff, nx.seed);
      const int tn = draw_timestep(b, a.tb.T, key);
      if (ntok == 0) nx.t_out[b] = tn;
      // the sampling rows of that timestep
      float rw = 0, rc = 0, rs = 0;
      if (lane < KC) {
        rw = dg(a.tb.qbar_digest, tn, 4, KC)[lane];
        rc = dg(a.tb.qbar_digest, tn, 5, KC)[lane];
        rs = dg(a.tb.qbar_digest, tn, 6, KC)[lane];
      }
      if (lane == 0) s_next[wave].cdf[0] = 0.0f;
      if (lane < KC) {
        s_next[wave].w[lane] = rw;
        s_next[wave].cdf[lane + 1] = rc;
        s_next[wave].stay[lane] = __builtin_bit_cast(uint32_t, rs);
      }
      wave_lds_fence();
      const uint64_t n = (uint64_t)b * (uint64_t)a.npos + ntok;
      if (ntok < p1) nx.x_t[n] = qs_draw_counted<KC>(s_next[wave], clamp_index(tok, KC), n, key);
      D3PM_TL(7);
      __builtin_amdgcn_s_setprio(0);
    }
  }
  if (ONE || base < p1) issue(base);
  D3PM_TL_AFTER(1, t);
  const bool need_post = (t > 0) && loss_type != D3PM_LOSS_CE;
  unsigned same_w = 0;          // bit q: classes 4q .. 4q+3 of both halves have one off-diagonal weight each
  float bdo = 0, sum_boe = 0;
  float vd = 0, vo = 0, v_ad = 0, v_ao = 0;
  if (need_post && lane < KC) {
    vd = dg(a.tb.qbar_digest, t - 1, 0, KC)[lane];
    vo = dg(a.tb.qbar_digest, t - 1, 1, KC)[lane];
    v_ad = dg(a.tb.qone_digest, t, 0, KC)[lane];
    v_ao = dg(a.tb.qone_digest, t, 1, KC)[lane];
  }
  // this lane's per-class weights off_j + eps stay in the wave's LDS table and are read four at a time (a broadcast
  // read per group): sixteen registers less than keeping them, which is what this kernel is short of
  const v4f_nt* const my_boe = reinterpret_cast<const v4f_nt*>(&s_tab[wave][2][HC * half]);
  float w_vb = w_vb0, w_ce = w_ce0;
  if (GRAD && grad_out) {                     // (uniform) an upstream gradient per batch element
    const float go = grad_out[b];
    w_vb *= go;
    w_ce *= go;
  }
  float acc_vb = 0, acc_ce = 0;
  bool first = true;
  for (;; base += step) {
    const bool work = ONE || base < p1;       // uniform over the wave
    // the lane's 16 classes as eight pairs: the per-class arithmetic below is written on pairs so that it maps onto
    // the packed float32 instructions (one issue slot for two classes)
    f2v e[HC / 2];
    float m = 0, inv = 0, l2s = 0, mneg = 0;
    if (work) {
#pragma unroll
      for (int h = 0; h < 4; ++h) *reinterpret_cast<v4f_nt*>(piece0 + h * 8 * TS) = stage[h];
      if (first && need_post && lane < KC) {
        // this wave's table copy goes to LDS together with the rows (the digest values were requested right behind
        // the rows and arrive with them): the fence below covers both, and nothing waits on LDS after phase A
        const float bo = vo + eps, bo2 = bo * bo;
        s_tab[wave][0][lane] = v_ad + eps;
        s_tab[wave][1][lane] = v_ao + eps;
        s_tab[wave][2][lane] = bo;
        s_w4[wave][lane >> 2] = bo2 * bo2;      // (four lanes, one slot: equal values wherever the slot is used)
      }
      if (first) D3PM_TL(2);
      wave_lds_fence();
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const v4f_nt q = *reinterpret_cast<const v4f_nt*>(myhalf + 16 * j);
        e[2 * j] = f2v{q.x, q.y};
        e[2 * j + 1] = f2v{q.z, q.w};
      }
      // ---- phase A: softmax statistics (no timestep needed)
      m = fmaxf(e[0].x, e[0].y);
#pragma unroll
      for (int j = 1; j < HC / 2; ++j) m = fmaxf(fmaxf(m, e[j].x), e[j].y);
      m = fmaxf(m, partner<1>(m));
      mneg = -m * LOG2E;
      const f2v mneg2 = {mneg, mneg}, l2e2 = {LOG2E, LOG2E};
      f2v s2 = {0.0f, 0.0f};
#pragma unroll
      for (int j = 0; j < HC / 2; ++j) {
        const f2v a2 = e[j] * l2e2 + mneg2;
        e[j] = f2v{fast_exp2(a2.x), fast_exp2(a2.y)};
        s2 += e[j];
      }
      float s_ = s2.x + s2.y;
      s_ += partner<1>(s_);
      inv = fast_rcp(s_);
      l2s = fast_log2(s_);
      if (first) D3PM_TL_AFTER(3, l2s);
    }
    if (first) {
      first = false;
      if (need_post) {
        bdo = lane_bcast(vd - vo, 0);
        sum_boe = (1.0f - bdo) + (float)KC * eps;           // rows of Qbar_{t-1} sum to one
        // bit q of same_w: classes 4q .. 4q+3 of BOTH halves have one off-diagonal weight each.  Lane j < 32 holds
        // class j's weight, so a group of four classes is a quad of lanes: two DPP compares and a ballot, no LDS
        const float bo = vo + eps;
        const bool quad_eq = (bo == partner<1>(bo)) && (bo == partner<2>(bo));
        unsigned mk = (unsigned)__builtin_amdgcn_ballot_w64(quad_eq);   // lanes 0 .. 31
        mk &= mk >> 1;
        mk &= mk >> 2;                                      // bit 4g: all four lanes of quad g agree
        unsigned sw = 0;
#pragma unroll
        for (int q = 0; q < HC / 4; ++q) sw |= ((mk >> (4 * q)) & (mk >> (4 * q + 16)) & 1u) << q;
        same_w = __builtin_amdgcn_readfirstlane(sw);
      }
      D3PM_TL(4);
    }
    if (!work) break;
    // ---- phase B
    x0 = clamp_index(x0, KC);
    xt = clamp_index(xt, KC);
    const float z_x0 = *reinterpret_cast<const float*>(tokrow + cls_off(x0));
    const float z_xt = *reinterpret_cast<const float*>(tokrow + cls_off(xt));
    const bool live = (base + ptok < p1) && half == 0;      // the token's sums are counted once
    const float ce_bits = fmaf(m - z_x0, LOG2E, l2s);       // -log2 softmax(z)[x0]
    acc_ce += live ? ce_bits : 0.0f;
    float g_x0 = 0, g_xt = 0;
    const float e_x0 = fast_exp2(fmaf(z_x0, LOG2E, mneg)), e_xt = fast_exp2(fmaf(z_xt, LOG2E, mneg));
    if (!need_post) {
      acc_vb += live ? ce_bits : 0.0f;
      if constexpr (GRAD) {
        const float w = (t == 0 ? w_vb : 0.0f) + w_ce, wi = w * inv;
        const f2v wi2 = {wi, wi};
#pragma unroll
        for (int j = 0; j < HC / 2; ++j) e[j] *= wi2;
        g_x0 = fmaf(e_x0, wi, -w);
        g_xt = (x0 == xt) ? g_x0 : e_xt * wi;
      }
    } else {
      const float a_d = s_tab[wave][0][xt], a_o = s_tab[wave][1][xt], da = a_d - a_o;
      const float boe_x0 = s_tab[wave][2][x0], boe_xt = s_tab[wave][2][xt];
      const float same = (x0 == xt) ? 1.0f : 0.0f;
      const float bdoi = bdo * inv;
      const float pi_x0 = e_x0 * inv, pi_xt = e_xt * inv;
      const float f2_x0 = fmaf(bdo, pi_x0, boe_x0), f2_xt = fmaf(bdo, pi_xt, boe_xt);
      const float rf_x0 = fast_rcp(f2_x0), rf_xt = fast_rcp(f2_xt);
      const float b_x0 = boe_x0 + bdo, b_xt = fmaf(same, bdo, boe_xt);
      const float a_x0 = fmaf(same, da, a_o);
      const float Zt = fmaf(a_o, sum_boe + bdo, da * b_xt);
      const float Zm = fmaf(a_o, sum_boe + bdo, da * f2_xt);
      const float izt = fast_rcp(Zt), izm = fast_rcp(Zm);
      const float zr = Zm * izt;
      f2v pr[HC / 2], E2 = {0.0f, 0.0f};
      float S1 = 0;
      const f2v bdoi2 = {bdoi, bdoi};
      const v4f_nt w4 = *reinterpret_cast<const v4f_nt*>(&s_w4[wave][(HC / 4) * half]);
#pragma unroll
      for (int q = 0; q < HC / 4; ++q) {
        // the four reciprocals 1 / f_j from one: with p = (f0 f2, f1 f3) and r = 1 / (p.x p.y), (1/f0, 1/f1) =
        // r (p.y, p.x) (f2, f3) and (1/f2, 1/f3) = r (p.y, p.x) (f0, f1): every step is one packed operation on the
        // pairs as they lie in registers.  A transcendental costs three to four ordinary operations on this part;
        // every f_j lies in [eps, 2], so the product stays in range.
        const v4f_nt w = my_boe[q];
        const f2v w01 = {w.x, w.y}, w23 = {w.z, w.w};
        const f2v f01 = e[2 * q] * bdoi2 + w01, f23 = e[2 * q + 1] * bdoi2 + w23;
        const f2v pp = f01 * f23;
        const float r = fast_rcp(pp.x * pp.y);
        const f2v rp = f2v{r, r} * f2v{pp.y, pp.x};
        pr[2 * q] = w01 * (rp * f23);
        pr[2 * q + 1] = w23 * (rp * f01);
        if constexpr (GRAD) {
          E2 += e[2 * q] * pr[2 * q];
          E2 += e[2 * q + 1] * pr[2 * q + 1];
        }
        if ((same_w >> q) & 1) {              // one logarithm for four classes of one weight (see k_hybrid_loss_tpl)
          S1 = fmaf(w.x, fast_log2(w4[q] * r), S1);                 // prod_j pr_j = w^4 / (f0 f1 f2 f3) >= 1e-24
        } else {
          S1 = fmaf(w.x, fast_log2(pr[2 * q].x), S1);
          S1 = fmaf(w.y, fast_log2(pr[2 * q].y), S1);
          S1 = fmaf(w.z, fast_log2(pr[2 * q + 1].x), S1);
          S1 = fmaf(w.w, fast_log2(pr[2 * q + 1].y), S1);
        }
      }
      float E = E2.x + E2.y;
      S1 += partner<1>(S1);
      if constexpr (GRAD) E += partner<1>(E);
      S1 = fmaf(sum_boe, fast_log2(zr), S1);
      float kls = a_o * S1;
      kls = fmaf(-a_o * boe_x0, fast_log2(boe_x0 * rf_x0 * zr), kls);
      kls = fmaf(a_x0 * b_x0, fast_log2(b_x0 * rf_x0 * zr), kls);
      kls = fmaf((1.0f - same) * da * boe_xt, fast_log2(boe_xt * rf_xt * zr), kls);
      acc_vb += live ? kls * izt : 0.0f;
      if constexpr (GRAD) {
        const float A1 = w_vb * a_o * izm, A2 = w_vb * a_o * izt;
        const float d_x0 = -A2 * bdo * rf_x0;
        const float d_xt = w_vb * da * (izm - b_xt * rf_xt * izt);
        const float p0 = pi_x0 * d_x0 * bdo, pt = pi_xt * d_xt * bdo;
        const float D = bdo * (A1 - A2 * inv * E);
        const float Cc = w_ce - (D + p0 + pt);
        const float nA2i = -A2 * bdoi, Ci = inv * fmaf(bdo, A1, Cc);
        const f2v nA2i2 = {nA2i, nA2i}, Ci2 = {Ci, Ci};
#pragma unroll
        for (int j = 0; j < HC / 2; ++j) e[j] *= nA2i2 * pr[j] + Ci2;
        g_x0 = fmaf(e_x0, fmaf(nA2i, boe_x0 * rf_x0, Ci), p0 - w_ce);
        g_xt = fmaf(e_xt, fmaf(nA2i, boe_xt * rf_xt, Ci), pt);
        if (x0 == xt) g_x0 = g_xt = g_x0 + pt;
      }
    }
    D3PM_TL_AFTER(5, e[HC / 2 - 1].y);
    if constexpr (GRAD) {
#pragma unroll
      for (int j = 0; j < 4; ++j)
        *reinterpret_cast<v4f_nt*>(myhalf + 16 * j) = v4f_nt{e[2 * j].x, e[2 * j].y, e[2 * j + 1].x, e[2 * j + 1].y};
      if (half == 0) {                        // after both lanes' generic values (LDS executes a wave's writes in order)
        *reinterpret_cast<float*>(tokrow + cls_off(x0)) = g_x0;
        *reinterpret_cast<float*>(tokrow + cls_off(xt)) = g_xt;
      }
      wave_lds_fence();
#ifdef D3PM_TIMELINE
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      D3PM_TL(6);
#endif
      const unsigned nrows = p1 - base;       // tokens of this wave that exist
      if (nrows >= 32) {
        char* dst = drow + (size_t)base * RB + lane16;
#pragma unroll
        for (int h = 0; h < 4; ++h) {
          const v4f_nt v = *reinterpret_cast<const v4f_nt*>(piece0 + h * 8 * TS);
          __builtin_nontemporal_store(v, reinterpret_cast<v4f_nt*>(dst + h * 1024));
        }
      } else {
#pragma unroll
        for (int h = 0; h < 4; ++h) {
          const unsigned q = h * 64 + lane, r = q / 8, j = q % 8;
          const v4f_nt v = *reinterpret_cast<const v4f_nt*>(piece0 + h * 8 * TS);
          if (r < nrows)
            __builtin_nontemporal_store(v, reinterpret_cast<v4f_nt*>(drow + ((base + r) * RB + j * 16)));
        }
      }
    }
    D3PM_TL(8);
    if (ONE || base + step >= p1) break;      // (uniform over the wave)
    wave_lds_fence();
    issue(base + step);
  }
  // Totals: rows of 16 lanes by DPP, the row totals of every wave to LDS, and the first wave adds the (at most) 64
  // pairs up, again by DPP (this is the tail of the wave's only pass: its latency is not covered by anything, and
  // every vector instruction here is one of 16 waves').
  acc_vb = group_sum<16>(acc_vb);
  acc_ce = group_sum<16>(acc_ce);
  if ((lane & 15) == 0) *reinterpret_cast<f2v*>(&red[(wave * 4 + (lane >> 4)) * 2]) = f2v{acc_vb, acc_ce};
  __syncthreads();
  if (wave == 0) {
    const int nrow = (blockDim.x >> 6) * 4;
    f2v v = {0.0f, 0.0f};
    if (lane < nrow) v = *reinterpret_cast<const f2v*>(&red[lane * 2]);
    const float svb = wave_sum(v.x), sce = wave_sum(v.y);
    if (lane == 0) write_losses(out, b, svb, sce, a.npos, loss_type, hybrid_coeff);
  }
  D3PM_TL_FLUSH;
}

// ------------------------------------------------------------------ fused hybrid loss, K = 64 / 128 / 256
// The arithmetic of k_hybrid_loss_pair32 with a token on G = K / 16 lanes, 16 classes a lane in four interleaved
// 16-byte pieces (piece h of lane g = classes 4 (g + G h) .. + 3: every access instruction of a lane group covers
// 16 G contiguous bytes of the row, no staging through LDS on the way in).  64 / G tokens per wave; the sums of a
// token cross its lanes inside a DPP row (log2 G steps).  Per-class constants come from a workgroup-wide LDS table by
// piece (one 16-byte broadcast read per piece).  The logits of the two special classes are read from, and their
// gradient values patched into, a wave-private LDS image of the wave's rows in natural class order.
// NW waves per workgroup: 4 (256 threads; a batch element split over chunks, the partial sums added by
// k_loss_finalize) or 16 (one workgroup per batch element, which writes its losses itself).
template <int G, bool GRAD, int NW>
__global__ __launch_bounds__(64 * NW) void k_hybrid_loss_wide(
    ImgArgs a, const float* __restrict__ logits, int loss_type, float hybrid_coeff, float w_vb0,
    const float* __restrict__ grad_out, LossOut out, float* __restrict__ dlogits, float w_ce0) {
  constexpr int K = 16 * G, HC = 16, RB = 4 * K, TPW = 64 / G;
  static_assert(G == 4 || G == 8 || G == 16, "64, 128 or 256 classes");
  __shared__ __align__(16) float s_tab[3][K];               // a_diag + eps, a_off + eps by x_t; off + eps of Qbar_{t-1}
  __shared__ __align__(16) float s_w4[K / 4];               // (off + eps)^4 of a piece whose classes share one weight, else -1
  __shared__ float s_bdo;
  __shared__ __align__(8) float red[NW * 4 * 2];
  __shared__ __align__(16) char wide_rows[NW * TPW * RB];   // [waves][tokens of a wave][K]: 4 KB per wave
  const float eps = a.tb.eps;
  const int b = blockIdx.x / a.chunks, chunk = blockIdx.x % a.chunks;
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int g = lane % G, slot = lane / G;
  const unsigned tpp = NW * TPW;                            // tokens per pass of the workgroup
  const unsigned p1 = (unsigned)min(a.npos, (chunk + 1) * a.chunk_pos);
  const char* __restrict__ lrow = (const char*)(logits + (size_t)b * a.npos * K);
  char* __restrict__ drow = GRAD ? (char*)(dlogits + (size_t)b * a.npos * K) : nullptr;
  const char* __restrict__ x0row = (const char*)a.x0 + (size_t)b * a.npos * a.x0_bytes;
  const char* __restrict__ xtrow = (const char*)a.xt + (size_t)b * a.npos * a.xt_bytes;
  const unsigned x0sh = a.x0_bytes == 8 ? 3 : 2, xtsh = a.xt_bytes == 8 ? 3 : 2;
  char* const tokrow = wide_rows + (size_t)(wave * TPW + slot) * RB;        // class c of this lane's token: + 4 c
  char* const mine = tokrow + 16 * g;                                       // this lane's piece h: + 16 G h

  if (blockIdx.x == 0 && threadIdx.x == 0 && out.rng_counter) *out.rng_counter += out.rng_step;

  const int t = load_index(a.t, b, a.t_bytes, a.tb.T);
  const bool need_post = (t > 0) && loss_type != D3PM_LOSS_CE;
  float vd = 0, vo = 0, v_ad = 0, v_ao = 0;
  if (need_post && threadIdx.x < K) {      // (K <= 256 <= blockDim.x)
    vd = dg(a.tb.qbar_digest, t - 1, 0, K)[threadIdx.x];
    vo = dg(a.tb.qbar_digest, t - 1, 1, K)[threadIdx.x];
    v_ad = dg(a.tb.qone_digest, t, 0, K)[threadIdx.x];
    v_ao = dg(a.tb.qone_digest, t, 1, K)[threadIdx.x];
  }
  // rows and tokens one pass ahead; positions past the end re-read the last one (never stored)
  v4f_nt zn[4];
  int x0n = 0, xtn = 0;
  auto issue = [&](unsigned pos) {
    const unsigned at = min(pos, p1 - 1);
    x0n = *reinterpret_cast<const int*>(x0row + ((size_t)at << x0sh));
    xtn = *reinterpret_cast<const int*>(xtrow + ((size_t)at << xtsh));
    const char* src = lrow + (size_t)at * RB + 16 * g;
#pragma unroll
    for (int h = 0; h < 4; ++h) zn[h] = *reinterpret_cast<const v4f_nt*>(src + 16 * G * h);
  };
  unsigned pos = (unsigned)(chunk * a.chunk_pos) + wave * TPW + slot;
  const unsigned wave_first = (unsigned)(chunk * a.chunk_pos) + wave * TPW;   // (scalar) first token of this wave's pass
  if (wave_first < p1) issue(pos);
  unsigned slow_mask = 0;       // bit h: piece h of some lane holds classes of different weights (scalar)
  float bdo = 0, sum_boe = 0;
  if (need_post) {
    if (threadIdx.x < K) {
      const float bo = vo + eps, bo2 = bo * bo;
      const float qe = ((bo == partner<1>(bo)) && (bo == partner<2>(bo))) ? 1.0f : 0.0f;
      const bool same4 = qe != 0.0f && partner<1>(qe) != 0.0f;               // all four classes of the piece
      s_tab[0][threadIdx.x] = v_ad + eps;
      s_tab[1][threadIdx.x] = v_ao + eps;
      s_tab[2][threadIdx.x] = bo;
      s_w4[threadIdx.x >> 2] = same4 ? bo2 * bo2 : -1.0f;                    // (four threads, one slot, one value)
      if (threadIdx.x == 0) s_bdo = vd - vo;
    }
    __syncthreads();
    bdo = s_bdo;
    sum_boe = (1.0f - bdo) + (float)K * eps;                // rows of Qbar_{t-1} sum to one
#pragma unroll
    for (int h = 0; h < 4; ++h)
      slow_mask |= (__builtin_amdgcn_ballot_w64(s_w4[g + G * h] < 0.0f) != 0 ? 1u : 0u) << h;
    slow_mask = __builtin_amdgcn_readfirstlane(slow_mask);
  }
  float w_vb = w_vb0, w_ce = w_ce0;
  if (GRAD && grad_out) {
    const float go = grad_out[b];
    w_vb *= go;
    w_ce *= go;
  }
  float acc_vb = 0, acc_ce = 0;
  for (unsigned first = wave_first; first < p1; first += tpp, pos += tpp) {
    const bool live = pos < p1;
    f2v e[HC / 2];
#pragma unroll
    for (int h = 0; h < 4; ++h) {
      e[2 * h] = f2v{zn[h].x, zn[h].y};
      e[2 * h + 1] = f2v{zn[h].z, zn[h].w};
      *reinterpret_cast<v4f_nt*>(mine + 16 * G * h) = zn[h];      // the row in natural order, for the two special logits
    }
    int x0 = clamp_index(x0n, K), xt = clamp_index(xtn, K);
    if (first + tpp < p1) issue(pos + tpp);
    wave_lds_fence();
    const float z_x0 = *reinterpret_cast<const float*>(tokrow + 4 * x0);
    const float z_xt = *reinterpret_cast<const float*>(tokrow + 4 * xt);
    // ---- softmax statistics
    float m = fmaxf(e[0].x, e[0].y);
#pragma unroll
    for (int j = 1; j < HC / 2; ++j) m = fmaxf(fmaxf(m, e[j].x), e[j].y);
    m = group_max<G>(m);
    const float mneg = -m * LOG2E;
    const f2v mneg2 = {mneg, mneg}, l2e2 = {LOG2E, LOG2E};
    f2v s2 = {0.0f, 0.0f};
#pragma unroll
    for (int j = 0; j < HC / 2; ++j) {
      const f2v a2 = e[j] * l2e2 + mneg2;
      e[j] = f2v{fast_exp2(a2.x), fast_exp2(a2.y)};
      s2 += e[j];
    }
    const float s_ = group_sum<G>(s2.x + s2.y);
    const float inv = fast_rcp(s_), l2s = fast_log2(s_);
    const bool count = live && g == 0;                      // the token's sums are counted once
    const float ce_bits = fmaf(m - z_x0, LOG2E, l2s);       // -log2 softmax(z)[x0]
    acc_ce += count ? ce_bits : 0.0f;
    float g_x0 = 0, g_xt = 0;
    const float e_x0 = fast_exp2(fmaf(z_x0, LOG2E, mneg)), e_xt = fast_exp2(fmaf(z_xt, LOG2E, mneg));
    if (!need_post) {
      acc_vb += count ? ce_bits : 0.0f;
      if constexpr (GRAD) {
        const float w = (t == 0 ? w_vb : 0.0f) + w_ce, wi = w * inv;
        const f2v wi2 = {wi, wi};
#pragma unroll
        for (int j = 0; j < HC / 2; ++j) e[j] *= wi2;
        g_x0 = fmaf(e_x0, wi, -w);
        g_xt = (x0 == xt) ? g_x0 : e_xt * wi;
      }
    } else {
      const float a_d = s_tab[0][xt], a_o = s_tab[1][xt], da = a_d - a_o;
      const float boe_x0 = s_tab[2][x0], boe_xt = s_tab[2][xt];
      const float same = (x0 == xt) ? 1.0f : 0.0f;
      const float bdoi = bdo * inv;
      const float pi_x0 = e_x0 * inv, pi_xt = e_xt * inv;
      const float f2_x0 = fmaf(bdo, pi_x0, boe_x0), f2_xt = fmaf(bdo, pi_xt, boe_xt);
      const float rf_x0 = fast_rcp(f2_x0), rf_xt = fast_rcp(f2_xt);
      const float b_x0 = boe_x0 + bdo, b_xt = fmaf(same, bdo, boe_xt);
      const float a_x0 = fmaf(same, da, a_o);
      const float Zt = fmaf(a_o, sum_boe + bdo, da * b_xt);
      const float Zm = fmaf(a_o, sum_boe + bdo, da * f2_xt);
      const float izt = fast_rcp(Zt), izm = fast_rcp(Zm);
      const float zr = Zm * izt;
      f2v pr[HC / 2], E2 = {0.0f, 0.0f};
      float S1 = 0;
      const f2v bdoi2 = {bdoi, bdoi};
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        // (the quad of k_hybrid_loss_pair32: four reciprocals from one, every step a packed operation)
        const v4f_nt w = *reinterpret_cast<const v4f_nt*>(&s_tab[2][4 * (g + G * q)]);
        const f2v w01 = {w.x, w.y}, w23 = {w.z, w.w};
        const f2v f01 = e[2 * q] * bdoi2 + w01, f23 = e[2 * q + 1] * bdoi2 + w23;
        const f2v pp = f01 * f23;
        const float r = fast_rcp(pp.x * pp.y);
        const f2v rp = f2v{r, r} * f2v{pp.y, pp.x};
        pr[2 * q] = w01 * (rp * f23);
        pr[2 * q + 1] = w23 * (rp * f01);
        if constexpr (GRAD) {
          E2 += e[2 * q] * pr[2 * q];
          E2 += e[2 * q + 1] * pr[2 * q + 1];
        }
        if (!((slow_mask >> q) & 1)) {        // (scalar) every lane's piece q has one weight: one logarithm for four classes
          S1 = fmaf(w.x, fast_log2(s_w4[g + G * q] * r), S1);
        } else {
          S1 = fmaf(w.x, fast_log2(pr[2 * q].x), S1);
          S1 = fmaf(w.y, fast_log2(pr[2 * q].y), S1);
          S1 = fmaf(w.z, fast_log2(pr[2 * q + 1].x), S1);
          S1 = fmaf(w.w, fast_log2(pr[2 * q + 1].y), S1);
        }
      }
      S1 = group_sum<G>(S1);
      float E = 0;
      if constexpr (GRAD) E = group_sum<G>(E2.x + E2.y);
      S1 = fmaf(sum_boe, fast_log2(zr), S1);
      float kls = a_o * S1;
      kls = fmaf(-a_o * boe_x0, fast_log2(boe_x0 * rf_x0 * zr), kls);
      kls = fmaf(a_x0 * b_x0, fast_log2(b_x0 * rf_x0 * zr), kls);
      kls = fmaf((1.0f - same) * da * boe_xt, fast_log2(boe_xt * rf_xt * zr), kls);
      acc_vb += count ? kls * izt : 0.0f;
      if constexpr (GRAD) {
        const float A1 = w_vb * a_o * izm, A2 = w_vb * a_o * izt;
        const float d_x0 = -A2 * bdo * rf_x0;
        const float d_xt = w_vb * da * (izm - b_xt * rf_xt * izt);
        const float p0 = pi_x0 * d_x0 * bdo, pt = pi_xt * d_xt * bdo;
        const float D = bdo * (A1 - A2 * inv * E);
        const float Cc = w_ce - (D + p0 + pt);
        const float nA2i = -A2 * bdoi, Ci = inv * fmaf(bdo, A1, Cc);
        const f2v nA2i2 = {nA2i, nA2i}, Ci2 = {Ci, Ci};
#pragma unroll
        for (int j = 0; j < HC / 2; ++j) e[j] *= nA2i2 * pr[j] + Ci2;
        g_x0 = fmaf(e_x0, fmaf(nA2i, boe_x0 * rf_x0, Ci), p0 - w_ce);
        g_xt = fmaf(e_xt, fmaf(nA2i, boe_xt * rf_xt, Ci), pt);
        if (x0 == xt) g_x0 = g_xt = g_x0 + pt;
      }
    }
    if constexpr (GRAD) {
      // the gradient row through the image; the two special classes are patched in after every lane's generic values
      // (a wave's LDS operations execute in order)
      wave_lds_fence();
#pragma unroll
      for (int h = 0; h < 4; ++h)
        *reinterpret_cast<v4f_nt*>(mine + 16 * G * h) = v4f_nt{e[2 * h].x, e[2 * h].y, e[2 * h + 1].x, e[2 * h + 1].y};
      if (g == 0) {
        *reinterpret_cast<float*>(tokrow + 4 * x0) = g_x0;
        *reinterpret_cast<float*>(tokrow + 4 * xt) = g_xt;
      }
      wave_lds_fence();
      if (live) {
        char* dst = drow + (size_t)pos * RB + 16 * g;
#pragma unroll
        for (int h = 0; h < 4; ++h)
          __builtin_nontemporal_store(*reinterpret_cast<const v4f_nt*>(mine + 16 * G * h),
                                      reinterpret_cast<v4f_nt*>(dst + 16 * G * h));
      }
      wave_lds_fence();                       // the image is written again at the top of the next pass
    }
  }
  acc_vb = group_sum<16>(acc_vb);
  acc_ce = group_sum<16>(acc_ce);
  if ((lane & 15) == 0) *reinterpret_cast<f2v*>(&red[(wave * 4 + (lane >> 4)) * 2]) = f2v{acc_vb, acc_ce};
  __syncthreads();
  if (wave == 0) {
    f2v v = {0.0f, 0.0f};
    if (lane < NW * 4) v = *reinterpret_cast<const f2v*>(&red[lane * 2]);
    const float svb = wave_sum(v.x), sce = wave_sum(v.y);
    if (lane == 0) {
      if (a.chunks == 1) {
        write_losses(out, b, svb, sce, a.npos, loss_type, hybrid_coeff);
      } else {
        out.part[((size_t)b * a.chunks + chunk) * 2 + 0] = svb;
        out.part[((size_t)b * a.chunks + chunk) * 2 + 1] = sce;
      }
    }
  }
}

// Sums of the per-workgroup partials in chunk order; `unit` converts them to
// bits (1 for partials already in bits, 1/ln2 for partials in nats).
__global__ void k_loss_finalize(LossOut out, int B, int chunks, long npos, int loss_type, float hybrid_coeff,
                                float unit) {
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= B) return;
  float svb = 0, sce = 0;
  for (int c = 0; c < chunks; ++c) {
    svb += out.part[((size_t)b * chunks + c) * 2 + 0];
    sce += out.part[((size_t)b * chunks + c) * 2 + 1];
  }
  write_losses(out, b, svb * unit, sce * unit, npos, loss_type, hybrid_coeff);
}

// ------------------------------------------------------------------ p_sample
// (with 16 classes a lane the kernel would take 149 registers, three waves per SIMD: held to 128, without spills)
template <class L, int NOISE>
__global__ __launch_bounds__(BLOCK, L::V >= 16 ? 4 : 1) void k_p_sample(ImgArgs a, const float* __restrict__ logits, long ld,
                                                    const float* __restrict__ noise, uint64_t seed, uint64_t offset,
                                                    const uint64_t* __restrict__ offset_dev,
                                                    int64_t* __restrict__ sample, float* __restrict__ probs) {
  if (NOISE == D3PM_NOISE_PHILOX && offset_dev) offset += *offset_dev;
  const int K = a.tb.K;
  const float eps = a.tb.eps;
  const int b = blockIdx.x / a.chunks, chunk = blockIdx.x % a.chunks;
  const int g = threadIdx.x % L::G, slot = threadIdx.x / L::G;
  const int t = load_index(a.t, b, a.t_bytes, a.tb.T);
  Post<L> P;
  if (t > 0) load_post<L>(a.tb, t, g, P);
  const long p1 = min(a.npos, (chunk + 1) * a.chunk_pos);
  for (long pos = chunk * a.chunk_pos + slot; pos < p1; pos += BLOCK / L::G) {
    const long n = (long)b * a.npos + pos;
    float z[L::V], pi[L::V], zmax, zsum;
    load_slots<L>(logits + n * ld, g, K, -INFINITY, z);
    const bool want_pi = (t > 0) || probs;
    if (NOISE == D3PM_NOISE_PHILOX && !probs) {
      // device noise and no probabilities asked for: nothing is compared bit for bit, hardware exp2 / rcp
      if (want_pi) {
        float m = z[0];
#pragma unroll
        for (int v = 1; v < L::V; ++v) m = fmaxf(m, z[v]);
        m = group_max<L::G>(m);
        float ssum = 0;
#pragma unroll
        for (int v = 0; v < L::V; ++v) { pi[v] = fast_exp2((z[v] - m) * LOG2E); ssum += pi[v]; }
        const float inv = fast_rcp(group_sum<L::G>(ssum));
#pragma unroll
        for (int v = 0; v < L::V; ++v) pi[v] *= inv;
      }
    } else if (want_pi) group_softmax<L>(z, pi, zmax, zsum);
    if (probs) store_slots<L>(probs + n * K, g, K, pi);
    float best = -INFINITY;
    int besti = 0x7fffffff;
    if (t == 0) {   // argmax of the logits themselves, no noise (mask of :394)
#pragma unroll
      for (int v = 0; v < L::V; ++v) {
        const int c = L::cls(g, v);
        if (c < K && (z[v] > best || besti == 0x7fffffff)) { best = z[v]; besti = c; }
      }
    } else if constexpr (NOISE == D3PM_NOISE_PHILOX) {
      // noise drawn on the device: nothing to match bit for bit, so one uniform per token and an inverse CDF over the
      // unnormalised posterior (a_j + eps)(f2_j + eps) -- no logarithm, no Gumbel transform
      const int xt = load_index(a.xt, n, a.xt_bytes, a.tb.K);
      const float a_d = dg(a.tb.qone_digest, t, 0, K)[xt] + eps, a_o = dg(a.tb.qone_digest, t, 1, K)[xt] + eps;
      float w[L::V];
#pragma unroll
      for (int v = 0; v < L::V; ++v) {
        const int c = L::cls(g, v);
        const float f2 = fmaf(P.bd[v] - P.bo[v], pi[v], P.bo[v]);
        w[v] = c < K ? ((c == xt) ? a_d : a_o) * (f2 + eps) : 0.0f;
      }
      const U4 r = philox4x32_10((uint64_t)n, offset, seed);
      const float uu = u01_open(r.x);
      besti = group_pick<L::G, L::V>(w, g, uu, xt, [](int gg, int vv) { return L::cls(gg, vv); });
      if (g == 0) sample[n] = besti;
      continue;
    } else {
      const int xt = load_index(a.xt, n, a.xt_bytes, a.tb.K);
      const float l1d = dg(a.tb.qone_digest, t, 2, K)[xt], l1o = dg(a.tb.qone_digest, t, 3, K)[xt];
      float u[L::V];
      load_slots<L>(noise + n * K, g, K, 0.5f, u);
#pragma unroll
      for (int v = 0; v < L::V; ++v) {
        const int c = L::cls(g, v);
        const float f2 = fmaf(P.bd[v] - P.bo[v], pi[v], P.bo[v]);
        const float gum = (NOISE == D3PM_NOISE_GUMBEL) ? u[v] : gumbel_from_uniform(u[v]);
        const float val = (((c == xt) ? l1d : l1o) + logf(f2 + eps)) + gum;
        if (c < K && (val > best || besti == 0x7fffffff)) { best = val; besti = c; }
      }
    }
    group_argmax<L::G>(best, besti);
    if (g == 0) sample[n] = besti;
  }
}

// ------------------------------------------------------------------ p_sample, device noise, one token per lane
// The reverse step of the sampling loop (no probabilities out, noise drawn on the device) for K = 32 with the tokens
// of the whole batch taken as one row, as k_lvb_reverse_flat does for the protein-style step: wave w owns tokens
// 64 w .. 64 w + 63, rows through the wave-private LDS image with division-free addressing, and the draw is an in-lane
// inverse CDF over the unnormalised posterior (a_j + eps)(f2_j + eps).  The factor a_off + eps is divided out: generic
// weights g_j = (off_j + eps) + (diag - off) pi_j, whose running sum is MO_j + c P_j with MO the digest's own prefix
// row (wave-uniform) and P_j the running sum of the softmax pass, plus an extra mass X on class x_t; one uniform
// decides both.  t == 0: argmax of the logits (first maximum), as with explicit noise.  A wave that straddles two
// batch elements with different timesteps runs the per-timestep part once for each.
template <int KC>
__global__ __launch_bounds__(256) void k_p_sample_flat(ImgArgs a, const float* __restrict__ logits, uint64_t seed,
                                                       uint64_t offset, const uint64_t* __restrict__ offset_dev,
                                                       int64_t* __restrict__ sample, unsigned ntok, float inv_npos) {
  using Img = RowImage<KC>;
  extern __shared__ __align__(16) char flat_rows[];         // [waves][64][STRIDE]
  __shared__ float s_tab[4][3][KC];                         // per wave: a_diag + eps, a_off + eps of Q_t by x_t; off + eps
  const float eps = a.tb.eps;
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const unsigned base = (blockIdx.x * (blockDim.x >> 6) + wave) * 64u;
  if (base >= ntok) return;                                 // (uniform over the wave; the kernel has no barrier)
  const unsigned np = (unsigned)a.npos;
  unsigned b0 = __builtin_amdgcn_readfirstlane((unsigned)((float)base * inv_npos));   // off by one at most (B <= 2^20)
  int r0 = (int)(base - b0 * np);
  if (r0 < 0) { b0 -= 1; r0 += (int)np; }
  else if (r0 >= (int)np) { b0 += 1; r0 -= (int)np; }
  const bool second = (unsigned)r0 + lane >= np;            // this lane's token belongs to batch element b0 + 1
  const unsigned nrows = min(64u, ntok - base);
  const bool spans = (unsigned)r0 + nrows > np;             // (uniform)
  char* const img = flat_rows + (size_t)wave * Img::BYTES;
  Img rows;
  rows.request_flat((const char*)logits + (size_t)base * Img::RB, nrows * Img::RB - Img::PB, lane, nrows == 64);
  const unsigned xsh = a.xt_bytes == 8 ? 3 : 2;             // the low dword of a token, either width
  int xt = *reinterpret_cast<const int*>((const char*)a.xt + ((size_t)(base + min((unsigned)lane, nrows - 1)) << xsh));
  const int t0 = load_index(a.t, b0, a.t_bytes, a.tb.T);
  const int t1 = spans ? load_index(a.t, b0 + 1, a.t_bytes, a.tb.T) : t0;
  const bool two = t1 != t0;                                // (uniform)
  float vd = 0, vw = 0, vc = 0, ad = 0, ao = 0;
  auto fetch = [&](int t) {
    if (t > 0 && lane < KC) {
      vd = dg(a.tb.qbar_digest, t - 1, 0, KC)[lane] - dg(a.tb.qbar_digest, t - 1, 1, KC)[lane];
      vw = dg(a.tb.qbar_digest, t - 1, 4, KC)[lane];        // off + eps
      vc = dg(a.tb.qbar_digest, t - 1, 5, KC)[lane];        // its inclusive prefix sums
      ad = dg(a.tb.qone_digest, t, 0, KC)[lane] + eps;
      ao = dg(a.tb.qone_digest, t, 1, KC)[lane] + eps;
    }
  };
  fetch(t0);
  if (offset_dev) offset += *offset_dev;
  const QsKey key = qs_key(offset, seed);
  const U2 rnd = philox2x32_10((base + lane) ^ key.x, key.y, key.z);
  const float u = u01_open(rnd.x);
  xt = clamp_index(xt, KC);
  rows.deposit_flat(img, lane, Img::split(lane));
  wave_lds_order();
  float z[KC];
  Img::read_row(img, lane, z);
  const float z_xt = *reinterpret_cast<const float*>(img + lane * Img::STRIDE + 4 * xt);
  float m = z[0];
  int amax = 0;
#pragma unroll
  for (int j = 1; j < KC; ++j) {
    amax = z[j] > m ? j : amax;                             // first maximum, as torch.argmax
    m = fmaxf(m, z[j]);
  }
  const float mneg = -m * LOG2E;
  float P[KC];                                              // running sums of e_j
  float acc = 0;
#pragma unroll
  for (int j = 0; j < KC; ++j) {
    acc += fast_exp2(fmaf(z[j], LOG2E, mneg));
    P[j] = acc;
  }
  const float e_xt = fast_exp2(fmaf(z_xt, LOG2E, mneg));
  const float inv_s = fast_rcp(acc);
  auto draw = [&](int t) {
    if (t == 0) return amax;                                // (uniform)
    if (lane < KC) {
      s_tab[wave][0][lane] = ad;
      s_tab[wave][1][lane] = ao;
      s_tab[wave][2][lane] = vw;
    }
    const float c = lane_broadcast(vd, 0) * inv_s;          // diag - off is one value for all classes
    float MO[KC];
#pragma unroll
    for (int j = 0; j < KC; ++j) MO[j] = lane_broadcast(vc, j);
    wave_lds_order();
    const float G = fmaf(P[KC - 1], c, MO[KC - 1]);
    const float a_d = s_tab[wave][0][xt], a_o = s_tab[wave][1][xt];
    const float g_xt = fmaf(e_xt, c, s_tab[wave][2][xt]);
    const float X = (a_d - a_o) * fast_rcp(a_o) * g_xt;
    const float target = u * (G + X) - X;                   // < 0: the extra mass of x_t was hit
    int cnt = 0;                                            // classes whose cumulative weight is <= target
#pragma unroll
    for (int j = 0; j < KC - 1; ++j) cnt += (fmaf(P[j], c, MO[j]) <= target) ? 1 : 0;
    return target < 0.0f ? xt : cnt;
  };
  int pick = draw(t0);
  if (two) {                                                // (uniform) the wave's second batch element, own timestep
    wave_lds_order();
    fetch(t1);
    const int p1 = draw(t1);
    pick = second ? p1 : pick;
  }
  if (lane < (int)nrows) sample[base + lane] = pick;
}

// ------------------------------------------------------------------ prior_bpd
template <class L>
__global__ __launch_bounds__(BLOCK) void k_prior(ImgArgs a, int absorbing, float* __restrict__ part) {
  __shared__ float red[BLOCK / 64];
  const int K = a.tb.K;
  const float eps = a.tb.eps;
  const int b = blockIdx.x / a.chunks, chunk = blockIdx.x % a.chunks;
  const int g = threadIdx.x % L::G, slot = threadIdx.x / L::G;
  const int t = a.tb.T - 1;
  float d[L::V], o[L::V], ld_[L::V], lo_[L::V];
  load_digest<L>(a.tb.qbar_digest, t, 0, K, g, d);
  load_digest<L>(a.tb.qbar_digest, t, 1, K, g, o);
  load_digest<L>(a.tb.qbar_digest, t, 2, K, g, ld_);
  load_digest<L>(a.tb.qbar_digest, t, 3, K, g, lo_);
  // same rounding as the digest's log(. + eps): float64 log rounded once
  const float lp_uniform = (float)log((double)(1.0f / (float)K + eps));
  const float lp_one = (float)log((double)(1.0f + eps)), lp_zero = (float)log((double)eps);
  float acc = 0;
  const long p1 = min(a.npos, (chunk + 1) * a.chunk_pos);
  for (long pos = chunk * a.chunk_pos + slot; pos < p1; pos += BLOCK / L::G) {
    const long n = (long)b * a.npos + pos;
    const int x0 = load_index(a.x0, n, a.x0_bytes, a.tb.K);
    float kl = 0;
#pragma unroll
    for (int v = 0; v < L::V; ++v) {
      const int c = L::cls(g, v);
      const float p = (c == x0) ? d[v] : o[v];
      const float lp = (c == x0) ? ld_[v] : lo_[v];
      const float lprior = absorbing ? (c == K / 2 ? lp_one : lp_zero) : lp_uniform;
      kl += (c < K) ? p * (lp - lprior) : 0.0f;
    }
    kl = group_sum<L::G>(kl);
    if (g == 0) acc += kl;
  }
  const float s = block_sum(acc, red);
  if (threadIdx.x == 0) {
    part[((size_t)b * a.chunks + chunk) * 2 + 0] = s;
    part[((size_t)b * a.chunks + chunk) * 2 + 1] = 0;
  }
}

// ------------------------------------------------------------------ rescale rows
__global__ __launch_bounds__(256) void k_scale_rows(float* __restrict__ gptr, const float* __restrict__ upstream,
                                                    float baked, long row_elems, int blocks_per_row) {
  const int b = blockIdx.x / blocks_per_row, part = blockIdx.x % blocks_per_row;
  const float up = upstream[b];
  if (up == baked) return;
  const float f = up / baked;
  float* row = gptr + (size_t)b * row_elems;
  const long per = (row_elems + blocks_per_row - 1) / blocks_per_row;
  const long e0 = part * per, e1 = min(row_elems, e0 + per);
  if ((((uintptr_t)row) & 15u) == 0 && (e0 & 3) == 0) {
    const long v0 = e0 / 4, v1 = e1 / 4;
    float4* r4 = reinterpret_cast<float4*>(row);
    for (long i = v0 + threadIdx.x; i < v1; i += blockDim.x) {
      float4 x = r4[i];
      x.x *= f; x.y *= f; x.z *= f; x.w *= f;
      r4[i] = x;
    }
    for (long i = v1 * 4 + threadIdx.x; i < e1; i += blockDim.x) row[i] *= f;
  } else {
    for (long i = e0 + threadIdx.x; i < e1; i += blockDim.x) row[i] *= f;
  }
}

// ------------------------------------------------------------------ host side
static int check_common(const d3pm_tables_f32* tb, int B, long npos) {
  D3PM_CHECK_ARG(tb, "tables descriptor is null");
  D3PM_CHECK_ARG(tb->K >= 2 && tb->K <= 256, "2 <= K <= 256 required, got %d", tb->K);
  D3PM_CHECK_ARG(tb->T >= 1, "T >= 1 required");
  D3PM_CHECK_ARG(B >= 0 && npos >= 0, "negative batch or position count");
  D3PM_CHECK_ARG((long)B * 64 < 0x7fffffffL, "batch too large");
  if (tb->structured) {
    D3PM_CHECK_ARG(tb->qbar_digest && tb->qone_digest, "structured tables need both digests");
  } else {
    D3PM_CHECK_ARG(tb->qbar && tb->qoneT, "dense tables need qbar and qoneT");
  }
  return D3PM_OK;
}
static bool bytes_ok(int b) { return b == 4 || b == 8; }

static ImgArgs make_args(const d3pm_tables_f32* tb, const void* x0, int x0b, const void* xt, int xtb, const void* t,
                         int tb_, int B, long npos, int G) {
  ImgArgs a;
  a.tb = *tb;
  a.x0 = x0; a.x0_bytes = x0b; a.xt = xt; a.xt_bytes = xtb; a.t = t; a.t_bytes = tb_;
  a.B = B; a.npos = npos;
  const TokenGrid tg = token_grid(B, npos, BLOCK / G);
  a.chunks = tg.chunks; a.chunk_pos = tg.chunk_pos;
  return a;
}

}  // namespace d3pm

#include "image_dense.inl"

using namespace d3pm;

extern "C" int d3pm_q_probs(const d3pm_tables_f32* tb, const void* x0, int tok_bytes, const void* t, int t_bytes, int B,
                            long npos, float* out, void* stream) {
  if (int rc = check_common(tb, B, npos)) return rc;
  D3PM_CHECK_ARG(bytes_ok(tok_bytes) && bytes_ok(t_bytes), "token/timestep width must be 4 or 8 bytes");
  if (B == 0 || npos == 0) return D3PM_OK;
  D3PM_CHECK_ARG(x0 && t && out, "null pointer");
  hipStream_t s = (hipStream_t)stream;
  if (!tb->structured) return dense_q_probs(tb, x0, tok_bytes, t, t_bytes, B, npos, out, s);
  return dispatch_layout(tb->K, aligned16(out), [&](auto lay) {
    using L = decltype(lay);
    ImgArgs a = make_args(tb, x0, tok_bytes, nullptr, 8, t, t_bytes, B, npos, L::G);
    hipLaunchKernelGGL(k_q_probs<L>, dim3(B * a.chunks), dim3(BLOCK), 0, s, a, out);
    D3PM_LAUNCH_CHECK("k_q_probs");
    return D3PM_OK;
  });
}

extern "C" int d3pm_q_sample(const d3pm_tables_f32* tb, const void* x0, int tok_bytes, const void* t, int t_bytes, int B,
                             long npos, const float* noise, int noise_kind, uint64_t seed, uint64_t offset,
                             const uint64_t* offset_dev, int64_t* x_t, void* stream) {
  if (int rc = check_common(tb, B, npos)) return rc;
  D3PM_CHECK_ARG(bytes_ok(tok_bytes) && bytes_ok(t_bytes), "token/timestep width must be 4 or 8 bytes");
  D3PM_CHECK_ARG(noise_kind == D3PM_NOISE_UNIFORM || noise_kind == D3PM_NOISE_GUMBEL || noise_kind == D3PM_NOISE_PHILOX,
                 "noise_kind %d not valid for q_sample", noise_kind);
  if (B == 0 || npos == 0) return D3PM_OK;
  D3PM_CHECK_ARG(x0 && t && x_t, "null pointer");
  D3PM_CHECK_ARG(noise_kind == D3PM_NOISE_PHILOX || noise, "noise tensor is null");
  hipStream_t s = (hipStream_t)stream;
  if (!tb->structured)
    return dense_q_sample(tb, x0, tok_bytes, t, t_bytes, B, npos, noise, noise_kind, seed, offset, offset_dev, x_t, s);
  if (noise_kind == D3PM_NOISE_PHILOX) {
    ImgArgs a = make_args(tb, x0, tok_bytes, nullptr, 8, t, t_bytes, B, npos, 1);    // one thread per token
    hipLaunchKernelGGL(k_q_sample_icdf, dim3(B * a.chunks), dim3(BLOCK), 0, s, a, seed, offset, offset_dev,
                       (int32_t*)nullptr, (float*)nullptr, x_t);
    D3PM_LAUNCH_CHECK("k_q_sample_icdf");
    return D3PM_OK;
  }
  return dispatch_layout(tb->K, noise_kind == D3PM_NOISE_PHILOX || aligned16(noise), [&](auto lay) {
    using L = decltype(lay);
    ImgArgs a = make_args(tb, x0, tok_bytes, nullptr, 8, t, t_bytes, B, npos, L::G);
    const dim3 grid(B * a.chunks), block(BLOCK);
    if (noise_kind == D3PM_NOISE_UNIFORM)
      hipLaunchKernelGGL((k_q_sample<L, D3PM_NOISE_UNIFORM>), grid, block, 0, s, a, noise, seed, offset, offset_dev,
                         x_t);
    else if (noise_kind == D3PM_NOISE_GUMBEL)
      hipLaunchKernelGGL((k_q_sample<L, D3PM_NOISE_GUMBEL>), grid, block, 0, s, a, noise, seed, offset, offset_dev,
                         x_t);
    else
      hipLaunchKernelGGL((k_q_sample<L, D3PM_NOISE_PHILOX>), grid, block, 0, s, a, noise, seed, offset, offset_dev,
                         x_t);
    D3PM_LAUNCH_CHECK("k_q_sample");
    return D3PM_OK;
  });
}

__global__ void k_draw_t(int B, int T, uint64_t seed, uint64_t offset, const uint64_t* __restrict__ offset_dev,
                         int32_t* __restrict__ t_draw, float* __restrict__ clear) {
  if (clear && blockIdx.x == 0 && threadIdx.x == 0) *clear = 0.0f;
  if (offset_dev) offset += *offset_dev;
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b < B) t_draw[b] = draw_timestep(b, T, offset, seed);
}

extern "C" int d3pm_q_sample_draw_t(const d3pm_tables_f32* tb, const void* x0, int tok_bytes, int B, long npos,
                                    uint64_t seed, uint64_t offset, const uint64_t* offset_dev, float* clear,
                                    int32_t* t_out, int64_t* x_t, void* stream) {
  if (int rc = check_common(tb, B, npos)) return rc;
  D3PM_CHECK_ARG(bytes_ok(tok_bytes), "token width must be 4 or 8 bytes");
  if (B == 0) return D3PM_OK;
  D3PM_CHECK_ARG(t_out, "null pointer");
  hipStream_t s = (hipStream_t)stream;
  if (npos == 0 || !tb->structured) {
    hipLaunchKernelGGL(k_draw_t, dim3((B + 255) / 256), dim3(256), 0, s, B, tb->T, seed, offset, offset_dev, t_out,
                       clear);
    D3PM_LAUNCH_CHECK("k_draw_t");
    if (npos == 0) return D3PM_OK;
    D3PM_CHECK_ARG(x0 && x_t, "null pointer");
    return dense_q_sample(tb, x0, tok_bytes, t_out, 4, B, npos, nullptr, D3PM_NOISE_PHILOX, seed, offset, offset_dev, x_t,
                          s);
  }
  D3PM_CHECK_ARG(x0 && x_t, "null pointer");
  ImgArgs a = make_args(tb, x0, tok_bytes, nullptr, 8, nullptr, 4, B, npos, 1);    // one thread per token
  hipLaunchKernelGGL(k_q_sample_icdf, dim3(B * a.chunks), dim3(BLOCK), 0, s, a, seed, offset, offset_dev, t_out, clear,
                     x_t);
  D3PM_LAUNCH_CHECK("k_q_sample_icdf");
  return D3PM_OK;
}

extern "C" int d3pm_q_posterior_logits(const d3pm_tables_f32* tb, const void* x_start, int x_start_is_logits, long ld,
                                       int tok_bytes, const void* x_t, int xt_bytes, const void* t, int t_bytes, int B,
                                       long npos, float* out, void* stream) {
  if (int rc = check_common(tb, B, npos)) return rc;
  D3PM_CHECK_ARG(bytes_ok(xt_bytes) && bytes_ok(t_bytes), "token/timestep width must be 4 or 8 bytes");
  D3PM_CHECK_ARG(x_start_is_logits ? ld >= tb->K : bytes_ok(tok_bytes), "bad x_start row stride or token width");
  if (B == 0 || npos == 0) return D3PM_OK;
  D3PM_CHECK_ARG(x_start && x_t && t && out, "null pointer");
  hipStream_t s = (hipStream_t)stream;
  if (!tb->structured)
    return dense_posterior_logits(tb, x_start, x_start_is_logits, ld, tok_bytes, x_t, xt_bytes, t, t_bytes, B, npos,
                                  out, s);
  const bool vec = aligned16(out) && (!x_start_is_logits || (aligned16(x_start) && ld == tb->K));
  return dispatch_layout(tb->K, vec, [&](auto lay) {
    using L = decltype(lay);
    ImgArgs a = make_args(tb, x_start_is_logits ? nullptr : x_start, tok_bytes, x_t, xt_bytes, t, t_bytes, B, npos, L::G);
    const dim3 grid(B * a.chunks), block(BLOCK);
    if (x_start_is_logits)
      hipLaunchKernelGGL((k_posterior_logits<L, true>), grid, block, 0, s, a, (const float*)x_start, ld, out);
    else
      hipLaunchKernelGGL((k_posterior_logits<L, false>), grid, block, 0, s, a, (const float*)nullptr, ld, out);
    D3PM_LAUNCH_CHECK("k_posterior_logits");
    return D3PM_OK;
  });
}

extern "C" int d3pm_q_posterior_logits_backward(const d3pm_tables_f32* tb, const float* x_start_logits, long ld,
                                                const void* t, int t_bytes, int B, long npos, const float* grad_out,
                                                float* grad_logits, void* stream) {
  if (int rc = check_common(tb, B, npos)) return rc;
  D3PM_CHECK_ARG(bytes_ok(t_bytes), "timestep width must be 4 or 8 bytes");
  D3PM_CHECK_ARG(ld >= tb->K, "row stride %ld smaller than K=%d", ld, tb->K);
  if (!tb->structured) {
    set_error("gradient of q_posterior_logits is implemented for column-structured (uniform / absorbing) tables");
    return D3PM_EUNSUPPORTED;
  }
  if (B == 0 || npos == 0) return D3PM_OK;
  D3PM_CHECK_ARG(x_start_logits && t && grad_out && grad_logits, "null pointer");
  const bool vec = aligned16(x_start_logits) && aligned16(grad_out) && aligned16(grad_logits) && ld == tb->K;
  return dispatch_layout(tb->K, vec, [&](auto lay) {
    using L = decltype(lay);
    ImgArgs a = make_args(tb, nullptr, 8, nullptr, 8, t, t_bytes, B, npos, L::G);
    hipLaunchKernelGGL(k_posterior_logits_backward<L>, dim3(B * a.chunks), dim3(BLOCK), 0, (hipStream_t)stream, a,
                       x_start_logits, ld, grad_out, grad_logits);
    D3PM_LAUNCH_CHECK("k_posterior_logits_backward");
    return D3PM_OK;
  });
}

extern "C" size_t d3pm_hybrid_loss_workspace(int B, long npos, int K) {
  (void)K;
  // one (vb, ce) pair per workgroup; the grid never has more than
  // max(2048, B) * 2 workgroups (see token_grid), bounded here generously.
  long passes = npos > 0 ? npos : 1;
  long chunks = passes < 4096 ? passes : 4096;
  return (size_t)(B > 0 ? B : 1) * (size_t)chunks * 2 * sizeof(float);
}

static int structured_hybrid_loss(const d3pm_tables_f32* tb, const float* logits, long ld, const void* x0, int tok_bytes,
                                  const void* x_t, int xt_bytes, const void* t, int t_bytes, int B, long npos,
                                  int loss_type, float hybrid_coeff, float grad_scale, const float* grad_out, LossOut o,
                                  float* dlogits, void* workspace, size_t workspace_bytes, hipStream_t s,
                                  const struct NextQ* next_q, bool* next_done);

static int hybrid_loss_impl(const d3pm_tables_f32* tb, const float* logits, long ld, const void* x0, int tok_bytes,
                            const void* x_t, int xt_bytes, const void* t, int t_bytes, int B, long npos,
                            int loss_type, float hybrid_coeff, float grad_scale, const float* grad_out, float* vb,
                            float* ce, float* loss, float* dlogits, float* loss_mean, uint64_t* rng_counter,
                            uint64_t rng_step, void* workspace, size_t workspace_bytes, void* stream,
                            const NextQ* next_q, bool* next_done) {
  if (int rc = check_common(tb, B, npos)) return rc;
  D3PM_CHECK_ARG(bytes_ok(tok_bytes) && bytes_ok(xt_bytes) && bytes_ok(t_bytes),
                 "token/timestep width must be 4 or 8 bytes");
  D3PM_CHECK_ARG(ld >= tb->K, "row stride %ld smaller than K=%d", ld, tb->K);
  if (loss_type != D3PM_LOSS_KL && loss_type != D3PM_LOSS_CE && loss_type != D3PM_LOSS_HYBRID) {
    set_error("loss_type %d not implemented", loss_type);
    return D3PM_EUNSUPPORTED;
  }
  if (B == 0) return D3PM_OK;
  D3PM_CHECK_ARG(npos > 0, "empty sample: mean over zero positions");
  D3PM_CHECK_ARG(npos * ld < 0x3fffffffL, "one batch element spans %ld logits: more than 2^30", npos * ld);
  D3PM_CHECK_ARG(logits && x0 && x_t && t, "null pointer");
  hipStream_t s = (hipStream_t)stream;
  LossOut o{vb, ce, loss, nullptr, (unsigned long long*)rng_counter, (unsigned long long)rng_step, loss_mean,
            1.0f / (float)B};
  int rc;
  if (!tb->structured) {
    rc = dense_hybrid_loss(tb, logits, ld, x0, tok_bytes, x_t, xt_bytes, t, t_bytes, B, npos, loss_type, hybrid_coeff,
                           grad_scale, grad_out, o, dlogits, workspace, workspace_bytes, s);
  } else {
    rc = structured_hybrid_loss(tb, logits, ld, x0, tok_bytes, x_t, xt_bytes, t, t_bytes, B, npos, loss_type,
                                hybrid_coeff, grad_scale, grad_out, o, dlogits, workspace, workspace_bytes, s, next_q,
                                next_done);
  }
  return rc;
}

extern "C" int d3pm_hybrid_loss(const d3pm_tables_f32* tb, const float* logits, long ld, const void* x0, int tok_bytes,
                                const void* x_t, int xt_bytes, const void* t, int t_bytes, int B, long npos,
                                int loss_type, float hybrid_coeff, float grad_scale, const float* grad_out, float* vb,
                                float* ce, float* loss, float* dlogits, float* loss_mean, uint64_t* rng_counter,
                                uint64_t rng_step, void* workspace, size_t workspace_bytes, void* stream) {
  bool unused = false;
  return hybrid_loss_impl(tb, logits, ld, x0, tok_bytes, x_t, xt_bytes, t, t_bytes, B, npos, loss_type, hybrid_coeff,
                          grad_scale, grad_out, vb, ce, loss, dlogits, loss_mean, rng_counter, rng_step, workspace,
                          workspace_bytes, stream, nullptr, &unused);
}

extern "C" int d3pm_hybrid_loss_then_q_sample(const d3pm_tables_f32* tb, const float* logits, long ld, const void* x0,
                                              int tok_bytes, const void* x_t, int xt_bytes, const void* t, int t_bytes,
                                              int B, long npos, int loss_type, float hybrid_coeff, float grad_scale,
                                              const float* grad_out, float* vb, float* ce, float* loss, float* dlogits,
                                              void* workspace, size_t workspace_bytes, const void* next_x0,
                                              int next_tok_bytes, uint64_t seed, uint64_t offset,
                                              const uint64_t* offset_dev, int32_t* next_t_out, int64_t* next_x_t,
                                              void* stream) {
  D3PM_CHECK_ARG(bytes_ok(next_tok_bytes), "token width must be 4 or 8 bytes");
  D3PM_CHECK_ARG(B == 0 || (next_x0 && next_t_out && (next_x_t || npos == 0)), "null pointer (next batch)");
  D3PM_CHECK_ARG(B == 0 || ((const void*)next_x_t != x_t && (const void*)next_t_out != t),
                 "the next batch's x_t / t must not be this step's x_t / t: the loss reads those while these are written");
  NextQ nq{next_x0, next_tok_bytes, seed, offset, offset_dev, next_t_out, next_x_t};
  bool fused = false;
  const int rc = hybrid_loss_impl(tb, logits, ld, x0, tok_bytes, x_t, xt_bytes, t, t_bytes, B, npos, loss_type,
                                  hybrid_coeff, grad_scale, grad_out, vb, ce, loss, dlogits, nullptr, nullptr, 0,
                                  workspace, workspace_bytes, stream, &nq, &fused);
  if (rc || fused || B == 0) return rc;
  // shapes the fused kernel does not take: the same two results from two launches
  return d3pm_q_sample_draw_t(tb, next_x0, next_tok_bytes, B, npos, seed, offset, offset_dev, nullptr, next_t_out,
                              next_x_t, stream);
}

// ------------------------------------------------------------------ the entry point that allocates nothing
// Partial sums of d3pm_hybrid_loss_static: one (vb, ce) pair per workgroup, written in full by the loss kernel
// and read by k_loss_finalize in the same call, so nothing in it has to start at zero.  Every GPU the library runs
// on holds its own copy of the array for the life of the process.
constexpr size_t STATIC_SCRATCH_BYTES = 128 * 1024;
static __device__ __align__(256) float g_loss_scratch[STATIC_SCRATCH_BYTES / sizeof(float)];

// Device address of g_loss_scratch on the current device (the array's name is a host-side stand-in, not a device
// address: kernels get this pointer).  Looked up once per device, with that device current.
static int static_scratch(float** out, hipStream_t stream, bool quiet = false) {
  constexpr int MAX_DEVICES = 64;
  static std::mutex mu;
  static float* ptrs[MAX_DEVICES] = {};
  int dev = -1;
  hipError_t e = hipGetDevice(&dev);
  if (e != hipSuccess) return hip_fail(e, "hipGetDevice");
  D3PM_CHECK_ARG(dev >= 0 && dev < MAX_DEVICES, "device ordinal %d out of range", dev);
  std::lock_guard<std::mutex> lock(mu);
  if (!ptrs[dev]) {
    if (!quiet) {
      // the look-up is not a stream operation: under a capture it would fail, or end the capture, with an error
      // that names neither
      hipStreamCaptureStatus st = hipStreamCaptureStatusNone;
      e = hipStreamIsCapturing(stream, &st);
      if (e != hipSuccess) return hip_fail(e, "hipStreamIsCapturing");
      D3PM_CHECK_ARG(st == hipStreamCaptureStatusNone,
                     "d3pm_hybrid_loss_static: the first call on a device resolves the scratch array's address and "
                     "cannot run under stream capture; call it (or d3pm_digest_f32) once on this device beforehand");
    }
    float* p = nullptr;
    e = hipGetSymbolAddress((void**)&p, HIP_SYMBOL(g_loss_scratch));
    if (e != hipSuccess) return hip_fail(e, "hipGetSymbolAddress(g_loss_scratch)");
    D3PM_CHECK_ARG(p, "hipGetSymbolAddress returned a null address");
    ptrs[dev] = p;
  }
  *out = ptrs[dev];
  return D3PM_OK;
}

namespace d3pm {
void warm_static_scratch() {
  float* p = nullptr;
  (void)static_scratch(&p, nullptr, true);
}
}  // namespace d3pm

extern "C" int d3pm_hybrid_loss_static(const d3pm_tables_f32* tb, const float* logits, long ld, const void* x0,
                                       int tok_bytes, const void* x_t, int xt_bytes, const void* t, int t_bytes, int B,
                                       long npos, int loss_type, float hybrid_coeff, float grad_scale,
                                       const float* grad_out, float* vb, float* ce, float* loss, float* dlogits,
                                       float* loss_mean, uint64_t* rng_counter, uint64_t rng_step, void* stream) {
  float* scratch = nullptr;
  if (int rc = static_scratch(&scratch, (hipStream_t)stream)) return rc;
  (void)take_workspace_need();
  const int rc = d3pm_hybrid_loss(tb, logits, ld, x0, tok_bytes, x_t, xt_bytes, t, t_bytes, B, npos, loss_type,
                                  hybrid_coeff, grad_scale, grad_out, vb, ce, loss, dlogits, loss_mean, rng_counter,
                                  rng_step, scratch, STATIC_SCRATCH_BYTES, stream);
  const size_t need = take_workspace_need();
  if (rc == D3PM_EINVAL && need > STATIC_SCRATCH_BYTES) {
    set_error("d3pm_hybrid_loss_static: B = %d x npos = %ld needs %zu bytes of partial sums, more than the %zu-byte "
              "static scratch holds (8 bytes per workgroup); call d3pm_hybrid_loss with a workspace of "
              "d3pm_hybrid_loss_workspace() bytes instead", B, npos, need, STATIC_SCRATCH_BYTES);
  }
  return rc;
}

static int structured_hybrid_loss(const d3pm_tables_f32* tb, const float* logits, long ld, const void* x0, int tok_bytes,
                                  const void* x_t, int xt_bytes, const void* t, int t_bytes, int B, long npos,
                                  int loss_type, float hybrid_coeff, float grad_scale, const float* grad_out, LossOut o,
                                  float* dlogits, void* workspace, size_t workspace_bytes, hipStream_t s,
                                  const struct NextQ* next_q, bool* next_done) {
  float* partials = (float*)workspace;
  const size_t partial_bytes = workspace_bytes;
  const bool vec = aligned16(logits) && ld == tb->K && (!dlogits || aligned16(dlogits));
  const int K = tb->K;
  if (vec && (K == 8 || K == 16 || K == 24 || K == 32)) {
    // one token per lane
    ImgArgs a = make_args(tb, x0, tok_bytes, x_t, xt_bytes, t, t_bytes, B, npos, 1);
    int block;
    // Up to one batch element per CU (256 of them), or a batch that nearly fills a second round of the CUs: one
    // workgroup per element (the pair kernel at K = 32).  Other batches above 256: the token-per-lane kernel in
    // 256-thread workgroups, three of them resident per CU at its 153 registers and out of step with each other,
    // each wave walking its chunk pass by pass with the next rows requested ahead.  K = 32, L = 512, one workgroup
    // per element against 256-thread workgroups: B = 320 11.9 / 9.8 us, 384 13.4 / 10.7, 448 12.3 / 15.0,
    // 512 13.9 / 14.2, 768 20.3 / 18.6, 2048 50.6 / 44.9 (0.70 / 0.79 of the HBM roofline: a 1024-thread workgroup
    // fills a CU alone and loads, computes and stores in lock step).
    const bool one_wg = B >= 192 && (B <= 256 || (B >= 432 && B <= 512));
    if (one_wg) {
      const long want = ((npos + 63) / 64) * 64;
      block = (int)(want > 512 ? 512 : want);      // (1024 threads leave 128 registers a lane: spills)
      a.chunks = 1;
      a.chunk_pos = npos;
    } else {
      block = 256;
      // (large batches: one chunk per element as soon as that gives two workgroups per CU -- no second launch)
      // (512 workgroups in all; long rows, B = 8 x 32768 positions: 24.1 -> 18.9 us against 1024)
      const TokenGrid tg = token_grid(B, npos, block, 512);
      a.chunks = tg.chunks;
      a.chunk_pos = tg.chunk_pos;
    }
    if (a.chunks > 1) {
      const size_t need = (size_t)B * a.chunks * 2 * sizeof(float);
      D3PM_CHECK_WORKSPACE(partials, partial_bytes, need);
      o.part = partials;
    }
    if (K == 32 && a.chunks == 1 && one_wg) {
      // two lanes per token: 32 tokens per wave, up to 16 waves per workgroup
      const long waves = (npos + 31) / 32;
      const int pblock = (int)(waves > 16 ? 16 : waves) * 64;
      const size_t sh = (size_t)(pblock / 64) * 32 * 144;    // a wave's image: 32 rows at a stride of 144 bytes
      // the weights of the two terms in the gradient: grad_scale / (npos ln 2) x {1, 0, hybrid_coeff} by loss type
      const float cnorm = (float)((double)grad_scale / ((double)npos * 0.6931471805599453));
      const float w_vb0 = loss_type == D3PM_LOSS_CE ? 0.0f : cnorm;
      const float w_ce0 = loss_type == D3PM_LOSS_KL ? 0.0f : (loss_type == D3PM_LOSS_CE ? cnorm : hybrid_coeff * cnorm);
      auto go_one = [&](auto grad, auto next, auto one) -> int {
        auto kern = k_hybrid_loss_pair32<decltype(grad)::value, decltype(next)::value, decltype(one)::value>;
        if (sh > 48 * 1024) {
          hipError_t er = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)sh);
          if (er != hipSuccess) return hip_fail(er, "hipFuncSetAttribute(MaxDynamicSharedMemorySize)");
        }
        hipLaunchKernelGGL(kern, dim3(B), dim3(pblock), sh, s, a, logits, loss_type, hybrid_coeff, w_vb0, grad_out,
                           o, dlogits, w_ce0, next_q ? *next_q : NextQ{});
        D3PM_LAUNCH_CHECK("k_hybrid_loss_pair32");
        return D3PM_OK;
      };
      // one pass: the workgroup's waves cover the element (pblock / 2 tokens), every wave has tokens
      auto go = [&](auto grad, auto next) -> int {
        return 2 * npos <= pblock ? go_one(grad, next, std::true_type{}) : go_one(grad, next, std::false_type{});
      };
      if (next_q && dlogits && npos <= pblock) {      // the launch that also corrupts the next batch
        *next_done = true;
        return go(std::true_type{}, std::true_type{});
      }
      return dlogits ? go(std::true_type{}, std::false_type{}) : go(std::false_type{}, std::false_type{});
    }
    auto launch_t = [&](auto kc, auto grad, auto maxt) -> int {
      constexpr int KC = decltype(kc)::value;
      auto kern = k_hybrid_loss_tpl<KC, decltype(grad)::value, decltype(maxt)::value>;
      const size_t sh = (size_t)(block / 64) * 64 * TplCfg<KC>::STRIDE;
      if (sh > 48 * 1024) {
        hipError_t er = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)sh);
        if (er != hipSuccess) return hip_fail(er, "hipFuncSetAttribute(MaxDynamicSharedMemorySize)");
      }
      hipLaunchKernelGGL(kern, dim3(B * a.chunks), dim3(block), sh, s, a, logits, loss_type, hybrid_coeff, grad_scale,
                         grad_out, o, dlogits);
      D3PM_LAUNCH_CHECK("k_hybrid_loss_tpl");
      return D3PM_OK;
    };
    auto launch = [&](auto kc, auto grad) -> int {          // block <= 512 always (see above)
      return launch_t(kc, grad, std::integral_constant<int, 512>{});
    };
    auto by_grad = [&](auto kc) -> int {
      return dlogits ? launch(kc, std::true_type{}) : launch(kc, std::false_type{});
    };
    int rc;
    switch (K) {
      case 8: rc = by_grad(std::integral_constant<int, 8>{}); break;
      case 16: rc = by_grad(std::integral_constant<int, 16>{}); break;
      case 24: rc = by_grad(std::integral_constant<int, 24>{}); break;
      default: rc = by_grad(std::integral_constant<int, 32>{}); break;
    }
    if (rc) return rc;
    if (a.chunks > 1) {
      hipLaunchKernelGGL(k_loss_finalize, dim3((B + 255) / 256), dim3(256), 0, s, o, B, a.chunks, npos, loss_type,
                         hybrid_coeff, 1.0f);
      D3PM_LAUNCH_CHECK("k_loss_finalize");
    }
    return D3PM_OK;
  }
  if (vec && (K == 64 || K == 128 || K == 256) && npos * (long)K * 4 < (1L << 31)) {
    // 16 classes a lane, K / 16 lanes a token: 256-thread workgroups over chunks of the batch elements, about 1024
    // of them (four resident per CU), and k_loss_finalize for the chunks' sums.  A caller with a batch that fills the
    // chip and no workspace gets one workgroup of 16 waves per batch element, which writes its losses itself
    // (B = 256, L = 512: 44.7 against 45.5 us at K = 256, 24.9 / 25.0 at K = 128, 17.6 / 14.4 at K = 64).
    const int G = K / 16;
    ImgArgs a = make_args(tb, x0, tok_bytes, x_t, xt_bytes, t, t_bytes, B, npos, 1);
    const TokenGrid tg = token_grid(B, npos, 256 / G, 1024);
    const bool one_wg = B >= 192 && tg.chunks > 1 &&
                        (!partials || partial_bytes < (size_t)B * tg.chunks * 2 * sizeof(float));
    if (!one_wg) {
      a.chunks = tg.chunks;
      a.chunk_pos = tg.chunk_pos;
    }
    if (a.chunks > 1) {
      const size_t need = (size_t)B * a.chunks * 2 * sizeof(float);
      D3PM_CHECK_WORKSPACE(partials, partial_bytes, need);
      o.part = partials;
    }
    const float cnorm = (float)((double)grad_scale / ((double)npos * 0.6931471805599453));
    const float w_vb0 = loss_type == D3PM_LOSS_CE ? 0.0f : cnorm;
    const float w_ce0 = loss_type == D3PM_LOSS_KL ? 0.0f : (loss_type == D3PM_LOSS_CE ? cnorm : hybrid_coeff * cnorm);
    auto go = [&](auto gc, auto grad, auto nw) {
      constexpr int NW = decltype(nw)::value;
      hipLaunchKernelGGL((k_hybrid_loss_wide<decltype(gc)::value, decltype(grad)::value, NW>), dim3(B * a.chunks),
                         dim3(64 * NW), 0, s, a, logits, loss_type, hybrid_coeff, w_vb0, grad_out, o, dlogits, w_ce0);
    };
    auto by_nw = [&](auto gc, auto grad) {
      if (one_wg) go(gc, grad, std::integral_constant<int, 16>{});
      else go(gc, grad, std::integral_constant<int, 4>{});
    };
    auto by_grad = [&](auto gc) { if (dlogits) by_nw(gc, std::true_type{}); else by_nw(gc, std::false_type{}); };
    if (K == 64) by_grad(std::integral_constant<int, 4>{});
    else if (K == 128) by_grad(std::integral_constant<int, 8>{});
    else by_grad(std::integral_constant<int, 16>{});
    D3PM_LAUNCH_CHECK("k_hybrid_loss_wide");
    if (a.chunks > 1) {
      hipLaunchKernelGGL(k_loss_finalize, dim3((B + 255) / 256), dim3(256), 0, s, o, B, a.chunks, npos, loss_type,
                         hybrid_coeff, 1.0f);
      D3PM_LAUNCH_CHECK("k_loss_finalize");
    }
    return D3PM_OK;
  }
  return dispatch_layout_wide(tb->K, vec, [&](auto lay) {
    using L = decltype(lay);
    ImgArgs a = make_args(tb, x0, tok_bytes, x_t, xt_bytes, t, t_bytes, B, npos, L::G);
    // A batch that fills the chip on its own gets one workgroup per batch element
    // (sized so each lane group makes one PIPE-deep trip) and writes its losses
    // directly; small batches are split over chunks and summed by k_loss_finalize.
    int block = BLOCK;
    if (B >= 192) {
      constexpr int tpg = 4;
      long want = ((npos + tpg - 1) / tpg) * L::G;       // lane groups make ~tpg trips each
      want = ((want + 63) / 64) * 64;
      const long cap = loss_max_threads<L>();
      if (want < tb->K) want = ((tb->K + 63) / 64) * 64;      // one thread per class stages the digests
      block = (int)(want < 64 ? 64 : (want > cap ? cap : want));
      a.chunks = 1;
      a.chunk_pos = npos;
    } else {
      const TokenGrid tg = token_grid(B, npos, pipe_depth<L>() * (BLOCK / L::G), 1024);
      a.chunks = tg.chunks;
      a.chunk_pos = tg.chunk_pos;
    }
    if (a.chunks > 1) {
      const size_t need = (size_t)B * a.chunks * 2 * sizeof(float);
      D3PM_CHECK_WORKSPACE(partials, partial_bytes, need);
      o.part = partials;
    }
    const dim3 grid(B * a.chunks), blk(block);
    if (dlogits)
      hipLaunchKernelGGL((k_hybrid_loss<L, true>), grid, blk, 0, s, a, logits, ld, loss_type, hybrid_coeff, grad_scale,
                         grad_out, o, dlogits);
    else
      hipLaunchKernelGGL((k_hybrid_loss<L, false>), grid, blk, 0, s, a, logits, ld, loss_type, hybrid_coeff, grad_scale,
                         grad_out, o, dlogits);
    D3PM_LAUNCH_CHECK("k_hybrid_loss");
    if (a.chunks > 1) {
      hipLaunchKernelGGL(k_loss_finalize, dim3((B + 255) / 256), dim3(256), 0, s, o, B, a.chunks, npos, loss_type,
                         hybrid_coeff, 1.0f);
      D3PM_LAUNCH_CHECK("k_loss_finalize");
    }
    return D3PM_OK;
  });
}

extern "C" int d3pm_scale_rows(float* g, const float* upstream, float baked, int B, long row_elems, void* stream) {
  D3PM_CHECK_ARG(g && upstream, "null pointer");
  D3PM_CHECK_ARG(baked != 0.0f, "baked scale must be non-zero");
  D3PM_CHECK_ARG(B >= 0 && row_elems >= 0, "negative size");
  if (B == 0 || row_elems == 0) return D3PM_OK;
  int bpr = (int)((2048 + B - 1) / B);
  const long max_bpr = (row_elems + 1023) / 1024;
  if (bpr > max_bpr) bpr = (int)max_bpr;
  if (bpr < 1) bpr = 1;
  // keep chunk starts 4-element aligned so the float4 path applies
  hipLaunchKernelGGL(k_scale_rows, dim3(B * bpr), dim3(256), 0, (hipStream_t)stream, g, upstream, baked, row_elems, bpr);
  D3PM_LAUNCH_CHECK("k_scale_rows");
  return D3PM_OK;
}

extern "C" int d3pm_p_sample(const d3pm_tables_f32* tb, const float* logits, long ld, const void* x_t, int xt_bytes,
                             const void* t, int t_bytes, int B, long npos, const float* noise, int noise_kind,
                             uint64_t seed, uint64_t offset, const uint64_t* offset_dev, int64_t* sample, float* probs,
                             void* stream) {
  if (int rc = check_common(tb, B, npos)) return rc;
  D3PM_CHECK_ARG(bytes_ok(xt_bytes) && bytes_ok(t_bytes), "token/timestep width must be 4 or 8 bytes");
  D3PM_CHECK_ARG(ld >= tb->K, "row stride %ld smaller than K=%d", ld, tb->K);
  D3PM_CHECK_ARG(noise_kind == D3PM_NOISE_UNIFORM || noise_kind == D3PM_NOISE_GUMBEL || noise_kind == D3PM_NOISE_PHILOX,
                 "noise_kind %d not valid for p_sample", noise_kind);
  if (B == 0 || npos == 0) return D3PM_OK;
  D3PM_CHECK_ARG(logits && x_t && t && sample, "null pointer");
  D3PM_CHECK_ARG(noise_kind == D3PM_NOISE_PHILOX || noise, "noise tensor is null");
  hipStream_t s = (hipStream_t)stream;
  if (!tb->structured)
    return dense_p_sample(tb, logits, ld, x_t, xt_bytes, t, t_bytes, B, npos, noise, noise_kind, seed, offset, offset_dev,
                          sample, probs, s);
  const bool vec = aligned16(logits) && ld == tb->K && (noise_kind == D3PM_NOISE_PHILOX || aligned16(noise)) &&
                   (!probs || aligned16(probs));
  if (noise_kind == D3PM_NOISE_PHILOX && !probs && vec && tb->K == 32 && npos >= 64 && B <= (1 << 20) &&
      (long)B * npos * ld < 0x3fffffffL) {
    // the sampling loop's call: tokens of the whole batch as one row, 64 per wave
    ImgArgs a = make_args(tb, nullptr, 8, x_t, xt_bytes, t, t_bytes, B, npos, 1);
    const unsigned ntok = (unsigned)((long)B * npos), waves = (ntok + 63) / 64, wpb = 4;
    hipLaunchKernelGGL((k_p_sample_flat<32>), dim3((waves + wpb - 1) / wpb), dim3(64 * wpb),
                       (size_t)wpb * RowImage<32>::BYTES, s, a, logits, seed, offset, offset_dev, sample, ntok,
                       1.0f / (float)npos);
    D3PM_LAUNCH_CHECK("k_p_sample_flat");
    return D3PM_OK;
  }
  if (noise_kind == D3PM_NOISE_PHILOX && !probs && vec && (tb->K == 64 || tb->K == 128 || tb->K == 256)) {
    // the sampling loop's call at K = 64 / 128 / 256: a token on 8 or 16 lanes (8 or 16 classes a lane, interleaved
    // 16-byte pieces) instead of 16 / 32 / 64, so that the softmax reductions, the prefix scan of the draw and its
    // search stay inside one DPP row and a wave takes 4 or 8 tokens per pass
    auto rows = [&](auto lay) {
      using L = decltype(lay);
      ImgArgs a = make_args(tb, nullptr, 8, x_t, xt_bytes, t, t_bytes, B, npos, L::G);
      hipLaunchKernelGGL((k_p_sample<L, D3PM_NOISE_PHILOX>), dim3(B * a.chunks), dim3(BLOCK), 0, s, a, logits, ld, noise,
                         seed, offset, offset_dev, sample, probs);
      D3PM_LAUNCH_CHECK("k_p_sample");
      return D3PM_OK;
    };
    if (tb->K == 256) return rows(Layout<16, 16, true, true>{});
    if (tb->K == 128) return rows(Layout<16, 8, true, true>{});
    return rows(Layout<8, 8, true, true>{});
  }
  return dispatch_layout(tb->K, vec, [&](auto lay) {
    using L = decltype(lay);
    ImgArgs a = make_args(tb, nullptr, 8, x_t, xt_bytes, t, t_bytes, B, npos, L::G);
    const dim3 grid(B * a.chunks), block(BLOCK);
    if (noise_kind == D3PM_NOISE_UNIFORM)
      hipLaunchKernelGGL((k_p_sample<L, D3PM_NOISE_UNIFORM>), grid, block, 0, s, a, logits, ld, noise, seed, offset,
                         offset_dev, sample, probs);
    else if (noise_kind == D3PM_NOISE_GUMBEL)
      hipLaunchKernelGGL((k_p_sample<L, D3PM_NOISE_GUMBEL>), grid, block, 0, s, a, logits, ld, noise, seed, offset,
                         offset_dev, sample, probs);
    else
      hipLaunchKernelGGL((k_p_sample<L, D3PM_NOISE_PHILOX>), grid, block, 0, s, a, logits, ld, noise, seed, offset,
                         offset_dev, sample, probs);
    D3PM_LAUNCH_CHECK("k_p_sample");
    return D3PM_OK;
  });
}

extern "C" int d3pm_prior_bpd(const d3pm_tables_f32* tb, const void* x0, int tok_bytes, int B, long npos, int absorbing,
                              float* out, void* workspace, size_t workspace_bytes, void* stream) {
  if (int rc = check_common(tb, B, npos)) return rc;
  D3PM_CHECK_ARG(bytes_ok(tok_bytes), "token width must be 4 or 8 bytes");
  if (B == 0) return D3PM_OK;
  D3PM_CHECK_ARG(npos > 0, "empty sample: mean over zero positions");
  D3PM_CHECK_ARG(x0 && out, "null pointer");
  hipStream_t s = (hipStream_t)stream;
  if (!tb->structured) return dense_prior(tb, x0, tok_bytes, B, npos, absorbing, out, workspace, workspace_bytes, s);
  return dispatch_layout(tb->K, false, [&](auto lay) {
    using L = decltype(lay);
    ImgArgs a = make_args(tb, x0, tok_bytes, nullptr, 8, nullptr, 8, B, npos, L::G);
    const size_t need = (size_t)B * a.chunks * 2 * sizeof(float);
    D3PM_CHECK_WORKSPACE(workspace, workspace_bytes, need);
    float* part = (float*)workspace;
    hipLaunchKernelGGL(k_prior<L>, dim3(B * a.chunks), dim3(BLOCK), 0, s, a, absorbing, part);
    D3PM_LAUNCH_CHECK("k_prior");
    hipLaunchKernelGGL(k_loss_finalize, dim3((B + 255) / 256), dim3(256), 0, s,
                       LossOut{out, nullptr, nullptr, part, nullptr, 0, nullptr, 0.0f}, B,
                       a.chunks, npos, D3PM_LOSS_KL, 0.0f, 1.0f / LN2);
    D3PM_LAUNCH_CHECK("k_loss_finalize");
    return D3PM_OK;
  });
}
