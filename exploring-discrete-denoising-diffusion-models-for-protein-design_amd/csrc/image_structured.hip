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
  return digest + ((size_t)t * 4 + c) * K;
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
  const int t = load_tok(a.t, b, a.t_bytes);
  float d[L::V], o[L::V];
  load_digest<L>(a.tb.qbar_digest, t, 0, K, g, d);
  load_digest<L>(a.tb.qbar_digest, t, 1, K, g, o);
  const long p1 = min(a.npos, (chunk + 1) * a.chunk_pos);
  for (long pos = chunk * a.chunk_pos + slot; pos < p1; pos += BLOCK / L::G) {
    const long n = (long)b * a.npos + pos;
    const int x0 = load_tok(a.x0, n, a.x0_bytes);
    float r[L::V];
#pragma unroll
    for (int v = 0; v < L::V; ++v) r[v] = (L::cls(g, v) == x0) ? d[v] : o[v];
    store_slots<L>(out + n * K, g, K, r);
  }
}

// ------------------------------------------------------------------ q_sample
template <class L, int NOISE>
__global__ __launch_bounds__(BLOCK) void k_q_sample(ImgArgs a, const float* __restrict__ noise, uint64_t seed,
                                                    uint64_t offset, int64_t* __restrict__ out) {
  const int K = a.tb.K;
  const int b = blockIdx.x / a.chunks, chunk = blockIdx.x % a.chunks;
  const int g = threadIdx.x % L::G, slot = threadIdx.x / L::G;
  const int t = load_tok(a.t, b, a.t_bytes);
  float ld_[L::V], lo_[L::V];
  load_digest<L>(a.tb.qbar_digest, t, 2, K, g, ld_);
  load_digest<L>(a.tb.qbar_digest, t, 3, K, g, lo_);
  const long p1 = min(a.npos, (chunk + 1) * a.chunk_pos);
  for (long pos = chunk * a.chunk_pos + slot; pos < p1; pos += BLOCK / L::G) {
    const long n = (long)b * a.npos + pos;
    const int x0 = load_tok(a.x0, n, a.x0_bytes);
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
      const float gum = (NOISE == D3PM_NOISE_GUMBEL) ? u[v] : gumbel_from_uniform(u[v]);
      const float val = ((c == x0) ? ld_[v] : lo_[v]) + gum;
      const bool ok = c < K;
      // NaN-free inputs: strict > keeps the first maximal class of this lane
      if (ok && (val > best || besti == 0x7fffffff)) { best = val; besti = c; }
    }
    group_argmax<L::G>(best, besti);
    if (g == 0) out[n] = besti;
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
  const int t = load_tok(a.t, b, a.t_bytes);
  Post<L> P;
  if (t > 0) load_post<L>(a.tb, t, g, P);
  const float log_one = (float)log((double)(1.0f + eps)), log_zero = (float)log((double)eps);
  const long p1 = min(a.npos, (chunk + 1) * a.chunk_pos);
  for (long pos = chunk * a.chunk_pos + slot; pos < p1; pos += BLOCK / L::G) {
    const long n = (long)b * a.npos + pos;
    const int xt = load_tok(a.xt, n, a.xt_bytes);
    float r[L::V];
    float z[L::V];
    int x0 = -1;
    if constexpr (FROM_LOGITS) load_slots<L>(xs_logits + n * ld, g, K, -INFINITY, z);
    else x0 = load_tok(a.x0, n, a.x0_bytes);
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

// ------------------------------------------------------------------ fused hybrid loss, forward + backward
// Per token (t > 0), with pi = softmax(z), A the column x_t of Q_t, Qm = Qbar_{t-1}:
//   f2 = pi Qm,  m = log(A+eps) + log(f2+eps),  tl = log(A+eps) + log(Qm[x0,:]+eps)
//   kl = sum softmax(tl+eps) (lsm(tl+eps) - lsm(m+eps)),  ce = -lsm(z)[x0]
//   d kl/d m = softmax(m) - softmax(tl);  chain through f2 and the softmax of z.
// t == 0: kl is replaced by ce.  Partial sums per workgroup go to `part`
// ([B][chunks][2]: vb numerator, ce numerator), combined by k_loss_finalize in a
// fixed order so the result does not depend on scheduling.
template <class L, bool GRAD>
__global__ __launch_bounds__(BLOCK) void k_hybrid_loss(ImgArgs a, const float* __restrict__ logits, long ld,
                                                       int loss_type, float hybrid_coeff, float grad_scale,
                                                       const float* __restrict__ grad_out, float* __restrict__ part,
                                                       float* __restrict__ dlogits) {
  __shared__ float red[2 * (BLOCK / 64)];
  const int K = a.tb.K;
  const float eps = a.tb.eps;
  const int b = blockIdx.x / a.chunks, chunk = blockIdx.x % a.chunks;
  const int g = threadIdx.x % L::G, slot = threadIdx.x / L::G;
  const int t = load_tok(a.t, b, a.t_bytes);
  const bool need_post = (t > 0) && loss_type != D3PM_LOSS_CE;
  Post<L> P;
  if (need_post) load_post<L>(a.tb, t, g, P);
  const float gb = grad_scale * ((GRAD && grad_out) ? grad_out[b] : 1.0f);
  const float cnorm = gb / ((float)a.npos * LN2);
  // weights of d(vb)/dz and d(ce)/dz in the requested loss
  const float w_vb = (loss_type == D3PM_LOSS_CE) ? 0.0f : cnorm;
  const float w_ce = (loss_type == D3PM_LOSS_KL) ? 0.0f : (loss_type == D3PM_LOSS_CE ? cnorm : hybrid_coeff * cnorm);
  float acc_vb = 0, acc_ce = 0;
  const long p1 = min(a.npos, (chunk + 1) * a.chunk_pos);
  for (long pos = chunk * a.chunk_pos + slot; pos < p1; pos += BLOCK / L::G) {
    const long n = (long)b * a.npos + pos;
    const int x0 = load_tok(a.x0, n, a.x0_bytes);
    float z[L::V], pi[L::V], zmax, zsum;
    load_slots<L>(logits + n * ld, g, K, -INFINITY, z);
    group_softmax<L>(z, pi, zmax, zsum);
    float zx0 = 0;
#pragma unroll
    for (int v = 0; v < L::V; ++v) zx0 += (L::cls(g, v) == x0) ? z[v] : 0.0f;
    zx0 = group_sum<L::G>(zx0);
    const float ce = (zmax + logf(zsum)) - zx0;
    float gz[L::V];
    float vb;
    if (!need_post) {
      vb = ce;
      if constexpr (GRAD) {
        const float w = (t == 0 ? w_vb : 0.0f) + w_ce;
#pragma unroll
        for (int v = 0; v < L::V; ++v) gz[v] = w * (pi[v] - ((L::cls(g, v) == x0) ? 1.0f : 0.0f));
      }
    } else {
      const int xt = load_tok(a.xt, n, a.xt_bytes);
      const float l1d = dg(a.tb.qone_digest, t, 2, K)[xt], l1o = dg(a.tb.qone_digest, t, 3, K)[xt];
      float f2e[L::V], lm[L::V], tl[L::V];
      float mmax = -INFINITY, tmax = -INFINITY;
#pragma unroll
      for (int v = 0; v < L::V; ++v) {
        const int c = L::cls(g, v);
        const bool ok = L::VEC || c < K;
        const float l1 = (c == xt) ? l1d : l1o;
        f2e[v] = fmaf(P.bd[v] - P.bo[v], pi[v], P.bo[v]) + eps;
        lm[v] = ok ? (l1 + logf(f2e[v])) + eps : -INFINITY;
        tl[v] = ok ? (l1 + ((c == x0) ? P.lbd[v] : P.lbo[v])) + eps : -INFINITY;
        mmax = fmaxf(mmax, lm[v]);
        tmax = fmaxf(tmax, tl[v]);
      }
      mmax = group_max<L::G>(mmax);
      tmax = group_max<L::G>(tmax);
      float em[L::V], et[L::V], sm = 0, st = 0;
#pragma unroll
      for (int v = 0; v < L::V; ++v) {
        em[v] = expf(lm[v] - mmax); sm += em[v];
        et[v] = expf(tl[v] - tmax); st += et[v];
      }
      sm = group_sum<L::G>(sm);
      st = group_sum<L::G>(st);
      const float lsm_off = mmax + logf(sm), lst_off = tmax + logf(st);
      const float ism = 1.0f / sm, ist = 1.0f / st;
      float kl = 0;
#pragma unroll
      for (int v = 0; v < L::V; ++v) {
        const bool ok = L::VEC || L::cls(g, v) < K;
        const float q = et[v] * ist;
        kl += ok ? q * ((tl[v] - lst_off) - (lm[v] - lsm_off)) : 0.0f;
      }
      vb = group_sum<L::G>(kl);
      if constexpr (GRAD) {
        float gf2[L::V], R = 0, dot = 0;
#pragma unroll
        for (int v = 0; v < L::V; ++v) {
          const float gm = (em[v] * ism - et[v] * ist) * w_vb;
          gf2[v] = gm / f2e[v];
          R += gf2[v] * P.bo[v];
        }
        R = group_sum<L::G>(R);
        float gpi[L::V];
#pragma unroll
        for (int v = 0; v < L::V; ++v) {
          gpi[v] = fmaf(gf2[v], P.bd[v] - P.bo[v], R);
          dot += pi[v] * gpi[v];
        }
        dot = group_sum<L::G>(dot);
#pragma unroll
        for (int v = 0; v < L::V; ++v)
          gz[v] = pi[v] * (gpi[v] - dot) + w_ce * (pi[v] - ((L::cls(g, v) == x0) ? 1.0f : 0.0f));
      }
    }
    if constexpr (GRAD) {
      store_slots<L>(dlogits + n * ld, g, K, gz);
      if (!L::VEC && ld > K) zero_tail<L>(dlogits + n * ld, g, K, ld);
    }
    if (g == 0) { acc_vb += vb; acc_ce += ce; }
  }
  // workgroup partial sums (fixed order: lanes by butterfly, waves in index order)
  acc_vb = wave_sum(acc_vb);
  acc_ce = wave_sum(acc_ce);
  const int wave = threadIdx.x >> 6;
  if ((threadIdx.x & 63) == 0) { red[2 * wave] = acc_vb; red[2 * wave + 1] = acc_ce; }
  __syncthreads();
  if (threadIdx.x == 0) {
    float svb = 0, sce = 0;
    for (int w = 0; w < BLOCK / 64; ++w) { svb += red[2 * w]; sce += red[2 * w + 1]; }
    part[((size_t)b * a.chunks + chunk) * 2 + 0] = svb;
    part[((size_t)b * a.chunks + chunk) * 2 + 1] = sce;
  }
}

__global__ void k_loss_finalize(const float* __restrict__ part, int B, int chunks, long npos, int loss_type,
                                float hybrid_coeff, float* __restrict__ vb, float* __restrict__ ce,
                                float* __restrict__ loss) {
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= B) return;
  float svb = 0, sce = 0;
  for (int c = 0; c < chunks; ++c) {
    svb += part[((size_t)b * chunks + c) * 2 + 0];
    sce += part[((size_t)b * chunks + c) * 2 + 1];
  }
  const float v = svb / (float)npos / LN2, c_ = sce / (float)npos / LN2;
  if (vb) vb[b] = v;
  if (ce) ce[b] = c_;
  if (loss) loss[b] = loss_type == D3PM_LOSS_KL ? v : (loss_type == D3PM_LOSS_CE ? c_ : v + hybrid_coeff * c_);
}

// ------------------------------------------------------------------ p_sample
template <class L, int NOISE>
__global__ __launch_bounds__(BLOCK) void k_p_sample(ImgArgs a, const float* __restrict__ logits, long ld,
                                                    const float* __restrict__ noise, uint64_t seed, uint64_t offset,
                                                    int64_t* __restrict__ sample, float* __restrict__ probs) {
  const int K = a.tb.K;
  const float eps = a.tb.eps;
  const int b = blockIdx.x / a.chunks, chunk = blockIdx.x % a.chunks;
  const int g = threadIdx.x % L::G, slot = threadIdx.x / L::G;
  const int t = load_tok(a.t, b, a.t_bytes);
  Post<L> P;
  if (t > 0) load_post<L>(a.tb, t, g, P);
  const long p1 = min(a.npos, (chunk + 1) * a.chunk_pos);
  for (long pos = chunk * a.chunk_pos + slot; pos < p1; pos += BLOCK / L::G) {
    const long n = (long)b * a.npos + pos;
    float z[L::V], pi[L::V], zmax, zsum;
    load_slots<L>(logits + n * ld, g, K, -INFINITY, z);
    const bool want_pi = (t > 0) || probs;
    if (want_pi) group_softmax<L>(z, pi, zmax, zsum);
    if (probs) store_slots<L>(probs + n * K, g, K, pi);
    float best = -INFINITY;
    int besti = 0x7fffffff;
    if (t == 0) {   // argmax of the logits themselves, no noise (mask of :394)
#pragma unroll
      for (int v = 0; v < L::V; ++v) {
        const int c = L::cls(g, v);
        if (c < K && (z[v] > best || besti == 0x7fffffff)) { best = z[v]; besti = c; }
      }
    } else {
      const int xt = load_tok(a.xt, n, a.xt_bytes);
      const float l1d = dg(a.tb.qone_digest, t, 2, K)[xt], l1o = dg(a.tb.qone_digest, t, 3, K)[xt];
      float u[L::V];
      if constexpr (NOISE == D3PM_NOISE_PHILOX) {
        const U4 r = philox4x32_10((uint64_t)n * L::G + g, offset, seed);
        const uint32_t rr[4] = {r.x, r.y, r.z, r.w};
#pragma unroll
        for (int v = 0; v < L::V; ++v) u[v] = u01(rr[v]);
      } else {
        load_slots<L>(noise + n * K, g, K, 0.5f, u);
      }
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
    const int x0 = load_tok(a.x0, n, a.x0_bytes);
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
                             int64_t* x_t, void* stream) {
  if (int rc = check_common(tb, B, npos)) return rc;
  D3PM_CHECK_ARG(bytes_ok(tok_bytes) && bytes_ok(t_bytes), "token/timestep width must be 4 or 8 bytes");
  D3PM_CHECK_ARG(noise_kind == D3PM_NOISE_UNIFORM || noise_kind == D3PM_NOISE_GUMBEL || noise_kind == D3PM_NOISE_PHILOX,
                 "noise_kind %d not valid for q_sample", noise_kind);
  if (B == 0 || npos == 0) return D3PM_OK;
  D3PM_CHECK_ARG(x0 && t && x_t, "null pointer");
  D3PM_CHECK_ARG(noise_kind == D3PM_NOISE_PHILOX || noise, "noise tensor is null");
  hipStream_t s = (hipStream_t)stream;
  if (!tb->structured)
    return dense_q_sample(tb, x0, tok_bytes, t, t_bytes, B, npos, noise, noise_kind, seed, offset, x_t, s);
  return dispatch_layout(tb->K, noise_kind == D3PM_NOISE_PHILOX || aligned16(noise), [&](auto lay) {
    using L = decltype(lay);
    ImgArgs a = make_args(tb, x0, tok_bytes, nullptr, 8, t, t_bytes, B, npos, L::G);
    const dim3 grid(B * a.chunks), block(BLOCK);
    if (noise_kind == D3PM_NOISE_UNIFORM)
      hipLaunchKernelGGL((k_q_sample<L, D3PM_NOISE_UNIFORM>), grid, block, 0, s, a, noise, seed, offset, x_t);
    else if (noise_kind == D3PM_NOISE_GUMBEL)
      hipLaunchKernelGGL((k_q_sample<L, D3PM_NOISE_GUMBEL>), grid, block, 0, s, a, noise, seed, offset, x_t);
    else
      hipLaunchKernelGGL((k_q_sample<L, D3PM_NOISE_PHILOX>), grid, block, 0, s, a, noise, seed, offset, x_t);
    D3PM_LAUNCH_CHECK("k_q_sample");
    return D3PM_OK;
  });
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

extern "C" size_t d3pm_hybrid_loss_workspace(int B, long npos, int K) {
  (void)K;
  // one (vb, ce) pair per workgroup; the grid never has more than
  // max(2048, B) * 2 workgroups (see token_grid), bounded here generously.
  long passes = npos > 0 ? npos : 1;
  long chunks = passes < 4096 ? passes : 4096;
  return (size_t)(B > 0 ? B : 1) * (size_t)chunks * 2 * sizeof(float);
}

extern "C" int d3pm_hybrid_loss(const d3pm_tables_f32* tb, const float* logits, long ld, const void* x0, int tok_bytes,
                                const void* x_t, int xt_bytes, const void* t, int t_bytes, int B, long npos,
                                int loss_type, float hybrid_coeff, float grad_scale, const float* grad_out, float* vb,
                                float* ce, float* loss, float* dlogits, void* workspace, size_t workspace_bytes,
                                void* stream) {
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
  D3PM_CHECK_ARG(logits && x0 && x_t && t, "null pointer");
  hipStream_t s = (hipStream_t)stream;
  if (!tb->structured)
    return dense_hybrid_loss(tb, logits, ld, x0, tok_bytes, x_t, xt_bytes, t, t_bytes, B, npos, loss_type, hybrid_coeff,
                             grad_scale, grad_out, vb, ce, loss, dlogits, workspace, workspace_bytes, s);
  const bool vec = aligned16(logits) && ld == tb->K && (!dlogits || aligned16(dlogits));
  return dispatch_layout(tb->K, vec, [&](auto lay) {
    using L = decltype(lay);
    ImgArgs a = make_args(tb, x0, tok_bytes, x_t, xt_bytes, t, t_bytes, B, npos, L::G);
    const size_t need = (size_t)B * a.chunks * 2 * sizeof(float);
    D3PM_CHECK_ARG(workspace && workspace_bytes >= need, "workspace too small: %zu < %zu bytes", workspace_bytes, need);
    float* part = (float*)workspace;
    const dim3 grid(B * a.chunks), block(BLOCK);
    if (dlogits)
      hipLaunchKernelGGL((k_hybrid_loss<L, true>), grid, block, 0, s, a, logits, ld, loss_type, hybrid_coeff, grad_scale,
                         grad_out, part, dlogits);
    else
      hipLaunchKernelGGL((k_hybrid_loss<L, false>), grid, block, 0, s, a, logits, ld, loss_type, hybrid_coeff, grad_scale,
                         grad_out, part, dlogits);
    D3PM_LAUNCH_CHECK("k_hybrid_loss");
    hipLaunchKernelGGL(k_loss_finalize, dim3((B + 255) / 256), dim3(256), 0, s, part, B, a.chunks, npos, loss_type,
                       hybrid_coeff, vb, ce, loss);
    D3PM_LAUNCH_CHECK("k_loss_finalize");
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
                             uint64_t seed, uint64_t offset, int64_t* sample, float* probs, void* stream) {
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
    return dense_p_sample(tb, logits, ld, x_t, xt_bytes, t, t_bytes, B, npos, noise, noise_kind, seed, offset, sample,
                          probs, s);
  const bool vec = aligned16(logits) && ld == tb->K && (noise_kind == D3PM_NOISE_PHILOX || aligned16(noise)) &&
                   (!probs || aligned16(probs));
  return dispatch_layout(tb->K, vec, [&](auto lay) {
    using L = decltype(lay);
    ImgArgs a = make_args(tb, nullptr, 8, x_t, xt_bytes, t, t_bytes, B, npos, L::G);
    const dim3 grid(B * a.chunks), block(BLOCK);
    if (noise_kind == D3PM_NOISE_UNIFORM)
      hipLaunchKernelGGL((k_p_sample<L, D3PM_NOISE_UNIFORM>), grid, block, 0, s, a, logits, ld, noise, seed, offset,
                         sample, probs);
    else if (noise_kind == D3PM_NOISE_GUMBEL)
      hipLaunchKernelGGL((k_p_sample<L, D3PM_NOISE_GUMBEL>), grid, block, 0, s, a, logits, ld, noise, seed, offset,
                         sample, probs);
    else
      hipLaunchKernelGGL((k_p_sample<L, D3PM_NOISE_PHILOX>), grid, block, 0, s, a, logits, ld, noise, seed, offset,
                         sample, probs);
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
    D3PM_CHECK_ARG(workspace && workspace_bytes >= need, "workspace too small: %zu < %zu bytes", workspace_bytes, need);
    float* part = (float*)workspace;
    hipLaunchKernelGGL(k_prior<L>, dim3(B * a.chunks), dim3(BLOCK), 0, s, a, absorbing, part);
    D3PM_LAUNCH_CHECK("k_prior");
    hipLaunchKernelGGL(k_loss_finalize, dim3((B + 255) / 256), dim3(256), 0, s, part, B, a.chunks, npos, D3PM_LOSS_KL,
                       0.0f, out, (float*)nullptr, (float*)nullptr);
    D3PM_LAUNCH_CHECK("k_loss_finalize");
    return D3PM_OK;
  });
}
