// Protein-style (EvoDiff-like) D3PM kernels: batched forward corruption, the LVB
// loss with its gradient, and the reverse step.  Arithmetic after the float32
// softmax is float64, as in protein_generation/losses.py:65 and
// generate.py:30.  Tables are column-structured digests (uniform schedule):
// entry(i,j) = i==j ? diag[j] : off[j], which turns the reference's [L,K,K]
// temporaries into O(K) work per token:
//   s_j = sum_i p_i^2 Qm[i,j] = off_j * (S2 - p_j^2) + diag_j * p_j^2.
#include "layout.h"

namespace d3pm {

constexpr int PBLOCK = 256;

struct ProtArgs {
  d3pm_tables_f64 tb;
  const int64_t* x0;
  const int64_t* xt;
  const int64_t* t;
  int B; long npos;
  int chunks; long chunk_pos;
};

__device__ __forceinline__ const double* dgd(const double* digest, int t, int c, int K) {
  return digest + ((size_t)t * 4 + c) * K;        // c: 0 diag, 1 off, 2 log(diag), 3 log(off)
}

template <class L>
__device__ __forceinline__ void load_digest_d(const double* digest, int t, int c, int K, int g, double (&out)[L::V]) {
  load_slots<L>(dgd(digest, t, c, K), g, K, 0.0, out);
}

// ------------------------------------------------------------------ forward corruption
template <class L>
__global__ __launch_bounds__(PBLOCK) void k_corrupt(ProtArgs a, const double* __restrict__ expo, uint64_t seed,
                                                    uint64_t offset, int64_t* __restrict__ x_t,
                                                    double* __restrict__ probs) {
  const int K = a.tb.K;
  const int b = blockIdx.x / a.chunks, chunk = blockIdx.x % a.chunks;
  const int g = threadIdx.x % L::G, slot = threadIdx.x / L::G;
  const int t = (int)a.t[b];
  double d[L::V], o[L::V];
  load_digest_d<L>(a.tb.qbar_digest, t, 0, K, g, d);
  load_digest_d<L>(a.tb.qbar_digest, t, 1, K, g, o);
  const long p1 = min(a.npos, (chunk + 1) * a.chunk_pos);
  for (long pos = chunk * a.chunk_pos + slot; pos < p1; pos += PBLOCK / L::G) {
    const long n = (long)b * a.npos + pos;
    const int x0 = (int)a.x0[n];
    double pv[L::V];
#pragma unroll
    for (int v = 0; v < L::V; ++v) {
      const int c = L::cls(g, v);
      pv[v] = (c < K) ? ((c == x0) ? d[v] : o[v]) : 0.0;
      if (probs && c < K) probs[n * K + c] = pv[v];
    }
    int besti;
    if (expo) {
      // the reference's draw: torch.multinomial on CPU = argmax_j(p_j / E_j), E ~ Exp(1) passed in
      double e[L::V];
      load_slots<L>(expo + n * K, g, K, 1.0, e);
      double best = -INFINITY;
      besti = 0x7fffffff;
#pragma unroll
      for (int v = 0; v < L::V; ++v) {
        const int c = L::cls(g, v);
        const double val = pv[v] / e[v];
        if (c < K && (val > best || besti == 0x7fffffff)) { best = val; besti = c; }
      }
      group_argmax<L::G>(best, besti);
    } else {
      // device noise: one uniform per token, inverse CDF (same distribution, K times fewer draws)
      const U4 r = philox4x32_10((uint64_t)n, offset, seed);
      besti = group_pick<L::G, L::V>(pv, g, u01d_open(r.x, r.y), x0, [](int gg, int vv) { return L::cls(gg, vv); });
    }
    if (g == 0) x_t[n] = besti;
  }
}

// ------------------------------------------------------------------ shared maths
// 1/x in float64 from the hardware estimate and two Newton steps (error of the
// order of 1e-16 relative, not correctly rounded; a third of the cost of a division).
__device__ __forceinline__ double rcp64(double x) {
  double r = __builtin_amdgcn_rcp(x);
  r = fma(fma(-x, r, 1.0), r, r);
  r = fma(fma(-x, r, 1.0), r, r);
  return r;
}

// Natural logarithm of a positive normal float64 (the argument-reduction and
// polynomial of the classic fdlibm e_log.c algorithm, error below 1 ulp with an
// exact division; here the division is rcp64, which adds about 1e-16 relative).
// A third of the instructions of the library routine, which also handles zero,
// negative, infinite and subnormal arguments -- those are sent to it.
__device__ __forceinline__ double log64(double x) {
  if (!(x >= 2.2250738585072014e-308 && x < 1.7976931348623157e308)) return log(x);
  int e;
  double m = frexp(x, &e);                       // [0.5, 1)
  if (m < 0.70710678118654752440) { m *= 2.0; e -= 1; }
  const double f = m - 1.0, dk = (double)e;
  const double sq = f * rcp64(2.0 + f), z = sq * sq, w = z * z;
  const double t1 = w * fma(w, fma(w, 1.531383769920937332e-01, 2.222219843214978396e-01), 3.999999999940941908e-01);
  const double t2 = z * fma(w, fma(w, fma(w, 1.479819860511658591e-01, 1.818357216161805012e-01),
                                   2.857142874366239149e-01), 6.666666666666735130e-01);
  const double R = t2 + t1, hfsq = 0.5 * f * f;
  return dk * 6.93147180369123816490e-01 - ((hfsq - (sq * (hfsq + R) + dk * 1.90821492927058770002e-10)) - f);
}

// float32 softmax over the first K logits, promoted to float64.
template <class L>
__device__ __forceinline__ void softmax_f32_to_f64(const float (&z)[L::V], int g, int K, double (&p)[L::V]) {
  float m = -INFINITY;
#pragma unroll
  for (int v = 0; v < L::V; ++v) m = fmaxf(m, (L::cls(g, v) < K) ? z[v] : -INFINITY);
  m = group_max<L::G>(m);
  float e[L::V], s = 0;
#pragma unroll
  for (int v = 0; v < L::V; ++v) {
    e[v] = (L::cls(g, v) < K) ? expf(z[v] - m) : 0.0f;
    s += e[v];
  }
  s = group_sum<L::G>(s);
#pragma unroll
  for (int v = 0; v < L::V; ++v) p[v] = (double)(e[v] / s);
}

// r_j = A_j * s_j (unnormalised p-squared marginal) and its sum over j.
template <class L>
__device__ __forceinline__ void marginal(const double (&p)[L::V], const double (&md)[L::V], const double (&mo)[L::V],
                                         double a_diag, double a_off, int xt, int g, int K, double (&A)[L::V],
                                         double (&r)[L::V], double& rsum) {
  double s2 = 0;
#pragma unroll
  for (int v = 0; v < L::V; ++v) s2 += p[v] * p[v];
  s2 = group_sum<L::G>(s2);
  double rs = 0;
#pragma unroll
  for (int v = 0; v < L::V; ++v) {
    const int c = L::cls(g, v);
    const double pp = p[v] * p[v];
    A[v] = (c == xt) ? a_diag : a_off;
    r[v] = (c < K) ? A[v] * (mo[v] * (s2 - pp) + md[v] * pp) : 0.0;
    rs += r[v];
  }
  rsum = group_sum<L::G>(rs);
}

// ------------------------------------------------------------------ LVB loss
template <class L, bool GRAD>
__global__ __launch_bounds__(PBLOCK) void k_lvb_loss(ProtArgs a, const float* __restrict__ pred, long ld,
                                                     const double* __restrict__ q, int tmax, int msa, double grad_scale,
                                                     double log_prior, double* __restrict__ part,
                                                     float* __restrict__ dpred) {
  __shared__ double red[PBLOCK / 64];
  const int K = a.tb.K;
  const int b = blockIdx.x / a.chunks, chunk = blockIdx.x % a.chunks;
  const int g = threadIdx.x % L::G, slot = threadIdx.x / L::G;
  const int t = (int)a.t[b];
  const int branch = (t == 1) ? 0 : (t == tmax ? 1 : 2);
  double md[L::V], mo[L::V], lmd[L::V], lmo[L::V];
  if (branch == 2) {
    load_digest_d<L>(a.tb.qbar_digest, t - 1, 0, K, g, md);
    load_digest_d<L>(a.tb.qbar_digest, t - 1, 1, K, g, mo);
    load_digest_d<L>(a.tb.qbar_digest, t - 1, 2, K, g, lmd);
    load_digest_d<L>(a.tb.qbar_digest, t - 1, 3, K, g, lmo);
  }
  const double cscale = grad_scale / (double)a.npos;
  double acc = 0;
  const long p1 = min(a.npos, (chunk + 1) * a.chunk_pos);
  for (long pos = chunk * a.chunk_pos + slot; pos < p1; pos += PBLOCK / L::G) {
    const long n = (long)b * a.npos + pos;
    const int x0 = (int)a.x0[n];
    float z[L::V];
    float gz[L::V];
#pragma unroll
    for (int v = 0; v < L::V; ++v) gz[v] = 0.0f;
    double tok_loss = 0;
    if (branch == 0) {
      // CrossEntropyLoss over all ld logits (losses.py:54), float32
      load_slots<L>(pred + n * ld, g, (int)ld, -INFINITY, z);
      float m = -INFINITY;
#pragma unroll
      for (int v = 0; v < L::V; ++v) m = fmaxf(m, z[v]);
      m = group_max<L::G>(m);
      float e[L::V], s = 0, zx0 = 0;
#pragma unroll
      for (int v = 0; v < L::V; ++v) {
        e[v] = expf(z[v] - m); s += e[v];
        zx0 += (L::cls(g, v) == x0) ? z[v] : 0.0f;
      }
      s = group_sum<L::G>(s);
      zx0 = group_sum<L::G>(zx0);
      tok_loss = (double)((m + logf(s)) - zx0);
      if constexpr (GRAD) {
#pragma unroll
        for (int v = 0; v < L::V; ++v)
          gz[v] = (float)(cscale * (double)(e[v] / s - ((L::cls(g, v) == x0) ? 1.0f : 0.0f)));
      }
    } else if (branch == 1) {
      // KL(q || uniform) on the collater's forward probabilities (losses.py:58-60)
      double qq[L::V];
      load_slots<L>(q + n * K, g, K, 0.0, qq);
      double s = 0;
#pragma unroll
      for (int v = 0; v < L::V; ++v) s += (qq[v] > 0.0 ? qq[v] * log(qq[v]) : 0.0) - qq[v] * log_prior;
      // 'batchmean' divides by the leading size of q[i]: L for a sequence (what the
      // finalize step does), but only the depth D for an MSA laid out [D,L,K]
      tok_loss = group_sum<L::G>(s) * (msa > 0 ? (double)a.npos / (double)msa : 1.0);
    } else {
      // KL(q(x_{t-1} | x_t, x_0) || p_theta(x_{t-1} | x_t)) (losses.py:62-86), with
      //   post_j = A_j B_j / den,  P_j = A_j s_j / rsum,  s_j = off_j (S2 - p_j^2) + diag_j p_j^2
      // (A: column x_t of Q_t, B: row x0 of Qbar_{t-1}; both take their diagonal or off-diagonal value).  A_j
      // cancels inside the logarithm and log B_j, log den come from the digests, so a class costs one float64
      // log (of s_j) and, for the gradient, one reciprocal:
      //   term_j = post_j (log B_j - log den + log rsum - log s_j)
      const int xt = (int)a.xt[n];
      load_slots<L>(pred + n * ld, g, (int)ld, -INFINITY, z);
      double p[L::V];
      softmax_f32_to_f64<L>(z, g, K, p);
      const double a_diag = dgd(a.tb.qone_digest, t, 0, K)[xt], a_off = dgd(a.tb.qone_digest, t, 1, K)[xt];
      // denominator Qbar_t[x0, x_t] (MSA variant: Qbar_{t-1}[x0, x_t], losses.py:148)
      const int td = msa ? t - 1 : t;
      const int dsel = (x0 == xt) ? 0 : 1;
      const double den = dgd(a.tb.qbar_digest, td, dsel, K)[xt], lden = dgd(a.tb.qbar_digest, td, 2 + dsel, K)[xt];
      const double iden = rcp64(den);
      double s2 = 0;
#pragma unroll
      for (int v = 0; v < L::V; ++v) s2 += p[v] * p[v];
      s2 = group_sum<L::G>(s2);
      double A[L::V], sj[L::V], post[L::V], rs = 0, psum = 0;
#pragma unroll
      for (int v = 0; v < L::V; ++v) {
        const int c = L::cls(g, v);
        const bool ok = c < K;
        const double pp = p[v] * p[v];
        A[v] = (c == xt) ? a_diag : a_off;
        sj[v] = ok ? mo[v] * (s2 - pp) + md[v] * pp : 1.0;
        post[v] = ok ? A[v] * ((c == x0) ? md[v] : mo[v]) * iden : 0.0;
        rs += ok ? A[v] * sj[v] : 0.0;
        psum += post[v];
      }
      const double rsum = group_sum<L::G>(rs);
      const double lshift = log64(rsum) - lden;
      double s = 0;
#pragma unroll
      for (int v = 0; v < L::V; ++v) {
        const double lB = (L::cls(g, v) == x0) ? lmd[v] : lmo[v];
        s += (post[v] > 0.0) ? post[v] * ((lB + lshift) - log64(sj[v])) : 0.0;
      }
      tok_loss = group_sum<L::G>(s);
      if constexpr (GRAD) {
        psum = group_sum<L::G>(psum);
        const double base = psum * rcp64(rsum);
        double gs[L::V], W = 0;
#pragma unroll
        for (int v = 0; v < L::V; ++v) {
          const int c = L::cls(g, v);
          // d term / d r_j times A_j:  (-post_j / r_j + psum / rsum) A_j, post_j / r_j = B_j / (den s_j)
          const double Bc = (c == x0) ? md[v] : mo[v];
          gs[v] = (c < K) ? (base - Bc * iden * rcp64(sj[v])) * cscale * A[v] : 0.0;
          W += gs[v] * mo[v];
        }
        W = group_sum<L::G>(W);
        double gp[L::V], dot = 0;
#pragma unroll
        for (int v = 0; v < L::V; ++v) {
          gp[v] = 2.0 * p[v] * (W + gs[v] * (md[v] - mo[v]));
          dot += p[v] * gp[v];
        }
        dot = group_sum<L::G>(dot);
#pragma unroll
        for (int v = 0; v < L::V; ++v) gz[v] = (L::cls(g, v) < K) ? (float)(p[v] * (gp[v] - dot)) : 0.0f;
      }
    }
    if constexpr (GRAD) store_slots<L>(dpred + n * ld, g, (int)ld, gz);
    if (g == 0) acc += tok_loss;
  }
  const double s = block_sum(acc, red);
  if (threadIdx.x == 0) part[(size_t)b * a.chunks + chunk] = s;
}

__global__ void k_lvb_finalize(const double* __restrict__ part, int B, int chunks, long npos,
                               double* __restrict__ per_seq) {
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= B) return;
  double s = 0;
  for (int c = 0; c < chunks; ++c) s += part[(size_t)b * chunks + c];
  per_seq[b] = s / (double)npos;
}

// ------------------------------------------------------------------ reverse step
template <class L>
__global__ __launch_bounds__(PBLOCK) void k_lvb_reverse(ProtArgs a, const float* __restrict__ pred, long ld,
                                                        const double* __restrict__ expo, uint64_t seed, uint64_t offset,
                                                        double last_class_div, long exclude_last_prefix,
                                                        int64_t* __restrict__ out) {
  const int K = a.tb.K;
  const int b = blockIdx.x / a.chunks, chunk = blockIdx.x % a.chunks;
  const int g = threadIdx.x % L::G, slot = threadIdx.x / L::G;
  const int t = (int)a.t[b];
  double md[L::V], mo[L::V];
  load_digest_d<L>(a.tb.qbar_digest, t - 1, 0, K, g, md);
  load_digest_d<L>(a.tb.qbar_digest, t - 1, 1, K, g, mo);
  const long p1 = min(a.npos, (chunk + 1) * a.chunk_pos);
  for (long pos = chunk * a.chunk_pos + slot; pos < p1; pos += PBLOCK / L::G) {
    const long n = (long)b * a.npos + pos;
    const int xt = (int)a.xt[n];
    float z[L::V];
    load_slots<L>(pred + n * ld, g, (int)ld, -INFINITY, z);
    double p[L::V], A[L::V], r[L::V], rsum, e[L::V];
    softmax_f32_to_f64<L>(z, g, K, p);
    const double a_diag = dgd(a.tb.qone_digest, t, 0, K)[xt], a_off = dgd(a.tb.qone_digest, t, 1, K)[xt];
    marginal<L>(p, md, mo, a_diag, a_off, xt, g, K, A, r, rsum);
    // MSA sampling (generate.py:84-88): the last class (the gap) is divided by 1 + penalty after the
    // normalisation, and cannot be drawn at all in the first `exclude_last_prefix` positions (the query row)
    const bool no_last = pos < exclude_last_prefix;
    int besti;
    if (expo) {
      // the reference's draw: torch.multinomial on CPU = argmax_j(P_j / E_j), E ~ Exp(1) passed in
      load_slots<L>(expo + n * K, g, K, 1.0, e);
      double best = -INFINITY;
      besti = 0x7fffffff;
#pragma unroll
      for (int v = 0; v < L::V; ++v) {
        const int c = L::cls(g, v);
        double P = r[v] / rsum;
        if (c == K - 1 && last_class_div != 1.0) P /= last_class_div;
        const double val = P / e[v];
        const bool ok = c < K && !(no_last && c == K - 1);
        if (ok && (val > best || besti == 0x7fffffff)) { best = val; besti = c; }
      }
      group_argmax<L::G>(best, besti);
    } else {
      // device noise: one uniform per token, inverse CDF over the unnormalised marginal
#pragma unroll
      for (int v = 0; v < L::V; ++v)
        if (L::cls(g, v) == K - 1) r[v] = no_last ? 0.0 : r[v] / last_class_div;
      const U4 rn = philox4x32_10((uint64_t)n, offset, seed);
      besti = group_pick<L::G, L::V>(r, g, u01d_open(rn.x, rn.y), xt, [](int gg, int vv) { return L::cls(gg, vv); });
    }
    if (g == 0) out[n] = besti;
  }
}

// The same step for on-device noise.  Nothing has to match a recorded draw, so the
// marginal is evaluated in float32 (weights accurate to 1e-6 relative), the softmax
// uses the hardware exp2, and the uniforms come from a per-wave pool: one Philox
// call per lane yields 256 words per wave, parked in LDS and handed out one word
// per token (a 32-bit uniform) over the next 256 / tokens-per-pass passes.
// PAIR: rows are read as 8-byte pairs (even row stride, 8-byte aligned base): lane g owns the class pairs g, g + G,
// ... -- half the load instructions and address arithmetic of the one-class-per-load strided layout, and no
// per-slot bounds branches (a pair past the row end re-reads the row's last pair and is masked afterwards).
// BLK: threads per workgroup (128, 192 or 256), chosen by the host so that the last, partly filled pass over a
// chunk wastes the fewest lanes (263 positions: five passes of 64 tokens idle 18 % of the slots, six of 48 idle 9 %).
template <class L, bool PAIR, int BLK>
__global__ __launch_bounds__(BLK) void k_lvb_reverse_device_noise(ProtArgs a, const float* __restrict__ pred, long ld,
                                                                     uint64_t seed, uint64_t offset,
                                                                     const uint64_t* __restrict__ offset_dev,
                                                                     float last_class_div, long exclude_last_prefix,
                                                                     int64_t* __restrict__ out) {
  if (offset_dev) offset += *offset_dev;
  constexpr int WAVES = BLK / 64, TPW = 64 / L::G;      // tokens per wave and pass
  constexpr int REFILL = 256 / TPW;                         // passes served by one pool
  constexpr int PF = 3;                                     // passes whose rows are in flight
  __shared__ uint32_t pool[WAVES][256];
  __shared__ float s_ad[256], s_ao[256];                    // column x_t of Q_t: diagonal / off-diagonal entry by x_t
  const int K = a.tb.K;
  static_assert(!PAIR || L::V % 2 == 0, "pairs need an even number of slots per lane");
  auto cls = [](int gg, int vv) { return PAIR ? 2 * (gg + L::G * (vv >> 1)) + (vv & 1) : L::cls(gg, vv); };
  const int b = blockIdx.x / a.chunks, chunk = blockIdx.x % a.chunks;
  const int g = threadIdx.x % L::G, slot = threadIdx.x / L::G;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, tok_in_wave = lane / L::G;
  const long p1 = min(a.npos, (chunk + 1) * a.chunk_pos);
  const long pbeg = chunk * a.chunk_pos + slot;
  constexpr long STEP = BLK / L::G;
  // rows and tokens of the first passes are requested before the timestep -> digest chain
  float zq[PF][L::V];
  int xq[PF];
  // this batch element's rows and tokens through workgroup-uniform bases and 32-bit byte offsets (the host checks
  // that one batch element spans less than 2^31 bytes): no 64-bit address arithmetic per lane
  const char* pred_b = reinterpret_cast<const char*>(pred + (long)b * a.npos * ld);
  const char* xt_b = reinterpret_cast<const char*>(a.xt + (long)b * a.npos);
  const unsigned ld_u = (unsigned)ld, last_pair = (unsigned)(ld >> 1) - 1u;
  auto fetch = [&](int u, long pos) {
    const unsigned pc = (unsigned)(pos < p1 ? pos : pbeg < p1 ? pbeg : 0);
    if constexpr (PAIR) {
#pragma unroll
      for (int j = 0; j < L::V / 2; ++j) {
        const unsigned off = (pc * ld_u + 2u * min((unsigned)(g + L::G * j), last_pair)) * 4u;
        const float2 q = *reinterpret_cast<const float2*>(pred_b + off);
        zq[u][2 * j] = q.x; zq[u][2 * j + 1] = q.y;
      }
    } else {
      load_slots<L>(reinterpret_cast<const float*>(pred_b) + (size_t)pc * ld_u, g, (int)ld, -INFINITY, zq[u]);
    }
    xq[u] = (int)*reinterpret_cast<const int64_t*>(xt_b + pc * 8u);
  };
  if (pbeg < p1 || chunk * a.chunk_pos < p1) {
#pragma unroll
    for (int u = 0; u < PF; ++u) fetch(u, pbeg + u * STEP);
  }
  const int t = (int)a.t[b];
  // Per-slot constants of the timestep.  Slots past K get zero weights and their logits are replaced by -inf on
  // arrival, so the loop carries no bounds checks; the divisor of the last class is folded into its two weights
  // (the draw is linear in them).
  const float inv_last = 1.0f / last_class_div;
  float md[L::V], mo[L::V];
  bool valid[L::V];
#pragma unroll
  for (int v = 0; v < L::V; ++v) {
    const int c = cls(g, v);
    valid[v] = c < K;
    const float w = (c == K - 1) ? inv_last : 1.0f;
    md[v] = valid[v] ? w * (float)dgd(a.tb.qbar_digest, t - 1, 0, K)[c] : 0.0f;
    mo[v] = valid[v] ? w * (float)dgd(a.tb.qbar_digest, t - 1, 1, K)[c] : 0.0f;
  }
  for (int c = threadIdx.x; c < K; c += BLK) {
    s_ad[c] = (float)dgd(a.tb.qone_digest, t, 0, K)[c];
    s_ao[c] = (float)dgd(a.tb.qone_digest, t, 1, K)[c];
  }
  __syncthreads();
  int it = 0;
  for (long pos0 = chunk * a.chunk_pos; pos0 < p1; pos0 += PF * STEP) {
#pragma unroll
    for (int u = 0; u < PF; ++u, ++it) {
      if (pos0 + u * STEP >= p1) break;                      // uniform over the workgroup
      if (it % REFILL == 0) {        // wave-private pool: LDS operations of a wave are in order
        const U4 r = philox4x32_10(((uint64_t)blockIdx.x << 32) + (uint64_t)(it / REFILL) * BLK + threadIdx.x, offset,
                                   seed);
        pool[wave][lane] = r.x; pool[wave][64 + lane] = r.y; pool[wave][128 + lane] = r.z; pool[wave][192 + lane] = r.w;
      }
      const long pos = pos0 + u * STEP + slot;
      float z[L::V];
#pragma unroll
      for (int v = 0; v < L::V; ++v) z[v] = valid[v] ? zq[u][v] : -INFINITY;
      const int xt = xq[u];
      if (pos + PF * STEP - slot < p1) fetch(u, pos + PF * STEP);   // refill this stage (uniform condition)
      const uint32_t word = pool[wave][(it % REFILL) * TPW + tok_in_wave];
      float m = -INFINITY;
#pragma unroll
      for (int v = 0; v < L::V; ++v) m = fmaxf(m, z[v]);
      m = group_max<L::G>(m);
      float p[L::V], s2 = 0;
#pragma unroll
      for (int v = 0; v < L::V; ++v) {
        p[v] = __builtin_amdgcn_exp2f((z[v] - m) * 1.44269504088896340736f);      // 0 on slots past K
        s2 = fmaf(p[v], p[v], s2);
      }
      s2 = group_sum<L::G>(s2);       // the softmax normaliser cancels in the draw: p stays unnormalised
      const float a_diag = s_ad[xt], a_off = s_ao[xt];
      float r[L::V];
#pragma unroll
      for (int v = 0; v < L::V; ++v) {
        const float pp = p[v] * p[v];
        r[v] = ((cls(g, v) == xt) ? a_diag : a_off) * fmaf(mo[v], s2 - pp, md[v] * pp);
      }
      if (exclude_last_prefix > 0) {      // MSA sampling only: the last class cannot be drawn in the query row
#pragma unroll
        for (int v = 0; v < L::V; ++v)
          if (cls(g, v) == K - 1 && pos < exclude_last_prefix) r[v] = 0.0f;
      }
      const float u01v = ((float)(word >> 8) + 0.5f) * 5.9604644775390625e-08f;      // (0, 1), 24 bits
      const int pick = group_pick<L::G, L::V>(r, g, u01v, xt, cls);
      if (g == 0 && pos < p1) out[(long)b * a.npos + pos] = pick;
    }
  }
}

static int check_prot(const d3pm_tables_f64* tb, int B, long npos) {
  D3PM_CHECK_ARG(tb && tb->qbar_digest && tb->qone_digest, "tables descriptor or digest is null");
  D3PM_CHECK_ARG(tb->K >= 2 && tb->K <= 256 && tb->T >= 1, "need 2 <= K <= 256, T >= 1 (K=%d, T=%d)", tb->K, tb->T);
  D3PM_CHECK_ARG(B >= 0 && npos >= 0, "negative batch or position count");
  return D3PM_OK;
}

static ProtArgs make_prot(const d3pm_tables_f64* tb, const int64_t* x0, const int64_t* xt, const int64_t* t, int B,
                          long npos, int G) {
  ProtArgs a;
  a.tb = *tb; a.x0 = x0; a.xt = xt; a.t = t; a.B = B; a.npos = npos;
  const TokenGrid tg = token_grid(B, npos, PBLOCK / G);
  a.chunks = tg.chunks; a.chunk_pos = tg.chunk_pos;
  return a;
}

}  // namespace d3pm

using namespace d3pm;

extern "C" int d3pm_corrupt(const d3pm_tables_f64* tb, const int64_t* x0, const int64_t* t, int B, long npos,
                            const double* expo, uint64_t seed, uint64_t offset, int64_t* x_t, double* probs,
                            void* stream) {
  if (int rc = check_prot(tb, B, npos)) return rc;
  if (B == 0 || npos == 0) return D3PM_OK;
  D3PM_CHECK_ARG(x0 && t && x_t, "null pointer");
  return dispatch_layout(tb->K, false, [&](auto lay) {
    using L = decltype(lay);
    ProtArgs a = make_prot(tb, x0, nullptr, t, B, npos, L::G);
    hipLaunchKernelGGL(k_corrupt<L>, dim3(B * a.chunks), dim3(PBLOCK), 0, (hipStream_t)stream, a, expo, seed, offset,
                       x_t, probs);
    D3PM_LAUNCH_CHECK("k_corrupt");
    return D3PM_OK;
  });
}

extern "C" size_t d3pm_lvb_loss_workspace(int B, long npos, int K) {
  (void)K;
  long passes = npos > 0 ? npos : 1;
  long chunks = passes < 4096 ? passes : 4096;
  return (size_t)(B > 0 ? B : 1) * (size_t)chunks * sizeof(double);
}

extern "C" int d3pm_lvb_loss(const d3pm_tables_f64* tb, const float* pred, long ld, const int64_t* x0,
                             const int64_t* x_t, const int64_t* t, const double* q, int B, long npos, int tmax, int msa,
                             double grad_scale, double* per_seq, float* dpred, void* workspace, size_t workspace_bytes,
                             void* stream) {
  if (int rc = check_prot(tb, B, npos)) return rc;
  D3PM_CHECK_ARG(ld >= tb->K && ld <= 256, "row stride (n_tokens) must be in [K, 256], got %ld", ld);
  if (B == 0) return D3PM_OK;
  D3PM_CHECK_ARG(npos > 0, "empty sequence: mean over zero positions");
  D3PM_CHECK_ARG(pred && x0 && x_t && t && per_seq, "null pointer");
  hipStream_t s = (hipStream_t)stream;
  // sample_prior builds a float32 1/K and takes its float32 log (losses.py:11-12,60)
  const double log_prior = (double)logf(1.0f / (float)tb->K);
  auto run = [&](auto lay) {
    using L = decltype(lay);
    ProtArgs a = make_prot(tb, x0, x_t, t, B, npos, L::G);
    const size_t need = (size_t)B * a.chunks * sizeof(double);
    D3PM_CHECK_ARG(workspace && workspace_bytes >= need, "workspace too small: %zu < %zu bytes", workspace_bytes, need);
    double* part = (double*)workspace;
    const dim3 grid(B * a.chunks), block(PBLOCK);
    if (dpred)
      hipLaunchKernelGGL((k_lvb_loss<L, true>), grid, block, 0, s, a, pred, ld, q, tmax, msa, grad_scale, log_prior,
                         part, dpred);
    else
      hipLaunchKernelGGL((k_lvb_loss<L, false>), grid, block, 0, s, a, pred, ld, q, tmax, msa, grad_scale, log_prior,
                         part, dpred);
    D3PM_LAUNCH_CHECK("k_lvb_loss");
    hipLaunchKernelGGL(k_lvb_finalize, dim3((B + 255) / 256), dim3(256), 0, s, part, B, a.chunks, npos, per_seq);
    D3PM_LAUNCH_CHECK("k_lvb_finalize");
    return D3PM_OK;
  };
  return dispatch_layout((int)ld, false, run);       // (four lanes per token measured no faster than eight)
}

extern "C" int d3pm_lvb_reverse_step(const d3pm_tables_f64* tb, const float* pred, long ld, const int64_t* x_t,
                                     const int64_t* t, int B, long npos, const double* expo, uint64_t seed,
                                     uint64_t offset, const uint64_t* offset_dev, double last_class_div,
                                     long exclude_last_prefix, int64_t* out, void* stream) {
  if (int rc = check_prot(tb, B, npos)) return rc;
  D3PM_CHECK_ARG(ld >= tb->K, "row stride %ld smaller than K=%d", ld, tb->K);
  D3PM_CHECK_ARG(last_class_div > 0.0 && exclude_last_prefix >= 0, "last-class divisor must be positive, prefix >= 0");
  if (B == 0 || npos == 0) return D3PM_OK;
  D3PM_CHECK_ARG(pred && x_t && t && out, "null pointer");
  D3PM_CHECK_ARG(npos * ld < 0x1fffffffL, "one batch element spans %ld predictions: more than 2^29", npos * ld);
  if (!expo && ld <= 32) {
    // few lanes per token: the draw is cheap per class, the cross-lane steps are what costs
    static const int rev_blocks = getenv("D3PM_REV_BLOCKS") ? atoi(getenv("D3PM_REV_BLOCKS")) : 1024;
    static const int rev_lanes = getenv("D3PM_REV_LANES") ? atoi(getenv("D3PM_REV_LANES")) : 4;
    auto launch_blk = [&](auto lay, auto blk) {
      using L = decltype(lay);
      constexpr int BLK = decltype(blk)::value;
      ProtArgs a = make_prot(tb, nullptr, x_t, t, B, npos, L::G);
      const TokenGrid tg = token_grid(B, npos, BLK / L::G, rev_blocks);   // long trips: rows are prefetched 3 passes ahead
      a.chunks = tg.chunks; a.chunk_pos = tg.chunk_pos;
      if constexpr (L::V % 2 == 0) {
        if (ld % 2 == 0 && ((uintptr_t)pred & 7) == 0) {
          hipLaunchKernelGGL((k_lvb_reverse_device_noise<L, true, BLK>), dim3(B * a.chunks), dim3(BLK), 0, (hipStream_t)stream,
                             a, pred, ld, seed, offset, offset_dev, (float)last_class_div, exclude_last_prefix, out);
          D3PM_LAUNCH_CHECK("k_lvb_reverse_device_noise");
          return D3PM_OK;
        }
      }
      hipLaunchKernelGGL((k_lvb_reverse_device_noise<L, false, BLK>), dim3(B * a.chunks), dim3(BLK), 0, (hipStream_t)stream, a,
                         pred, ld, seed, offset, offset_dev, (float)last_class_div, exclude_last_prefix, out);
      D3PM_LAUNCH_CHECK("k_lvb_reverse_device_noise");
      return D3PM_OK;
    };
    auto launch = [&](auto lay) {
      using L = decltype(lay);
      // workgroup size with the fewest idle token slots over a chunk (ties: the larger workgroup)
      static const int force = getenv("D3PM_REV_BLOCK") ? atoi(getenv("D3PM_REV_BLOCK")) : 0;
      int best = 256;
      long best_slots = -1;
      for (int blk : {256, 192, 128}) {
        const TokenGrid tg = token_grid(B, npos, blk / L::G, rev_blocks);
        const long tpp = blk / L::G;
        long slots = 0;
        for (long c0 = 0; c0 < npos; c0 += tg.chunk_pos) {
          const long n = (npos - c0 < tg.chunk_pos) ? npos - c0 : tg.chunk_pos;
          slots += (n + tpp - 1) / tpp * tpp;
        }
        if (best_slots < 0 || slots < best_slots) { best_slots = slots; best = blk; }
      }
      if (force) best = force;
      if (best == 192) return launch_blk(lay, std::integral_constant<int, 192>{});
      if (best == 128) return launch_blk(lay, std::integral_constant<int, 128>{});
      return launch_blk(lay, std::integral_constant<int, 256>{});
    };
    if (rev_lanes == 2) return ld <= 24 ? launch(Layout<2, 12, false>{}) : launch(Layout<2, 16, false>{});
    if (rev_lanes == 4) return ld <= 24 ? launch(Layout<4, 6, false>{}) : launch(Layout<4, 8, false>{});
    return ld <= 24 ? launch(Layout<8, 3, false>{}) : launch(Layout<8, 4, false>{});
  }
  return dispatch_layout(tb->K, false, [&](auto lay) {
    using L = decltype(lay);
    ProtArgs a = make_prot(tb, nullptr, x_t, t, B, npos, L::G);
    if (!expo) {
      const TokenGrid tg = token_grid(B, npos, PBLOCK / L::G, 1024);
      a.chunks = tg.chunks; a.chunk_pos = tg.chunk_pos;
      hipLaunchKernelGGL((k_lvb_reverse_device_noise<L, false, PBLOCK>), dim3(B * a.chunks), dim3(PBLOCK), 0,
                         (hipStream_t)stream, a, pred, ld, seed, offset, offset_dev, (float)last_class_div,
                         exclude_last_prefix, out);
      D3PM_LAUNCH_CHECK("k_lvb_reverse_device_noise");
      return D3PM_OK;
    }
    hipLaunchKernelGGL(k_lvb_reverse<L>, dim3(B * a.chunks), dim3(PBLOCK), 0, (hipStream_t)stream, a, pred, ld, expo,
                       seed, offset, last_class_div, exclude_last_prefix, out);
    D3PM_LAUNCH_CHECK("k_lvb_reverse");
    return D3PM_OK;
  });
}
