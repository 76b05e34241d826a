// Protein-style (EvoDiff-like) D3PM kernels: batched forward corruption, the LVB
// loss with its gradient, and the reverse step.  Arithmetic after the float32
// softmax is float64, as in protein_generation/losses.py:65 and
// generate.py:30.  Tables are column-structured digests (uniform schedule):
// entry(i,j) = i==j ? diag[j] : off[j], which turns the reference's [L,K,K]
// temporaries into O(K) work per token:
//   s_j = sum_i p_i^2 Qm[i,j] = off_j * (S2 - p_j^2) + diag_j * p_j^2.
#include "layout.h"
#include "rowimage.h"

#include <cstdlib>

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
// The loss and the reverse step read Qbar_{t-1} and Q_t: their timesteps lie in [1, T - 1] (the loss also takes
// t == tmax, which reads no table).  Anything else is brought to the nearest valid step, so nothing is read outside
// the digests; tokens are clamped to [0, K - 1] in the same way.
__device__ __forceinline__ int lvb_timestep(int t, int T) { return max(1, min(t, T - 1)); }

template <class L>
__global__ __launch_bounds__(PBLOCK) void k_corrupt(ProtArgs a, const double* __restrict__ expo, uint64_t seed,
                                                    uint64_t offset, int64_t* __restrict__ x_t,
                                                    double* __restrict__ probs) {
  const int K = a.tb.K;
  const int b = blockIdx.x / a.chunks, chunk = blockIdx.x % a.chunks;
  const int g = threadIdx.x % L::G, slot = threadIdx.x / L::G;
  const int t = clamp_index((int)a.t[b], a.tb.T);
  double d[L::V], o[L::V];
  load_digest_d<L>(a.tb.qbar_digest, t, 0, K, g, d);
  load_digest_d<L>(a.tb.qbar_digest, t, 1, K, g, o);
  const long p1 = min(a.npos, (chunk + 1) * a.chunk_pos);
  for (long pos = chunk * a.chunk_pos + slot; pos < p1; pos += PBLOCK / L::G) {
    const long n = (long)b * a.npos + pos;
    const int x0 = clamp_index((int)a.x0[n], K);
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

// Forward corruption with noise drawn on the device and no probabilities asked for: one token per lane, float32.
// Row x0 of a column-structured Qbar_t is off_j everywhere except diag_x0 at j = x0, so the draw is two-staged:
//   x_t == x0 with probability diag_x0 / (sum_j off_j - off_x0 + diag_x0), decided on a 32-bit integer threshold that
//   the workgroup's first wave forms in float64 (a step that moves one token in 10^5 keeps five digits of that
//   probability);
//   otherwise class j != x0 in proportion to off_j: inverse CDF over the prefix sums of off with x0's share taken
//   out, one float32 uniform.
// Both words come from one Philox2x32-10 call per token, keyed per launch as in k_lvb_reverse_flat.  Nothing here is
// matched bit for bit by a host computation (the recorded-draw variant above is).
__global__ __launch_bounds__(256) void k_corrupt_lane(ProtArgs a, uint64_t seed, uint64_t offset,
                                                      int64_t* __restrict__ x_t) {
  __shared__ float s_off[64], s_pre[64];
  __shared__ uint32_t s_stay[64];
  const int K = a.tb.K;
  const int b = blockIdx.x / a.chunks, chunk = blockIdx.x % a.chunks;
  const int t = clamp_index((int)a.t[b], a.tb.T);
  if (threadIdx.x < 64) {
    const int lane = threadIdx.x;
    const double o = lane < K ? dgd(a.tb.qbar_digest, t, 1, K)[lane] : 0.0;
    const double d = lane < K ? dgd(a.tb.qbar_digest, t, 0, K)[lane] : 0.0;
    const double tot = wave_sum(o);
    const double stay = d / (tot - o + d);                   // (lanes past K: 0 / 0, never read)
    s_stay[lane] = stay >= 1.0 ? 0xffffffffu : (uint32_t)(stay * 4294967296.0);
    s_off[lane] = (float)o;
    s_pre[lane] = group_scan_inclusive<64>((float)o, lane);
  }
  __syncthreads();
  const float tot_off = s_pre[K - 1];
  const U4 pad = philox4x32_10(offset, 0x13198A2E03707344ull, seed);
  const long p1 = min(a.npos, (chunk + 1) * a.chunk_pos);
  for (long pos = chunk * a.chunk_pos + threadIdx.x; pos < p1; pos += blockDim.x) {
    const long n = (long)b * a.npos + pos;
    const int x0 = clamp_index((int)a.x0[n], K);
    const U2 r = philox2x32_10((uint32_t)n ^ pad.x, (uint32_t)((uint64_t)n >> 32) ^ pad.y, pad.z);
    const float off_x0 = s_off[x0];
    const float target = u01_open(r.x) * (tot_off - off_x0);
    const float target_hi = target + off_x0;                 // for the classes from x0 on
    int cnt = 0;
    for (int j = 0; j < K - 1; ++j) cnt += (s_pre[j] <= (j < x0 ? target : target_hi)) ? 1 : 0;
    x_t[n] = r.y < s_stay[x0] ? x0 : cnt;
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
// `hyb`: the driver's cross-entropy term as well (Protein Generation Notebook cell-12:
// loss = (lvb + lambda * CrossEntropyLoss(outputs, tgt)) * n_tokens): per token -log_softmax(z)[x0] over ALL ld
// logits, whatever the element's timestep, summed into part_ce; its gradient, weighted ce_scale / npos per token,
// is added to the LVB gradient in the same store.
template <class L, bool GRAD>
__global__ __launch_bounds__(PBLOCK) void k_lvb_loss(ProtArgs a, const float* __restrict__ pred, long ld,
                                                     const double* __restrict__ q, int tmax, int msa, double grad_scale,
                                                     double log_prior, double* __restrict__ part,
                                                     float* __restrict__ dpred, int hyb, double ce_scale,
                                                     double* __restrict__ part_ce) {
  __shared__ double red[PBLOCK / 64];
  const int K = a.tb.K;
  const int b = blockIdx.x / a.chunks, chunk = blockIdx.x % a.chunks;
  const int g = threadIdx.x % L::G, slot = threadIdx.x / L::G;
  const int t0 = (int)a.t[b];
  const int t = (t0 == tmax) ? t0 : lvb_timestep(t0, a.tb.T);
  const int branch = (t == 1) ? 0 : (t == tmax ? 1 : 2);
  double md[L::V], mo[L::V], lmd[L::V], lmo[L::V];
  if (branch == 2) {
    load_digest_d<L>(a.tb.qbar_digest, t - 1, 0, K, g, md);
    load_digest_d<L>(a.tb.qbar_digest, t - 1, 1, K, g, mo);
    load_digest_d<L>(a.tb.qbar_digest, t - 1, 2, K, g, lmd);
    load_digest_d<L>(a.tb.qbar_digest, t - 1, 3, K, g, lmo);
  }
  const double cscale = grad_scale / (double)a.npos;
  const double ces = hyb ? ce_scale / (double)a.npos : 0.0;
  double acc = 0, acc_ce = 0;
  const long p1 = min(a.npos, (chunk + 1) * a.chunk_pos);
  for (long pos = chunk * a.chunk_pos + slot; pos < p1; pos += PBLOCK / L::G) {
    const long n = (long)b * a.npos + pos;
    const int x0 = clamp_index((int)a.x0[n], K);
    float z[L::V];
    float gz[L::V];
#pragma unroll
    for (int v = 0; v < L::V; ++v) gz[v] = 0.0f;
    double tok_loss = 0;
    if (branch == 0 || hyb) {
      // CrossEntropyLoss over all ld logits (losses.py:54 for t == 1; the driver's second loss term), float32
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
      const double ce_tok = (double)((m + logf(s)) - zx0);
      if (branch == 0) tok_loss = ce_tok;
      if (g == 0) acc_ce += ce_tok;
      if constexpr (GRAD) {
        const double w = (branch == 0 ? cscale : 0.0) + ces;
#pragma unroll
        for (int v = 0; v < L::V; ++v)
          gz[v] = (float)(w * (double)(e[v] / s - ((L::cls(g, v) == x0) ? 1.0f : 0.0f)));
      }
    }
    if (branch == 0) {
    } else if (branch == 1) {
      // KL(q || uniform) on the collater's forward probabilities (losses.py:58-60)
      // (no probabilities given: the element's loss is NaN, nothing is read)
      double qq[L::V];
      if (q) load_slots<L>(q + n * K, g, K, 0.0, qq);
      double s = q ? 0.0 : (double)NAN;
#pragma unroll
      for (int v = 0; v < L::V; ++v) s += q ? (qq[v] > 0.0 ? qq[v] * log(qq[v]) : 0.0) - qq[v] * log_prior : 0.0;
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
      const int xt = clamp_index((int)a.xt[n], K);
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
        // (gz holds the cross-entropy term's gradient, or zero)
#pragma unroll
        for (int v = 0; v < L::V; ++v) gz[v] += (L::cls(g, v) < K) ? (float)(p[v] * (gp[v] - dot)) : 0.0f;
      }
    }
    if constexpr (GRAD) store_slots<L>(dpred + n * ld, g, (int)ld, gz);
    if (g == 0) acc += tok_loss;
  }
  const double s = block_sum(acc, red);
  if (threadIdx.x == 0) part[(size_t)b * a.chunks + chunk] = s;
  if (hyb) {
    const double sc = block_sum(acc_ce, red);
    if (threadIdx.x == 0) part_ce[(size_t)b * a.chunks + chunk] = sc;
  }
}

// ------------------------------------------------------------------ LVB loss, one token per lane, float32
// The training form of the loss (a gradient is asked for): every lane takes a whole token, rows go through a
// wave-private LDS image (rowimage.h), per-class constants of the timestep are wave-uniform scalars, and the
// arithmetic after the softmax is float32 in a form that does not cancel:
//   e_j = 2^(z_j - max),  E2 = sum e_j^2,  s'_j = s_j / S2 = mo_j + (mdo / E2) e_j^2
// (S2 = sum p_i^2 divides out of every ratio; mdo = diag - off of Qbar_{t-1} is one value for all classes because
// every row of the table has the same sum, so sum_j s'_j = sum_mo + mdo needs no reduction), and with
// c = rsum' / den the Kullback-Leibler term of a token is
//   sum_j post_j log(post_j / P_j),   post_j / P_j = B_j c / s'_j
// -- every logarithm is of a ratio that is close to one wherever post_j is large.  Every class is first treated
// as a generic one (A_j = a_off, B_j = mo_j); the classes x0 (B = mo + mdo) and x_t (A = a_diag) are corrected
// with per-token scalars (their two logarithms in a 1 + d form, see below) and patched into the lane's own LDS row.
// The MSA variant divides the posterior by Qbar_{t-1}[x0, x_t] instead of Qbar_t[x0, x_t] (losses.py:148): its
// target is rho post_j with rho = Qbar_t[x0, x_t] / Qbar_{t-1}[x0, x_t], a number that depends on the timestep, on
// the class x_t and on whether x0 == x_t only, and since the post_j sum to one its token term is
//   sum_j rho post_j log(rho post_j / P_j) = rho (log rho + KL),
// the gradient rho times the plain one.  rho and log rho are formed per timestep in float64 (log rho as a
// difference of the digests' float64 logarithms: rho is close to one at late timesteps) and kept as float32 rows
// beside the other per-timestep constants.  Gradient:
//   gs_j = A_j cs / S2 (psum / rsum' - B_j / (den s'_j)) = g0 - g1 q_j (+ d0 at x0, + dt at x_t),  q_j = mo_j / s'_j
//   W = sum_j gs_j mo_j,  dot = 2 (W S2 + mdo sum_i p_i^2 gs_i),  gz_i = 2 p_i^2 (W + mdo gs_i) - p_i dot.
// The sequence sums are accumulated in float64.  Forward-only evaluation (ELBO scoring, where differences of
// nearly equal sums are taken) stays with the float64 kernel above.
// log2(1 + d) to float32 relative accuracy for any d > -1 (for small d the rounding of 1 + d is divided out).
__device__ __forceinline__ float log2_1p(float d) {
  // log(1 + d) = log(w) + (d - (w - 1)) / w with w = fl(1 + d); the correction is at most one rounding of w, so its
  // division by w is dropped (that changes the result by less than 6e-8 |d|)
  d = fmaxf(d, -0.99999994f);          // a class of zero weight (d = -1) keeps a finite logarithm: 0 * finite = 0
  const float w = 1.0f + d;
  return fmaf(d - (w - 1.0f), 1.44269504088896340736f, __builtin_amdgcn_logf(w));
}

// ------------------------------------------------------------------ per-timestep constants of the float32 loss
// What k_lvb_loss_tpl needs of the tables for a loss at timestep t, as float32 rows of 32 (d3pm_lvb_digest_f32):
//   row 0 / 1: a_diag / a_off by x_t (column x_t of Q_t)          row 2: mo_j (off-diagonal of Qbar_{t-1} by column)
//   row 3 / 4: Qbar_t[x0, x_t] for x0 == x_t / x0 != x_t by x_t   row 5 / 6: rho for x0 == x_t / x0 != x_t (MSA variant)
//   row 7 / 8: log rho likewise                                   row 9: mdo = diag - off of Qbar_{t-1}, sum_j mo_j
// Formed once per table in float64 (log rho as a difference of the float64 digests' logarithms: rho is close to one
// at late timesteps), so that the loss kernel's set-up is a handful of scalar loads and one 32-lane copy into LDS.
constexpr int LVB_ROWS = 10;
__global__ __launch_bounds__(64) void k_lvb_digest32(d3pm_tables_f64 tb, float* __restrict__ out) {
  const int K = tb.K, t = blockIdx.x, lane = threadIdx.x;
  const int tm = max(0, min(t - 1, tb.T - 1)), tq = min(t, tb.T - 1);
  float* o = out + (size_t)t * LVB_ROWS * 32;
  const bool ok = lane < K;
  const double vd64 = ok ? dgd(tb.qbar_digest, tm, 0, K)[lane] : 0.0, vo64 = ok ? dgd(tb.qbar_digest, tm, 1, K)[lane] : 0.0;
  const double dd = ok ? dgd(tb.qbar_digest, tq, 0, K)[lane] : 0.0, dof = ok ? dgd(tb.qbar_digest, tq, 1, K)[lane] : 0.0;
  const float vd = (float)vd64, vo = (float)vo64;
  const float sum_mo = wave_sum(vo);
  const float mdo = lane_broadcast(vd - vo, 0);
  if (lane < 32) {
    o[0 * 32 + lane] = ok ? (float)dgd(tb.qone_digest, tq, 0, K)[lane] : 0.0f;
    o[1 * 32 + lane] = ok ? (float)dgd(tb.qone_digest, tq, 1, K)[lane] : 0.0f;
    o[2 * 32 + lane] = vo;
    o[3 * 32 + lane] = (float)dd;
    o[4 * 32 + lane] = (float)dof;
    // rho = Qbar_t[x0, x_t] / Qbar_{t-1}[x0, x_t]; a pair that Qbar_{t-1} gives no weight never occurs (x_t was
    // drawn from it), its rho is left at zero
    o[5 * 32 + lane] = vd64 > 0.0 ? (float)(dd / vd64) : 0.0f;
    o[6 * 32 + lane] = vo64 > 0.0 ? (float)(dof / vo64) : 0.0f;
    o[7 * 32 + lane] = (ok && vd64 > 0.0) ? (float)(dgd(tb.qbar_digest, tq, 2, K)[lane] - dgd(tb.qbar_digest, tm, 2, K)[lane]) : 0.0f;
    o[8 * 32 + lane] = (ok && vo64 > 0.0) ? (float)(dgd(tb.qbar_digest, tq, 3, K)[lane] - dgd(tb.qbar_digest, tm, 3, K)[lane]) : 0.0f;
    o[9 * 32 + lane] = lane == 0 ? mdo : (lane == 1 ? sum_mo : 0.0f);
  }
}

// HYB: the driver's cross-entropy term in the same pass (Protein Generation Notebook cell-12, loss = (lvb + lambda *
// CrossEntropyLoss(outputs, tgt)) * n_tokens): -log_softmax(z)[x0] over ALL LD logits for every token whatever its
// element's timestep, summed per sequence into ce_seq; its gradient ces * (softmax(z) - e_x0) rides in the
// coefficients of the gradient's last loop (softmax_j = e_j f / S_all: one more term on e_j), so a class costs no
// further operation.  The softmax of the KL branch runs over the first K logits with their own maximum; the LD - K
// further logits enter through m_all = max(m, z_K..) and f = 2^(m - m_all), every exponent <= 0.
//
// The class x0 in the KL branch.  Its logit is read from the lane's LDS row and the slot is then overwritten with
// -inf before the row is read into registers, so the row the loops see has e_x0 = 0: the running sums over the
// other classes (R0 = sum_{j != x0} e_j^2, which must not be formed as a difference) need no per-class select, the
// slot's generic terms are those of a class of zero probability (taken out again with the same expressions, bit for
// bit), and the gradient loop leaves a zero in the slot, over which the token's own value for x0 is stored.
// cs = grad_scale / npos, ces = ce_grad_scale / npos from the host.
// T = float for one class, f2v for a pair of classes (the class loops run on pairs: packed float32 instructions,
// one issue slot for two classes).
__device__ __forceinline__ float rcp_each(float v) { return __builtin_amdgcn_rcpf(v); }
__device__ __forceinline__ f2v rcp_each(f2v v) { return f2v{__builtin_amdgcn_rcpf(v.x), __builtin_amdgcn_rcpf(v.y)}; }
__device__ __forceinline__ f2v exp2_each(f2v v) { return f2v{__builtin_amdgcn_exp2f(v.x), __builtin_amdgcn_exp2f(v.y)}; }
__device__ __forceinline__ f2v both(float v) { return f2v{v, v}; }
__device__ __forceinline__ f2v log2_1p(f2v d) {          // log2_1p(float) on both halves
  d = f2v{fmaxf(d.x, -0.99999994f), fmaxf(d.y, -0.99999994f)};
  const f2v w = d + both(1.0f);
  const f2v lw = {__builtin_amdgcn_logf(w.x), __builtin_amdgcn_logf(w.y)};
  return (d - (w - both(1.0f))) * both(1.44269504088896340736f) + lw;
}
template <class T> struct LvbGeneric { T qv, tj, d; };
template <class T>
__device__ __forceinline__ LvbGeneric<T> lvb_generic(T e2, T mo, T c3, T k1, T nc3) {
  // s'_j = mo_j + c3 e_j^2;  qv = mo_j / s'_j;  tj = e_j^2 / s'_j (so that qv + c3 tj = 1);
  // d_j = post_j / P_j - 1 = qv (rs - den) / den - c3 tj: both parts without cancellation
  const T r = rcp_each(e2 * c3 + mo);
  LvbGeneric<T> g;
  g.qv = mo * r;
  g.tj = e2 * r;
  g.d = g.qv * k1 + g.tj * nc3;
  return g;
}

template <int K, int LD, bool MSA, bool HYB>
__global__ __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(3, 4))) void k_lvb_loss_tpl(ProtArgs a, const float* __restrict__ lvb32,
                                                      const float* __restrict__ pred, const double* __restrict__ q,
                                                      int tmax, int msa, float cs, float ces, double log_prior,
                                                      double* __restrict__ part, double* __restrict__ per_seq,
                                                      double* __restrict__ ce_seq, float* __restrict__ dpred) {
  using Img = RowImage<LD>;
  constexpr float LOG2E = 1.44269504088896340736f, LN2 = 0.693147180559945309417f;
  constexpr int TR = MSA ? 9 : 5;                           // rows of the digest indexed by a token's classes
  extern __shared__ __align__(16) char lvb_rows[];          // [waves][64][STRIDE]
  static_assert(K % 2 == 0 && LD % 2 == 0, "classes are taken in pairs");
  constexpr int KP = K / 2, LP = LD / 2;
  __shared__ double red[HYB ? 64 : 32];                     // the sums of every 16-lane row of every wave
  __shared__ float s_tab[8][TR][32];                        // a copy per wave: no workgroup barrier
  const int b = blockIdx.x / a.chunks, chunk = blockIdx.x % a.chunks;
  // (the wave index through the scalar unit: what derives from it is scalar arithmetic and scalar branches)
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const unsigned p1 = (unsigned)min(a.npos, (chunk + 1) * a.chunk_pos);
  const unsigned pbeg = (unsigned)(chunk * a.chunk_pos);
  const char* __restrict__ prow = (const char*)(pred + (size_t)b * a.npos * LD);
  char* __restrict__ drow = (char*)(dpred + (size_t)b * a.npos * LD);
  const int64_t* __restrict__ x0row = a.x0 + (size_t)b * a.npos;
  const int64_t* __restrict__ xtrow = a.xt + (size_t)b * a.npos;
  char* const img = lvb_rows + (size_t)wave * Img::BYTES;
  char* const myrow = img + lane * Img::STRIDE;
  Img rows;
  int x0 = 0, xt = 0;
  const typename Img::LaneSplit ls = Img::split(lane);
  auto issue = [&](unsigned base) {
    const unsigned last = p1 - 1 - base, mine = min((unsigned)lane, last);
    x0 = (int)x0row[base + mine];
    xt = (int)xtrow[base + mine];
    // the wave's rows are one contiguous block of the sequence
    rows.request_flat(prow + (size_t)base * Img::RB, (last + 1) * Img::RB - Img::PB, lane, last >= 63);
  };
  unsigned base = pbeg + wave * 64;
  if (base < p1) issue(base);
  const int t0 = (int)a.t[b];
  const int t = (t0 == tmax) ? t0 : lvb_timestep(t0, a.tb.T);
  const int branch = (t == 1) ? 0 : (t == tmax ? 1 : 2);
  f2v mo[KP];
  float mdo = 0, sum_mo = 0;
  if (branch == 2) {
    // (t is uniform over the workgroup: these are scalar loads)
    const float* __restrict__ dg = lvb32 + (size_t)__builtin_amdgcn_readfirstlane(t) * (LVB_ROWS * 32);
    if (lane < 32) {
#pragma unroll
      for (int r = 0; r < TR; ++r) s_tab[wave][r][lane] = dg[r * 32 + lane];
    }
#pragma unroll
    for (int j = 0; j < KP; ++j) mo[j] = f2v{dg[2 * 32 + 2 * j], dg[2 * 32 + 2 * j + 1]};
    mdo = dg[9 * 32];
    sum_mo = dg[9 * 32 + 1];
  }
  double acc = 0, acc_ce = 0;
  if (branch == 1) {
    // (uniform, one element in T) KL(q || uniform) on the collater's forward probabilities (losses.py:58-60), no
    // gradient.  A loop of its own: inside the main loop the compiler hoists this path's double-precision constants
    // and its division in front of the loop, where every wave of every element pays for them.
    for (unsigned bq = pbeg + wave * 64; bq < p1; bq += blockDim.x) {
      if (bq + lane >= p1) continue;
      if (!q) { acc = (double)NAN; continue; }               // no probabilities given: NaN, nothing is read
      const double* qq = q + ((size_t)b * a.npos + bq + lane) * K;
      double s = 0;
      for (int j = 0; j < K; ++j) s += (qq[j] > 0.0 ? qq[j] * log(qq[j]) : 0.0) - qq[j] * log_prior;
      // 'batchmean' divides by the leading size of q[i]: L for a sequence (the final division), but only the
      // depth D for an MSA laid out [D,L,K]
      acc += s * (msa > 0 ? (double)a.npos / (double)msa : 1.0);
    }
  }
  for (;; base += blockDim.x) {
    if (base >= p1) break;                                   // (uniform over the wave)
    const bool live = base + lane < p1;
    x0 = clamp_index(x0, K);                                 // out-of-range tokens are clamped: nothing faults
    xt = clamp_index(xt, K);
    f2v e[LP];                                               // the lane's row as pairs of classes
    float tok_loss = 0, tok_ce = 0, set0 = 0, addt = 0;
    bool set_x0 = false;                                     // (uniform) the x0 slot is stored, not added to
    auto from_rows = [&]() {
      rows.deposit_flat(img, lane, ls);
      wave_lds_order();
      const float z_x0 = *reinterpret_cast<const float*>(myrow + 4 * x0);
      if (branch != 2) {
        // CrossEntropyLoss over all LD logits: the loss of an element at t == 1 (losses.py:54) and the driver's
        // second term (every element)
        Img::read_row_pairs(img, lane, e);
        float m = fmaxf(e[0].x, e[0].y);
#pragma unroll
        for (int j = 1; j < LP; ++j) m = fmaxf(fmaxf(m, e[j].x), e[j].y);
        const float mneg = -m * LOG2E;
        f2v s2 = {0.0f, 0.0f};
#pragma unroll
        for (int j = 0; j < LP; ++j) { e[j] = exp2_each(e[j] * both(LOG2E) + both(mneg)); s2 += e[j]; }
        const float s = s2.x + s2.y;
        tok_ce = fmaf(__builtin_amdgcn_logf(s), LN2, m - z_x0);
        tok_loss = branch == 0 ? tok_ce : 0.0f;                // (t == tmax: the element's term went to acc above)
        const float w = (branch == 0 ? cs : 0.0f) + ces;
        const float wi = w * __builtin_amdgcn_rcpf(s);
#pragma unroll
        for (int j = 0; j < LP; ++j) e[j] *= both(wi);
        set0 = -w;                                             // (added to the slot: set_x0 stays false)
      } else {
        const float z_xt = *reinterpret_cast<const float*>(myrow + 4 * xt);
        *reinterpret_cast<float*>(myrow + 4 * x0) = -INFINITY;  // after both reads (a wave's LDS operations are in order)
        wave_lds_order();
        Img::read_row_pairs(img, lane, e);
        float m = fmaxf(e[0].x, e[0].y);
#pragma unroll
        for (int j = 1; j < KP; ++j) m = fmaxf(fmaxf(m, e[j].x), e[j].y);
        m = fmaxf(m, z_x0);
        const float mneg = -m * LOG2E;
        f2v s2 = {0.0f, 0.0f}, R2 = {0.0f, 0.0f};
#pragma unroll
        for (int j = 0; j < KP; ++j) {
          e[j] = exp2_each(e[j] * both(LOG2E) + both(mneg));
          s2 += e[j];
          R2 = e[j] * e[j] + R2;
        }
        float s = s2.x + s2.y;
        const float R0 = R2.x + R2.y;                          // sum of e_j^2 over the classes other than x0
        const float e_x0 = __builtin_amdgcn_exp2f(fmaf(z_x0, LOG2E, mneg)), e_xt = __builtin_amdgcn_exp2f(fmaf(z_xt, LOG2E, mneg));
        s += e_x0;
        const float ee_x0 = e_x0 * e_x0, ee_xt = e_xt * e_xt;
        const float E2 = ee_x0 + R0;
        const float inv = __builtin_amdgcn_rcpf(s), inv2 = inv * inv;
        const float S2 = E2 * inv2;
        const float c3 = mdo * __builtin_amdgcn_rcpf(E2), nc3 = -c3;
        const float a_d = s_tab[wave][0][xt], a_o = s_tab[wave][1][xt], da = a_d - a_o;
        const float mo_x0 = s_tab[wave][2][x0], mo_xt = s_tab[wave][2][xt];
        const bool same_b = x0 == xt;
        const float same = same_b ? 1.0f : 0.0f;
        const float den = same_b ? s_tab[wave][3][xt] : s_tab[wave][4][xt];
        const float iden = __builtin_amdgcn_rcpf(den);
        const float w1 = c3 * R0, w2 = c3 * ee_xt;                    // b_x0 - s'_x0 and s'_xt - mo_xt, both >= 0
        const float sp_x0 = fmaf(ee_x0, c3, mo_x0), sp_xt = mo_xt + w2;                           // s'_x0, s'_xt
        const float Sm = a_o * (sum_mo + mdo);
        const float rs = fmaf(da, sp_xt, Sm);                        // rsum' (sum_j s'_j = sum_mo + mdo)
        // Every ratio post_j / P_j = mo_j rs / (s'_j den) is taken as 1 + d_j with
        //   d_j = (mo_j (rs - den) - (mdo / E2) e_j^2 den) / (s'_j den) = qv_j (rs - den) / den - c3 tj_j,
        // rs - den = da w2 (x0 != x_t) or -da w1 (x0 == x_t) known without cancellation (den = Sm + da B_xt, the way
        // Qbar_t is built from Qbar_{t-1}): the first-order parts of the terms cancel in the sum over classes, so a
        // logarithm of the rounded ratio (6e-8 absolute) would leave an error of that size in a sum that can be
        // far smaller.
        const float Dl = same_b ? -da * w1 : da * w2, k1 = Dl * iden;
        f2v qv[KP], G2 = {0.0f, 0.0f}, Hq2 = {0.0f, 0.0f}, Hp2 = {0.0f, 0.0f};
        // (One logarithm for several classes of equal weight, as the image kernel takes it, does not carry over: a
        // product of the ratios kept as 1 + D loses the factor's relative accuracy wherever a ratio is far below
        // one (early timesteps), and kept as a plain product it is rounded against one (late timesteps).)
#pragma unroll
        for (int j = 0; j < KP; ++j) {
          const LvbGeneric<f2v> g = lvb_generic(e[j] * e[j], mo[j], both(c3), both(k1), both(nc3));
          qv[j] = g.qv;
          G2 = mo[j] * log2_1p(g.d) + G2;
          Hq2 = mo[j] * g.qv + Hq2;
          Hp2 = mo[j] * g.tj + Hp2;                                   // sum_j e_j^2 qv_j
        }
        const float G1 = G2.x + G2.y, Hq = Hq2.x + Hq2.y, Hp = Hp2.x + Hp2.y;
        // special classes: x0 has B = mo + mdo, x_t has A = a_diag (both at once when x0 == x_t)
        const float r_x0 = __builtin_amdgcn_rcpf(sp_x0), r_xt = __builtin_amdgcn_rcpf(sp_xt);
        const float b_x0 = mo_x0 + mdo, a_x0 = fmaf(same, da, a_o);
        // the slot of x0 as the loop saw it (a generic class with e = 0: the same expressions, so that its terms
        // cancel bit for bit), then the class with its own B = mo + mdo; class x_t with A = a_diag.  Their d
        // without cancellation:
        //   x0 == x_t:  d0 = w1 Sm / (s'_x0 den)
        //   otherwise:  d0 = (w1 den + da w2 b_x0) / (s'_x0 den),   dt = -w2 Sm / (s'_xt den)
        const LvbGeneric<float> g0s = lvb_generic(0.0f, mo_x0, c3, k1, nc3);
        const float lg_x0 = log2_1p(g0s.d);
        const float d0r = (same_b ? w1 * Sm : fmaf(w1, den, da * w2 * b_x0)) * r_x0 * iden;
        const float dtr = -w2 * Sm * r_xt * iden;
        const float lb_x0 = log2_1p(d0r);
        const float lg_xt = log2_1p(dtr);
        float kl = a_o * G1;
        kl = fmaf(-a_o * mo_x0, lg_x0, kl);
        kl = fmaf(a_x0 * b_x0, lb_x0, kl);
        kl = fmaf((1.0f - same) * da * mo_xt, lg_xt, kl);
        tok_loss = kl * iden * LN2;                                  // logarithms above are base 2
        float rho = 1.0f;
        if constexpr (MSA) {
          rho = same_b ? s_tab[wave][5][xt] : s_tab[wave][6][xt];
          tok_loss = rho * ((same_b ? s_tab[wave][7][xt] : s_tab[wave][8][xt]) + tok_loss);
        }
        // gradient: gs_j = g0 - g1 qv_j for the generic slots (the slot of x0 among them, with e = 0), the exact
        // values gx0 at x0 and gxt at x_t
        const float b_xt = fmaf(same, mdo, mo_xt);
        const float psum = iden * fmaf(a_o, sum_mo + mdo, da * b_xt);
        const float iS2 = __builtin_amdgcn_rcpf(S2), irs = __builtin_amdgcn_rcpf(rs);
        const float k = (MSA ? cs * rho : cs) * iS2;
        const float g0 = a_o * k * psum * irs, g1 = a_o * k * iden;
        const float gx0 = a_x0 * k * (psum * irs - b_x0 * iden * r_x0);          // exact gs at x0
        const float d0 = gx0 - (g0 - g1 * g0s.qv);                                // over the slot's generic value
        const float gxt = a_d * k * (psum * irs - mo_xt * iden * r_xt);          // exact gs at x_t (x_t != x0)
        const float dt = (1.0f - same) * (gxt - (g0 - g1 * mo_xt * r_xt));
        const float pp_x0 = ee_x0 * inv2, pp_xt = ee_xt * inv2;
        const float W = fmaf(g0, sum_mo, fmaf(-g1, Hq, fmaf(d0, mo_x0, dt * mo_xt)));
        const float Sg = fmaf(inv2, fmaf(g0, R0, -g1 * Hp), fmaf(pp_x0, gx0, pp_xt * dt));   // sum_i p_i^2 gs_i
        const float dot = 2.0f * fmaf(W, S2, mdo * Sg);
        const float c4 = 2.0f * inv2 * fmaf(mdo, g0, W), c5 = 2.0f * inv2 * mdo * g1;
        float c6 = inv * dot;
        float ce_x0 = 0.0f;
        if constexpr (HYB) {
          // softmax over all LD logits from the K exponentials at hand: softmax_j = e_j f / S_all
          float S_all = s, f = 1.0f, m_all = m;
          if constexpr (LD > K) {
            float mx = fmaxf(e[KP].x, e[KP].y);
#pragma unroll
            for (int j = KP + 1; j < LP; ++j) mx = fmaxf(fmaxf(mx, e[j].x), e[j].y);
            m_all = fmaxf(m, mx);
            const float mneg_all = -m_all * LOG2E;
            f = __builtin_amdgcn_exp2f(fmaf(m, LOG2E, mneg_all));
            S_all = s * f;
#pragma unroll
            for (int j = KP; j < LP; ++j) {
              e[j] = exp2_each(e[j] * both(LOG2E) + both(mneg_all));
              S_all += e[j].x + e[j].y;
            }
          }
          tok_ce = fmaf(__builtin_amdgcn_logf(S_all), LN2, m_all - z_x0);
          const float wi = ces * __builtin_amdgcn_rcpf(S_all);
          c6 = fmaf(-wi, f, c6);
          ce_x0 = -ces;
#pragma unroll
          for (int j = KP; j < LP; ++j) e[j] *= both(wi);
        } else {
#pragma unroll
          for (int j = KP; j < LP; ++j) e[j] = both(0.0f);
        }
        // the token's own value for x0 (stored over the loop's zero) and the extra of x_t (added)
        set0 = fmaf(2.0f * pp_x0, fmaf(mdo, gx0, W), fmaf(-c6, e_x0, ce_x0));
        addt = 2.0f * pp_xt * mdo * dt;
        set_x0 = true;
#pragma unroll
        for (int j = 0; j < KP; ++j) e[j] *= e[j] * (qv[j] * both(-c5) + both(c4)) + both(-c6);
      }
      acc += live ? (double)tok_loss : 0.0;
      if constexpr (HYB) acc_ce += live ? (double)tok_ce : 0.0;
    };
    if constexpr (HYB) {
      from_rows();
    } else {
      if (branch == 1) {
#pragma unroll
        for (int j = 0; j < LP; ++j) e[j] = both(0.0f);
      } else {
        from_rows();
      }
    }
    // gradient rows back through the image; the special classes are patched in the lane's own row
    Img::write_row_pairs(img, lane, e);
    if (set_x0) *reinterpret_cast<float*>(myrow + 4 * x0) = set0;
    else *reinterpret_cast<float*>(myrow + 4 * x0) += set0;
    *reinterpret_cast<float*>(myrow + 4 * xt) += addt;
    wave_lds_order();
    Img::store_flat(img, drow + (size_t)base * Img::RB, min(64u, p1 - base), lane, ls);
    if (base + blockDim.x >= p1) break;
    wave_lds_order();
    // (the next pass's rows are requested here, where the pass's registers are free.  Requested right after the
    // deposit instead they travel while the pass computes and hold 26 more registers through it: 142 -> 168, still
    // three waves per SIMD, measured 36.4 against 36.9 us at B = 2048 and 6.6 against 6.45 us with the
    // cross-entropy term at B = 256 -- the large launch runs at what the memory system gives a mixed read / write
    // stream either way, so the simpler order stays.  Held to 128 registers for a fourth wave the kernel spills
    // 10 of them: 42.6 us.)
    issue(base + blockDim.x);
  }
  // Totals: rows of 16 lanes by DPP, every wave's four row totals to LDS, and the first wave adds the (at most 32)
  // of them up -- two reduction steps less on every wave, on the tail of its last pass.
  acc = group_sum<16>(acc);
  if constexpr (HYB) acc_ce = group_sum<16>(acc_ce);
  if ((lane & 15) == 0) {
    red[wave * 4 + (lane >> 4)] = acc;
    if constexpr (HYB) red[32 + wave * 4 + (lane >> 4)] = acc_ce;
  }
  __syncthreads();
  if (wave == 0) {
    const int nrow = (int)(blockDim.x >> 6) * 4;
    double sacc = lane < nrow ? red[lane] : 0.0, sce = 0;
    sacc = group_sum<32>(sacc);
    if constexpr (HYB) {
      sce = lane < nrow ? red[32 + lane] : 0.0;
      sce = group_sum<32>(sce);
    }
    if (lane != 0) return;
    if (a.chunks == 1) per_seq[b] = sacc / (double)a.npos;
    else part[(size_t)b * a.chunks + chunk] = sacc;
    if constexpr (HYB) {
      if (a.chunks == 1) { if (ce_seq) ce_seq[b] = sce / (double)a.npos; }
      else part[(size_t)a.B * a.chunks + (size_t)b * a.chunks + chunk] = sce;
    }
  }
}

// per_seq[b] = sum of the element's chunk partials / npos; the same for the cross-entropy partials (second half of
// `part`) when ce_seq is given.
__global__ void k_lvb_finalize(const double* __restrict__ part, int B, int chunks, long npos,
                               double* __restrict__ per_seq, double* __restrict__ ce_seq) {
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= B) return;
  double s = 0;
  for (int c = 0; c < chunks; ++c) s += part[(size_t)b * chunks + c];
  per_seq[b] = s / (double)npos;
  if (ce_seq) {
    double sc = 0;
    for (int c = 0; c < chunks; ++c) sc += part[(size_t)B * chunks + (size_t)b * chunks + c];
    ce_seq[b] = sc / (double)npos;
  }
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
  const int t = lvb_timestep((int)a.t[b], a.tb.T);
  double md[L::V], mo[L::V];
  load_digest_d<L>(a.tb.qbar_digest, t - 1, 0, K, g, md);
  load_digest_d<L>(a.tb.qbar_digest, t - 1, 1, K, g, mo);
  const long p1 = min(a.npos, (chunk + 1) * a.chunk_pos);
  for (long pos = chunk * a.chunk_pos + slot; pos < p1; pos += PBLOCK / L::G) {
    const long n = (long)b * a.npos + pos;
    const int xt = clamp_index((int)a.xt[n], K);
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
  const int t = lvb_timestep((int)a.t[b], a.tb.T);
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
      const int xt = clamp_index(xq[u], K);
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
      const float u01v = u01_open(word);
      const int pick = group_pick<L::G, L::V>(r, g, u01v, xt, cls);
      if (g == 0 && pos < p1) out[(long)b * a.npos + pos] = pick;
    }
  }
}

// ------------------------------------------------------------------ reverse step, device noise, one token per lane
// As k_lvb_reverse_device_noise (float32 marginal, one uniform per token, nothing to match bit for bit), with a
// whole token per lane: rows through the wave-private LDS image, per-class constants in scalar registers, and the
// inverse CDF a running sum inside the lane -- no cross-lane scan at all.  With e_j = 2^(z_j - max) and
// s'_j = mo_j + (mdo / sum e^2) e_j^2 the unnormalised marginal is A_j s'_j; the factor a_off is divided out, which
// leaves generic weights g_j = s'_j and an extra mass X = (a_diag / a_off - 1) g_xt on class x_t.  One uniform
// decides both: target = u (G + X); target < X picks x_t, otherwise the generic scan runs with target - X.
// The last class (the gap of an MSA) has its weight divided by `last_class_div`, or removed in the query row.
template <int K, int LD>
__global__ __launch_bounds__(512) void k_lvb_reverse_tpl(ProtArgs a, const float* __restrict__ pred, uint64_t seed,
                                                         uint64_t offset, const uint64_t* __restrict__ offset_dev,
                                                         float last_class_div, long exclude_last_prefix,
                                                         int64_t* __restrict__ out) {
  using Img = RowImage<LD>;
  constexpr float LOG2E = 1.44269504088896340736f;
  extern __shared__ __align__(16) char rev_rows[];          // [waves][64][STRIDE]
  __shared__ float s_tab[8][3][32];                         // per wave: a_diag, a_off of Q_t by x_t; mo by class
  if (offset_dev) offset += *offset_dev;
  const int b = blockIdx.x / a.chunks, chunk = blockIdx.x % a.chunks;
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const unsigned p1 = (unsigned)min(a.npos, (chunk + 1) * a.chunk_pos);
  const unsigned pbeg = (unsigned)(chunk * a.chunk_pos);
  const char* __restrict__ prow = (const char*)(pred + (size_t)b * a.npos * LD);
  const int64_t* __restrict__ xtrow = a.xt + (size_t)b * a.npos;
  int64_t* __restrict__ orow = out + (size_t)b * a.npos;
  char* const img = rev_rows + (size_t)wave * Img::BYTES;
  Img rows;
  int xt = 0;
  auto issue = [&](unsigned base) {
    const unsigned last = p1 - 1 - base, mine = min((unsigned)lane, last);
    xt = (int)xtrow[base + mine];
    rows.request(prow, base, last, lane);
  };
  unsigned base = pbeg + wave * 64;
  if (base < p1) issue(base);
  const int t = lvb_timestep((int)a.t[b], a.tb.T);
  const float vd = lane < K ? (float)dgd(a.tb.qbar_digest, t - 1, 0, K)[lane] : 0.0f;
  const float vo = lane < K ? (float)dgd(a.tb.qbar_digest, t - 1, 1, K)[lane] : 0.0f;
  if (lane < K) {
    s_tab[wave][0][lane] = (float)dgd(a.tb.qone_digest, t, 0, K)[lane];
    s_tab[wave][1][lane] = (float)dgd(a.tb.qone_digest, t, 1, K)[lane];
    s_tab[wave][2][lane] = vo;
  }
  const float mdo = lane_broadcast(vd - vo, 0);
  float mo[K];
#pragma unroll
  for (int j = 0; j < K; ++j) mo[j] = lane_broadcast(vo, j);
  const float inv_last = 1.0f / last_class_div;
  wave_lds_order();
  for (;; base += blockDim.x) {
    if (base >= p1) break;                                   // (uniform over the wave)
    const unsigned pos = base + lane;
    // the uniform of this token: Philox counter = global token index
    const U4 rnd = philox4x32_10((uint64_t)b * (uint64_t)a.npos + pos, offset, seed);
    const float u = u01_open(rnd.x);
    xt = clamp_index(xt, K);                                 // out-of-range tokens are clamped: nothing faults
    rows.deposit(img, lane);
    wave_lds_order();
    float e[LD];
    Img::read_row(img, lane, e);
    const float z_xt = *reinterpret_cast<const float*>(img + lane * Img::STRIDE + 4 * xt);
    float m = e[0];
#pragma unroll
    for (int j = 1; j < K; ++j) m = fmaxf(m, e[j]);
    const float mneg = -m * LOG2E;
    float E2 = 0;
#pragma unroll
    for (int j = 0; j < K; ++j) {
      const float ej = __builtin_amdgcn_exp2f(fmaf(e[j], LOG2E, mneg));
      e[j] = ej * ej;
      E2 += e[j];
    }
    const float c3 = mdo * __builtin_amdgcn_rcpf(E2);
    const float w_last = (long)pos < exclude_last_prefix ? 0.0f : inv_last;
    float G = 0;
#pragma unroll
    for (int j = 0; j < K; ++j) {
      e[j] = fmaf(e[j], c3, mo[j]);                          // g_j = s'_j
      if (j == K - 1) e[j] *= w_last;
      G += e[j];
    }
    const float a_d = s_tab[wave][0][xt], a_o = s_tab[wave][1][xt];
    const float e_xt = __builtin_amdgcn_exp2f(fmaf(z_xt, LOG2E, mneg));
    const float g_xt = fmaf(e_xt * e_xt, c3, s_tab[wave][2][xt]) * (xt == K - 1 ? w_last : 1.0f);
    const float X = (a_d - a_o) * __builtin_amdgcn_rcpf(a_o) * g_xt;
    const float target = u * (G + X) - X;                    // < 0: the extra mass of x_t was hit
    float run = 0;
    int cnt = 0;                                             // classes whose cumulative weight is <= target
#pragma unroll
    for (int j = 0; j < K - 1; ++j) {
      run += e[j];
      cnt += (run <= target) ? 1 : 0;
    }
    int pick = target < 0.0f ? xt : cnt;
    // rounding can carry the scan onto a class of zero weight (the excluded last class): step back to the last
    // class that has weight
    if (pick == K - 1 && w_last == 0.0f) pick = K - 2;
    if (pos < p1) orow[pos] = pick;
    if (base + blockDim.x >= p1) break;
    wave_lds_order();
    issue(base + blockDim.x);
  }
}

// ------------------------------------------------------------------ reverse step, device noise, tokens of the whole batch in one row
// The same draw as k_lvb_reverse_tpl with the [B, npos] tokens taken as one flat sequence: wave w owns tokens
// 64 w .. 64 w + 63 wherever the batch elements begin, so a length that is no multiple of 64 (263 = 4 x 64 + 7)
// leaves no idle lanes except in the launch's very last wave.  With npos >= 64 a wave touches at most two batch
// elements; if their timesteps differ the per-timestep part runs once for each with the other's lanes masked (the
// samplers step every sequence at one timestep, so this is the rare path).  No workgroup-level step: waves are
// independent and a workgroup is only a launch unit.
//   * rows: the wave's 64 rows are one contiguous block, addressed without any division (request_flat);
//   * weights: with P2_j = sum_{i<=j} e_i^2 (the running sum of the softmax pass itself) and MO_j = sum_{i<=j} mo_i
//     (per timestep, wave-uniform) the cumulative weight of classes 0..j is MO_j + c3 P2_j: one fused multiply-add
//     and one compare per class, no second running sum;
//   * noise: one 32-bit word per token from Philox2x32-10, keyed per launch by a Philox4x32-10 call on
//     (offset, seed) that the scalar unit evaluates (its inputs are uniform).
template <int K, int LD>
__global__ __launch_bounds__(256) void k_lvb_reverse_flat(ProtArgs a, const float* __restrict__ pred, uint64_t seed,
                                                          uint64_t offset, const uint64_t* __restrict__ offset_dev,
                                                          float last_class_div, long exclude_last_prefix,
                                                          int64_t* __restrict__ out, unsigned ntok, float inv_npos) {
  using Img = RowImage<LD>;
  constexpr float LOG2E = 1.44269504088896340736f;
  extern __shared__ __align__(16) char flat_rows[];         // [waves][64][STRIDE]
  __shared__ float s_tab[4][3][32];                         // per wave: a_diag, a_off of Q_t by x_t; mo by class
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const unsigned base = (blockIdx.x * (blockDim.x >> 6) + wave) * 64u;
  if (base >= ntok) return;                                 // (uniform over the wave; the kernel has no barrier)
  const unsigned np = (unsigned)a.npos;
  // batch element and position of the wave's first token: float quotient, off by one at most (B <= 2^20)
  unsigned b0 = __builtin_amdgcn_readfirstlane((unsigned)((float)base * inv_npos));
  int r0 = (int)(base - b0 * np);
  if (r0 < 0) { b0 -= 1; r0 += (int)np; }
  else if (r0 >= (int)np) { b0 += 1; r0 -= (int)np; }
  unsigned pos = (unsigned)r0 + lane;
  const bool second = pos >= np;                            // this lane's token belongs to batch element b0 + 1
  if (second) pos -= np;
  const unsigned nrows = min(64u, ntok - base);
  const bool spans = (unsigned)r0 + nrows > np;             // (uniform)
  char* const img = flat_rows + (size_t)wave * Img::BYTES;
  Img rows;
  rows.request_flat((const char*)pred + (size_t)base * Img::RB, nrows * Img::RB - Img::PB, lane, nrows == 64);
  int xt = (int)a.xt[base + min((unsigned)lane, nrows - 1)];
  const int t0 = lvb_timestep((int)a.t[b0], a.tb.T);
  const int t1 = spans ? lvb_timestep((int)a.t[b0 + 1], a.tb.T) : t0;
  const bool two = t1 != t0;                                // (uniform)
  float vd = 0, vo = 0, ad = 0, ao = 0;
  auto fetch = [&](int t) {
    if (lane < K) {
      vd = (float)dgd(a.tb.qbar_digest, t - 1, 0, K)[lane];
      vo = (float)dgd(a.tb.qbar_digest, t - 1, 1, K)[lane];
      ad = (float)dgd(a.tb.qone_digest, t, 0, K)[lane];
      ao = (float)dgd(a.tb.qone_digest, t, 1, K)[lane];
    }
  };
  fetch(t0);
  // the uniform of this token
  if (offset_dev) offset += *offset_dev;
  const U4 pad = philox4x32_10(offset, 0x243F6A8885A308D3ull, seed);
  const U2 rnd = philox2x32_10((base + lane) ^ pad.x, pad.y, pad.z);
  const float u = u01_open(rnd.x);
  xt = clamp_index(xt, K);                                   // out-of-range tokens are clamped: nothing faults
  rows.deposit_flat(img, lane, Img::split(lane));
  wave_lds_order();
  float z[LD];
  Img::read_row(img, lane, z);
  const float z_xt = *reinterpret_cast<const float*>(img + lane * Img::STRIDE + 4 * xt);
  float m = z[0];
#pragma unroll
  for (int j = 1; j < K; ++j) m = fmaxf(m, z[j]);
  const float mneg = -m * LOG2E;
  float P2[K];                                               // running sums of e_j^2
  float acc = 0;
#pragma unroll
  for (int j = 0; j < K; ++j) {
    const float ej = __builtin_amdgcn_exp2f(fmaf(z[j], LOG2E, mneg));
    acc = fmaf(ej, ej, acc);
    P2[j] = acc;
  }
  const float e2_last = P2[K - 1] - P2[K - 2];
  const float e_xt = __builtin_amdgcn_exp2f(fmaf(z_xt, LOG2E, mneg));
  const float inv_e2 = __builtin_amdgcn_rcpf(acc);
  const float inv_last = 1.0f / last_class_div;
  const float w_last = (long)pos < exclude_last_prefix ? 0.0f : inv_last;
  int pick = 0;
  auto draw = [&]() {
    // per-timestep constants from the lanes that fetched them
    if (lane < 32) {
      s_tab[wave][0][lane] = ad;
      s_tab[wave][1][lane] = ao;
      s_tab[wave][2][lane] = vo;
    }
    const float mdo = lane_broadcast(vd - vo, 0);
    float sc = vo;                                           // inclusive scan of mo over the lanes (zero from lane K on)
    sc += dpp_mov<0x111>(sc);                                // row_shr 1, 2, 4, 8: lanes without a source read zero
    sc += dpp_mov<0x112>(sc);
    sc += dpp_mov<0x114>(sc);
    sc += dpp_mov<0x118>(sc);
    const float row0 = lane_broadcast(sc, 15);
    if (lane >= 16) sc += row0;
    float MO[K - 1];
#pragma unroll
    for (int j = 0; j < K - 1; ++j) MO[j] = lane_broadcast(sc, j);
    const float mo_last = lane_broadcast(vo, K - 1);
    wave_lds_order();
    const float c3 = mdo * inv_e2;
    const float g_last = fmaf(e2_last, c3, mo_last) * w_last;
    const float G = fmaf(P2[K - 2], c3, MO[K - 2]) + g_last;
    const float a_d = s_tab[wave][0][xt], a_o = s_tab[wave][1][xt];
    const float g_xt = fmaf(e_xt * e_xt, c3, s_tab[wave][2][xt]) * (xt == K - 1 ? w_last : 1.0f);
    const float X = (a_d - a_o) * __builtin_amdgcn_rcpf(a_o) * g_xt;
    const float target = u * (G + X) - X;                    // < 0: the extra mass of x_t was hit
    int cnt = 0;                                             // classes whose cumulative weight is <= target
#pragma unroll
    for (int j = 0; j < K - 1; ++j) cnt += (fmaf(P2[j], c3, MO[j]) <= target) ? 1 : 0;
    int p = target < 0.0f ? xt : cnt;
    // rounding can carry the count onto a class of zero weight (the excluded last class): step back
    if (p == K - 1 && w_last == 0.0f) p = K - 2;
    return p;
  };
  pick = draw();
  if (two) {                                                 // (uniform) the wave's second batch element, own timestep
    wave_lds_order();
    fetch(t1);
    const int p1 = draw();
    pick = second ? p1 : pick;
  }
  if (lane < (int)nrows) out[base + lane] = pick;
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
  if (!expo && !probs && tb->K <= 64) {
    ProtArgs a = make_prot(tb, x0, nullptr, t, B, npos, 1);
    hipLaunchKernelGGL(k_corrupt_lane, dim3(B * a.chunks), dim3(256), 0, (hipStream_t)stream, a, seed, offset, x_t);
    D3PM_LAUNCH_CHECK("k_corrupt_lane");
    return D3PM_OK;
  }
  return dispatch_layout(tb->K, false, [&](auto lay) {
    using L = decltype(lay);
    ProtArgs a = make_prot(tb, x0, nullptr, t, B, npos, L::G);
    hipLaunchKernelGGL(k_corrupt<L>, dim3(B * a.chunks), dim3(PBLOCK), 0, (hipStream_t)stream, a, expo, seed, offset,
                       x_t, probs);
    D3PM_LAUNCH_CHECK("k_corrupt");
    return D3PM_OK;
  });
}

extern "C" size_t d3pm_lvb_digest_f32_size(int T) { return (size_t)(T > 0 ? T : 0) * LVB_ROWS * 32 * sizeof(float); }

extern "C" int d3pm_lvb_digest_f32(const d3pm_tables_f64* tb, float* out, void* stream) {
  D3PM_CHECK_ARG(tb && tb->qbar_digest && tb->qone_digest && out, "null pointer");
  D3PM_CHECK_ARG(tb->K >= 2 && tb->K <= 32 && tb->T >= 1, "the float32 loss digest holds up to 32 classes (K=%d, T=%d)",
                 tb->K, tb->T);
  hipLaunchKernelGGL(k_lvb_digest32, dim3(tb->T), dim3(64), 0, (hipStream_t)stream, *tb, out);
  D3PM_LAUNCH_CHECK("k_lvb_digest32");
  return D3PM_OK;
}

extern "C" size_t d3pm_lvb_loss_workspace(int B, long npos, int K) {
  (void)K;
  long passes = npos > 0 ? npos : 1;
  long chunks = passes < 4096 ? passes : 4096;
  // (two partial sums per workgroup: the element's loss term and its cross-entropy)
  return 2 * (size_t)(B > 0 ? B : 1) * (size_t)chunks * sizeof(double);
}

static int lvb_loss_impl(const d3pm_tables_f64* tb, const float* pred, long ld, const int64_t* x0,
                         const int64_t* x_t, const int64_t* t, const double* q, int B, long npos, int tmax, int msa,
                         double grad_scale, bool hyb, double ce_grad_scale, double* per_seq, double* ce_seq,
                         float* dpred, void* workspace, size_t workspace_bytes, void* stream) {
  if (int rc = check_prot(tb, B, npos)) return rc;
  D3PM_CHECK_ARG(ld >= tb->K && ld <= 256, "row stride (n_tokens) must be in [K, 256], got %ld", ld);
  if (B == 0) return D3PM_OK;
  D3PM_CHECK_ARG(npos > 0, "empty sequence: mean over zero positions");
  D3PM_CHECK_ARG(pred && x0 && x_t && t && per_seq, "null pointer");
  hipStream_t s = (hipStream_t)stream;
  // sample_prior builds a float32 1/K and takes its float32 log (losses.py:11-12,60)
  const double log_prior = (double)logf(1.0f / (float)tb->K);
  // training form (both D3PMLVBLoss and its MSA variant): one token per lane, float32 (K = 22 with rows of 22 or
  // 24 logits, the two alphabets of the reference's denoisers); every other shape, and forward-only evaluation,
  // on the float64 kernel
  if (dpred && tb->lvb_digest && tb->K == 22 && (ld == 22 || ld == 24) && ((uintptr_t)pred & 15) == 0 &&
      ((uintptr_t)dpred & 15) == 0 && npos * ld < 0x1fffffffL) {
    ProtArgs a = make_prot(tb, x0, x_t, t, B, npos, 1);
    int block;
    // The kernel holds 168 registers a lane: three waves per SIMD.  Up to one batch element per CU: one workgroup
    // of up to eight waves per element (two per SIMD, every wave a single pass).  Larger batches: 256-thread
    // workgroups, three of them resident per CU (a 512-thread workgroup leaves no room for a second one), each wave
    // walking its chunk pass by pass with the next rows requested ahead (L = 512, one workgroup per element against
    // 256-thread workgroups: B = 320 12.5 / 11.4 us, 512 14.0 / 12.1, 768 23.9 / 18.2, 2048 60.8 / 40.6).  Smaller
    // batches: 256-thread chunks.
    if (B >= 192 && B <= 256) {
      const long want = ((npos + 63) / 64) * 64;
      block = (int)(want > 512 ? 512 : want);
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
    double* part = (double*)workspace;
    if (a.chunks > 1) {
      const size_t need = (hyb ? 2 : 1) * (size_t)B * a.chunks * sizeof(double);
      D3PM_CHECK_WORKSPACE(workspace, workspace_bytes, need);
    }
    auto go = [&](auto ldc, auto msac, auto hybc) -> int {
      constexpr int LDC = decltype(ldc)::value;
      auto kern = k_lvb_loss_tpl<22, LDC, decltype(msac)::value, decltype(hybc)::value>;
      const size_t sh = (size_t)(block / 64) * RowImage<LDC>::BYTES;
      if (sh > 48 * 1024) {
        hipError_t er = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)sh);
        if (er != hipSuccess) return hip_fail(er, "hipFuncSetAttribute(MaxDynamicSharedMemorySize)");
      }
      hipLaunchKernelGGL(kern, dim3(B * a.chunks), dim3(block), sh, s, a, tb->lvb_digest, pred, q, tmax, msa,
                         (float)(grad_scale / (double)npos), (float)(ce_grad_scale / (double)npos), log_prior, part, per_seq,
                         ce_seq, dpred);
      D3PM_LAUNCH_CHECK("k_lvb_loss_tpl");
      return D3PM_OK;
    };
    using I22 = std::integral_constant<int, 22>;
    using I24 = std::integral_constant<int, 24>;
    auto go_h = [&](auto ldc, auto msac) -> int {
      return hyb ? go(ldc, msac, std::true_type{}) : go(ldc, msac, std::false_type{});
    };
    const int rc = msa > 0 ? (ld == 22 ? go_h(I22{}, std::true_type{}) : go_h(I24{}, std::true_type{}))
                           : (ld == 22 ? go_h(I22{}, std::false_type{}) : go_h(I24{}, std::false_type{}));
    if (rc) return rc;
    if (a.chunks > 1) {
      hipLaunchKernelGGL(k_lvb_finalize, dim3((B + 255) / 256), dim3(256), 0, s, part, B, a.chunks, npos, per_seq,
                         hyb ? ce_seq : nullptr);
      D3PM_LAUNCH_CHECK("k_lvb_finalize");
    }
    return D3PM_OK;
  }
  auto run = [&](auto lay) {
    using L = decltype(lay);
    ProtArgs a = make_prot(tb, x0, x_t, t, B, npos, L::G);
    const size_t need = (hyb ? 2 : 1) * (size_t)B * a.chunks * sizeof(double);
    D3PM_CHECK_WORKSPACE(workspace, workspace_bytes, need);
    double* part = (double*)workspace;
    double* part_ce = part + (size_t)B * a.chunks;
    const dim3 grid(B * a.chunks), block(PBLOCK);
    if (dpred)
      hipLaunchKernelGGL((k_lvb_loss<L, true>), grid, block, 0, s, a, pred, ld, q, tmax, msa, grad_scale, log_prior,
                         part, dpred, hyb ? 1 : 0, ce_grad_scale, part_ce);
    else
      hipLaunchKernelGGL((k_lvb_loss<L, false>), grid, block, 0, s, a, pred, ld, q, tmax, msa, grad_scale, log_prior,
                         part, dpred, hyb ? 1 : 0, ce_grad_scale, part_ce);
    D3PM_LAUNCH_CHECK("k_lvb_loss");
    hipLaunchKernelGGL(k_lvb_finalize, dim3((B + 255) / 256), dim3(256), 0, s, part, B, a.chunks, npos, per_seq,
                       hyb ? ce_seq : nullptr);
    D3PM_LAUNCH_CHECK("k_lvb_finalize");
    return D3PM_OK;
  };
  return dispatch_layout((int)ld, false, run);       // (four lanes per token measured no faster than eight)
}

extern "C" int d3pm_lvb_loss(const d3pm_tables_f64* tb, const float* pred, long ld, const int64_t* x0,
                             const int64_t* x_t, const int64_t* t, const double* q, int B, long npos, int tmax, int msa,
                             double grad_scale, double* per_seq, float* dpred, void* workspace, size_t workspace_bytes,
                             void* stream) {
  return lvb_loss_impl(tb, pred, ld, x0, x_t, t, q, B, npos, tmax, msa, grad_scale, false, 0.0, per_seq, nullptr, dpred,
                       workspace, workspace_bytes, stream);
}

extern "C" int d3pm_lvb_hybrid_loss(const d3pm_tables_f64* tb, const float* pred, long ld, const int64_t* x0,
                                    const int64_t* x_t, const int64_t* t, const double* q, int B, long npos, int tmax,
                                    int msa, double grad_scale, double ce_grad_scale, double* per_seq, double* ce_seq,
                                    float* dpred, void* workspace, size_t workspace_bytes, void* stream) {
  D3PM_CHECK_ARG(ce_seq, "null pointer (ce_seq)");
  return lvb_loss_impl(tb, pred, ld, x0, x_t, t, q, B, npos, tmax, msa, grad_scale, true, ce_grad_scale, per_seq, ce_seq,
                       dpred, workspace, workspace_bytes, stream);
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
  if (!expo && tb->K == 22 && (ld == 22 || ld == 24) && ((uintptr_t)pred & 15) == 0 &&
      npos >= 64 && B <= (1 << 20) && (long)B * npos * ld < 0x3fffffffL) {
    // tokens of the whole batch as one row, 64 per wave
    ProtArgs a = make_prot(tb, nullptr, x_t, t, B, npos, 1);
    const unsigned ntok = (unsigned)((long)B * npos);
    const unsigned waves = (ntok + 63) / 64;
    constexpr int block = 256;
    const unsigned wpb = block / 64;
    auto go = [&](auto ldc) -> int {
      constexpr int LDC = decltype(ldc)::value;
      const size_t sh = (size_t)wpb * RowImage<LDC>::BYTES;
      hipLaunchKernelGGL((k_lvb_reverse_flat<22, LDC>), dim3((waves + wpb - 1) / wpb), dim3(block), sh,
                         (hipStream_t)stream, a, pred, seed, offset, offset_dev, (float)last_class_div,
                         exclude_last_prefix, out, ntok, 1.0f / (float)npos);
      D3PM_LAUNCH_CHECK("k_lvb_reverse_flat");
      return D3PM_OK;
    };
    return ld == 22 ? go(std::integral_constant<int, 22>{}) : go(std::integral_constant<int, 24>{});
  }
  if (!expo && tb->K == 22 && (ld == 22 || ld == 24) && ((uintptr_t)pred & 15) == 0) {
    // one token per lane
    ProtArgs a = make_prot(tb, nullptr, x_t, t, B, npos, 1);
    int block;
    if (B >= 192) {
      const long want = ((npos + 63) / 64) * 64;
      block = (int)(want > 512 ? 512 : want);
      a.chunks = 1;
      a.chunk_pos = npos;
    } else {
      block = 256;
      const TokenGrid tg = token_grid(B, npos, block, 1024);
      a.chunks = tg.chunks;
      a.chunk_pos = tg.chunk_pos;
    }
    auto go = [&](auto ldc) -> int {
      constexpr int LDC = decltype(ldc)::value;
      auto kern = k_lvb_reverse_tpl<22, LDC>;
      const size_t sh = (size_t)(block / 64) * RowImage<LDC>::BYTES;
      if (sh > 48 * 1024) {
        hipError_t er = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)sh);
        if (er != hipSuccess) return hip_fail(er, "hipFuncSetAttribute(MaxDynamicSharedMemorySize)");
      }
      hipLaunchKernelGGL(kern, dim3(B * a.chunks), dim3(block), sh, (hipStream_t)stream, a, pred, seed, offset,
                         offset_dev, (float)last_class_div, exclude_last_prefix, out);
      D3PM_LAUNCH_CHECK("k_lvb_reverse_tpl");
      return D3PM_OK;
    };
    return ld == 22 ? go(std::integral_constant<int, 22>{}) : go(std::integral_constant<int, 24>{});
  }
  if (!expo && ld <= 32) {
    // few lanes per token: the draw is cheap per class, the cross-lane steps are what costs
    constexpr int rev_blocks = 1024;
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
      if (best == 192) return launch_blk(lay, std::integral_constant<int, 192>{});
      if (best == 128) return launch_blk(lay, std::integral_constant<int, 128>{});
      return launch_blk(lay, std::integral_constant<int, 256>{});
    };
    // four lanes per token (two and eight measured slower)
    return ld <= 24 ? launch(Layout<4, 6, false>{}) : launch(Layout<4, 8, false>{});
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
