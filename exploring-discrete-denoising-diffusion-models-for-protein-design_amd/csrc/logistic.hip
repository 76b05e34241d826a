// Discretised-logistic output head of the image models: logits of the K pixel
// bins from a location and a log-scale per position, and the gradient back to
// the two parameters.  Follows _get_logits_from_logistic_pars
// (d3pm/diffusion_discrete.py:288-309) and utils.log_min_exp (d3pm/utils.py:15-18)
// operation by operation in float32 (same order, same intermediate roundings up
// to the last ulp of exp / log1p), so that the bins whose value is dominated by
// rounding (far right tail: 1 + eps - exp(lo - hi) with lo - hi -> 0) scatter
// the way the reference's tensor expression does.
//
// Lane layout: a position's K bins are spread over G = min(K, 64) consecutive
// lanes, lane g owning bins g, g + G, ...; writes of a wave are contiguous.
// Forward is bound by the [N, K] float32 store, backward by the [N, K] load.
#include "d3pm_device.h"

namespace d3pm {

constexpr int LG_BLOCK = 256;
constexpr float LN2 = 0.693147180559945309417f;
__device__ __forceinline__ float fast_exp2(float x) { return __builtin_amdgcn_exp2f(x); }   // v_exp_f32
__device__ __forceinline__ float fast_log2(float x) { return __builtin_amdgcn_logf(x); }    // v_log_f32
__device__ __forceinline__ float fast_rcp(float x) { return __builtin_amdgcn_rcpf(x); }     // v_rcp_f32

// exp and log1p on the hardware exp2 / log2 (v_exp_f32, v_log_f32: 1 ulp) with the
// rounding of their argument folded back in, so that both stay within 2 ulp over the
// ranges used here; the library versions cost 5-10x the instructions and made the
// head VALU-bound at 6% of its memory roofline.
__device__ __forceinline__ float exp_acc(float x) {
  const float L2E = 1.44269504088896341f, L2E_LO = 1.92596299e-8f;      // log2(e) split in two floats
  const float t = x * L2E;
  const float r = fmaf(x, L2E_LO, fmaf(x, L2E, -t));                    // what the product's rounding dropped
  const float p = fast_exp2(t);
  return fmaf(p, r * LN2, p);
}
// log(1 + q), q > -1: log of the rounded sum plus the first-order term of what the sum dropped
__device__ __forceinline__ float log1p_acc(float q) {
  const float u = 1.0f + q;
  return fmaf(fast_log2(u), LN2, (q - (u - 1.0f)) * fast_rcp(u));
}

// The same for 0 <= q <= 1 (u in [1, 2]): 1 / u in the first-order term replaced by 2 - u, which is within 25 % of it;
// the term itself is at most half an ulp of u, so the result moves by less than a fifth of an ulp -- and a
// transcendental (four issue slots) becomes one subtraction.
__device__ __forceinline__ float log1p_unit(float q) {
  const float u = 1.0f + q;
  return fmaf(fast_log2(u), LN2, (q - (u - 1.0f)) * (2.0f - u));
}

// log(sigmoid(x)) = min(x, 0) - log1p(exp(-|x|))
__device__ __forceinline__ float log_sigmoid(float x) { return fminf(x, 0.0f) - log1p_unit(exp_acc(-fabsf(x))); }
// its derivative sigmoid(-x), in the two-sided form that keeps small values exact
__device__ __forceinline__ float log_sigmoid_grad(float x) {
  const float z = exp_acc(-fabsf(x));
  const float r = z * fast_rcp(1.0f + z);
  return x < 0.0f ? 1.0f - r : r;
}

struct LogisticArgs {
  const float* loc; const float* log_scale; const float* centers;
  long N; int K; int G; float half_width; float eps;
};

__global__ __launch_bounds__(LG_BLOCK) void k_logistic_logits(LogisticArgs a, float* __restrict__ out) {
  const int g = threadIdx.x % a.G, slot = threadIdx.x / a.G, per_block = LG_BLOCK / a.G;
  for (long n = (long)blockIdx.x * per_block + slot; n < a.N; n += (long)gridDim.x * per_block) {
    const float loc = a.loc[n];
    const float inv_scale = exp_acc(-(a.log_scale[n] - 2.0f));
    for (int k = g; k < a.K; k += a.G) {
      const float d = a.centers[k] - loc;
      const float lo = log_sigmoid(inv_scale * (d - a.half_width));
      const float hi = log_sigmoid(inv_scale * (d + a.half_width));
      // (a streamed output: nothing in this launch reads it back)
      __builtin_nontemporal_store(hi + log1p_acc(-exp_acc(lo - hi) + a.eps), out + (n * a.K + k));
    }
  }
}

__global__ __launch_bounds__(LG_BLOCK) void k_logistic_logits_backward(LogisticArgs a, const float* __restrict__ dlogits,
                                                                       float* __restrict__ dloc,
                                                                       float* __restrict__ dlog_scale) {
  const int g = threadIdx.x % a.G, slot = threadIdx.x / a.G, per_block = LG_BLOCK / a.G;
  // every lane of a wave runs the same number of trips, so the shuffles below see all their partners
  const long trips = (a.N + (long)gridDim.x * per_block - 1) / ((long)gridDim.x * per_block);
  for (long it = 0; it < trips; ++it) {
    const long n = (it * gridDim.x + blockIdx.x) * per_block + slot;
    const bool live = n < a.N;
    const long nn = live ? n : 0;
    const float loc = a.loc[nn];
    const float inv_scale = exp_acc(-(a.log_scale[nn] - 2.0f));
    float g_d = 0.0f, g_s = 0.0f;        // d loss / d (centre - loc) summed over bins, d loss / d inv_scale
    for (int k = g; k < a.K; k += a.G) {
      const float go = live ? dlogits[nn * a.K + k] : 0.0f;
      const float d = a.centers[k] - loc;
      const float u_lo = inv_scale * (d - a.half_width), u_hi = inv_scale * (d + a.half_width);
      const float lo = log_sigmoid(u_lo), hi = log_sigmoid(u_hi);
      // y = hi + log1p(q), q = -exp(lo - hi) + eps
      const float e = exp_acc(lo - hi);
      const float gq = go * fast_rcp(1.0f + (-e + a.eps));
      const float g_lo = -gq * e, g_hi = go - g_lo;
      const float gu_lo = g_lo * log_sigmoid_grad(u_lo), gu_hi = g_hi * log_sigmoid_grad(u_hi);
      g_d += inv_scale * (gu_lo + gu_hi);
      g_s += gu_lo * (d - a.half_width) + gu_hi * (d + a.half_width);
    }
    for (int s = a.G >> 1; s > 0; s >>= 1) {
      g_d += __shfl_xor(g_d, s, 64);
      g_s += __shfl_xor(g_s, s, 64);
    }
    if (g == 0 && live) {
      dloc[n] = -g_d;                         // d (centre - loc) / d loc = -1
      dlog_scale[n] = -inv_scale * g_s;       // d exp(-(ls - 2)) / d ls = -inv_scale
    }
  }
}

static int logistic_args(const float* loc, const float* log_scale, const float* centers, long N, int K, float eps,
                         LogisticArgs* a, int* blocks) {
  D3PM_CHECK_ARG(K >= 2 && N >= 0, "logistic head: K = %d, N = %ld", K, N);
  D3PM_CHECK_ARG(N == 0 || (loc && log_scale && centers), "logistic head: null input");
  int G = 2;
  while (G < K && G < 64) G <<= 1;
  const double width = 2.0 / ((double)K - 1.0);
  *a = LogisticArgs{loc, log_scale, centers, N, K, G, (float)(0.5 * width), eps};
  const long per_block = LG_BLOCK / G;
  const long want = (N + per_block - 1) / per_block;
  *blocks = (int)(want < 1 ? 1 : (want > 8192 ? 8192 : want));
  return D3PM_OK;
}

}  // namespace d3pm

using namespace d3pm;

extern "C" int d3pm_logistic_logits(const float* loc, const float* log_scale, const float* centers, long N, int K,
                                    float eps, float* logits, void* stream) {
  LogisticArgs a;
  int blocks;
  if (int rc = logistic_args(loc, log_scale, centers, N, K, eps, &a, &blocks)) return rc;
  D3PM_CHECK_ARG(logits != nullptr || N == 0, "d3pm_logistic_logits: null output");
  if (N == 0) return D3PM_OK;
  hipLaunchKernelGGL(k_logistic_logits, dim3(blocks), dim3(LG_BLOCK), 0, (hipStream_t)stream, a, logits);
  D3PM_LAUNCH_CHECK("k_logistic_logits");
  return D3PM_OK;
}

extern "C" int d3pm_logistic_logits_backward(const float* loc, const float* log_scale, const float* centers,
                                             const float* dlogits, long N, int K, float eps, float* dloc,
                                             float* dlog_scale, void* stream) {
  LogisticArgs a;
  int blocks;
  if (int rc = logistic_args(loc, log_scale, centers, N, K, eps, &a, &blocks)) return rc;
  D3PM_CHECK_ARG((dlogits && dloc && dlog_scale) || N == 0, "d3pm_logistic_logits_backward: null pointer");
  if (N == 0) return D3PM_OK;
  hipLaunchKernelGGL(k_logistic_logits_backward, dim3(blocks), dim3(LG_BLOCK), 0, (hipStream_t)stream, a, dlogits, dloc,
                     dlog_scale);
  D3PM_LAUNCH_CHECK("k_logistic_logits_backward");
  return D3PM_OK;
}
