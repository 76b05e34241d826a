// Transition-table kernels: one-step matrices, cumulative products, the
// column-structured digests and float32 copies the hot kernels read.
//
// The cumulative product Qbar_t = Qbar_{t-1} Q_t is sequential in t but its
// rows are independent (row i of Qbar_t is row i of Qbar_{t-1} times Q_t), so
// one workgroup carries one row through all T steps as a vector-matrix chain
// with the vector in LDS; all workgroups read the same Q_t at about the same
// time, so it is served from L2.
#include "d3pm_device.h"

#include <cstdarg>
#include <cstdio>

namespace d3pm {

static thread_local char g_err[512] = "";

void set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

static thread_local size_t g_workspace_need = 0;
void note_workspace_need(size_t bytes) { g_workspace_need = bytes; }
size_t take_workspace_need() {
  const size_t n = g_workspace_need;
  g_workspace_need = 0;
  return n;
}

int hip_fail(hipError_t e, const char* what) {
  set_error("%s: %s", what, hipGetErrorString(e));
  return D3PM_EHIP;
}

// ------------------------------------------------------------ one-step, image
// One workgroup per timestep.  Gaussian weights follow the reference
// literally: grid values linspace(0,255,K)*2/(K-1) (the 255 is not K-1),
// softmax over the mirrored vector, and off-diagonal k takes the weight of
// distance k-1 (d3pm/diffusion_discrete.py:166-181).
__global__ __launch_bounds__(256) void k_onestep_image(int kind, int K, int bands, const double* __restrict__ betas,
                                                       double* __restrict__ out) {
  extern __shared__ double sh[];   // gaussian: w[0..nb+1], prefix[0..nb+1]
  const int t = blockIdx.x;
  const double beta = betas[t];
  double* q = out + (size_t)t * K * K;
  const int nb = bands > 0 ? bands : K - 1;   // `if self.transition_bands` is false for 0 as well
  if (kind == D3PM_KIND_GAUSSIAN) {
    double* w = sh;
    double* pre = sh + (nb + 2);
    __shared__ double zsum;
    // val_k = -(v_k^2)/beta, v_k = k*(255/(K-1))*2/(K-1); max over the vector is val_0 = 0
    if (threadIdx.x == 0) {
      double z = 0;
      for (int k = 0; k <= nb; ++k) {
        double v = (255.0 / (K - 1.0)) * k;     // linspace(0,255,K)[k]
        v = v * 2.0 / (K - 1.0);
        z += exp(-v * v / beta);
      }
      zsum = 2.0 * z;
    }
    __syncthreads();
    for (int k = threadIdx.x; k <= nb + 1; k += blockDim.x) {
      if (k == 0) { w[0] = 0; continue; }
      double v = (255.0 / (K - 1.0)) * (k - 1);
      v = v * 2.0 / (K - 1.0);
      w[k] = exp(-v * v / beta) / zsum;         // weight of off-diagonal k
    }
    __syncthreads();
    if (threadIdx.x == 0) {
      double s = 0;
      pre[0] = 0;
      for (int k = 1; k <= nb; ++k) { s += w[k]; pre[k] = s; }
    }
    __syncthreads();
    for (int e = threadIdx.x; e < K * K; e += blockDim.x) {
      const int i = e / K, j = e - i * K;
      const int d = i > j ? i - j : j - i;
      double v;
      if (d == 0) {
        const int up = min(nb, K - 1 - i), dn = min(nb, i);
        v = 1.0 - (pre[up] + pre[dn]);
      } else {
        v = d <= nb ? w[d] : 0.0;
      }
      q[e] = v;
    }
    return;
  }
  for (int e = threadIdx.x; e < K * K; e += blockDim.x) {
    const int i = e / K, j = e - i * K;
    double v;
    if (kind == D3PM_KIND_UNIFORM) {
      if (bands < 0) {
        v = (i == j) ? 1.0 - beta * (K - 1.0) / K : beta / (double)K;
      } else {
        const int d = i > j ? i - j : j - i;
        if (d == 0) {
          const int cnt = min(bands, K - 1 - i) + min(bands, i);
          double s = 0;
          for (int c = 0; c < cnt; ++c) s += beta / (double)K;
          v = 1.0 - s;
        } else {
          v = d <= bands ? beta / (double)K : 0.0;
        }
      }
    } else {  // absorbing: diag(1-beta) with beta added to column K/2
      v = (i == j ? 1.0 - beta : 0.0) + (j == K / 2 ? beta : 0.0);
    }
    q[e] = v;
  }
}

// ------------------------------------------------------------ one-step, protein
__global__ void k_onestep_protein(int K, const float* __restrict__ off, const float* __restrict__ norm,
                                  double* __restrict__ out) {
  const int t = blockIdx.x;
  const double o = (double)off[t], n = (double)norm[t];
  double* q = out + (size_t)t * K * K;
  for (int e = threadIdx.x; e < K * K; e += blockDim.x) {
    const int i = e / K, j = e - i * K;
    q[e] = (i == j ? n : 0.0) + o;
  }
}

// ------------------------------------------------------------ cumulative product
// grid = K rows, block = K rounded up to 64 (<= 1024).  v lives in LDS, double
// buffered; thread j owns output column j.
__global__ __launch_bounds__(1024) void k_chain(int K, int T, const double* __restrict__ mats, long st, long sr, long sc,
                                                double* __restrict__ out) {
  extern __shared__ double sh[];
  double* v0 = sh;
  double* v1 = sh + K;
  const int i = blockIdx.x, j = threadIdx.x;
  if (j < K) {
    const double x = mats[(long)i * sr + (long)j * sc];
    v0[j] = x;
    out[(size_t)i * K + j] = x;
  }
  __syncthreads();
  for (int t = 1; t < T; ++t) {
    const double* m = mats + (long)t * st;
    double acc = 0;
    if (j < K) {
#pragma unroll 8
      for (int k = 0; k < K; ++k) acc = fma(v0[k], m[(long)k * sr + (long)j * sc], acc);
      v1[j] = acc;
      out[((size_t)t * K + i) * K + j] = acc;
    }
    __syncthreads();
    double* tmp = v0; v0 = v1; v1 = tmp;
  }
}

// ------------------------------------------------------------ digests
template <typename OutT>
__global__ __launch_bounds__(256) void k_digest(const double* __restrict__ table, long st, long sr, long sc, int K,
                                                double tol_or_eps, OutT* __restrict__ digest, int* violations) {
  const int t = blockIdx.x;
  const double* m = table + (long)t * st;
  constexpr bool F32 = sizeof(OutT) == 4;
  // diag, off, log(diag [+eps]), log(off [+eps]); the float32 digest carries four more rows for sampling from a row of
  // the table with weights p + eps (what q_sample does): w = off + eps, its inclusive prefix sums, the probability of
  // staying in class j when starting there as a 32-bit integer threshold, and a spare row
  constexpr int NV = F32 ? 8 : 4;
  OutT* dg = digest + (size_t)t * NV * K;
  int bad = 0;
  for (int e = threadIdx.x; e < K * K; e += blockDim.x) {
    const int i = e / K, j = e - i * K;
    const double x = m[(long)i * sr + (long)j * sc];
    const int irep = (j == 0) ? 1 : 0;            // representative off-diagonal row of column j
    const double rep = m[(long)irep * sr + (long)j * sc];
    if (i == j) {
      // a diagonal entry below the column's off-diagonal value also counts as a break: the one-uniform samplers
      // take the diagonal's surplus over the off-diagonal value as a non-negative extra mass of that class.  (Below
      // by more than rounding: where a cumulative product has mixed completely the two are equal up to the last
      // bits of independent float64 dot products, and a surplus of -1e-6 of the class weight moves no draw.)
      if (x < rep * (1.0 - 1e-6)) ++bad;
      if constexpr (F32) {
        const float d = (float)x;
        dg[0 * K + j] = d;
        dg[2 * K + j] = (float)log((double)(d + (float)tol_or_eps));
      } else {
        dg[0 * K + j] = x;
        dg[2 * K + j] = log(x);          // -inf for a zero entry: readers mask on the probability
      }
    } else {
      if constexpr (F32) {
        if ((float)x != (float)rep) ++bad;
      } else {
        if (fabs(x - rep) > tol_or_eps * fmax(fabs(x), fabs(rep))) ++bad;
      }
      if (i == irep) {
        if constexpr (F32) {
          const float o = (float)x;
          dg[1 * K + j] = o;
          dg[3 * K + j] = (float)log((double)(o + (float)tol_or_eps));
        } else {
          dg[1 * K + j] = x;
          dg[3 * K + j] = log(x);
        }
      }
    }
  }
  if (bad) atomicAdd(violations, bad);
  if constexpr (F32) {
    __syncthreads();                    // the rows above are this workgroup's own writes
    if (threadIdx.x == 0) {
      const float eps = (float)tol_or_eps;
      float run = 0;
      for (int j = 0; j < K; ++j) {     // float32 prefix sums in class order: the inverse CDF the samplers search
        const float w = dg[1 * K + j] + eps;
        run += w;
        dg[4 * K + j] = w;
        dg[5 * K + j] = run;
      }
      const double total = (double)run;
      for (int j = 0; j < K; ++j) {
        const double d = (double)(dg[0 * K + j] + eps), w = (double)dg[4 * K + j];
        const double stay = d / (total - w + d);             // in float64: a probability of 1 - 1e-5 keeps its digits
        const unsigned bits = stay >= 1.0 ? 0xffffffffu : (unsigned)(stay * 4294967296.0);
        dg[6 * K + j] = __builtin_bit_cast(float, bits);
        dg[7 * K + j] = 0.0f;
      }
    }
  }
}

__global__ void k_cast_f32(const double* __restrict__ table, long st, long sr, long sc, int K, float* __restrict__ out) {
  const int t = blockIdx.y;
  const double* m = table + (long)t * st;
  for (int e = blockIdx.x * blockDim.x + threadIdx.x; e < K * K; e += gridDim.x * blockDim.x) {
    const int i = e / K, j = e - i * K;
    out[(size_t)t * K * K + e] = (float)m[(long)i * sr + (long)j * sc];
  }
}

// float32 [T][K][K] (row = reduction index k, column = output class m) -> the operand image of the float16 matrix
// instructions, [T][K/32][K/16][2][64][8] float16: reduction block u, output-class tile mt, piece p (hi, lo), lane,
// slot.  Lane l = (kg << 4) | mm holds, for output class 16 mt + mm, the eight reduction indices
// k = 32 u + 16 (s >> 2) + 4 kg + (s & 3), s = 0..7 -- the A operand of v_mfma_f32_16x16x32_f16 (row = lane & 15,
// reduction group = lane >> 4) under a reduction order in which a lane's eight indices are two runs of four
// consecutive classes 16 apart: the order the accumulator tiles of a previous 16x16 product hold their rows in, so
// that a result can be the next product's operand without leaving its registers.  The pieces are those of 2^14 x
// (entries are probabilities: at most 1): hi = float16(2^14 x), lo = float16(2^14 x - hi), round to nearest.
// One thread per (u, mt, lane): eight strided reads, two 16-byte writes; consecutive threads write consecutive
// 16-byte pieces.
__global__ __launch_bounds__(256) void k_split_f16x2(const float* __restrict__ table, int K, uint16_t* __restrict__ out) {
  const int t = blockIdx.y;
  const int e = blockIdx.x * blockDim.x + threadIdx.x;       // (u * (K / 16) + mt) * 64 + lane
  if (e >= (K / 32) * (K / 16) * 64) return;
  const int lane = e & 63, um = e >> 6;
  const int mt = um % (K / 16), u = um / (K / 16);
  const int mm = lane & 15, kg = lane >> 4;
  const float* m = table + (size_t)t * K * K + 16 * mt + mm;
  uint16_t h[8], lo[8];
#pragma unroll
  for (int s = 0; s < 8; ++s) {
    const int k = 32 * u + 16 * (s >> 2) + 4 * kg + (s & 3);
    const float x = m[(size_t)k * K] * 16384.0f;            // entries are probabilities: 2^14 x fits float16
    const _Float16 xh = (_Float16)x;
    const _Float16 xl = (_Float16)(x - (float)xh);
    h[s] = __builtin_bit_cast(uint16_t, xh); lo[s] = __builtin_bit_cast(uint16_t, xl);
  }
  uint16_t* o = out + (size_t)t * 2 * K * K + ((size_t)um * 2 * 64 + lane) * 8;
#pragma unroll
  for (int s = 0; s < 8; ++s) { o[s] = h[s]; o[64 * 8 + s] = lo[s]; }
}

// 64-bit content fingerprint of a strided [T,K,K] float64 table: the sum, modulo 2^64, of every element's bit
// pattern times an odd multiplier derived from its index (splitmix64 of the index).  Integer addition is
// associative, so the order in which workgroups add their partial sums does not matter.
__global__ __launch_bounds__(256) void k_fingerprint(const double* __restrict__ table, long st, long sr, long sc, int K,
                                                     unsigned long long* __restrict__ out) {
  const int t = blockIdx.x;
  const double* m = table + (long)t * st;
  unsigned long long acc = 0;
  for (int e = threadIdx.x; e < K * K; e += blockDim.x) {
    const int i = e / K, j = e - i * K;
    unsigned long long z = (unsigned long long)t * (unsigned long long)(K * K) + e + 0x9E3779B97F4A7C15ull;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    z ^= z >> 31;
    acc += __builtin_bit_cast(unsigned long long, m[(long)i * sr + (long)j * sc]) * (z | 1ull);
  }
  __shared__ unsigned long long red[256];
  red[threadIdx.x] = acc;
  __syncthreads();
  for (int w = 128; w > 0; w >>= 1) {
    if ((int)threadIdx.x < w) red[threadIdx.x] += red[threadIdx.x + w];
    __syncthreads();
  }
  if (threadIdx.x == 0) atomicAdd(out, red[0]);
}

static int launch_chain(int K, int T, const double* mats, long st, long sr, long sc, double* out, hipStream_t s) {
  D3PM_CHECK_ARG(K >= 1 && K <= 1024, "cumulative product supports 1 <= K <= 1024, got %d", K);
  const int block = ((K + 63) / 64) * 64;
  hipLaunchKernelGGL(k_chain, dim3(K), dim3(block), 2 * K * sizeof(double), s, K, T, mats, st, sr, sc, out);
  D3PM_LAUNCH_CHECK("k_chain");
  return D3PM_OK;
}

}  // namespace d3pm

using namespace d3pm;

extern "C" const char* d3pm_last_error(void) { return g_err; }
extern "C" int d3pm_abi_version(void) { return 10; }

extern "C" int d3pm_build_tables_f64(int kind, int K, int T, int bands, const double* betas, double* q_onestep,
                                     double* q_mats, void* stream) {
  D3PM_CHECK_ARG(kind == D3PM_KIND_UNIFORM || kind == D3PM_KIND_GAUSSIAN || kind == D3PM_KIND_ABSORBING,
                 "transition_mat_type must be 'gaussian', 'uniform', 'absorbing' (kind=%d)", kind);
  D3PM_CHECK_ARG(K >= 2 && K <= 1024 && T >= 1, "need 2 <= K <= 1024 and T >= 1 (K=%d, T=%d)", K, T);
  D3PM_CHECK_ARG(betas && q_onestep && q_mats, "null pointer");
  D3PM_CHECK_ARG(bands < K, "transition_bands must be < num_pixel_vals");
  hipStream_t s = (hipStream_t)stream;
  const int nb = bands > 0 ? bands : K - 1;
  const size_t sh = kind == D3PM_KIND_GAUSSIAN ? 2 * (size_t)(nb + 2) * sizeof(double) : 0;
  hipLaunchKernelGGL(k_onestep_image, dim3(T), dim3(256), sh, s, kind, K, bands, betas, q_onestep);
  D3PM_LAUNCH_CHECK("k_onestep_image");
  return launch_chain(K, T, q_onestep, (long)K * K, K, 1, q_mats, s);
}

extern "C" int d3pm_protein_uniform_tables_f64(int K, int T, const float* off, const float* norm, double* q_onestep,
                                               double* q_bar, void* stream) {
  D3PM_CHECK_ARG(K >= 2 && K <= 1024 && T >= 1, "need 2 <= K <= 1024 and T >= 1 (K=%d, T=%d)", K, T);
  D3PM_CHECK_ARG(off && norm && q_onestep && q_bar, "null pointer");
  hipStream_t s = (hipStream_t)stream;
  hipLaunchKernelGGL(k_onestep_protein, dim3(T), dim3(256), 0, s, K, off, norm, q_onestep);
  D3PM_LAUNCH_CHECK("k_onestep_protein");
  return launch_chain(K, T, q_onestep, (long)K * K, K, 1, q_bar, s);
}

extern "C" int d3pm_cumprod_f64(int K, int T, const double* mats, long st, long sr, long sc, double* out,
                                void* stream) {
  D3PM_CHECK_ARG(mats && out && T >= 1, "null pointer or T < 1");
  return launch_chain(K, T, mats, st, sr, sc, out, (hipStream_t)stream);
}

extern "C" int d3pm_digest_f32(const double* table, long st, long sr, long sc, int T, int K, float eps, float* digest,
                               int* violations, void* stream) {
  D3PM_CHECK_ARG(table && digest && violations, "null pointer");
  D3PM_CHECK_ARG(K >= 2 && T >= 1, "need K >= 2 and T >= 1");
  warm_static_scratch();     // a set-up call, outside any capture: d3pm_hybrid_loss_static finds its array resolved
  hipLaunchKernelGGL(k_digest<float>, dim3(T), dim3(256), 0, (hipStream_t)stream, table, st, sr, sc, K, (double)eps,
                     digest, violations);
  D3PM_LAUNCH_CHECK("k_digest<float>");
  return D3PM_OK;
}

extern "C" int d3pm_digest_f64(const double* table, long st, long sr, long sc, int T, int K, double rtol,
                               double* digest, int* violations, void* stream) {
  D3PM_CHECK_ARG(table && digest && violations, "null pointer");
  D3PM_CHECK_ARG(K >= 2 && T >= 1, "need K >= 2 and T >= 1");
  hipLaunchKernelGGL(k_digest<double>, dim3(T), dim3(256), 0, (hipStream_t)stream, table, st, sr, sc, K, rtol, digest,
                     violations);
  D3PM_LAUNCH_CHECK("k_digest<double>");
  return D3PM_OK;
}

extern "C" int d3pm_table_fingerprint(const double* table, long st, long sr, long sc, int T, int K, uint64_t* out,
                                      void* stream) {
  D3PM_CHECK_ARG(table && out && K >= 1 && T >= 1, "null pointer or empty table");
  hipLaunchKernelGGL(k_fingerprint, dim3(T), dim3(256), 0, (hipStream_t)stream, table, st, sr, sc, K,
                     (unsigned long long*)out);
  D3PM_LAUNCH_CHECK("k_fingerprint");
  return D3PM_OK;
}

extern "C" int d3pm_table_to_f32(const double* table, long st, long sr, long sc, int T, int K, float* out,
                                 void* stream) {
  D3PM_CHECK_ARG(table && out && K >= 1 && T >= 1, "null pointer or empty table");
  const int nb = min(64, (K * K + 255) / 256);
  hipLaunchKernelGGL(k_cast_f32, dim3(nb, T), dim3(256), 0, (hipStream_t)stream, table, st, sr, sc, K, out);
  D3PM_LAUNCH_CHECK("k_cast_f32");
  return D3PM_OK;
}

extern "C" int d3pm_table_split_f16x2(const float* table, int T, int K, uint16_t* out, void* stream) {
  D3PM_CHECK_ARG(table && out && T >= 1, "null pointer or empty table");
  D3PM_CHECK_ARG(K >= 32 && K % 32 == 0, "the operand image needs K to be a multiple of 32, got %d", K);
  const int n = (K / 32) * (K / 16) * 64;
  hipLaunchKernelGGL(k_split_f16x2, dim3((n + 255) / 256, T), dim3(256), 0, (hipStream_t)stream, table, K, out);
  D3PM_LAUNCH_CHECK("k_split_f16x2");
  return D3PM_OK;
}
