// Device-side helpers shared by the D3PM kernels (gfx950, wave64).
//
// Lane layout used everywhere: a token's K classes are spread over a group of
// G consecutive lanes (G a power of two, 64/G tokens per wave); lane g of the
// group owns V class slots.  In the vector layout (K == 4*G) slot v is class
// 4*g+v and a lane's slots are one 16-byte access; in the strided layout slot
// v is class g+G*v (any K <= G*V, slots past K masked).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <math.h>

#include "../../include/d3pm_hip.h"

namespace d3pm {

void set_error(const char* fmt, ...);
int hip_fail(hipError_t e, const char* what);

#define D3PM_CHECK_ARG(cond, ...)                    \
  do {                                               \
    if (!(cond)) {                                   \
      ::d3pm::set_error(__VA_ARGS__);                \
      return D3PM_EINVAL;                            \
    }                                                \
  } while (0)

// A caller-supplied workspace that is too small: the usual D3PM_EINVAL, and the number of bytes that were needed is
// noted for the calling thread (take_workspace_need() reads and clears it) -- entry points that bring a fixed
// scratch buffer of their own tell "does not fit" from every other argument error by that, not by message text.
void note_workspace_need(size_t bytes);
size_t take_workspace_need();
// Looks up the device address of the library's static scratch array on the current device if that has not
// happened yet (a set-up call outside any stream capture; failures are ignored here and surface at first use).
void warm_static_scratch();
#define D3PM_CHECK_WORKSPACE(ptr, have, need)                                                        \
  do {                                                                                               \
    if (!((ptr) && (size_t)(have) >= (size_t)(need))) {                                              \
      ::d3pm::note_workspace_need((size_t)(need));                                                   \
      ::d3pm::set_error("workspace too small: %zu < %zu bytes", (size_t)(have), (size_t)(need));     \
      return D3PM_EINVAL;                                                                            \
    }                                                                                                \
  } while (0)

#define D3PM_LAUNCH_CHECK(what)                                  \
  do {                                                           \
    hipError_t e__ = hipGetLastError();                          \
    if (e__ != hipSuccess) return ::d3pm::hip_fail(e__, what);   \
  } while (0)

// ---------------------------------------------------------------- DPP moves
template <int CTRL>
__device__ __forceinline__ float dpp_mov(float v) {
  return __builtin_bit_cast(
      float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xF, 0xF, true));
}
template <int CTRL>
__device__ __forceinline__ int dpp_mov(int v) {
  return __builtin_amdgcn_update_dpp(0, v, CTRL, 0xF, 0xF, true);
}
template <int CTRL>
__device__ __forceinline__ double dpp_mov(double v) {
  long long b = __builtin_bit_cast(long long, v);
  int lo = dpp_mov<CTRL>((int)(b & 0xffffffffll));
  int hi = dpp_mov<CTRL>((int)(b >> 32));
  return __builtin_bit_cast(double, ((long long)hi << 32) | (unsigned int)lo);
}

// Exchange partner of step S (1,2,4,8 inside a 16-lane row by DPP; 16 and 32
// through the LDS crossbar).  Steps 4 and 8 are mirrors, which pair lanes that
// already hold equal partial results of the lower steps, so for a commutative
// reduction they are as good as an xor butterfly.
template <int S, typename T>
__device__ __forceinline__ T partner(T v) {
  if constexpr (S == 1) return dpp_mov<0xB1>(v);         // quad_perm [1,0,3,2]
  else if constexpr (S == 2) return dpp_mov<0x4E>(v);    // quad_perm [2,3,0,1]
  else if constexpr (S == 4) return dpp_mov<0x141>(v);   // row_half_mirror
  else if constexpr (S == 8) return dpp_mov<0x140>(v);   // row_mirror
  else return __shfl_xor(v, S, 64);
}

template <int G, typename T>
__device__ __forceinline__ T group_sum(T v) {
  if constexpr (G >= 2) v += partner<1>(v);
  if constexpr (G >= 4) v += partner<2>(v);
  if constexpr (G >= 8) v += partner<4>(v);
  if constexpr (G >= 16) v += partner<8>(v);
  if constexpr (G >= 32) v += partner<16>(v);
  if constexpr (G >= 64) v += partner<32>(v);
  return v;
}
template <int G, typename T>
__device__ __forceinline__ T group_max(T v) {
  if constexpr (G >= 2) v = fmax(v, partner<1>(v));
  if constexpr (G >= 4) v = fmax(v, partner<2>(v));
  if constexpr (G >= 8) v = fmax(v, partner<4>(v));
  if constexpr (G >= 16) v = fmax(v, partner<8>(v));
  if constexpr (G >= 32) v = fmax(v, partner<16>(v));
  if constexpr (G >= 64) v = fmax(v, partner<32>(v));
  return v;
}

// argmax with first-index tie-break (torch.argmax on CPU returns the first
// maximal index).  Every lane of the group ends with the same (value, index).
template <int S, typename T>
__device__ __forceinline__ void argmax_step(T& v, int& i) {
  T ov = partner<S>(v);
  int oi = partner<S>(i);
  bool take = (ov > v) || (ov == v && oi < i);
  v = take ? ov : v;
  i = take ? oi : i;
}
template <int G, typename T>
__device__ __forceinline__ void group_argmax(T& v, int& i) {
  if constexpr (G >= 2) argmax_step<1>(v, i);
  if constexpr (G >= 4) argmax_step<2>(v, i);
  if constexpr (G >= 8) argmax_step<4>(v, i);
  if constexpr (G >= 16) argmax_step<8>(v, i);
  if constexpr (G >= 32) argmax_step<16>(v, i);
  if constexpr (G >= 64) argmax_step<32>(v, i);
}

template <int G>
__device__ __forceinline__ int group_min(int v) {
  if constexpr (G >= 2) v = min(v, partner<1>(v));
  if constexpr (G >= 4) v = min(v, partner<2>(v));
  if constexpr (G >= 8) v = min(v, partner<4>(v));
  if constexpr (G >= 16) v = min(v, partner<8>(v));
  if constexpr (G >= 32) v = min(v, partner<16>(v));
  if constexpr (G >= 64) v = min(v, partner<32>(v));
  return v;
}

// Inclusive prefix sum over the lanes of a group, in lane order (g = lane index in the group).
// Distances below 16 stay inside a DPP row (row_shr); 16 and 32 go through the LDS crossbar.
// (groups of up to 16 lanes never straddle a row; wider groups use the crossbar for every step)
template <int G, int D, typename T>
__device__ __forceinline__ T lane_up(T v) {
  if constexpr (G > 16) return __shfl_up(v, D, 64);
  else if constexpr (D == 1) return dpp_mov<0x111>(v);   // row_shr:1
  else if constexpr (D == 2) return dpp_mov<0x112>(v);   // row_shr:2
  else if constexpr (D == 4) return dpp_mov<0x114>(v);   // row_shr:4
  else return dpp_mov<0x118>(v);                         // row_shr:8
}
template <int G, int D, typename T>
__device__ __forceinline__ void scan_step(T& v, int g) {
  if constexpr (D < G) {
    const T up = lane_up<G, D>(v);
    if (g >= D) v += up;
  }
}
template <int G, typename T>
__device__ __forceinline__ T group_scan_inclusive(T v, int g) {
  scan_step<G, 1>(v, g);
  scan_step<G, 2>(v, g);
  scan_step<G, 4>(v, g);
  scan_step<G, 8>(v, g);
  scan_step<G, 16>(v, g);
  scan_step<G, 32>(v, g);
  return v;
}

// One draw from the categorical distribution with unnormalised weights w over a
// token's classes (w = 0 on masked slots), from ONE uniform u in [0,1) shared by
// the group: inverse CDF over the classes in (lane, slot) order.  V slots per
// lane; `cls(g, v)` maps a slot to its class.  Every lane returns the class.
template <int G, int V, typename T, typename ClsFn>
__device__ __forceinline__ int group_pick(const T (&w)[V], int g, T u, int fallback, ClsFn cls) {
  T lane_tot = 0;
#pragma unroll
  for (int v = 0; v < V; ++v) lane_tot += w[v];
  const T incl = group_scan_inclusive<G>(lane_tot, g);
  const T total = group_max<G>(incl);          // weights are non-negative: the last lane's prefix is the largest
  const T target = u * total;
  T run = incl - lane_tot;
  int key = 0x7fffffff;
#pragma unroll
  for (int v = 0; v < V; ++v) {
    run += w[v];
    if (key == 0x7fffffff && w[v] > 0 && target < run) key = g * V + v;
  }
  key = group_min<G>(key);
  return key == 0x7fffffff ? fallback : cls(key / V, key % V);
}

// Sum over the whole wave, result valid in every lane.
template <typename T>
__device__ __forceinline__ T wave_sum(T v) { return group_sum<64>(v); }

// Block-wide sum of one value per thread; result valid in thread 0.
// `scratch` holds one T per wave.
template <typename T>
__device__ __forceinline__ T block_sum(T v, T* scratch) {
  v = wave_sum(v);
  const int wave = threadIdx.x >> 6, nw = (blockDim.x + 63) >> 6;
  if ((threadIdx.x & 63) == 0) scratch[wave] = v;
  __syncthreads();
  T r = 0;
  if (threadIdx.x == 0)
    for (int w = 0; w < nw; ++w) r += scratch[w];
  __syncthreads();
  return r;
}

// ---------------------------------------------------------------- tokens
__device__ __forceinline__ int load_tok(const void* p, long idx, int bytes) {
  return bytes == 8 ? (int)((const int64_t*)p)[idx] : ((const int32_t*)p)[idx];
}
// Timesteps and class indices address tables, LDS rows and logits columns: values outside their range are brought
// to the nearest valid one (one v_med3 per value, no fault, no host synchronisation).  Results for such inputs are
// those of the clamped values.
__device__ __forceinline__ int clamp_index(int v, int n) { return max(0, min(v, n - 1)); }
__device__ __forceinline__ int load_index(const void* p, long idx, int bytes, int n) {
  return clamp_index(load_tok(p, idx, bytes), n);
}

// ---------------------------------------------------------------- Philox
struct U4 { uint32_t x, y, z, w; };
__device__ __forceinline__ U4 philox4x32_10(uint64_t counter, uint64_t stream_id, uint64_t seed) {
  uint32_t c0 = (uint32_t)counter, c1 = (uint32_t)(counter >> 32);
  uint32_t c2 = (uint32_t)stream_id, c3 = (uint32_t)(stream_id >> 32);
  uint32_t k0 = (uint32_t)seed, k1 = (uint32_t)(seed >> 32);
#pragma unroll
  for (int r = 0; r < 10; ++r) {
    // (one 32 x 32 -> 64 bit multiply per product: v_mad_u64_u32 gives both halves)
    const uint64_t p0 = (uint64_t)0xD2511F53u * c0, p1 = (uint64_t)0xCD9E8D57u * c2;
    uint32_t h0 = (uint32_t)(p0 >> 32), l0 = (uint32_t)p0;
    uint32_t h1 = (uint32_t)(p1 >> 32), l1 = (uint32_t)p1;
    uint32_t n0 = h1 ^ c1 ^ k0, n1 = l1, n2 = h0 ^ c3 ^ k1, n3 = l0;
    c0 = n0; c1 = n1; c2 = n2; c3 = n3;
    k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
  }
  return U4{c0, c1, c2, c3};
}
// Philox2x32-10 (Salmon et al., "Parallel random numbers: as easy as 1, 2, 3"): a 64-bit counter and a 32-bit key,
// one multiply per round -- half the work of the 4x32 form per call, for kernels that need one word per token.
struct U2 { uint32_t x, y; };
__device__ __forceinline__ U2 philox2x32_10(uint32_t c0, uint32_t c1, uint32_t key) {
#pragma unroll
  for (int r = 0; r < 10; ++r) {
    const uint64_t p = (uint64_t)0xD256D193u * c0;
    const uint32_t n0 = (uint32_t)(p >> 32) ^ key ^ c1;
    c1 = (uint32_t)p;
    c0 = n0;
    key += 0x9E3779B9u;
  }
  return U2{c0, c1};
}
// U[0,1) with 24 random bits, like torch.rand for float32.
__device__ __forceinline__ float u01(uint32_t r) { return (float)(r >> 8) * 5.9604644775390625e-08f; }
// Strictly inside (0, 1): 23 random bits at the centres of 2^-23-wide cells, [2^-24, 1 - 2^-24].  (24 bits plus a
// half would round to 1.0 for the topmost cell.)
__device__ __forceinline__ float u01_open(uint32_t r) {
  return fmaf((float)(r >> 9), 1.1920928955078125e-07f, 5.9604644775390625e-08f);
}
// U[0,1) with 53 random bits.
__device__ __forceinline__ double u01d_open(uint32_t hi, uint32_t lo) {
  uint64_t m = (((uint64_t)hi << 32) | lo) >> 11;
  return (double)m * 1.1102230246251565e-16;
}
// U(0,1] with 53 random bits.
__device__ __forceinline__ double u01d(uint32_t hi, uint32_t lo) {
  uint64_t m = (((uint64_t)hi << 32) | lo) >> 11;
  return ((double)m + 1.0) * 1.1102230246251565e-16;
}

// The reference's Gumbel transform (diffusion_discrete.py:283-284):
// clip to [float32 tiny, 1], then -log(-log(u)).
__device__ __forceinline__ float gumbel_from_uniform(float u) {
  u = fminf(fmaxf(u, 1.17549435e-38f), 1.0f);
  return -logf(-logf(u));
}

// The same transform on hardware log2 (v_log_f32), for noise drawn on the device,
// where no bit pattern has to match a host computation.
__device__ __forceinline__ float gumbel_from_uniform_fast(float u) {
  u = fmaxf(u, 1.17549435e-38f);
  const float ln2 = 0.693147180559945309417f;
  return -ln2 * __builtin_amdgcn_logf(-ln2 * __builtin_amdgcn_logf(u));
}

}  // namespace d3pm
