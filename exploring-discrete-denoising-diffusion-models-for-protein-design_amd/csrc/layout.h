// Lane layouts for [N, K] row-major class tensors and the host-side choice
// between them.  See d3pm_device.h for the convention.
#pragma once
#include "d3pm_device.h"

#include <cstdlib>

namespace d3pm {

// QUAD (vector layouts only): the lane's 16-byte pieces are interleaved across the
// group, piece h of lane g being classes 4 (g + G h) .. +3, so that one access
// instruction of the group covers 16 G contiguous bytes (fully coalesced in
// global memory and conflict-free in LDS) however many classes a lane holds.
template <int G_, int V_, bool VEC_, bool QUAD_ = false>
struct Layout {
  static constexpr int G = G_;        // lanes per token
  static constexpr int V = V_;        // class slots per lane
  static constexpr bool VEC = VEC_;   // slots in 16-byte pieces
  static constexpr bool QUAD = QUAD_;
  static constexpr int TOKENS_PER_WAVE = 64 / G_;
  static_assert(!VEC_ || V_ % 4 == 0, "vector layouts hold whole 16-byte pieces");
  static_assert(!QUAD_ || VEC_, "interleaved pieces are a vector layout");
  // first class of the lane's 16-byte piece h
  __device__ static __forceinline__ int piece(int g, int h) { return QUAD_ ? 4 * (g + G_ * h) : V_ * g + 4 * h; }
  __device__ static __forceinline__ int cls(int g, int v) { return VEC_ ? piece(g, v >> 2) + (v & 3) : g + G_ * v; }
};

// Load one token's slots of a float row; masked slots read `fill`.
template <class L>
__device__ __forceinline__ void load_slots(const float* __restrict__ row, int g, int K, float fill, float (&z)[L::V]) {
  if constexpr (L::VEC) {
#pragma unroll
    for (int h = 0; h < L::V / 4; ++h) {
      const float4 q = *reinterpret_cast<const float4*>(row + L::piece(g, h));
      z[4 * h + 0] = q.x; z[4 * h + 1] = q.y; z[4 * h + 2] = q.z; z[4 * h + 3] = q.w;
    }
  } else {
#pragma unroll
    for (int v = 0; v < L::V; ++v) {
      const int c = L::cls(g, v);
      z[v] = c < K ? row[c] : fill;
    }
  }
}
template <class L>
__device__ __forceinline__ void load_slots(const double* __restrict__ row, int g, int K, double fill, double (&z)[L::V]) {
#pragma unroll
  for (int v = 0; v < L::V; ++v) {
    const int c = L::cls(g, v);
    z[v] = c < K ? row[c] : fill;
  }
}

// The same through a wave-uniform base and a 32-bit byte offset of the row, so
// the access is one scalar-base + vector-offset instruction with no 64-bit
// address arithmetic per lane.
template <class L>
__device__ __forceinline__ void load_slots_off(const char* __restrict__ base, unsigned row_off, int g, int K,
                                               float fill, float (&z)[L::V]) {
  if constexpr (L::VEC) {
#pragma unroll
    for (int h = 0; h < L::V / 4; ++h) {
      const float4 q = *reinterpret_cast<const float4*>(base + (row_off + 4u * (unsigned)L::piece(g, h)));
      z[4 * h + 0] = q.x; z[4 * h + 1] = q.y; z[4 * h + 2] = q.z; z[4 * h + 3] = q.w;
    }
  } else {
#pragma unroll
    for (int v = 0; v < L::V; ++v) {
      const int c = L::cls(g, v);
      z[v] = c < K ? *reinterpret_cast<const float*>(base + (row_off + 4u * (unsigned)c)) : fill;
    }
  }
}
template <class L>
__device__ __forceinline__ void store_slots_off(char* __restrict__ base, unsigned row_off, int g, int K,
                                                const float (&z)[L::V]) {
  if constexpr (L::VEC) {
#pragma unroll
    for (int h = 0; h < L::V / 4; ++h)
      *reinterpret_cast<float4*>(base + (row_off + 4u * (unsigned)L::piece(g, h))) =
          make_float4(z[4 * h + 0], z[4 * h + 1], z[4 * h + 2], z[4 * h + 3]);
  } else {
#pragma unroll
    for (int v = 0; v < L::V; ++v) {
      const int c = L::cls(g, v);
      if (c < K) *reinterpret_cast<float*>(base + (row_off + 4u * (unsigned)c)) = z[v];
    }
  }
}

// The same with the non-temporal cache policy: a streamed output that no workgroup of the launch reads again
// (dirty lines left in the XCDs' L2 are written back only when the kernel ends; streamed out, that traffic
// overlaps the kernel's own run time).
typedef float v4f_nt __attribute__((ext_vector_type(4)));
typedef float f2v __attribute__((ext_vector_type(2)));      // a pair of classes: arithmetic on it maps onto v_pk_*_f32
template <class L>
__device__ __forceinline__ void store_slots_off_nt(char* __restrict__ base, unsigned row_off, int g, int K,
                                                   const float (&z)[L::V]) {
  if constexpr (L::VEC) {
#pragma unroll
    for (int h = 0; h < L::V / 4; ++h)
      __builtin_nontemporal_store(v4f_nt{z[4 * h + 0], z[4 * h + 1], z[4 * h + 2], z[4 * h + 3]},
                                  reinterpret_cast<v4f_nt*>(base + (row_off + 4u * (unsigned)L::piece(g, h))));
  } else {
#pragma unroll
    for (int v = 0; v < L::V; ++v) {
      const int c = L::cls(g, v);
      if (c < K) __builtin_nontemporal_store(z[v], reinterpret_cast<float*>(base + (row_off + 4u * (unsigned)c)));
    }
  }
}

template <class L>
__device__ __forceinline__ void store_slots(float* __restrict__ row, int g, int K, const float (&z)[L::V]) {
  if constexpr (L::VEC) {
#pragma unroll
    for (int h = 0; h < L::V / 4; ++h)
      *reinterpret_cast<float4*>(row + L::piece(g, h)) =
          make_float4(z[4 * h + 0], z[4 * h + 1], z[4 * h + 2], z[4 * h + 3]);
  } else {
#pragma unroll
    for (int v = 0; v < L::V; ++v) {
      const int c = L::cls(g, v);
      if (c < K) row[c] = z[v];
    }
  }
}

// Zero the padding columns [K, ld) of a row (rows wider than K only occur in
// the strided layout).
template <class L>
__device__ __forceinline__ void zero_tail(float* __restrict__ row, int g, int K, long ld) {
  for (long c = K + g; c < ld; c += L::G) row[c] = 0.0f;
}

inline bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; }

// Calls f(Layout<...>{}) for the layout that fits K.  `vec_ok` says whether
// every float row the kernel touches is 16-byte aligned with stride K.
template <class F>
int dispatch_layout(int K, bool vec_ok, F&& f) {
  if (vec_ok) {
    switch (K) {
      case 16: return f(Layout<4, 4, true>{});
      case 32: return f(Layout<8, 4, true>{});
      case 64: return f(Layout<16, 4, true>{});
      case 128: return f(Layout<32, 4, true>{});
      case 256: return f(Layout<64, 4, true>{});
      default: break;
    }
  }
  if (K <= 8) return f(Layout<2, 4, false>{});
  if (K <= 24) return f(Layout<8, 3, false>{});
  if (K <= 32) return f(Layout<8, 4, false>{});
  if (K <= 64) return f(Layout<16, 4, false>{});
  if (K <= 128) return f(Layout<32, 4, false>{});
  if (K <= 256) return f(Layout<64, 4, false>{});
  set_error("K = %d is not supported (2 <= K <= 256)", K);
  return D3PM_EUNSUPPORTED;
}

// As dispatch_layout, but with 8 classes per lane where K allows: fewer lanes
// per token means fewer cross-lane reduction steps and less per-token work
// repeated in every lane.  For kernels whose per-class register footprint is
// small enough to afford it.
template <class F>
int dispatch_layout_wide(int K, bool vec_ok, F&& f) {
  if (vec_ok) {
    switch (K) {
      case 32: return f(Layout<4, 8, true>{});
      case 64: return f(Layout<8, 8, true>{});
      case 128: return f(Layout<16, 8, true>{});
      case 256: return f(Layout<32, 8, true>{});
      default: break;
    }
  }
  return dispatch_layout(K, vec_ok, f);
}

// Grid for kernels that walk [B, npos] tokens with one batch element per
// blockIdx.y: enough workgroups to fill 256 CUs several times over, whole
// passes of `tokens_per_pass` per workgroup.
struct TokenGrid {
  int chunks;        // gridDim.x
  long chunk_pos;    // positions per chunk (multiple of tokens_per_pass)
};
inline TokenGrid token_grid(int B, long npos, int tokens_per_pass, int target_blocks = 2048) {
  long passes = (npos + tokens_per_pass - 1) / tokens_per_pass;
  long want = (target_blocks + B - 1) / B;
  if (want < 1) want = 1;
  long chunks = passes < want ? passes : want;
  if (chunks < 1) chunks = 1;
  long ppc = (passes + chunks - 1) / chunks;
  chunks = (passes + ppc - 1) / ppc;
  if (chunks < 1) chunks = 1;
  return TokenGrid{(int)chunks, ppc * tokens_per_pass};
}

}  // namespace d3pm
