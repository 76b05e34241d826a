// Wave-private LDS image that turns coalesced rows into one row per lane and back.
//
// Kernels that give every lane a whole token (its K classes in registers: no cross-lane reduction, per-class
// constants wave-uniform) still have to move rows of LD floats between global memory and that layout.  A wave's
// 64 consecutive rows are contiguous in memory: they are fetched as PB-byte pieces in lane order (fully
// coalesced), written to LDS at [row][STRIDE], and read back one row per lane; the reverse for output rows.
// STRIDE / PB is odd, which spreads the lanes of every LDS access group over all banks.
#pragma once
#include "layout.h"

#include <type_traits>

namespace d3pm {

typedef float v2f_t __attribute__((ext_vector_type(2)));
typedef float v4f_t __attribute__((ext_vector_type(4)));

// LDS operations of one wave complete in order; this only stops the compiler from moving them across.
__device__ __forceinline__ void wave_lds_order() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

__device__ __forceinline__ float lane_broadcast(float v, int lane) {
  return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), lane));
}

// LD floats per row; 16-byte pieces when rows are 16-byte multiples (the caller checks the base alignment),
// 8-byte pieces otherwise (LD even).
template <int LD>
struct RowImage {
  static_assert(LD % 2 == 0 && LD >= 2 && LD <= 32, "even row length up to 32 floats");
  static constexpr int RB = 4 * LD;
  static constexpr int PB = (LD % 4 == 0) ? 16 : 8;
  static constexpr int PIECES = RB / PB;                                // per row = per lane
  static constexpr int STRIDE = ((PIECES % 2) == 1) ? RB : RB + PB;     // bytes between rows in LDS
  static constexpr int BYTES = 64 * STRIDE;                             // per wave
  using piece_t = typename std::conditional<PB == 16, v4f_t, v2f_t>::type;

  piece_t stage[PIECES];

  // Request the wave's rows [base, base + 64) of a [.., LD] float tensor starting at `rows` (row index 0 = rows);
  // rows past `last` (index of the last existing row, relative to base) re-read row `last`.
  __device__ __forceinline__ void request(const char* __restrict__ rows, unsigned base, unsigned last, int lane) {
#pragma unroll
    for (int h = 0; h < PIECES; ++h) {
      const unsigned q = h * 64 + lane, r = min(q / PIECES, last), j = q % PIECES;
      stage[h] = *reinterpret_cast<const piece_t*>(rows + ((base + r) * RB + j * PB));
    }
  }
  // The same for rows that are contiguous in memory over the whole wave (`first` = first byte of the wave's first
  // row): piece h of a lane is byte (64 h + lane) PB of that block, whatever the row length -- no division.  Pieces
  // past `limit` (byte offset of the last piece that exists) re-read that piece; `full` (uniform over the wave) says
  // that every piece exists.
  __device__ __forceinline__ void request_flat(const char* __restrict__ first, unsigned limit, int lane, bool full) {
    if (full) {
#pragma unroll
      for (int h = 0; h < PIECES; ++h)
        stage[h] = *reinterpret_cast<const piece_t*>(first + (unsigned)(h * 64 + lane) * PB);
    } else {
#pragma unroll
      for (int h = 0; h < PIECES; ++h)
        stage[h] = *reinterpret_cast<const piece_t*>(first + min((unsigned)(h * 64 + lane) * PB, limit));
    }
  }
  // Where piece 64 h + lane of the wave's block lies in the image: byte (64 h + lane) PB plus the padding of the rows
  // before it, (64 h + lane) / PIECES of them.  With 64 h = QH PIECES + RH that quotient is lane / PIECES + QH, plus
  // one where lane % PIECES + RH carries: the lane keeps its own byte offset with and without the carry for each
  // distinct RH (equal expressions: one register each), and everything that depends on h is a constant.
  struct LaneSplit { unsigned row, col, at; };
  __device__ static __forceinline__ LaneSplit split(int lane) {
    const unsigned row = (unsigned)lane / PIECES, col = (unsigned)lane % PIECES;
    return LaneSplit{row, col, (unsigned)lane * PB + row * (STRIDE - RB)};
  }
  __device__ static __forceinline__ unsigned piece_at(LaneSplit ls, int h) {
    const unsigned QH = (64u * h) / PIECES, RH = (64u * h) % PIECES;     // constants once unrolled
    unsigned at = ls.at + 64u * h * PB + QH * (STRIDE - RB);
    if (RH != 0) at += (ls.col >= (unsigned)PIECES - RH) ? (unsigned)(STRIDE - RB) : 0u;
    return at;
  }
  __device__ __forceinline__ void deposit_flat(char* img, int lane, LaneSplit ls) const {
#pragma unroll
    for (int h = 0; h < PIECES; ++h) {
      if constexpr (STRIDE == RB) *reinterpret_cast<piece_t*>(img + (unsigned)(h * 64 + lane) * PB) = stage[h];
      else *reinterpret_cast<piece_t*>(img + piece_at(ls, h)) = stage[h];
    }
  }
  // The requested pieces into the image.
  __device__ __forceinline__ void deposit(char* img, int lane) const {
#pragma unroll
    for (int h = 0; h < PIECES; ++h) {
      const unsigned q = h * 64 + lane;
      *reinterpret_cast<piece_t*>(img + (q / PIECES) * STRIDE + (q % PIECES) * PB) = stage[h];
    }
  }
  // This lane's row out of / into the image.
  __device__ static __forceinline__ void read_row(const char* img, int lane, float (&z)[LD]) {
#pragma unroll
    for (int j = 0; j < PIECES; ++j) {
      const piece_t q = *reinterpret_cast<const piece_t*>(img + lane * STRIDE + j * PB);
      if constexpr (PB == 16) { z[4 * j] = q.x; z[4 * j + 1] = q.y; z[4 * j + 2] = q.z; z[4 * j + 3] = q.w; }
      else { z[2 * j] = q.x; z[2 * j + 1] = q.y; }
    }
  }
  __device__ static __forceinline__ void write_row(char* img, int lane, const float (&z)[LD]) {
#pragma unroll
    for (int j = 0; j < PIECES; ++j) {
      if constexpr (PB == 16)
        *reinterpret_cast<piece_t*>(img + lane * STRIDE + j * PB) = piece_t{z[4 * j], z[4 * j + 1], z[4 * j + 2], z[4 * j + 3]};
      else
        *reinterpret_cast<piece_t*>(img + lane * STRIDE + j * PB) = piece_t{z[2 * j], z[2 * j + 1]};
    }
  }
  // The same with the row as LD / 2 pairs of neighbouring elements (arithmetic on pairs maps onto packed float32).
  __device__ static __forceinline__ void read_row_pairs(const char* img, int lane, f2v (&z)[LD / 2]) {
#pragma unroll
    for (int j = 0; j < PIECES; ++j) {
      const piece_t q = *reinterpret_cast<const piece_t*>(img + lane * STRIDE + j * PB);
      if constexpr (PB == 16) { z[2 * j] = f2v{q.x, q.y}; z[2 * j + 1] = f2v{q.z, q.w}; }
      else z[j] = f2v{q.x, q.y};
    }
  }
  __device__ static __forceinline__ void write_row_pairs(char* img, int lane, const f2v (&z)[LD / 2]) {
#pragma unroll
    for (int j = 0; j < PIECES; ++j) {
      if constexpr (PB == 16)
        *reinterpret_cast<piece_t*>(img + lane * STRIDE + j * PB) = piece_t{z[2 * j].x, z[2 * j].y, z[2 * j + 1].x, z[2 * j + 1].y};
      else
        *reinterpret_cast<piece_t*>(img + lane * STRIDE + j * PB) = piece_t{z[j].x, z[j].y};
    }
  }
  // The image's first `nrows` rows to a contiguous block of rows starting at `first`, addressed as in request_flat.
  __device__ static __forceinline__ void store_flat(const char* img, char* __restrict__ first, unsigned nrows, int lane,
                                                    LaneSplit ls) {
    if (nrows >= 64) {                        // (uniform over the wave) every piece exists
#pragma unroll
      for (int h = 0; h < PIECES; ++h) {
        const unsigned q = h * 64 + lane;
        const unsigned at = (STRIDE != RB) ? piece_at(ls, h) : q * PB;
        __builtin_nontemporal_store(*reinterpret_cast<const piece_t*>(img + at), reinterpret_cast<piece_t*>(first + q * PB));
      }
    } else {
#pragma unroll
      for (int h = 0; h < PIECES; ++h) {
        const unsigned q = h * 64 + lane;
        const unsigned at = (STRIDE != RB) ? piece_at(ls, h) : q * PB;
        const piece_t v = *reinterpret_cast<const piece_t*>(img + at);
        if (q * PB < nrows * RB) __builtin_nontemporal_store(v, reinterpret_cast<piece_t*>(first + q * PB));
      }
    }
  }
  // The image's first `nrows` rows to global memory, coalesced, non-temporal (a streamed output).
  __device__ static __forceinline__ void store(const char* img, char* __restrict__ rows, unsigned base, unsigned nrows,
                                               int lane) {
#pragma unroll
    for (int h = 0; h < PIECES; ++h) {
      const unsigned q = h * 64 + lane, r = q / PIECES, j = q % PIECES;
      const piece_t v = *reinterpret_cast<const piece_t*>(img + r * STRIDE + j * PB);
      if (r < nrows) __builtin_nontemporal_store(v, reinterpret_cast<piece_t*>(rows + ((base + r) * RB + j * PB)));
    }
  }
};

}  // namespace d3pm
