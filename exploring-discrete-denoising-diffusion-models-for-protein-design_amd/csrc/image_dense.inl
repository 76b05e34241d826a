// Image-style D3PM kernels on dense float32 tables (Gaussian and banded
// transitions): row gathers for the factors that depend on integer tokens and a
// K x K contraction of the predicted x0 distribution against Qbar_{t-1}.
// Included by image_structured.hip (same translation unit, namespace d3pm).
//
// The contraction kernels work on tiles of TM tokens of one batch element: the
// tile's softmax rows sit in LDS, each thread owns one output class for TPT
// tokens and streams its column of Qbar_{t-1} once per tile (coalesced across
// the workgroup, L2-resident across tiles of the same timestep).

namespace d3pm {

constexpr int next_pow2_c(int x) { int p = 1; while (p < x) p <<= 1; return p; }

// MFMA = true: K = 256 only; the contraction runs on v_mfma_f32_32x32x2_f32 over
// tiles of 64 tokens (every table element fetched from L2 feeds two matrix
// instructions) whose LDS rows are padded by 16 bytes (conflict-free
// ds_read_b128 of the A fragments).  One 65 KB LDS buffer per workgroup and at
// most 256 registers per lane, so two workgroups share a CU and one's per-token
// phases run under the other's contraction.
template <class L, bool MFMA = false>
struct DenseCfg {
  static constexpr int TPP = BLOCK / L::G;              // tokens per epilogue pass
  static constexpr int TM = MFMA ? 64 : (TPP > 16 ? TPP : 16);   // tokens per tile
  static constexpr int PAD = MFMA ? 4 : 0;              // floats of padding per LDS row
  static constexpr int KC = next_pow2_c(L::G * L::V);   // threads per token group in the contraction
  static constexpr int CG = BLOCK / KC;                 // token groups
  static constexpr int TPT = TM / CG;                   // tokens per thread
  static_assert(TM % CG == 0 && TPT >= 1, "tile does not split over the workgroup");
  static_assert(TM % TPP == 0, "tile is a whole number of epilogue passes");
};

// ------------------------------------------------------------------ gathers only
template <class L>
__global__ __launch_bounds__(BLOCK) void k_dense_q_probs(ImgArgs a, float* __restrict__ out) {
  const int K = a.tb.K;
  const int b = blockIdx.x / a.chunks, chunk = blockIdx.x % a.chunks;
  const int g = threadIdx.x % L::G, slot = threadIdx.x / L::G;
  const int t = load_index(a.t, b, a.t_bytes, a.tb.T);
  const long p1 = min(a.npos, (chunk + 1) * a.chunk_pos);
  for (long pos = chunk * a.chunk_pos + slot; pos < p1; pos += BLOCK / L::G) {
    const long n = (long)b * a.npos + pos;
    const int x0 = load_index(a.x0, n, a.x0_bytes, a.tb.K);
    float r[L::V];
    load_slots<L>(a.tb.qbar + ((size_t)t * K + x0) * K, g, K, 0.0f, r);
    store_slots<L>(out + n * K, g, K, r);
  }
}

template <class L, int NOISE>
__global__ __launch_bounds__(BLOCK) void k_dense_q_sample(ImgArgs a, const float* __restrict__ noise, uint64_t seed,
                                                          uint64_t offset, const uint64_t* __restrict__ offset_dev,
                                                          int64_t* __restrict__ out) {
  if (NOISE == D3PM_NOISE_PHILOX && offset_dev) offset += *offset_dev;
  const int K = a.tb.K;
  const float eps = a.tb.eps;
  const int b = blockIdx.x / a.chunks, chunk = blockIdx.x % a.chunks;
  const int g = threadIdx.x % L::G, slot = threadIdx.x / L::G;
  const int t = load_index(a.t, b, a.t_bytes, a.tb.T);
  const long p1 = min(a.npos, (chunk + 1) * a.chunk_pos);
  for (long pos = chunk * a.chunk_pos + slot; pos < p1; pos += BLOCK / L::G) {
    const long n = (long)b * a.npos + pos;
    const int x0 = load_index(a.x0, n, a.x0_bytes, a.tb.K);
    float p[L::V], u[L::V];
    load_slots<L>(a.tb.qbar + ((size_t)t * K + x0) * K, g, K, 0.0f, p);
    if constexpr (NOISE == D3PM_NOISE_PHILOX) {
      // Noise drawn on the device: nothing to match bit for bit, so the draw is one uniform per token and an inverse
      // CDF over the weights q + eps (what the Gumbel-max over log(q + eps) samples from) -- no logarithms, one
      // Philox call per token instead of one per four classes.
      float w[L::V];
#pragma unroll
      for (int v = 0; v < L::V; ++v) w[v] = (L::VEC || L::cls(g, v) < K) ? p[v] + eps : 0.0f;
      const U4 r = philox4x32_10((uint64_t)n, offset, seed);
      const int pick = group_pick<L::G, L::V>(w, g, u01_open(r.x), x0, [](int gg, int vv) { return L::cls(gg, vv); });
      if (g == 0) out[n] = pick;
      continue;
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
      const float val = logf(p[v] + eps) + gum;
      if (c < K && (val > best || besti == 0x7fffffff)) { best = val; besti = c; }
    }
    group_argmax<L::G>(best, besti);
    if (g == 0) out[n] = besti;
  }
}

template <class L>
__global__ __launch_bounds__(BLOCK) void k_dense_posterior_tokens(ImgArgs a, float* __restrict__ out) {
  const int K = a.tb.K;
  const float eps = a.tb.eps;
  const int b = blockIdx.x / a.chunks, chunk = blockIdx.x % a.chunks;
  const int g = threadIdx.x % L::G, slot = threadIdx.x / L::G;
  const int t = load_index(a.t, b, a.t_bytes, a.tb.T);
  const float log_one = (float)log((double)(1.0f + eps)), log_zero = (float)log((double)eps);
  const long p1 = min(a.npos, (chunk + 1) * a.chunk_pos);
  for (long pos = chunk * a.chunk_pos + slot; pos < p1; pos += BLOCK / L::G) {
    const long n = (long)b * a.npos + pos;
    const int x0 = load_index(a.x0, n, a.x0_bytes, a.tb.K);
    const int xt = load_index(a.xt, n, a.xt_bytes, a.tb.K);
    float r[L::V];
    if (t == 0) {
#pragma unroll
      for (int v = 0; v < L::V; ++v) r[v] = (L::cls(g, v) == x0) ? log_one : log_zero;
    } else {
      float f1[L::V], f2[L::V];
      load_slots<L>(a.tb.qoneT + ((size_t)t * K + xt) * K, g, K, 0.0f, f1);
      load_slots<L>(a.tb.qbar + ((size_t)(t - 1) * K + x0) * K, g, K, 0.0f, f2);
#pragma unroll
      for (int v = 0; v < L::V; ++v) r[v] = logf(f1[v] + eps) + logf(f2[v] + eps);
    }
    store_slots<L>(out + n * K, g, K, r);
  }
}

template <class L>
__global__ __launch_bounds__(BLOCK) void k_dense_prior(ImgArgs a, int absorbing, float* __restrict__ part) {
  __shared__ float red[BLOCK / 64];
  const int K = a.tb.K;
  const float eps = a.tb.eps;
  const int b = blockIdx.x / a.chunks, chunk = blockIdx.x % a.chunks;
  const int g = threadIdx.x % L::G, slot = threadIdx.x / L::G;
  const int t = a.tb.T - 1;
  const float lp_uniform = logf(1.0f / (float)K + eps);
  const float lp_one = logf(1.0f + eps), lp_zero = logf(eps);
  float acc = 0;
  const long p1 = min(a.npos, (chunk + 1) * a.chunk_pos);
  for (long pos = chunk * a.chunk_pos + slot; pos < p1; pos += BLOCK / L::G) {
    const long n = (long)b * a.npos + pos;
    const int x0 = load_index(a.x0, n, a.x0_bytes, a.tb.K);
    float p[L::V];
    load_slots<L>(a.tb.qbar + ((size_t)t * K + x0) * K, g, K, 0.0f, p);
    float kl = 0;
#pragma unroll
    for (int v = 0; v < L::V; ++v) {
      const int c = L::cls(g, v);
      const float lprior = absorbing ? (c == K / 2 ? lp_one : lp_zero) : lp_uniform;
      kl += (c < K) ? p[v] * (logf(p[v] + eps) - lprior) : 0.0f;
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

// ------------------------------------------------------------------ tile kernel
// out[tok][j] = sum_i in[tok][i] * M[i][j] for the TM tokens of a tile; `in` and
// `out` are LDS [TM][K]; M is a global [K][K] float32 matrix.
template <class L>
__device__ __forceinline__ void tile_contract(const float* __restrict__ M, int K, const float* in, float* out) {
  using C = DenseCfg<L>;
  const int j = threadIdx.x % C::KC, tg = threadIdx.x / C::KC;
  float acc[C::TPT];
#pragma unroll
  for (int q = 0; q < C::TPT; ++q) acc[q] = 0.0f;
  if (j < K) {
    const float* col = M + j;
    const float* rows = in + (size_t)tg * C::TPT * K;
#pragma unroll 4
    for (int i = 0; i < K; ++i) {
      const float m = col[(size_t)i * K];
#pragma unroll
      for (int q = 0; q < C::TPT; ++q) acc[q] = fmaf(rows[q * K + i], m, acc[q]);
    }
#pragma unroll
    for (int q = 0; q < C::TPT; ++q) out[(size_t)(tg * C::TPT + q) * K + j] = acc[q];
  }
}

// The same contraction for K = 256 on the matrix cores, exact float32
// (v_mfma_f32_32x32x2_f32 is a k-ordered fmaf chain).  Four waves; wave w owns
// output columns [64w, 64w+64) of all TM rows: TM/32 x 2 accumulator tiles of
// 32 x 32.  The 256-long reduction is split over the two lane halves as
// k = 128 h + s, so a lane's A operands for four consecutive steps s are one
// 16-byte LDS read and its B operands are coalesced 128-byte row pieces of M,
// read straight from global memory (every wave needs different columns, so LDS
// staging of M would add traffic, not save it) one group of four steps ahead.
// `in` and `out` are LDS [TM][260]; they may be the same buffer.
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int TM>
__device__ __forceinline__ void tile_contract_mfma256(const float* __restrict__ M, const float* in, float* out) {
  constexpr int K = 256, LDK = K + 4, RT = TM / 32;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int r = lane & 31, h = lane >> 5;
  f32x16 acc[RT][2];
#pragma unroll
  for (int rt = 0; rt < RT; ++rt)
#pragma unroll
    for (int ct = 0; ct < 2; ++ct)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[rt][ct][e] = 0.0f;
  // M[128h + s][64w + r (+32)] through a buffer descriptor: per-lane 32-bit byte offset, the group's row offset in
  // a scalar register and the rest in the instruction's immediate field, so the loop carries no address registers
  const __amdgpu_buffer_rsrc_t Mrs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(M), 0, K * K * 4, 0x00020000);
  const int boff = ((128 * h) * K + 64 * wave + r) * 4;
  const float* arow = in + (size_t)r * LDK + 128 * h;                 // in[r (+32 rt)][128h + s]
  // groups of GS reduction steps; the B operands of the next group are requested
  // before the current group's MFMAs (GS * RT * 2 * 64 cycles of cover)
  constexpr int GS = 8, NG = 128 / GS;
  auto load_b = [&](float (&b)[GS][2], int grp) {
#pragma unroll
    for (int q = 0; q < GS; ++q) {
      const int so = (grp * GS + (q / 4) * 4) * K * 4;      // immediates reach 4095 bytes: four rows
#pragma unroll
      for (int ct = 0; ct < 2; ++ct)
        b[q][ct] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(Mrs, boff + ((q % 4) * K + 32 * ct) * 4, so, 0));
    }
  };
  auto load_a = [&](float4 (&av)[RT][GS / 4], int grp) {
#pragma unroll
    for (int rt = 0; rt < RT; ++rt)
#pragma unroll
      for (int f = 0; f < GS / 4; ++f)
        av[rt][f] = *reinterpret_cast<const float4*>(arow + (size_t)rt * 32 * LDK + grp * GS + 4 * f);
  };
  auto compute = [&](const float4 (&av)[RT][GS / 4], const float (&b)[GS][2]) {
#pragma unroll
    for (int q = 0; q < GS; ++q) {
#pragma unroll
      for (int rt = 0; rt < RT; ++rt) {
        const float4 a4 = av[rt][q / 4];
        const float aq = (q % 4) == 0 ? a4.x : ((q % 4) == 1 ? a4.y : ((q % 4) == 2 ? a4.z : a4.w));
#pragma unroll
        for (int ct = 0; ct < 2; ++ct)
          acc[rt][ct] = __builtin_amdgcn_mfma_f32_32x32x2f32(aq, b[q][ct], acc[rt][ct], 0, 0, 0);
      }
    }
  };
  // Both operand streams are double-buffered by hand.  Each half trip is scheduled as
  // one MFMA, one global load, one MFMA, ... so that issuing the next group's loads
  // hides under the matrix pipe instead of in front of it.
  auto interleave = [&]() {
#pragma unroll
    for (int i = 0; i < GS * 2 * RT; ++i) {
      __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);   // MFMA
      __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);   // VMEM read
      if (i < RT * GS / 4) __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);   // DS read
    }
    __builtin_amdgcn_sched_barrier(0);
  };
  float b0[GS][2], b1[GS][2];
  float4 a0[RT][GS / 4], a1[RT][GS / 4];
  load_b(b0, 0);
  load_a(a0, 0);
#pragma unroll 1
  for (int grp = 0; grp < NG; grp += 2) {
    load_b(b1, grp + 1);
    load_a(a1, grp + 1);
    compute(a0, b0);
    interleave();
    const int nxt = grp + 2 < NG ? grp + 2 : grp;     // unconditional (a repeat on the last trip): exact wait counts
    load_b(b0, nxt);
    load_a(a0, nxt);
    compute(a1, b1);
    interleave();
  }
  __syncthreads();     // every wave has read its A operands: `out` may alias `in`
#pragma unroll
  for (int rt = 0; rt < RT; ++rt)
#pragma unroll
    for (int ct = 0; ct < 2; ++ct)
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const int row = rt * 32 + (e & 3) + 8 * (e >> 2) + 4 * h;      // C/D layout of the 32x32 MFMA
        out[(size_t)row * LDK + 64 * wave + 32 * ct + r] = acc[rt][ct][e];
      }
}

// The same contraction with float32 emulated on the 16-bit matrix instructions, which on this part run at sixteen times
// the float32 matrix rate.  An operand x, scaled by a power of two into float16's range, is split into two float16
// pieces: hi = x rounded towards zero (11 significant bits), lo = float16 of the exact remainder x - hi (11 more bits
// wherever the remainder is a normal float16, an absolute 2^-25 of the scale below that).  Of the four piece products
// three are issued (lo hi', hi lo', hi hi': each exact in the float32 accumulator); lo lo' is below 2^-22 |x y|, about
// four float32 roundings, where an in-order float32 fma chain makes one rounding on each of its 256 steps.  Scales:
// table entries and softmax probabilities are at most 1 and carry 2^14 (an entry below 2^-39 is lost: against the eps
// = 1e-6 that every contracted value is added to before its logarithm, 2e-6 relative in the worst case); the backward
// operand d loss / d f2 carries a power of two per token that puts its largest element into [2^13, 2^14).
// v_mfma_f32_32x32x16_f16 takes 32 cycles: three per 16 reduction steps against eight 64-cycle v_mfma_f32_32x32x2_f32,
// 96 matrix-pipe cycles against 512.
// The table arrives pre-split (d3pm_table_split_f16x2: a lane's operand of eight reduction indices for one output
// class is one 16-byte piece); the LDS rows stay float32 and every wave splits the row pieces it reads.  Waves, output
// tiles and the C layout are those of tile_contract_mfma256.
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

// eight floats (already scaled into float16's range) -> their hi / lo float16 operands: hi = the value rounded towards
// zero, lo = float16 of the exact remainder
__device__ __forceinline__ void split_f16x2(const float4& p, const float4& q, u32x4& hi, u32x4& lo) {
  const float x[8] = {p.x, p.y, p.z, p.w, q.x, q.y, q.z, q.w};
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const float a = x[2 * i], b = x[2 * i + 1];
    const auto h = __builtin_amdgcn_cvt_pkrtz(a, b);
    const auto l = __builtin_amdgcn_cvt_pkrtz(a - (float)h[0], b - (float)h[1]);
    hi[i] = __builtin_bit_cast(unsigned, h);
    lo[i] = __builtin_bit_cast(unsigned, l);
  }
}

template <int TM>
__device__ __forceinline__ void tile_contract_f16x2(const uint16_t* __restrict__ Ms, const float* in, float* out) {
  constexpr int K = 256, LDK = K + 4, RT = TM / 32, NC = K / 16;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int r = lane & 31, h = lane >> 5;
  f32x16 acc[RT][2];
#pragma unroll
  for (int rt = 0; rt < RT; ++rt)
#pragma unroll
    for (int ct = 0; ct < 2; ++ct)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[rt][ct][e] = 0.0f;
  // Operand image (d3pm_table_split_f16x2): piece p of (reduction block u, class tile mt, lane kg << 4 | mm) at byte
  // ((u 16 + mt) 2 + p) 1024 + (kg 16 + mm) 16, holding reduction indices 32 u + 16 (s >> 2) + 4 kg + (s & 3).
  // Step c of lane half h takes (u, kg) = (c >> 1, 2 (c & 1) + h); its output class is 64 wave + 32 ct + r.
  const __amdgpu_buffer_rsrc_t Mrs =
      __builtin_amdgcn_make_buffer_rsrc(const_cast<uint16_t*>(Ms), 0, 2 * K * K * 2, 0x00020000);
  const int boff = (4 * wave + (r >> 4)) * 2048 + h * 256 + (r & 15) * 16;
  const float* arow = in + (size_t)r * LDK + 4 * h;
  auto load_b = [&](u32x4 (&b)[2][2], int c) {
#pragma unroll
    for (int s = 0; s < 2; ++s)
#pragma unroll
      for (int ct = 0; ct < 2; ++ct)
        b[s][ct] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(
            Mrs, boff + s * 1024, (c >> 1) * 32768 + (c & 1) * 512 + ct * 4096, 0));
  };
  auto load_a = [&](float4 (&av)[RT][2], int c) {
#pragma unroll
    for (int rt = 0; rt < RT; ++rt)
#pragma unroll
      for (int f = 0; f < 2; ++f)
        av[rt][f] = *reinterpret_cast<const float4*>(arow + (size_t)rt * 32 * LDK + 32 * (c >> 1) + 8 * (c & 1) + 16 * f);
  };
  auto mm = [&](f32x16& d, const u32x4& x, const u32x4& y) {
    d = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, x), __builtin_bit_cast(f16x8, y), d, 0, 0, 0);
  };
  auto compute = [&](const float4 (&av)[RT][2], const u32x4 (&b)[2][2]) {
#pragma unroll
    for (int rt = 0; rt < RT; ++rt) {
      u32x4 ah, al;
      split_f16x2(av[rt][0], av[rt][1], ah, al);
#pragma unroll
      for (int ct = 0; ct < 2; ++ct) {      // small terms first
        mm(acc[rt][ct], al, b[0][ct]);
        mm(acc[rt][ct], ah, b[1][ct]);
        mm(acc[rt][ct], ah, b[0][ct]);
      }
    }
  };
  // The table pieces come from L2: a ring of NB buffers keeps NB - 1 steps' worth of them in flight ((NB - 1) x 24
  // matrix instructions x 32 cycles of cover); the LDS row pieces are read one step ahead.
  constexpr int NB = 2;      // measured: a ring of four spills (210 registers) and is 18 % slower
  static_assert(NC % NB == 0, "the ring is unrolled over whole trips");
  u32x4 b[NB][2][2];
  float4 av[2][RT][2];
#pragma unroll
  for (int i = 0; i < NB - 1; ++i) load_b(b[i], i);
  load_a(av[0], 0);
#pragma unroll 1
  for (int c0 = 0; c0 < NC; c0 += NB) {
#pragma unroll
    for (int i = 0; i < NB; ++i) {
      const int c = c0 + i;
      load_b(b[(i + NB - 1) % NB], c + NB - 1 < NC ? c + NB - 1 : NC - 1);     // unconditional (repeats at the end)
      load_a(av[(i + 1) & 1], c + 1 < NC ? c + 1 : NC - 1);
      compute(av[i & 1], b[i]);
    }
  }
  __syncthreads();     // every wave has read its A operands: `out` may alias `in`
#pragma unroll
  for (int rt = 0; rt < RT; ++rt)
#pragma unroll
    for (int ct = 0; ct < 2; ++ct)
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const int row = rt * 32 + (e & 3) + 8 * (e >> 2) + 4 * h;
        out[(size_t)row * LDK + 64 * wave + 32 * ct + r] = acc[rt][ct][e];
      }
}

enum { DENSE_POST = 0, DENSE_LOSS = 1, DENSE_SAMPLE = 2 };

#ifndef D3PM_VALU_PRIO
#define D3PM_VALU_PRIO 2     // wave priority (0..3) of the per-token phases on the matrix-core path
#endif

struct DenseExtra {
  int B;
  unsigned long long* rng_counter; unsigned long long rng_step;   // see LossOut
  const float* logits; long ld;
  // loss
  int loss_type; float hybrid_coeff; float grad_scale; const float* grad_out; float* part; float* dlogits;
  // posterior logits out / probs out
  float* out; float* probs;
  // sampling
  const float* noise; uint64_t seed, offset; const uint64_t* offset_dev; int64_t* sample;
};

template <class L, int MODE, bool GRAD, int NOISE, bool MFMA, bool SPLIT = false>
__global__ __launch_bounds__(BLOCK, MFMA ? 2 : 1) void k_dense_tile(ImgArgs a, DenseExtra e) {
  static_assert(MFMA || !SPLIT, "the split contraction is a matrix-core path");
  static_assert(!(SPLIT && GRAD), "with split tables the training form runs on k_dense_loss256");
  constexpr float PSC = SPLIT ? 16384.0f : 1.0f;        // float16 operand pieces: pi x 2^14 into LDS, f2 = acc / 2^28
  using C = DenseCfg<L, MFMA>;
  extern __shared__ __align__(16) float lds[];
  __shared__ float red[2 * (BLOCK / 64)];
  // matrix-core path: per-token scalars (x0, x_t | CE term, softmax max, 1 / sum) live here between phases
  // instead of in registers, which are what limits two workgroups per CU
  __shared__ int s_toki[MFMA ? 2 * C::TM : 1];
  __shared__ float s_tokf[MFMA ? 3 * C::TM : 1];
  const int K = a.tb.K;
  const int LDK = K + C::PAD;              // LDS row stride
  const float eps = a.tb.eps;
  // VALU path: softmax(z), pi @ Qm (later d loss / d f2) and d loss / d pi in buffers of their own.  Matrix-core
  // path: ONE [TM][LDK] buffer that every phase transforms in place (the contraction reads all its operands before
  // it writes); the softmax needed again by the backward phase is recomputed from the logits.
  float* s_pi = lds;
  float* s_f2 = MFMA ? lds : lds + (size_t)C::TM * LDK;
  float* s_gp = MFMA ? lds : s_f2 + (size_t)C::TM * LDK;
  // Workgroups are dealt to the 8 XCDs round-robin by index; all chunks of a batch
  // element go to the same XCD so that its two 256 KB matrices (one timestep) are
  // fetched into one L2 only.  The grid is B rounded up to a multiple of 8.
  if (blockIdx.x == 0 && threadIdx.x == 0 && e.rng_counter) *e.rng_counter += e.rng_step;
  const int xcd = blockIdx.x % 8, within = blockIdx.x / 8;
  const int b = (within / a.chunks) * 8 + xcd, chunk = within % a.chunks;
  if (b >= e.B) return;
  const int g = threadIdx.x % L::G, slot = threadIdx.x / L::G;
  // vector layouts cover exactly K classes: no masked slots
  auto valid = [&](int c) { return L::VEC || c < K; };
  const int t = __builtin_amdgcn_readfirstlane(load_index(a.t, b, a.t_bytes, a.tb.T));   // one timestep per workgroup
  const bool need_post = (t > 0) && !(MODE == DENSE_LOSS && e.loss_type == D3PM_LOSS_CE);
  const float* Qm = need_post ? a.tb.qbar + (size_t)(t - 1) * K * K : nullptr;
  const float* QmT = (need_post && GRAD) ? a.tb.qbarT + (size_t)(t - 1) * K * K : nullptr;
  const uint16_t* QmS = (SPLIT && need_post) ? a.tb.qbar_f16x2 + (size_t)(t - 1) * 2 * K * K : nullptr;
  const uint16_t* QmTS = nullptr;

  float w_vb = 0, w_ce = 0;
  if constexpr (MODE == DENSE_LOSS) {
    const float gb = e.grad_scale * ((GRAD && e.grad_out) ? e.grad_out[b] : 1.0f);
    const float cnorm = gb / ((float)a.npos * LN2);
    w_vb = (e.loss_type == D3PM_LOSS_CE) ? 0.0f : cnorm;
    w_ce = (e.loss_type == D3PM_LOSS_KL) ? 0.0f : (e.loss_type == D3PM_LOSS_CE ? cnorm : e.hybrid_coeff * cnorm);
  }
  float acc_vb = 0, acc_ce = 0;
  if (MODE == DENSE_SAMPLE && NOISE == D3PM_NOISE_PHILOX && e.offset_dev) e.offset += *e.offset_dev;
  const long p1 = min(a.npos, (chunk + 1) * a.chunk_pos);
#ifdef D3PM_TIMELINE
#ifndef D3PM_TL_FIRST
#define D3PM_TL_FIRST 0      // first of the 256 workgroups whose clocks are kept
#endif
  // debug builds: time spent per phase, summed over this wave's tiles (slot 0: start, 6: end)
  unsigned long long ph_[TL_POINTS] = {}, last_ = wall_clock64();
  ph_[0] = last_;
#define D3PM_PHASE(i) do { const unsigned long long now_ = wall_clock64(); ph_[i] += now_ - last_; last_ = now_; } while (0)
#else
#define D3PM_PHASE(i) do {} while (0)
#endif
  // A lane group keeps the same tokens through all phases of a tile (token ps * TPP + slot in pass ps), so
  // their indices stay in registers.  Global reads run ahead of their use by a whole phase: a tile's logits rows
  // and tokens are requested during the previous tile's second contraction (or before its first, without a
  // backward pass), the table rows of the epilogue before the first contraction.
  // On the matrix-core path nothing is carried across a contraction: rows, tokens and table rows are requested one
  // pass ahead inside each phase through two rolling buffers (the next tile's first pass during this tile's last),
  // and the backward phase re-reads the tile's logits rows.
  constexpr int NP = C::TM / C::TPP;       // passes per phase
  constexpr int NZ = MFMA ? 2 : NP;        // buffers for rows, tokens, table rows
  float zn[NZ][L::V];
  int x0n[NZ], xtn[NZ];
  auto row_index = [&](long tile, int ps) {
    const long pos = tile + ps * C::TPP + slot;
    return (long)b * a.npos + (pos < p1 ? pos : tile);
  };
  auto fetch_row = [&](long tile, int ps) { load_slots<L>(e.logits + row_index(tile, ps) * e.ld, g, K, -INFINITY, zn[ps % NZ]); };
  auto fetch_pass = [&](long tile, int ps) {
    const long n = row_index(tile, ps);
    fetch_row(tile, ps);
    xtn[ps % NZ] = load_index(a.xt, n, a.xt_bytes, a.tb.K);
    x0n[ps % NZ] = (MODE == DENSE_LOSS) ? load_index(a.x0, n, a.x0_bytes, a.tb.K) : 0;
  };
  auto fetch_tile = [&](long tile) {
#pragma unroll
    for (int ps = 0; ps < (MFMA ? 1 : NP); ++ps) fetch_pass(tile, ps);
  };
  fetch_tile(chunk * a.chunk_pos);
  if constexpr (MFMA) __builtin_amdgcn_s_setprio(D3PM_VALU_PRIO);
  for (long tile0 = chunk * a.chunk_pos; tile0 < p1; tile0 += C::TM) {
    // ---- phase 1: softmax of the tile's logits into LDS, CE term
    constexpr int NKEEP = MFMA ? 1 : NP;
    float zkeep[NKEEP][L::V];   // VALU path: logits of this lane's tokens (used at t == 0); matrix-core path: re-read
    float cekeep[NP];
    int x0c[NP], xtc[NP];
#pragma unroll
    for (int ps = 0; ps < NP; ++ps) {
      const int tok = ps * C::TPP + slot;
      const bool live = tile0 + tok < p1;
      x0c[ps] = x0n[ps % NZ]; xtc[ps] = xtn[ps % NZ];
      float z[L::V], pi[L::V];
#pragma unroll
      for (int v = 0; v < L::V; ++v) z[v] = zn[ps % NZ][v];
      if constexpr (MFMA) {
        if (ps + 1 < NP) fetch_pass(tile0, ps + 1);
      }
      float m = z[0];
#pragma unroll
      for (int v = 1; v < L::V; ++v) m = fmaxf(m, z[v]);
      m = group_max<L::G>(m);
      float sum = 0;
#pragma unroll
      for (int v = 0; v < L::V; ++v) { pi[v] = fast_exp2((z[v] - m) * LOG2E); sum += pi[v]; }   // 0 on masked slots
      sum = group_sum<L::G>(sum);
      const float inv = fast_rcp(sum);
      if constexpr (L::VEC) {      // whole 16-byte pieces: ds_write_b128
        float pn[L::V];
#pragma unroll
        for (int v = 0; v < L::V; ++v) {
          if constexpr (!MFMA) zkeep[ps][v] = z[v];
          pn[v] = live ? pi[v] * (inv * PSC) : 0.0f;
        }
        store_slots<L>(s_pi + (size_t)tok * LDK, g, K, pn);
      } else {
#pragma unroll
        for (int v = 0; v < L::V; ++v) {
          zkeep[ps][v] = z[v];
          const int c = L::cls(g, v);
          if (c < K) s_pi[(size_t)tok * LDK + c] = live ? pi[v] * inv : 0.0f;
        }
      }
      cekeep[ps] = 0;
      if constexpr (MODE == DENSE_LOSS) {
        float zx0 = 0;
#pragma unroll
        for (int v = 0; v < L::V; ++v) zx0 += (L::cls(g, v) == x0c[ps]) ? z[v] : 0.0f;
        zx0 = group_sum<L::G>(zx0);
        cekeep[ps] = fmaf(fast_log2(sum), LN2, m) - zx0;
      }
      if constexpr (MFMA) {
        if (g == 0) {
          s_toki[2 * tok] = x0c[ps]; s_toki[2 * tok + 1] = xtc[ps];
          s_tokf[3 * tok] = cekeep[ps]; s_tokf[3 * tok + 1] = m; s_tokf[3 * tok + 2] = inv;
        }
      }
      if constexpr (MODE == DENSE_SAMPLE) {
        if (e.probs && live) {
#pragma unroll
          for (int v = 0; v < L::V; ++v) pi[v] *= inv;
          const long n = (long)b * a.npos + tile0 + tok;
          store_slots<L>(e.probs + n * K, g, K, pi);
        }
      }
      if constexpr (MFMA) __builtin_amdgcn_sched_barrier(0);   // keeps the next passes' loads from being hoisted (registers)
    }
    // table rows of the epilogue: column x_t of Q_t and (loss only) row x0 of Qbar_{t-1}; those of the first
    // pass are requested here, ahead of the contraction, the others one pass ahead
    float f1n[NZ][L::V], f2n[NZ][L::V];
    auto tok_x0 = [&](int ps) { return MFMA ? s_toki[2 * (ps * C::TPP + slot)] : x0c[ps]; };
    auto tok_xt = [&](int ps) { return MFMA ? s_toki[2 * (ps * C::TPP + slot) + 1] : xtc[ps]; };
    auto gather = [&](int ps) {
      load_slots<L>(a.tb.qoneT + ((size_t)t * K + tok_xt(ps)) * K, g, K, 0.0f, f1n[ps % NZ]);
      if constexpr (MODE == DENSE_LOSS)
        load_slots<L>(a.tb.qbar + ((size_t)(t - 1) * K + tok_x0(ps)) * K, g, K, 0.0f, f2n[ps % NZ]);
    };
    if (!MFMA && need_post) gather(0);
    const bool more = tile0 + C::TM < p1;
    if (!MFMA && !GRAD && more) fetch_tile(tile0 + C::TM);
    __syncthreads();
    D3PM_PHASE(1);
    // ---- phase 2: f2 = pi @ Qbar_{t-1}
    if (need_post) {
      if constexpr (MFMA) {
        // the co-resident workgroup's per-token phase issues ahead of this stream of long matrix instructions,
        // which loses next to nothing by it
        __builtin_amdgcn_s_setprio(0);
        if constexpr (SPLIT) tile_contract_f16x2<C::TM>(QmS, s_pi, s_f2);
        else tile_contract_mfma256<C::TM>(Qm, s_pi, s_f2);
        __builtin_amdgcn_s_setprio(D3PM_VALU_PRIO);
      } else tile_contract<L>(Qm, K, s_pi, s_f2);
    }
    __syncthreads();
    D3PM_PHASE(2);
    // ---- phase 3: per-token epilogue
    if (MFMA && need_post) gather(0);
#pragma unroll
    for (int ps = 0; ps < NP; ++ps) {
      const int tok = ps * C::TPP + slot;
      const long pos = tile0 + tok;
      const bool live = pos < p1;
      const long n = (long)b * a.npos + (live ? pos : tile0);
      float f2e[L::V];
      if constexpr (MFMA && MODE != DENSE_LOSS) {
        if (t == 0) load_slots<L>(e.logits + n * e.ld, g, K, -INFINITY, zkeep[0]);    // rare: 1 / T of the workgroups
      }
      if (need_post) {
        if (ps + 1 < NP) gather(ps + 1);
        load_slots<L>(s_f2 + (size_t)tok * LDK, g, K, 1.0f, f2e);
#pragma unroll
        for (int v = 0; v < L::V; ++v) f2e[v] = fmaf(f2e[v], 1.0f / (PSC * PSC), eps);
      }
      if constexpr (MFMA && !GRAD) {
        if (ps == NP - 1 && more) fetch_tile(tile0 + C::TM);
      }
      if constexpr (MODE == DENSE_LOSS) {
        // kl = sum_j q_j log(q_j / p_j) with q ~ a.*b, p ~ a.*f2 (a = column x_t of Q_t + eps, b = row x0 of
        // Qbar_{t-1} + eps, f2 = pi Qbar_{t-1} + eps): the reference's softmax(log a + log .) pairs folded, one
        // log per class (of a ratio close to one where q is large).
        const float ce = MFMA ? s_tokf[3 * tok] : cekeep[ps];
        float vb = ce;
        if (need_post) {
          float av[L::V], bv[L::V], zm = 0, zt = 0;
#pragma unroll
          for (int v = 0; v < L::V; ++v) {
            const bool ok = valid(L::cls(g, v));
            av[v] = ok ? f1n[ps % NZ][v] + eps : 0.0f;
            bv[v] = f2n[ps % NZ][v] + eps;
            zm = fmaf(av[v], f2e[v], zm);
            zt = fmaf(av[v], bv[v], zt);
          }
          zm = group_sum<L::G>(zm);
          zt = group_sum<L::G>(zt);
          const float izm = fast_rcp(zm), izt = fast_rcp(zt), zr = zm * izt;
          float kl = 0, gf2[L::V];
#pragma unroll
          for (int v = 0; v < L::V; ++v) {
            const float br = bv[v] * fast_rcp(f2e[v]);                     // b_j / f2_j
            kl = fmaf(av[v] * bv[v], fast_log2(br * zr), kl);              // av = 0 on masked slots
            if constexpr (GRAD) gf2[v] = live ? w_vb * av[v] * fmaf(-br, izt, izm) : 0.0f;
          }
          if constexpr (GRAD) {
            if constexpr (L::VEC) {
              store_slots<L>(s_f2 + (size_t)tok * LDK, g, K, gf2);
            } else {
#pragma unroll
              for (int v = 0; v < L::V; ++v) {
                const int c = L::cls(g, v);
                if (c < K) s_f2[(size_t)tok * LDK + c] = gf2[v];
              }
            }
          }
          vb = group_sum<L::G>(kl) * izt * LN2;                            // nats, like ce
        }
        if (g == 0 && live) { acc_vb += vb; acc_ce += ce; }
      } else {
        constexpr bool DEVICE_DRAW = MODE == DENSE_SAMPLE && NOISE == D3PM_NOISE_PHILOX;
        float lm[L::V];
        if (need_post && !DEVICE_DRAW) {
#pragma unroll
          for (int v = 0; v < L::V; ++v) lm[v] = logf(f1n[ps % NZ][v] + eps) + logf(f2e[v]);
        }
        if constexpr (MODE == DENSE_POST) {
          float r[L::V];
#pragma unroll
          for (int v = 0; v < L::V; ++v) r[v] = (t == 0) ? zkeep[ps % NKEEP][v] : lm[v];
          if (live) store_slots<L>(e.out + n * K, g, K, r);
        } else if constexpr (DEVICE_DRAW) {
          // Noise drawn on the device: nothing to match bit for bit, so the draw is one uniform per token and an
          // inverse CDF over the unnormalised posterior (a_j f2_j) -- no logarithm, no Gumbel transform, one Philox
          // call per token instead of one per four classes.  t == 0: argmax of the logits, as with explicit noise.
          int pick;
          if (t == 0) {
            float best = -INFINITY;
            pick = 0x7fffffff;
#pragma unroll
            for (int v = 0; v < L::V; ++v) {
              const int c = L::cls(g, v);
              const float val = zkeep[ps % NKEEP][v];
              if (valid(c) && (val > best || pick == 0x7fffffff)) { best = val; pick = c; }
            }
            group_argmax<L::G>(best, pick);
          } else {
            float w[L::V];
#pragma unroll
            for (int v = 0; v < L::V; ++v) w[v] = valid(L::cls(g, v)) ? (f1n[ps % NZ][v] + eps) * f2e[v] : 0.0f;
            const U4 r = philox4x32_10((uint64_t)n, e.offset, e.seed);
            const float uu = u01_open(r.x);
            pick = group_pick<L::G, L::V>(w, g, uu, tok_xt(ps), [](int gg, int vv) { return L::cls(gg, vv); });
          }
          if (g == 0 && live) e.sample[n] = pick;
        } else {   // DENSE_SAMPLE, noise tensor given
          float best = -INFINITY;
          int besti = 0x7fffffff;
          float u[L::V];
          if (t > 0) load_slots<L>(e.noise + n * K, g, K, 0.5f, u);
#pragma unroll
          for (int v = 0; v < L::V; ++v) {
            const int c = L::cls(g, v);
            float val;
            if (t == 0) val = zkeep[ps % NKEEP][v];
            else val = lm[v] + ((NOISE == D3PM_NOISE_GUMBEL) ? u[v] : gumbel_from_uniform(u[v]));
            if (valid(c) && (val > best || besti == 0x7fffffff)) { best = val; besti = c; }
          }
          group_argmax<L::G>(best, besti);
          if (g == 0 && live) e.sample[n] = besti;
        }
      }
      if constexpr (MFMA) __builtin_amdgcn_sched_barrier(0);
    }
    if constexpr (MODE == DENSE_LOSS && GRAD) {
      if constexpr (MFMA) fetch_row(tile0, 0);
      else if (more) fetch_tile(tile0 + C::TM);
      __syncthreads();
      D3PM_PHASE(3);
      // ---- phase 4: d loss / d pi = (d loss / d f2) @ Qbar_{t-1}^T
      if (need_post) {
        if constexpr (MFMA) {
          __builtin_amdgcn_s_setprio(0);
          if constexpr (SPLIT) tile_contract_f16x2<C::TM>(QmTS, s_f2, s_gp);
          else tile_contract_mfma256<C::TM>(QmT, s_f2, s_gp);
          __builtin_amdgcn_s_setprio(D3PM_VALU_PRIO);
        } else tile_contract<L>(QmT, K, s_f2, s_gp);
      }
      __syncthreads();
      D3PM_PHASE(4);
      // ---- phase 5: through the softmax of z, plus the CE term
#pragma unroll
      for (int ps = 0; ps < NP; ++ps) {
        const int tok = ps * C::TPP + slot;
        const long pos = tile0 + tok;
        const int x0 = tok_x0(ps);
        float pi[L::V], gz[L::V];
        if constexpr (MFMA) {
          // softmax again from the re-read row (same max and 1 / sum: bit-identical to phase 1); the row after it,
          // or the next tile's first pass, is requested meanwhile
          const float m = s_tokf[3 * tok + 1], inv = s_tokf[3 * tok + 2];
#pragma unroll
          for (int v = 0; v < L::V; ++v) pi[v] = fast_exp2((zn[ps % NZ][v] - m) * LOG2E) * inv;
          if (ps + 1 < NP) fetch_row(tile0, ps + 1);
          else if (more) fetch_tile(tile0 + C::TM);
        }
        if (pos >= p1) continue;     // whole lane groups leave together
        const long n = (long)b * a.npos + pos;
        if constexpr (!MFMA) load_slots<L>(s_pi + (size_t)tok * LDK, g, K, 0.0f, pi);
        if (need_post) {
          float gp[L::V], dot = 0;
          load_slots<L>(s_gp + (size_t)tok * LDK, g, K, 0.0f, gp);
#pragma unroll
          for (int v = 0; v < L::V; ++v) dot = fmaf(pi[v], gp[v], dot);
          dot = group_sum<L::G>(dot);
#pragma unroll
          for (int v = 0; v < L::V; ++v)
            gz[v] = pi[v] * (gp[v] - dot) + w_ce * (pi[v] - ((L::cls(g, v) == x0) ? 1.0f : 0.0f));
        } else {
          const float w = (t == 0 ? w_vb : 0.0f) + w_ce;
#pragma unroll
          for (int v = 0; v < L::V; ++v) gz[v] = w * (pi[v] - ((L::cls(g, v) == x0) ? 1.0f : 0.0f));
        }
        store_slots<L>(e.dlogits + n * e.ld, g, K, gz);
        if (!L::VEC && e.ld > K) zero_tail<L>(e.dlogits + n * e.ld, g, K, e.ld);
        if constexpr (MFMA) __builtin_amdgcn_sched_barrier(0);
      }
    }
    __syncthreads();   // LDS tiles are reused by the next tile
    D3PM_PHASE(5);
  }
#ifdef D3PM_TIMELINE
  ph_[6] = wall_clock64();
  ph_[7] = __builtin_amdgcn_s_getreg((31 << 11) | 4);              // HW_REG_HW_ID
  ph_[8] = __builtin_amdgcn_s_getreg((3 << 11) | 20);              // HW_REG_XCC_ID
  if (blockIdx.x >= D3PM_TL_FIRST && blockIdx.x < D3PM_TL_FIRST + 256 && (threadIdx.x & 63) == 0)
    for (int i_ = 0; i_ < TL_POINTS; ++i_) g_timeline[blockIdx.x - D3PM_TL_FIRST][threadIdx.x >> 6][i_] = ph_[i_];
#endif
  if constexpr (MODE == DENSE_LOSS) {
    acc_vb = wave_sum(acc_vb);
    acc_ce = wave_sum(acc_ce);
    const int wave = threadIdx.x >> 6;
    if ((threadIdx.x & 63) == 0) { red[2 * wave] = acc_vb; red[2 * wave + 1] = acc_ce; }
    __syncthreads();
    if (threadIdx.x == 0) {
      float svb = 0, sce = 0;
      for (int w = 0; w < BLOCK / 64; ++w) { svb += red[2 * w]; sce += red[2 * w + 1]; }
      e.part[((size_t)b * a.chunks + chunk) * 2 + 0] = svb;
      e.part[((size_t)b * a.chunks + chunk) * 2 + 1] = sce;
    }
  }
}

}  // namespace d3pm

#include "image_dense256.inl"

namespace d3pm {

// ------------------------------------------------------------------ host side
template <class L, bool MFMA>
static ImgArgs make_dense_args(const d3pm_tables_f32* tb, const void* x0, int x0b, const void* xt, int xtb, const void* t,
                               int tb_, int B, long npos) {
  ImgArgs a = make_args(tb, x0, x0b, xt, xtb, t, tb_, B, npos, L::G);
  // matrix-core path: two workgroups per CU, two rounds of them over the chip (measured best of 512..4096)
  const TokenGrid tg = token_grid(B, npos, DenseCfg<L, MFMA>::TM, MFMA ? 1024 : 2048);
  a.chunks = tg.chunks; a.chunk_pos = tg.chunk_pos;
  return a;
}

// The matrix-core contraction is built for K = 256.  Around it the per-token
// phases run 16 lanes per token with interleaved 16-byte pieces (4 tokens per
// wave, reductions inside a DPP row); f(layout, use_mfma) gets that layout, or
// the caller's with the VALU contraction.
using MfmaLayout = Layout<16, 16, true, true>;

// third argument of f: 0 = VALU contraction, 1 = float32 matrix instructions, 2 = float32 emulated with two float16
// pieces per operand (taken when the caller built both split tables)
template <class L, class F>
static int with_mfma_choice(const d3pm_tables_f32* tb, F&& f) {
  if constexpr (L::VEC) {
    if (tb->K == 256) {
      if (tb->qbar_f16x2 && tb->qbarT_f16x2) return f(MfmaLayout{}, std::integral_constant<int, 2>{});
      return f(MfmaLayout{}, std::integral_constant<int, 1>{});
    }
  }
  return f(L{}, std::integral_constant<int, 0>{});
}

template <class L, int MODE, bool GRAD, int NOISE, bool MFMA, bool SPLIT = false>
static int launch_tile(const d3pm_tables_f32* tb, const ImgArgs& a, const DenseExtra& e, int B, hipStream_t s) {
  using C = DenseCfg<L, MFMA>;
  const size_t sh = (size_t)C::TM * (tb->K + C::PAD) * sizeof(float) * (MFMA ? 1 : (GRAD ? 3 : 2));
  auto kern = k_dense_tile<L, MODE, GRAD, NOISE, MFMA, SPLIT>;
  if (sh > 48 * 1024) {
    hipError_t er = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)sh);
    if (er != hipSuccess) return hip_fail(er, "hipFuncSetAttribute(MaxDynamicSharedMemorySize)");
  }
  DenseExtra ee = e;
  ee.B = B;
  hipLaunchKernelGGL(kern, dim3(((B + 7) / 8) * 8 * a.chunks), dim3(BLOCK), sh, s, a, ee);
  D3PM_LAUNCH_CHECK("k_dense_tile");
  return D3PM_OK;
}

static int dense_q_probs(const d3pm_tables_f32* tb, const void* x0, int tok_bytes, const void* t, int t_bytes, int B,
                         long npos, float* out, hipStream_t s) {
  return dispatch_layout(tb->K, aligned16(out) && aligned16(tb->qbar), [&](auto lay) {
    using L = decltype(lay);
    ImgArgs a = make_args(tb, x0, tok_bytes, nullptr, 8, t, t_bytes, B, npos, L::G);
    hipLaunchKernelGGL(k_dense_q_probs<L>, dim3(B * a.chunks), dim3(BLOCK), 0, s, a, out);
    D3PM_LAUNCH_CHECK("k_dense_q_probs");
    return D3PM_OK;
  });
}

static int dense_q_sample(const d3pm_tables_f32* tb, const void* x0, int tok_bytes, const void* t, int t_bytes, int B,
                          long npos, const float* noise, int noise_kind, uint64_t seed, uint64_t offset,
                          const uint64_t* offset_dev, int64_t* x_t, hipStream_t s) {
  const bool vec = aligned16(tb->qbar) && (noise_kind == D3PM_NOISE_PHILOX || aligned16(noise));
  if (noise_kind == D3PM_NOISE_PHILOX && vec && tb->K == 256) {
    // 16 lanes per token, 16-byte pieces interleaved across the group: the inverse CDF's scan stays inside a DPP row
    // and a wave draws four tokens at a time
    using L = Layout<16, 16, true, true>;
    ImgArgs a = make_args(tb, x0, tok_bytes, nullptr, 8, t, t_bytes, B, npos, L::G);
    hipLaunchKernelGGL((k_dense_q_sample<L, D3PM_NOISE_PHILOX>), dim3(B * a.chunks), dim3(BLOCK), 0, s, a, noise, seed,
                       offset, offset_dev, x_t);
    D3PM_LAUNCH_CHECK("k_dense_q_sample");
    return D3PM_OK;
  }
  return dispatch_layout(tb->K, vec, [&](auto lay) {
    using L = decltype(lay);
    ImgArgs a = make_args(tb, x0, tok_bytes, nullptr, 8, t, t_bytes, B, npos, L::G);
    const dim3 grid(B * a.chunks), block(BLOCK);
    if (noise_kind == D3PM_NOISE_UNIFORM)
      hipLaunchKernelGGL((k_dense_q_sample<L, D3PM_NOISE_UNIFORM>), grid, block, 0, s, a, noise, seed, offset, offset_dev,
                         x_t);
    else if (noise_kind == D3PM_NOISE_GUMBEL)
      hipLaunchKernelGGL((k_dense_q_sample<L, D3PM_NOISE_GUMBEL>), grid, block, 0, s, a, noise, seed, offset, offset_dev,
                         x_t);
    else
      hipLaunchKernelGGL((k_dense_q_sample<L, D3PM_NOISE_PHILOX>), grid, block, 0, s, a, noise, seed, offset, offset_dev,
                         x_t);
    D3PM_LAUNCH_CHECK("k_dense_q_sample");
    return D3PM_OK;
  });
}

static int dense_posterior_logits(const d3pm_tables_f32* tb, const void* x_start, int x_start_is_logits, long ld,
                                  int tok_bytes, const void* x_t, int xt_bytes, const void* t, int t_bytes, int B,
                                  long npos, float* out, hipStream_t s) {
  const bool vec = aligned16(out) && aligned16(tb->qbar) && aligned16(tb->qoneT) &&
                   (!x_start_is_logits || (aligned16(x_start) && ld == tb->K));
  return dispatch_layout(tb->K, vec, [&](auto lay) {
    using L = decltype(lay);
    if (!x_start_is_logits) {
      ImgArgs a = make_args(tb, x_start, tok_bytes, x_t, xt_bytes, t, t_bytes, B, npos, L::G);
      hipLaunchKernelGGL(k_dense_posterior_tokens<L>, dim3(B * a.chunks), dim3(BLOCK), 0, s, a, out);
      D3PM_LAUNCH_CHECK("k_dense_posterior_tokens");
      return D3PM_OK;
    }
    return with_mfma_choice<decltype(lay)>(tb, [&](auto lay2, auto mf) {
      using L = decltype(lay2);
      constexpr bool MF = decltype(mf)::value != 0, SP = decltype(mf)::value == 2;
      ImgArgs a = make_dense_args<L, MF>(tb, nullptr, 8, x_t, xt_bytes, t, t_bytes, B, npos);
      DenseExtra e{};
      e.logits = (const float*)x_start; e.ld = ld; e.out = out;
      return launch_tile<L, DENSE_POST, false, D3PM_NOISE_GUMBEL, MF, SP>(tb, a, e, B, s);
    });
  });
}

static int dense_hybrid_loss(const d3pm_tables_f32* tb, const float* logits, long ld, const void* x0, int tok_bytes,
                             const void* x_t, int xt_bytes, const void* t, int t_bytes, int B, long npos, int loss_type,
                             float hybrid_coeff, float grad_scale, const float* grad_out, LossOut o,
                             float* dlogits, void* workspace, size_t workspace_bytes, hipStream_t s) {
  D3PM_CHECK_ARG(!dlogits || tb->qbarT, "dense backward needs the transposed cumulative table (qbarT)");
  const bool vec = aligned16(logits) && ld == tb->K && (!dlogits || aligned16(dlogits)) && aligned16(tb->qbar) &&
                   aligned16(tb->qoneT);
  // 64, 128 or 256 classes with the float16 split tables: the register-tile kernel (k_dense_loss256) -- at 256
  // classes for the training form only (forward only the tile kernel below is the faster one, 95 against 108 us at
  // B = 128 x 1024 tokens); at 64 and 128 classes, where the tile kernel contracts on the vector ALUs, for both
  const int K = tb->K;
  if (vec && tb->qbar_f16x2 && tb->qbarT_f16x2 && (K == 64 || K == 128 || (K == 256 && dlogits))) {
    ImgArgs a = make_args(tb, x0, tok_bytes, x_t, xt_bytes, t, t_bytes, B, npos, 1);
    const TokenGrid tg = token_grid(B, npos, 64, 1024);      // tiles of 64 tokens, two workgroups per CU, two rounds
    a.chunks = tg.chunks; a.chunk_pos = tg.chunk_pos;
    const size_t need = (size_t)B * a.chunks * 2 * sizeof(float);
    D3PM_CHECK_WORKSPACE(workspace, workspace_bytes, need);
    DenseExtra e{};
    e.logits = logits; e.ld = ld; e.loss_type = loss_type; e.hybrid_coeff = hybrid_coeff; e.grad_scale = grad_scale;
    e.grad_out = grad_out; e.part = (float*)workspace; e.dlogits = dlogits;
    e.rng_counter = o.rng_counter; e.rng_step = o.rng_step;
    int rc;
    if (K == 256) rc = launch_loss256<256, true>(a, e, B, s);
    else if (K == 128) rc = dlogits ? launch_loss256<128, true>(a, e, B, s) : launch_loss256<128, false>(a, e, B, s);
    else rc = dlogits ? launch_loss256<64, true>(a, e, B, s) : launch_loss256<64, false>(a, e, B, s);
    if (rc) return rc;
    LossOut oo = o;
    oo.part = e.part;
    hipLaunchKernelGGL(k_loss_finalize, dim3((B + 255) / 256), dim3(256), 0, s, oo, B, a.chunks, npos, loss_type,
                       hybrid_coeff, 1.0f / LN2);
    D3PM_LAUNCH_CHECK("k_loss_finalize");
    return D3PM_OK;
  }
  return dispatch_layout(tb->K, vec, [&](auto lay) {
    using L = decltype(lay);
    return with_mfma_choice<decltype(lay)>(tb, [&](auto lay2, auto mf) {
      using L = decltype(lay2);
      constexpr bool MF = decltype(mf)::value != 0, SP = decltype(mf)::value == 2;
      ImgArgs a = make_dense_args<L, MF>(tb, x0, tok_bytes, x_t, xt_bytes, t, t_bytes, B, npos);
      const size_t need = (size_t)B * a.chunks * 2 * sizeof(float);
      D3PM_CHECK_WORKSPACE(workspace, workspace_bytes, need);
      DenseExtra e{};
      e.logits = logits; e.ld = ld; e.loss_type = loss_type; e.hybrid_coeff = hybrid_coeff; e.grad_scale = grad_scale;
      e.grad_out = grad_out; e.part = (float*)workspace; e.dlogits = dlogits;
      e.rng_counter = o.rng_counter; e.rng_step = o.rng_step;
      int rc;
      if constexpr (SP) {
        // (the training form took the register-tile kernel above)
        rc = launch_tile<L, DENSE_LOSS, false, D3PM_NOISE_GUMBEL, MF, SP>(tb, a, e, B, s);
      } else {
        rc = dlogits ? launch_tile<L, DENSE_LOSS, true, D3PM_NOISE_GUMBEL, MF, SP>(tb, a, e, B, s)
                     : launch_tile<L, DENSE_LOSS, false, D3PM_NOISE_GUMBEL, MF, SP>(tb, a, e, B, s);
      }
      if (rc) return rc;
      LossOut oo = o;
      oo.part = e.part;
      hipLaunchKernelGGL(k_loss_finalize, dim3((B + 255) / 256), dim3(256), 0, s, oo, B,
                         a.chunks, npos, loss_type, hybrid_coeff, 1.0f / LN2);
      D3PM_LAUNCH_CHECK("k_loss_finalize");
      return D3PM_OK;
    });
  });
}

static int dense_p_sample(const d3pm_tables_f32* tb, const float* logits, long ld, const void* x_t, int xt_bytes,
                          const void* t, int t_bytes, int B, long npos, const float* noise, int noise_kind,
                          uint64_t seed, uint64_t offset, const uint64_t* offset_dev, int64_t* sample, float* probs,
                          hipStream_t s) {
  const bool vec = aligned16(logits) && ld == tb->K && (noise_kind == D3PM_NOISE_PHILOX || aligned16(noise)) &&
                   (!probs || aligned16(probs)) && aligned16(tb->qbar) && aligned16(tb->qoneT);
  // 64, 128 or 256 classes with the float16 split tables, device noise, no probabilities out (the sampling loop's
  // call): the register-tile kernel (76 against 85 us on the tile kernel below at 256 classes, B = 128 x 1024 tokens)
  if (vec && tb->qbar_f16x2 && noise_kind == D3PM_NOISE_PHILOX && !probs && (tb->K == 64 || tb->K == 128 || tb->K == 256)) {
    ImgArgs a = make_args(tb, nullptr, 8, x_t, xt_bytes, t, t_bytes, B, npos, 1);
    const TokenGrid tg = token_grid(B, npos, 64, 1024);
    a.chunks = tg.chunks; a.chunk_pos = tg.chunk_pos;
    DenseExtra e{};
    e.logits = logits; e.ld = ld; e.seed = seed; e.offset = offset; e.offset_dev = offset_dev; e.sample = sample;
    return tb->K == 256 ? launch_sample256<256>(a, e, B, s) : tb->K == 128 ? launch_sample256<128>(a, e, B, s) : launch_sample256<64>(a, e, B, s);
  }
  return dispatch_layout(tb->K, vec, [&](auto lay) {
    using L = decltype(lay);
    return with_mfma_choice<decltype(lay)>(tb, [&](auto lay2, auto mf) {
      using L = decltype(lay2);
      constexpr bool MF = decltype(mf)::value != 0, SP = decltype(mf)::value == 2;
      ImgArgs a = make_dense_args<L, MF>(tb, nullptr, 8, x_t, xt_bytes, t, t_bytes, B, npos);
      DenseExtra e{};
      e.logits = logits; e.ld = ld; e.noise = noise; e.seed = seed; e.offset = offset; e.offset_dev = offset_dev;
      e.sample = sample; e.probs = probs;
      if (noise_kind == D3PM_NOISE_UNIFORM) return launch_tile<L, DENSE_SAMPLE, false, D3PM_NOISE_UNIFORM, MF, SP>(tb, a, e, B, s);
      if (noise_kind == D3PM_NOISE_GUMBEL) return launch_tile<L, DENSE_SAMPLE, false, D3PM_NOISE_GUMBEL, MF, SP>(tb, a, e, B, s);
      return launch_tile<L, DENSE_SAMPLE, false, D3PM_NOISE_PHILOX, MF, SP>(tb, a, e, B, s);
    });
  });
}

static int dense_prior(const d3pm_tables_f32* tb, const void* x0, int tok_bytes, int B, long npos, int absorbing,
                       float* out, void* workspace, size_t workspace_bytes, hipStream_t s) {
  return dispatch_layout(tb->K, aligned16(tb->qbar), [&](auto lay) {
    using L = decltype(lay);
    ImgArgs a = make_args(tb, x0, tok_bytes, nullptr, 8, nullptr, 8, B, npos, L::G);
    const size_t need = (size_t)B * a.chunks * 2 * sizeof(float);
    D3PM_CHECK_WORKSPACE(workspace, workspace_bytes, need);
    float* part = (float*)workspace;
    hipLaunchKernelGGL(k_dense_prior<L>, dim3(B * a.chunks), dim3(BLOCK), 0, s, a, absorbing, part);
    D3PM_LAUNCH_CHECK("k_dense_prior");
    hipLaunchKernelGGL(k_loss_finalize, dim3((B + 255) / 256), dim3(256), 0, s,
                       LossOut{out, nullptr, nullptr, part, nullptr, 0, nullptr, 0.0f}, B,
                       a.chunks, npos, D3PM_LOSS_KL, 0.0f, 1.0f / LN2);
    D3PM_LAUNCH_CHECK("k_loss_finalize");
    return D3PM_OK;
  });
}

}  // namespace d3pm
