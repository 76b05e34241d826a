// Dense transition tables (Gaussian, banded) with K = 64, 128 or 256 classes: the
// training loss (vb_terms_bpd + cross_entropy_start + their gradient,
// d3pm/diffusion_discrete.py:462-582) and the sampling loop's p_sample
// (:384-403) with every per-token quantity held in registers from the logits
// load to the gradient / sample store.  Included by image_structured.hip after
// image_dense.inl (same translation unit, namespace d3pm).
//
// A wave owns 16 tokens.  Lane l = (kg << 4) | tok holds, for token `tok`, the
// K / 4 classes 16 mt + 4 kg + r (mt = 0..K/16-1, r = 0..3): element (mt, r) of
// every per-class array.  In that order
//   * a token's row is read and written as K / 16 16-byte pieces per lane;
//   * the two K x K contractions run as  out^T = table-image x act^T  on
//     v_mfma_f32_16x16x32_f16 with the table as the A operand (rows = output
//     classes) and the wave's activations as the B operand (columns = tokens):
//     accumulator tile mt of the result is element (mt, .) of the output array,
//     and elements (2u, .), (2u+1, .) of an input array are the B operand of
//     reduction block u -- so softmax -> contraction -> KL terms -> contraction
//     -> softmax backward never leave the lane's registers;
//   * sums over a token's classes are K / 4 - 1 additions in registers and two
//     cross-lane steps (lanes tok, tok + 16, tok + 32, tok + 48).
// float32 is emulated on the float16 matrix pipe as in tile_contract_f16x2 (two
// pieces per operand, three products); the activations are split once per use
// by the wave that owns them.  The table (d3pm_table_split_f16x2's operand image,
// 4 K^2 bytes per timestep and orientation: 256 KB at K = 256) streams from L2
// through a four-stage LDS ring of 16 KB stages shared by the workgroup's four
// waves, one barrier per stage; a stage is 8 output-class tiles x 2 pieces x 1 KB,
// byte for byte the image in global memory, and a wave's A-operand read is 1 KB
// of consecutive bytes (no bank conflicts).  At most 64 KB of LDS and 256
// registers: two workgroups per CU, one's per-token arithmetic under the other's
// contraction.

#ifndef D3PM_RT_AHEAD
#define D3PM_RT_AHEAD 2
#endif

namespace d3pm {

typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int RT_BS = 8;               // blocks (class tiles of one reduction block) per ring stage
constexpr int RT_STAGE = RT_BS * 2 * 1024;    // bytes of one ring stage
constexpr float RT_SCALE = 16384.0f;          // 2^14: what the table image and the forward activations are scaled by
// K classes: K / 16 class tiles (elements (mt, .) of a lane's arrays), K / 32 reduction blocks, K^2 / 512 blocks of
// (reduction block, class tile) = K^2 / 4096 ring stages per contraction (256 classes: 16, 128: 4, 64: 1)
// The ring: four stages (3 x 24 KB, 6 or 8 x 8 KB, 16 x 4 KB measured equal), a stage's DMA issued three stages ahead.
// (One workgroup of eight waves per CU on tiles of 128 tokens with a ring of nine stages -- half the table stream per
// token -- measured slower, 213 against 200 us: nothing then runs beside a workgroup's per-token phases.)
template <int K>
struct RtCfg {
  static_assert(K == 64 || K == 128 || K == 256, "alphabets of 64, 128 or 256 classes");
  static constexpr int RING = 4;                                  // stages in the LDS ring
  static constexpr int LA = RING - 1;                             // a stage's DMA is issued this many stages ahead
  static constexpr int DMA = RT_STAGE / 4096;                     // DMA instructions per wave and stage
  static constexpr int NMT = K / 16, NU = K / 32, NBLK = NU * NMT, NS = NBLK / RT_BS;
  static constexpr int TABLE = 2 * K * K * 2;                     // bytes of one timestep's operand image
  static constexpr int SPLIT_AT = NMT >= 8 ? NMT - 6 : NMT - 2;   // class tile (pair) after which the next block's pieces are formed
};

__device__ __forceinline__ float quad_lanes_sum(float x) {      // over lanes tok, tok+16, tok+32, tok+48
  x += __shfl_xor(x, 16);
  x += __shfl_xor(x, 32);
  return x;
}
__device__ __forceinline__ float quad_lanes_max(float x) {
  x = fmaxf(x, __shfl_xor(x, 16));
  x = fmaxf(x, __shfl_xor(x, 32));
  return x;
}

// Table stages go from L2 to the ring by LDS-DMA loads (buffer_load ... lds: no register destination; a wave's
// instruction moves 64 x 16 consecutive bytes, which is what a stage's image is made of).
typedef __attribute__((address_space(3))) void* lds_ptr;

__device__ __forceinline__ void ring_dma(const __amdgpu_buffer_rsrc_t& trs, char* ring, int s) {
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  // the lane's byte offset is formed here, every time, from the lane index (three vector operations): kept in a
  // register across the kernel it gets spilled, and a scratch reload in front of each DMA instruction waits for
  // the one before it (the counter of vector-memory operations completes in order)
  int off;
  asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0\n\tv_lshlrev_b32 %0, 4, %0" : "=v"(off));
#pragma unroll
  for (int i = 0; i < RT_STAGE / 4096; ++i)
    __builtin_amdgcn_raw_ptr_buffer_load_lds(trs, (lds_ptr)(ring + (s % 4) * RT_STAGE + (i * 4 + wave) * 1024), 16, off,
                                             s * RT_STAGE + (i * 4 + wave) * 1024, 0, 0);
}
template <int K>
__device__ __forceinline__ __amdgpu_buffer_rsrc_t table_rsrc(const char* table) {
  return __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(table), 0, RtCfg<K>::TABLE, 0x00020000);
}
// Requests the first stages of `table` (as many as are kept in flight).  Call once every wave of the workgroup is past its last read of the
// ring (regtile_contract ends with a barrier), as early as the table is known: the stages arrive under whatever the
// waves do until the contraction.
template <int K>
__device__ __forceinline__ void ring_begin(const char* table, char* ring) {
  using C = RtCfg<K>;
  static_assert(C::RING == 4, "ring_dma addresses a ring of four stages");
  const __amdgpu_buffer_rsrc_t trs = table_rsrc<K>(table);
#pragma unroll
  for (int s = 0; s < C::LA && s < C::NS; ++s) ring_dma(trs, ring, s);
}
// Barrier of the ring: waits until all but the wave's `pending` most recent DMA instructions have landed (and its
// LDS reads have returned), then meets the other waves.  Not __syncthreads(): its fence waits for every DMA.
template <int PENDING>
__device__ __forceinline__ void ring_barrier() {
  if constexpr (PENDING == 0) asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");
  else asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)\n\ts_barrier" ::"n"(PENDING) : "memory");
}

// The same with the count as a value (a constant once the caller's loops are unrolled).  n < 0: no stage has to land,
// the barrier only orders the waves' LDS reads against the next requests.
__device__ __forceinline__ void ring_barrier_n(int n) {
  switch (n) {
#define D3PM_RB(i) case i: ring_barrier<i>(); break;
    D3PM_RB(0) D3PM_RB(1) D3PM_RB(2) D3PM_RB(3) D3PM_RB(4) D3PM_RB(5) D3PM_RB(6) D3PM_RB(7) D3PM_RB(8) D3PM_RB(9)
    D3PM_RB(10) D3PM_RB(11) D3PM_RB(12) D3PM_RB(13) D3PM_RB(14) D3PM_RB(15) D3PM_RB(16) D3PM_RB(17) D3PM_RB(18)
    D3PM_RB(19) D3PM_RB(20) D3PM_RB(21) D3PM_RB(22) D3PM_RB(23) D3PM_RB(24) D3PM_RB(25) D3PM_RB(26) D3PM_RB(27)
    D3PM_RB(28) D3PM_RB(29) D3PM_RB(30) D3PM_RB(31) D3PM_RB(32) D3PM_RB(33) D3PM_RB(34) D3PM_RB(35) D3PM_RB(36)
    D3PM_RB(37) D3PM_RB(38) D3PM_RB(39) D3PM_RB(40)
#undef D3PM_RB
    default:
      if (n < 0) asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
      else ring_barrier<0>();
  }
}
// A caller's loads issued from inside the contraction (their latency passes under the matrix instructions; their
// destination registers are the ones the consumed activations leave free): hook(u, mt) is called once per pair of
// blocks, issues its loads and returns how many vector-memory instructions that was -- the counted waits of the ring
// barriers (vector-memory operations complete in order) leave that many more in flight.
struct NoHook {
  __device__ __forceinline__ int operator()(int, int) const { return 0; }
};

// out(mt, r) = sum_k table[k][16 mt + 4 kg + r] * act[k] for the lane's token, all K classes of `act` taking
// part (act in element order).  Every wave of the workgroup calls this together, after ring_begin(table).
//
// The contraction is a flat sequence of K^2 / 512 blocks (reduction block u, class tile mt): two 1 KB piece reads and
// three matrix instructions each.  Stage s = blocks 8 s .. 8 s + 7 sits in ring buffer s % 4; its DMA is issued at
// the start of stage s - 3 (its buffer's previous content, stage s - 4, was last read before the barrier that ends
// that stage), must have landed at the barrier that ends stage s - 2 (which waits for all but the most recent
// stage's DMA instructions: two stages of cover) and is read from stage s - 1 on: the piece reads run two pairs of
// blocks ahead of the matrix instructions and cross the stage boundaries, so neither a barrier nor an LDS latency
// sits between two stages' matrix instructions.  (One or two pairs ahead: within 1 %.)
template <int K, class Hook = NoHook>
__device__ __forceinline__ void regtile_contract(const char* __restrict__ table, char* ring,
                                                 const f32x4 (&act)[K / 16], f32x4 (&out)[K / 16], Hook hook = Hook()) {
  using C = RtCfg<K>;
  constexpr int NMT = C::NMT, NU = C::NU, NBLK = C::NBLK, RT_NS = C::NS;
  constexpr int RT_RING = C::RING, RT_LA = C::LA, RT_DMA = C::DMA;
  const int lane = threadIdx.x & 63;
  const __amdgpu_buffer_rsrc_t trs = table_rsrc<K>(table);
  auto mm = [&](f32x4& d, const u32x4& x, const u32x4& y) {
    d = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, x), __builtin_bit_cast(f16x8, y), d, 0, 0, 0);
  };
  struct Frag { u32x4 hi, lo; };
  auto read_frag = [&](int j) {       // block j = 8 s + ml
    const char* st = ring + ((j / RT_BS) % RT_RING) * RT_STAGE + (j % RT_BS) * 2048 + lane * 16;
    Frag f;
    f.hi = *reinterpret_cast<const u32x4*>(st);
    f.lo = *reinterpret_cast<const u32x4*>(st + 1024);
    return f;
  };
  // what the caller computed for `act` is complete here: left free, the compiler sinks the producers of act's
  // later elements into the loop and keeps their inputs alive, spilled, across it.  An empty asm statement that
  // takes every element as an in-out operand (no instruction is emitted) pins them, the scheduling fence keeps
  // the machine scheduler from moving anything across
#pragma unroll
  for (int mt = 0; mt < NMT; ++mt) asm volatile("" : "+v"(const_cast<f32x4&>(act[mt])));
  __builtin_amdgcn_sched_barrier(0);
#pragma unroll
  for (int mt = 0; mt < NMT; ++mt) out[mt] = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
  // stages 0 and 1 have landed and are visible; later ones may be on their way
  ring_barrier<((RT_LA < RT_NS ? RT_LA : RT_NS) > 2 ? (RT_LA < RT_NS ? RT_LA : RT_NS) - 2 : 0) * RT_DMA>();
  __builtin_amdgcn_s_setprio(0);     // the co-resident workgroup's per-token arithmetic issues ahead of this stream
  // The activations' float16 pieces, all reduction blocks, formed in place of the float32 values before the first
  // matrix instruction (block by block, so that the allocator can give a block's pieces the registers of its inputs).
  struct Pieces { u32x4 hi, lo; };
  Pieces pc[NU];
  __builtin_amdgcn_sched_barrier(0);
#pragma unroll
  for (int u = 0; u < NU; ++u) {
    split_f16x2(__builtin_bit_cast(float4, act[2 * u]), __builtin_bit_cast(float4, act[2 * u + 1]), pc[u].hi, pc[u].lo);
    __builtin_amdgcn_sched_barrier(0);
  }
  // blocks go in pairs (two class tiles: two independent accumulators, their matrix instructions interleaved);
  // the piece reads run AH pairs ahead
  constexpr int AH = D3PM_RT_AHEAD, NF = 2 * (AH + 1);
  static_assert(NBLK >= 2 * AH, "prologue reads");
  // a wave has 4 (AH + 1) piece reads in flight at most; the counter that s_waitcnt lgkmcnt reads has four bits
  static_assert(4 * (AH + 1) <= 15, "LDS reads in flight per wave");
  Frag fr[NF];
#pragma unroll
  for (int q = 0; q < 2 * AH; ++q) fr[q] = read_frag(q);
  __builtin_amdgcn_sched_barrier(0);
  int ex[RT_NS] = {};                // the hook's loads issued during each stage
#pragma unroll
  for (int u = 0; u < NU; ++u) {
    const Pieces& b = pc[u];
#pragma unroll
    for (int mt = 0; mt < NMT; mt += 2) {
      const int j = u * NMT + mt, s = j / RT_BS;              // block (u, mt) is block j of the image; its stage
      if (j % RT_BS == 0 && s + RT_LA < RT_NS) ring_dma(trs, ring, s + RT_LA);
      ex[s] += hook(u, mt);
      // The four piece reads go out in a row, ahead of the pair's six matrix instructions, and the fence keeps them
      // there.  Issued one by one between the matrix instructions (sched_group_barrier) the kernel was 1 % faster and
      // WRONG: at full size 1 - 3 % of the gradient elements differed from launch to launch -- a read's destination
      // registers had been operands of a matrix instruction issued just before it.
      if (j + 2 * AH < NBLK) {
        fr[(j + 2 * AH) % NF] = read_frag(j + 2 * AH);
        fr[(j + 2 * AH + 1) % NF] = read_frag(j + 2 * AH + 1);
      }
      __builtin_amdgcn_sched_barrier(0);
      const Frag &f0 = fr[j % NF], &f1 = fr[(j + 1) % NF];
      f32x4 &d0 = out[mt], &d1 = out[mt + 1];
      mm(d0, f0.lo, b.hi);      // small terms first
      mm(d1, f1.lo, b.hi);
      mm(d0, f0.hi, b.lo);
      mm(d1, f1.hi, b.lo);
      mm(d0, f0.hi, b.hi);
      mm(d1, f1.hi, b.hi);
      __builtin_amdgcn_sched_barrier(0);
      // end of a stage: stage s + 2 has landed (the piece reads reach it before the next barrier); later ones may be
      // in flight
      // Issued behind stage s + 2's DMA (which went out at the start of stage s + 2 - LA, or in ring_begin): the
      // DMA of the stages after it up to the latest one requested, and the hook's loads from that stage on.
      if (j % RT_BS == RT_BS - 2) {
        int n = -1;
        if (s + 2 < RT_NS) {
          const int last = s + RT_LA < RT_NS ? s + RT_LA : RT_NS - 1;
          n = (last - (s + 2)) * RT_DMA;
#pragma unroll
          for (int q = (s + 2 - RT_LA > 0 ? s + 2 - RT_LA : 0); q <= s; ++q) n += ex[q];
        }
        ring_barrier_n(n);
      }
    }
  }
  __builtin_amdgcn_s_setprio(D3PM_VALU_PRIO);
  __builtin_amdgcn_sched_barrier(0);
}

template <int K, bool GRAD>
__global__ __launch_bounds__(256, 2) void k_dense_loss256(ImgArgs a, DenseExtra e) {
  extern __shared__ __align__(16) char ring[];
  __shared__ float red[2 * 4];
  constexpr int TILE = 64;                      // tokens per tile: 16 per wave
  constexpr int NMT = K / 16;                   // class tiles = 16-byte pieces of a lane's share of a row
  constexpr int RT_TABLE = RtCfg<K>::TABLE;
  const float eps = a.tb.eps;
  if (blockIdx.x == 0 && threadIdx.x == 0 && e.rng_counter) *e.rng_counter += e.rng_step;
  // workgroup -> (batch element, chunk) as in k_dense_tile: all chunks of an element on one XCD
  const int xcd = blockIdx.x % 8, within = blockIdx.x / 8;
  const int b = (within / a.chunks) * 8 + xcd, chunk = within % a.chunks;
  if (b >= e.B) return;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int tok = lane & 15, kg = lane >> 4;
  const int t = __builtin_amdgcn_readfirstlane(load_index(a.t, b, a.t_bytes, a.tb.T));
  const bool need_post = (t > 0) && e.loss_type != D3PM_LOSS_CE;
  const char* tableF = need_post ? reinterpret_cast<const char*>(a.tb.qbar_f16x2) + (size_t)(t - 1) * RT_TABLE : nullptr;
  const char* tableB =
      (need_post && GRAD) ? reinterpret_cast<const char*>(a.tb.qbarT_f16x2) + (size_t)(t - 1) * RT_TABLE : nullptr;
  const float gb = e.grad_scale * ((GRAD && e.grad_out) ? e.grad_out[b] : 1.0f);
  const float cnorm = gb / ((float)a.npos * LN2);
  const float w_vb = (e.loss_type == D3PM_LOSS_CE) ? 0.0f : cnorm;
  const float w_ce = (e.loss_type == D3PM_LOSS_KL) ? 0.0f : (e.loss_type == D3PM_LOSS_CE ? cnorm : e.hybrid_coeff * cnorm);
  float acc_vb = 0, acc_ce = 0;
  const long p1 = min(a.npos, (chunk + 1) * a.chunk_pos);
#ifdef D3PM_TIMELINE
  // debug builds: time spent per phase, summed over this wave's tiles (slot 0: start, 6: end)
  unsigned long long ph_[TL_POINTS] = {}, last_ = wall_clock64();
  ph_[0] = last_;
#define D3PM_PHASE256(i) do { const unsigned long long now_ = wall_clock64(); ph_[i] += now_ - last_; last_ = now_; } while (0)
#else
#define D3PM_PHASE256(i) do {} while (0)
#endif
  __builtin_amdgcn_s_setprio(D3PM_VALU_PRIO);
  if (need_post) ring_begin<K>(tableF, ring);
  auto load_row4 = [&](const __amdgpu_buffer_rsrc_t& rs, int off, int mt) {
    return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, off + 64 * mt, 0, 0));
  };
  // rows of a tile through descriptors based at the tile's first row (rows are K floats: ld == K here)
  auto tile_rsrc = [&](const float* base, long tile0) {
    const long row0 = (long)b * a.npos + tile0;
    const int rows = (int)max((long)0, min((long)TILE, p1 - tile0));       // past the chunk's end: no rows, loads return 0
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(base) + row0 * K, 0, rows * K * 4, 0x00020000);
  };
  auto tile_rowoff = [&](long tile0) {
    return (tile0 + 16 * wave + tok < p1 ? 16 * wave + tok : 0) * (K * 4) + 16 * kg;
  };
  // The token's logits.  With 256 classes they are requested one tile ahead (training form: under the last
  // per-token phase of the tile before) and re-read for the softmax backward from inside the second contraction:
  // 205 -> 194 us at B = 128 x 1024 tokens.  With 64 and 128 classes, whose contractions are a quarter and a
  // sixteenth as long, the same measured 1 - 2 us slower: there every row is requested where it is needed.
  constexpr bool AHEAD = K == 256;
  f32x4 v[NMT];
  auto request_rows = [&](long tile0) {
    const __amdgpu_buffer_rsrc_t rs = tile_rsrc(e.logits, tile0);
    const int off = tile_rowoff(tile0);
#pragma unroll
    for (int mt = 0; mt < NMT; ++mt) v[mt] = load_row4(rs, off, mt);
  };
  if (AHEAD) request_rows(chunk * a.chunk_pos);
  for (long tile0 = chunk * a.chunk_pos; tile0 < p1; tile0 += TILE) {
    const long pos = tile0 + 16 * wave + tok;
    const bool live = pos < p1;
    const long n = (long)b * a.npos + (live ? pos : tile0);
    const int x0 = load_index(a.x0, n, a.x0_bytes, K);
    const int xt = load_index(a.xt, n, a.xt_bytes, K);
    const __amdgpu_buffer_rsrc_t zrs = tile_rsrc(e.logits, tile0);
    const int rowoff = tile_rowoff(tile0);
    if (!AHEAD) request_rows(tile0);
    // ---- softmax of the token's logits, cross-entropy term
    const float zx0 = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(zrs, rowoff - 16 * kg + 4 * x0, 0, 0));
    float m = v[0][0];
#pragma unroll
    for (int mt = 0; mt < NMT; ++mt)
#pragma unroll
      for (int r = 0; r < 4; ++r) m = fmaxf(m, v[mt][r]);
    m = quad_lanes_max(m);
    float sum = 0;
#pragma unroll
    for (int mt = 0; mt < NMT; ++mt)
#pragma unroll
      for (int r = 0; r < 4; ++r) { v[mt][r] = fast_exp2((v[mt][r] - m) * LOG2E); sum += v[mt][r]; }
    sum = quad_lanes_sum(sum);
    const float inv = fast_rcp(sum);
    const float ce = fmaf(fast_log2(sum), LN2, m) - zx0;
    const float pscale = live ? inv * RT_SCALE : 0.0f;      // the contraction's operand: pi x 2^14 (float16 pieces)
#pragma unroll
    for (int mt = 0; mt < NMT; ++mt) v[mt] *= pscale;
    float vb = ce;
    D3PM_PHASE256(1);
    if (need_post) {
      // ---- f2 = pi @ Qbar_{t-1}
      f32x4 f2[NMT];
      regtile_contract<K>(tableF, ring, v, f2);
      if (GRAD) ring_begin<K>(tableB, ring);
      else if (tile0 + TILE < p1) ring_begin<K>(tableF, ring);
      D3PM_PHASE256(2);
      // ---- KL(q || p) with q ~ a.*b, p ~ a.*f2 (a = column x_t of Q_t + eps, b = row x0 of Qbar_{t-1} + eps,
      // f2 += eps), one log per class; d loss / d f2 replaces f2 in its registers.  (Requesting the two table rows
      // from inside the contraction, which fits at 64 and 128 classes, measured equal with the gradient and 3 us
      // slower without at 128 classes.)
      const __amdgpu_buffer_rsrc_t ars = __builtin_amdgcn_make_buffer_rsrc(
          const_cast<float*>(a.tb.qoneT) + (size_t)t * K * K, 0, K * K * 4, 0x00020000);
      const __amdgpu_buffer_rsrc_t brs = __builtin_amdgcn_make_buffer_rsrc(
          const_cast<float*>(a.tb.qbar) + (size_t)(t - 1) * K * K, 0, K * K * 4, 0x00020000);
      const int aoff = xt * (K * 4) + 16 * kg, boff = x0 * (K * 4) + 16 * kg;
      f32x4 av[NMT], bv[NMT];
#pragma unroll
      for (int mt = 0; mt < NMT; ++mt) {
        av[mt] = load_row4(ars, aoff, mt) + eps;
        bv[mt] = load_row4(brs, boff, mt) + eps;
      }
      float zm = 0, zt = 0;
#pragma unroll
      for (int mt = 0; mt < NMT; ++mt)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          f2[mt][r] = fmaf(f2[mt][r], 1.0f / (RT_SCALE * RT_SCALE), eps);
          zm = fmaf(av[mt][r], f2[mt][r], zm);
          zt = fmaf(av[mt][r], bv[mt][r], zt);
        }
      zm = quad_lanes_sum(zm);
      zt = quad_lanes_sum(zt);
      const float izm = fast_rcp(zm), izt = fast_rcp(zt), zr = zm * izt;
      const float wv = live ? w_vb : 0.0f;
      float kl = 0;
#pragma unroll
      for (int mt = 0; mt < NMT; ++mt)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const float br = bv[mt][r] * fast_rcp(f2[mt][r]);
          kl = fmaf(av[mt][r] * bv[mt][r], fast_log2(br * zr), kl);
          if constexpr (GRAD) f2[mt][r] = wv * av[mt][r] * fmaf(-br, izt, izm);
        }
      vb = quad_lanes_sum(kl) * izt * LN2;
      D3PM_PHASE256(3);
      float unscale = 0;
      if constexpr (GRAD) {
        // the token's d loss / d f2 row scaled by a power of two into [2^13, 2^14): float16 pieces keep 22
        // significant bits of the largest elements and an absolute 2^-39 of that scale below
        float mx = 0;
#pragma unroll
        for (int mt = 0; mt < NMT; ++mt)
#pragma unroll
          for (int r = 0; r < 4; ++r) mx = fmaxf(mx, fabsf(f2[mt][r]));
        mx = quad_lanes_max(mx);
        const int ex = max(-80, min(80, __builtin_amdgcn_frexp_expf(mx)));       // mx = m 2^ex, m in [0.5, 1)
        const float sc = __builtin_ldexpf(1.0f, 14 - ex);
        unscale = __builtin_ldexpf(1.0f, ex - 28);                                // 1 / (sc x 2^14)
#pragma unroll
        for (int mt = 0; mt < NMT; ++mt) f2[mt] *= sc;
      }
      if constexpr (GRAD) {
        // ---- d loss / d pi = (d loss / d f2) @ Qbar_{t-1}^T, then through the softmax (re-formed from the
        // re-read row with the same max and 1 / sum: bit-identical) plus the cross-entropy term
        // (256 classes: the row is re-read from inside the contraction, four pieces per reduction block of its second half)
        f32x4 gp[NMT], pi[NMT];
        regtile_contract<K>(tableB, ring, f2, gp, [&](int u, int mt) {
          constexpr int NU = K / 32;
          if (!AHEAD || mt != 0 || u < NU / 2) return 0;
#pragma unroll
          for (int k = 0; k < 4; ++k) pi[4 * (u - NU / 2) + k] = load_row4(zrs, rowoff, 4 * (u - NU / 2) + k);
          return 4;
        });
        if (!AHEAD) {
#pragma unroll
          for (int mt = 0; mt < NMT; ++mt) pi[mt] = load_row4(zrs, rowoff, mt);
        }
        if (tile0 + TILE < p1) ring_begin<K>(tableF, ring);
        // the next tile's rows arrive under this tile's last phase.  (Unconditionally: past the chunk's end the
        // descriptor holds no rows.  Under a condition the previous contents of v would stay live, in scratch,
        // from the forward contraction to this point.)
        if (AHEAD) request_rows(tile0 + TILE);
        D3PM_PHASE256(4);
        float dot = 0;
#pragma unroll
        for (int mt = 0; mt < NMT; ++mt) {
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            pi[mt][r] = fast_exp2((pi[mt][r] - m) * LOG2E) * inv;
            gp[mt][r] *= unscale;
            dot = fmaf(pi[mt][r], gp[mt][r], dot);
          }
        }
        dot = quad_lanes_sum(dot);
        if (live) {
          const int d = x0 - 4 * kg;                               // element (d >> 4, d & 3) is class x0, if this lane has it
          const int mt_hit = (d >= 0 && (d & 12) == 0) ? (d >> 4) : -1;
          float sel[4];
#pragma unroll
          for (int r = 0; r < 4; ++r) sel[r] = ((d & 3) == r) ? w_ce : 0.0f;
          const __amdgpu_buffer_rsrc_t grs = tile_rsrc(e.dlogits, tile0);
#pragma unroll
          for (int mt = 0; mt < NMT; ++mt) {
            const float hit = (mt == mt_hit) ? 1.0f : 0.0f;
            f32x4 gz;
#pragma unroll
            for (int r = 0; r < 4; ++r)
              gz[r] = fmaf(-hit, sel[r], pi[mt][r] * (gp[mt][r] - dot) + w_ce * pi[mt][r]);
            __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, gz), grs, rowoff + 64 * mt, 0, 0);
          }
        }
      }
    } else if constexpr (GRAD) {
      // t == 0 (the bound term is the cross-entropy) or a cross-entropy-only loss: no contraction
      if (live) {
        const float w = (t == 0 ? w_vb : 0.0f) + w_ce;
        const int d = x0 - 4 * kg;
        const int mt_hit = (d >= 0 && (d & 12) == 0) ? (d >> 4) : -1;
        float sel[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) sel[r] = ((d & 3) == r) ? w : 0.0f;
        const __amdgpu_buffer_rsrc_t grs = tile_rsrc(e.dlogits, tile0);
#pragma unroll
        for (int mt = 0; mt < NMT; ++mt) {
          const float hit = (mt == mt_hit) ? 1.0f : 0.0f;
          f32x4 gz;
          // (the row again: the probabilities of the softmax above are not kept past the branch, so that in the
          // other branch their registers are free from the forward contraction on)
          const f32x4 z = load_row4(zrs, rowoff, mt);
#pragma unroll
          for (int r = 0; r < 4; ++r) gz[r] = fmaf(-hit, sel[r], w * (fast_exp2((z[r] - m) * LOG2E) * inv));
          __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, gz), grs, rowoff + 64 * mt, 0, 0);
        }
      }
    }
    if (kg == 0 && live) { acc_vb += vb; acc_ce += ce; }
    if (AHEAD && !(GRAD && need_post)) request_rows(tile0 + TILE);
    D3PM_PHASE256(5);
  }
#ifdef D3PM_TIMELINE
  ph_[6] = wall_clock64();
  ph_[7] = __builtin_amdgcn_s_getreg((31 << 11) | 4);              // HW_REG_HW_ID
  ph_[8] = __builtin_amdgcn_s_getreg((3 << 11) | 20);              // HW_REG_XCC_ID
  if (blockIdx.x < 256 && lane == 0)
    for (int i_ = 0; i_ < TL_POINTS; ++i_) g_timeline[blockIdx.x][wave][i_] = ph_[i_];
#endif
  acc_vb = wave_sum(acc_vb);
  acc_ce = wave_sum(acc_ce);
  if (lane == 0) { red[2 * wave] = acc_vb; red[2 * wave + 1] = acc_ce; }
  __syncthreads();
  if (threadIdx.x == 0) {
    float svb = 0, sce = 0;
    for (int w = 0; w < 4; ++w) { svb += red[2 * w]; sce += red[2 * w + 1]; }
    e.part[((size_t)b * a.chunks + chunk) * 2 + 0] = svb;
    e.part[((size_t)b * a.chunks + chunk) * 2 + 1] = sce;
  }
}

// p_sample (d3pm/diffusion_discrete.py:384-403) with noise drawn on the device and no probabilities asked for, on
// the same register tile: softmax -> contraction -> posterior weights (a_j + eps)(f2_j + eps) -> one uniform per
// token and an inverse CDF over the token's classes in (lane, element) order.  t == 0: the argmax of the logits
// (first index on ties), as with explicit noise.  The draw consumes word x of Philox4x32-10 keyed (token index,
// offset, seed), as the tile kernel's does; the class order of the cumulative sums differs from that kernel's, so the
// two draw different (equally distributed) tokens from one stream.
template <int K>
__global__ __launch_bounds__(256, 2) void k_dense_sample256(ImgArgs a, DenseExtra e) {
  extern __shared__ __align__(16) char ring[];
  constexpr int NMT = K / 16;
  constexpr int RT_TABLE = RtCfg<K>::TABLE;
  const float eps = a.tb.eps;
  const int xcd = blockIdx.x % 8, within = blockIdx.x / 8;
  const int b = (within / a.chunks) * 8 + xcd, chunk = within % a.chunks;
  if (b >= e.B) return;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int tok = lane & 15, kg = lane >> 4;
  const int t = __builtin_amdgcn_readfirstlane(load_index(a.t, b, a.t_bytes, a.tb.T));
  const bool need_post = t > 0;
  const char* tableF = need_post ? reinterpret_cast<const char*>(a.tb.qbar_f16x2) + (size_t)(t - 1) * RT_TABLE : nullptr;
  if (e.offset_dev) e.offset += *e.offset_dev;
  const long p1 = min(a.npos, (chunk + 1) * a.chunk_pos);
  __builtin_amdgcn_s_setprio(D3PM_VALU_PRIO);
  if (need_post) ring_begin<K>(tableF, ring);
  for (long tile0 = chunk * a.chunk_pos; tile0 < p1; tile0 += 64) {
    const long pos = tile0 + 16 * wave + tok;
    const bool live = pos < p1;
    const long n = (long)b * a.npos + (live ? pos : tile0);
    const int xt = load_index(a.xt, n, a.xt_bytes, K);
    const long row0 = (long)b * a.npos + tile0;
    const int rows = (int)min((long)64, p1 - tile0);
    const __amdgpu_buffer_rsrc_t zrs =
        __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(e.logits) + row0 * K, 0, rows * K * 4, 0x00020000);
    const int rowoff = (live ? 16 * wave + tok : 0) * (K * 4) + 16 * kg;
    auto load_row4 = [&](const __amdgpu_buffer_rsrc_t& rs, int off, int mt) {
      return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, off + 64 * mt, 0, 0));
    };
    f32x4 v[NMT];
#pragma unroll
    for (int mt = 0; mt < NMT; ++mt) v[mt] = load_row4(zrs, rowoff, mt);
    int pick;
    if (!need_post) {
      // argmax of the logits, first index on ties: element order within the lane is not class order across lanes, so
      // ties are broken on the class index explicitly
      float best = v[0][0];
      int bc = 4 * kg;
#pragma unroll
      for (int mt = 0; mt < NMT; ++mt)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int c = 16 * mt + 4 * kg + r;
          if (v[mt][r] > best) { best = v[mt][r]; bc = c; }       // (classes ascend within the lane: > keeps the first)
        }
#pragma unroll
      for (int d = 16; d <= 32; d <<= 1) {
        const float ob = __shfl_xor(best, d);
        const int oc = __shfl_xor(bc, d);
        if (ob > best || (ob == best && oc < bc)) { best = ob; bc = oc; }
      }
      pick = bc;
    } else {
      float m = v[0][0];
#pragma unroll
      for (int mt = 0; mt < NMT; ++mt)
#pragma unroll
        for (int r = 0; r < 4; ++r) m = fmaxf(m, v[mt][r]);
      m = quad_lanes_max(m);
      float sum = 0;
#pragma unroll
      for (int mt = 0; mt < NMT; ++mt)
#pragma unroll
        for (int r = 0; r < 4; ++r) { v[mt][r] = fast_exp2((v[mt][r] - m) * LOG2E); sum += v[mt][r]; }
      sum = quad_lanes_sum(sum);
      const float pscale = live ? fast_rcp(sum) * RT_SCALE : 0.0f;
#pragma unroll
      for (int mt = 0; mt < NMT; ++mt) v[mt] *= pscale;
      f32x4 f2[NMT];
      regtile_contract<K>(tableF, ring, v, f2);
      if (tile0 + 64 < p1) ring_begin<K>(tableF, ring);
      const __amdgpu_buffer_rsrc_t ars = __builtin_amdgcn_make_buffer_rsrc(
          const_cast<float*>(a.tb.qoneT) + (size_t)t * K * K, 0, K * K * 4, 0x00020000);
      const int aoff = xt * (K * 4) + 16 * kg;
      float lane_tot = 0;
#pragma unroll
      for (int mt = 0; mt < NMT; ++mt) {
        const f32x4 av = load_row4(ars, aoff, mt) + eps;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          f2[mt][r] = av[r] * fmaf(f2[mt][r], 1.0f / (RT_SCALE * RT_SCALE), eps);      // the class's weight
          lane_tot += f2[mt][r];
        }
      }
      // the four lanes' totals, in lane order
      float before = 0, total = 0;
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const float tk = __shfl(lane_tot, tok + 16 * k);
        total += tk;
        before += (k < kg) ? tk : 0.0f;
      }
      const U4 rnd = philox4x32_10((uint64_t)n, e.offset, e.seed);
      const float target = u01_open(rnd.x) * total;
      float run = before;
      int key = 0x7fffffff;
#pragma unroll
      for (int mt = 0; mt < NMT; ++mt)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          run += f2[mt][r];
          if (key == 0x7fffffff && f2[mt][r] > 0 && target < run) key = kg * (K / 4) + 4 * mt + r;
        }
      key = min(key, __shfl_xor(key, 16));
      key = min(key, __shfl_xor(key, 32));
      const int kk = key / (K / 4), el = key % (K / 4);
      pick = key == 0x7fffffff ? xt : 16 * (el >> 2) + 4 * kk + (el & 3);
    }
    if (kg == 0 && live) e.sample[n] = pick;
  }
}

template <int K>
static int launch_sample256(const ImgArgs& a, const DenseExtra& e, int B, hipStream_t s) {
  auto kern = k_dense_sample256<K>;
  const int sh = (RtCfg<K>::NS < RtCfg<K>::RING ? RtCfg<K>::NS : RtCfg<K>::RING) * RT_STAGE;
  hipError_t er = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, sh);
  if (er != hipSuccess) return hip_fail(er, "hipFuncSetAttribute(MaxDynamicSharedMemorySize)");
  DenseExtra ee = e;
  ee.B = B;
  hipLaunchKernelGGL(kern, dim3(((B + 7) / 8) * 8 * a.chunks), dim3(256), sh, s, a, ee);
  D3PM_LAUNCH_CHECK("k_dense_sample256");
  return D3PM_OK;
}

template <int K, bool GRAD>
static int launch_loss256(const ImgArgs& a, const DenseExtra& e, int B, hipStream_t s) {
  using C = RtCfg<K>;
  auto kern = k_dense_loss256<K, GRAD>;
  const int sh = (C::NS < C::RING ? C::NS : C::RING) * RT_STAGE;
  hipError_t er = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, sh);
  if (er != hipSuccess) return hip_fail(er, "hipFuncSetAttribute(MaxDynamicSharedMemorySize)");
  DenseExtra ee = e;
  ee.B = B;
  hipLaunchKernelGGL(kern, dim3(((B + 7) / 8) * 8 * a.chunks), dim3(256), sh, s, a, ee);
  D3PM_LAUNCH_CHECK("k_dense_loss256");
  return D3PM_OK;
}

}  // namespace d3pm
