// Micro-benchmark: issue rate of v_mfma_f32_32x32x2_f32 from one or two waves per SIMD,
// with 2 or 4 independent accumulator chains, alone and with one global load per MFMA.
//   hipcc -O3 --offload-arch=gfx950 mfma_rate.hip -o mfma_rate.bin
#include <hip/hip_runtime.h>
#include <cstdio>

typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int CHAINS, bool LOADS>
__global__ __launch_bounds__(256) void k(const float* __restrict__ src, float* out, int iters) {
  f32x16 acc[CHAINS];
  for (int c = 0; c < CHAINS; ++c)
    for (int e = 0; e < 16; ++e) acc[c][e] = 0.0f;
  float a = threadIdx.x * 0.001f, b = 1.0f;
  const float* p = src + threadIdx.x;
  float b0[16], b1[16];
  auto load = [&](float (&bl)[16], int it) {
#pragma unroll
    for (int q = 0; q < 16; ++q) bl[q] = p[(size_t)((it * 16 + q) & 1023) * 256];
  };
  auto mfmas = [&](const float (&bl)[16]) {
#pragma unroll
    for (int q = 0; q < 16; ++q) {
      const float bb = LOADS ? bl[q] : b;
      acc[q % CHAINS] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, bb, acc[q % CHAINS], 0, 0, 0);
    }
  };
  auto interleave = [&]() {
    if (LOADS) {
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
        __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
      }
      __builtin_amdgcn_sched_barrier(0);
    }
  };
  if (LOADS) load(b0, 0);
#pragma unroll 1
  for (int it = 0; it < iters; it += 2) {       // operands double-buffered one group of 16 ahead
    if (LOADS) load(b1, it + 1);
    mfmas(b0);
    interleave();
    if (LOADS) load(b0, it + 2);
    mfmas(b1);
    interleave();
  }
  float s = 0;
  for (int c = 0; c < CHAINS; ++c)
    for (int e = 0; e < 16; ++e) s += acc[c][e];
  if (s == 123.456f) out[0] = s;
}

template <int CHAINS, bool LOADS>
void run(const char* name, int blocks_per_cu, const float* src, float* out) {
  const int iters = 2000;
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  k<CHAINS, LOADS><<<256 * blocks_per_cu, 256>>>(src, out, iters);
  (void)hipDeviceSynchronize();
  (void)hipEventRecord(e0);
  k<CHAINS, LOADS><<<256 * blocks_per_cu, 256>>>(src, out, iters);
  (void)hipEventRecord(e1);
  (void)hipEventSynchronize(e1);
  float ms;
  (void)hipEventElapsedTime(&ms, e0, e1);
  const double mfma_per_simd = (double)iters * 16 * blocks_per_cu;
  const double tflops = 256.0 * 4 * mfma_per_simd * 4096 / (ms * 1e-3) / 1e12;
  printf("%-26s waves/SIMD=%d  %8.1f us  %.1f ns per MFMA per SIMD  %.1f TFLOP/s\n", name, blocks_per_cu, ms * 1e3,
         ms * 1e6 / mfma_per_simd, tflops);
}

int main() {
  float *src, *out;
  (void)hipMalloc(&src, 1024 * 256 * 4 + 4096);
  (void)hipMemset(src, 0, 1024 * 256 * 4 + 4096);
  (void)hipMalloc(&out, 4);
  for (int w : {1, 2}) run<2, false>("2 chains", w, src, out);
  for (int w : {1, 2}) run<4, false>("4 chains", w, src, out);
  for (int w : {1, 2}) run<2, true>("2 chains + load per MFMA", w, src, out);
  for (int w : {1, 2}) run<4, true>("4 chains + load per MFMA", w, src, out);
  return 0;
}
