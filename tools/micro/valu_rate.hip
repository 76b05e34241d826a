// Micro-benchmark: VALU issue rate per SIMD as a function of waves per SIMD.
// Each wave runs ITER iterations of 16 independent v_fma_f32 (or mixed with
// v_exp_f32 / DPP adds); the grid places `wps` waves on every SIMD of the chip.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

template <int MODE>
__global__ void k(float* out, int iters) {
  float a[16];
  for (int i = 0; i < 16; ++i) a[i] = threadIdx.x * 0.001f + i;
  const float b = 1.0001f, c = 0.5f;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      if (MODE == 0) a[i] = __builtin_fmaf(a[i], b, c);
      if (MODE == 1) a[i] = (i & 3) == 0 ? __builtin_amdgcn_exp2f(a[i]) : __builtin_fmaf(a[i], b, c);
      if (MODE == 2) {
        float p = __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, a[i]), 0xB1, 0xF, 0xF, true));
        a[i] = a[i] * b + p;
      }
    }
  }
  float s = 0;
  for (int i = 0; i < 16; ++i) s += a[i];
  if (s == 123.456f) out[0] = s;
}

template <int MODE>
void run(const char* name, int wps) {
  float* d; hipMalloc(&d, 4);
  const int iters = 2000;
  // blocks of 256 threads = 4 waves = one per SIMD; wps blocks per CU
  dim3 grid(256 * wps), block(256);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  k<MODE><<<grid, block>>>(d, iters);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  k<MODE><<<grid, block>>>(d, iters);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  const double instr_per_simd = (double)iters * 16 * wps * (MODE == 2 ? 2 : 1);
  printf("%-10s waves/SIMD=%d  %.1f us  -> %.2f ns per wave-instr per SIMD (%.2f cycles @2.4GHz)\n", name, wps,
         ms * 1e3, ms * 1e6 / instr_per_simd, ms * 1e6 / instr_per_simd * 2.4);
  hipFree(d);
}

int main() {
  for (int w : {1, 2, 4, 8}) run<0>("fma", w);
  for (int w : {1, 2, 4, 8}) run<1>("fma+exp", w);
  for (int w : {1, 2, 4, 8}) run<2>("fma+dpp", w);
  return 0;
}
