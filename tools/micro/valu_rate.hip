// Micro-benchmark: VALU issue rate per SIMD as a function of waves per SIMD and instruction kind.
// Each wave runs ITER iterations of 16 independent operations; the grid places `wps` waves on every SIMD of the chip.
//   hipcc -O3 --offload-arch=gfx950 valu_rate.hip -o valu_rate.bin
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

typedef float f2 __attribute__((ext_vector_type(2)));

enum { FMA, PKFMA, PKMUL, PKADD, EXP, RCP, LOG, DPPADD, FMA_EXP4, CNDMASK, FMA_DEP };

template <int MODE>
__global__ void k(float* out, int iters) {
  float a[16];
  f2 p[16];
  for (int i = 0; i < 16; ++i) { a[i] = threadIdx.x * 0.001f + i; p[i] = f2{a[i], a[i] + 0.5f}; }
  const float b = 1.0001f, c = 0.5f;
  const f2 b2 = {1.0001f, 0.9999f}, c2 = {0.5f, 0.25f};
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      if (MODE == FMA) a[i] = __builtin_fmaf(a[i], b, c);
      if (MODE == FMA_DEP) a[0] = __builtin_fmaf(a[0], b, c);
      if (MODE == PKFMA) asm volatile("v_pk_fma_f32 %0, %1, %2, %3" : "=v"(p[i]) : "v"(p[i]), "v"(b2), "v"(c2));
      if (MODE == PKMUL) asm volatile("v_pk_mul_f32 %0, %1, %2" : "=v"(p[i]) : "v"(p[i]), "v"(b2));
      if (MODE == PKADD) asm volatile("v_pk_add_f32 %0, %1, %2" : "=v"(p[i]) : "v"(p[i]), "v"(c2));
      if (MODE == EXP) a[i] = __builtin_amdgcn_exp2f(a[i]);
      if (MODE == RCP) a[i] = __builtin_amdgcn_rcpf(a[i]);
      if (MODE == LOG) a[i] = __builtin_amdgcn_logf(a[i]);
      if (MODE == FMA_EXP4) a[i] = (i & 3) == 0 ? __builtin_amdgcn_exp2f(a[i]) : __builtin_fmaf(a[i], b, c);
      if (MODE == DPPADD) {
        float q = __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, a[i]), 0xB1, 0xF, 0xF, true));
        a[i] = a[i] + q;
      }
      if (MODE == CNDMASK) a[i] = (a[i] > c) ? a[i] - b : a[i] + b;
    }
  }
  float s = 0;
  for (int i = 0; i < 16; ++i) s += a[i] + p[i].x + p[i].y;
  if (s == 123.456f) out[0] = s;
}

template <int MODE>
void run(const char* name, int per_iter) {
  float* d; hipMalloc(&d, 4);
  const int iters = 2000;
  printf("%-28s", name);
  for (int wps : {1, 2, 4, 8}) {
    dim3 grid(256 * wps), block(256);     // blocks of 256 threads = 4 waves = one per SIMD; wps blocks per CU
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    k<MODE><<<grid, block>>>(d, iters);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    k<MODE><<<grid, block>>>(d, iters);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double instr_per_simd = (double)iters * per_iter * wps;
    printf("  w%d: %5.2f cyc", wps, ms * 1e6 / instr_per_simd * 2.4);
  }
  printf("   (cycles @2.4 GHz per wave-instruction per SIMD)\n");
  hipFree(d);
}

int main() {
  run<FMA>("v_fma_f32", 16);
  run<FMA_DEP>("v_fma_f32 dependent chain", 16);
  run<PKFMA>("v_pk_fma_f32", 16);
  run<PKMUL>("v_pk_mul_f32", 16);
  run<PKADD>("v_pk_add_f32", 16);
  run<EXP>("v_exp_f32", 16);
  run<RCP>("v_rcp_f32", 16);
  run<LOG>("v_log_f32", 16);
  run<FMA_EXP4>("3 fma + 1 exp", 16);
  run<DPPADD>("v_add_f32 dpp (quad_perm)", 16);
  run<CNDMASK>("cmp + cndmask + add/sub", 16);
  return 0;
}
