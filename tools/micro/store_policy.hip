// Micro-benchmark: does the cache policy of the output stores change the duration of a streaming kernel with the
// fused loss kernel's traffic (B=256, L=512, K=32: 16.8 MB in, 16.8 MB out, back-to-back launches)?  Dirty lines
// left in the XCDs' L2 are written back when the kernel ends; write-through stores move that traffic into the
// kernel's own run time.
//   hipcc -O3 --offload-arch=gfx950 store_policy.hip -o store_policy.bin
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>

constexpr int B = 256, L = 512, K = 32;
typedef float v4f __attribute__((ext_vector_type(4)));
enum { PLAIN, NT, SC1, SC0SC1, NT_SC1 };

template <int MODE>
__device__ __forceinline__ void store4(float* p, float4 v_) {
  const v4f v = {v_.x, v_.y, v_.z, v_.w};
  if constexpr (MODE == PLAIN) *reinterpret_cast<v4f*>(p) = v;
  else if constexpr (MODE == NT) __builtin_nontemporal_store(v, reinterpret_cast<v4f*>(p));
  else if constexpr (MODE == SC1) asm volatile("global_store_dwordx4 %0, %1, off sc1" ::"v"(p), "v"(v) : "memory");
  else if constexpr (MODE == SC0SC1) asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1" ::"v"(p), "v"(v) : "memory");
  else asm volatile("global_store_dwordx4 %0, %1, off sc1 nt" ::"v"(p), "v"(v) : "memory");
}

template <int MODE, bool NTLOAD>
__global__ __launch_bounds__(512) void k_rows(const float* __restrict__ in, const int64_t* __restrict__ x0,
                                               const int64_t* __restrict__ xt, float* __restrict__ out) {
  const int b = blockIdx.x, g = threadIdx.x & 3, slot = threadIdx.x >> 2;
  for (int pos = slot; pos < L; pos += 128) {
    const size_t n = (size_t)b * L + pos;
    float4 u, v;
    if (NTLOAD) {
      const v4f uu = __builtin_nontemporal_load(reinterpret_cast<const v4f*>(in + n * K + 8 * g));
      const v4f vv = __builtin_nontemporal_load(reinterpret_cast<const v4f*>(in + n * K + 8 * g + 4));
      u = make_float4(uu.x, uu.y, uu.z, uu.w); v = make_float4(vv.x, vv.y, vv.z, vv.w);
    } else {
      u = *reinterpret_cast<const float4*>(in + n * K + 8 * g);
      v = *reinterpret_cast<const float4*>(in + n * K + 8 * g + 4);
    }
    const float s = (float)((int)x0[n] + (int)xt[n]);
    u.x += s;
    store4<MODE>(out + n * K + 8 * g, u);
    store4<MODE>(out + n * K + 8 * g + 4, v);
  }
}

int main() {
  const size_t N = (size_t)B * L;
  constexpr int SETS = 10;
  float *in[SETS], *out[SETS]; int64_t *x0, *xt;
  for (int i = 0; i < SETS; ++i) { hipMalloc(&in[i], N * K * 4); hipMalloc(&out[i], N * K * 4); hipMemset(in[i], 0, N * K * 4); }
  hipMalloc(&x0, N * 8); hipMalloc(&xt, N * 8);
  hipMemset(x0, 0, N * 8); hipMemset(xt, 0, N * 8);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  const double bytes = (double)N * (8 * K + 16);
  auto time = [&](const char* name, auto launch) {
    for (int sets : {1, SETS}) {
      for (int i = 0; i < 20; ++i) launch(i % sets);
      hipDeviceSynchronize();
      hipEventRecord(e0);
      for (int i = 0; i < 200; ++i) launch(i % sets);
      hipEventRecord(e1); hipEventSynchronize(e1);
      float ms; hipEventElapsedTime(&ms, e0, e1);
      printf("%-26s %2d buffer set(s): %.2f us per launch back to back  (%.0f GB/s)\n", name, sets, ms * 1e3 / 200,
             bytes / (ms * 1e-3 / 200) / 1e9);
    }
  };
  time("plain stores", [&](int s) { k_rows<PLAIN, false><<<B, 512>>>(in[s], x0, xt, out[s]); });
  time("nt stores", [&](int s) { k_rows<NT, false><<<B, 512>>>(in[s], x0, xt, out[s]); });
  time("sc1 stores", [&](int s) { k_rows<SC1, false><<<B, 512>>>(in[s], x0, xt, out[s]); });
  time("sc0 sc1 stores", [&](int s) { k_rows<SC0SC1, false><<<B, 512>>>(in[s], x0, xt, out[s]); });
  time("sc1 nt stores", [&](int s) { k_rows<NT_SC1, false><<<B, 512>>>(in[s], x0, xt, out[s]); });
  time("nt loads, plain stores", [&](int s) { k_rows<PLAIN, true><<<B, 512>>>(in[s], x0, xt, out[s]); });
  time("nt loads, sc1 stores", [&](int s) { k_rows<SC1, true><<<B, 512>>>(in[s], x0, xt, out[s]); });
  time("nt loads, nt stores", [&](int s) { k_rows<NT, true><<<B, 512>>>(in[s], x0, xt, out[s]); });
  return 0;
}
