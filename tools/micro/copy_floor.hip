// Micro-benchmark: the practical floor of a kernel with the fused loss kernel's
// algorithmic traffic at the protein benchmark shape.  Per token it reads a
// float32 row of K classes and two int64 tokens and writes a float32 row of K
// classes, with no arithmetic to speak of; grid shapes as the loss kernel's
// (one workgroup per batch element) and as a flat grid-stride copy.
//   hipcc -O3 --offload-arch=gfx950 copy_floor.hip -o copy_floor.bin
//   rocprofv3 --kernel-trace --stats -- ./copy_floor.bin
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>

constexpr int B = 256, L = 512, K = 32;

// one workgroup of 1024 threads per batch element, 4 lanes x 32 bytes per token (as the loss kernel)
__global__ __launch_bounds__(1024) void k_rows_per_sample(const float* __restrict__ in, const int64_t* __restrict__ x0,
                                                          const int64_t* __restrict__ xt, float* __restrict__ out) {
  const int b = blockIdx.x, g = threadIdx.x & 3, slot = threadIdx.x >> 2;
  for (int pos = slot; pos < L; pos += 256) {
    const size_t n = (size_t)b * L + pos;
    const float4 u = *reinterpret_cast<const float4*>(in + n * K + 8 * g);
    const float4 v = *reinterpret_cast<const float4*>(in + n * K + 8 * g + 4);
    const float s = (float)((int)x0[n] + (int)xt[n]);
    *reinterpret_cast<float4*>(out + n * K + 8 * g) = make_float4(u.x + s, u.y, u.z, u.w);
    *reinterpret_cast<float4*>(out + n * K + 8 * g + 4) = v;
  }
}

// flat copy, one float4 per thread per step, 2048 workgroups of 256
__global__ __launch_bounds__(256) void k_flat(const float4* __restrict__ in, const int64_t* __restrict__ x0,
                                              const int64_t* __restrict__ xt, float4* __restrict__ out, size_t n4) {
  for (size_t i = blockIdx.x * 256ull + threadIdx.x; i < n4; i += gridDim.x * 256ull) {
    float4 u = in[i];
    if ((i & 7) == 0) u.x += (float)((int)x0[i >> 3] + (int)xt[i >> 3]);
    out[i] = u;
  }
}

__global__ void k_empty(float* p) { if (p == nullptr) *p = 0; }

int main() {
  const size_t N = (size_t)B * L;
  float *in, *out; int64_t *x0, *xt;
  hipMalloc(&in, N * K * 4); hipMalloc(&out, N * K * 4); hipMalloc(&x0, N * 8); hipMalloc(&xt, N * 8);
  hipMemset(in, 0, N * K * 4); hipMemset(x0, 0, N * 8); hipMemset(xt, 0, N * 8);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  const double bytes = (double)N * (8 * K + 16);
  auto time = [&](const char* name, auto launch) {
    for (int i = 0; i < 20; ++i) launch();
    hipDeviceSynchronize();
    hipEventRecord(e0);
    for (int i = 0; i < 200; ++i) launch();
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    printf("%-20s %.2f us per launch back to back  (%.0f GB/s)\n", name, ms * 1e3 / 200, bytes / (ms * 1e-3 / 200) / 1e9);
  };
  time("rows_per_sample", [&] { k_rows_per_sample<<<B, 1024>>>(in, x0, xt, out); });
  time("flat_2048x256", [&] { k_flat<<<2048, 256>>>((const float4*)in, x0, xt, (float4*)out, N * K / 4); });
  time("flat_1024x256", [&] { k_flat<<<1024, 256>>>((const float4*)in, x0, xt, (float4*)out, N * K / 4); });
  time("empty", [&] { k_empty<<<256, 1024>>>(out); });
  // the same copies over ten disjoint sets of buffers in rotation (357 MB: more than the 256 MiB Infinity Cache holds,
  // every launch reads its rows from HBM proper) -- the floor of the figure the benchmark's roofline.frac is quoted on
  constexpr int SETS = 10;
  float *ins[SETS], *outs[SETS]; int64_t *x0s[SETS], *xts[SETS];
  for (int i = 0; i < SETS; ++i) {
    hipMalloc(&ins[i], N * K * 4); hipMalloc(&outs[i], N * K * 4); hipMalloc(&x0s[i], N * 8); hipMalloc(&xts[i], N * 8);
    hipMemset(ins[i], 0, N * K * 4); hipMemset(x0s[i], 0, N * 8); hipMemset(xts[i], 0, N * 8);
  }
  int r = 0;
  time("rows_per_sample/rot", [&] { k_rows_per_sample<<<B, 1024>>>(ins[r], x0s[r], xts[r], outs[r]); r = (r + 1) % SETS; });
  time("flat_2048x256/rot", [&] { k_flat<<<2048, 256>>>((const float4*)ins[r], x0s[r], xts[r], (float4*)outs[r], N * K / 4); r = (r + 1) % SETS; });
  time("flat_1024x256/rot", [&] { k_flat<<<1024, 256>>>((const float4*)ins[r], x0s[r], xts[r], (float4*)outs[r], N * K / 4); r = (r + 1) % SETS; });
  return 0;
}
