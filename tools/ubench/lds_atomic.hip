// microbenchmark: LDS atomic-add throughput by type and conflict pattern (development tool)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
template <int MODE>
__global__ __launch_bounds__(512) void k(float *out, int iters, int stride) {
  __shared__ float sf[8192];
  __shared__ unsigned long long s64[4096];
  unsigned *su = reinterpret_cast<unsigned *>(sf);
  for (int i = threadIdx.x; i < 8192; i += 512) sf[i] = 0.f;
  for (int i = threadIdx.x; i < 4096; i += 512) s64[i] = 0;
  __syncthreads();
  unsigned a = (threadIdx.x * stride) & 8191;
  for (int it = 0; it < iters; it++) {
    if (MODE == 0) atomicAdd(&sf[a], 1.0f);
    if (MODE == 1) atomicAdd(&su[a], 1u);
    if (MODE == 2) atomicAdd(&s64[a & 4095], 1ull);
    if (MODE == 3) sf[a] += 1.0f;                       // plain RMW
    if (MODE == 4) __hip_atomic_fetch_add(&sf[a], 1.0f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    a = (a * 1664525u + 1013904223u) & 8191;            // pseudo-random next address
  }
  __syncthreads();
  if (threadIdx.x == 0) out[blockIdx.x] = sf[0] + (float)s64[0];
}
int main() {
  float *d; hipMalloc(&d, 4096 * 4);
  const char *names[] = {"ds_add_f32", "ds_add_u32", "ds_add_u64", "plain rmw f32", "hip_atomic f32 wg"};
  for (int mode = 0; mode < 5; mode++) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    int iters = 2000, blocks = 512;
    for (int rep = 0; rep < 2; rep++) {
      hipEventRecord(e0);
      if (mode == 0) k<0><<<blocks, 512>>>(d, iters, 1);
      if (mode == 1) k<1><<<blocks, 512>>>(d, iters, 1);
      if (mode == 2) k<2><<<blocks, 512>>>(d, iters, 1);
      if (mode == 3) k<3><<<blocks, 512>>>(d, iters, 1);
      if (mode == 4) k<4><<<blocks, 512>>>(d, iters, 1);
      hipEventRecord(e1); hipEventSynchronize(e1);
    }
    float ms; hipEventElapsedTime(&ms, e0, e1);
    double waveops = (double)blocks * 8 * iters;       // wave-instructions
    printf("%-20s %8.3f ms  -> %.1f ns per wave-op per CU-slot (%.2f cycles/lane @2.1GHz, 2 WG/CU)\n", names[mode], ms,
           ms * 1e6 / (waveops / 256.0), ms * 1e6 / (waveops / 256.0) * 2.1 / 64.0);
  }
  return 0;
}
