// development microbenchmark (round 6): where can the p.V kernel keep its outlier sums while the dense loop runs?
// Rates of the NATIVE add instructions at scattered addresses, one workgroup of 1024 lanes per CU (the geometry of the
// in-loop outlier evaluation) and two of 512:
//   mode 0: ds_add_f32   over 4096 floats  (16 KB)          -- lds_atomic.hip measured hipcc's atomicAdd(float) = a CAS loop
//   mode 1: ds_add_u32   over 4096 words
//   mode 2: ds_add_u64   over 4096 quad-words (32 KB)
//   mode 3: global_atomic_add_f32 (no return) into the workgroup's own 16 KB slab (stays in the XCD's L2)
//   mode 4: global_atomic_add_x2  (no return) into the workgroup's own 32 KB slab
// build: hipcc --offload-arch=gfx950 -O3 -o tools/ubench/atomic_rate tools/ubench/atomic_rate.hip
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>

template <int MODE, int NT>
__global__ __launch_bounds__(NT) void k(float *slabs, float *out, int iters) {
  __shared__ __attribute__((aligned(16))) unsigned char smem[32768];
  for (int i = threadIdx.x; i < 8192; i += NT) reinterpret_cast<uint32_t *>(smem)[i] = 0;
  __syncthreads();
  unsigned a = (threadIdx.x * 2654435761u) >> 20;   // 12 bits
  float *slab = slabs + (size_t)blockIdx.x * 8192;
  const float one = 1.0f;
  const unsigned long long one64 = 1ull;
  for (int it = 0; it < iters; it++) {
    a &= 4095u;
    if constexpr (MODE == 0) asm volatile("ds_add_f32 %0, %1" ::"v"(a * 4u), "v"(one) : "memory");
    if constexpr (MODE == 1) asm volatile("ds_add_u32 %0, %1" ::"v"(a * 4u), "v"(1u) : "memory");
    if constexpr (MODE == 2) asm volatile("ds_add_u64 %0, %1" ::"v"(a * 8u), "v"(one64) : "memory");
    if constexpr (MODE == 3) asm volatile("global_atomic_add_f32 %0, %1, %2" ::"v"(a * 4u), "v"(one), "s"(slab) : "memory");
    if constexpr (MODE == 4) asm volatile("global_atomic_add_x2 %0, %1, %2" ::"v"(a * 8u), "v"(one64), "s"(slab) : "memory");
    a = a * 1664525u + 1013904223u + (a >> 7);
  }
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
  __syncthreads();
  if (threadIdx.x == 0) out[blockIdx.x] = reinterpret_cast<float *>(smem)[0];
}

template <int MODE, int NT>
static void run(const char *name, float *slabs, float *out) {
  const int blocks = 256 * (1024 / NT), iters = 2000;
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  float ms = 0;
  for (int rep = 0; rep < 3; rep++) {
    hipEventRecord(e0);
    k<MODE, NT><<<blocks, NT>>>(slabs, out, iters);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    hipEventElapsedTime(&ms, e0, e1);
  }
  const double waveops_per_cu = 16.0 * iters;   // 16 waves per CU in both geometries
  printf("%-28s NT %4d  %8.3f ms  -> %6.1f ns per wave-instruction per CU (%.1f cycles at 2.1 GHz)\n", name, NT, ms,
         ms * 1e6 / waveops_per_cu, ms * 1e6 / waveops_per_cu * 2.1);
}

int main() {
  float *slabs, *out;
  hipMalloc(&slabs, (size_t)512 * 8192 * 4);
  hipMemset(slabs, 0, (size_t)512 * 8192 * 4);
  hipMalloc(&out, 4096);
  run<0, 1024>("ds_add_f32", slabs, out);
  run<1, 1024>("ds_add_u32", slabs, out);
  run<2, 1024>("ds_add_u64", slabs, out);
  run<3, 1024>("global_atomic_add_f32", slabs, out);
  run<4, 1024>("global_atomic_add_x2", slabs, out);
  run<0, 512>("ds_add_f32", slabs, out);
  run<2, 512>("ds_add_u64", slabs, out);
  run<3, 512>("global_atomic_add_f32", slabs, out);
  run<4, 512>("global_atomic_add_x2", slabs, out);
  return 0;
}
