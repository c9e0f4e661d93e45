// development microbenchmark: how fast does a CU take the score kernel's packed-word traffic as a function of the load
// width?  Same grid (512 workgroups x 8 waves, two per CU), same lines per wave and head (2 x 8 segments of 128 B), same
// look-ahead (the loads of two heads in flight), no other work:
//   mode 0: 16 x global_load_dword per wave and head-pair... i.e. 8 per head, 256 B per instruction (what score_k issues)
//   mode 1: 4 x global_load_dwordx2 per head (512 B per instruction, 2 tokens per lane)
//   mode 2: 2 x global_load_dwordx4 per head (1 KiB per instruction, 4 tokens per lane)
// build: hipcc --offload-arch=gfx950 -O3 -o tools/ubench/load_rate tools/ubench/load_rate.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
#include <algorithm>

constexpr int H = 32, ROWS = 16;

template <int MODE>
__global__ __launch_bounds__(512, 4) void k(const uint32_t *__restrict__ mat, int64_t max_len, uint32_t *sink) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int64_t tok0 = (int64_t)blockIdx.x * 256 + wave * 32;
  uint32_t acc = 0;
  if constexpr (MODE == 0) {
    const int role = lane >> 5;
    const uint32_t off = (uint32_t)(((int64_t)role * 8 * max_len + tok0 + (lane & 31)) * 4);
    uint32_t r[2][8];
    auto issue = [&](int h, uint32_t (&d)[8]) {
      const uint32_t *b = mat + (int64_t)h * ROWS * max_len;
#pragma unroll
      for (int j = 0; j < 8; j++) asm volatile("global_load_dword %0, %1, %2 nt" : "=v"(d[j]) : "v"(off), "s"(b + j * max_len) : "memory");
    };
    issue(0, r[0]); issue(1, r[1]);
    for (int h = 0; h < H; h += 2) {
      asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
#pragma unroll
      for (int j = 0; j < 8; j++) { asm volatile("" : "+v"(r[0][j])); acc ^= r[0][j]; }
      issue(h + 2 < H ? h + 2 : h, r[0]);
      asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
#pragma unroll
      for (int j = 0; j < 8; j++) { asm volatile("" : "+v"(r[1][j])); acc ^= r[1][j]; }
      issue(h + 3 < H ? h + 3 : h + 1, r[1]);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
    for (int j = 0; j < 8; j++) asm volatile("" ::"v"(r[0][j]), "v"(r[1][j]));
  } else {
    constexpr int W = MODE == 1 ? 2 : 4;          // dwords per lane
    constexpr int LPR = 32 / W;                   // lanes per 128-byte row segment
    constexpr int RPI = 64 / LPR;                 // row segments per instruction
    constexpr int NI = 16 / RPI;                  // instructions per head
    typedef uint32_t vec __attribute__((ext_vector_type(W)));
    const uint32_t off = (uint32_t)(((int64_t)(lane / LPR) * max_len + tok0 + (lane % LPR) * W) * 4);
    vec r[2][NI];
    auto issue = [&](int h, vec (&d)[NI]) {
      const uint32_t *b = mat + (int64_t)h * ROWS * max_len;
#pragma unroll
      for (int j = 0; j < NI; j++) {
        if constexpr (W == 2) asm volatile("global_load_dwordx2 %0, %1, %2 nt" : "=v"(d[j]) : "v"(off), "s"(b + (int64_t)j * RPI * max_len) : "memory");
        else asm volatile("global_load_dwordx4 %0, %1, %2 nt" : "=v"(d[j]) : "v"(off), "s"(b + (int64_t)j * RPI * max_len) : "memory");
      }
    };
    auto use = [&](vec (&d)[NI]) {
#pragma unroll
      for (int j = 0; j < NI; j++) { asm volatile("" : "+v"(d[j])); for (int c = 0; c < W; c++) acc ^= d[j][c]; }
    };
    issue(0, r[0]); issue(1, r[1]);
    for (int h = 0; h < H; h += 2) {
      if constexpr (NI == 4) asm volatile("s_waitcnt vmcnt(4)" ::: "memory"); else asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
      use(r[0]);
      issue(h + 2 < H ? h + 2 : h, r[0]);
      if constexpr (NI == 4) asm volatile("s_waitcnt vmcnt(4)" ::: "memory"); else asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
      use(r[1]);
      issue(h + 3 < H ? h + 3 : h + 1, r[1]);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
    for (int j = 0; j < NI; j++) asm volatile("" ::"v"(r[0][j]), "v"(r[1][j]));
  }
  if (acc == 0x12345u) sink[0] = acc;
}

int main() {
  const int64_t L = 131072, max_len = L + 128;
  const size_t bytes = (size_t)H * ROWS * max_len * 4;
  uint32_t *buf[2], *sink;
  for (auto &b : buf) { hipMalloc(&b, bytes); hipMemset(b, 1, bytes); }
  hipMalloc(&sink, 64);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int rep = 0; rep < 2; rep++)
  for (int mode = 0; mode < 3; mode++) {
    std::vector<float> ts;
    for (int i = 0; i < 60; i++) {
      hipEventRecord(e0, 0);
      if (mode == 0) k<0><<<512, 512>>>(buf[i & 1], max_len, sink);
      if (mode == 1) k<1><<<512, 512>>>(buf[i & 1], max_len, sink);
      if (mode == 2) k<2><<<512, 512>>>(buf[i & 1], max_len, sink);
      hipEventRecord(e1, 0); hipEventSynchronize(e1);
      float ms; hipEventElapsedTime(&ms, e0, e1); if (i >= 10) ts.push_back(ms * 1000);
    }
    std::sort(ts.begin(), ts.end());
    const double mb = (double)H * ROWS * L * 4 / 1e6;
    printf("mode %d (%s): median %.1f us (min %.1f)  %.0f GB/s for %.0f MB\n", mode, mode == 0 ? "dword" : mode == 1 ? "dwordx2" : "dwordx4",
           ts[ts.size() / 2], ts[0], mb / ts[ts.size() / 2] * 1e3 / 1e3, mb);
  }
  return 0;
}
