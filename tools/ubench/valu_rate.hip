// microbenchmark: VALU issue cost per instruction type on gfx950 (development tool).
// 256 CUs x 4 waves/SIMD (grid 512 x 512 threads), each wave runs a loop of 64 independent instructions.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
typedef float f32x2 __attribute__((ext_vector_type(2)));

#define REP8(X) X X X X X X X X
template <int MODE>
__global__ __launch_bounds__(512, 2) void k(float *out, int iters, unsigned mask) {
  f32x2 a0 = {1.f, 2.f}, a1 = a0, a2 = a0, a3 = a0, a4 = a0, a5 = a0, a6 = a0, a7 = a0;
  f32x2 c = {0.5f, 0.25f}, d = {(float)threadIdx.x, 1.f};
  unsigned u0 = threadIdx.x, u1 = u0 + 1, u2 = u0 + 2, u3 = u0 + 3, u4 = u0 + 4, u5 = u0 + 5, u6 = u0 + 6, u7 = u0 + 7;
  unsigned w = threadIdx.x * 2654435761u;
  for (int it = 0; it < iters; it++) {
    if (MODE == 0) {   // v_fma_f32
      REP8(asm volatile("v_fma_f32 %0, %8, %9, %0\n v_fma_f32 %1, %8, %9, %1\n v_fma_f32 %2, %8, %9, %2\n v_fma_f32 %3, %8, %9, %3\n"
                        "v_fma_f32 %4, %8, %9, %4\n v_fma_f32 %5, %8, %9, %5\n v_fma_f32 %6, %8, %9, %6\n v_fma_f32 %7, %8, %9, %7"
                        : "+v"(a0.x), "+v"(a1.x), "+v"(a2.x), "+v"(a3.x), "+v"(a4.x), "+v"(a5.x), "+v"(a6.x), "+v"(a7.x) : "v"(c.x), "v"(d.x));)
    }
    if (MODE == 1) {   // v_pk_fma_f32
      REP8(asm volatile("v_pk_fma_f32 %0, %8, %9, %0\n v_pk_fma_f32 %1, %8, %9, %1\n v_pk_fma_f32 %2, %8, %9, %2\n v_pk_fma_f32 %3, %8, %9, %3\n"
                        "v_pk_fma_f32 %4, %8, %9, %4\n v_pk_fma_f32 %5, %8, %9, %5\n v_pk_fma_f32 %6, %8, %9, %6\n v_pk_fma_f32 %7, %8, %9, %7"
                        : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(c), "v"(d));)
    }
    if (MODE == 2) {   // v_bfe_u32 with immediates
      REP8(asm volatile("v_bfe_u32 %0, %8, 8, 8\n v_bfe_u32 %1, %8, 16, 8\n v_bfe_u32 %2, %8, 8, 8\n v_bfe_u32 %3, %8, 16, 8\n"
                        "v_bfe_u32 %4, %8, 8, 8\n v_bfe_u32 %5, %8, 16, 8\n v_bfe_u32 %6, %8, 8, 8\n v_bfe_u32 %7, %8, 16, 8"
                        : "=v"(u0), "=v"(u1), "=v"(u2), "=v"(u3), "=v"(u4), "=v"(u5), "=v"(u6), "=v"(u7) : "v"(w));)
    }
    if (MODE == 3) {   // v_and_or_b32 with SGPR mask
      REP8(asm volatile("v_and_or_b32 %0, %8, %9, %10\n v_and_or_b32 %1, %8, %9, %10\n v_and_or_b32 %2, %8, %9, %10\n v_and_or_b32 %3, %8, %9, %10\n"
                        "v_and_or_b32 %4, %8, %9, %10\n v_and_or_b32 %5, %8, %9, %10\n v_and_or_b32 %6, %8, %9, %10\n v_and_or_b32 %7, %8, %9, %10"
                        : "=v"(u0), "=v"(u1), "=v"(u2), "=v"(u3), "=v"(u4), "=v"(u5), "=v"(u6), "=v"(u7) : "v"(w), "s"(mask), "v"(d.x));)
    }
    if (MODE == 4) {   // v_and_b32 (VOP2, literal)
      REP8(asm volatile("v_and_b32 %0, 0xff, %8\n v_and_b32 %1, 0xff, %8\n v_and_b32 %2, 0xff, %8\n v_and_b32 %3, 0xff, %8\n"
                        "v_and_b32 %4, 0xff, %8\n v_and_b32 %5, 0xff, %8\n v_and_b32 %6, 0xff, %8\n v_and_b32 %7, 0xff, %8"
                        : "=v"(u0), "=v"(u1), "=v"(u2), "=v"(u3), "=v"(u4), "=v"(u5), "=v"(u6), "=v"(u7) : "v"(w));)
    }
    if (MODE == 5) {   // v_pk_mul_f32
      REP8(asm volatile("v_pk_mul_f32 %0, %8, %0\n v_pk_mul_f32 %1, %8, %1\n v_pk_mul_f32 %2, %8, %2\n v_pk_mul_f32 %3, %8, %3\n"
                        "v_pk_mul_f32 %4, %8, %4\n v_pk_mul_f32 %5, %8, %5\n v_pk_mul_f32 %6, %8, %6\n v_pk_mul_f32 %7, %8, %7"
                        : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(c));)
    }
    if (MODE == 6) {   // v_dot2_f32_f16 (fp16 pairs, fp32 accumulate)
      REP8(asm volatile("v_dot2_f32_f16 %0, %8, %9, %0\n v_dot2_f32_f16 %1, %8, %9, %1\n v_dot2_f32_f16 %2, %8, %9, %2\n v_dot2_f32_f16 %3, %8, %9, %3\n"
                        "v_dot2_f32_f16 %4, %8, %9, %4\n v_dot2_f32_f16 %5, %8, %9, %5\n v_dot2_f32_f16 %6, %8, %9, %6\n v_dot2_f32_f16 %7, %8, %9, %7"
                        : "+v"(a0.x), "+v"(a1.x), "+v"(a2.x), "+v"(a3.x), "+v"(a4.x), "+v"(a5.x), "+v"(a6.x), "+v"(a7.x) : "v"(w), "v"(u0));)
    }
    if (MODE == 7) {   // v_lshlrev_b32 sdwa byte select
      REP8(asm volatile("v_lshlrev_b32_sdwa %0, %9, %8 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_1\n"
                        "v_lshlrev_b32_sdwa %1, %9, %8 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_2\n"
                        "v_lshlrev_b32_sdwa %2, %9, %8 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_1\n"
                        "v_lshlrev_b32_sdwa %3, %9, %8 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_2\n"
                        "v_lshlrev_b32_sdwa %4, %9, %8 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_1\n"
                        "v_lshlrev_b32_sdwa %5, %9, %8 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_2\n"
                        "v_lshlrev_b32_sdwa %6, %9, %8 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_1\n"
                        "v_lshlrev_b32_sdwa %7, %9, %8 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_2"
                        : "=v"(u0), "=v"(u1), "=v"(u2), "=v"(u3), "=v"(u4), "=v"(u5), "=v"(u6), "=v"(u7) : "v"(w), "v"(3u));)
    }
    if (MODE == 8) {   // alternating bfe / pk_fma (the kernel's mix)
      REP8(asm volatile("v_bfe_u32 %8, %16, 8, 8\n v_pk_fma_f32 %0, %17, %18, %0\n v_bfe_u32 %9, %16, 16, 8\n v_pk_fma_f32 %1, %17, %18, %1\n"
                        "v_bfe_u32 %10, %16, 8, 8\n v_pk_fma_f32 %2, %17, %18, %2\n v_bfe_u32 %11, %16, 16, 8\n v_pk_fma_f32 %3, %17, %18, %3"
                        : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7),
                          "=v"(u0), "=v"(u1), "=v"(u2), "=v"(u3), "=v"(u4), "=v"(u5), "=v"(u6), "=v"(u7) : "v"(w), "v"(c), "v"(d));)
    }
  }
  f32x2 s = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7;
  unsigned us = u0 ^ u1 ^ u2 ^ u3 ^ u4 ^ u5 ^ u6 ^ u7;
  if (s.x + s.y + (float)us == 12345.678f) out[0] = s.x;
}
template <int MODE>
static void run(const char *name, float *d) {
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  int iters = 2000, blocks = 512;
  float ms = 0;
  for (int rep = 0; rep < 2; rep++) {
    hipEventRecord(e0);
    k<MODE><<<blocks, 512>>>(d, iters, 0x78787878u);
    hipEventRecord(e1); hipEventSynchronize(e1);
    hipEventElapsedTime(&ms, e0, e1);
  }
  // per SIMD: 4 waves x iters x 64 instructions
  double inst_per_simd = 4.0 * iters * 64.0;
  printf("%-28s %8.3f ms -> %.2f ns per wave-instruction per SIMD (= %.2f clk @2.4 GHz)\n", name, ms,
         ms * 1e6 / inst_per_simd, ms * 1e6 / inst_per_simd * 2.4);
}
int main() {
  float *d; hipMalloc(&d, 4096);
  run<0>("v_fma_f32", d); run<1>("v_pk_fma_f32", d); run<2>("v_bfe_u32 imm", d); run<3>("v_and_or_b32 sgpr", d);
  run<4>("v_and_b32 literal", d); run<5>("v_pk_mul_f32", d); run<6>("v_dot2_f32_f16", d); run<7>("v_lshlrev_b32_sdwa byte", d);
  run<8>("bfe/pk_fma alternating", d);
  return 0;
}
