// microbenchmark 2: VALU cost per instruction form on gfx950 (development tool).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
typedef float f32x2 __attribute__((ext_vector_type(2)));
#define REP8(X) X X X X X X X X
// 8 instructions, integer destinations u0..u7 (%0..%7), sources: %8 = w (vgpr), %9 = x (vgpr), %10 = sgpr
#define U8(I) asm volatile(I(0) I(1) I(2) I(3) I(4) I(5) I(6) I(7) : "=v"(u0), "=v"(u1), "=v"(u2), "=v"(u3), "=v"(u4), "=v"(u5), "=v"(u6), "=v"(u7) : "v"(w), "v"(x), "s"(sg));
// 8 instructions, float accumulators f0..f7 (%0..%7, read-write), sources %8 = a, %9 = b (vgpr floats)
#define F8(I) asm volatile(I(0) I(1) I(2) I(3) I(4) I(5) I(6) I(7) : "+v"(f0), "+v"(f1), "+v"(f2), "+v"(f3), "+v"(f4), "+v"(f5), "+v"(f6), "+v"(f7) : "v"(fa), "v"(fb));
#define P8(I) asm volatile(I(0) I(1) I(2) I(3) I(4) I(5) I(6) I(7) : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3), "+v"(p4), "+v"(p5), "+v"(p6), "+v"(p7) : "v"(pa), "v"(pb));

#define I_LSHR(n) "v_lshrrev_b32 %" #n ", 5, %8\n"
#define I_ANDI(n) "v_and_b32 %" #n ", 15, %8\n"
#define I_ANDV(n) "v_and_b32 %" #n ", %9, %8\n"
#define I_ANDS(n) "v_and_b32 %" #n ", %10, %8\n"
#define I_OR(n) "v_or_b32 %" #n ", %9, %8\n"
#define I_ADDU(n) "v_add_u32 %" #n ", %9, %8\n"
#define I_LSHLADD(n) "v_lshl_add_u32 %" #n ", %8, 3, %9\n"
#define I_PERM(n) "v_perm_b32 %" #n ", %8, %9, %10\n"
#define I_MAD24(n) "v_mad_u32_u24 %" #n ", %8, 8, %9\n"
#define I_ALIGN(n) "v_alignbit_b32 %" #n ", %8, %9, 5\n"
#define I_BFE(n) "v_bfe_u32 %" #n ", %8, 8, 8\n"
#define I_BFEV(n) "v_bfe_u32 %" #n ", %8, %9, %9\n"
#define I_ANDOR(n) "v_and_or_b32 %" #n ", %8, %10, %9\n"
#define I_LSHLOR(n) "v_lshl_or_b32 %" #n ", %8, 3, %9\n"
#define I_CVTUB(n) "v_cvt_f32_ubyte1 %" #n ", %8\n"
#define I_MOV(n) "v_mov_b32 %" #n ", %8\n"
#define I_XOR3(n) "v_xor_b32 %" #n ", %8, %9\n"
#define I_ADD3(n) "v_add3_u32 %" #n ", %8, %9, %9\n"
#define I_FMAC(n) "v_fmac_f32 %" #n ", %8, %9\n"
#define I_FMA(n) "v_fma_f32 %" #n ", %8, %9, %" #n "\n"
#define I_MUL(n) "v_mul_f32 %" #n ", %8, %" #n "\n"
#define I_ADDF(n) "v_add_f32 %" #n ", %8, %" #n "\n"
#define I_SIN(n) "v_sin_f32 %" #n ", %" #n "\n"
#define I_PKFMA(n) "v_pk_fma_f32 %" #n ", %8, %9, %" #n "\n"
#define I_PKADD(n) "v_pk_add_f32 %" #n ", %8, %" #n "\n"
#define I_PKFMA16(n) "v_pk_fma_f16 %" #n ", %8, %9, %" #n "\n"
#define I_DOT2C(n) "v_dot2c_f32_f16 %" #n ", %8, %9\n"
#define I_ANDLIT(n) "v_and_b32 %" #n ", 0x78787878, %8\n"
#define I_ANDFF(n) "v_and_b32 %" #n ", 0xff, %8\n"
#define I_MOVSDWA(n) "v_mov_b32_sdwa %" #n ", %8 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_1\n"
#define I_ORSDWA(n) "v_or_b32_sdwa %" #n ", %9, %8 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_2\n"
#define I_LSHLSDWA(n) "v_lshlrev_b32_sdwa %" #n ", %9, %8 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_1\n"
#define I_LSHLI(n) "v_lshlrev_b32 %" #n ", 3, %8\n"
#define I_CNDMASK(n) "v_cndmask_b32 %" #n ", %8, %9, vcc\n"
#define I_FMACS(n) "v_fmac_f32 %" #n ", %10, %9\n"
#define I_DOT2(n) "v_dot2_f32_f16 %" #n ", %8, %9, %" #n "\n"

template <int MODE>
__global__ __launch_bounds__(512, 2) void k(float *out, int iters, unsigned sg) {
  float f0 = 1, f1 = 2, f2 = 3, f3 = 4, f4 = 5, f5 = 6, f6 = 7, f7 = 8, fa = 0.5f, fb = (float)threadIdx.x;
  f32x2 p0 = {1, 2}, p1 = p0, p2 = p0, p3 = p0, p4 = p0, p5 = p0, p6 = p0, p7 = p0, pa = {0.5f, 0.25f}, pb = {fb, 1.f};
  unsigned u0 = 0, u1 = 0, u2 = 0, u3 = 0, u4 = 0, u5 = 0, u6 = 0, u7 = 0;
  unsigned w = threadIdx.x * 2654435761u, x = threadIdx.x & 7;
  for (int it = 0; it < iters; it++) {
    if (MODE == 0) { REP8(U8(I_LSHR)) }
    if (MODE == 1) { REP8(U8(I_ANDI)) }
    if (MODE == 2) { REP8(U8(I_ANDV)) }
    if (MODE == 3) { REP8(U8(I_ANDS)) }
    if (MODE == 4) { REP8(U8(I_OR)) }
    if (MODE == 5) { REP8(U8(I_ADDU)) }
    if (MODE == 6) { REP8(U8(I_LSHLADD)) }
    if (MODE == 7) { REP8(U8(I_PERM)) }
    if (MODE == 8) { REP8(U8(I_MAD24)) }
    if (MODE == 9) { REP8(U8(I_ALIGN)) }
    if (MODE == 10) { REP8(U8(I_BFE)) }
    if (MODE == 11) { REP8(U8(I_BFEV)) }
    if (MODE == 12) { REP8(U8(I_ANDOR)) }
    if (MODE == 13) { REP8(U8(I_LSHLOR)) }
    if (MODE == 14) { REP8(U8(I_CVTUB)) }
    if (MODE == 15) { REP8(U8(I_MOV)) }
    if (MODE == 16) { REP8(U8(I_XOR3)) }
    if (MODE == 17) { REP8(U8(I_ADD3)) }
    if (MODE == 18) { REP8(F8(I_FMAC)) }
    if (MODE == 19) { REP8(F8(I_FMA)) }
    if (MODE == 20) { REP8(F8(I_MUL)) }
    if (MODE == 21) { REP8(F8(I_ADDF)) }
    if (MODE == 22) { REP8(F8(I_SIN)) }
    if (MODE == 23) { REP8(P8(I_PKFMA)) }
    if (MODE == 24) { REP8(P8(I_PKADD)) }
    if (MODE == 25) { REP8(F8(I_PKFMA16)) }
    if (MODE == 26) { REP8(F8(I_DOT2C)) }
    if (MODE == 27) { REP8(U8(I_ANDLIT)) }
    if (MODE == 28) { REP8(U8(I_ANDFF)) }
    if (MODE == 29) { REP8(U8(I_MOVSDWA)) }
    if (MODE == 30) { REP8(U8(I_ORSDWA)) }
    if (MODE == 31) { REP8(U8(I_LSHLSDWA)) }
    if (MODE == 32) { REP8(U8(I_LSHLI)) }
    if (MODE == 33) { REP8(U8(I_CNDMASK)) }
  }
  float s = f0 + f1 + f2 + f3 + f4 + f5 + f6 + f7 + (p0 + p1 + p2 + p3 + p4 + p5 + p6 + p7).x;
  unsigned us = u0 ^ u1 ^ u2 ^ u3 ^ u4 ^ u5 ^ u6 ^ u7;
  if (s + (float)us == 12345.678f) out[0] = s;
}
template <int MODE>
static void run(const char *name, float *d, int blocks) {
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  int iters = 2000;
  float ms = 0;
  for (int rep = 0; rep < 2; rep++) {
    hipEventRecord(e0);
    k<MODE><<<blocks, 512>>>(d, iters, 0x07060504u);
    hipEventRecord(e1); hipEventSynchronize(e1);
    hipEventElapsedTime(&ms, e0, e1);
  }
  double inst_per_simd = (blocks / 256.0) * 2.0 * iters * 64.0;   // waves per SIMD x instructions
  printf("%-22s %d waves/SIMD: %7.3f ms -> %.2f clk per wave-instruction per SIMD @2.4 GHz\n", name, blocks / 128, ms,
         ms * 1e6 / inst_per_simd * 2.4);
}
#define RUN(M, NAME) run<M>(NAME, d, 512); run<M>(NAME, d, 256);
int main() {
  float *d; hipMalloc(&d, 4096);
  RUN(0, "v_lshrrev_b32 imm") RUN(1, "v_and_b32 inline") RUN(2, "v_and_b32 v,v") RUN(3, "v_and_b32 s,v") RUN(4, "v_or_b32 v,v")
  RUN(5, "v_add_u32") RUN(6, "v_lshl_add_u32") RUN(7, "v_perm_b32") RUN(8, "v_mad_u32_u24") RUN(9, "v_alignbit_b32")
  RUN(10, "v_bfe_u32 imm") RUN(11, "v_bfe_u32 vgpr") RUN(12, "v_and_or_b32") RUN(13, "v_lshl_or_b32") RUN(14, "v_cvt_f32_ubyte1")
  RUN(15, "v_mov_b32") RUN(16, "v_xor3_b32") RUN(17, "v_add3_u32") RUN(18, "v_fmac_f32") RUN(19, "v_fma_f32") RUN(20, "v_mul_f32")
  RUN(21, "v_add_f32") RUN(22, "v_sin_f32") RUN(23, "v_pk_fma_f32") RUN(24, "v_pk_add_f32") RUN(25, "v_pk_fma_f16") RUN(26, "v_dot2c_f32_f16")
  RUN(27, "v_and_b32 literal") RUN(28, "v_and_b32 0xff") RUN(29, "v_mov_b32_sdwa byte") RUN(30, "v_or_b32_sdwa byte") RUN(31, "v_lshlrev_b32_sdwa byte") RUN(32, "v_lshlrev_b32 imm") RUN(33, "v_cndmask_b32 vcc")
  return 0;
}
