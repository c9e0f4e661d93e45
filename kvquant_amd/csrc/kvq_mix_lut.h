// Hand-scheduled instruction groups of the per-row p.V look-up loops (4 / 3 / 2 bit), shared by kvq_mix_v.hip and
// kvq_fused_decode.hip.
#pragma once
#include "kvq_common.h"

namespace kvq {

// ---- hand-scheduled pieces of the 4-bit look-up loop ---------------------------------------------------------
// hipcc's own schedule of the loop runs at 8.4 ns per code-step per SIMD, VALU and LDS time ADDED UP; the same
// instructions issued as below -- the look-ups of token t+1 in flight while token t is accumulated, plain v_fmac
// (v_pk_fma_f32 takes two passes), one prepare step per word -- run at 5.7 (tools/ubench/lut_loop.hip).
// LDS operations return in order, so s_waitcnt lgkmcnt(N) = "all but the last N issued have landed".
template <int OFF>
__device__ __forceinline__ void lds_read16(uint4 &w, uint32_t addr) {
  asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(w) : "v"(addr), "n"(OFF) : "memory");
}
template <int OFF>
__device__ __forceinline__ void lds_read16(float4 &w, uint32_t addr) {
  asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(w) : "v"(addr), "n"(OFF) : "memory");
}
// nibbles of w -> bytes holding code*4 + slot*64: even nibbles in we, odd in wo
__device__ __forceinline__ void nib_prep(uint32_t &we, uint32_t &wo, uint32_t w, uint32_t slotpat) {
  asm volatile("v_lshlrev_b32 %0, 2, %2\n\tv_lshrrev_b32 %1, 2, %2\n\tv_and_or_b32 %0, %0, %4, %3\n\tv_and_or_b32 %1, %1, %4, %3"
               : "=&v"(we), "=&v"(wo) : "v"(w), "v"(slotpat), "s"(0x3C3C3C3Cu));
}
__device__ __forceinline__ void nib_extract(uint32_t (&u)[8], uint32_t we, uint32_t wo) {
  asm volatile("v_and_b32 %0, 0xff, %8\n\tv_and_b32 %1, 0xff, %9\n\tv_bfe_u32 %2, %8, 8, 8\n\tv_bfe_u32 %3, %9, 8, 8\n\t"
               "v_bfe_u32 %4, %8, 16, 8\n\tv_bfe_u32 %5, %9, 16, 8\n\tv_lshrrev_b32 %6, 24, %8\n\tv_lshrrev_b32 %7, 24, %9"
               : "=&v"(u[0]), "=&v"(u[1]), "=&v"(u[2]), "=&v"(u[3]), "=&v"(u[4]), "=&v"(u[5]), "=&v"(u[6]), "=&v"(u[7])
               : "v"(we), "v"(wo));
}
template <int OFF>
__device__ __forceinline__ void lut_read8(float (&v)[8], const uint32_t (&u)[8]) {
  asm volatile("ds_read_b32 %0, %8 offset:%16\n\tds_read_b32 %1, %9 offset:%16\n\tds_read_b32 %2, %10 offset:%16\n\t"
               "ds_read_b32 %3, %11 offset:%16\n\tds_read_b32 %4, %12 offset:%16\n\tds_read_b32 %5, %13 offset:%16\n\t"
               "ds_read_b32 %6, %14 offset:%16\n\tds_read_b32 %7, %15 offset:%16"
               : "=&v"(v[0]), "=&v"(v[1]), "=&v"(v[2]), "=&v"(v[3]), "=&v"(v[4]), "=&v"(v[5]), "=&v"(v[6]), "=&v"(v[7])
               : "v"(u[0]), "v"(u[1]), "v"(u[2]), "v"(u[3]), "v"(u[4]), "v"(u[5]), "v"(u[6]), "v"(u[7]), "n"(OFF)
               : "memory");
}
__device__ __forceinline__ void fmac8(float (&a)[8], const float (&v)[8], float p) {
  asm volatile("v_fmac_f32 %0, %8, %16\n\tv_fmac_f32 %1, %9, %16\n\tv_fmac_f32 %2, %10, %16\n\tv_fmac_f32 %3, %11, %16\n\t"
               "v_fmac_f32 %4, %12, %16\n\tv_fmac_f32 %5, %13, %16\n\tv_fmac_f32 %6, %14, %16\n\tv_fmac_f32 %7, %15, %16"
               : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(a[4]), "+v"(a[5]), "+v"(a[6]), "+v"(a[7])
               : "v"(v[0]), "v"(v[1]), "v"(v[2]), "v"(v[3]), "v"(v[4]), "v"(v[5]), "v"(v[6]), "v"(v[7]), "v"(p));
}
template <int O, int NA>
__device__ __forceinline__ void fmac8_at(float (&a)[NA], const float (&v)[8], float p) {
  asm volatile("v_fmac_f32 %0, %8, %16\n\tv_fmac_f32 %1, %9, %16\n\tv_fmac_f32 %2, %10, %16\n\tv_fmac_f32 %3, %11, %16\n\t"
               "v_fmac_f32 %4, %12, %16\n\tv_fmac_f32 %5, %13, %16\n\tv_fmac_f32 %6, %14, %16\n\tv_fmac_f32 %7, %15, %16"
               : "+v"(a[O]), "+v"(a[O + 1]), "+v"(a[O + 2]), "+v"(a[O + 3]), "+v"(a[O + 4]), "+v"(a[O + 5]), "+v"(a[O + 6]), "+v"(a[O + 7])
               : "v"(v[0]), "v"(v[1]), "v"(v[2]), "v"(v[3]), "v"(v[4]), "v"(v[5]), "v"(v[6]), "v"(v[7]), "v"(p));
}
template <int N>
__device__ __forceinline__ void lds_wait() {
  asm volatile("s_waitcnt lgkmcnt(%0)" ::"n"(N) : "memory");
}

// ---- 2 bit: a word holds 16 codes; pre-masked word k holds code*4 (+ the slot's row offset, 16 bytes) of channel
// 4b + k in byte b, so one byte extraction per code is its look-up address
__device__ __forceinline__ void duo_prep(uint32_t (&pk)[4], uint32_t w, uint32_t slotpat) {
  asm volatile("v_lshlrev_b32 %0, 2, %4\n\tv_lshrrev_b32 %2, 2, %4\n\tv_lshrrev_b32 %3, 4, %4\n\t"
               "v_and_or_b32 %0, %0, %5, %6\n\tv_and_or_b32 %1, %4, %5, %6\n\tv_and_or_b32 %2, %2, %5, %6\n\tv_and_or_b32 %3, %3, %5, %6"
               : "=&v"(pk[0]), "=&v"(pk[1]), "=&v"(pk[2]), "=&v"(pk[3]) : "v"(w), "s"(0x0C0C0C0Cu), "v"(slotpat));
}
// channels 0..7 (bytes 0 and 1 of the four pre-masked words) / 8..15 (bytes 2 and 3)
__device__ __forceinline__ void duo_extract_a(uint32_t (&u)[8], const uint32_t (&pk)[4]) {
  asm volatile("v_and_b32 %0, 0xff, %8\n\tv_and_b32 %1, 0xff, %9\n\tv_and_b32 %2, 0xff, %10\n\tv_and_b32 %3, 0xff, %11\n\t"
               "v_bfe_u32 %4, %8, 8, 8\n\tv_bfe_u32 %5, %9, 8, 8\n\tv_bfe_u32 %6, %10, 8, 8\n\tv_bfe_u32 %7, %11, 8, 8"
               : "=&v"(u[0]), "=&v"(u[1]), "=&v"(u[2]), "=&v"(u[3]), "=&v"(u[4]), "=&v"(u[5]), "=&v"(u[6]), "=&v"(u[7])
               : "v"(pk[0]), "v"(pk[1]), "v"(pk[2]), "v"(pk[3]));
}
__device__ __forceinline__ void duo_extract_b(uint32_t (&u)[8], const uint32_t (&pk)[4]) {
  asm volatile("v_bfe_u32 %0, %8, 16, 8\n\tv_bfe_u32 %1, %9, 16, 8\n\tv_bfe_u32 %2, %10, 16, 8\n\tv_bfe_u32 %3, %11, 16, 8\n\t"
               "v_lshrrev_b32 %4, 24, %8\n\tv_lshrrev_b32 %5, 24, %9\n\tv_lshrrev_b32 %6, 24, %10\n\tv_lshrrev_b32 %7, 24, %11"
               : "=&v"(u[0]), "=&v"(u[1]), "=&v"(u[2]), "=&v"(u[3]), "=&v"(u[4]), "=&v"(u[5]), "=&v"(u[6]), "=&v"(u[7])
               : "v"(pk[0]), "v"(pk[1]), "v"(pk[2]), "v"(pk[3]));
}

// ---- 3 bit: the 32 codes of a unit are one 96-bit stream over three words; a lane decodes 16 of them (its half).
// `src` holds consecutive 3-bit codes from bit 0: the even ones masked in place and shifted left by 2, the odd ones
// shifted right by 1 and masked, land as code*4 in bits [6j+2, 6j+5) of 6-bit fields -- bit 6j+5 takes the slot's table
// offset (32 bytes), so ONE v_bfe_u32 per code yields the complete variable part of its look-up address.
__device__ __forceinline__ void tri_prep(uint32_t &ev, uint32_t &od, uint32_t src, uint32_t slotpat) {
  asm volatile("v_and_b32 %0, %3, %2\n\tv_lshrrev_b32 %1, 1, %2\n\tv_lshl_or_b32 %0, %0, 2, %5\n\tv_and_or_b32 %1, %1, %4, %5"
               : "=&v"(ev), "=&v"(od) : "v"(src), "s"(0x071C71C7u), "s"(0x1C71C71Cu), "v"(slotpat));
}
// fields 0..3 of (ev, od) interleaved: codes 0..7 of the stream
__device__ __forceinline__ void tri_extract_a(uint32_t (&u)[8], uint32_t ev, uint32_t od) {
  asm volatile("v_and_b32 %0, 63, %8\n\tv_and_b32 %1, 63, %9\n\tv_bfe_u32 %2, %8, 6, 6\n\tv_bfe_u32 %3, %9, 6, 6\n\t"
               "v_bfe_u32 %4, %8, 12, 6\n\tv_bfe_u32 %5, %9, 12, 6\n\tv_bfe_u32 %6, %8, 18, 6\n\tv_bfe_u32 %7, %9, 18, 6"
               : "=&v"(u[0]), "=&v"(u[1]), "=&v"(u[2]), "=&v"(u[3]), "=&v"(u[4]), "=&v"(u[5]), "=&v"(u[6]), "=&v"(u[7])
               : "v"(ev), "v"(od));
}
// field 4 of (ev, od) (codes 8, 9) and fields 0..2 of the second pair (codes 10..15)
__device__ __forceinline__ void tri_extract_b(uint32_t (&u)[8], uint32_t ev, uint32_t od, uint32_t ev2, uint32_t od2) {
  asm volatile("v_bfe_u32 %0, %8, 24, 6\n\tv_bfe_u32 %1, %9, 24, 6\n\tv_and_b32 %2, 63, %10\n\tv_and_b32 %3, 63, %11\n\t"
               "v_bfe_u32 %4, %10, 6, 6\n\tv_bfe_u32 %5, %11, 6, 6\n\tv_bfe_u32 %6, %10, 12, 6\n\tv_bfe_u32 %7, %11, 12, 6"
               : "=&v"(u[0]), "=&v"(u[1]), "=&v"(u[2]), "=&v"(u[3]), "=&v"(u[4]), "=&v"(u[5]), "=&v"(u[6]), "=&v"(u[7])
               : "v"(ev), "v"(od), "v"(ev2), "v"(od2));
}
// the two 3-bit streams of a lane's half: codes [0, 10) and [10, 16) of the half, from the unit's three words
//   half 0 (channels 0..15):  bits 0.. of w0, and bits 30.. of (w1:w0);   half 1 (16..31): bits 16.. of (w2:w1), bits 14.. of w2
__device__ __forceinline__ void tri_streams(uint32_t &s1, uint32_t &s2, uint32_t w0, uint32_t w1, uint32_t w2, int hf) {
  if (hf == 0) {   // (wave-uniform)
    asm volatile("v_mov_b32 %0, %2\n\tv_alignbit_b32 %1, %3, %2, 30" : "=&v"(s1), "=&v"(s2) : "v"(w0), "v"(w1));
  } else {
    asm volatile("v_alignbit_b32 %0, %3, %2, 16\n\tv_lshrrev_b32 %1, 14, %3" : "=&v"(s1), "=&v"(s2) : "v"(w1), "v"(w2));
  }
}

}  // namespace kvq
