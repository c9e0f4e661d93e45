// Shared device helpers for the gfx950 KVQuant kernels.
// Built with -ffp-contract=off: the places that must round like the reference
// (LUT rows, rescale, nearest-code compare) are plain mul/add/div sequences;
// everything that may fuse uses fmaf explicitly.
#pragma once
#include <hip/hip_fp16.h>
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/kvq.h"

// measurement hook of kvq_decode_step (kvq_decode_step.hip): the fused p.V route calls it in front of its p.V kernel
extern "C" void kvq_step_mark_pv(hipStream_t st);
extern "C" void kvq_step_mark_fused(hipStream_t st);

namespace kvq {

// half(half(raw) * inv): the reference casts the scores to fp16 and divides that tensor by a Python scalar
// on the GPU (modeling_llama.py:873-874, 1972-1973), which torch evaluates as an fp32 multiply by the
// fp32 reciprocal, rounded back to fp16.
__device__ __forceinline__ float scaled(float raw, float inv) {
  const float h = __half2float(__float2half_rn(raw));
  return __half2float(__float2half_rn(h * inv));
}

// exp(d) for d <= 0 on the hardware exponential: 2^(d log2 e) with the product carried to ~2^-48 (hi + lo) and a
// first-order correction -- 7 VALU operations against ~25 for expf, <= 1.5 ulp (results below 2^-126 flush to 0).
__device__ __forceinline__ float exp_neg(float d) {
  const float t = d * 1.44269502f;
  const float e = fmaf(d, 1.44269502f, -t) + d * 1.92596299e-8f;
  const float r = __builtin_amdgcn_exp2f(t);
  return fmaf(r, e * 0.693147182f, r);
}
// scaled score -> fp16-rounded probability, exp(x - M) / Z rounded to fp16 (modeling_llama.py:1976), with rZ = 1/Z.
// The softmax passes evaluate 4 M of these per layer at 128K: with expf and a true division (~45 operations) the
// normalising kernel is VALU bound (14 us); this form costs 12 and is <= 2 ulp from the exact fp32 quotient BEFORE the
// fp16 rounding -- about one probability in 2000 lands on the neighbouring fp16 value, the scale of the difference
// between two exact-to-an-ulp softmax implementations (torch's on the GPU and libm on the host differ the same way).
__device__ __forceinline__ float prob_fp16(float x, float M, float rZ) {
  return __half2float(__float2half_rn(exp_neg(x - M) * rZ));
}

// attention output of the fp16 sink tokens for channel c of head h: torch.matmul(probs[:, :n_sink] (fp16),
// value_states_fp16) (modeling_llama.py:1987-1995) -- fp32 accumulation, ONE rounding to fp16 -- from the sink scores
// and the row's (max, 1 / normaliser); the probabilities are re-evaluated per lane (n_sink is a handful)
__device__ __forceinline__ float sink_output(const __half *sink, const __half *v_sink, int n_sink, int h, int c,
                                             float M, float rZ) {
  float acc = 0.f;
  for (int i = 0; i < n_sink; i++) {
    const float pi = __half2float(__float2half_rn(prob_fp16(__half2float(sink[h * n_sink + i]), M, rZ)));
    acc = fmaf(pi, __half2float(v_sink[((int64_t)h * n_sink + i) * 128 + c]), acc);
  }
  return __half2float(__float2half_rn(acc));
}

constexpr int kWave = 64;
constexpr int kHeadDim = 128;  // score / mix kernels (reference BLOCKWIDTH, KCU:43)

// words of packed cache per 32 channels == bits
template <int BITS>
struct Fmt {
  static constexpr int kN = 1 << BITS;               // LUT entries
  static constexpr int kWordsPer32 = BITS;           // int32 words per 32 channels
  static constexpr int kWordsPerHead = kHeadDim / 32 * BITS;
  static constexpr unsigned kZeroCode = BITS == 4 ? 7u : (BITS == 3 ? 3u : 1u);  // KCU:2084/2442/3020
};

// argmin_v |lut[v]-x|, strict '<' from v=0 (KCU:1222-1237)
template <int N>
__device__ __forceinline__ unsigned nearest_code(const float (&lut)[N], float x) {
  unsigned best = 0;
  float prev = fabsf(lut[0] - x);
#pragma unroll
  for (int v = 1; v < N; v++) {
    float d = fabsf(lut[v] - x);
    if (d < prev) {
      prev = d;
      best = (unsigned)v;
    }
  }
  return best;
}

// Pack 32 codes (one 32-channel group) into BITS words; layout of
// KCU:1240-1244 (4b), 1395-1424 (3b), 2712-2716 (2b).
template <int BITS>
__device__ __forceinline__ void pack32(const unsigned (&code)[32], uint32_t (&w)[BITS]) {
#pragma unroll
  for (int i = 0; i < BITS; i++) w[i] = 0;
  if constexpr (BITS == 4) {
#pragma unroll
    for (int i = 0; i < 32; i++) w[i / 8] |= code[i] << (4 * (i % 8));
  } else if constexpr (BITS == 2) {
#pragma unroll
    for (int i = 0; i < 32; i++) w[i / 16] |= code[i] << (2 * (i % 16));
  } else {
#pragma unroll
    for (int i = 0; i < 32; i++) {
      if (i == 10) {
        w[0] |= code[i] << 30;
        w[1] |= code[i] >> 2;
      } else if (i == 21) {
        w[1] |= code[i] << 31;
        w[2] |= code[i] >> 1;
      } else {
        w[i / 11] |= code[i] << ((i * 3) % 32);
      }
    }
  }
}

// code of channel i (0..31, compile-time) of a 32-channel group held in BITS words.
template <int BITS, int I>
__device__ __forceinline__ unsigned code_of(const uint32_t (&w)[BITS]) {
  if constexpr (BITS == 4) {
    return (w[I / 8] >> (4 * (I % 8))) & 0xfu;
  } else if constexpr (BITS == 2) {
    return (w[I / 16] >> (2 * (I % 16))) & 0x3u;
  } else {
    if constexpr (I < 10) return (w[0] >> (3 * I)) & 0x7u;
    else if constexpr (I == 10) return ((w[0] >> 30) & 0x3u) | ((w[1] & 0x1u) << 2);
    else if constexpr (I < 21) return (w[1] >> ((3 * I) % 32)) & 0x7u;
    else if constexpr (I == 21) return ((w[1] >> 31) & 0x1u) | ((w[2] & 0x3u) << 1);
    else return (w[2] >> ((3 * I) % 32)) & 0x7u;
  }
}

// compile-time loop helper
template <int I, int N, typename F>
__device__ __forceinline__ void static_for(F &&f) {
  if constexpr (I < N) {
    f(std::integral_constant<int, I>{});
    static_for<I + 1, N>(f);
  }
}

// ---- sincos of a (possibly huge) fp32 angle, |error| ~1e-6 ----------------
// The reference evaluates cosf/sinf(fl32(theta_k * pos)) (KCU:3122-3123); the
// rounding of the product is part of the contract, so the
// caller forms `ang` exactly that way and this only has to be an accurate
// sincos of that float.  x/(2*pi) is split hi/lo with FMAs (C1+C2 = 1/(2*pi) to
// ~2^-50) so that the fractional revolution is exact to ~1e-7 for any
// |ang| < 2^31; the hardware v_sin/v_cos take revolutions.
__device__ __forceinline__ void sincos_rev(float ang, float &s, float &c) {
  constexpr float C1 = 0.15915494f;         // fl32(1/(2*pi)) = 0x3E22F983
  constexpr float C2 = 6.4206382e-9f;       // 1/(2*pi) - C1
  float hi = ang * C1;
  float e = fmaf(ang, C1, -hi);             // exact rounding error of hi
  float lo = fmaf(ang, C2, e);
  float r = __builtin_amdgcn_fractf(hi) + lo;
  s = __builtin_amdgcn_sinf(r);
  c = __builtin_amdgcn_cosf(r);
}

// theta_j = powf(rope_theta, -2j/128), j = 0..63, evaluated ON THE DEVICE exactly as the reference writes it
// (KCU:3081: powf(rope_theta, (-2 * __int2float_rd(off % headdim2) / __int2float_rd(headdim)))).  The device
// powf is part of the contract: on MI355X it differs from the correctly rounded value by 1 ulp for 21 of the
// 64 frequencies, and 1 ulp of theta_j is 6e-8 * pos radians -- 8e-3 rad at 128K, 6e-2 at 1M
// (tests/test_ref_gpu.py measures both against the reference's own kernels built for this GPU).
__device__ __forceinline__ float rope_freq(float rope_theta, int j) {
  return powf(rope_theta, (-2 * (float)j / (float)kHeadDim));
}

__device__ __forceinline__ int lane_id() { return threadIdx.x & 63; }

// ---- LDS-DMA (global_load_lds) -------------------------------------------------------------
typedef __attribute__((address_space(3))) unsigned char lds_byte_t;

// LDS-DMA issued from inline asm so that hipcc does not put it on its own vmcnt scoreboard (with the
// builtin it waits vmcnt(0) in front of every later ds_read and the copy never overlaps the math;
// cdna_hip_programming.md 5.7).  lds_dst: wave-uniform LDS byte address (goes to M0), gsrc: per-lane
// source.  Completion is waited for explicitly (dma_wait_all) before the barrier that publishes a stage.
__device__ __forceinline__ void dma16(const void *sbase, uint32_t voff, uint32_t lds_dst) {
  unsigned keep;   // source = wave-uniform 64-bit base (SGPR pair) + per-lane 32-bit byte offset
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\ts_mov_b32 m0, %0"
               : "=&s"(keep) : "v"(voff), "s"(sbase), "s"(lds_dst) : "memory");
}
// the same with the non-temporal policy: rows that ONE workgroup reads ONCE (the packed value rows of p.V)
__device__ __forceinline__ void dma16_nt(const void *sbase, uint32_t voff, uint32_t lds_dst) {
  unsigned keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2 nt\n\ts_mov_b32 m0, %0"
               : "=&s"(keep) : "v"(voff), "s"(sbase), "s"(lds_dst) : "memory");
}
__device__ __forceinline__ void dma4(const void *sbase, uint32_t voff, uint32_t lds_dst) {
  unsigned keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dword %1, %2\n\ts_mov_b32 m0, %0"
               : "=&s"(keep) : "v"(voff), "s"(sbase), "s"(lds_dst) : "memory");
}
__device__ __forceinline__ void dma_wait_all() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
__device__ __forceinline__ uint32_t lds_addr(const void *p) {
  return (uint32_t)(uintptr_t)(const lds_byte_t *)p;
}
// wait until at most N of this wave's VMEM operations (DMA pieces and ordinary loads, in issue order)
// are still outstanding
template <int N>
__device__ __forceinline__ void vm_wait() {
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

// p.V: the row unit a lane owns (one packed word-row = 8 / 16 channels at 4 / 2 bit, three rows = 32 channels at 3 bit)
template <int BITS>
struct Unit {
  static constexpr int kWords = BITS == 3 ? 3 : 1;                   // word-rows per unit
  static constexpr int kCh = BITS == 4 ? 8 : (BITS == 3 ? 32 : 16);  // channels per unit
  static constexpr int kPerHead = kHeadDim / kCh;
  static constexpr int kBatch = BITS == 3 ? 8 : 16;  // tokens per register batch (16-byte loads)
};

}  // namespace kvq
