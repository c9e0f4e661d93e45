// Query-premultiplied K codebook image shared by kvq_score_k.hip (consumer) and the producers
// (kvq_score_k's own prep launch, or the combined decode prologue in kvq_fused_append.hip).
#pragma once
#include "kvq_common.h"

namespace kvq {

typedef float f32x2 __attribute__((ext_vector_type(2)));

// LDS image of one head's codebook, pre-multiplied by the query.  Rotation pair i (0..31) of
// wave-half ("role") r covers channel k_lo = 32r+i and k_hi = 64+32r+i.  Entry order:
//   TLO[(i*2 + r)*N + code] = (L[k_lo][code]*q[k_lo],  L[k_lo][code]*q[k_lo+64])
//   THI[(i*2 + r)*N + code] = (L[k_hi][code]*q[k_hi], -L[k_hi][code]*q[k_hi-64])
template <int BITS>
struct KTab {
  static constexpr int N = Fmt<BITS>::kN;
  static constexpr int HALF_B = 32 * 2 * N * 8;     // bytes of TLO (= THI)
  static constexpr int BUF_B = 2 * HALF_B;          // one head
};

// element i of an activation vector that is fp32 or fp16 (the model's dtype)
__device__ __forceinline__ float ld_act(const void *p, int64_t i, int is_half) {
  return is_half ? __half2float(reinterpret_cast<const __half *>(p)[i]) : reinterpret_cast<const float *>(p)[i];
}

// 3 bit: PAIR-SUM image of one head's codebook in fp16 (round 5).  The two channels of a rotation pair, k_lo = 32r+i and
// k_hi = k_lo + 64, share (cos, sin), so their two look-ups and two packed FMAs collapse into ONE look-up indexed by both
// codes and one v_dot2_f32_f16 against the lane's half2 (cos, sin):
//   P[((i*2 + r)*64 + (c_lo | c_hi << 3))] = half2(a*q[k_lo] + b*q[k_hi],  a*q[k_hi] - b*q[k_lo]),  a = L[k_lo][c_lo], b = L[k_hi][c_hi]
// 64 entries x 4 B per (pair, role) = 16 KB per head (the fp32 image of 4 bit: 16 KB; a 4-bit pair table would be 64 KB).
// The sums are formed in fp32 and rounded once to fp16 (2^-11 relative): scores agree with the fp32 path to ~1e-4 of the
// row's largest score (bar: 1e-3, BASELINE.json; the reference rounds the score itself to fp16, modeling_llama.py:873).
struct KTabPair3 {
  static constexpr int PAIR_B = 64 * 4;             // one (pair, role)
  static constexpr int BUF_B = 32 * 2 * PAIR_B;     // one head: 16 KB
};
// 3 bit: the same pair sums in fp32 (round 6) -- EXACT arithmetic (a sum of two fp32 products, as the per-channel path forms it,
// in another order), one ds_read_b64 + one v_pk_fma_f32 per TWO codes.  64 entries x 8 B per (pair, role) = 32 KB per head: two
// table buffers + the score tile of 512 tokens need a 1024-lane workgroup, one per CU (score_k_kernel<3, ..., NWAVES = 16, PAIR = 2>).
//   P32[((i*2 + r)*64 + (c_lo | c_hi << 3))] = (a*q[k_lo] + b*q[k_hi],  a*q[k_hi] - b*q[k_lo])
// The image lives where the fp16 one does (a layer scores through one of them): the region is sized for the larger.
struct KTabPair32 {
  static constexpr int PAIR_B = 64 * 8;             // one (pair, role)
  static constexpr int BUF_B = 32 * 2 * PAIR_B;     // one head: 32 KB
};
// does this width keep a pair-sum image next to the fp32 one?
template <int BITS>
struct KTabHasPair { static constexpr bool value = BITS == 3; };

// workspace layout of kvq_score_k: [q_len*H tables of BUF_B bytes][q_len*H*128 floats: q as fp32]
// [3 bit: q_len*H pair-sum images of KTabPair3::BUF_B bytes]
template <int BITS>
__host__ __device__ __forceinline__ size_t ktab_q_offset(int q_len, int H) {
  return (size_t)q_len * H * KTab<BITS>::BUF_B;
}
template <int BITS>
__host__ __device__ __forceinline__ size_t ktab_pair_offset(int q_len, int H) {
  return ktab_q_offset<BITS>(q_len, H) + (size_t)q_len * H * kHeadDim * sizeof(float);
}
template <int BITS>
__host__ __device__ __forceinline__ size_t ktab_total_bytes(int q_len, int H) {
  return ktab_pair_offset<BITS>(q_len, H) + (KTabHasPair<BITS>::value ? (size_t)q_len * H * KTabPair32::BUF_B : 0);
}

// builds the image of head h / query row b in global memory (L2-resident: H*16 KB) and the fp32 copy
// of that q row; called by all threads of a workgroup
template <int BITS>
__device__ __forceinline__ void lutq_prep_head(const float *__restrict__ lut, const void *__restrict__ q,
                                               int q_is_half, unsigned char *__restrict__ tab,
                                               float *__restrict__ q32, unsigned char *__restrict__ pair_tab, int H,
                                               int h, int b, int pair_mode = 1) {
  constexpr int N = Fmt<BITS>::kN;
  const float *lh = lut + (int64_t)h * kHeadDim * N;
  const int64_t qoff = ((int64_t)b * H + h) * kHeadDim;
  unsigned char *dst = tab + ((int64_t)b * H + h) * KTab<BITS>::BUF_B;
  for (int k = threadIdx.x; k < kHeadDim; k += blockDim.x) q32[qoff + k] = ld_act(q, qoff + k, q_is_half);
  for (int e4 = threadIdx.x; e4 < kHeadDim * N / 4; e4 += blockDim.x) {
    const int e0 = e4 * 4;
    const int k = e0 / N, v0 = e0 % N;
    const float4 l4 = *reinterpret_cast<const float4 *>(lh + e0);
    const float qa = ld_act(q, qoff + k, q_is_half);
    const float qb = (k < 64) ? ld_act(q, qoff + k + 64, q_is_half) : -ld_act(q, qoff + k - 64, q_is_half);
    const int kk = k & 63;
    const int r = kk >> 5, i = kk & 31;
    float4 *d = reinterpret_cast<float4 *>(dst + (k >> 6) * KTab<BITS>::HALF_B + (((i * 2 + r) * N + v0) << 3));
    d[0] = make_float4(l4.x * qa, l4.x * qb, l4.y * qa, l4.y * qb);
    d[1] = make_float4(l4.z * qa, l4.z * qb, l4.w * qa, l4.w * qb);
  }
  if constexpr (KTabHasPair<BITS>::value) {
    // the pair-sum image (KTabPair3): 4096 entries per head
    if (pair_tab == nullptr || pair_mode == 0) return;
    if (pair_mode == 2) {       // fp32 pair sums (KTabPair32)
      unsigned char *pdst = pair_tab + ((int64_t)b * H + h) * KTabPair32::BUF_B;
      for (int e = threadIdx.x; e < 32 * 2 * 64; e += blockDim.x) {
        const int idx = e & 63, ir = e >> 6;           // ir = i*2 + r
        const int i = ir >> 1, r = ir & 1;
        const int k_lo = 32 * r + i, k_hi = k_lo + 64;
        const float a = lh[k_lo * N + (idx & 7)], bb = lh[k_hi * N + (idx >> 3)];
        const float q_lo = ld_act(q, qoff + k_lo, q_is_half), q_hi = ld_act(q, qoff + k_hi, q_is_half);
        // (products rounded one by one, then added: the per-channel tables hold exactly these products)
        reinterpret_cast<f32x2 *>(pdst)[e] = f32x2{a * q_lo + bb * q_hi, a * q_hi - bb * q_lo};
      }
      return;
    }
    unsigned char *pdst = pair_tab + ((int64_t)b * H + h) * KTabPair3::BUF_B;
    for (int e = threadIdx.x; e < 32 * 2 * 64; e += blockDim.x) {
      const int idx = e & 63, ir = e >> 6;           // ir = i*2 + r
      const int i = ir >> 1, r = ir & 1;
      const int k_lo = 32 * r + i, k_hi = k_lo + 64;
      const float a = lh[k_lo * N + (idx & 7)], bb = lh[k_hi * N + (idx >> 3)];
      const float q_lo = ld_act(q, qoff + k_lo, q_is_half), q_hi = ld_act(q, qoff + k_hi, q_is_half);
      const __half2 v = __floats2half2_rn(fmaf(a, q_lo, bb * q_hi), fmaf(a, q_hi, -(bb * q_lo)));
      reinterpret_cast<__half2 *>(pdst)[e] = v;
    }
  }
}


template <int BITS>
__global__ __launch_bounds__(256) void lutq_prep_kernel(const float *__restrict__ lut, const void *__restrict__ q,
                                                        int q_is_half, unsigned char *__restrict__ tab,
                                                        float *__restrict__ q32, unsigned char *__restrict__ pair_tab,
                                                        int H, int pair_mode = 1) {
  lutq_prep_head<BITS>(lut, q, q_is_half, tab, q32, pair_tab, H, blockIdx.x, blockIdx.y, pair_mode);
}

}  // namespace kvq
