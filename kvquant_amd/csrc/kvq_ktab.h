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

// workspace layout of kvq_score_k: [q_len*H tables of BUF_B bytes][q_len*H*128 floats: q as fp32]
template <int BITS>
__device__ __forceinline__ size_t ktab_q_offset(int q_len, int H) {
  return (size_t)q_len * H * KTab<BITS>::BUF_B;
}

// builds the image of head h / query row b in global memory (L2-resident: H*16 KB) and the fp32 copy
// of that q row; called by all threads of a workgroup
template <int BITS>
__device__ __forceinline__ void lutq_prep_head(const float *__restrict__ lut, const void *__restrict__ q,
                                               int q_is_half, unsigned char *__restrict__ tab,
                                               float *__restrict__ q32, int H, int h, int b) {
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
}


template <int BITS>
__global__ __launch_bounds__(256) void lutq_prep_kernel(const float *__restrict__ lut, const void *__restrict__ q,
                                                        int q_is_half, unsigned char *__restrict__ tab,
                                                        float *__restrict__ q32, int H) {
  lutq_prep_head<BITS>(lut, q, q_is_half, tab, q32, H, blockIdx.x, blockIdx.y);
}

}  // namespace kvq
