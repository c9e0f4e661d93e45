// Scale + softmax between the K score kernel and the V mix kernel, fused:
// modeling_llama.py:873-874 (scores -> fp16), 1972-1973 (/sqrt(head_dim) in
// fp16), 1950-1962 (fp16 attention-sink scores concatenated in front),
// 1976 (softmax in fp32, result cast to fp16) -- five torch launches and three
// dtype round trips in the reference.  Output: the fp16-rounded probabilities
// widened back to fp32 in the [H][L] layout the V kernel consumes (what
// `score.float()` produces at modeling_llama.py:1083), plus the sink part as
// fp16.  Rows are split over several workgroups (two passes: partial max/sum,
// then normalise) so that 32 heads fill the chip.
#include "kvq_common.h"
#include "kvq_host.h"

#include <hip/hip_fp16.h>

namespace kvq {

__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
  for (int d = 32; d >= 1; d >>= 1) v = fmaxf(v, __shfl_xor(v, d));
  return v;
}
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int d = 32; d >= 1; d >>= 1) v += __shfl_xor(v, d);
  return v;
}

// 2^(d log2 e) with the hardware exponential, for the weights that merge partial sums into the row normaliser
// (a few ulp there are far below the fp16 rounding of the probabilities; the per-element exponentials of the
// normalising pass use expf)
__device__ __forceinline__ float exp_w(float d) { return __builtin_amdgcn_exp2f(d * 1.4426950408889634f); }

// block-wide (max, sum) merge of per-lane online-softmax state (NW waves, red[2 * NW])
template <int NW = 4>
__device__ __forceinline__ void block_merge(float &m, float &s, float *red) {
  // wave level
#pragma unroll
  for (int d = 32; d >= 1; d >>= 1) {
    const float mo = __shfl_xor(m, d), so = __shfl_xor(s, d);
    const float mn = fmaxf(m, mo);
    s = (mn == -INFINITY) ? 0.f : s * exp_w(m - mn) + so * exp_w(mo - mn);
    m = mn;
  }
  const int w = threadIdx.x >> 6;
  if ((threadIdx.x & 63) == 0) { red[w] = m; red[NW + w] = s; }
  __syncthreads();
  float M = red[0];
#pragma unroll
  for (int i = 1; i < NW; i++) M = fmaxf(M, red[i]);
  float S = 0.f;
#pragma unroll
  for (int i = 0; i < NW; i++) if (red[i] > -INFINITY) S += red[NW + i] * exp_w(red[i] - M);
  m = M;
  s = S;
}

// pass 1: per (split, head) max and sum of exp in ONE sweep over the slice (online softmax per lane,
// 16-byte loads): the scores are read once here and once in pass 2
__global__ __launch_bounds__(256) void softmax_partial_kernel(const float *__restrict__ scores,
                                                              const __half *__restrict__ sink, float *__restrict__ ws,
                                                              int64_t L, int n_sink, float inv, int nsplit) {
  __shared__ float red[8];
  const int h = blockIdx.y, sp = blockIdx.x;
  const int64_t per = ((L + nsplit - 1) / nsplit + 3) & ~(int64_t)3;
  const int64_t t0 = sp * per, t1 = (t0 + per < L) ? (t0 + per) : L;
  const float *row = scores + (int64_t)h * L;
  float m = -INFINITY, s = 0.f;
  auto push = [&](float x) {
    if (x > m) { s = s * expf(m - x) + 1.f; m = x; }   // (m = -inf: s = 0*0 + 1)
    else s += expf(x - m);
  };
  // scalar head up to the first 16-byte aligned element (rows start at h*L floats: any alignment),
  // vector body, scalar tail
  int64_t ta = t0 + ((4 - (int64_t)(((reinterpret_cast<uintptr_t>(row) >> 2) + (uint64_t)t0) & 3)) & 3);
  if (ta > t1) ta = t1;
  if ((int64_t)threadIdx.x < ta - t0) push(scaled(row[t0 + threadIdx.x], inv));
  int64_t t = ta + threadIdx.x * 4;
  for (; t + 3 < t1; t += 1024) {
    const float4 v = *reinterpret_cast<const float4 *>(row + t);
    push(scaled(v.x, inv)); push(scaled(v.y, inv)); push(scaled(v.z, inv)); push(scaled(v.w, inv));
  }
  for (int64_t u = t; u < t1 && u < t + 4; u++) push(scaled(row[u], inv));
  if (sp == 0)
    for (int i = threadIdx.x; i < n_sink; i += 256) push(__half2float(sink[h * n_sink + i]));
  block_merge(m, s, red);
  if (threadIdx.x == 0) {
    ws[(h * nsplit + sp) * 2] = m;
    ws[(h * nsplit + sp) * 2 + 1] = s;
  }
}

// pass 2: combine the partials of the row, normalise, round to fp16
template <int NT>
__global__ __launch_bounds__(NT) void softmax_final_kernel(const float *__restrict__ scores,
                                                            const __half *__restrict__ sink,
                                                            const float *__restrict__ ws, float *__restrict__ probs,
                                                            __half *__restrict__ sink_probs, int64_t L, int n_sink,
                                                            float inv, int nsplit, int n_parts, int sinks_in_parts,
                                                            const __half *__restrict__ v_sink = nullptr,
                                                            float *__restrict__ sink_out = nullptr) {
  __shared__ float red[2 * (NT / 64)];
  const int h = blockIdx.y, sp = blockIdx.x;
  const int64_t per = ((L + nsplit - 1) / nsplit + 3) & ~(int64_t)3;
  const int64_t t0 = sp * per, t1 = (t0 + per < L) ? (t0 + per) : L;
  const float *row = scores + (int64_t)h * L;
  float *out = probs + (int64_t)h * L;
  // The slice's scores are requested BEFORE the (max, sum) merge of the row's partials, so that the two
  // memory round trips overlap: up to PRE 16-byte loads per lane are in flight during the merge.
  constexpr int PRE = 4;
  const bool vec = ((reinterpret_cast<uintptr_t>(row) ^ reinterpret_cast<uintptr_t>(out)) & 15) == 0;
  int64_t ta = t0 + ((4 - (int64_t)(((reinterpret_cast<uintptr_t>(row) >> 2) + (uint64_t)t0) & 3)) & 3);
  if (ta > t1) ta = t1;
  const int64_t tv = ta + threadIdx.x * 4;      // this lane's first vector element
  float4 pre[PRE];
  if (vec) {
#pragma unroll
    for (int k = 0; k < PRE; k++) {
      const int64_t t = tv + (int64_t)k * NT * 4;
      pre[k] = (t + 3 < t1) ? *reinterpret_cast<const float4 *>(row + t) : make_float4(0.f, 0.f, 0.f, 0.f);
    }
  }
  // (max, sum) of the whole row from its n_parts partials (+ the sink scores when the producer of the
  // partials did not see them): per-lane online merge, then across the block
  float M = -INFINITY, Z = 0.f;
  for (int i = threadIdx.x; i < n_parts; i += NT) {
    const float2 ms = *reinterpret_cast<const float2 *>(ws + ((int64_t)h * n_parts + i) * 2);
    if (ms.x > -INFINITY) {
      const float mn = fmaxf(M, ms.x);
      Z = Z * exp_w(M - mn) + ms.y * exp_w(ms.x - mn);   // (M = -inf: Z = 0)
      M = mn;
    }
  }
  if (!sinks_in_parts)
    for (int i = threadIdx.x; i < n_sink; i += NT) {
      const float x = __half2float(sink[h * n_sink + i]);
      const float mn = fmaxf(M, x);
      Z = Z * expf(M - mn) + expf(x - mn);
      M = mn;
    }
  block_merge<NT / 64>(M, Z, red);
  const float rZ = 1.0f / Z;
  auto f = [&](float x) { return prob_fp16(scaled(x, inv), M, rZ); };
  if (vec) {
    if ((int64_t)threadIdx.x < ta - t0) out[t0 + threadIdx.x] = f(row[t0 + threadIdx.x]);
    int64_t t = tv;
#pragma unroll
    for (int k = 0; k < PRE; k++) {
      if (t + 3 < t1) {
        *reinterpret_cast<float4 *>(out + t) = make_float4(f(pre[k].x), f(pre[k].y), f(pre[k].z), f(pre[k].w));
        t += NT * 4;
      }
    }
    for (; t + 3 < t1; t += NT * 4) {
      const float4 v = *reinterpret_cast<const float4 *>(row + t);
      *reinterpret_cast<float4 *>(out + t) = make_float4(f(v.x), f(v.y), f(v.z), f(v.w));
    }
    for (int64_t u = t; u < t1 && u < t + 4; u++) out[u] = f(row[u]);
  } else {
    for (int64_t u = t0 + threadIdx.x; u < t1; u += NT) out[u] = f(row[u]);
  }
  if (sp == 0) {
    for (int i = threadIdx.x; i < n_sink; i += NT)
      sink_probs[h * n_sink + i] = __float2half_rn(prob_fp16(__half2float(sink[h * n_sink + i]), M, rZ));
    if (v_sink != nullptr && n_sink > 0)   // the sink tokens' share of the attention output (kvq_mix_v then accumulates)
      for (int c = threadIdx.x; c < kHeadDim; c += NT) sink_out[h * kHeadDim + c] = sink_output(sink, v_sink, n_sink, h, c, M, rZ);
  }
}

static int pick_split(int H, int64_t L) {
  int64_t s = 512 / (H > 0 ? H : 1);
  const int64_t cap = (L + 1023) / 1024;
  if (s > cap) s = cap;
  if (s < 1) s = 1;
  if (s > 64) s = 64;
  return (int)s;
}

}  // namespace kvq

using namespace kvq;

extern "C" {

size_t kvq_softmax_workspace_bytes(int H, int64_t L) {
  if (H <= 0 || L < 0) return 0;
  return (size_t)H * pick_split(H, L) * 2 * sizeof(float);
}

int kvq_softmax_scale(const float *scores, const uint16_t *sink_scores, float *probs, uint16_t *sink_probs,
                      int H, int64_t L, int n_sink, float inv_sqrt_hd, void *workspace, size_t workspace_bytes,
                      void *stream) {
  if (!scores || !probs || H <= 0 || L <= 0 || n_sink < 0) return KVQ_EINVAL;
  if (n_sink > 0 && (!sink_scores || !sink_probs)) return KVQ_EINVAL;
  if (!workspace || workspace_bytes < kvq_softmax_workspace_bytes(H, L)) return KVQ_EWORKSPACE;
  const int nsplit = pick_split(H, L);
  dim3 grid(nsplit, H), block(256);
  hipStream_t st = (hipStream_t)stream;
  softmax_partial_kernel<<<grid, block, 0, st>>>(scores, reinterpret_cast<const __half *>(sink_scores),
                                                 reinterpret_cast<float *>(workspace), L, n_sink, inv_sqrt_hd, nsplit);
  int rc = check_launch();
  if (rc) return rc;
  softmax_final_kernel<256><<<grid, block, 0, st>>>(scores, reinterpret_cast<const __half *>(sink_scores),
                                               reinterpret_cast<const float *>(workspace), probs,
                                               reinterpret_cast<__half *>(sink_probs), L, n_sink, inv_sqrt_hd,
                                               nsplit, nsplit, 1);
  return check_launch();
}

int kvq_softmax_finish(const float *scores, const uint16_t *sink_scores, const float *parts, int n_parts,
                       float *probs, uint16_t *sink_probs, int H, int64_t L, int n_sink, float inv_sqrt_hd,
                       const uint16_t *v_sink, float *sink_out, void *stream) {
  if (!scores || !probs || !parts || n_parts <= 0 || H <= 0 || L <= 0 || n_sink < 0) return KVQ_EINVAL;
  if (n_sink > 0 && (!sink_scores || !sink_probs)) return KVQ_EINVAL;
  if (v_sink != nullptr && (n_sink <= 0 || !sink_out)) return KVQ_EINVAL;
  // 1024-lane workgroups: the (max, sum) merge of the row's partials (one per 256-token tile of the score
  // kernel) is paid once per 1024 lanes and the streaming part runs at 8 waves per SIMD
  const int nsplit = pick_split(H, L);
  dim3 grid(nsplit, H), block(1024);
  softmax_final_kernel<1024><<<grid, block, 0, (hipStream_t)stream>>>(
      scores, reinterpret_cast<const __half *>(sink_scores), parts, probs, reinterpret_cast<__half *>(sink_probs), L,
      n_sink, inv_sqrt_hd, nsplit, n_parts, 0, reinterpret_cast<const __half *>(v_sink), sink_out);
  return check_launch();
}

}  // extern "C"
