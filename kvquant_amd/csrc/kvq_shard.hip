// Token-sharded attention of one decode stream (the context split along the token axis over the GPUs of a node):
// the three pieces a shard needs besides the ordinary decode kernels --
//   kvq_score_k_tables    the query-premultiplied K codebook images WITHOUT an append or a score launch (a shard that
//                         does not own the newest token only scores),
//   kvq_softmax_stats     (max, normaliser) of every softmax row of the shard from the score kernel's partials, i.e.
//                         what the merge across shards needs -- the non-per-token part of kvq_softmax_finish,
//   kvq_combine_shards    the exact merge of R shards' locally normalised outputs after ONE all-gather per layer.
// The reference has no counterpart (its multi-GPU placement is by layer, modeling_llama.py:2428-2453; see DESIGN.md 6
// for why the token split is the cut that speeds a single stream up).
#include "kvq_common.h"
#include "kvq_host.h"
#include "kvq_ktab.h"

namespace kvq {

__device__ __forceinline__ float mzw(float d) { return __builtin_amdgcn_exp2f(d * 1.4426950408889634f); }

// one workgroup per head: merge the (max, sum exp) partials of the head's tiles (same arithmetic as the merge inside
// kvq_softmax_finish / softmax_merge_kernel) -> stats[h] = (M, Z)
__global__ __launch_bounds__(256) void softmax_stats_kernel(const float *__restrict__ parts, int n_parts,
                                                            float *__restrict__ stats) {
  __shared__ float red[8];
  const int h = blockIdx.x, tid = threadIdx.x;
  const float2 *pr = reinterpret_cast<const float2 *>(parts) + (int64_t)h * n_parts;
  float M = -INFINITY, Z = 0.f;
  for (int i = tid; i < n_parts; i += 256) {
    const float2 ms = pr[i];
    if (ms.x > -INFINITY) {
      const float mn = fmaxf(M, ms.x);
      Z = Z * mzw(M - mn) + ms.y * mzw(ms.x - mn);
      M = mn;
    }
  }
#pragma unroll
  for (int d = 32; d >= 1; d >>= 1) {
    const float mo = __shfl_xor(M, d), zo = __shfl_xor(Z, d);
    const float mn = fmaxf(M, mo);
    Z = (mn == -INFINITY) ? 0.f : Z * mzw(M - mn) + zo * mzw(mo - mn);
    M = mn;
  }
  if ((tid & 63) == 0) { red[tid >> 6] = M; red[4 + (tid >> 6)] = Z; }
  __syncthreads();
  if (tid == 0) {
    const float Mb = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
    float Zb = 0.f;
#pragma unroll
    for (int i = 0; i < 4; i++)
      if (red[i] > -INFINITY) Zb += red[4 + i] * mzw(red[i] - Mb);
    stats[2 * h] = Mb;
    stats[2 * h + 1] = Zb;
  }
}

// out[h][c] = sum_r w_r out_r[h][c] / sum_r w_r,  w_r = Z_r exp(M_r - max_r M_r): every shard's output is normalised over
// its own tokens, (M_r, Z_r) are the shard's softmax maximum and normaliser.  packed: R records of
// [H*hd floats out][H x (M, Z)]; a shard without tokens carries (-inf, 0).  One workgroup per head, lane = channel.
__global__ __launch_bounds__(128) void combine_shards_kernel(const float *__restrict__ packed, int R, int H, int hd,
                                                             float *__restrict__ out) {
  const int h = blockIdx.x, c = threadIdx.x;
  const int64_t rec = (int64_t)H * hd + 2 * H;
  float Mg = -INFINITY;
  for (int r = 0; r < R; r++) Mg = fmaxf(Mg, packed[r * rec + (int64_t)H * hd + 2 * h]);
  float num = 0.f, den = 0.f;
  for (int r = 0; r < R; r++) {
    const float *p = packed + r * rec;
    const float M = p[(int64_t)H * hd + 2 * h], Z = p[(int64_t)H * hd + 2 * h + 1];
    if (!(M > -INFINITY) || !(Z > 0.f)) continue;            // (empty shard)
    const float w = Z * expf(M - Mg);
    den += w;
    if (c < hd) num = fmaf(w, p[(int64_t)h * hd + c], num);
  }
  if (c < hd) out[(int64_t)h * hd + c] = den > 0.f ? num / den : 0.f;
}

}  // namespace kvq

using namespace kvq;

extern "C" {

int kvq_score_k_tables(int bits, const void *q, int q_is_half, const float *lut, int H, int hd, void *workspace,
                       size_t workspace_bytes, void *stream) {
  if (!q || !lut || H <= 0 || hd != kHeadDim || bits < 2 || bits > 4) return KVQ_EINVAL;
  if (!workspace || workspace_bytes < kvq_score_k_workspace_bytes(bits, 1, H) || reinterpret_cast<uintptr_t>(workspace) % 16)
    return KVQ_EWORKSPACE;
  unsigned char *tab = reinterpret_cast<unsigned char *>(workspace);
  hipStream_t st = (hipStream_t)stream;
  switch (bits) {
    case 4: lutq_prep_kernel<4><<<dim3(H, 1), 256, 0, st>>>(lut, q, q_is_half, tab, reinterpret_cast<float *>(tab + (size_t)H * KTab<4>::BUF_B), H); break;
    case 3: lutq_prep_kernel<3><<<dim3(H, 1), 256, 0, st>>>(lut, q, q_is_half, tab, reinterpret_cast<float *>(tab + (size_t)H * KTab<3>::BUF_B), H); break;
    default: lutq_prep_kernel<2><<<dim3(H, 1), 256, 0, st>>>(lut, q, q_is_half, tab, reinterpret_cast<float *>(tab + (size_t)H * KTab<2>::BUF_B), H); break;
  }
  return check_launch();
}

int kvq_softmax_stats(const float *parts, int n_parts, int H, float *stats, void *stream) {
  if (!parts || !stats || n_parts <= 0 || H <= 0) return KVQ_EINVAL;
  softmax_stats_kernel<<<H, 256, 0, (hipStream_t)stream>>>(parts, n_parts, stats);
  return check_launch();
}

int kvq_combine_shards(const float *packed, int n_shards, int H, int hd, float *out, void *stream) {
  if (!packed || !out || n_shards <= 0 || H <= 0 || hd <= 0 || hd > 128) return KVQ_EINVAL;
  combine_shards_kernel<<<H, 128, 0, (hipStream_t)stream>>>(packed, n_shards, H, hd, out);
  return check_launch();
}

}  // extern "C"
