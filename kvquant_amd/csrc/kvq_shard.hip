// Sharded attention of one decode stream.  (1) TOKEN-sharded (the context split along the token axis over the GPUs of a node):
// the three pieces a shard needs besides the ordinary decode kernels --
//   kvq_score_k_tables    the query-premultiplied K codebook images WITHOUT an append or a score launch (a shard that
//                         does not own the newest token only scores),
//   kvq_softmax_stats     (max, normaliser) of every softmax row of the shard from the score kernel's partials, i.e.
//                         what the merge across shards needs -- the non-per-token part of kvq_softmax_finish,
//   kvq_combine_shards    the exact merge of R shards' locally normalised outputs after ONE all-gather per layer.
// The reference has no counterpart (its multi-GPU placement is by layer, modeling_llama.py:2428-2453; see DESIGN.md 6
// for why the token split is the cut that speeds a single stream up).
// (2) HEAD-sharded (SURVEY 8e "by head"): every rank holds H / N heads of every layer for ALL tokens.  The outlier
// selection of a token is over all H*hd channels (top-21 / bottom-21 of the whole token, ML:742, 1093-1096; V's codebook
// row comes from the token's 22nd largest / smallest value), so a rank appends the WHOLE token into a full-width staging
// column with the ordinary append kernels -- bit-identical selection on every rank -- and
//   kvq_extract_heads     copies its heads' packed words, the token's V codebook row and its heads' share of the 42
//                         outlier entries (channels rebased to the shard, foreign entries zeroed) into the shard's cache.
#include "kvq_common.h"
#include "kvq_host.h"
#include "kvq_ktab.h"
#include "kvq_shard.h"

namespace kvq {

__device__ __forceinline__ float mzw(float d) { return __builtin_amdgcn_exp2f(d * 1.4426950408889634f); }

// one workgroup per head: merge the (max, sum exp) partials of the head's tiles (same arithmetic as the merge inside
// kvq_softmax_finish / softmax_merge_kernel) -> stats[h] = (M, Z)
__global__ __launch_bounds__(256) void softmax_stats_kernel(const float *__restrict__ parts, int n_parts,
                                                            const __half *__restrict__ sink, int n_sink,
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
  // (the fp16 sink tokens of the shard that holds them: their scaled scores join the row like any other token's)
  for (int i = tid; i < n_sink; i += 256) {
    const float x = __half2float(sink[h * n_sink + i]);
    const float mn = fmaxf(M, x);
    Z = Z * expf(M - mn) + expf(x - mn);
    M = mn;
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


// scaled scores of the fp16 sink tokens, one workgroup per head: half(half(q . k_sink) * inv_sqrt_hd) (ML:1950-1962; what
// the table workgroups of kvq_decode_prologue write when the step appends)
__global__ __launch_bounds__(64) void sink_scores_kernel(const void *__restrict__ q, int q_is_half, const __half *__restrict__ k_sink,
                                                         __half *__restrict__ sink_scores, int n_sink, float inv) {
  const int h = blockIdx.x;
  for (int i = threadIdx.x; i < n_sink; i += 64) {
    float acc = 0.f;
    for (int c = 0; c < kHeadDim; c++)
      acc = fmaf(ld_act(q, h * kHeadDim + c, q_is_half), __half2float(k_sink[((int64_t)h * kHeadDim + c) * n_sink + i]), acc);
    sink_scores[h * n_sink + i] = __float2half_rn(scaled(acc, inv));
  }
}

// Everything a head shard needs from the staging column(s) in ONE launch -- and, for the decode step, the query tables of
// its heads in the same launch (kvq_head_shard_step: append -> THIS -> q.K^T -> p.V): blocks [0, n_tab) build the table
// of one head each (+ that head's fp16 sink scores), the others copy, grid-stride over one flat index space:
// [K words | V words | K outlier entries | V outlier entries | V codebook rows | Q-Norm rows].
__device__ __forceinline__ void extract_entry(const float *val, const int32_t *idx, float *dval, int32_t *didx, float *dval_t,
                                              int32_t *didx_t, const ExtractArgs &a, int64_t e) {
  const int64_t t = e / a.n_out;
  const int j = (int)(e % a.n_out);
  const float v = val[(a.src_col + t) * a.n_out + j];
  const int c = idx[(a.src_col + t) * a.n_out + j] - a.c0;
  const bool own = c >= 0 && c < a.cn;
  const float ov = own ? v : 0.f;
  const int oc = own ? c : (c < 0 ? 0 : a.cn - 1);
  dval[(a.dst_col + t) * a.n_out + j] = ov;
  didx[(a.dst_col + t) * a.n_out + j] = oc;
  if (dval_t) {
    dval_t[(int64_t)j * a.dst_max_len + a.dst_col + t] = ov;
    didx_t[(int64_t)j * a.dst_max_len + a.dst_col + t] = oc;
  }
}

__global__ __launch_bounds__(256) void extract_fused_kernel(ExtractArgs a) {
  if ((int)blockIdx.x < a.n_tab) {
    const int h = blockIdx.x;
    switch (a.bits) {
      case 4: lutq_prep_head<4>(a.lut, a.q, a.q_is_half, a.tab, a.q32, a.pair_tab, a.n_tab, h, 0); break;
      case 3: lutq_prep_head<3>(a.lut, a.q, a.q_is_half, a.tab, a.q32, a.pair_tab, a.n_tab, h, 0); break;
      default: lutq_prep_head<2>(a.lut, a.q, a.q_is_half, a.tab, a.q32, a.pair_tab, a.n_tab, h, 0); break;
    }
    if (a.k_sink != nullptr)
      for (int i = threadIdx.x; i < a.n_sink; i += 256) {
        float acc = 0.f;
        for (int c = 0; c < kHeadDim; c++)
          acc = fmaf(ld_act(a.q, h * kHeadDim + c, a.q_is_half), __half2float(a.k_sink[((int64_t)h * kHeadDim + c) * a.n_sink + i]), acc);
        a.sink_scores[h * a.n_sink + i] = __float2half_rn(scaled(acc, a.sink_inv));
      }
    return;
  }
  const int64_t nw = (int64_t)a.rows_w * a.n, ne = a.n_out > 0 ? a.n * a.n_out : 0, nr = a.n * a.n_codes;
  const int64_t total = 2 * nw + 2 * ne + nr + (a.rows2_src ? nr : 0);
  const int64_t stride = (int64_t)(gridDim.x - a.n_tab) * 256;
  for (int64_t i = (int64_t)(blockIdx.x - a.n_tab) * 256 + threadIdx.x; i < total; i += stride) {
    int64_t e = i;
    if (e < 2 * nw) {
      const bool isv = e >= nw;
      if (isv) e -= nw;
      const int r = (int)(e / a.n);
      const int64_t t = e % a.n;
      const int32_t *src = isv ? a.v_src : a.k_src;
      int32_t *dst = isv ? a.v_dst : a.k_dst;
      dst[(int64_t)r * a.dst_max_len + a.dst_col + t] = src[(int64_t)(a.row0 + r) * a.src_max_len + a.src_col + t];
      continue;
    }
    e -= 2 * nw;
    if (e < ne) { extract_entry(a.k_out, a.k_idx, a.k_out_d, a.k_idx_d, a.k_out_t, a.k_idx_t, a, e); continue; }
    e -= ne;
    if (e < ne) { extract_entry(a.v_out, a.v_idx, a.v_out_d, a.v_idx_d, nullptr, nullptr, a, e); continue; }
    e -= ne;
    if (e < nr) { a.rows_dst[a.dst_col * a.n_codes + e] = a.rows_src[a.src_col * a.n_codes + e]; continue; }
    e -= nr;
    a.rows2_dst[a.dst_col * a.n_codes + e] = a.rows2_src[a.src_col * a.n_codes + e];
  }
}

void fill_extract_args(ExtractArgs &a, int bits, int hd, int h0, int n_heads, int n_out, const int32_t *k_src, const int32_t *v_src,
                       int64_t src_max_len, int64_t src_col, const float *k_out_src, const int32_t *k_idx_src,
                       const float *v_out_src, const int32_t *v_idx_src, const float *v_rows_src, int32_t *k_dst, int32_t *v_dst,
                       int64_t dst_max_len, int64_t dst_col, float *k_out_dst, int32_t *k_idx_dst, float *k_out_t_dst,
                       int32_t *k_idx_t_dst, float *v_out_dst, int32_t *v_idx_dst, float *v_rows_dst, const float *v_rows2_src,
                       float *v_rows2_dst, int64_t n) {
  const int W = hd / 32 * bits;
  a.k_src = k_src; a.v_src = v_src; a.k_dst = k_dst; a.v_dst = v_dst;
  a.src_max_len = src_max_len; a.src_col = src_col; a.dst_max_len = dst_max_len; a.dst_col = dst_col; a.n = n;
  a.rows_w = n_heads * W; a.row0 = h0 * W;
  a.k_out = k_out_src; a.k_idx = k_idx_src; a.v_out = v_out_src; a.v_idx = v_idx_src;
  a.k_out_d = k_out_dst; a.k_idx_d = k_idx_dst; a.k_out_t = k_out_t_dst; a.k_idx_t = k_idx_t_dst;
  a.v_out_d = v_out_dst; a.v_idx_d = v_idx_dst;
  a.n_out = n_out; a.c0 = h0 * hd; a.cn = n_heads * hd; a.n_codes = 1 << bits;
  a.rows_src = v_rows_src; a.rows_dst = v_rows_dst; a.rows2_src = v_rows2_src; a.rows2_dst = v_rows2_dst;
  a.n_tab = 0; a.bits = bits; a.lut = nullptr; a.q = nullptr; a.q_is_half = 0; a.tab = nullptr; a.q32 = nullptr; a.pair_tab = nullptr;
  a.k_sink = nullptr; a.sink_scores = nullptr; a.n_sink = 0; a.sink_inv = 0.f;
}

// kvq_extract_heads as one launch, optionally with the shard's query tables (+ sink scores) as extra roles of it
int launch_extract_fused(ExtractArgs a, hipStream_t st) {
  const int64_t nw = (int64_t)a.rows_w * a.n, ne = a.n_out > 0 ? a.n * a.n_out : 0, nr = a.n * a.n_codes;
  const int64_t total = 2 * nw + 2 * ne + nr + (a.rows2_src ? nr : 0);
  int64_t blocks = (total + 255) / 256;
  if (blocks > 4096) blocks = 4096;
  if (blocks < 1) blocks = 1;
  extract_fused_kernel<<<(unsigned)(blocks + a.n_tab), 256, 0, st>>>(a);
  return check_launch();
}

}  // namespace kvq

using namespace kvq;

extern "C" {

int kvq_score_k_tables(int bits, const void *q, int q_is_half, const float *lut, int H, int hd, const kvq_sinks *sinks,
                       void *workspace, size_t workspace_bytes, void *stream) {
  if (!q || !lut || H <= 0 || hd != kHeadDim || bits < 2 || bits > 4) return KVQ_EINVAL;
  if (sinks != nullptr && sinks->n_sink > 0 && (!sinks->k_sink || !sinks->sink_scores)) return KVQ_EINVAL;
  if (!workspace || workspace_bytes < kvq_score_k_workspace_bytes(bits, 1, H) || reinterpret_cast<uintptr_t>(workspace) % 16)
    return KVQ_EWORKSPACE;
  unsigned char *tab = reinterpret_cast<unsigned char *>(workspace);
  hipStream_t st = (hipStream_t)stream;
  switch (bits) {
    case 4: lutq_prep_kernel<4><<<dim3(H, 1), 256, 0, st>>>(lut, q, q_is_half, tab, reinterpret_cast<float *>(tab + ktab_q_offset<4>(1, H)), KTabHasPair<4>::value ? tab + ktab_pair_offset<4>(1, H) : nullptr, H); break;
    case 3: lutq_prep_kernel<3><<<dim3(H, 1), 256, 0, st>>>(lut, q, q_is_half, tab, reinterpret_cast<float *>(tab + ktab_q_offset<3>(1, H)), KTabHasPair<3>::value ? tab + ktab_pair_offset<3>(1, H) : nullptr, H); break;
    default: lutq_prep_kernel<2><<<dim3(H, 1), 256, 0, st>>>(lut, q, q_is_half, tab, reinterpret_cast<float *>(tab + ktab_q_offset<2>(1, H)), KTabHasPair<2>::value ? tab + ktab_pair_offset<2>(1, H) : nullptr, H); break;
  }
  int rc = check_launch();
  if (rc) return rc;
  if (sinks != nullptr && sinks->n_sink > 0) {
    sink_scores_kernel<<<H, 64, 0, st>>>(q, q_is_half, reinterpret_cast<const __half *>(sinks->k_sink),
                                         reinterpret_cast<__half *>(sinks->sink_scores), sinks->n_sink, sinks->inv_sqrt_hd);
    rc = check_launch();
  }
  return rc;
}

int kvq_softmax_stats(const float *parts, int n_parts, const uint16_t *sink_scores, int n_sink, int H, float *stats,
                      void *stream) {
  if (!parts || !stats || n_parts <= 0 || H <= 0 || n_sink < 0 || (n_sink > 0 && !sink_scores)) return KVQ_EINVAL;
  softmax_stats_kernel<<<H, 256, 0, (hipStream_t)stream>>>(parts, n_parts, reinterpret_cast<const __half *>(sink_scores),
                                                           n_sink, stats);
  return check_launch();
}

int kvq_combine_shards(const float *packed, int n_shards, int H, int hd, float *out, void *stream) {
  if (!packed || !out || n_shards <= 0 || H <= 0 || hd <= 0 || hd > 128) return KVQ_EINVAL;
  combine_shards_kernel<<<H, 128, 0, (hipStream_t)stream>>>(packed, n_shards, H, hd, out);
  return check_launch();
}


int kvq_extract_heads(int bits, int H, int hd, int h0, int n_heads, int n_out, const int32_t *k_src, const int32_t *v_src,
                      int64_t src_max_len, int64_t src_col, const float *k_out_src, const int32_t *k_idx_src,
                      const float *v_out_src, const int32_t *v_idx_src, const float *v_rows_src, int32_t *k_dst,
                      int32_t *v_dst, int64_t dst_max_len, int64_t dst_col, float *k_out_dst, int32_t *k_idx_dst,
                      float *k_out_t_dst, int32_t *k_idx_t_dst, float *v_out_dst, int32_t *v_idx_dst, float *v_rows_dst,
                      const float *v_rows2_src, float *v_rows2_dst, int64_t n, void *stream) {
  if (bits < 2 || bits > 4 || hd != kHeadDim || H <= 0 || h0 < 0 || n_heads <= 0 || h0 + n_heads > H || n < 0) return KVQ_EINVAL;
  if (!k_src || !v_src || !k_dst || !v_dst || !v_rows_src || !v_rows_dst || ((v_rows2_src == nullptr) != (v_rows2_dst == nullptr)))
    return KVQ_EINVAL;
  if (src_col < 0 || dst_col < 0 || src_col + n > src_max_len || dst_col + n > dst_max_len) return KVQ_EINVAL;
  const bool sparse = n_out > 0;
  if (sparse && (!k_out_src || !k_idx_src || !v_out_src || !v_idx_src || !k_out_dst || !k_idx_dst || !v_out_dst || !v_idx_dst ||
                 ((k_out_t_dst == nullptr) != (k_idx_t_dst == nullptr))))
    return KVQ_EINVAL;
  if (n == 0) return KVQ_OK;
  ExtractArgs a;
  fill_extract_args(a, bits, hd, h0, n_heads, n_out, k_src, v_src, src_max_len, src_col, k_out_src, k_idx_src, v_out_src, v_idx_src,
                    v_rows_src, k_dst, v_dst, dst_max_len, dst_col, k_out_dst, k_idx_dst, k_out_t_dst, k_idx_t_dst, v_out_dst,
                    v_idx_dst, v_rows_dst, v_rows2_src, v_rows2_dst, n);
  return launch_extract_fused(a, (hipStream_t)stream);
}

}  // extern "C"
