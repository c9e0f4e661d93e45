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


// Head window of full-width cache columns -> a shard's cache.  Packed words: row r of the shard = row h0*W + r of the
// source (W = hd/32*bits word rows per head), tokens along x (coalesced on both sides).
__global__ __launch_bounds__(256) void extract_words_kernel(const int32_t *__restrict__ src, int64_t src_max_len, int64_t src_col,
                                                            int32_t *__restrict__ dst, int64_t dst_max_len, int64_t dst_col,
                                                            int row0, int64_t n) {
  const int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x;
  const int r = blockIdx.y;
  if (t < n) dst[(int64_t)r * dst_max_len + dst_col + t] = src[(int64_t)(row0 + r) * src_max_len + src_col + t];
}

// Outlier rows [max_len][n_out] (value, global channel): an entry of the shard's heads keeps its value and gets its channel
// rebased; an entry of another rank's heads becomes (0, first / last channel of the shard) -- zero entries are skipped by
// the matvec kernels (as the reference's capped-away slots are, ML:745-747), and the row stays sorted by channel, which
// the row-format score kernel relies on.  `dst_t` / `dst_idx_t`: the token-contiguous mirror [n_out][max_len] (K) or null.
// rows_src / rows_dst: the per-token V codebook rows [max_len][n_codes] or null.
__global__ __launch_bounds__(256) void extract_outliers_kernel(const float *__restrict__ val, const int32_t *__restrict__ idx,
                                                               int64_t src_col, float *__restrict__ dval,
                                                               int32_t *__restrict__ didx, float *__restrict__ dval_t,
                                                               int32_t *__restrict__ didx_t, int64_t dst_max_len,
                                                               int64_t dst_col, int n_out, int c0, int cn,
                                                               const float *__restrict__ rows_src,
                                                               float *__restrict__ rows_dst, int n_codes, int64_t n) {
  const int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x;      // (token, slot) flat
  if (e < n * n_out) {
    const int64_t t = e / n_out;
    const int j = (int)(e % n_out);
    const float v = val[(src_col + t) * n_out + j];
    const int c = idx[(src_col + t) * n_out + j] - c0;
    const bool own = c >= 0 && c < cn;
    const float ov = own ? v : 0.f;
    const int oc = own ? c : (c < 0 ? 0 : cn - 1);
    dval[(dst_col + t) * n_out + j] = ov;
    didx[(dst_col + t) * n_out + j] = oc;
    if (dval_t) {
      dval_t[(int64_t)j * dst_max_len + dst_col + t] = ov;
      didx_t[(int64_t)j * dst_max_len + dst_col + t] = oc;
    }
  }
  if (rows_src && e < n * n_codes) rows_dst[dst_col * n_codes + e] = rows_src[src_col * n_codes + e];
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
    case 4: lutq_prep_kernel<4><<<dim3(H, 1), 256, 0, st>>>(lut, q, q_is_half, tab, reinterpret_cast<float *>(tab + ktab_q_offset<4>(1, H)), KTabHasPair<4>::value ? tab + ktab_pair_offset<4>(1, H) : nullptr, H); break;
    case 3: lutq_prep_kernel<3><<<dim3(H, 1), 256, 0, st>>>(lut, q, q_is_half, tab, reinterpret_cast<float *>(tab + ktab_q_offset<3>(1, H)), KTabHasPair<3>::value ? tab + ktab_pair_offset<3>(1, H) : nullptr, H); break;
    default: lutq_prep_kernel<2><<<dim3(H, 1), 256, 0, st>>>(lut, q, q_is_half, tab, reinterpret_cast<float *>(tab + ktab_q_offset<2>(1, H)), KTabHasPair<2>::value ? tab + ktab_pair_offset<2>(1, H) : nullptr, H); break;
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


int kvq_extract_heads(int bits, int H, int hd, int h0, int n_heads, int n_out, const int32_t *k_src, const int32_t *v_src,
                      int64_t src_max_len, int64_t src_col, const float *k_out_src, const int32_t *k_idx_src,
                      const float *v_out_src, const int32_t *v_idx_src, const float *v_rows_src, int32_t *k_dst,
                      int32_t *v_dst, int64_t dst_max_len, int64_t dst_col, float *k_out_dst, int32_t *k_idx_dst,
                      float *k_out_t_dst, int32_t *k_idx_t_dst, float *v_out_dst, int32_t *v_idx_dst, float *v_rows_dst,
                      int64_t n, void *stream) {
  if (bits < 2 || bits > 4 || hd != kHeadDim || H <= 0 || h0 < 0 || n_heads <= 0 || h0 + n_heads > H || n < 0) return KVQ_EINVAL;
  if (!k_src || !v_src || !k_dst || !v_dst || !v_rows_src || !v_rows_dst) return KVQ_EINVAL;
  if (src_col < 0 || dst_col < 0 || src_col + n > src_max_len || dst_col + n > dst_max_len) return KVQ_EINVAL;
  const bool sparse = n_out > 0;
  if (sparse && (!k_out_src || !k_idx_src || !v_out_src || !v_idx_src || !k_out_dst || !k_idx_dst || !v_out_dst || !v_idx_dst ||
                 ((k_out_t_dst == nullptr) != (k_idx_t_dst == nullptr))))
    return KVQ_EINVAL;
  if (n == 0) return KVQ_OK;
  hipStream_t st = (hipStream_t)stream;
  const int W = hd / 32 * bits, n_codes = 1 << bits;
  dim3 wgrid((unsigned)((n + 255) / 256), (unsigned)(n_heads * W));
  extract_words_kernel<<<wgrid, 256, 0, st>>>(k_src, src_max_len, src_col, k_dst, dst_max_len, dst_col, h0 * W, n);
  extract_words_kernel<<<wgrid, 256, 0, st>>>(v_src, src_max_len, src_col, v_dst, dst_max_len, dst_col, h0 * W, n);
  const int c0 = h0 * hd, cn = n_heads * hd;
  const int per = sparse ? (n_out > n_codes ? n_out : n_codes) : n_codes;
  const unsigned eblocks = (unsigned)((n * per + 255) / 256);
  if (sparse) {
    extract_outliers_kernel<<<eblocks, 256, 0, st>>>(k_out_src, k_idx_src, src_col, k_out_dst, k_idx_dst, k_out_t_dst,
                                                     k_idx_t_dst, dst_max_len, dst_col, n_out, c0, cn, nullptr, nullptr, 0, n);
    extract_outliers_kernel<<<eblocks, 256, 0, st>>>(v_out_src, v_idx_src, src_col, v_out_dst, v_idx_dst, nullptr, nullptr,
                                                     dst_max_len, dst_col, n_out, c0, cn, v_rows_src, v_rows_dst, n_codes, n);
  } else {
    // (dense caches: only the codebook rows travel; the kernel's outlier half sees n_out = 0 and touches nothing)
    extract_outliers_kernel<<<eblocks, 256, 0, st>>>(nullptr, nullptr, src_col, nullptr, nullptr, nullptr, nullptr, dst_max_len,
                                                     dst_col, 0, c0, cn, v_rows_src, v_rows_dst, n_codes, n);
  }
  return check_launch();
}

}  // extern "C"
