// Head-shard extract (kvq_shard.hip), shared with the one-call head-shard step (kvq_decode_step.hip).
#pragma once
#include "kvq_common.h"
#include "kvq_host.h"

namespace kvq {

struct ExtractArgs {
  const int32_t *k_src, *v_src;
  int32_t *k_dst, *v_dst;
  int64_t src_max_len, src_col, dst_max_len, dst_col, n;
  int rows_w;                 // packed word rows of the shard (n_heads * W), first source row row0
  int row0;
  const float *k_out, *v_out;
  const int32_t *k_idx, *v_idx;
  float *k_out_d, *v_out_d, *k_out_t;
  int32_t *k_idx_d, *v_idx_d, *k_idx_t;
  int n_out, c0, cn, n_codes;
  const float *rows_src, *rows2_src;
  float *rows_dst, *rows2_dst;
  // table roles (optional)
  int n_tab;
  int bits;
  const float *lut;
  const void *q;
  int q_is_half;
  unsigned char *tab;
  float *q32;
  unsigned char *pair_tab;
  const __half *k_sink;
  __half *sink_scores;
  int n_sink;
  float sink_inv;
};


void fill_extract_args(ExtractArgs &a, int bits, int hd, int h0, int n_heads, int n_out, const int32_t *k_src, const int32_t *v_src,
                       int64_t src_max_len, int64_t src_col, const float *k_out_src, const int32_t *k_idx_src,
                       const float *v_out_src, const int32_t *v_idx_src, const float *v_rows_src, int32_t *k_dst, int32_t *v_dst,
                       int64_t dst_max_len, int64_t dst_col, float *k_out_dst, int32_t *k_idx_dst, float *k_out_t_dst,
                       int32_t *k_idx_t_dst, float *v_out_dst, int32_t *v_idx_dst, float *v_rows_dst, const float *v_rows2_src,
                       float *v_rows2_dst, int64_t n);
int launch_extract_fused(ExtractArgs a, hipStream_t st);

}  // namespace kvq
