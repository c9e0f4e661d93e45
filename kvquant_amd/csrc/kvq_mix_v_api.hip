// C entry points of p.V (include/kvq.h: kvq_mix_v, kvq_mix_v_softmax, kvq_mix_v_workspace_bytes): argument checks and the
// choice between the streaming kernel (kvq_mix_v.hip) and the row-per-lane fallback for shapes it does not take
// (kvq_mix_v_rows.h).  Reference launchers: KCU:3491-3538, 3625-3690.
#include "kvq_mix_v_stage.h"
#include "kvq_mix_v_rows.h"

#include <cstdlib>

namespace kvq {

static size_t ws_bytes(int bits, int q_len, int H, int64_t L) {
  size_t a = mix_plan_bytes(bits, q_len, H, L);
  size_t b = plan_mix_rows(bits, q_len, H, L, true).bytes;
  return a > b ? a : b;
}

}  // namespace kvq

using namespace kvq;

extern "C" {

size_t kvq_mix_v_workspace_bytes(int bits, int q_len, int H, int hd, int64_t L) {
  if (bits < 2 || bits > 4 || q_len <= 0 || H <= 0 || hd != kHeadDim || L < 0) return 0;
  return ws_bytes(bits, q_len, H, L > 0 ? L : 1);
}

static bool mix_fast_shape(const int32_t *mat, const float *lut_rows, int H, int hd, int64_t L, int64_t max_len, int bits) {
  return (max_len % 4 == 0) && (max_len >= 4) && ((int64_t)H * (hd / 32 * bits) * max_len < (1ll << 31)) &&
         ((reinterpret_cast<uintptr_t>(mat) | reinterpret_cast<uintptr_t>(lut_rows)) % 16 == 0);
}

// p: the probabilities, or with `fs` the raw scores (fast shapes only)
static int mix_v_any(int bits, const float *p, const FusedSoftmax *fs, const int32_t *mat, float *mul, const float *lut_rows,
                     int q_len, int H, int hd, int64_t L, int64_t max_len, const float *outliers,
                     const int32_t *outlier_idx, int n_out, int accumulate, void *workspace, size_t workspace_bytes,
                     void *stream) {
  if (!p || !mat || !mul || !lut_rows || q_len <= 0 || H <= 0 || hd != kHeadDim || L < 0 || L > max_len ||
      bits < 2 || bits > 4)
    return KVQ_EINVAL;
  const bool sparse = outlier_idx != nullptr;     // (without `outliers`: compact rows, packed entries in outlier_idx)
  if ((outliers && !outlier_idx) || (sparse && n_out <= 0)) return KVQ_EINVAL;
  hipStream_t st = (hipStream_t)stream;
  if (L == 0) {
    if (!accumulate) {
      if (hipMemsetAsync(mul, 0, (size_t)q_len * H * hd * sizeof(float), st) != hipSuccess) return KVQ_ELAUNCH;
    }
    return KVQ_OK;
  }
  if (!workspace || workspace_bytes < ws_bytes(bits, q_len, H, L)) return KVQ_EWORKSPACE;
  const bool fast = mix_fast_shape(mat, lut_rows, H, hd, L, max_len, bits);
  if (!fast) {
    if (fs || (sparse && !outliers)) return KVQ_EINVAL;      // (the row-per-lane fallback reads the reference format only)
    MixPlan pl = plan_mix_rows(bits, q_len, H, L, sparse);
    MixVArgs a;
    a.p = p;
    a.mat = reinterpret_cast<const uint32_t *>(mat);
    a.lut_rows = lut_rows;
    a.outliers = outliers;
    a.idx = outlier_idx;
    a.partial = reinterpret_cast<float *>(workspace);
    a.H = H;
    a.q_len = q_len;
    a.L = L;
    a.max_len = max_len;
    a.tr = pl.tr;
    a.n_ranges = pl.n_ranges;
    a.ubg = pl.ubg;
    a.n_units = pl.n_units;
    a.n_out = n_out;
    switch (bits) {
      case 4: return launch_mix_rows<4>(a, pl, mul, accumulate, st);
      case 3: return launch_mix_rows<3>(a, pl, mul, accumulate, st);
      default: return launch_mix_rows<2>(a, pl, mul, accumulate, st);
    }
  }
  MixArgs a;
  a.p = p;
  a.mat = reinterpret_cast<const uint32_t *>(mat);
  a.lut_rows = lut_rows;
  a.outliers = outliers;
  a.idx = outlier_idx;
  a.partial = reinterpret_cast<float *>(workspace);
  a.H = H;
  a.q_len = q_len;
  a.L = L;
  a.max_len = max_len;
  a.tr = 0;
  a.groups = 1;
  a.n_units = 0;
  a.n_out = n_out;
  a.split = 0;
  a.n_out_magic = sparse ? (uint32_t)(((1ull << 32) + (uint64_t)n_out - 1) / (uint64_t)n_out) : 0u;
  a.scores = nullptr;
  a.mz = nullptr;
  a.parts = nullptr;
  a.n_parts = 0;
  a.sink = nullptr;
  a.sink_probs = nullptr;
  a.n_sink = 0;
  a.v_sink = nullptr;
  a.sink_out = nullptr;
  a.inv = 0.f;
#if KVQ_TRACE
  a.trace = reinterpret_cast<unsigned long long *>(strtoull(getenv("KVQ_TRACE_PTR") ? getenv("KVQ_TRACE_PTR") : "0", nullptr, 0));
  if (!a.trace) return KVQ_EINVAL;
#endif
  return launch_mix_bits(bits, a, mul, accumulate, st, fs);
}

int kvq_mix_v(int bits, const float *p, const int32_t *mat, float *mul, const float *lut_rows, int q_len,
              int H, int hd, int64_t L, int64_t max_len, const float *outliers, const int32_t *outlier_idx,
              int n_out, int accumulate, void *workspace, size_t workspace_bytes, void *stream) {
  return mix_v_any(bits, p, nullptr, mat, mul, lut_rows, q_len, H, hd, L, max_len, outliers, outlier_idx, n_out, accumulate,
                   workspace, workspace_bytes, stream);
}

int kvq_mix_v_softmax(int bits, const float *scores, const float *parts, int n_parts, float inv_sqrt_hd,
                      const uint16_t *sink_scores, uint16_t *sink_probs, int n_sink, const uint16_t *v_sink,
                      float *probs, const int32_t *mat, float *mul, const float *lut_rows, int H, int hd, int64_t L,
                      int64_t max_len, const float *outliers, const int32_t *outlier_idx, int n_out,
                      int accumulate, void *workspace, size_t workspace_bytes, void *stream) {
  if (!scores || !parts || n_parts <= 0 || n_sink < 0 || H <= 0 || L <= 0) return KVQ_EINVAL;
  if (n_sink > 0 && (!sink_scores || !sink_probs)) return KVQ_EINVAL;
  if (v_sink != nullptr && (n_sink <= 0 || accumulate)) return KVQ_EINVAL;
  const bool fast = mix_fast_shape(mat, lut_rows, H, hd, L, max_len, bits) && H <= VCfg<4>::MZ_HEADS;
  if (!fast) {
    // shapes the streaming kernel does not take: the two passes separately
    if (!probs) return KVQ_EINVAL;
    int rc = kvq_softmax_finish(scores, sink_scores, parts, n_parts, probs, sink_probs, H, L, n_sink, inv_sqrt_hd, v_sink,
                                mul, stream);
    if (rc) return rc;
    return mix_v_any(bits, probs, nullptr, mat, mul, lut_rows, 1, H, hd, L, max_len, outliers, outlier_idx, n_out,
                     v_sink ? 1 : accumulate, workspace, workspace_bytes, stream);
  }
  FusedSoftmax f;
  f.scores = scores;
  f.parts = parts;
  f.n_parts = n_parts;
  f.inv = inv_sqrt_hd;
  f.sink = reinterpret_cast<const __half *>(sink_scores);
  f.sink_probs = reinterpret_cast<__half *>(sink_probs);
  f.n_sink = n_sink;
  f.v_sink = reinterpret_cast<const __half *>(v_sink);
  return mix_v_any(bits, scores, &f, mat, mul, lut_rows, 1, H, hd, L, max_len, outliers, outlier_idx, n_out, accumulate,
                   workspace, workspace_bytes, stream);
}

}  // extern "C"
