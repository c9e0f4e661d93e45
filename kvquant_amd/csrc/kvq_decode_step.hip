// kvq_decode_step: one decode token through ONE layer's compressed KV path in a single call -- the launch sequence
// of kvquant_amd.cache.decode_kv (prologue -> q.K^T + softmax partials -> softmax finish -> p.V -> slab reduce, or
// the 4-launch form with the softmax inside the p.V kernel for short caches) issued from C++ instead of from five
// Python / ctypes round trips.  What it replaces in the reference: the per-token body of the patched
// LlamaAttention.forward around QuantK.forward_fused_sparse / QuantV.forward_fused_sparse (ML:1930-2000), i.e. ~25
// torch launches, two host top-k round trips and a side stream per layer.
//
// Why a call and not a hipGraph: every launch depends on the cache length (grid sizes, partial counts, append
// column), so a captured graph would have to be re-parameterised every token; the boundary between two dependent
// kernels costs the same eager or replayed (MI355X_MICROARCH.md price list: "boundary ... eager == hipGraph"), and
// what made short contexts host-bound was the ~78 us of interpreter time per layer, not the launches themselves
// (~4 us each from C++).  kvq_decode_steps takes a whole stack of layers for callers whose layers follow each other
// without other work in between (the KV-path benchmark).
#include "kvq_common.h"
#include "kvq_host.h"
#include "kvq_ktab.h"
#include "kvq_shard.h"

namespace kvq {

struct StepPlan {
  size_t score_ws, parts_off, parts_b, mix_off, mix_b, scores_off, scores_b, probs_off, probs_b, total;
  int n_parts;
};

static size_t align256(size_t x) { return (x + 255) & ~(size_t)255; }

static StepPlan plan_step(int bits, int H, int hd, int64_t L) {
  StepPlan p;
  p.score_ws = align256(kvq_score_k_workspace_bytes(bits, 1, H));
  p.n_parts = kvq_score_k_softmax_parts(bits, L, 1);
  p.parts_off = p.score_ws;
  p.parts_b = align256((size_t)H * (p.n_parts > 0 ? p.n_parts : 1) * 8);
  p.mix_off = p.parts_off + p.parts_b;
  p.mix_b = align256(kvq_mix_v_workspace_bytes(bits, 1, H, hd, L));
  p.scores_off = p.mix_off + p.mix_b;
  p.scores_b = align256((size_t)H * L * 4);
  {   // (the fused kernel keeps its tile slabs and statistics where the other routes keep the scores)
    const size_t fb = align256(kvq_fused_attend_workspace_bytes(bits, H, hd, L));
    if (fb > p.scores_b) p.scores_b = fb;
  }
  p.probs_off = p.scores_off + p.scores_b;
  p.probs_b = p.scores_b;
  p.total = p.probs_off + p.probs_b;
  return p;
}

// measurement hook: events recorded around the two matvec launches of the next kvq_decode_step on this thread
static thread_local hipEvent_t *step_events = nullptr;

static thread_local bool mark2_pending = false;
// which pair-sum image of the 3-bit score tables a layer's step builds and reads (kvq.h: KVQ_LAYER_SCORE_*): the fp32 one from
// 16K cached tokens on (below, the per-channel tables, which are always built), else the fp16 one if asked for
static inline int pair_mode_of(const kvq_layer *ly, int64_t L) {
  if (ly->bits != 3) return 0;
  if ((ly->flags & KVQ_LAYER_SCORE_F32_PAIR) && L >= 16384 && ly->kidx_t != nullptr) return 2;
  return (ly->flags & KVQ_LAYER_SCORE_F16_PAIR) ? 1 : 0;
}
static inline int score_flags_of(const kvq_layer *ly, int64_t L) {
  const int m = pair_mode_of(ly, L);
  return m == 2 ? KVQ_SCORE_F32_PAIR_TABLES : (m == 1 ? KVQ_SCORE_F16_PAIR_TABLES : 0);
}

static thread_local int last_route = -1;      // which launch sequence the last kvq_decode_step on this thread took
static thread_local int step_fused_mark = 0;
static void record(int i, hipStream_t st) {
  if (step_events && step_events[i]) (void)hipEventRecord(step_events[i], st);
}

}  // namespace kvq

using namespace kvq;

extern "C" {

// called by the fused p.V route right in front of its p.V kernel (kvq_mix_v.hip)
void kvq_step_mark_pv(hipStream_t st) {
  if (mark2_pending) {
    mark2_pending = false;
    record(2, st);
  }
}

// called by kvq_fused_attend between its two launches (kvq_fused_decode.hip)
void kvq_step_mark_fused(hipStream_t st) {
  if (step_fused_mark) {
    record(1, st);
    record(2, st);
  }
}

int kvq_decode_step_route(void) { return last_route; }

int kvq_decode_step_events(void *const *events4) {
  step_events = reinterpret_cast<hipEvent_t *>(const_cast<void **>(events4));
  return KVQ_OK;
}

// softmax + p.V (+ slab reduce) of a step, given the raw scores and the softmax partials
static int step_tail(const kvq_layer *ly, const kvq_sinks *sinks, const uint16_t *v_sink, uint16_t *sink_probs, float *out,
                     int fuse_softmax, float *scores, float *probs, float *parts, int n_parts, unsigned char *ws,
                     const StepPlan &p, int64_t L, int n_out, int n_sink, float inv, hipStream_t st) {
  const int bits = ly->bits, H = ly->H, hd = ly->hd;
  void *stream = (void *)st;
  int rc;
  const float *vrows = ly->v_mix_rows ? ly->v_mix_rows : ly->vlut_rows;
  const uint16_t *sink_scores = sinks ? sinks->sink_scores : nullptr;
  if (fuse_softmax) {
    // (event 2 goes behind the small softmax-merge launch, in front of the p.V kernel: kvq_step_mark_pv below)
    mark2_pending = true;
    rc = kvq_mix_v_softmax(bits, scores, parts, n_parts, inv, sink_scores, sink_probs, n_sink, v_sink, probs, ly->vmat,
                           out, vrows, H, hd, L, ly->max_len, ly->voutliers, ly->vidx, n_out, 0, ws + p.mix_off, p.mix_b,
                           stream);
    if (mark2_pending) { mark2_pending = false; record(2, st); }   // (a route that does not pass the mark: the whole call)
    record(3, st);
    return rc;
  }
  rc = kvq_softmax_finish(scores, sink_scores, parts, n_parts, probs, sink_probs, H, L, n_sink, inv, v_sink, out, stream);
  if (rc) return rc;
  record(2, st);
  rc = kvq_mix_v(bits, probs, ly->vmat, out, vrows, 1, H, hd, L, ly->max_len, ly->voutliers, ly->vidx, n_out,
                 v_sink ? 1 : 0, ws + p.mix_off, p.mix_b, stream);
  record(3, st);
  return rc;
}

size_t kvq_decode_step_workspace_bytes(int bits, int H, int hd, int64_t L) {
  if (bits < 2 || bits > 4 || H <= 0 || hd != kHeadDim || L <= 0) return 0;
  return plan_step(bits, H, hd, L).total;
}

int kvq_decode_step(const kvq_layer *ly, int64_t kcol, int64_t vcol, const void *q, const void *k, const void *v,
                    int acts_are_half, const kvq_sinks *sinks, const uint16_t *v_sink, uint16_t *sink_probs,
                    float *out, int fuse_softmax, void *workspace, size_t workspace_bytes, void *stream) {
  if (!ly || !q || !k || !v || !out || kcol < 0 || vcol != kcol) return KVQ_EINVAL;
  // Dense-and-Sparse caches with the K mirror; value arrays may be absent (COMPACT formats: packed entries in the index
  // arrays, include/kvq.h)
  if (!ly->kidx_t || !ly->vidx || (ly->koutliers && !ly->kidx) || (ly->koutliers_t && !ly->kidx_t)) return KVQ_EINVAL;
  if (v_sink && (!sinks || !sink_probs)) return KVQ_EINVAL;
  const int bits = ly->bits, H = ly->H, hd = ly->hd;
  const int64_t L = kcol + 1;                       // cached tokens after the append (sink tokens not counted)
  if (L > ly->max_len) return KVQ_EINVAL;
  const StepPlan p = plan_step(bits, H, hd, L);
  if (p.n_parts <= 0) return KVQ_EINVAL;
  if (!workspace || workspace_bytes < p.total || reinterpret_cast<uintptr_t>(workspace) % 256) return KVQ_EWORKSPACE;
  unsigned char *ws = reinterpret_cast<unsigned char *>(workspace);
  float *parts = reinterpret_cast<float *>(ws + p.parts_off);
  float *scores = reinterpret_cast<float *>(ws + p.scores_off);
  float *probs = reinterpret_cast<float *>(ws + p.probs_off);
  const int n_out = 2 * ly->thr_k;
  const int n_sink = sinks ? sinks->n_sink : 0;
  const float inv = 1.0f / sqrtf((float)hd);
  const float *ktab = ly->klut_score ? ly->klut_score : ly->klut;
  int rc = decode_prologue(bits, ly->kmat, ly->klut, ly->klut_off, k, ly->klo, ly->khi, ly->koutliers, ly->kidx, kcol,
                           ly->vmat, ly->vlut_rows, ly->vlut_sorted, v, ly->voutliers, ly->vidx, vcol, q,
                           acts_are_half, ly->thr_k, H, hd, ly->max_len, ly->koutliers_t, ly->kidx_t, ly->klut_ends,
                           ly->klut_score, ly->vnorm, sinks, ws, p.score_ws,
                           pair_mode_of(ly, L), stream);
  if (rc) return rc;
  hipStream_t st = (hipStream_t)stream;
  struct Clear { ~Clear() { step_events = nullptr; } } clear_events_on_exit;
  if (fuse_softmax == 3 && ly->koutliers_t && ly->voutliers && !ly->v_mix_rows && !ly->klut_score &&
      kvq_fused_attend_supported(bits, H, hd, L, ly->max_len, n_out)) {
    // one kernel for q.K^T + softmax + p.V of every 256-token tile, then the merge (events: 0-1 the kernel, 2-3 the merge)
    last_route = 3;
    record(0, st);
    step_fused_mark = 1;
    rc = kvq_fused_attend(bits, ly->kmat, ktab, ws, ly->vmat, ly->vlut_rows, H, hd, L, ly->max_len, ly->rope_theta,
                          ly->pos_offset, ly->koutliers_t, ly->kidx_t, ly->voutliers, ly->vidx, n_out, inv,
                          sinks ? sinks->sink_scores : nullptr, sink_probs, n_sink, v_sink, out, ws + p.scores_off, p.scores_b,
                          stream);
    step_fused_mark = 0;
    record(3, st);
    return rc;
  }
  last_route = fuse_softmax ? 1 : 0;
  record(0, st);
  rc = kvq_score_k_prepared_softmax_ex(bits, ly->kmat, scores, ktab, H, hd, L, ly->max_len, ly->rope_theta,
                                       ly->pos_offset, ly->koutliers, ly->kidx, n_out, ly->koutliers_t, ly->kidx_t, ws,
                                       p.score_ws, inv, parts, p.n_parts,
                                       score_flags_of(ly, L), stream);
  record(1, st);
  if (rc) return rc;
  return step_tail(ly, sinks, v_sink, sink_probs, out, fuse_softmax, scores, probs, parts, p.n_parts, ws, p, L, n_out, n_sink, inv, st);
}

/* One decode token's attention over a layer's compressed cache WITHOUT an append: query tables (+ fp16 sink scores) ->
 * q.K^T -> softmax -> p.V over the L cached tokens (the launches of kvq_decode_step behind its prologue).  For shards of
 * a split stream: the tokens were appended elsewhere (head shard: through the staging cache + kvq_extract_heads), or this
 * shard does not hold the newest token at all. */
static int attend_step(const kvq_layer *ly, int64_t L, const void *q, int acts_are_half, const kvq_sinks *sinks,
                       const uint16_t *v_sink, uint16_t *sink_probs, float *out, int fuse_softmax, void *workspace,
                       size_t workspace_bytes, void *stream, bool tables_ready) {
  if (!ly || !q || !out || L <= 0 || L > ly->max_len) return KVQ_EINVAL;
  if (!ly->kidx_t || !ly->vidx || (ly->koutliers && !ly->kidx) || (ly->koutliers_t && !ly->kidx_t)) return KVQ_EINVAL;
  if (v_sink && (!sinks || !sink_probs)) return KVQ_EINVAL;
  const int bits = ly->bits, H = ly->H, hd = ly->hd;
  const StepPlan p = plan_step(bits, H, hd, L);
  if (p.n_parts <= 0) return KVQ_EINVAL;
  if (!workspace || workspace_bytes < p.total || reinterpret_cast<uintptr_t>(workspace) % 256) return KVQ_EWORKSPACE;
  unsigned char *ws = reinterpret_cast<unsigned char *>(workspace);
  float *parts = reinterpret_cast<float *>(ws + p.parts_off);
  float *scores = reinterpret_cast<float *>(ws + p.scores_off);
  float *probs = reinterpret_cast<float *>(ws + p.probs_off);
  const int n_out = 2 * ly->thr_k;
  const int n_sink = sinks ? sinks->n_sink : 0;
  const float inv = 1.0f / sqrtf((float)hd);
  const float *ktab = ly->klut_score ? ly->klut_score : ly->klut;
  int rc = tables_ready ? KVQ_OK : kvq_score_k_tables(bits, q, acts_are_half, ktab, H, hd, sinks, ws, p.score_ws, stream);
  if (rc) return rc;
  hipStream_t st = (hipStream_t)stream;
  struct Clear { ~Clear() { step_events = nullptr; } } clear_events_on_exit;
  last_route = fuse_softmax ? 1 : 0;
  record(0, st);
  rc = kvq_score_k_prepared_softmax_ex(bits, ly->kmat, scores, ktab, H, hd, L, ly->max_len, ly->rope_theta, ly->pos_offset,
                                       ly->koutliers, ly->kidx, n_out, ly->koutliers_t, ly->kidx_t, ws, p.score_ws, inv,
                                       parts, p.n_parts,
                                       (ly->flags & KVQ_LAYER_SCORE_F16_PAIR) ? KVQ_SCORE_F16_PAIR_TABLES : 0, stream);   // (tables built by the caller: kvq_score_k_tables / the extract, which know the fp16 image only)
  record(1, st);
  if (rc) return rc;
  return step_tail(ly, sinks, v_sink, sink_probs, out, fuse_softmax, scores, probs, parts, p.n_parts, ws, p, L, n_out, n_sink, inv, st);
}

int kvq_attend_step(const kvq_layer *ly, int64_t L, const void *q, int acts_are_half, const kvq_sinks *sinks,
                    const uint16_t *v_sink, uint16_t *sink_probs, float *out, int fuse_softmax, void *workspace,
                    size_t workspace_bytes, void *stream) {
  return attend_step(ly, L, q, acts_are_half, sinks, v_sink, sink_probs, out, fuse_softmax, workspace, workspace_bytes, stream,
                     false);
}

/* One decode token through one HEAD SHARD of a layer (kvquant_amd.cache.HeadShard; SURVEY 8e "by head"), one call:
 * the WHOLE token is appended into column 0 of the full-width staging cache `full` (kvq_append_kv_fused: selection and
 * codes bit-identical on every rank), heads [h0, h0 + shard->H) of that column are extracted into column `col` of the
 * shard's cache (kvq_extract_heads) and the shard's heads attend over its col + 1 tokens (kvq_attend_step).  q: the
 * shard's heads of the RoPE'd query [shard->H][128]; k, v: the whole token [full->H * hd]; sinks / v_sink / sink_probs: the
 * shard's heads of the fp16 sink caches.  out f32 [shard->H][hd]. */
int kvq_head_shard_step(const kvq_layer *full, const kvq_layer *shard, int h0, int64_t col, const void *q, const void *k,
                        const void *v, int acts_are_half, const kvq_sinks *sinks, const uint16_t *v_sink,
                        uint16_t *sink_probs, float *out, int fuse_softmax, void *workspace, size_t workspace_bytes,
                        void *stream) {
  if (!full || !shard || h0 < 0 || h0 + shard->H > full->H || full->bits != shard->bits || full->hd != shard->hd ||
      full->thr_k != shard->thr_k || col < 0 || col >= shard->max_len)
    return KVQ_EINVAL;
  if (!full->koutliers || !full->voutliers || !shard->koutliers || !shard->voutliers) return KVQ_EINVAL;   // (reference outlier format)
  // (every argument is checked before the first launch: a refused call leaves the staging column and the shard untouched)
  if (!q || !k || !v || !out) return KVQ_EINVAL;
  if (!workspace || reinterpret_cast<uintptr_t>(workspace) % 256 ||
      workspace_bytes < kvq_decode_step_workspace_bytes(shard->bits, shard->H, shard->hd, col + 1))
    return KVQ_EWORKSPACE;
  if (sinks != nullptr && sinks->n_sink > 0 && (!sinks->k_sink || !sinks->sink_scores)) return KVQ_EINVAL;
  int rc = kvq_append_kv_fused(full, 0, k, v, acts_are_half, stream);
  if (rc) return rc;
  // the extract of the staged column and the shard's query tables (+ fp16 sink scores) as ONE launch
  const kvq_vopts *vn = full->vnorm;
  const kvq_vopts *sn = shard->vnorm;
  ExtractArgs a;
  fill_extract_args(a, full->bits, full->hd, h0, shard->H, 2 * full->thr_k, full->kmat, full->vmat, full->max_len, 0,
                    full->koutliers, full->kidx, full->voutliers, full->vidx, full->vlut_rows, shard->kmat, shard->vmat,
                    shard->max_len, col, shard->koutliers, shard->kidx, shard->koutliers_t, shard->kidx_t, shard->voutliers,
                    shard->vidx, shard->vlut_rows, (vn && sn) ? vn->lut_rows2 : nullptr, (vn && sn) ? sn->lut_rows2 : nullptr, 1);
  const int bits = shard->bits, Hs = shard->H;
  unsigned char *tab = reinterpret_cast<unsigned char *>(workspace);
  const size_t qoff = bits == 4 ? ktab_q_offset<4>(1, Hs) : (bits == 3 ? ktab_q_offset<3>(1, Hs) : ktab_q_offset<2>(1, Hs));
  a.n_tab = Hs;
  a.lut = shard->klut_score ? shard->klut_score : shard->klut;
  a.q = q;
  a.q_is_half = acts_are_half;
  a.tab = tab;
  a.q32 = reinterpret_cast<float *>(tab + qoff);
  a.pair_tab = bits == 3 ? tab + ktab_pair_offset<3>(1, Hs) : nullptr;
  if (sinks != nullptr && sinks->n_sink > 0) {
    a.k_sink = reinterpret_cast<const __half *>(sinks->k_sink);
    a.sink_scores = reinterpret_cast<__half *>(sinks->sink_scores);
    a.n_sink = sinks->n_sink;
    a.sink_inv = sinks->inv_sqrt_hd;
  }
  rc = launch_extract_fused(a, (hipStream_t)stream);
  if (rc) return rc;
  return attend_step(shard, col + 1, q, acts_are_half, sinks, v_sink, sink_probs, out, fuse_softmax, workspace, workspace_bytes,
                     stream, true);
}

int kvq_decode_steps(int n_layers, const kvq_layer *layers, int64_t col, const void *const *q, const void *const *k,
                     const void *const *v, int acts_are_half, float *const *out, int fuse_softmax, void *workspace,
                     size_t workspace_bytes, void *stream) {
  if (n_layers <= 0 || !layers || !q || !k || !v || !out) return KVQ_EINVAL;
  for (int i = 0; i < n_layers; i++) {
    int rc = kvq_decode_step(&layers[i], col, col, q[i], k[i], v[i], acts_are_half, nullptr, nullptr, nullptr, out[i],
                             fuse_softmax, workspace, workspace_bytes, stream);
    if (rc) return rc;
  }
  return KVQ_OK;
}

}  // extern "C"
