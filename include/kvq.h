/*
 * kvq.h -- C ABI of the MI355X-native KVQuant hot path (libkvq.so).
 *
 * This is the drop-in boundary: plain device pointers, sizes, scalars and a
 * hipStream_t (passed as void*).  No torch types.  Every entry point replaces
 * one family of functions of the reference's pybind11 module `quant_cuda`
 * (KCPP = /root/reference/deployment/kvquant/quant_cuda.cpp, kernels in
 * KCU = /root/reference/deployment/kvquant/quant_cuda_kernel.cu), with
 * `bits` (2, 3 or 4) as a parameter instead of three copies.  INTEGRATION.md
 * shows the binding a reference maintainer would add.
 *
 * Conventions
 *   - all pointers are DEVICE pointers on the current HIP device unless the
 *     name says host; buffers are contiguous and owned by the caller;
 *   - cache `mat`: int32 [H][hd/32*bits][max_len], token index contiguous,
 *     bit layout of KCU:1240-1244 (4b), 1395-1424 (3b), 2712-2716 (2b);
 *   - K look-up table `lut`: float [H][hd][2^bits] (per channel);
 *   - V look-up table `lut_rows`: float [max_len][2^bits] (per token);
 *   - sparse buffers: `outliers` float [max_len][n_out], `outlier_idx` int32
 *     [max_len][n_out], global channel indices ascending within a row;
 *   - head_dim must be 128 for the score/mix kernels (the reference kernels
 *     hard-code BLOCKWIDTH = 128 = head_dim, KCU:43, 3101), any multiple of 32
 *     for the append/pack kernels;
 *   - every function is asynchronous on `stream` and returns 0 on success or
 *     a negative KVQ_E* code (kvq_strerror); nothing is allocated or freed.
 *   - the reference asserts on shapes and otherwise faults; here violated
 *     preconditions are reported as KVQ_EINVAL before any launch.
 */
#ifndef KVQ_H
#define KVQ_H

#include <stddef.h>
#include <stdint.h>

#if defined(__GNUC__)
#define KVQ_API __attribute__((visibility("default")))
#else
#define KVQ_API
#endif

#ifdef __cplusplus
extern "C" {
#endif

#define KVQ_OK 0
#define KVQ_EINVAL (-1)    /* bad argument (null pointer, unsupported bits / head_dim / size) */
#define KVQ_ELAUNCH (-2)   /* hipLaunch / runtime error, see kvq_last_hip_error() */
#define KVQ_EWORKSPACE (-3) /* workspace missing or too small */

/* major * 100 + minor.  History of the ABI:
 *   200  round 2: one-call decode step (kvq_decode_step), prepared score tables
 *   300  round 4: (a) every entry point that takes a token-contiguous outlier mirror (`outliers_t`, `outlier_idx_t`) or
 *        outlier rows (`outliers`, `outlier_idx`) reads a NULL value array next to a non-NULL index array as the COMPACT
 *        format -- index words are fp16 residual << 16 | channel -- where 200 fell back to the row layout / rejected it: a
 *        caller that used to pass reference-format int32 indices without values must now pass both arrays;
 *        (b) kvq_vopts NULL now means the reference's behaviour including its tie quirk (reference_tie_quirk = 1);
 *        (c) new entries: kvq_mix_v_softmax_affine, kvq_mix_v_affine_*, kvq_fused_attend*, kvq_decode_step
 *        fuse_softmax modes 2 and 3, kvq_extract_heads, kvq_rope_q_f16.
 *   400  round 5: (a) struct kvq_layer grew (`flags` at the end): a caller compiled against 300 hands over a shorter struct;
 *        (b) the constant-table p.V kernel left the library (kvq_mix_v_softmax_affine, kvq_mix_v_affine_*: measured slower
 *        than the per-row kernel everywhere, tools/experiments/kvq_mix_va.hip keeps the source); kvq_decode_step
 *        fuse_softmax 1 and 2 are the same route now;  (c) kvq_score_k_workspace_bytes is larger at 3 bit (the fp16
 *        pair-sum tables live behind the fp32 ones);  (d) kvq_score_k_tables takes the shard's fp16 sink tokens, kvq_softmax_stats their
 *        scores, kvq_extract_heads the Q-Norm rows;  (e) new entries: kvq_score_k_prepared_softmax_ex, kvq_decode_step_route,
 *        kvq_append_kv_fused, kvq_attend_step, kvq_head_shard_step.
 *   401  round 6: no entry point added, removed or re-typed.  Behaviour: kvq_mix_v / kvq_mix_v_softmax (and kvq_decode_step
 *        through them) run the one-workgroup-per-CU kernel (kvq_mix_v_wide.hip) for q_len = 1 from 6144 cached tokens on --
 *        same arguments, same workspace size, the outlier sums are still exact (64-bit fixed point) and order independent;
 *        kvq_head_shard_step validates every argument before its first launch.
 *   402  round 6: new entries kvq_score_k_mirror, kvq_outlier_mirror_rows (the mirror variant of q.K^T for callers that
 *        keep the reference's outlier rows); new flag values KVQ_SCORE_F32_PAIR_TABLES / KVQ_LAYER_SCORE_F32_PAIR;
 *        kvq_score_k_workspace_bytes is 16 KB per head larger at 3 bit.
 *        A binding checks `kvq_version() / 100 == KVQ_ABI_MAJOR`. */
#define KVQ_ABI_MAJOR 4
KVQ_API int kvq_version(void);
KVQ_API const char *kvq_strerror(int code);
/* last hipError_t seen by a failing call on this thread (0 = none) */
KVQ_API int kvq_last_hip_error(void);

/* RoPE of the decode query in fp16, the glue of the patched attention (ML:1851-1853: query_states * cos +
 * rotate_half(query_states) * sin on fp16 tensors) as ONE launch with the same roundings -- both products and the sum
 * rounded to fp16 --, bit-identical to the three torch kernels.  q, out: fp16 [H][128]; cosv, sinv: fp16 [128], the
 * position's row of the reference's rotary tables. */
KVQ_API int kvq_rope_q_f16(const uint16_t *q, const uint16_t *cosv, const uint16_t *sinv, uint16_t *out, int H, int hd,
                   void *stream);

/* The 64 RoPE frequencies theta_j = powf(rope_theta, -2j/128) exactly as the score
 * kernels evaluate them -- on the device, like the reference (KCU:3081, 508, 584) --
 * written to out[0..63] (device).  For callers and checkers that need the same
 * values (the device powf differs from the correctly rounded one by 1 ulp for some j,
 * which is 6e-8 * position radians in the angle). */
KVQ_API int kvq_rope_freqs(float rope_theta, float *out, void *stream);

/* ---- append one token -------------------------------------------------- */

/* vecquant{2,3,4}appendvecK (KCPP:5-33; KCU:1167-1245, 1322-1425, 1528-1607):
 * code[c] = argmin_v |lut[c][v] - x[c]| (first minimum wins), packed into
 * column `col`, which must be zero. */
KVQ_API int kvq_append_k(int bits, int32_t *mat, const float *lut, const float *x,
                 int H, int hd, int64_t max_len, int64_t col, void *stream);

/* vecquant{2,3,4}appendvecV (KCPP:99-124; KCU:1248-1320, 1427-1526,
 * 1609-1682): as above with the per-token row lut_rows[col]. */
KVQ_API int kvq_append_v(int bits, int32_t *mat, const float *lut_rows, const float *x,
                 int H, int hd, int64_t max_len, int64_t col, void *stream);

/* vecquant{2,3,4}appendvecKsparse (KCPP:35-43, 58-63, 78-83; KCU:1684-1781,
 * 2104-2227, 2620-2717): append_k plus
 * rescaled[c] = (x[c] - (hi[c]+lo[c])/2) / ((hi[c]-lo[c])/2). */
KVQ_API int kvq_append_k_sparse(int bits, int32_t *mat, const float *lut, const float *x,
                        float *rescaled, const float *lo, const float *hi,
                        int H, int hd, int64_t max_len, int64_t col, void *stream);

/* vecquant{2,3,4}appendvecVsparse (KCPP:126-134, 150-155, 170-175;
 * KCU:2011-2102, 2369-2486, 2947-3038): code = (x<lo || x>hi) ? 7/3/1 :
 * argmin over lut_rows[col].  (The reference's `zeropoint` argument is unused
 * by its kernel and therefore not part of this ABI.) */
KVQ_API int kvq_append_v_sparse(int bits, int32_t *mat, const float *lut_rows,
                        const float *x, float lo, float hi, int H, int hd,
                        int64_t max_len, int64_t col, void *stream);

/* ---- pack S prompt tokens (prefill) ------------------------------------ */

/* vecquant{2,3,4}appendvecKsparseParallel (KCPP:45-53, 65-73, 85-93;
 * KCU:1783-1898, 2229-2366, 2719-2834): x and rescaled are [H][hd][S]
 * (token contiguous); writes columns col0 .. col0+S-1 (reference: col0 = 0). */
KVQ_API int kvq_pack_k_sparse_parallel(int bits, int32_t *mat, const float *lut,
                               const float *x, float *rescaled, const float *lo,
                               const float *hi, int H, int hd, int64_t S,
                               int64_t max_len, int64_t col0, void *stream);

/* vecquant{2,3,4}appendvecVsparseParallel (KCPP:136-144, 157-165, 177-185;
 * KCU:1900-2009, 2488-2618, 2836-2945): per-token rows lut_rows[col0+t] and
 * per-token thresholds lo[t], hi[t] (t < S). */
KVQ_API int kvq_pack_v_sparse_parallel(int bits, int32_t *mat, const float *lut_rows,
                               const float *x, const float *lo, const float *hi,
                               int H, int hd, int64_t S, int64_t max_len,
                               int64_t col0, void *stream);

/* ---- decode matvecs ------------------------------------------------------ */

/* vecquant{b}matmul_nuq_perchannel_transposed_rope_mha_batched_fused_opt and
 * ..._opt2 (KCPP:188-212 ...; KCU:3040-3209, 3437-3622, 473-521):
 *   mul[b][h][t] (+)= sum_k Khat[h][k][t] * (cos(a)*q[b][h][k]
 *                       + sgn_k*sin(a)*q[b][h][(k+64)%128]),
 *   a = fl32(theta^(-2(k%64)/128) * (t+pos_offset)),
 * Khat = LUT value + the token's sparse residuals when outliers != NULL (the
 * sparse part only for b = 0, as the reference).  q: float [q_len][H][128];
 * mul: float [q_len][H][L].  accumulate != 0: add into mul (the reference's
 * pre-zeroed contract); 0: overwrite.  Needs a 16-byte aligned device workspace of
 * kvq_score_k_workspace_bytes(...) bytes (the query-premultiplied codebook
 * images, H * 2^bits KiB).  outlier_idx of a token in ascending order (as the
 * reference's glue stores it) takes the fast path; any other order is still
 * summed correctly, one entry at a time. */
KVQ_API size_t kvq_score_k_workspace_bytes(int bits, int q_len, int H);
KVQ_API int kvq_score_k(int bits, const float *q, const int32_t *mat, float *mul,
                const float *lut, int q_len, int H, int hd, int64_t L,
                int64_t max_len, float rope_theta, int pos_offset,
                const float *outliers, const int32_t *outlier_idx, int n_out,
                int accumulate, void *workspace, size_t workspace_bytes,
                void *stream);

/* vecquant{b}matmul_nuq_perchannel_transposed_mha_batched_fused_opt and
 * ..._opt2 (KCPP:214-238 ...; KCU:3211-3433, 3491-3538, 3625-3690, 437-470):
 *   mul[b][h][c] (+)= sum_t Vhat[h][c][t] * p[b][h][t]
 * p: float [q_len][H][L]; mul: float [q_len][H][128].  Needs a device
 * workspace of kvq_mix_v_workspace_bytes(...) bytes (partial sums; no atomics,
 * deterministic). */
KVQ_API size_t kvq_mix_v_workspace_bytes(int bits, int q_len, int H, int hd, int64_t L);
KVQ_API int kvq_mix_v(int bits, const float *p, const int32_t *mat, float *mul,
              const float *lut_rows, int q_len, int H, int hd, int64_t L,
              int64_t max_len, const float *outliers, const int32_t *outlier_idx,
              int n_out, int accumulate, void *workspace, size_t workspace_bytes,
              void *stream);

/* ---- GPU-resident decode append (replaces the reference's CPU topk round trips) -- */

/* One launch = vecquant{b}appendvecKsparse + the host glue of
 * QuantK.forward_fused_sparse (modeling_llama.py:706-751): top-thr_k largest and
 * smallest rescaled values, residual = x - lut_off[c][n-1] / x - lut_off[c][0],
 * zeroed where the rescaled value is inside [-1, 1], in ascending channel order,
 * stored to row `col` of outliers / outlier_idx (width 2*thr_k).  lut_off = lut,
 * or the Q-Norm table.  Ties at the selection boundary: lowest channel first
 * (torch.topk leaves this unspecified).  H*hd <= 8192. */
KVQ_API int kvq_append_k_fused(int bits, int32_t *mat, const float *lut,
                       const float *lut_off, const float *x, const float *lo,
                       const float *hi, float *outliers, int32_t *outlier_idx,
                       int thr_k, int H, int hd, int64_t max_len, int64_t col,
                       float *outliers_t, int32_t *outlier_idx_t, void *stream);

/* Options of the fused V appends (host struct).  NULL = the reference's behaviour: no Q-Norm, reference_tie_quirk = 1
 * (codes bit-identical to vecquant{b}appendvecVsparse on every token).
 *  - Q-Norm (modeling_llama.py:1116-1118, 1153-1156, 1237-1240, 1369-1375): lut_rows2 != NULL makes the kernel write
 *    the second per-token codebook row lut_rows2[col] = (lut_sorted*normscale + normoffset)*sf + off next to
 *    lut_rows[col]; zp_from_rows2 != 0 makes the sparse residuals refer to lut_rows2[col][zero code] instead of
 *    lut_rows[col][zero code] (the reference does so at 2 bit in decode and at every width in its prefill glue).
 *  - reference_tie_quirk: the reference's V append clips an element to the zero-point code only when it lies
 *    STRICTLY outside [22nd smallest, 22nd largest] (KCU:2084), while its glue stores the 21 largest / smallest as
 *    sparse residuals (modeling_llama.py:1093-1096).  When the 21st and 22nd largest values are equal -- one token in
 *    ten for fp16 activations -- that element keeps its nearest code AND gets a residual: it is counted twice
 *    (SURVEY App. A.5).  1 (what NULL options mean): replicate that, as kvquant_amd.cache.QuantV does by default.
 *    0: an element is clipped iff it is stored sparse, which is what the reference's simulated path computes
 *    (simquant_module_quantizer.py:95-108, 347-350). */
typedef struct kvq_vopts {
  float *lut_rows2;      /* device, float [max_len][2^bits], or NULL */
  float normscale;
  float normoffset;
  int zp_from_rows2;
  int reference_tie_quirk;
} kvq_vopts;

/* One launch = the V top-(thr_k+1) selection of modeling_llama.py:1537-1545, the
 * per-token codebook row lut_sorted*sf+off written to lut_rows[col] (1086-1114),
 * vecquant{b}appendvecVsparse, and the sparse row (1168-1176).
 * lut_sorted: float [2^bits] ascending.  opts: host pointer or NULL. */
KVQ_API int kvq_append_v_fused(int bits, int32_t *mat, float *lut_rows,
                       const float *lut_sorted, const float *x, float *outliers,
                       int32_t *outlier_idx, int thr_k, int H, int hd,
                       int64_t max_len, int64_t col, const kvq_vopts *norm,
                       void *stream);

/* Prefill forms of the two fused appends: S prompt tokens in ONE launch (one workgroup
 * per token), replacing vecquant{b}appendvec{K,V}sparseParallel + the torch.topk /
 * gather / sort glue of QuantK/QuantV.parallel_pack (modeling_llama.py:879-972,
 * 1294-1382).  x: the prompt channel-major, float [H*hd][S] (KCPP:48-53); columns
 * col0 .. col0+S-1 of the cache (must be zero) and rows col0.. of the outlier buffers
 * (and of the optional K mirror) are written. */
KVQ_API int kvq_pack_k_fused(int bits, int32_t *mat, const float *lut, const float *lut_off,
                     const float *x, const float *lo, const float *hi, float *outliers,
                     int32_t *outlier_idx, int thr_k, int H, int hd, int64_t max_len,
                     int64_t col0, int64_t S, float *outliers_t, int32_t *outlier_idx_t,
                     void *stream);
KVQ_API int kvq_pack_v_fused(int bits, int32_t *mat, float *lut_rows, const float *lut_sorted,
                     const float *x, float *outliers, int32_t *outlier_idx, int thr_k,
                     int H, int hd, int64_t max_len, int64_t col0, int64_t S,
                     const kvq_vopts *norm, void *stream);

/* modeling_llama.py:873-874, 1950-1962, 1972-1977 in two launches: scores fp32
 * [H][L] (raw q.K^T) -> half -> * inv_sqrt_hd (fp16) -> softmax in fp32 over
 * [sink_scores | scores] -> fp16.  probs: the fp16 values widened to fp32, [H][L]
 * (the layout kvq_mix_v reads); sink_scores / sink_probs: fp16 [H][n_sink]
 * (already scaled; NULL when n_sink = 0). */
KVQ_API size_t kvq_softmax_workspace_bytes(int H, int64_t L);
KVQ_API int kvq_softmax_scale(const float *scores, const uint16_t *sink_scores,
                      float *probs, uint16_t *sink_probs, int H, int64_t L,
                      int n_sink, float inv_sqrt_hd, void *workspace,
                      size_t workspace_bytes, void *stream);

/* ---- one-launch decode prologue ------------------------------------------------ */

/* K fused append + V fused append + the query-premultiplied K codebook images for
 * kvq_score_k_prepared, as three roles of ONE launch (the jobs are independent and
 * latency-bound).  k, v: [H*hd], q: [H][128] (already RoPE'd), all fp32
 * (acts_are_half = 0) or all fp16 (1, the model's activation dtype: saves the
 * reference's three .float() launches).  score_workspace as for kvq_score_k.
 * klut_ends (optional): float [H*hd][2] = (klut_off[c][0], klut_off[c][2^bits-1]),
 * the two codebook values the outlier residuals refer to, contiguous -- a constant
 * of the layer that saves the selection workgroup a dependent 256 KB-strided read.
 * klut_score (optional, NULL = klut): the table the score images are built from --
 * the K Q-Norm table at 2 bit (modeling_llama.py:811-815).  vnorm: options of the V append or NULL. */
/* sinks (optional, NULL = none): the fp16 attention-sink tokens of modeling_llama.py:1464-1466.  The head's table
 * workgroup also writes the scaled sink scores half(half(q . k_sink) * inv_sqrt_hd) (ML:1950-1962) that
 * kvq_softmax_finish / kvq_mix_v_softmax take, replacing a torch.matmul and a division per layer. */
typedef struct kvq_sinks {
  const uint16_t *k_sink;   /* fp16 [H][128][n_sink], post-RoPE keys */
  uint16_t *sink_scores;    /* fp16 [H][n_sink] (out) */
  int n_sink;
  float inv_sqrt_hd;
} kvq_sinks;

KVQ_API int kvq_decode_prologue(int bits, int32_t *kmat, const float *klut,
                        const float *klut_off, const void *k, const float *lo,
                        const float *hi, float *koutliers, int32_t *kidx, int64_t kcol,
                        int32_t *vmat, float *vlut_rows, const float *vlut_sorted,
                        const void *v, float *voutliers, int32_t *vidx, int64_t vcol,
                        const void *q, int acts_are_half, int thr_k, int H, int hd,
                        int64_t max_len, float *koutliers_t, int32_t *kidx_t,
                        const float *klut_ends, const float *klut_score,
                        const kvq_vopts *vnorm, const kvq_sinks *sinks, void *score_workspace,
                        size_t score_workspace_bytes, void *stream);
/* kvq_score_k for q_len = 1 with the tables (and the fp32 query copy) already in
 * `workspace` (written by kvq_decode_prologue on the same stream). */
KVQ_API int kvq_score_k_prepared(int bits, const int32_t *mat, float *mul, const float *lut,
                         int H, int hd, int64_t L, int64_t max_len, float rope_theta,
                         int pos_offset, const float *outliers, const int32_t *outlier_idx,
                         int n_out, int accumulate, void *workspace,
                         size_t workspace_bytes, void *stream);

/* First softmax pass fused into the sparse score kernel (it holds each 256-token x H
 * tile of scores in LDS anyway): kvq_score_k_prepared with accumulate = 0 that also
 * writes, per head and tile, (max, sum of exp) of the scaled scores
 * half(half(score) * inv_sqrt_hd) to softmax_parts[H][n_parts][2];
 * n_parts = kvq_score_k_softmax_parts(bits, L, 1) (0 = shape not supported: use
 * kvq_softmax_scale).  kvq_softmax_finish is the second pass of kvq_softmax_scale
 * (modeling_llama.py:1976) on such partials; the fp16 sink scores are merged there.
 * outliers_t / outlier_idx_t (optional, replace outliers / outlier_idx): the
 * token-contiguous mirror [n_out][max_len] of the K outlier rows that
 * kvq_decode_prologue / kvq_append_k_fused maintain when given one -- entries of a
 * token in the same (ascending channel) order.  With it a lane owns its token's
 * entries: coalesced loads, no segmented scan (the reference layout costs +23 us at
 * 128K for exactly that). */
KVQ_API int kvq_score_k_softmax_parts(int bits, int64_t L, int sparse);
/* Host logic of the score launches (no device work; exported for tests and capacity planning): into how many head
 * groups the launch cuts each full token tile, given `tiles` full tiles, q_len query rows, the kernel's limit on
 * heads per workgroup and `slots` = workgroups the GPU holds at once (512 on MI355X for the 256-token kernel).  The
 * reference launches one block per (128 tokens, head) whatever the size (KCU:3437-3488); here the grid is planned with
 * a makespan model (kvq_score_k.hip, pick_groups).  0 = invalid arguments. */
/* round 6 (ABI 402).  kvq_score_k (above) over the token-contiguous outlier MIRROR f32 / i32 [n_out][max_len] instead of
 * the reference's rows [max_len][n_out]: the decode kernel's fast variant (a lane owns its token's entries; 82 - 87 us
 * against 117 us through the rows at 128K nuq4) behind the legacy call's semantics -- tables built from q, `mul`
 * accumulated (accumulate = 1, what quant_cuda's _opt2 functions do, KCPP:205-212) or overwritten.  q [H][128] f32, q_len = 1.
 * kvq_outlier_mirror_rows fills columns [t0, t1) of a mirror from rows [t0, t1) of the reference layout (KCU:473-521 reads
 * the rows; modeling_llama.py:742-751 writes them): a caller that keeps the rows as the source of truth transposes what
 * it appended since its last call.  kvquant_amd.quant_cuda does exactly that behind the module swap. */
KVQ_API int kvq_score_k_mirror(int bits, const float *q, const int32_t *mat, float *mul, const float *lut, int H,
                               int hd, int64_t L, int64_t max_len, float rope_theta, int pos_offset,
                               const float *outliers_t, const int32_t *outlier_idx_t, int n_out, int accumulate,
                               void *workspace, size_t workspace_bytes, void *stream);
KVQ_API int kvq_outlier_mirror_rows(const float *outliers, const int32_t *outlier_idx, float *outliers_t,
                                    int32_t *outlier_idx_t, int n_out, int64_t max_len, int64_t t0, int64_t t1,
                                    void *stream);
KVQ_API int kvq_score_k_head_groups(int H, int64_t tiles, int q_len, int max_heads_per_group, int slots);
KVQ_API int kvq_score_k_prepared_softmax(int bits, const int32_t *mat, float *mul,
                         const float *lut, int H, int hd, int64_t L, int64_t max_len,
                         float rope_theta, int pos_offset, const float *outliers,
                         const int32_t *outlier_idx, int n_out, const float *outliers_t,
                         const int32_t *outlier_idx_t, void *workspace,
                         size_t workspace_bytes, float inv_sqrt_hd, float *softmax_parts,
                         int n_parts, void *stream);
/* As above with `flags`:
 * KVQ_SCORE_F16_PAIR_TABLES (3 bit, token-contiguous mirror only; ignored elsewhere): the score kernel reads fp16
 * PAIR-SUM tables -- the two channels of a rotation pair (k, k + 64) share their (cos, sin), so one table entry indexed by
 * both codes holds the pre-added, query-premultiplied pair (64 entries x 4 B per pair, 16 KB per head; written by
 * kvq_decode_prologue / kvq_score_k_tables next to the fp32 tables) and ONE LDS look-up + one v_dot2_f32_f16 replace two
 * look-ups and two packed FMAs.  The entries and the (cos, sin) are rounded to fp16, the sums accumulate in fp32: scores
 * agree with the fp32 tables to ~1e-4 of a row's largest score (contract: 1e-3; the reference rounds the scores
 * themselves to fp16, modeling_llama.py:873).  128K nuq3: q.K^T 80.4 us (exact fp32 tables) -> 70.6 us
 * (DESIGN.md 3.6b); scores within 4e-4 of the reference's, attention outputs within 4e-3 (1.4e-3 .. 1.8e-3 measured). */
/* KVQ_SCORE_F32_PAIR_TABLES (round 6; 3 bit, token-contiguous mirror, L >= 16384; ignored elsewhere): the same pair sums in
 * FP32 (64 entries x 8 B per pair, 32 KB per head, in the place of the fp16 image: a workspace holds one of the two) --
 * exact arithmetic, one ds_read_b64 + one v_pk_fma_f32 per two codes, one 1024-lane workgroup of 512 tokens per CU.
 * Opt-in (kvquant_amd: KVQ_SCORE_F32_PAIR=1): equal to the per-channel tables at 128K, 3 % faster at 1M, slower at 32K
 * (profiles/r06_p_pair32_ab.txt). */
#define KVQ_SCORE_F16_PAIR_TABLES 1
#define KVQ_SCORE_F32_PAIR_TABLES 2
KVQ_API int kvq_score_k_prepared_softmax_ex(int bits, const int32_t *mat, float *mul,
                         const float *lut, int H, int hd, int64_t L, int64_t max_len,
                         float rope_theta, int pos_offset, const float *outliers,
                         const int32_t *outlier_idx, int n_out, const float *outliers_t,
                         const int32_t *outlier_idx_t, void *workspace,
                         size_t workspace_bytes, float inv_sqrt_hd, float *softmax_parts,
                         int n_parts, int flags, void *stream);
/* v_sink (optional, with sink_out): fp16 [H][n_sink][128] values of the sink tokens; the workgroup that writes
 * a head's sink probabilities also writes the sink tokens' share of the attention output,
 * sink_out[h][c] = float(half(sum_i sink_probs[h][i] * v_sink[h][i][c])) (torch.matmul in fp16, ML:1987-1995), so
 * that kvq_mix_v with accumulate = 1 completes the layer's output without further launches. */
KVQ_API int kvq_softmax_finish(const float *scores, const uint16_t *sink_scores,
                       const float *parts, int n_parts, float *probs,
                       uint16_t *sink_probs, int H, int64_t L, int n_sink,
                       float inv_sqrt_hd, const uint16_t *v_sink, float *sink_out,
                       void *stream);

/* kvq_softmax_finish + kvq_mix_v (q_len = 1) in ONE streaming pass: the p.V kernel
 * merges the (max, sum) partials itself and turns raw scores into the fp16-rounded
 * probabilities in LDS, a chunk ahead of its look-up loop -- the [H][L] probabilities
 * of modeling_llama.py:1976 never exist in memory (one launch, 13 us and 8 bytes per
 * token and head less at 128K).  Same arithmetic per element as kvq_softmax_finish;
 * the row normaliser is merged in a different order (last-bit differences).  Shapes
 * the streaming kernel does not take (max_len % 4 != 0, H > 128) run the two passes
 * separately through `probs` (float [H][L] scratch; may be NULL otherwise).
 * v_sink (optional, needs accumulate = 0): as in kvq_softmax_finish, `mul` then also holds the sink tokens' share.
 * workspace: kvq_mix_v_workspace_bytes(bits, 1, H, hd, L). */
KVQ_API int kvq_mix_v_softmax(int bits, const float *scores, const float *parts, int n_parts,
                      float inv_sqrt_hd, const uint16_t *sink_scores, uint16_t *sink_probs,
                      int n_sink, const uint16_t *v_sink, float *probs, const int32_t *mat, float *mul,
                      const float *lut_rows, int H, int hd, int64_t L, int64_t max_len,
                      const float *outliers, const int32_t *outlier_idx, int n_out,
                      int accumulate, void *workspace, size_t workspace_bytes, void *stream);

/* The attention of one decode token over a layer's compressed cache as ONE streaming kernel + a small merge
 * (kvq_fused_decode.hip): every workgroup takes a 256-token tile through q.K^T (+ RoPE, + K outlier entries: the tile
 * body of kvq_score_k_prepared_softmax), a tile-local softmax and p.V (+ V outlier entries) of the same tokens; the merge
 * combines the tiles' (max, sum, partial output) records and the fp16 sink tokens exactly ("flash-decoding" over the
 * token axis).  Replaces kvq_score_k_prepared_softmax + kvq_mix_v_softmax (KCU:3040-3209, 473-521, 3211-3433, 437-470 and
 * the softmax between them, modeling_llama.py:1948-1995): the scores never exist in memory.  A probability is rounded
 * to fp16 relative to its tile's maximum instead of the row's normaliser (same 2^-11 relative rounding; tolerance of the
 * attention output against the reference pipeline: 1e-3).  score_workspace: the query-premultiplied tables + fp32 query
 * written by kvq_decode_prologue / kvq_score_k_tables.  Reference outlier formats with the K mirror, n_out <= 48,
 * H <= 32, max_len % 4 == 0 (kvq_fused_attend_supported); out: float [H][hd]. */
KVQ_API int kvq_fused_attend_supported(int bits, int H, int hd, int64_t L, int64_t max_len, int n_out);
KVQ_API size_t kvq_fused_attend_workspace_bytes(int bits, int H, int hd, int64_t L);
KVQ_API int kvq_fused_attend(int bits, const int32_t *kmat, const float *klut, const void *score_workspace,
                     const int32_t *vmat, const float *vlut_rows, int H, int hd, int64_t L, int64_t max_len,
                     float rope_theta, int pos_offset, const float *koutliers_t, const int32_t *kidx_t,
                     const float *voutliers, const int32_t *vidx, int n_out, float inv_sqrt_hd,
                     const uint16_t *sink_scores, uint16_t *sink_probs, int n_sink, const uint16_t *v_sink,
                     float *out, void *workspace, size_t workspace_bytes, void *stream);

/* ---- one decode token through one layer, one call -------------------------------- */

/* The buffers of one layer's compressed KV cache: what the reference keeps as attributes of QuantK / QuantV
 * (modeling_llama.py:352-1385) plus the token-contiguous K outlier mirror.  Dense-and-Sparse caches
 * (include_sparse) only; thr_k = outliers per side and token (21 at 1 % of 4096 channels). */
typedef struct kvq_layer {
  int bits, H, hd, thr_k;
  int64_t max_len;
  float rope_theta;
  int pos_offset;             /* first_few_fp16: position of cached token 0 */
  /* K (QuantK) */
  int32_t *kmat;              /* kcache       int32 [H][hd/32*bits][max_len] */
  const float *klut;          /* lookup_table f32 [H][hd][2^bits] (pack) */
  const float *klut_off;      /* table the outlier residuals refer to (lookup_table2 with Q-Norm, else klut) */
  const float *klo, *khi;     /* outlier_threshold_lower / _upper f32 [H*hd] */
  float *koutliers;           /* [max_len][2*thr_k] */
  int32_t *kidx;
  float *koutliers_t;         /* mirror [2*thr_k][max_len] */
  int32_t *kidx_t;
  const float *klut_ends;     /* optional [H*hd][2] */
  const float *klut_score;    /* optional: table the scores dequantise with (Q-Norm at 2 bit), NULL = klut */
  /* V (QuantV) */
  int32_t *vmat;              /* vcache */
  float *vlut_rows;           /* lookup_table f32 [max_len][2^bits] */
  const float *vlut_sorted;   /* lut f32 [2^bits], ascending */
  float *voutliers;           /* [max_len][2*thr_k] */
  int32_t *vidx;
  const kvq_vopts *vnorm;     /* optional */
  const float *v_mix_rows;    /* optional: table p.V dequantises with (lookup_table2 at 2 bit with Q-Norm), NULL = vlut_rows */
  int flags;                  /* KVQ_LAYER_*: options of the decode step */
} kvq_layer;
/* kvq_layer.flags: 3-bit caches score through the fp16 pair-sum tables (KVQ_SCORE_F16_PAIR_TABLES above) */
#define KVQ_LAYER_SCORE_F16_PAIR 1
/* ... through the exact fp32 pair-sum tables (KVQ_SCORE_F32_PAIR_TABLES above; wins over the fp16 flag) */
#define KVQ_LAYER_SCORE_F32_PAIR 2

/* One decode token through one layer: K / V fused appends at column kcol (= vcol) + query tables, q.K^T with RoPE +
 * outliers + first softmax pass, softmax, p.V + outliers, slab reduce -- the launches of kvq_decode_prologue,
 * kvq_score_k_prepared_softmax, kvq_softmax_finish (or, fuse_softmax != 0: inside kvq_mix_v_softmax) and kvq_mix_v
 * issued back to back on `stream` from one call.  Replaces the decode branch of the patched LlamaAttention.forward
 * around QuantK / QuantV.forward_fused_sparse (modeling_llama.py:1930-2000).  q [H][128] (RoPE'd), k, v [H*hd]:
 * all fp32 or all fp16.  sinks / v_sink / sink_probs: the fp16 attention-sink tokens (as kvq_decode_prologue /
 * kvq_softmax_finish), or NULL.  out f32 [H][hd]: the complete attention output.  Scores, probabilities, partials and
 * slabs live in `workspace` (kvq_decode_step_workspace_bytes(bits, H, hd, L) with L = kcol + 1; 256-byte aligned).
 * fuse_softmax: 0 = softmax as its own launch, 1 (or 2) = inside the p.V kernel,
 * 3 = kvq_fused_attend (one kernel for q.K^T, softmax and p.V; shapes it does not take run as 1). */
KVQ_API size_t kvq_decode_step_workspace_bytes(int bits, int H, int hd, int64_t L);
KVQ_API int kvq_decode_step(const kvq_layer *layer, int64_t kcol, int64_t vcol, const void *q, const void *k,
                    const void *v, int acts_are_half, const kvq_sinks *sinks, const uint16_t *v_sink,
                    uint16_t *sink_probs, float *out, int fuse_softmax, void *workspace, size_t workspace_bytes,
                    void *stream);
/* measurement hook: events4 = four hipEvent_t (or NULL entries) recorded on the stream before / after the q.K^T launch
 * and before / after the p.V launches (p.V kernel + slab reduce; with fuse_softmax: from the fused kernel's launch, i.e.
 * behind the small merge of the softmax partials that long caches need) of the NEXT
 * kvq_decode_step on this thread; the hook clears itself after that call.  NULL: no events. */
KVQ_API int kvq_decode_step_events(void *const *events4);
/* the launch sequence the last kvq_decode_step on this thread took: 0 = softmax as its own launch, 1 = softmax inside the
 * p.V kernel, 3 = kvq_fused_attend (a request for 3 that the shape does not allow runs as 1); -1 = none yet.  For callers
 * that account bytes per kernel (bench.py). */
KVQ_API int kvq_decode_step_route(void);
/* the same for a stack of layers that follow each other without other work in between (all at the same column;
 * q / k / v / out: arrays of n_layers pointers; one workspace, reused layer after layer) */
KVQ_API int kvq_decode_steps(int n_layers, const kvq_layer *layers, int64_t col, const void *const *q,
                     const void *const *k, const void *const *v, int acts_are_half, float *const *out,
                     int fuse_softmax, void *workspace, size_t workspace_bytes, void *stream);

/* ---- one decode stream over a context split along the token axis (one shard per GPU) ---- */

/* The query-premultiplied K codebook images of kvq_score_k_prepared[_softmax] into `workspace`
 * (kvq_score_k_workspace_bytes(bits, 1, H)) without an append: the shards that do not hold the newest token only score.
 * q: [H][128] RoPE'd query, fp32 or fp16; lut: the table the scores dequantise with.  sinks (optional): the fp16 sink
 * tokens this shard holds -- their scaled scores are written as kvq_decode_prologue writes them. */
KVQ_API int kvq_score_k_tables(int bits, const void *q, int q_is_half, const float *lut, int H, int hd,
                       const kvq_sinks *sinks, void *workspace, size_t workspace_bytes, void *stream);
/* stats[h] = (max, sum of exp(x - max)) of head h's scaled scores over the shard, merged from the partials that
 * kvq_score_k_prepared_softmax wrote (the per-row part of kvq_softmax_finish) and, for the shard that holds them, the
 * scaled scores of the fp16 sink tokens (fp16 [H][n_sink] or NULL / 0); float [H][2]. */
KVQ_API int kvq_softmax_stats(const float *parts, int n_parts, const uint16_t *sink_scores, int n_sink, int H,
                      float *stats, void *stream);
/* Exact merge of n_shards locally normalised attention outputs (flash-decoding across devices): packed = n_shards
 * records of [H*hd floats: the shard's output][H x (max, normaliser): kvq_softmax_stats], e.g. the result of ONE
 * all-gather per layer; a shard without tokens carries (-inf, 0).  out: float [H][hd]. */
KVQ_API int kvq_combine_shards(const float *packed, int n_shards, int H, int hd, float *out, void *stream);

/* HEAD-sharded cache (SURVEY 8e "by head"; no counterpart in the reference, whose placement is by layer, ML:2428-2453):
 * a rank that holds heads [h0, h0 + n_heads) of a layer appends the WHOLE token into a full-width staging cache with the
 * ordinary append / pack entries above -- the outlier selection and V's codebook row are properties of the whole token
 * (ML:742, 1093-1119), so every rank selects bit-identically -- and this call copies columns [src_col, src_col + n) of
 * the staging caches into columns [dst_col, dst_col + n) of the shard's: the heads' packed words of K and V, the tokens' V
 * codebook rows, and the outlier rows with the shard's entries kept (channel rebased by -h0*hd) and the other ranks'
 * entries zeroed (value 0, channel clamped to the shard's first / last, so rows stay sorted; zero entries are skipped by
 * the matvecs like the reference's capped-away slots, ML:745-747).  k_out_t_dst / k_idx_t_dst: the shard's
 * token-contiguous K mirror [n_out][dst_max_len] or NULL.  n_out = 0: dense caches.  Reference outlier format only. */
KVQ_API int kvq_extract_heads(int bits, int H, int hd, int h0, int n_heads, int n_out, const int32_t *k_src,
                      const int32_t *v_src, int64_t src_max_len, int64_t src_col, const float *k_out_src,
                      const int32_t *k_idx_src, const float *v_out_src, const int32_t *v_idx_src,
                      const float *v_rows_src, int32_t *k_dst, int32_t *v_dst, int64_t dst_max_len, int64_t dst_col,
                      float *k_out_dst, int32_t *k_idx_dst, float *k_out_t_dst, int32_t *k_idx_t_dst, float *v_out_dst,
                      int32_t *v_idx_dst, float *v_rows_dst, const float *v_rows2_src, float *v_rows2_dst, int64_t n,
                      void *stream);
/* K append | V append of one token into column `col` of a layer's cache as ONE launch (the two selection workgroups of
 * kvq_decode_prologue without its table roles; k, v: [H*hd] fp32 or fp16). */
KVQ_API int kvq_append_kv_fused(const kvq_layer *layer, int64_t col, const void *k, const void *v, int acts_are_half,
                        void *stream);
/* One decode token's attention over the L tokens a layer's cache already holds, WITHOUT an append: query tables (+ fp16
 * sink scores) -> q.K^T -> softmax -> p.V (+ slab reduce): the launches of kvq_decode_step behind its prologue.  For the
 * shards of a split stream.  Arguments as kvq_decode_step; workspace: kvq_decode_step_workspace_bytes(bits, H, hd, L). */
KVQ_API int kvq_attend_step(const kvq_layer *layer, int64_t L, const void *q, int acts_are_half, const kvq_sinks *sinks,
                    const uint16_t *v_sink, uint16_t *sink_probs, float *out, int fuse_softmax, void *workspace,
                    size_t workspace_bytes, void *stream);
/* One decode token through one HEAD SHARD of a layer, ONE call: kvq_append_kv_fused into column 0 of the full-width
 * staging cache `full` (the selection is a property of the whole token: bit-identical on every rank), kvq_extract_heads
 * of heads [h0, h0 + shard->H) into column `col` of the shard's cache (Q-Norm rows travel when both layers carry them),
 * kvq_attend_step over the shard's col + 1 tokens.  q: the shard's heads of the RoPE'd query [shard->H][128]; k, v: the
 * whole token; sinks / v_sink / sink_probs: the shard's heads of the fp16 sink caches; out f32 [shard->H][hd].
 * workspace: kvq_decode_step_workspace_bytes(bits, shard->H, hd, col + 1).  `v_rows2_*` of kvq_extract_heads above: the
 * Q-Norm codebook rows (kvq_vopts.lut_rows2) of the extracted tokens, or NULL. */
KVQ_API int kvq_head_shard_step(const kvq_layer *full, const kvq_layer *shard, int h0, int64_t col, const void *q,
                        const void *k, const void *v, int acts_are_half, const kvq_sinks *sinks, const uint16_t *v_sink,
                        uint16_t *sink_probs, float *out, int fuse_softmax, void *workspace, size_t workspace_bytes,
                        void *stream);

/* ---- prefill attention (the MFMA path of BASELINE config 4) ----------------------- */

/* Causal self-attention of the S prompt tokens of one sequence, all heads, flash-style on the gfx950 matrix cores
 * (v_mfma_f32_32x32x16_f16, fp32 accumulation, online softmax): what the reference delegates to flash-attn in its
 * prefill branch (modeling_llama.py:1861-1874, 2013-2070).  q, k, v: fp16, already RoPE'd, element [h][s][d] at
 * h*stride_h + s*stride_s + d (strides in elements, multiples of 8, bases 16-byte aligned; d contiguous);
 * out: fp16, same addressing (strides multiples of 4).  head_dim must be 128.
 * out[h][s][:] = softmax_j<=s(q[h][s].k[h][j] * softmax_scale) . v[h][j][:]. */
KVQ_API int kvq_prefill_attention(const void *q, const void *k, const void *v, void *out, int H, int S, int hd,
                          int64_t q_stride_h, int64_t q_stride_s, int64_t k_stride_h, int64_t k_stride_s,
                          int64_t v_stride_h, int64_t v_stride_s, int64_t o_stride_h, int64_t o_stride_s,
                          float softmax_scale, void *stream);

/* ---- uncapped ("orig") Dense-and-Sparse variants, 4 bit only ------------------ */

/* VecQuant4AppendVecKSparseOrig + ...2Orig (KCU:691-931): outlier iff x<lo || x>hi ->
 * code 7 and a sparse entry (channel, x - zeropoint[c]) in ascending channel order;
 * out_idx / out_val need room for H*hd entries; *out_count (device) = number written.
 * Growing the CSR arrays is left to the caller, as in the reference's host code. */
KVQ_API int kvq_append_k_sparse_orig(int32_t *mat, const float *lut, const float *x,
                             const float *zeropoint, const float *lo, const float *hi,
                             int32_t *out_idx, float *out_val, int32_t *out_count,
                             int H, int hd, int64_t max_len, int64_t col, void *stream);
/* VecQuant4AppendVecVSparseOrig (KCU:933-1163): scalar thresholds / zero point,
 * per-token codebook row lut_rows[col]. */
KVQ_API int kvq_append_v_sparse_orig(int32_t *mat, const float *lut_rows, const float *x,
                             float zeropoint, float lo, float hi, int32_t *out_idx,
                             float *out_val, int32_t *out_count, int H, int hd,
                             int64_t max_len, int64_t col, void *stream);
/* SPMV_ATOMIC_CSR_ROPE_BALANCED (KCU:524-614): mul[head][t] += sum over the CSR row t
 * (rowptr[num_rows+1], cols = global channel) of val * RoPE-weighted q; q row 0 only. */
KVQ_API int kvq_spmv_k_rope_csr(const int32_t *rowptr, const int32_t *cols, const float *vals,
                        const float *q, float *mul, int64_t num_rows, int64_t L, int hd,
                        float rope_theta, int pos_offset, void *stream);
/* SPMV_ATOMIC_CSC_BALANCED (KCU:617-689): mul[row] += val * p[row/128][t] over the CSC
 * column t (colptr[num_cols+1]). */
KVQ_API int kvq_spmv_v_csc(const int32_t *colptr, const int32_t *rows, const float *vals,
                   const float *p, float *mul, int64_t num_cols, int64_t L, int H,
                   int hd, void *stream);

#ifdef __cplusplus
}
#endif
#endif /* KVQ_H */
