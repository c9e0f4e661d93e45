/*
 * kvq_oracle.c -- CPU restatement of the KVQuant deployment kernels.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under kvquant_amd/ may import, link or
 * call this file.  Only tests/, __graft_entry__.smoke() and bench.py's
 * cpu_baseline leg use it, and only as the checker.
 *
 * Parity status: PINNED to the running reference.  The reference ships no
 * golden vectors / KATs for this path, but its own extension
 * (quant_cuda.cpp + quant_cuda_kernel.cu) is built for gfx950 from the sources
 * where they lie by oracle/build_ref.py -> oracle/_ref/ (git-ignored), and
 * tests/test_ref_gpu.py runs all 34 of its ops on the MI355X next to this
 * file and next to the HIP kernels: packed codes, rescaled values and CSR/CSC
 * bookkeeping bit for bit, q.K^T / p.V within the 1e-3 tolerance (measured
 * ~1e-6 .. 3e-4; the reference sums with atomics and is not bit-reproducible
 * against itself).  Known reference defects that show up there are listed in
 * that file's header.  The host glue around the kernels (oracle/glue.py) is
 * pinned separately: tests/golden/ holds outputs of the reference's own
 * QuantK/QuantV Python classes (tests/golden/gen_golden.py).
 *
 * Short names:
 *   KCU = /root/reference/deployment/kvquant/quant_cuda_kernel.cu
 *
 * Cache layout (KCU:1240-1244, 1395-1424, 2712-2716; shared by K and V):
 *   mat is int32 [n_rows][max_len] with n_rows = C/32*bits; the token index
 *   is the contiguous one.  For global channel c (= head*head_dim + k):
 *     4-bit: row c/8,  bits 4*(c%8)
 *     2-bit: row c/16, bits 2*(c%16)
 *     3-bit: group g=c/32, loc=c%32, rows 3g..3g+2 hold 32 codes:
 *            loc 0..9  -> row 3g   bits 3*loc
 *            loc 10    -> low 2 bits at row 3g [31:30], high bit row 3g+1 [0]
 *            loc 11..20-> row 3g+1 bits (3*loc)%32 = 1,4,..,28
 *            loc 21    -> low bit row 3g+1 [31], high 2 bits row 3g+2 [1:0]
 *            loc 22..31-> row 3g+2 bits (3*loc)%32 = 2,5,..,29
 *
 * Build: see oracle/Makefile (gcc -O2 -ffp-contract=off: the reference
 * kernels' a*b+c sequences are NOT contracted here; float division and
 * sqrt are IEEE).
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#ifdef _OPENMP
#include <omp.h>
#endif

#define KVQO_EXPORT __attribute__((visibility("default")))

/* ------------------------------------------------------------------ */
/* bit layout helpers                                                  */
/* ------------------------------------------------------------------ */

/* OR code (0..2^bits-1) of global channel c into column col.
 * Mirrors the atomicAdd-into-zeroed-word of KCU:1240-1244 (4b),
 * KCU:1395-1424 (3b), KCU:1602-1606 (2b): on a zero column add == or. */
static void put_code(int bits, int32_t *mat, int64_t max_len, int64_t col,
                     int c, unsigned code) {
  uint32_t *m = (uint32_t *)mat;
  if (bits == 4) {
    m[(int64_t)(c / 8) * max_len + col] += code << (4 * (c % 8));
  } else if (bits == 2) {
    m[(int64_t)(c / 16) * max_len + col] += code << (2 * (c % 16));
  } else { /* 3 */
    int g = c / 32, loc = c % 32;
    int64_t r0 = (int64_t)(3 * g) * max_len + col;
    if (loc == 10) {
      m[r0] += code << 30;               /* low 2 bits survive the shift */
      m[r0 + max_len] += code >> 2;
    } else if (loc == 21) {
      m[r0 + max_len] += code << 31;
      m[r0 + 2 * max_len] += code >> 1;
    } else {
      m[r0 + (int64_t)(loc / 11) * max_len] += code << ((loc * 3) % 32);
    }
  }
}

/* Decode side: KCU:3103-3199 (4b), 3771-3890 (3b), 4747-4995 (2b). */
static unsigned get_code(int bits, const int32_t *mat, int64_t max_len,
                         int64_t col, int c) {
  const uint32_t *m = (const uint32_t *)mat;
  if (bits == 4) {
    return (m[(int64_t)(c / 8) * max_len + col] >> (4 * (c % 8))) & 0xf;
  } else if (bits == 2) {
    return (m[(int64_t)(c / 16) * max_len + col] >> (2 * (c % 16))) & 0x3;
  } else {
    int g = c / 32, loc = c % 32;
    int64_t r0 = (int64_t)(3 * g) * max_len + col;
    uint32_t w0 = m[r0], w1 = m[r0 + max_len], w2 = m[r0 + 2 * max_len];
    if (loc < 10) return (w0 >> (3 * loc)) & 0x7;
    if (loc == 10) return ((w0 >> 30) & 0x3) | ((w1 & 0x1) << 2);
    if (loc < 21) return (w1 >> ((3 * loc) % 32)) & 0x7;
    if (loc == 21) return ((w1 >> 31) & 0x1) | ((w2 & 0x3) << 1);
    return (w2 >> ((3 * loc) % 32)) & 0x7;
  }
}

/* argmin_v |lut[v] - x|, strict '<' scan from v=0 so the first minimum
 * wins (KCU:1222-1237). */
static unsigned nearest_code(const float *lut, int n, float x) {
  unsigned best = 0;
  float prev = fabsf(lut[0] - x);
  for (int v = 1; v < n; v++) {
    float d = fabsf(lut[v] - x);
    if (d < prev) {
      prev = d;
      best = (unsigned)v;
    }
  }
  return best;
}

/* ------------------------------------------------------------------ */
/* append / pack                                                       */
/* ------------------------------------------------------------------ */

/* vecquant{b}appendvecK (KCU:1167-1245, 1322-1425, 1528-1607):
 * per-channel LUT lut[C][n]; x[C]; writes column col. */
KVQO_EXPORT void kvqo_append_k(int bits, int32_t *mat, const float *lut,
                               const float *x, int C, int64_t max_len,
                               int64_t col) {
  int n = 1 << bits;
  for (int c = 0; c < C; c++)
    put_code(bits, mat, max_len, col, c, nearest_code(lut + (int64_t)c * n, n, x[c]));
}

/* vecquant{b}appendvecV (KCU:1248-1320, 1427-1526, 1609-1682):
 * per-token LUT row lut_rows[col][n]. */
KVQO_EXPORT void kvqo_append_v(int bits, int32_t *mat, const float *lut_rows,
                               const float *x, int C, int64_t max_len,
                               int64_t col) {
  int n = 1 << bits;
  const float *lut = lut_rows + col * n;
  for (int c = 0; c < C; c++)
    put_code(bits, mat, max_len, col, c, nearest_code(lut, n, x[c]));
}

/* vecquant{b}appendvecKsparse (KCU:1684-1781, 2104-2227, 2620-2717):
 * as append_k, plus rescaled[c] = (x - zp)/range with zp=(up+lo)/2,
 * range=(up-lo)/2 in fp32 (KCU:1759-1764).  Outliers are not masked; they
 * saturate to an end code. */
KVQO_EXPORT void kvqo_append_k_sparse(int bits, int32_t *mat, const float *lut,
                                      const float *x, float *rescaled,
                                      const float *lo, const float *hi, int C,
                                      int64_t max_len, int64_t col) {
  int n = 1 << bits;
  for (int c = 0; c < C; c++) {
    float rangeval = (hi[c] - lo[c]) / 2;
    float zeropoint = (hi[c] + lo[c]) / 2;
    rescaled[c] = (x[c] - zeropoint) / rangeval;
    put_code(bits, mat, max_len, col, c, nearest_code(lut + (int64_t)c * n, n, x[c]));
  }
}

/* vecquant{b}appendvecVsparse (KCU:2011-2102, 2369-2486, 2947-3038):
 * code = (x<lo || x>hi) ? zero-point code (7/3/1) : argmin over the
 * per-token LUT row. */
static unsigned zero_code(int bits) { return bits == 4 ? 7u : (bits == 3 ? 3u : 1u); }

KVQO_EXPORT void kvqo_append_v_sparse(int bits, int32_t *mat,
                                      const float *lut_rows, const float *x,
                                      float lo, float hi, int C,
                                      int64_t max_len, int64_t col) {
  int n = 1 << bits;
  const float *lut = lut_rows + col * n;
  for (int c = 0; c < C; c++) {
    unsigned code;
    if (x[c] < lo || x[c] > hi) code = zero_code(bits);
    else code = nearest_code(lut, n, x[c]);
    put_code(bits, mat, max_len, col, c, code);
  }
}

/* vecquant{b}appendvecKsparseParallel (KCU:1783-1898, 2229-2366,
 * 2719-2834): x is [C][S] (channel-major, token contiguous); writes columns
 * col0..col0+S-1 (the reference always has col0 = 0); rescaled is [C][S].
 * The reference reads LDS written by other threads without a barrier
 * (KCU:1857-1877); this restates the intended semantics. */
KVQO_EXPORT void kvqo_pack_k_sparse_parallel(int bits, int32_t *mat,
                                             const float *lut, const float *x,
                                             float *rescaled, const float *lo,
                                             const float *hi, int C, int64_t S,
                                             int64_t max_len, int64_t col0) {
  int n = 1 << bits;
#pragma omp parallel for schedule(static)
  for (int64_t t = 0; t < S; t++) {
    for (int c = 0; c < C; c++) {
      float rangeval = (hi[c] - lo[c]) / 2;
      float zeropoint = (hi[c] + lo[c]) / 2;
      float xv = x[(int64_t)c * S + t];
      rescaled[(int64_t)c * S + t] = (xv - zeropoint) / rangeval;
      put_code(bits, mat, max_len, col0 + t, c,
               nearest_code(lut + (int64_t)c * n, n, xv));
    }
  }
}

/* vecquant{b}appendvecVsparseParallel (KCU:1900-2009, 2488-2618,
 * 2836-2945): per-token LUT rows lut_rows[col0+t], per-token thresholds
 * lo[t], hi[t].  (The 3-bit reference scans the wrong LUT column,
 * KCU:2574-2579; intended semantics here.) */
KVQO_EXPORT void kvqo_pack_v_sparse_parallel(int bits, int32_t *mat,
                                             const float *lut_rows,
                                             const float *x, const float *lo,
                                             const float *hi, int C, int64_t S,
                                             int64_t max_len, int64_t col0) {
  int n = 1 << bits;
#pragma omp parallel for schedule(static)
  for (int64_t t = 0; t < S; t++) {
    const float *lut = lut_rows + (col0 + t) * n;
    for (int c = 0; c < C; c++) {
      float xv = x[(int64_t)c * S + t];
      unsigned code;
      if (xv < lo[t] || xv > hi[t]) code = zero_code(bits);
      else code = nearest_code(lut, n, xv);
      put_code(bits, mat, max_len, col0 + t, c, code);
    }
  }
}

/* ------------------------------------------------------------------ */
/* unpack (test helper, no reference counterpart)                      */
/* ------------------------------------------------------------------ */
KVQO_EXPORT void kvqo_unpack_codes(int bits, const int32_t *mat, uint8_t *codes,
                                   int C, int64_t L, int64_t max_len) {
  for (int64_t t = 0; t < L; t++)
    for (int c = 0; c < C; c++)
      codes[t * C + c] = (uint8_t)get_code(bits, mat, max_len, t, c);
}

/* ------------------------------------------------------------------ */
/* K score: q.K^T with RoPE applied to the dequantised pre-RoPE key      */
/* ------------------------------------------------------------------ */

/* theta_k = powf(rope_theta, -2*(k % (hd/2))/hd) (KCU:3081, 508, 584).  The
 * reference evaluates this with the DEVICE's powf; 1 ulp of theta_k moves the
 * angle by 6e-8 * position radians, so at long contexts the platform's powf is
 * part of the result (measured against the reference's own kernels on MI355X,
 * tests/test_ref_gpu.py: the device powf differs from the correctly rounded
 * value by 1 ulp for 21 of the 64 frequencies of theta = 10000, which is up to
 * 2e-2 relative in a score at position 1e6).  Default here: the host libm's
 * powf.  A checker that compares with a GPU run installs the device's table
 * with kvqo_set_rope_freqs (obtained through kvq_rope_freqs of include/kvq.h or
 * from the reference extension itself). */
static float g_freq_tab[256];
static float g_freq_theta = 0.f;
static int g_freq_n = 0;

KVQO_EXPORT void kvqo_set_rope_freqs(float rope_theta, const float *f, int n) {
  if (!f || n <= 0 || n > 256) {
    g_freq_n = 0;
    return;
  }
  for (int i = 0; i < n; i++) g_freq_tab[i] = f[i];
  g_freq_theta = rope_theta;
  g_freq_n = n;
}

KVQO_EXPORT float kvqo_rope_freq(float rope_theta, int k, int hd) {
  int j = k % (hd / 2);
  if (g_freq_n == hd / 2 && g_freq_theta == rope_theta) return g_freq_tab[j];
  float e = -2.0f * (float)j / (float)hd;
  return powf(rope_theta, e);
}

/* vecquant{b}matmul_nuq_perchannel_transposed_rope_mha_batched_fused_opt
 * (KCU:3040-3209 / 3692-4115 / 4747-4995):
 *   mul[b][h][t] += sum_k lut[h*hd+k][code] * (cos(th_k*pos) * q[b][h][k]
 *                      + sign_k * sin(th_k*pos) * q[b][h][(k+hd/2)%hd])
 * pos = t + pos_offset, fp32 throughout, k ascending, the two products of a
 * channel added one after the other exactly as KCU:3122-3126. */
KVQO_EXPORT void kvqo_score_k(int bits, const float *q, const int32_t *mat,
                              float *mul, const float *lut, int q_len, int H,
                              int hd, int64_t L, int64_t max_len,
                              float rope_theta, int pos_offset) {
  int n = 1 << bits;
  int hd2 = hd / 2;
  float *freq = (float *)malloc(sizeof(float) * hd);
  for (int k = 0; k < hd; k++) freq[k] = kvqo_rope_freq(rope_theta, k, hd);
#pragma omp parallel for schedule(static)
  for (int64_t t = 0; t < L; t++) {
    int pos = (int)t + pos_offset;
    for (int b = 0; b < q_len; b++) {
      for (int h = 0; h < H; h++) {
        const float *qh = q + ((int64_t)b * H + h) * hd;
        float res = 0;
        for (int k = 0; k < hd; k++) {
          int c = h * hd + k;
          float sign = (k < hd2) ? 1.0f : -1.0f;
          float tmp1 = lut[(int64_t)c * n + get_code(bits, mat, max_len, t, c)];
          float theta = freq[k];
          float cs = cosf(theta * pos);
          float sn = sinf(theta * pos);
          res += tmp1 * cs * qh[k];
          res += sign * tmp1 * sn * qh[(k + hd2) % hd];
        }
        mul[((int64_t)b * H + h) * L + t] += res;
      }
    }
  }
  free(freq);
}

/* SPMV_ATOMIC_ROPE_BALANCED (KCU:473-521): fixed-width sparse K.  Only
 * batch 0 of q / mul is touched (KCU:3605 "TODO batching"). */
KVQO_EXPORT void kvqo_spmv_k_rope(const float *outliers, const int32_t *idx,
                                  const float *q, float *mul, int64_t L, int H,
                                  int hd, int num_outliers, float rope_theta,
                                  int pos_offset) {
  (void)H;
  int hd2 = hd / 2;
  for (int64_t t = 0; t < L; t++) {
    for (int i = 0; i < num_outliers; i++) {
      int col = idx[t * num_outliers + i];
      float mat_tmp = outliers[t * num_outliers + i];
      int headid = col / hd;
      int ch = col % hd;
      float theta = kvqo_rope_freq(rope_theta, ch, hd);
      float sign = (ch < hd2) ? 1.0f : -1.0f;
      float cs = cosf(theta * (float)((int)t + pos_offset));
      float sn = sinf(theta * (float)((int)t + pos_offset));
      int col2 = ((ch + hd2) % hd) + headid * hd;
      float dot = mat_tmp * cs * q[col];
      dot += sign * mat_tmp * sn * q[col2];
      mul[(int64_t)headid * L + t] += dot;
    }
  }
}

/* ------------------------------------------------------------------ */
/* V mix: p.V with per-token LUT rows                                  */
/* ------------------------------------------------------------------ */

/* vecquant{b}matmul_nuq_perchannel_transposed_mha_batched_fused_opt
 * (KCU:3211-3433 / 4117-4491 / 4998-5248):
 *   mul[b][h][c] += sum_t lut_rows[t][code(h,c,t)] * p[b][h][t]. */
KVQO_EXPORT void kvqo_mix_v(int bits, const float *p, const int32_t *mat,
                            float *mul, const float *lut_rows, int q_len, int H,
                            int hd, int64_t L, int64_t max_len) {
  int n = 1 << bits;
#pragma omp parallel for schedule(static) collapse(2)
  for (int b = 0; b < q_len; b++) {
    for (int h = 0; h < H; h++) {
      const float *ph = p + ((int64_t)b * H + h) * L;
      for (int k = 0; k < hd; k++) {
        int c = h * hd + k;
        float res = 0;
        for (int64_t t = 0; t < L; t++)
          res += lut_rows[t * n + get_code(bits, mat, max_len, t, c)] * ph[t];
        mul[((int64_t)b * H + h) * hd + k] += res;
      }
    }
  }
}

/* SPMV_ATOMIC_BALANCED (KCU:437-470): fixed-width sparse V, batch 0 only. */
KVQO_EXPORT void kvqo_spmv_v(const float *outliers, const int32_t *idx,
                             const float *p, float *mul, int64_t L, int H,
                             int hd, int num_outliers) {
  (void)H;
  for (int64_t t = 0; t < L; t++) {
    for (int i = 0; i < num_outliers; i++) {
      int row = idx[t * num_outliers + i];
      int headid = row / hd;
      mul[row] += outliers[t * num_outliers + i] * p[(int64_t)headid * L + t];
    }
  }
}

/* ------------------------------------------------------------------ */
/* uncapped ("orig") CSR / CSC path, 4-bit only                         */
/* ------------------------------------------------------------------ */

/* VecQuant4AppendVecKSparseOrig + ...2Orig (KCU:868-931, 832-866):
 * outlier iff x<lo || x>hi -> code forced to 7 and a sparse entry
 * (c, x - (hi+lo)/2) is emitted in ascending c; otherwise nearest code.
 * Returns the number of outliers written to out_idx/out_val (capacity C). */
KVQO_EXPORT int kvqo_append_k_sparse_orig(int32_t *mat, const float *lut,
                                          const float *x, const float *zeropoint,
                                          const float *lo, const float *hi,
                                          int32_t *out_idx, float *out_val,
                                          int C, int64_t max_len, int64_t col) {
  int cnt = 0;
  for (int c = 0; c < C; c++) {
    unsigned code;
    if (x[c] < lo[c] || x[c] > hi[c]) {
      code = 7;
      out_idx[cnt] = c;
      out_val[cnt] = x[c] - zeropoint[c];
      cnt++;
    } else {
      code = nearest_code(lut + (int64_t)c * 16, 16, x[c]);
    }
    put_code(4, mat, max_len, col, c, code);
  }
  return cnt;
}

/* VecQuant4AppendVecVSparseOrig (KCU:1103-1163, 1068-1101): scalar
 * thresholds and scalar zero point, per-token LUT row. */
KVQO_EXPORT int kvqo_append_v_sparse_orig(int32_t *mat, const float *lut_rows,
                                          const float *x, float zeropoint,
                                          float lo, float hi, int32_t *out_idx,
                                          float *out_val, int C,
                                          int64_t max_len, int64_t col) {
  int cnt = 0;
  const float *lut = lut_rows + col * 16;
  for (int c = 0; c < C; c++) {
    unsigned code;
    if (x[c] < lo || x[c] > hi) {
      code = 7;
      out_idx[cnt] = c;
      out_val[cnt] = x[c] - zeropoint;
      cnt++;
    } else {
      code = nearest_code(lut, 16, x[c]);
    }
    put_code(4, mat, max_len, col, c, code);
  }
  return cnt;
}

/* SPMV_ATOMIC_CSR_ROPE_BALANCED (KCU:524-614): CSR rows = tokens,
 * rowptr[num_rows+1], cols = global channel.  The thread balancing of the
 * reference does not change the sum; this walks rows directly. */
KVQO_EXPORT void kvqo_spmv_k_rope_csr(const int32_t *rowptr, const int32_t *cols,
                                      const float *vals, const float *q,
                                      float *mul, int64_t num_rows, int64_t L,
                                      int hd, float rope_theta, int pos_offset) {
  int hd2 = hd / 2;
  for (int64_t t = 0; t < num_rows; t++) {
    for (int i = rowptr[t]; i < rowptr[t + 1]; i++) {
      int col = cols[i];
      int headid = col / hd, ch = col % hd;
      float theta = kvqo_rope_freq(rope_theta, ch, hd);
      float sign = (ch < hd2) ? 1.0f : -1.0f;
      float cs = cosf(theta * (float)((int)t + pos_offset));
      float sn = sinf(theta * (float)((int)t + pos_offset));
      int col2 = ((ch + hd2) % hd) + headid * hd;
      float dot = vals[i] * cs * q[col];
      dot += sign * vals[i] * sn * q[col2];
      mul[(int64_t)headid * L + t] += dot;
    }
  }
}

/* SPMV_ATOMIC_CSC_BALANCED (KCU:617-689): CSC columns = tokens. */
KVQO_EXPORT void kvqo_spmv_v_csc(const int32_t *colptr, const int32_t *rows,
                                 const float *vals, const float *p, float *mul,
                                 int64_t num_cols, int64_t L, int hd) {
  for (int64_t t = 0; t < num_cols; t++)
    for (int i = colptr[t]; i < colptr[t + 1]; i++) {
      int row = rows[i];
      mul[row] += vals[i] * p[(int64_t)(row / hd) * L + t];
    }
}

/* ------------------------------------------------------------------ */
/* simulated-quant CPU baseline (bench.py cpu_baseline leg)             */
/* ------------------------------------------------------------------ */

/* One decode step of the reference's CPU path on already reconstructed
 * (simulated-quant) K/V, as quant/llama_simquant.py evaluates it: RoPE on
 * K-hat, q.K^T/sqrt(hd), fp32 softmax, p.V-hat.  khat/vhat are [L][C]
 * (token-major), q is post-RoPE [C], out [C].  Not a parity oracle: a timing
 * baseline whose arithmetic is the plain fp32 attention the simulated path
 * feeds (SURVEY.md 8d "CPU baseline"). */
KVQO_EXPORT void kvqo_sim_decode_step(const float *khat, const float *vhat,
                                      const float *q, float *out, float *scores,
                                      int H, int hd, int64_t L, float rope_theta,
                                      int pos_offset) {
  int C = H * hd, hd2 = hd / 2;
  float *freq = (float *)malloc(sizeof(float) * hd2);
  for (int k = 0; k < hd2; k++) freq[k] = kvqo_rope_freq(rope_theta, k, hd);
  float inv = 1.0f / sqrtf((float)hd);
#pragma omp parallel for schedule(static)
  for (int64_t t = 0; t < L; t++) {
    float pos = (float)((int)t + pos_offset);
    const float *kt = khat + t * C;
    for (int h = 0; h < H; h++) {
      const float *kh = kt + h * hd, *qh = q + h * hd;
      float res = 0;
      for (int j = 0; j < hd2; j++) {
        float cs = cosf(freq[j] * pos), sn = sinf(freq[j] * pos);
        float k0 = kh[j], k1 = kh[j + hd2];
        res += (k0 * cs - k1 * sn) * qh[j];
        res += (k1 * cs + k0 * sn) * qh[j + hd2];
      }
      scores[(int64_t)h * L + t] = res * inv;
    }
  }
#pragma omp parallel for schedule(static)
  for (int h = 0; h < H; h++) {
    float *s = scores + (int64_t)h * L;
    float m = -INFINITY, z = 0;
    for (int64_t t = 0; t < L; t++) m = s[t] > m ? s[t] : m;
    for (int64_t t = 0; t < L; t++) { s[t] = expf(s[t] - m); z += s[t]; }
    float *o = out + h * hd;
    for (int k = 0; k < hd; k++) o[k] = 0;
    for (int64_t t = 0; t < L; t++) {
      float p = s[t] / z;
      const float *vh = vhat + t * C + h * hd;
      for (int k = 0; k < hd; k++) o[k] += p * vh[k];
    }
  }
  free(freq);
}

KVQO_EXPORT int kvqo_num_threads(void) {
#ifdef _OPENMP
  return omp_get_max_threads();
#else
  return 1;
#endif
}
