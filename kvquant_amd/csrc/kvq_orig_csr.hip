// Uncapped ("orig") Dense-and-Sparse variants, 4 bit only: every value outside its thresholds
// is an outlier -- code forced to 7, exact residual kept in a growing CSR (K: rows = tokens) /
// CSC (V: columns = tokens) matrix.  Reference: KCU:691-931 (K append), 933-1163 (V append),
// 524-614 (CSR SpMV with RoPE), 617-689 (CSC SpMV), launchers 5506-5668.
//
// The reference compacts with one serial thread per 32-channel block plus a host round trip for
// the count; here one 1024-lane workgroup packs, flags and compacts in channel order with a block
// scan and leaves the count on the device (the Python shim reads it to grow the CSR arrays, as
// the reference's host code does).  The SpMVs need no atomics: a lane owns a CSR row (token) and
// folds runs of equal heads before its plain store; the CSC product accumulates per workgroup in
// 32.32 fixed point in LDS (ds_add_u64) and adds each workgroup's sums to `mul` once.
#include "kvq_common.h"
#include "kvq_host.h"

#include <cmath>

namespace kvq {

constexpr int kOrigThreads = 1024;

__device__ __forceinline__ uint32_t block_scan_excl_1024(uint32_t v, uint32_t *ws, uint32_t &total) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  uint32_t inc = v;
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) {
    uint32_t o = __shfl_up(inc, d);
    if (lane >= d) inc += o;
  }
  __syncthreads();
  if (lane == 63) ws[wave] = inc;
  __syncthreads();
  uint32_t base = 0, tot = 0;
  for (int w = 0; w < kOrigThreads / 64; w++) {
    const uint32_t s = ws[w];
    if (w < wave) base += s;
    tot += s;
  }
  total = tot;
  return base + inc - v;
}

// IS_V: per-token LUT row lut[col], scalar thresholds / zero point; else per-channel everything.
template <bool IS_V>
__global__ __launch_bounds__(kOrigThreads) void append_orig_kernel(
    uint32_t *__restrict__ mat, const float *__restrict__ lut, const float *__restrict__ x,
    const float *__restrict__ zp_c, const float *__restrict__ lo_c, const float *__restrict__ hi_c, float zp_s,
    float lo_s, float hi_s, int32_t *__restrict__ out_idx, float *__restrict__ out_val,
    int32_t *__restrict__ out_count, int C, int64_t max_len, int64_t col) {
  __shared__ unsigned codes[8192];
  __shared__ uint32_t ws[kOrigThreads / 64];
  const int tid = threadIdx.x;
  const int per = (C + kOrigThreads - 1) / kOrigThreads;
  const int c0 = tid * per;
  uint32_t nout = 0;
  for (int e = 0; e < per; e++) {
    const int c = c0 + e;
    if (c >= C) break;
    const float xv = x[c];
    const float lo = IS_V ? lo_s : lo_c[c], hi = IS_V ? hi_s : hi_c[c];
    float row[16];
    const float *src = IS_V ? lut + col * 16 : lut + (int64_t)c * 16;
#pragma unroll
    for (int v = 0; v < 16; v += 4) {
      const float4 t4 = *reinterpret_cast<const float4 *>(src + v);
      row[v] = t4.x; row[v + 1] = t4.y; row[v + 2] = t4.z; row[v + 3] = t4.w;
    }
    const bool outl = (xv < lo) || (xv > hi);                 // KCU:909 / 1144
    codes[c] = outl ? 7u : nearest_code<16>(row, xv);
    nout += outl;
  }
  uint32_t total;
  uint32_t pos = block_scan_excl_1024(nout, ws, total);      // (barriers inside publish `codes`)
  for (int e = 0; e < per; e++) {
    const int c = c0 + e;
    if (c >= C) break;
    const float xv = x[c];
    const float lo = IS_V ? lo_s : lo_c[c], hi = IS_V ? hi_s : hi_c[c];
    if ((xv < lo) || (xv > hi)) {
      out_idx[pos] = c;
      out_val[pos] = xv - (IS_V ? zp_s : zp_c[c]);            // KCU:855-857 / 1091-1092
      pos++;
    }
  }
  if (tid == 0) *out_count = (int32_t)total;
  for (int g = tid; g < C / 32; g += kOrigThreads) {
    unsigned cd[32];
#pragma unroll
    for (int i = 0; i < 32; i++) cd[i] = codes[g * 32 + i];
    uint32_t w[4];
    pack32<4>(cd, w);
#pragma unroll
    for (int i = 0; i < 4; i++) mat[((int64_t)g * 4 + i) * max_len + col] = w[i];
  }
}

// one lane per CSR row (token): runs of equal heads are folded, then one plain add per (head, token)
__global__ __launch_bounds__(256) void spmv_k_rope_csr_kernel(const int32_t *__restrict__ rowptr,
                                                              const int32_t *__restrict__ cols,
                                                              const float *__restrict__ vals,
                                                              const float *__restrict__ q, float *__restrict__ mul,
                                                              int64_t num_rows, int64_t L, int pos_offset,
                                                              float rope_theta) {
  __shared__ float theta[64];
  if (threadIdx.x < 64) theta[threadIdx.x] = rope_freq(rope_theta, threadIdx.x);   // KCU:584
  __syncthreads();
  const int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (t >= num_rows || t >= L) return;
  const float posf = (float)((int)t + pos_offset);
  int head = -1;
  float sum = 0.f;
  for (int i = rowptr[t]; i < rowptr[t + 1]; i++) {
    const int col = cols[i];
    const int h = col >> 7, ch = col & 127;
    if (h != head) {
      if (head >= 0) mul[(int64_t)head * L + t] += sum;
      head = h;
      sum = 0.f;
    }
    float s, c;
    sincos_rev(theta[ch & 63] * posf, s, c);
    const float q1 = q[col], q2 = q[(h << 7) + ((ch + 64) & 127)];
    sum += vals[i] * fmaf(c, q1, ((ch < 64) ? s : -s) * q2);
  }
  if (head >= 0) mul[(int64_t)head * L + t] += sum;
}

// CSC columns = tokens.  One workgroup per 1024 tokens, all channels in LDS fixed point.
__global__ __launch_bounds__(1024) void spmv_v_csc_kernel(const int32_t *__restrict__ colptr,
                                                          const int32_t *__restrict__ rows,
                                                          const float *__restrict__ vals,
                                                          const float *__restrict__ p, float *__restrict__ mul,
                                                          int64_t num_cols, int64_t L, int C) {
  __shared__ unsigned long long acc[4096];
  const int64_t t = (int64_t)blockIdx.x * 1024 + threadIdx.x;
  for (int c0 = 0; c0 < C; c0 += 4096) {
    const int cn = C - c0 < 4096 ? C - c0 : 4096;
    for (int i = threadIdx.x; i < cn; i += 1024) acc[i] = 0;
    __syncthreads();
    if (t < num_cols && t < L) {
      for (int i = colptr[t]; i < colptr[t + 1]; i++) {
        const int row = rows[i];
        const unsigned rel = (unsigned)(row - c0);
        if (rel < (unsigned)cn) {
          const float xx = vals[i] * p[(int64_t)(row >> 7) * L + t];
          const float fl = floorf(xx);
          const unsigned lo = (unsigned)((xx - fl) * 4294967296.0f);
          atomicAdd(&acc[rel], ((unsigned long long)(unsigned)(int)fl << 32) | lo);
        }
      }
    }
    __syncthreads();
    for (int i = threadIdx.x; i < cn; i += 1024) {
      const long long v = (long long)acc[i];
      if (v != 0) atomicAdd(&mul[c0 + i], (float)((double)v * (1.0 / 4294967296.0)));
    }
    __syncthreads();
  }
}

}  // namespace kvq

using namespace kvq;

extern "C" {

int kvq_append_k_sparse_orig(int32_t *mat, const float *lut, const float *x, const float *zeropoint,
                             const float *lo, const float *hi, int32_t *out_idx, float *out_val,
                             int32_t *out_count, int H, int hd, int64_t max_len, int64_t col, void *stream) {
  const int C = H * hd;
  if (!mat || !lut || !x || !zeropoint || !lo || !hi || !out_idx || !out_val || !out_count || H <= 0 || hd <= 0 ||
      hd % 32 || C > 8192 || col < 0 || col >= max_len)
    return KVQ_EINVAL;
  append_orig_kernel<false><<<1, kOrigThreads, 0, (hipStream_t)stream>>>(
      reinterpret_cast<uint32_t *>(mat), lut, x, zeropoint, lo, hi, 0.f, 0.f, 0.f, out_idx, out_val, out_count, C,
      max_len, col);
  return check_launch();
}

int kvq_append_v_sparse_orig(int32_t *mat, const float *lut_rows, const float *x, float zeropoint, float lo,
                             float hi, int32_t *out_idx, float *out_val, int32_t *out_count, int H, int hd,
                             int64_t max_len, int64_t col, void *stream) {
  const int C = H * hd;
  if (!mat || !lut_rows || !x || !out_idx || !out_val || !out_count || H <= 0 || hd <= 0 || hd % 32 || C > 8192 ||
      col < 0 || col >= max_len)
    return KVQ_EINVAL;
  append_orig_kernel<true><<<1, kOrigThreads, 0, (hipStream_t)stream>>>(
      reinterpret_cast<uint32_t *>(mat), lut_rows, x, nullptr, nullptr, nullptr, zeropoint, lo, hi, out_idx, out_val,
      out_count, C, max_len, col);
  return check_launch();
}

int kvq_spmv_k_rope_csr(const int32_t *rowptr, const int32_t *cols, const float *vals, const float *q, float *mul,
                        int64_t num_rows, int64_t L, int hd, float rope_theta, int pos_offset, void *stream) {
  if (!rowptr || !q || !mul || num_rows < 0 || L < 0 || hd != kHeadDim) return KVQ_EINVAL;
  if (num_rows == 0 || L == 0) return KVQ_OK;
  if (!cols || !vals) return KVQ_EINVAL;
  spmv_k_rope_csr_kernel<<<(unsigned)((num_rows + 255) / 256), 256, 0, (hipStream_t)stream>>>(
      rowptr, cols, vals, q, mul, num_rows, L, pos_offset, rope_theta);
  return check_launch();
}

int kvq_spmv_v_csc(const int32_t *colptr, const int32_t *rows, const float *vals, const float *p, float *mul,
                   int64_t num_cols, int64_t L, int H, int hd, void *stream) {
  if (!colptr || !p || !mul || num_cols < 0 || L < 0 || H <= 0 || hd != kHeadDim) return KVQ_EINVAL;
  if (num_cols == 0 || L == 0) return KVQ_OK;
  if (!rows || !vals) return KVQ_EINVAL;
  spmv_v_csc_kernel<<<(unsigned)((num_cols + 1023) / 1024), 1024, 0, (hipStream_t)stream>>>(
      colptr, rows, vals, p, mul, num_cols, L, H * hd);
  return check_launch();
}

}  // extern "C"
