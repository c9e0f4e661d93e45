// NUQ quantize-and-pack kernels: one token (decode append) and S tokens
// (prefill pack).  Reference semantics: KCU:1167-3038 (see include/kvq.h).
//
// These are tiny (decode) or purely streaming (prefill) kernels:
//   append:  one lane per channel, 256 channels per workgroup; codes meet in
//            LDS and 8 lanes assemble the group's words (no atomics: the
//            reference OR-s nibbles with atomicAdd, KCU:1244).
//   pack:    one lane per token (coalesced along the token axis, which is the
//            contiguous axis of both the [H][hd][S] input and the cache), one
//            32-channel group per workgroup row; the per-channel K LUT row is
//            wave-uniform and lives in SGPRs, the per-token V LUT row is
//            lane-private in VGPRs.  HBM-bound: reads 4*C*S (+ writes 4*C*S for
//            K's rescaled output) and writes C*bits/8*S bytes.
#include "kvq_common.h"
#include "kvq_host.h"

namespace kvq {

enum AppendMode { K_DENSE = 0, K_SPARSE = 1, V_DENSE = 2, V_SPARSE = 3 };

template <int BITS, int MODE>
__global__ __launch_bounds__(256) void append_token_kernel(
    uint32_t *__restrict__ mat, const float *__restrict__ lut, const float *__restrict__ x,
    float *__restrict__ rescaled, const float *__restrict__ lo_c, const float *__restrict__ hi_c,
    float lo_s, float hi_s, int C, int64_t max_len, int64_t col) {
  constexpr int N = Fmt<BITS>::kN;
  __shared__ unsigned codes[256];
  const int c = blockIdx.x * 256 + threadIdx.x;
  if (c < C) {
    float row[N];
    const float *src = (MODE == K_DENSE || MODE == K_SPARSE) ? lut + (int64_t)c * N : lut + col * N;
#pragma unroll
    for (int v = 0; v < N; v += (N >= 4 ? 4 : N)) {
      if constexpr (N >= 4) {
        float4 t = *reinterpret_cast<const float4 *>(src + v);
        row[v] = t.x; row[v + 1] = t.y; row[v + 2] = t.z; row[v + 3] = t.w;
      }
    }
    const float xv = x[c];
    unsigned code;
    if constexpr (MODE == V_SPARSE) {
      code = (xv < lo_s || xv > hi_s) ? Fmt<BITS>::kZeroCode : nearest_code<N>(row, xv);
    } else {
      code = nearest_code<N>(row, xv);
    }
    if constexpr (MODE == K_SPARSE) {
      const float lo = lo_c[c], hi = hi_c[c];
      const float rangeval = (hi - lo) / 2;   // KCU:1759-1764
      const float zeropoint = (hi + lo) / 2;
      rescaled[c] = (xv - zeropoint) / rangeval;
    }
    codes[threadIdx.x] = code;
  }
  __syncthreads();
  if (threadIdx.x < 8) {
    const int g = blockIdx.x * 8 + threadIdx.x;  // 32-channel group
    if (g * 32 < C) {
      unsigned cd[32];
#pragma unroll
      for (int i = 0; i < 32; i++) cd[i] = codes[threadIdx.x * 32 + i];
      uint32_t w[BITS];
      pack32<BITS>(cd, w);
#pragma unroll
      for (int i = 0; i < BITS; i++) mat[((int64_t)g * BITS + i) * max_len + col] = w[i];
    }
  }
}

template <int BITS, bool IS_K>
__global__ __launch_bounds__(256) void pack_parallel_kernel(
    uint32_t *__restrict__ mat, const float *__restrict__ lut, const float *__restrict__ x,
    float *__restrict__ rescaled, const float *__restrict__ lo, const float *__restrict__ hi,
    int64_t S, int64_t max_len, int64_t col0) {
  constexpr int N = Fmt<BITS>::kN;
  const int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x;
  const int g = blockIdx.y;  // 32-channel group
  if (t >= S) return;
  unsigned cd[32];
  if constexpr (IS_K) {
#pragma unroll 4
    for (int i = 0; i < 32; i++) {
      const int c = g * 32 + i;  // wave-uniform -> scalar loads
      float row[N];
#pragma unroll
      for (int v = 0; v < N; v++) row[v] = lut[(int64_t)c * N + v];
      const float l = lo[c], h = hi[c];
      const float rangeval = (h - l) / 2;
      const float zeropoint = (h + l) / 2;
      const float xv = x[(int64_t)c * S + t];
      if (rescaled != nullptr) rescaled[(int64_t)c * S + t] = (xv - zeropoint) / rangeval;   // (null: codes only)
      cd[i] = nearest_code<N>(row, xv);
    }
  } else {
    float row[N];
    const float *src = lut + (col0 + t) * N;
    if constexpr (N >= 4) {
#pragma unroll
      for (int v = 0; v < N; v += 4) {
        float4 q4 = *reinterpret_cast<const float4 *>(src + v);
        row[v] = q4.x; row[v + 1] = q4.y; row[v + 2] = q4.z; row[v + 3] = q4.w;
      }
    }
    const float l = lo[t], h = hi[t];
#pragma unroll 8
    for (int i = 0; i < 32; i++) {
      const float xv = x[(int64_t)(g * 32 + i) * S + t];
      cd[i] = (xv < l || xv > h) ? Fmt<BITS>::kZeroCode : nearest_code<N>(row, xv);
    }
  }
  uint32_t w[BITS];
  pack32<BITS>(cd, w);
#pragma unroll
  for (int i = 0; i < BITS; i++) mat[((int64_t)g * BITS + i) * max_len + col0 + t] = w[i];
}

template <int MODE>
static int launch_append(int bits, int32_t *mat, const float *lut, const float *x, float *rescaled,
                         const float *lo_c, const float *hi_c, float lo_s, float hi_s, int H, int hd,
                         int64_t max_len, int64_t col, hipStream_t st) {
  if (!mat || !lut || !x || H <= 0 || hd <= 0 || hd % 32 || col < 0 || col >= max_len) return KVQ_EINVAL;
  if (MODE == K_SPARSE && (!rescaled || !lo_c || !hi_c)) return KVQ_EINVAL;
  const int C = H * hd;
  dim3 grid((C + 255) / 256), block(256);
  auto m = reinterpret_cast<uint32_t *>(mat);
  switch (bits) {
    case 4: append_token_kernel<4, MODE><<<grid, block, 0, st>>>(m, lut, x, rescaled, lo_c, hi_c, lo_s, hi_s, C, max_len, col); break;
    case 3: append_token_kernel<3, MODE><<<grid, block, 0, st>>>(m, lut, x, rescaled, lo_c, hi_c, lo_s, hi_s, C, max_len, col); break;
    case 2: append_token_kernel<2, MODE><<<grid, block, 0, st>>>(m, lut, x, rescaled, lo_c, hi_c, lo_s, hi_s, C, max_len, col); break;
    default: return KVQ_EINVAL;
  }
  return check_launch();
}

template <bool IS_K>
static int launch_pack(int bits, int32_t *mat, const float *lut, const float *x, float *rescaled,
                       const float *lo, const float *hi, int H, int hd, int64_t S, int64_t max_len,
                       int64_t col0, hipStream_t st) {
  if (!mat || !lut || !x || !lo || !hi || H <= 0 || hd <= 0 || hd % 32 || S < 0 || col0 < 0 ||
      col0 + S > max_len)
    return KVQ_EINVAL;
  if (S == 0) return KVQ_OK;
  const int C = H * hd;
  dim3 grid((unsigned)((S + 255) / 256), C / 32), block(256);
  auto m = reinterpret_cast<uint32_t *>(mat);
  switch (bits) {
    case 4: pack_parallel_kernel<4, IS_K><<<grid, block, 0, st>>>(m, lut, x, rescaled, lo, hi, S, max_len, col0); break;
    case 3: pack_parallel_kernel<3, IS_K><<<grid, block, 0, st>>>(m, lut, x, rescaled, lo, hi, S, max_len, col0); break;
    case 2: pack_parallel_kernel<2, IS_K><<<grid, block, 0, st>>>(m, lut, x, rescaled, lo, hi, S, max_len, col0); break;
    default: return KVQ_EINVAL;
  }
  return check_launch();
}

// codes + pack of S prompt tokens of K without the rescaled output (the fused prefill pack selects the outliers in
// its own kernel and leaves the per-channel codebook search to this streaming one: lane per token, the channel's
// codebook row wave-uniform in SGPRs, i.e. read once per 256 tokens instead of once per token)
int pack_k_codes(int bits, int32_t *mat, const float *lut, const float *x, const float *lo, const float *hi, int H,
                 int hd, int64_t S, int64_t max_len, int64_t col0, hipStream_t st) {
  return launch_pack<true>(bits, mat, lut, x, nullptr, lo, hi, H, hd, S, max_len, col0, st);
}

}  // namespace kvq

using namespace kvq;

extern "C" {

int kvq_append_k(int bits, int32_t *mat, const float *lut, const float *x, int H, int hd,
                 int64_t max_len, int64_t col, void *stream) {
  return launch_append<K_DENSE>(bits, mat, lut, x, nullptr, nullptr, nullptr, 0.f, 0.f, H, hd, max_len, col,
                                (hipStream_t)stream);
}

int kvq_append_v(int bits, int32_t *mat, const float *lut_rows, const float *x, int H, int hd,
                 int64_t max_len, int64_t col, void *stream) {
  return launch_append<V_DENSE>(bits, mat, lut_rows, x, nullptr, nullptr, nullptr, 0.f, 0.f, H, hd, max_len,
                                col, (hipStream_t)stream);
}

int kvq_append_k_sparse(int bits, int32_t *mat, const float *lut, const float *x, float *rescaled,
                        const float *lo, const float *hi, int H, int hd, int64_t max_len, int64_t col,
                        void *stream) {
  return launch_append<K_SPARSE>(bits, mat, lut, x, rescaled, lo, hi, 0.f, 0.f, H, hd, max_len, col,
                                 (hipStream_t)stream);
}

int kvq_append_v_sparse(int bits, int32_t *mat, const float *lut_rows, const float *x, float lo, float hi,
                        int H, int hd, int64_t max_len, int64_t col, void *stream) {
  return launch_append<V_SPARSE>(bits, mat, lut_rows, x, nullptr, nullptr, nullptr, lo, hi, H, hd, max_len,
                                 col, (hipStream_t)stream);
}

int kvq_pack_k_sparse_parallel(int bits, int32_t *mat, const float *lut, const float *x, float *rescaled,
                               const float *lo, const float *hi, int H, int hd, int64_t S, int64_t max_len,
                               int64_t col0, void *stream) {
  if (!rescaled) return KVQ_EINVAL;
  return launch_pack<true>(bits, mat, lut, x, rescaled, lo, hi, H, hd, S, max_len, col0, (hipStream_t)stream);
}

int kvq_pack_v_sparse_parallel(int bits, int32_t *mat, const float *lut_rows, const float *x,
                               const float *lo, const float *hi, int H, int hd, int64_t S, int64_t max_len,
                               int64_t col0, void *stream) {
  return launch_pack<false>(bits, mat, lut_rows, x, nullptr, lo, hi, H, hd, S, max_len, col0,
                            (hipStream_t)stream);
}

}  // extern "C"
