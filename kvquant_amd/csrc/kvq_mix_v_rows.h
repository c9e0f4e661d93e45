// FALLBACK variant (any max_len alignment) of softmax(p).V -- see kvq_mix_v.hip for
// the main kernel.  softmax(p).V over the packed NUQ value cache (per-token codebooks), fused
// with the fixed-width sparse-outlier SpMV.  Reference semantics:
// KCU:3211-3433 (+4117-4491, 4998-5248), SPMV_ATOMIC_BALANCED KCU:437-470,
// launchers KCU:3491-3538, 3625-3690.
//
// CDNA4 design.  The reduction runs over tokens, so (unlike the reference, which
// puts a token on every thread and transposes 32 KB of products through LDS per
// block, then issues 128 atomics) a LANE OWNS A ROW UNIT of the cache -- one
// packed word-row of one head (8 / 16 channels for 4 / 2 bit, three rows = 32
// channels for 3 bit) -- and walks the token axis, which is the contiguous
// axis of that row: 16-byte loads, 64 B of a row per batch.  The 8..32 channel
// sums of a unit live in VGPRs for the whole token range: no cross-lane
// reduction, no transposes, no atomics.  All lanes of a wave decode the SAME
// token at the same time, so the per-token codebook row (64 B) is one
// conflict-free broadcast LDS read per code; rows are staged per 64-token chunk.
// A workgroup = 4 waves = 4 unit blocks (256 units) over one token range; every
// workgroup writes one slab of partial sums and a second tiny kernel adds the
// slabs into `mul` in a fixed order (deterministic, unlike atomics).
// The sparse residuals are handled by extra workgroups of the same launch that
// accumulate val*p into an LDS copy of the output vector (ds_add_f32).
// Algorithmic HBM bytes per cached token: C*bits/8 + 4*2^bits (codebook row)
// + 4*H (probabilities) (+ 8*n_out sparse).
#pragma once
#include "kvq_common.h"
#include "kvq_host.h"

namespace kvq {

constexpr int kChunk = 64;     // tokens per staged codebook chunk
constexpr int kMixWaves = 4;
constexpr int kSparseTokens = 2048;  // tokens per sparse workgroup

typedef uint32_t u32x4_u __attribute__((ext_vector_type(4), aligned(4)));
typedef float f32x4_u __attribute__((ext_vector_type(4), aligned(4)));

struct MixVArgs {
  const float *p;          // [q_len][H][L]
  const uint32_t *mat;     // [rows][max_len]
  const float *lut_rows;   // [max_len][N]
  const float *outliers;
  const int32_t *idx;
  float *partial;          // [slabs][q_len][C]
  int H;
  int q_len;
  int64_t L;
  int64_t max_len;
  int64_t tr;              // tokens per dense range (multiple of kChunk)
  int n_ranges;
  int ubg;                 // unit-block groups (workgroups per range)
  int n_units;
  int n_out;
};

template <int BITS, int I, int WORDS>
__device__ __forceinline__ unsigned unit_code(const uint32_t (&w)[WORDS]) {
  if constexpr (BITS == 3) {
    return code_of<3, I>(w);
  } else if constexpr (BITS == 4) {
    return (w[0] >> (4 * I)) & 0xfu;
  } else {
    return (w[0] >> (2 * I)) & 0x3u;
  }
}

template <int BITS>
__global__ __launch_bounds__(kMixWaves * 64) void mix_v_rows_kernel(MixVArgs a) {
  constexpr int N = Fmt<BITS>::kN;
  constexpr int WORDS = Unit<BITS>::kWords;
  constexpr int CH = Unit<BITS>::kCh;
  constexpr int NT = kMixWaves * 64;
  constexpr int kBatch = Unit<BITS>::kBatch;
  const int C = a.H * kHeadDim;
  const int b = blockIdx.z;
  const int n_dense = a.n_ranges * a.ubg;

  __shared__ __attribute__((aligned(16))) float lds[4096];  // 16 KB: codebook chunks / sparse accumulator

  if ((int)blockIdx.x >= n_dense) {
    // ---------------- sparse residual workgroup -------------------------------
    const int s = blockIdx.x - n_dense;
    float *slab = a.partial + ((int64_t)(a.n_ranges * a.ubg + s) * a.q_len + b) * C;
    // reference: the sparse part only sees batch 0 (KCU:3675)
    const bool active = (b == 0) && a.outliers != nullptr;
    for (int c0 = 0; c0 < C; c0 += 4096) {
      const int cn = (C - c0 < 4096) ? (C - c0) : 4096;
      for (int i = threadIdx.x; i < cn; i += NT) lds[i] = 0.f;
      __syncthreads();
      if (active) {
        const int64_t t0 = (int64_t)s * kSparseTokens;
        const int64_t t1 = (t0 + kSparseTokens < a.L) ? (t0 + kSparseTokens) : a.L;
        const int64_t e0 = t0 * a.n_out, e1 = t1 * a.n_out;
        for (int64_t e = e0 + threadIdx.x; e < e1; e += NT) {
          const float val = a.outliers[e];
          const int row = a.idx[e];
          if (row < c0 || row >= c0 + cn) continue;
          const int64_t t = e / a.n_out;
          const float pt = a.p[(int64_t)(row >> 7) * a.L + t];
          atomicAdd(&lds[row - c0], val * pt);
        }
      }
      __syncthreads();
      for (int i = threadIdx.x; i < cn; i += NT) slab[c0 + i] = lds[i];
      __syncthreads();
    }
    return;
  }

  // ---------------- dense workgroup ---------------------------------------------
  const int range = blockIdx.x / a.ubg;
  const int ub = (blockIdx.x % a.ubg) * kMixWaves + (threadIdx.x >> 6);
  const int u = ub * 64 + (threadIdx.x & 63);
  const bool uvalid = u < a.n_units;
  const int uc = uvalid ? u : a.n_units - 1;
  const int h = uc / Unit<BITS>::kPerHead;
  const int64_t t_begin = (int64_t)range * a.tr;
  const int64_t t_end = (t_begin + a.tr < a.L) ? (t_begin + a.tr) : a.L;
  const uint32_t *rowp = a.mat + (int64_t)uc * WORDS * a.max_len;
  const float *ph = a.p + ((int64_t)b * a.H + h) * a.L;

  float acc[CH];
#pragma unroll
  for (int i = 0; i < CH; i++) acc[i] = 0.f;

  for (int64_t c0 = t_begin; c0 < t_end; c0 += kChunk) {
    // stage the chunk's codebook rows: lds[tl*N + v]
    __syncthreads();
    for (int i = threadIdx.x; i < kChunk * N / 4; i += NT) {
      int64_t tok = c0 + (i * 4) / N;
      if (tok >= a.L) tok = a.L - 1;
      const float4 r = *reinterpret_cast<const float4 *>(a.lut_rows + tok * N + (i * 4) % N);
      *reinterpret_cast<float4 *>(&lds[i * 4]) = r;
    }
    __syncthreads();
#pragma unroll 1
    for (int bt = 0; bt < kChunk / kBatch; bt++) {
      const int64_t tb = c0 + bt * kBatch;
      if (tb >= t_end) break;
      u32x4_u wv[WORDS][kBatch / 4];
      f32x4_u pv[kBatch / 4];
      if (tb + kBatch <= t_end) {
#pragma unroll
        for (int j = 0; j < kBatch / 4; j++) {
#pragma unroll
          for (int wi = 0; wi < WORDS; wi++)
            wv[wi][j] = *reinterpret_cast<const u32x4_u *>(rowp + (int64_t)wi * a.max_len + tb + j * 4);
          pv[j] = *reinterpret_cast<const f32x4_u *>(ph + tb + j * 4);
        }
      } else {  // ragged tail: clamp the address, zero the probability
#pragma unroll
        for (int j = 0; j < kBatch / 4; j++) {
#pragma unroll
          for (int e = 0; e < 4; e++) {
            const int64_t tt = tb + j * 4 + e;
            const int64_t tc = tt < t_end ? tt : t_end - 1;
#pragma unroll
            for (int wi = 0; wi < WORDS; wi++) wv[wi][j][e] = rowp[(int64_t)wi * a.max_len + tc];
            pv[j][e] = tt < t_end ? ph[tc] : 0.f;
          }
        }
      }
      const float *tabb = lds + bt * kBatch * N;
#pragma unroll
      for (int tt = 0; tt < kBatch; tt++) {
        uint32_t w[WORDS];
#pragma unroll
        for (int wi = 0; wi < WORDS; wi++) w[wi] = wv[wi][tt / 4][tt % 4];
        const float pt = pv[tt / 4][tt % 4];
        const float *tab = tabb + tt * N;
        static_for<0, CH>([&](auto I) {
          constexpr int i = decltype(I)::value;
          acc[i] = fmaf(tab[unit_code<BITS, i, WORDS>(w)], pt, acc[i]);
        });
      }
    }
  }
  if (uvalid) {
    float *slab = a.partial + ((int64_t)blockIdx.x * a.q_len + b) * C + (int64_t)u * CH;
#pragma unroll
    for (int i = 0; i < CH; i += 4)
      *reinterpret_cast<float4 *>(slab + i) = make_float4(acc[i], acc[i + 1], acc[i + 2], acc[i + 3]);
  }
}

// mul[b][c] (+)= sum over slabs, fixed order
__global__ __launch_bounds__(256) void mix_v_rows_reduce_kernel(const float *__restrict__ partial,
                                                           float *__restrict__ mul, int n_ranges, int ubg,
                                                           int n_sparse, int q_len, int C, int units_ch,
                                                           int accumulate) {
  const int c = blockIdx.x * 256 + threadIdx.x;
  const int b = blockIdx.y;
  if (c >= C) return;
  // the dense slab of range r that holds channel c belongs to unit-block group g
  const int g = (c / units_ch) / (kMixWaves * 64);
  float s = accumulate ? mul[(int64_t)b * C + c] : 0.f;
  for (int r = 0; r < n_ranges; r++) s += partial[((int64_t)(r * ubg + g) * q_len + b) * C + c];
  for (int k = 0; k < n_sparse; k++) s += partial[((int64_t)(n_ranges * ubg + k) * q_len + b) * C + c];
  mul[(int64_t)b * C + c] = s;
}

struct MixPlan {
  int64_t tr;
  int n_ranges, ubg, n_units, n_sparse;
  size_t bytes;
};

static MixPlan plan_mix_rows(int bits, int q_len, int H, int64_t L, bool sparse) {
  MixPlan pl;
  const int ch = bits == 4 ? 8 : (bits == 3 ? 32 : 16);
  pl.n_units = H * kHeadDim / ch;
  const int ub = (pl.n_units + 63) / 64;
  pl.ubg = (ub + kMixWaves - 1) / kMixWaves;
  // ~1024 dense workgroups (4 per CU), ranges a multiple of the 64-token chunk
  int64_t want = 1024 / pl.ubg;
  if (want < 1) want = 1;
  int64_t tr = (L + want - 1) / want;
  tr = (tr + kChunk - 1) / kChunk * kChunk;
  if (tr < 4 * kChunk) tr = 4 * kChunk;
  pl.tr = tr;
  pl.n_ranges = (int)((L + tr - 1) / tr);
  if (pl.n_ranges < 1) pl.n_ranges = 1;
  pl.n_sparse = sparse ? (int)((L + kSparseTokens - 1) / kSparseTokens) : 0;
  pl.bytes = (size_t)(pl.n_ranges * pl.ubg + pl.n_sparse) * q_len * H * kHeadDim * sizeof(float);
  return pl;
}

template <int BITS>
static int launch_mix_rows(const MixVArgs &a, const MixPlan &pl, float *mul, int accumulate, hipStream_t st) {
  const int C = a.H * kHeadDim;
  dim3 grid(pl.n_ranges * pl.ubg + pl.n_sparse, 1, a.q_len), block(kMixWaves * 64);
  mix_v_rows_kernel<BITS><<<grid, block, 0, st>>>(a);
  int rc = check_launch();
  if (rc) return rc;
  dim3 rgrid((C + 255) / 256, a.q_len);
  mix_v_rows_reduce_kernel<<<rgrid, 256, 0, st>>>(a.partial, mul, pl.n_ranges, pl.ubg, pl.n_sparse, a.q_len, C,
                                             Unit<BITS>::kCh, accumulate);
  return check_launch();
}

}  // namespace kvq
