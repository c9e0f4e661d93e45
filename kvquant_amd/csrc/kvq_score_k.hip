// q.K^T over the packed NUQ key cache with RoPE applied on the fly to the
// dequantised pre-RoPE keys, fused with the fixed-width sparse-outlier SpMV.
// Reference semantics: KCU:3040-3209 (+3692-4115, 4747-4995) and
// SPMV_ATOMIC_ROPE_BALANCED KCU:473-521; launchers KCU:3437-3622.
//
// CDNA4 design (not the reference's thread-per-token/atomic layout):
//   * a workgroup owns a tile of T = 32*NWAVES cached tokens and a group of
//     heads.  The RoPE angles of a token depend only on (token, j) -- not on
//     the head -- so each lane evaluates the sincos of ITS token once per tile
//     and reuses it for every head of the group (the reference evaluates 256
//     transcendentals per (token, head));
//   * a wave is split in two 32-lane halves that serve the same 32 tokens:
//     half r handles rotation pairs j in [32r, 32r+32), i.e. channels
//     [32r,32r+32) and [64+32r, 64+32r+32).  That halves the trig registers
//     (64 VGPRs) and keeps the two halves in different LDS lane groups;
//   * per head the 128 x 2^bits codebook is staged in LDS PRE-MULTIPLIED by
//     the query: entry (k, v) = (L*q[k], sgn_k*L*q[(k+64)%128]).  One
//     conflict-free ds_read_b64 per code (all lanes of a 32-lane group read the
//     same channel k; the 2^bits distinct addresses fall into distinct banks)
//     yields both RoPE operands, so a code costs 2 address ops + 2 FMAs;
//   * packed words are read with lanes along the token axis (the contiguous
//     axis): 128 B per half-wave per row, prefetched one head ahead;
//   * the sparse residuals of the tile are scattered into an LDS score tile
//     with ds_add_f32 before the dense loop and folded into the single store of
//     each score (no global atomics; the reference does one per outlier).
// Algorithmic HBM bytes per cached token: C*bits/8 (+ 8*n_out sparse) + 4*H.
#include "kvq_common.h"
#include "kvq_host.h"

#include <cmath>

namespace kvq {

struct ScoreKArgs {
  const float *q;          // [q_len][H][128]
  const uint32_t *mat;     // [H][WPH][max_len]
  float *mul;              // [q_len][H][L]
  const float *lut;        // [H][128][N]
  const float *outliers;   // [max_len][n_out] or null
  const int32_t *idx;
  int H;
  int hpg;                 // heads per workgroup
  int64_t L;
  int64_t max_len;
  int pos_offset;
  int n_out;
  int accumulate;
};

typedef float f32x2 __attribute__((ext_vector_type(2)));

// LDS image of one head's codebook, pre-multiplied by the query.  Rotation pair i (0..31) of
// wave-half ("role") r covers channel k_lo = 32r+i and k_hi = 64+32r+i.  Entry order:
//   TLO[(i*2 + r)*N + code] = (L[k_lo][code]*q[k_lo],  L[k_lo][code]*q[k_lo+64])
//   THI[(i*2 + r)*N + code] = (L[k_hi][code]*q[k_hi], -L[k_hi][code]*q[k_hi-64])
// so the variable part of a look-up address is (r*N + code)*8 bytes -- for 4 bit that is ONE byte
// (role in bit 7, code*8 below) which a single v_bfe_u32 cuts out of a pre-masked word -- and the
// pair index i is an instruction immediate.
template <int BITS>
struct KTab {
  static constexpr int N = Fmt<BITS>::kN;
  static constexpr int HALF_B = 32 * 2 * N * 8;     // bytes of TLO (= THI)
  static constexpr int BUF_B = 2 * HALF_B;          // one head
};

template <int BITS>
__device__ __forceinline__ void stage_lutq(unsigned char *dst, const float *__restrict__ lut,
                                           const float *__restrict__ qh, int nthreads) {
  constexpr int N = Fmt<BITS>::kN;
  for (int e4 = threadIdx.x; e4 < kHeadDim * N / 4; e4 += nthreads) {
    const int e0 = e4 * 4;
    const int k = e0 / N, v0 = e0 % N;
    const float4 l4 = *reinterpret_cast<const float4 *>(lut + e0);
    const float qa = qh[k];
    const float qb = (k < 64) ? qh[k + 64] : -qh[k - 64];
    const int kk = k & 63;
    const int r = kk >> 5, i = kk & 31;
    float4 *d = reinterpret_cast<float4 *>(dst + (k >> 6) * KTab<BITS>::HALF_B + (((i * 2 + r) * N + v0) << 3));
    d[0] = make_float4(l4.x * qa, l4.x * qb, l4.y * qa, l4.y * qb);
    d[1] = make_float4(l4.z * qa, l4.z * qb, l4.w * qa, l4.w * qb);
  }
}

template <int BITS>
__device__ __forceinline__ void load_words(uint32_t (&w)[BITS], const uint32_t *__restrict__ mat,
                                           int64_t row0, int64_t max_len, int64_t t) {
#pragma unroll
  for (int i = 0; i < BITS; i++) w[i] = __builtin_nontemporal_load(mat + (row0 + i) * max_len + t);
}

constexpr int kSparseHpg = 16;   // heads per workgroup cap of the sparse variant (LDS budget)

template <int BITS, bool SPARSE, int NWAVES>
__global__ __launch_bounds__(NWAVES * 64) void score_k_kernel(ScoreKArgs a, RopeFreqs fr) {
  constexpr int N = Fmt<BITS>::kN;
  constexpr int WPH = Fmt<BITS>::kWordsPerHead;
  constexpr int T = NWAVES * 32;
  constexpr int NT = NWAVES * 64;
  constexpr int TAB_B = KTab<BITS>::BUF_B;
  constexpr int SC_B = SPARSE ? kSparseHpg * T * 4 : 16;
  constexpr int QL_B = SPARSE ? kSparseHpg * kHeadDim * 4 : 16;

  // static LDS: every table offset below is a compile-time constant that folds into ds immediates
  __shared__ __attribute__((aligned(16))) unsigned char smem[2 * TAB_B + 256 + SC_B + QL_B];
  unsigned char *lutq = smem;                                                    // [2][TAB_B]
  float *theta = reinterpret_cast<float *>(smem + 2 * TAB_B);                    // [64]
  float *sc = reinterpret_cast<float *>(smem + 2 * TAB_B + 256);                 // [hpg][T]
  float *ql = reinterpret_cast<float *>(smem + 2 * TAB_B + 256 + SC_B);          // [hpg][128]

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;
  const int role = lane >> 5;
  const int tl = wave * 32 + (lane & 31);
  const int64_t tile0 = (int64_t)blockIdx.x * T;
  const int64_t t = tile0 + tl;
  const bool valid = t < a.L;
  const int64_t tc = valid ? t : a.L - 1;
  const int h0 = blockIdx.y * a.hpg;
  const int b = blockIdx.z;
  const float *qb = a.q + (int64_t)b * a.H * kHeadDim;

  if (tid < 64) theta[tid] = fr.f[tid];
  if constexpr (SPARSE) {
    for (int i = tid; i < a.hpg * T; i += NT) sc[i] = 0.f;
    for (int i = tid; i < a.hpg * kHeadDim; i += NT) ql[i] = qb[h0 * kHeadDim + i];
  }
  stage_lutq<BITS>(lutq, a.lut + (int64_t)h0 * kHeadDim * N, qb + h0 * kHeadDim, NT);

  // first head's packed words (role r: channel groups r and 2+r)
  uint32_t wlo[BITS], whi[BITS];
  load_words<BITS>(wlo, a.mat, (int64_t)h0 * WPH + role * BITS, a.max_len, tc);
  load_words<BITS>(whi, a.mat, (int64_t)h0 * WPH + (2 + role) * BITS, a.max_len, tc);

  __syncthreads();

  // RoPE angles of this lane's token for its 32 rotation pairs (KCU:3083, 3122-3123)
  f32x2 cs[32];   // (cos, sin)
  const float posf = (float)((int)tc + a.pos_offset);
  static_for<0, 32>([&](auto I) {
    constexpr int i = decltype(I)::value;
    const float ang = theta[role * 32 + i] * posf;
    float sn, c;
    sincos_rev(ang, sn, c);
    cs[i].x = c;
    cs[i].y = sn;
  });

  if constexpr (SPARSE) {
    if (b == 0 && a.outliers != nullptr) {  // reference: sparse part ignores q_len > 1 (KCU:3605)
      const int ntok = (a.L - tile0 < T) ? (int)(a.L - tile0) : T;
      const unsigned nent = (unsigned)ntok * (unsigned)a.n_out;
      const float *ov = a.outliers + tile0 * a.n_out;
      const int32_t *oi = a.idx + tile0 * a.n_out;
      const int pos0 = (int)tile0 + a.pos_offset;
      for (unsigned e = tid; e < nent; e += NT) {
        const float val = ov[e];
        const int col = oi[e];
        const int hh = (col >> 7) - h0;
        if (val == 0.f || (unsigned)hh >= (unsigned)a.hpg) continue;   // capped-away slot / other head group
        const unsigned tle = e / (unsigned)a.n_out;
        const int ch = col & 127;
        const float ang = theta[ch & 63] * (float)(pos0 + (int)tle);
        float s, c;
        sincos_rev(ang, s, c);
        const float q1 = ql[hh * kHeadDim + ch];
        const float q2 = ql[hh * kHeadDim + ((ch + 64) & 127)];
        const float sg = (ch < 64) ? s : -s;
        atomicAdd(&sc[hh * T + tle], val * fmaf(c, q1, sg * q2));
      }
    }
  }

  // per-lane constant part of every look-up address
  const uint32_t rolepat = role ? 0x80808080u : 0u;     // 4 bit: role*128 in every byte
  const uint32_t rolebytes = (uint32_t)role * N * 8;    // generic

  auto head = [&](auto BUF, int hh) {
    constexpr int buf = decltype(BUF)::value;
    const int h = h0 + hh;
    __syncthreads();  // table `buf` staged (and sc complete); the other table is free
    uint32_t nlo[BITS], nhi[BITS];
    if (hh + 1 < a.hpg) {
      stage_lutq<BITS>(lutq + (1 - buf) * TAB_B, a.lut + (int64_t)(h + 1) * kHeadDim * N, qb + (h + 1) * kHeadDim, NT);
      load_words<BITS>(nlo, a.mat, (int64_t)(h + 1) * WPH + role * BITS, a.max_len, tc);
      load_words<BITS>(nhi, a.mat, (int64_t)(h + 1) * WPH + (2 + role) * BITS, a.max_len, tc);
    }
    const unsigned char *tlo = lutq + buf * TAB_B;
    const unsigned char *thi = lutq + buf * TAB_B + KTab<BITS>::HALF_B;
    f32x2 acc = {0.f, 0.f};
    if constexpr (BITS == 4) {
      static_for<0, 4>([&](auto J) {
        constexpr int j = decltype(J)::value;
        // even / odd nibbles as bytes = role*128 + code*8
        const uint32_t elo = ((wlo[j] << 3) & 0x78787878u) | rolepat, olo = ((wlo[j] >> 1) & 0x78787878u) | rolepat;
        const uint32_t ehi = ((whi[j] << 3) & 0x78787878u) | rolepat, ohi = ((whi[j] >> 1) & 0x78787878u) | rolepat;
        static_for<0, 8>([&](auto NN) {
          constexpr int n = decltype(NN)::value;
          constexpr int i = 8 * j + n;
          const uint32_t fl = (((n & 1) ? olo : elo) >> (8 * (n / 2))) & 0xffu;
          const uint32_t fh = (((n & 1) ? ohi : ehi) >> (8 * (n / 2))) & 0xffu;
          const f32x2 lo = *reinterpret_cast<const f32x2 *>(tlo + i * 2 * N * 8 + fl);
          const f32x2 hi = *reinterpret_cast<const f32x2 *>(thi + i * 2 * N * 8 + fh);
          acc = __builtin_elementwise_fma(cs[i], lo, acc);
          acc = __builtin_elementwise_fma(cs[i], hi, acc);
        });
      });
    } else {
      static_for<0, 32>([&](auto I) {
        constexpr int i = decltype(I)::value;
        const uint32_t fl = (code_of<BITS, i>(wlo) << 3) + rolebytes;
        const uint32_t fh = (code_of<BITS, i>(whi) << 3) + rolebytes;
        const f32x2 lo = *reinterpret_cast<const f32x2 *>(tlo + i * 2 * N * 8 + fl);
        const f32x2 hi = *reinterpret_cast<const f32x2 *>(thi + i * 2 * N * 8 + fh);
        acc = __builtin_elementwise_fma(cs[i], lo, acc);
        acc = __builtin_elementwise_fma(cs[i], hi, acc);
      });
    }
    float res = acc.x + acc.y;
    res += __shfl_xor(res, 32);
    if (role == 0 && valid) {
      if constexpr (SPARSE) res += sc[hh * T + tl];
      float *dst = a.mul + ((int64_t)b * a.H + h) * a.L + t;
      if (a.accumulate) res += *dst;
      *dst = res;
    }
    if (hh + 1 < a.hpg) {
#pragma unroll
      for (int i = 0; i < BITS; i++) {
        wlo[i] = nlo[i];
        whi[i] = nhi[i];
      }
    }
  };
  for (int hh = 0; hh < a.hpg; hh += 2) {
    head(std::integral_constant<int, 0>{}, hh);
    if (hh + 1 < a.hpg) head(std::integral_constant<int, 1>{}, hh + 1);
  }
}

// theta_j = powf(rope_theta, -2j/128) (KCU:3083): correctly rounded from double
// (the oracle uses the same definition).
static RopeFreqs make_freqs(float rope_theta) {
  RopeFreqs fr;
  for (int j = 0; j < kHeadDim / 2; j++) {
    float e = -2.0f * (float)j / (float)kHeadDim;
    fr.f[j] = (float)std::pow((double)rope_theta, (double)e);
  }
  return fr;
}

static int pick_groups(int H, int64_t tiles, int max_hpg) {
  // enough workgroups to fill 256 CUs twice over, heads per group as large as possible (trig
  // amortisation) but <= max_hpg; groups must divide H.
  int64_t want = (512 + tiles - 1) / tiles;
  int g = H;
  for (int d = 1; d <= H; d++)
    if (H % d == 0 && H / d <= max_hpg && d >= want) {
      g = d;
      break;
    }
  return g;
}

template <int BITS, bool SPARSE, int NWAVES>
static int launch_score(const ScoreKArgs &a0, int q_len, float rope_theta, hipStream_t st) {
  constexpr int T = NWAVES * 32;
  ScoreKArgs a = a0;
  const int64_t tiles = (a.L + T - 1) / T;
  const int groups = pick_groups(a.H, tiles, SPARSE ? kSparseHpg : 1 << 30);
  a.hpg = a.H / groups;
  dim3 grid((unsigned)tiles, groups, q_len), block(NWAVES * 64);
  score_k_kernel<BITS, SPARSE, NWAVES><<<grid, block, 0, st>>>(a, make_freqs(rope_theta));
  return check_launch();
}

template <int BITS>
static int dispatch_score(const ScoreKArgs &a, int q_len, float theta, bool sparse, hipStream_t st) {
  // big tiles (8 waves) once there are enough of them, small tiles for short caches
  if (a.L >= 16384) {
    return sparse ? launch_score<BITS, true, 8>(a, q_len, theta, st) : launch_score<BITS, false, 8>(a, q_len, theta, st);
  }
  return sparse ? launch_score<BITS, true, 4>(a, q_len, theta, st) : launch_score<BITS, false, 4>(a, q_len, theta, st);
}

}  // namespace kvq

using namespace kvq;

extern "C" int kvq_score_k(int bits, const float *q, const int32_t *mat, float *mul, const float *lut,
                           int q_len, int H, int hd, int64_t L, int64_t max_len, float rope_theta,
                           int pos_offset, const float *outliers, const int32_t *outlier_idx, int n_out,
                           int accumulate, void *stream) {
  if (!q || !mat || !mul || !lut || q_len <= 0 || H <= 0 || hd != kHeadDim || L < 0 || L > max_len)
    return KVQ_EINVAL;
  const bool sparse = outliers != nullptr;
  if (sparse && (!outlier_idx || n_out <= 0)) return KVQ_EINVAL;
  if (L == 0) return KVQ_OK;
  ScoreKArgs a;
  a.q = q;
  a.mat = reinterpret_cast<const uint32_t *>(mat);
  a.mul = mul;
  a.lut = lut;
  a.outliers = outliers;
  a.idx = outlier_idx;
  a.H = H;
  a.hpg = H;
  a.L = L;
  a.max_len = max_len;
  a.pos_offset = pos_offset;
  a.n_out = n_out;
  a.accumulate = accumulate;
  hipStream_t st = (hipStream_t)stream;
  switch (bits) {
    case 4: return dispatch_score<4>(a, q_len, rope_theta, sparse, st);
    case 3: return dispatch_score<3>(a, q_len, rope_theta, sparse, st);
    case 2: return dispatch_score<2>(a, q_len, rope_theta, sparse, st);
    default: return KVQ_EINVAL;
  }
}
