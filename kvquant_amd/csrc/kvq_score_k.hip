// q.K^T over the packed NUQ key cache with RoPE applied on the fly to the
// dequantised pre-RoPE keys, fused with the fixed-width sparse-outlier SpMV.
// Reference semantics: KCU:3040-3209 (+3692-4115, 4747-4995) and
// SPMV_ATOMIC_ROPE_BALANCED KCU:473-521; launchers KCU:3437-3622.
//
// CDNA4 design (not the reference's thread-per-token/atomic layout):
//   * a workgroup owns a tile of T = 32*NWAVES cached tokens and a group of
//     heads.  The RoPE angles of a token depend only on (token, j) -- not on
//     the head -- so each lane evaluates the sincos of ITS token once per tile
//     (hardware v_sin/v_cos after an exact hi/lo range reduction) and reuses it
//     for every head of the group; the reference evaluates 256 transcendentals
//     per (token, head);
//   * a wave is split in two 32-lane halves that serve the same 32 tokens:
//     half r handles rotation pairs i in [32r, 32r+32), i.e. channels
//     [32r,32r+32) and [64+32r, 64+32r+32).  That halves the trig registers
//     (64 VGPRs) and keeps the two halves in different LDS lane groups;
//   * the per-head codebook, PRE-MULTIPLIED by the query, is built once per
//     call by a tiny kernel (entry = (L*q[k], sgn_k*L*q[(k+64)%128])) as the
//     exact LDS image and copied per head with LDS-DMA one head ahead of the
//     math (no VALU, no registers, no ds_write).  One conflict-free
//     ds_read_b64 per code yields both RoPE operands and feeds one packed FMA
//     against the lane's (cos, sin) pair.  The variable part of a look-up
//     address is one byte (role*128 + code*8) cut out of a pre-masked word by
//     a single v_bfe_u32; everything else is an instruction immediate (static
//     LDS);
//   * packed words are read with lanes along the token axis (the contiguous
//     axis): 128 B per half-wave per row, one head ahead of the math;
//   * sparse residuals: the tile's scores collect in an LDS [token][head] tile
//     (dense results and residuals alike) that is written out once -- no
//     global atomics, one store per score (the reference does one atomic per
//     outlier and one per score).  The residual work is spread over the head
//     loop (one piece per head iteration, its loads one iteration ahead).
//     Reference row layout [max_len][n_out]: 64-lane chunks of the wave's own
//     tokens' entries, segmented wave scan over (token, head) runs.  Token-
//     contiguous mirror [n_out][max_len] (kvquant_amd's own cache): a lane
//     owns its token's entries, no scan;
//   * the first softmax pass (max, sum exp of the fp16-scaled scores per head
//     and tile) can be taken from the LDS tile before it is written out;
//   * the ragged last tile is cut into short head-group workgroups so that it
//     is a tail, not a second round of the grid.
// Algorithmic HBM bytes per cached token: C*bits/8 (+ 8*n_out sparse) + 4*H.
#include "kvq_common.h"
#include "kvq_host.h"
#include "kvq_ktab.h"

#include "kvq_score_k_tile.h"

namespace kvq {

template <int BITS, bool SPARSE, int NWAVES, bool TRANSPOSED = false, bool COMPACT = false, int PAIR = 0>
__global__ __launch_bounds__(NWAVES * 64, 4)
void score_k_kernel(ScoreKArgs a) {
  using G = KGeom<BITS, SPARSE, NWAVES, TRANSPOSED, PAIR>;
  constexpr int T = G::T, NT = G::NT, SCS = G::SCS;
  // static LDS: every table offset in the tile body is a compile-time constant that folds into ds immediates
  __shared__ __attribute__((aligned(16))) unsigned char smem[G::SMEM_B];
  const KTile kt = score_k_tile<BITS, SPARSE, NWAVES, TRANSPOSED, COMPACT, PAIR>(a, smem);
  if constexpr (SPARSE) {
    float *sc = reinterpret_cast<float *>(smem + G::SC_OFF);
    const int tid = threadIdx.x;
    const int nh = kt.nh, h0 = kt.h0, tl = kt.tl, role = kt.role, b = kt.b, ntok = kt.ntok, tile_i = kt.tile_i;
    const int64_t t = kt.t;
    const bool valid = kt.valid;
    // write the tile out: the two wave halves take alternate heads, 128 B per half-wave per head
    if (valid) {
      for (int hh = role; hh < nh; hh += 2) {
        float res = sc[tl * SCS + ((hh + tl) & (SCS - 1))];
        float *dst = a.mul + ((int64_t)b * a.H + h0 + hh) * a.L + t;
        if (a.accumulate) res += *dst;
        __builtin_nontemporal_store(res, dst);
      }
    }
    // First softmax pass on the tile while it is still in LDS (saves a launch and a 4*H*L-byte sweep):
    // NT/32 lanes per head read the head's column (tokens strided by NT/32), reduce (max, sum of exp) of
    // the scaled scores among themselves, and lane 0 of each head writes the tile's partial.
    if (a.sm_parts != nullptr) {
      __syncthreads();   // a wave wrote only its own tokens' rows
      // (a partial covers PT = 256 tokens -- the unit kvq_score_k_softmax_parts counts in: the 512-token tile of the 16-wave
      //  variant writes two, lanes [0, 16) and [16, 32) of a head's 32 taking a half each)
      constexpr int TPH = NT / 32;                       // lanes per head
      constexpr int PT = T > 256 ? 256 : T;              // tokens per partial
      constexpr int NP = T / PT;                         // partials per tile
      constexpr int LPP = TPH / NP;                      // lanes per partial
      const int hh = tid / TPH, r = tid % TPH;
      const int half = r / LPP, rr = r % LPP;
      float x[PT / LPP];
      float m = -INFINITY, sm = 0.f;
#pragma unroll
      for (int k = 0; k < PT / LPP; k++) {
        const int j = half * PT + rr + k * LPP;
        x[k] = (j < ntok && hh < nh) ? scaled(sc[j * SCS + ((hh + j) & (SCS - 1))], a.sm_inv) : -INFINITY;
        m = fmaxf(m, x[k]);
      }
      // (hardware 2^x: these sums only feed the row normaliser Z, where a few ulp are far below the fp16
      // rounding of the probabilities; the per-element exponentials of the second pass stay exact.  The
      // accurate expf was ~5 % of the kernel: 24 calls per lane per tile.)
      constexpr float kLog2e = 1.4426950408889634f;
      auto ex = [&](float d) { return __builtin_amdgcn_exp2f(d * kLog2e); };   // d <= 0; 2^-inf = 0
      if (m > -INFINITY) {
#pragma unroll
        for (int k = 0; k < PT / LPP; k++) sm += ex(x[k] - m);     // 0 for the padding
      }
#pragma unroll
      for (int d = LPP / 2; d >= 1; d >>= 1) {
        const float mo = __shfl_xor(m, d), so = __shfl_xor(sm, d);
        const float mn = fmaxf(m, mo);
        sm = (mn == -INFINITY) ? 0.f : sm * ex(m - mn) + so * ex(mo - mn);
        m = mn;
      }
      const int part = tile_i * NP + half;
      if (rr == 0 && hh < nh && part < a.sm_nparts) {
        float *dst = a.sm_parts + ((int64_t)(h0 + hh) * a.sm_nparts + part) * 2;
        dst[0] = m;
        dst[1] = sm;
      }
    }
  }
#if KVQ_TRACE
  {   // exit stamp of the wave (slot 3 of the kernel-level timeline, kvq_score_k_tile.h)
    unsigned long long rr;
    asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(rr)::"memory");
    const int wave = threadIdx.x >> 6;
    if ((threadIdx.x & 63) == 0 && blockIdx.x < 1024)
      a.trace[(int64_t)1024 * 8 * 32 * 8 + ((int64_t)blockIdx.x * NWAVES + wave) * 8 + 3] = rr;
  }
#endif
}

static int cu_count() {
  static int cus = 0;
  if (cus == 0) {
    int dev = 0, n = 0;
    if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n <= 0)
      n = 256;
    cus = n;
  }
  return cus;
}

// workgroups of one kernel variant that are resident at a time (registers AND LDS: the 4-wave sparse variants hold
// 64 KB of LDS, two per CU, not the four their wave count would allow)
template <typename K>
static int workgroup_slots(K kernel, int threads, int fallback_per_cu) {
  int per_cu = 0;
  if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, kernel, threads, 0) != hipSuccess || per_cu <= 0)
    per_cu = fallback_per_cu;
  (void)hipGetLastError();
  return cu_count() * per_cu;
}

// Head groups per full tile.  A workgroup of hpg heads costs ~(hpg + 3.2) units with two workgroups per CU: a head
// iteration is 2.3 us, the tile's trig + prologue + epilogue 7.3 us (profiles/r03_a_traces.txt); alone on its CU it takes
// 3/4 of that.  `slots` workgroups run side by side; full generations cost a workgroup each, the last partial one
// between 3/4 (every CU holds at most one) and a whole workgroup, and every generation ~1.5 units of ramp.
// Fitted to the launch times of 2K ... 256K caches (profiles/r03_b_planner.txt); the round-counting model it replaces
// cut a 96K cache into 1536 workgroups of 8 heads (97 us, now 80) and ran 160K as two rounds of 32 heads (162 us, now
// 131).  Pick the divisor of H with the shortest makespan; ties go to the larger group (better trig amortisation).
static int pick_groups(int H, int64_t tiles, int q_len, int max_hpg, int slots) {
  int best = H;
  double best_cost = 1e30;
  for (int d = 1; d <= H; d++) {
    if (H % d || H / d > max_hpg) continue;
    const int64_t n = tiles * d * q_len;
    const double wg = H / d + 3.2;
    const int64_t full = n / slots, r = n % slots;
    // (partial generation: 3/4 of a workgroup while no CU holds two of them, rising to a whole one as the CUs fill up)
    const double part = 2 * r <= slots ? 0.75 : 0.75 + 0.25 * (double)(2 * r - slots) / slots;
    const double cost = (double)full * wg + (r ? wg * part : 0.0) + 1.5 * (double)(full + (r ? 1 : 0));
    if (cost < best_cost - 1e-9) {
      best_cost = cost;
      best = d;
    }
  }
  return best;
}

template <int BITS, bool SPARSE, int NWAVES, bool TRANSPOSED = false, bool COMPACT = false, int PAIR = 0>
static int launch_score(const ScoreKArgs &a0, int q_len, float rope_theta, hipStream_t st) {
  constexpr int T = NWAVES * 32;
  ScoreKArgs a = a0;
  const int64_t full_tiles = a.L / T;
  const int rem = (int)(a.L % T);
  const int max_hpg = SPARSE ? kSparseHpg : 1 << 30;
  static const int slots = workgroup_slots(score_k_kernel<BITS, SPARSE, NWAVES, TRANSPOSED, COMPACT, PAIR>, NWAVES * 64, 16 / NWAVES);
  a.groups = full_tiles ? pick_groups(a.H, full_tiles, q_len, max_hpg, slots) : 1;
  a.hpg = a.H / a.groups;
  if (full_tiles && a.hpg > max_hpg) return KVQ_EINVAL;   // (no full tile: only the ragged one's head groups exist)
  if (!full_tiles && a.hpg > max_hpg) a.hpg = max_hpg;     // (unused, kept in range)
  a.full_blocks = (int)(full_tiles * a.groups);
  // ragged last tile: few heads per workgroup, so that it is a short tail rather than an extra round
#ifndef KVQ_HPG_TAIL
#define KVQ_HPG_TAIL 1   // measured at 128K + 1 (mirror variant, in bench.py): 1 -> 86.4, 2 -> 87.2, 4 -> 88.6, 8 -> 93.2 us
#endif
  a.hpg_tail = full_tiles ? (SPARSE ? KVQ_HPG_TAIL : 2) : a.H / pick_groups(a.H, 1, q_len, max_hpg, slots);
  if (a.hpg_tail > a.H) a.hpg_tail = a.H;
  const int tail_blocks = rem ? (a.H + a.hpg_tail - 1) / a.hpg_tail : 0;
  dim3 grid((unsigned)(a.full_blocks + tail_blocks), 1, q_len), block(NWAVES * 64);
  constexpr int PT = T > 256 ? 256 : T;        // tokens per softmax partial
  if (a.sm_parts != nullptr && (!SPARSE || a.sm_nparts != (int)((a.L + PT - 1) / PT))) return KVQ_EINVAL;
  a.rope_theta = rope_theta;
#if KVQ_TRACE
  a.trace = reinterpret_cast<unsigned long long *>(strtoull(getenv("KVQ_TRACE_PTR") ? getenv("KVQ_TRACE_PTR") : "0", nullptr, 0));
  if (!a.trace) return KVQ_EINVAL;
#endif
  score_k_kernel<BITS, SPARSE, NWAVES, TRANSPOSED, COMPACT, PAIR><<<grid, block, 0, st>>>(a);
  return check_launch();
}

// cached tokens from which the sparse score kernels take 256-token tiles (8 waves); below: 128-token tiles (4 waves).
// kvq_score_k_softmax_parts counts the same tiles.  KVQ_SCORE_T8_FROM: A/B runs.
static int64_t big_tiles_from() {
  static const int64_t v = [] {
    const char *e = getenv("KVQ_SCORE_T8_FROM");
    return e ? (int64_t)atoll(e) : (int64_t)16384;
  }();
  return v;
}

template <int BITS>
static int dispatch_score(ScoreKArgs a, const float *lut, const void *q_in, int q_is_half, int tables_ready,
                          void *ws, int q_len, float theta, bool sparse, hipStream_t st) {
  unsigned char *tab = reinterpret_cast<unsigned char *>(ws);
  float *q32 = reinterpret_cast<float *>(tab + (size_t)q_len * a.H * KTab<BITS>::BUF_B);
  if (!tables_ready) {
    lutq_prep_kernel<BITS><<<dim3(a.H, q_len), 256, 0, st>>>(
        lut, q_in, q_is_half, tab, q32, KTabHasPair<BITS>::value ? tab + ktab_pair_offset<BITS>(q_len, a.H) : nullptr, a.H,
        a.pair == 2 ? 2 : 1);
    int rc = check_launch();
    if (rc) return rc;
  }
  a.tab = tab;
  a.tab_pair = KTabHasPair<BITS>::value ? tab + ktab_pair_offset<BITS>(q_len, a.H) : nullptr;
  a.q = q32;   // fp32 copy made by the prep (the sparse phase reads q directly)
  if constexpr (BITS == 3) {
    // fp32 pair-sum tables (round 6; decode, q_len = 1, mirror formats, >= 16K tokens): exact, half the look-ups, one
    // 1024-lane workgroup per CU (kvq_ktab.h: KTabPair32); shorter caches take the per-channel tables, which are always built
    if (a.pair == 2 && q_len == 1 && a.idx_t != nullptr && a.L >= 16384) {
      return a.out_t == nullptr ? launch_score<3, true, 16, true, true, 2>(a, q_len, theta, st)
                                : launch_score<3, true, 16, true, false, 2>(a, q_len, theta, st);
    }
    // fp16 pair-sum tables (decode, q_len = 1, mirror formats): half the look-ups (kvq_ktab.h: KTabPair3)
    if (a.pair == 1 && q_len == 1 && a.idx_t != nullptr) {
      if (a.out_t == nullptr)
        return a.L >= 16384 ? launch_score<3, true, 8, true, true, 1>(a, q_len, theta, st)
                            : launch_score<3, true, 4, true, true, 1>(a, q_len, theta, st);
      return a.L >= 16384 ? launch_score<3, true, 8, true, false, 1>(a, q_len, theta, st)
                          : launch_score<3, true, 4, true, false, 1>(a, q_len, theta, st);
    }
  }
  // big tiles (8 waves) once there are enough of them, small tiles for short caches
  if (a.idx_t != nullptr && a.out_t == nullptr) {     // compact mirror
    return a.L >= big_tiles_from() ? launch_score<BITS, true, 8, true, true>(a, q_len, theta, st)
                        : launch_score<BITS, true, 4, true, true>(a, q_len, theta, st);
  }
  if (a.out_t != nullptr) {
    return a.L >= big_tiles_from() ? launch_score<BITS, true, 8, true>(a, q_len, theta, st)
                        : launch_score<BITS, true, 4, true>(a, q_len, theta, st);
  }
  // (row-layout outliers -- the legacy entry points: 4-wave tiles measured 53.0 -> 46.4 us at 32K and 105 -> 123 at 128K,
  //  profiles/r06_t_rows_ab.txt; not adopted: the fused-softmax form of the same call counts 256-token tiles from 16K on, and the
  //  two forms are held bit-identical -- tests/test_fused_gpu.py)
  if (a.L >= (sparse ? big_tiles_from() : (int64_t)16384)) {
    return sparse ? launch_score<BITS, true, 8>(a, q_len, theta, st) : launch_score<BITS, false, 8>(a, q_len, theta, st);
  }
  return sparse ? launch_score<BITS, true, 4>(a, q_len, theta, st) : launch_score<BITS, false, 4>(a, q_len, theta, st);
}

static size_t ws_bytes(int bits, int q_len, int H) {
  return bits == 4 ? ktab_total_bytes<4>(q_len, H) : (bits == 3 ? ktab_total_bytes<3>(q_len, H) : ktab_total_bytes<2>(q_len, H));
}

}  // namespace kvq


namespace kvq {
// rows [t0, t1) of the reference's outlier rows [max_len][n_out] -> columns of the token-contiguous mirror [n_out][max_len]
// (64 tokens per block through an LDS tile: both sides coalesced)
__global__ __launch_bounds__(256) void mirror_rows_kernel(const float *__restrict__ rows_v, const int32_t *__restrict__ rows_i,
                                                          float *__restrict__ out_t, int32_t *__restrict__ idx_t, int n_out,
                                                          int64_t max_len, int64_t t0, int64_t t1) {
  __shared__ float tv[64][65];
  __shared__ int32_t ti[64][65];
  const int64_t b0 = t0 + (int64_t)blockIdx.x * 64;
  const int nt = (t1 - b0 < 64) ? (int)(t1 - b0) : 64;
  for (int s0 = 0; s0 < n_out; s0 += 64) {
    const int ns = (n_out - s0 < 64) ? n_out - s0 : 64;
    for (int e = threadIdx.x; e < nt * ns; e += 256) {
      const int t = e / ns, sl = e % ns;
      tv[t][sl] = rows_v[(b0 + t) * n_out + s0 + sl];
      ti[t][sl] = rows_i[(b0 + t) * n_out + s0 + sl];
    }
    __syncthreads();
    for (int e = threadIdx.x; e < ns * 64; e += 256) {
      const int sl = e / 64, t = e % 64;
      if (t < nt) {
        out_t[(int64_t)(s0 + sl) * max_len + b0 + t] = tv[t][sl];
        idx_t[(int64_t)(s0 + sl) * max_len + b0 + t] = ti[t][sl];
      }
    }
    __syncthreads();
  }
}
}  // namespace kvq

using namespace kvq;

extern "C" {

int kvq_score_k_head_groups(int H, int64_t tiles, int q_len, int max_heads_per_group, int slots) {
  if (H <= 0 || tiles <= 0 || q_len <= 0 || max_heads_per_group <= 0 || slots <= 0) return 0;
  return pick_groups(H, tiles, q_len, max_heads_per_group, slots);
}

size_t kvq_score_k_workspace_bytes(int bits, int q_len, int H) {
  if (bits < 2 || bits > 4 || q_len <= 0 || H <= 0) return 0;
  return ws_bytes(bits, q_len, H);
}

static int score_entry(int bits, const void *q, int q_is_half, int tables_ready, const int32_t *mat, float *mul,
                       const float *lut, int q_len, int H, int hd, int64_t L, int64_t max_len, float rope_theta,
                       int pos_offset, const float *outliers, const int32_t *outlier_idx, int n_out, int accumulate,
                       void *workspace, size_t workspace_bytes, void *stream, float *sm_parts = nullptr,
                       float sm_inv = 0.f, int sm_nparts = 0, const float *outliers_t = nullptr,
                       const int32_t *idx_t = nullptr, int pair = 0) {
  if ((!q && !tables_ready) || !mat || !mul || !lut || q_len <= 0 || H <= 0 || hd != kHeadDim || L < 0 ||
      L > max_len || bits < 2 || bits > 4)
    return KVQ_EINVAL;
  const bool sparse = outliers != nullptr || idx_t != nullptr;     // (idx_t alone: the compact mirror)
  if (sparse && ((outliers && !outlier_idx) || (outliers_t && !idx_t) || n_out <= 0 || n_out > 4096)) return KVQ_EINVAL;
  if (idx_t && (int64_t)n_out * max_len * 4 >= (1ll << 32)) return KVQ_EINVAL;   // 32-bit lane offsets
  if (L == 0) return KVQ_OK;
  if (!workspace || workspace_bytes < kvq_score_k_workspace_bytes(bits, q_len, H) ||
      reinterpret_cast<uintptr_t>(workspace) % 16)
    return KVQ_EWORKSPACE;
  if ((int64_t)hd / 32 * bits * max_len * 4 >= (1ll << 32)) return KVQ_EINVAL;   // 32-bit lane offsets
  ScoreKArgs a;
  a.q = nullptr;
  a.mat = reinterpret_cast<const uint32_t *>(mat);
  a.mul = mul;
  a.tab = nullptr;
  a.outliers = outliers;
  a.idx = outlier_idx;
  a.out_t = outliers_t;
  a.idx_t = idx_t;
  a.H = H;
  a.hpg = H;
  a.L = L;
  a.max_len = max_len;
  a.pos_offset = pos_offset;
  a.n_out = n_out;
  a.n_out_magic = sparse ? (uint32_t)(((1ull << 32) + (uint64_t)n_out - 1) / (uint64_t)n_out) : 0u;
  a.accumulate = accumulate;
  a.pair = pair;
  a.tab_pair = nullptr;
  a.sm_parts = sm_parts;
  a.sm_inv = sm_inv;
  a.sm_nparts = sm_nparts;
  if (sm_parts && (q_len != 1 || accumulate || !sparse)) return KVQ_EINVAL;
  hipStream_t st = (hipStream_t)stream;
  switch (bits) {
    case 4: return dispatch_score<4>(a, lut, q, q_is_half, tables_ready, workspace, q_len, rope_theta, sparse, st);
    case 3: return dispatch_score<3>(a, lut, q, q_is_half, tables_ready, workspace, q_len, rope_theta, sparse, st);
    default: return dispatch_score<2>(a, lut, q, q_is_half, tables_ready, workspace, q_len, rope_theta, sparse, st);
  }
}

int kvq_score_k(int bits, const float *q, const int32_t *mat, float *mul, const float *lut, int q_len, int H,
                int hd, int64_t L, int64_t max_len, float rope_theta, int pos_offset, const float *outliers,
                const int32_t *outlier_idx, int n_out, int accumulate, void *workspace, size_t workspace_bytes,
                void *stream) {
  return score_entry(bits, q, 0, 0, mat, mul, lut, q_len, H, hd, L, max_len, rope_theta, pos_offset, outliers,
                     outlier_idx, n_out, accumulate, workspace, workspace_bytes, stream);
}


/* kvq_score_k over the token-contiguous outlier MIRROR [n_out][max_len] (kvq.h 2: what kvquant_amd.cache.QuantK keeps next to
 * the reference's rows) instead of the rows: the decode kernel's fast variant behind the legacy call's semantics (tables built
 * from q here, `mul` accumulated or overwritten).  q_len = 1. */
int kvq_score_k_mirror(int bits, const float *q, const int32_t *mat, float *mul, const float *lut, int H, int hd, int64_t L,
                       int64_t max_len, float rope_theta, int pos_offset, const float *outliers_t,
                       const int32_t *outlier_idx_t, int n_out, int accumulate, void *workspace, size_t workspace_bytes,
                       void *stream) {
  if (!outliers_t || !outlier_idx_t) return KVQ_EINVAL;
  return score_entry(bits, q, 0, 0, mat, mul, lut, 1, H, hd, L, max_len, rope_theta, pos_offset, nullptr, nullptr, n_out,
                     accumulate, workspace, workspace_bytes, stream, nullptr, 0.f, 0, outliers_t, outlier_idx_t);
}

/* Rows [t0, t1) of outlier rows in the reference's layout (f32 / i32 [max_len][n_out]) -> the same columns of the
 * token-contiguous mirror (f32 / i32 [n_out][max_len]).  For callers that keep the reference's rows as the source of truth and
 * want the decode kernel's mirror variant (kvquant_amd.quant_cuda does this behind the module swap). */
int kvq_outlier_mirror_rows(const float *outliers, const int32_t *outlier_idx, float *outliers_t, int32_t *outlier_idx_t,
                            int n_out, int64_t max_len, int64_t t0, int64_t t1, void *stream) {
  if (!outliers || !outlier_idx || !outliers_t || !outlier_idx_t || n_out <= 0 || t0 < 0 || t1 > max_len || t0 > t1)
    return KVQ_EINVAL;
  if (t0 == t1) return KVQ_OK;
  const unsigned blocks = (unsigned)((t1 - t0 + 63) / 64);
  mirror_rows_kernel<<<blocks, 256, 0, (hipStream_t)stream>>>(outliers, outlier_idx, outliers_t, outlier_idx_t, n_out, max_len, t0, t1);
  return check_launch();
}

int kvq_score_k_prepared(int bits, const int32_t *mat, float *mul, const float *lut, int H, int hd, int64_t L,
                         int64_t max_len, float rope_theta, int pos_offset, const float *outliers,
                         const int32_t *outlier_idx, int n_out, int accumulate, void *workspace,
                         size_t workspace_bytes, void *stream) {
  return score_entry(bits, nullptr, 0, 1, mat, mul, lut, 1, H, hd, L, max_len, rope_theta, pos_offset, outliers,
                     outlier_idx, n_out, accumulate, workspace, workspace_bytes, stream);
}

/* tiles of the sparse score kernel = (max, sum) partials per head it can write; 0: no fusion for this shape */
int kvq_score_k_softmax_parts(int bits, int64_t L, int sparse) {
  if (bits < 2 || bits > 4 || L <= 0 || !sparse) return 0;
  const int T = L >= big_tiles_from() ? 256 : 128;
  return (int)((L + T - 1) / T);
}

int kvq_score_k_prepared_softmax(int bits, const int32_t *mat, float *mul, const float *lut, int H, int hd,
                                 int64_t L, int64_t max_len, float rope_theta, int pos_offset,
                                 const float *outliers, const int32_t *outlier_idx, int n_out,
                                 const float *outliers_t, const int32_t *outlier_idx_t, void *workspace,
                                 size_t workspace_bytes, float inv_sqrt_hd, float *softmax_parts, int n_parts,
                                 void *stream) {
  return kvq_score_k_prepared_softmax_ex(bits, mat, mul, lut, H, hd, L, max_len, rope_theta, pos_offset, outliers,
                                         outlier_idx, n_out, outliers_t, outlier_idx_t, workspace, workspace_bytes,
                                         inv_sqrt_hd, softmax_parts, n_parts, 0, stream);
}

int kvq_score_k_prepared_softmax_ex(int bits, const int32_t *mat, float *mul, const float *lut, int H, int hd,
                                    int64_t L, int64_t max_len, float rope_theta, int pos_offset,
                                    const float *outliers, const int32_t *outlier_idx, int n_out,
                                    const float *outliers_t, const int32_t *outlier_idx_t, void *workspace,
                                    size_t workspace_bytes, float inv_sqrt_hd, float *softmax_parts, int n_parts,
                                    int flags, void *stream) {
  if (!softmax_parts || (!outliers && !outlier_idx_t) || n_parts != kvq_score_k_softmax_parts(bits, L, 1))
    return KVQ_EINVAL;
  return score_entry(bits, nullptr, 0, 1, mat, mul, lut, 1, H, hd, L, max_len, rope_theta, pos_offset,
                     outlier_idx_t ? nullptr : outliers, outlier_idx_t ? nullptr : outlier_idx, n_out, 0, workspace,
                     workspace_bytes, stream, softmax_parts, inv_sqrt_hd, n_parts, outliers_t, outlier_idx_t,
                     (flags & KVQ_SCORE_F32_PAIR_TABLES) ? 2 : ((flags & KVQ_SCORE_F16_PAIR_TABLES) ? 1 : 0));
}

}  // extern "C"
