// GPU-resident decode append: NUQ pack + exact top-k outlier selection + outlier
// row assembly in ONE launch per tensor, replacing the reference's
//   kernel -> .cpu() -> torch.topk on the host -> 4 x .cuda() -> ~10 tiny GPU ops
// round trip (modeling_llama.py:706-751 for K, 1803-1820 + 1086-1176 for V).
//
// One workgroup of 1024 lanes per token; lane l owns channels 4l..4l+3 (for
// C = 4096; larger C loops).  Selection is an exact radix select on the
// order-preserving integer image of the floats (4 passes of 8-bit digits, LDS
// histograms), both tails at once.  Ties at the k-th value are broken by LOWEST
// channel index (torch.topk leaves this unspecified).  The selected entries are
// compacted in channel order with a ballot/prefix scan, which is exactly the
// "sort by index" of the reference (modeling_llama.py:742, 1171), so no sort.
#include "kvq_common.h"
#include "kvq_host.h"
#include "kvq_ktab.h"
#include "kvq_select.h"

#include <cstdlib>

#ifndef KVQ_FAST_SELECT
#define KVQ_FAST_SELECT 1    // outlier selection by pruning + bitwise search (kvq_select.h); 0: the four-pass radix select (A/B runs)
#endif
#ifndef KVQ_PACK_TILED
#define KVQ_PACK_TILED 1     // prefill pack: four tokens per workgroup, the prompt read once (kvq_pack_tiled.h); 0: the
#endif                       // per-token workgroups (A/B runs)

namespace kvq {

constexpr int kSelThreads = 1024;
constexpr int kSelWaves = kSelThreads / 64;
constexpr int kMaxPerLane = 8;  // channels per lane: C <= 8192

// COMPACT outlier entry (opt-in format): fp16 residual in the high half, channel (< 65536) in the low half --
// 4 bytes instead of the reference's f32 value + i32 index.  Convention of every entry point that takes an
// (outliers, outlier_idx) or a mirror pair: a NULL value pointer with a non-NULL index pointer means "packed entries in
// the index array".
__device__ __forceinline__ int32_t pack_entry(float val, int c) {
  return (int32_t)(((uint32_t)__half_as_ushort(__float2half_rn(val)) << 16) | (uint32_t)c);
}

__device__ __forceinline__ uint32_t fkey(float x) {  // ascending order-preserving
  uint32_t b = __float_as_uint(x);
  return b ^ ((b >> 31) ? 0xffffffffu : 0x80000000u);
}

template <int NT, int E>
struct SelShared {
  // [pass parity][side][copy][digit]: the next pass's histogram is zeroed during this pass's scan; lanes spread over
  // HC copies of a histogram (same-address LDS atomics serialise: the first pass puts 4096 elements into ~10 bins)
  static constexpr int HC = NT >= 1024 ? 4 : 1;
  uint32_t hist[2][2][HC][256];    // (the candidate lists of the pruning select alias it: kvq_select.h, FselShared)
  FselCtl fctl;
  uint32_t fallback;
  uint32_t wsum[NT / 64 + 1];
  uint32_t prefix[2];
  uint32_t krem[2];
  uint32_t scan[NT / 64];
  unsigned codes[E * NT + E * NT / 32];   // channel c at c + c/32: the pack's lane-per-group reads are conflict free
};

static_assert(sizeof(FselShared) <= sizeof(uint32_t) * 2 * 2 * 256, "the candidate lists alias one histogram copy");

// inclusive prefix sum over the 64 lanes of a wave with DPP row operations: six dependent VALU instructions instead of
// six ds_bpermute round trips (the select is a chain of latencies: four passes x (atomics, barrier, scan, barrier))
__device__ __forceinline__ uint32_t wave_incl_scan(uint32_t v) {
  v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x111, 0xf, 0xf, false);   // row_shr:1
  v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x112, 0xf, 0xf, false);   // row_shr:2
  v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x114, 0xf, 0xf, false);   // row_shr:4
  v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x118, 0xf, 0xf, false);   // row_shr:8
  v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x142, 0xa, 0xf, false);   // row_bcast:15 -> rows 1, 3
  v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x143, 0xc, 0xf, false);   // row_bcast:31 -> rows 2, 3
  return v;
}

// exclusive block scan of one uint per lane (wave shuffles + LDS across waves)
template <int NT>
__device__ __forceinline__ uint32_t block_excl_scan(uint32_t v, uint32_t *ws, uint32_t &total) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const uint32_t inc = wave_incl_scan(v);
  __syncthreads();
  if (lane == 63) ws[wave] = inc;
  __syncthreads();
  uint32_t base = 0, tot = 0;
#pragma unroll
  for (int w = 0; w < NT / 64; w++) {
    const uint32_t s = ws[w];
    if (w < wave) base += s;
    tot += s;
  }
  total = tot;
  return base + inc - v;
}

// Exact k-th order statistics of both tails of keys[0..n) (n per lane valid):
//   T[0] = key of the k-th LARGEST element, gt[0] = #keys > T[0]
//   T[1] = key of the k-th SMALLEST element, gt[1] = #keys < T[1]
template <int NT, int E>
__device__ __forceinline__ void radix_select_both(const uint32_t (&key)[E], const bool (&ok)[E], uint32_t k,
                                                  SelShared<NT, E> &sh, uint32_t (&T)[2], uint32_t (&gt)[2]) {
  const int tid = threadIdx.x;
  if (tid < 2) {
    sh.prefix[tid] = 0;
    sh.krem[tid] = k;
  }
  constexpr int HC = SelShared<NT, E>::HC;
  for (int i = tid; i < 512 * HC; i += NT) (&sh.hist[1][0][0][0])[i] = 0;   // (the first pass is pass 3: parity 1)
  __syncthreads();
  for (int pass = 3; pass >= 0; pass--) {
    uint32_t (*hist)[HC][256] = sh.hist[pass & 1];
    const int cp = tid & (HC - 1);
    const uint32_t p0 = sh.prefix[0], p1 = sh.prefix[1];
#pragma unroll
    for (int e = 0; e < E; e++) {
      if (!ok[e]) continue;
      const uint32_t kk = key[e];
      const uint32_t hi = (pass == 3) ? 0u : (kk >> (8 * (pass + 1)));
      const uint32_t d = (kk >> (8 * pass)) & 0xffu;
      // (the LDS atomics are what bounds the prefill pack -- 8 tokens per CU histogram at once; in the first pass
      //  both sides count every element, so they share one histogram: 40 % fewer atomics over the four passes)
      if (hi == p0) atomicAdd(&hist[0][cp][d], 1u);
      if (pass != 3 && hi == p1) atomicAdd(&hist[1][cp][d], 1u);
    }
    __syncthreads();
    // wave 0 resolves the "largest" side (scan bins downward), wave 1 the "smallest" side (upward); the other
    // lanes zero the next pass's histogram meanwhile (two barriers per pass instead of three)
    const int wave = tid >> 6, lane = tid & 63;
    if (wave < 2) {
      const int side = wave;
      // lane handles 4 consecutive bins in scan order
      uint32_t c[4];
      uint32_t s = 0;
#pragma unroll
      for (int j = 0; j < 4; j++) {
        const int pos = lane * 4 + j;                      // position in scan order
        const int bin = side == 0 ? 255 - pos : pos;
        c[j] = 0;
#pragma unroll
        for (int q = 0; q < HC; q++) c[j] += hist[pass == 3 ? 0 : side][q][bin];
        s += c[j];
      }
      const uint32_t inc = wave_incl_scan(s);
      const uint32_t before = inc - s;                     // elements in bins scanned before this lane's
      const uint32_t kr = sh.krem[side];
      if (before < kr && kr <= inc) {                      // the k-th element is in one of my 4 bins
        uint32_t run = before;
#pragma unroll
        for (int j = 0; j < 4; j++) {
          if (run < kr && kr <= run + c[j]) {
            const int pos = lane * 4 + j;
            const int bin = side == 0 ? 255 - pos : pos;
            sh.prefix[side] = (sh.prefix[side] << 8) | (uint32_t)bin;
            sh.krem[side] = kr - run;                      // rank inside the chosen bin
          }
          run += c[j];
        }
      }
    } else if (pass > 0) {
      for (int i = tid - 128; i < 512 * HC; i += NT - 128) (&sh.hist[(pass - 1) & 1][0][0][0])[i] = 0;
    }
    __syncthreads();
  }
  T[0] = sh.prefix[0];
  T[1] = sh.prefix[1];
  // after the last pass krem = rank among the elements EQUAL to T; #strictly beyond = k - krem
  gt[0] = k - sh.krem[0];
  gt[1] = k - sh.krem[1];
}

struct AppendArgs {
  uint32_t *mat;
  const float *lut;        // K: [C][N]
  const float *lut_off;    // K: table the residuals refer to (lut, or the Q-Norm table)
  float *lut_rows;         // V: [max_len][N], row `col` is written
  const float *lut_sorted; // V: [N]
  float *lut_rows2;        // V Q-Norm (optional): [max_len][N], row `col` = (lut_sorted*normscale+normoffset)*sf+off
  float normscale, normoffset;
  int zp_from_rows2;       // V Q-Norm: the residuals refer to lut_rows2[col][zero code] (ML:1153-1156, 1369-1375)
  int tie_quirk;           // V: replicate the reference's double count of an outlier that equals the clip threshold
  const void *x;           // element of channel c: x[c * x_stride] (fp32 or fp16); decode: [C], stride 1
  int x_is_half;
  int64_t x_stride;        // prefill pack: channel-major [C][S] input, stride S, x points at the token's column
  int codes_elsewhere;     // K in the decode prologue: the per-head table workgroups quantize and pack the token
  const float *lut_ends;   // K, optional: [C][2] = (lut_off[c][0], lut_off[c][N-1]) contiguous, so that the
                           // selection workgroup gets the residual end points without touching the codebook
  const float *lo, *hi;    // K thresholds
  float *outliers;
  int32_t *outlier_idx;
  float *outliers_t;       // K, optional: token-contiguous mirror [2*thr_k][max_len] of the outlier rows
  int32_t *outlier_idx_t;
  int thr_k;
  int C;
  int64_t max_len;
  int64_t col;
#if KVQ_TRACE
  unsigned long long *trace;   // development: [16] phase stamps of lane 0 (tools/dbg/trace_prologue.py)
#endif
};

#ifndef KVQ_TRACE
#define KVQ_TRACE 0
#endif
#if KVQ_TRACE
#define KVQ_STAMP(A, k)                                                                         \
  do {                                                                                          \
    if (threadIdx.x == 0 && (A).trace) {                                                        \
      unsigned long long tt;                                                                    \
      asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(tt)::"memory");           \
      (A).trace[k] = tt;                                                                        \
    }                                                                                           \
  } while (0)
#else
#define KVQ_STAMP(A, k)
#endif

// K: rescaled selection, per-channel LUT; V: raw selection, per-token LUT row built here.
// Executed by one whole workgroup of NT lanes (NT = 1024: the decode append, one latency-critical workgroup;
// NT = 256: the prefill pack, where four times as many tokens in flight per CU hide the select's barriers),
// E = channels per lane the arrays are sized for (C <= E * NT).
template <int BITS, bool IS_V, int NT = kSelThreads, int E = kMaxPerLane>
__device__ __forceinline__ void fused_append_body(const AppendArgs &A) {
  uint32_t *__restrict__ mat = A.mat;
  const float *__restrict__ lut = A.lut;
  const float *__restrict__ lut_off = A.lut_off;
  float *__restrict__ lut_rows = A.lut_rows;
  const float *__restrict__ lut_sorted = A.lut_sorted;
  const float *__restrict__ lo = A.lo;
  const float *__restrict__ hi = A.hi;
  float *__restrict__ outliers = A.outliers;
  int32_t *__restrict__ outlier_idx = A.outlier_idx;
  const int thr_k = A.thr_k, C = A.C;
  const int64_t max_len = A.max_len, col = A.col;
  constexpr int N = Fmt<BITS>::kN;
  __shared__ SelShared<NT, E> sh;
  __shared__ float vrow[16], vrow2[16];
  const int tid = threadIdx.x;
  const int per = (C + NT - 1) / NT;   // channels per lane (4 at C = 4096 / 1024 lanes), <= E
  const int c0 = tid * per;

  float xv[E], sel[E];
  uint32_t key[E];
  bool ok[E];
  KVQ_STAMP(A, 0);
#pragma unroll
  for (int e = 0; e < E; e++) {
    ok[e] = e < per && (c0 + e) < C;
    xv[e] = ok[e] ? ld_act(A.x, (int64_t)(c0 + e) * A.x_stride, A.x_is_half) : 0.f;
  }
  if constexpr (!IS_V) {
#pragma unroll
    for (int e = 0; e < E; e++) {
      if (!ok[e]) { sel[e] = 0.f; continue; }
      const float l = lo[c0 + e], h = hi[c0 + e];
      const float rangeval = (h - l) / 2;      // KCU:1759-1764
      const float zeropoint = (h + l) / 2;
      sel[e] = (xv[e] - zeropoint) / rangeval;
    }
  } else {
#pragma unroll
    for (int e = 0; e < E; e++) sel[e] = xv[e];
  }
#pragma unroll
  for (int e = 0; e < E; e++) key[e] = fkey(sel[e]);

  // K: the codes do not depend on the selection, so the per-channel codebook rows (256 KB, the largest
  // read of the whole append) are fetched together with x / lo / hi -- one memory round trip instead of two --
  // and the end points the outlier residuals refer to are kept from the same rows when no Q-Norm table is
  // in play.
  float end_lo[E], end_hi[E];
  const bool same_tab = lut_off == lut;
  const bool own_codes = IS_V || !A.codes_elsewhere;
  const bool pre_ends = !IS_V && A.lut_ends != nullptr;
  if constexpr (!IS_V) {
#pragma unroll
    for (int e = 0; e < E; e++) {
      end_lo[e] = end_hi[e] = 0.f;
      if (ok[e] && pre_ends) {
        const float2 t = *reinterpret_cast<const float2 *>(A.lut_ends + 2 * (c0 + e));
        end_lo[e] = t.x;
        end_hi[e] = t.y;
      }
      if (!ok[e] || !own_codes) continue;
      float row[N];
      const float *src = lut + (int64_t)(c0 + e) * N;
#pragma unroll
      for (int v = 0; v < N; v += 4) {
        const float4 t = *reinterpret_cast<const float4 *>(src + v);
        row[v] = t.x; row[v + 1] = t.y; row[v + 2] = t.z; row[v + 3] = t.w;
      }
      sh.codes[(c0 + e) + ((c0 + e) >> 5)] = nearest_code<N>(row, xv[e]);
      if (!pre_ends) {
        end_lo[e] = row[0];
        end_hi[e] = row[N - 1];
      }
    }
  }

  uint32_t T[2], gt[2];
  const uint32_t ksel = IS_V ? (uint32_t)(thr_k + 1) : (uint32_t)thr_k;   // V: threshold is the (thr_k+1)-th
  asm volatile("" :: "v"(key[0]), "v"(end_lo[0]));
  KVQ_STAMP(A, 1);
  uint32_t eq_hi = 0, eq_lo = 0;          // #keys equal to each threshold
  bool radix = !KVQ_FAST_SELECT;
  if constexpr (KVQ_FAST_SELECT != 0) {
    // pruning select (kvq_select.h): candidates of every wave -> one list per side (aliases the histograms), one wave per
    // side resolves it; a token whose list overflows -- hundreds of keys tied at the bound -- takes the radix select below
    FselShared &fs = reinterpret_cast<FselShared &>(sh.hist);
    fsel_bounds<E>(key, ok, ksel, NT / 64, tid >> 6, sh.fctl);
    if (tid < 2) sh.fctl.ncand[tid] = 0;
    if (tid == 0) sh.fallback = 0;
    KVQ_STAMP(A, 7);
    __syncthreads();
    KVQ_STAMP(A, 8);
    fsel_collect<E>(key, ok, NT / 64, fs, sh.fctl);
    KVQ_STAMP(A, 9);
    __syncthreads();
    KVQ_STAMP(A, 10);
    if (tid < 128) {
      if (!fsel_resolve(tid >> 6, ksel, fs, sh.fctl)) sh.fallback = 1;
    }
    KVQ_STAMP(A, 11);
    __syncthreads();
    KVQ_STAMP(A, 12);
    radix = sh.fallback != 0;              // (block-uniform)
    if (!radix) {
      T[0] = sh.fctl.res[0][0]; gt[0] = sh.fctl.res[0][1]; eq_hi = sh.fctl.res[0][2];
      T[1] = sh.fctl.res[1][0]; gt[1] = sh.fctl.res[1][1]; eq_lo = sh.fctl.res[1][2];
    }
  }
  if (radix) {
    radix_select_both<NT, E>(key, ok, ksel, sh, T, gt);
    // (the last radix pass left the number of elements equal to each threshold in its histogram, pass 0: parity 0)
    for (int q = 0; q < SelShared<NT, E>::HC; q++) {
      eq_hi += sh.hist[0][0][q][T[0] & 0xffu];
      eq_lo += sh.hist[0][1][q][T[1] & 0xffu];
    }
  }
  KVQ_STAMP(A, 2);

  // ---- membership: strictly beyond the threshold, plus the first ties in channel order ----------
  // K keeps k = thr_k per side; V keeps the top thr_k of the thr_k+1 selected (the last-ranked one,
  // i.e. the highest-index tie, is the clipping threshold itself: modeling_llama.py:1091-1096).
  // (when the number of elements equal to a threshold is exactly the number still wanted, or none is wanted -- no run of
  //  ties is cut, the usual case -- the ranks are not needed and the block scan is skipped)
  const uint32_t w_hi = (uint32_t)thr_k - gt[0], w_lo = (uint32_t)thr_k - gt[1];
  const bool cut = !((w_hi == 0 || eq_hi == w_hi) && (w_lo == 0 || eq_lo == w_lo));   // (block-uniform)
  uint32_t rank_hi = 0, rank_lo = 0;
  if (cut) {
    uint32_t ntie_hi = 0, ntie_lo = 0;
#pragma unroll
    for (int e = 0; e < E; e++) {
      if (!ok[e]) continue;
      ntie_hi += key[e] == T[0];
      ntie_lo += key[e] == T[1];
    }
    uint32_t tot;
    const uint32_t packed = block_excl_scan<NT>(ntie_hi | (ntie_lo << 16), sh.scan, tot);
    rank_hi = packed & 0xffffu;
    rank_lo = packed >> 16;
  }
  const uint32_t want_hi = (uint32_t)thr_k - gt[0], want_lo = (uint32_t)thr_k - gt[1];
  bool in_hi[E], in_lo[E];
  uint32_t nsel = 0;
#pragma unroll
  for (int e = 0; e < E; e++) {
    in_hi[e] = in_lo[e] = false;
    if (!ok[e]) continue;
    if (key[e] > T[0]) in_hi[e] = true;
    else if (key[e] == T[0]) { in_hi[e] = rank_hi < want_hi; rank_hi++; }
    if (key[e] < T[1]) in_lo[e] = true;
    else if (key[e] == T[1]) { in_lo[e] = rank_lo < want_lo; rank_lo++; }
    nsel += (in_hi[e] || in_lo[e]);
  }

  KVQ_STAMP(A, 3);
  // ---- V: thresholds, scale/offset and the per-token codebook row ---------------------------------
  float vmin = 0.f, vmax = 0.f, zp = 0.f;
  if constexpr (IS_V) {
    // threshold VALUES: decode the keys back
    const uint32_t bh = T[0] ^ ((T[0] >> 31) ? 0x80000000u : 0xffffffffu);
    const uint32_t bl = T[1] ^ ((T[1] >> 31) ? 0x80000000u : 0xffffffffu);
    vmax = __uint_as_float(bh);
    vmin = __uint_as_float(bl);
    const float offset = (vmax + vmin) / 2;    // modeling_llama.py:1097-1098 (fp32)
    const float sf = (vmax - vmin) / 2;
    if (tid < N) {
      const float r = lut_sorted[tid] * sf + offset;   // two roundings (-ffp-contract=off), ML:1113
      vrow[tid] = r;
      lut_rows[col * N + tid] = r;
      if (A.lut_rows2 != nullptr) {                    // Q-Norm row, ML:1116-1118 (fp32, one rounding per op)
        const float r2 = (lut_sorted[tid] * A.normscale + A.normoffset) * sf + offset;
        vrow2[tid] = r2;
        A.lut_rows2[col * N + tid] = r2;
      }
    }
    __syncthreads();
    zp = (A.lut_rows2 != nullptr && A.zp_from_rows2) ? vrow2[Fmt<BITS>::kZeroCode] : vrow[Fmt<BITS>::kZeroCode];
  }

  // ---- V codes (need the clip thresholds and the row built above) ----------------------------------------
  if constexpr (IS_V) {
#pragma unroll
    for (int e = 0; e < E; e++) {
      if (!ok[e]) continue;
      float row[N];
#pragma unroll
      for (int v = 0; v < N; v++) row[v] = vrow[v];
      // clipped to the zero-point code iff stored sparse (include/kvq.h: kvq_vopts); with the reference's quirk:
      // iff strictly outside the thresholds (KCU:2084), which misses a selected element that equals one of them
      const bool clip = A.tie_quirk ? (xv[e] < vmin || xv[e] > vmax) : (in_hi[e] || in_lo[e]);
      sh.codes[(c0 + e) + ((c0 + e) >> 5)] = clip ? Fmt<BITS>::kZeroCode : nearest_code<N>(row, xv[e]);
    }
  }

  KVQ_STAMP(A, 4);
  // ---- outlier row: compaction in channel order ------------------------------------------------------
  uint32_t tot2;
  uint32_t pos = block_excl_scan<NT>(nsel, sh.scan, tot2);   // (the barriers inside also publish sh.codes)
  const int n_out = 2 * thr_k;
  float *orow = outliers + col * n_out;
  int32_t *irow = outlier_idx + col * n_out;
#pragma unroll
  for (int e = 0; e < E; e++) {
    if (!ok[e] || !(in_hi[e] || in_lo[e])) continue;
    float val;
    if constexpr (IS_V) {
      val = xv[e] - zp;                                   // modeling_llama.py:1169
    } else {
      const int c = c0 + e;
      // residual to the saturated end point; zero when the rescaled value is inside [-1, 1]
      // (modeling_llama.py:729-747)
      const bool have_ends = pre_ends || (same_tab && own_codes);
      if (in_hi[e]) val = (sel[e] <= 1.0f) ? 0.f : xv[e] - (have_ends ? end_hi[e] : lut_off[(int64_t)c * N + (N - 1)]);
      else val = (sel[e] >= -1.0f) ? 0.f : xv[e] - (have_ends ? end_lo[e] : lut_off[(int64_t)c * N]);
    }
    if ((int)pos < n_out) {
      if (outliers != nullptr) {
        orow[pos] = val;
        irow[pos] = c0 + e;
      } else if (outlier_idx != nullptr) {
        irow[pos] = pack_entry(val, c0 + e);                  // compact rows
      }
      if constexpr (!IS_V) {
        if (A.outliers_t != nullptr) {
          A.outliers_t[(int64_t)pos * max_len + col] = val;
          A.outlier_idx_t[(int64_t)pos * max_len + col] = c0 + e;
        } else if (A.outlier_idx_t != nullptr) {
          A.outlier_idx_t[(int64_t)pos * max_len + col] = pack_entry(val, c0 + e);   // compact mirror
        }
      }
    }
    pos++;
  }

  KVQ_STAMP(A, 5);
  // ---- pack: one lane per 32-channel group ------------------------------------------------------------
  if (!own_codes) return;
  for (int g = tid; g < C / 32; g += NT) {
    unsigned cd[32];
#pragma unroll
    for (int i = 0; i < 32; i++) cd[i] = sh.codes[g * 33 + i];
    uint32_t w[BITS];
    pack32<BITS>(cd, w);
#pragma unroll
    for (int i = 0; i < BITS; i++) mat[((int64_t)g * BITS + i) * max_len + col] = w[i];
  }
  KVQ_STAMP(A, 6);
}

}  // namespace kvq
#include "kvq_pack_tiled.h"
namespace kvq {

template <int BITS, bool IS_V>
__global__ __launch_bounds__(kSelThreads) void fused_append_kernel(AppendArgs A) {
  fused_append_body<BITS, IS_V>(A);
}

// Prefill: the same body, one workgroup per prompt token (the reference runs torch.topk over [S, C] and ~10
// elementwise launches around its pack kernel, modeling_llama.py:879-972 / 1294-1382).  The prompt arrives
// channel-major [C][S] (KCPP:48-53): token s reads a column, 4-byte elements S apart -- the 64-byte sectors
// are shared by 16 neighbouring tokens and come out of L2 / the Infinity Cache for all but the first.
// kPackThreads lanes per token: 16 channels per lane at C = 4096.  Tokens are handed to the XCDs in contiguous
// ranges (workgroup b runs on XCD b % 8, MI355X_MICROARCH.md): the 32 tokens that share each 128-byte line of the
// channel-major input are then read through ONE L2, by workgroups that are dispatched back to back.
constexpr int kPackThreads = 256;
constexpr int kPackPerLane = 16;

template <int BITS, bool IS_V, int NT, int E>
__global__ __launch_bounds__(NT) void fused_pack_kernel(AppendArgs A, int64_t S) {
  const int64_t nb = gridDim.x;
  const int64_t per_xcd = (nb + 7) / 8;
  int64_t s = (int64_t)(blockIdx.x % 8) * per_xcd + blockIdx.x / 8;
  if (nb < 64) s = blockIdx.x;
  if (s >= S) return;
  A.col += s;
  A.x = reinterpret_cast<const float *>(A.x) + s;   // (fp32 prompt)
  fused_append_body<BITS, IS_V, NT, E>(A);
}

// Decode prologue of one layer in ONE launch: workgroup 0 = K fused append, 1 = V fused append,
// 2.. = query-premultiplied K codebook images (one head each) for kvq_score_k_prepared.  The three jobs
// are independent; run back to back they cost 27 + 16 + 4 us of mostly latency.
struct PrologueArgs {
  AppendArgs k, v;
  const float *klut;       // [H][128][N]: source of the score tables (k.lut, or the Q-Norm table at 2 bit)
  const void *q;           // [H][128] fp32 or fp16
  int q_is_half;
  unsigned char *tab;      // score workspace: tables, then q as fp32, then (3 bit) the pair-sum images
  float *q32;
  unsigned char *pair_tab;
  int pair_mode;           // 0: none, 1: fp16 pair sums (KTabPair3), 2: fp32 pair sums (KTabPair32)
  int H;
  // fp16 attention-sink tokens (optional): scaled scores q . k_sink of head h by the head's table workgroup -- the
  // reference's torch.matmul(query_states, key_states_fp16) / sqrt(d) (ML:1950-1962): fp32 accumulation, the fp16
  // result divided in fp16
  const __half *k_sink;    // [H][128][n_sink], post-RoPE
  __half *sink_scores;     // [H][n_sink]
  int n_sink;
  float sink_inv;
};

// codes + pack of head h of the new K token (vecquant{b}appendvecK semantics, KCU:1202-1245 ...): the per-head
// table workgroups read their head's codebook anyway, and it takes the 256 KB codebook read and the
// nearest-code search off the single selection workgroup, which is the critical path of the prologue.
template <int BITS>
__device__ __forceinline__ void quantize_head(const AppendArgs &A, int h) {
  constexpr int N = Fmt<BITS>::kN;
  __shared__ unsigned hcodes[kHeadDim];
  const int tid = threadIdx.x;
  if (tid < kHeadDim) {
    const int c = h * kHeadDim + tid;
    const float x = ld_act(A.x, c, A.x_is_half);
    float row[N];
    const float *src = A.lut + (int64_t)c * N;
#pragma unroll
    for (int v = 0; v < N; v += 4) {
      const float4 t = *reinterpret_cast<const float4 *>(src + v);
      row[v] = t.x; row[v + 1] = t.y; row[v + 2] = t.z; row[v + 3] = t.w;
    }
    hcodes[tid] = nearest_code<N>(row, x);
  }
  __syncthreads();
  if (tid < kHeadDim / 32) {
    unsigned cd[32];
#pragma unroll
    for (int i = 0; i < 32; i++) cd[i] = hcodes[tid * 32 + i];
    uint32_t w[BITS];
    pack32<BITS>(cd, w);
    const int g = h * (kHeadDim / 32) + tid;
#pragma unroll
    for (int i = 0; i < BITS; i++) A.mat[((int64_t)g * BITS + i) * A.max_len + A.col] = w[i];
  }
}

template <int BITS>
__global__ __launch_bounds__(kSelThreads) void decode_prologue_kernel(PrologueArgs P) {
  if (blockIdx.x == 0) {
    fused_append_body<BITS, false>(P.k);
  } else if (blockIdx.x == 1) {
    fused_append_body<BITS, true>(P.v);
  } else {
    const int h = (int)blockIdx.x - 2;
    quantize_head<BITS>(P.k, h);
    lutq_prep_head<BITS>(P.klut, P.q, P.q_is_half, P.tab, P.q32, P.pair_tab, P.H, h, 0, P.pair_mode);
    if (P.k_sink != nullptr) {
      // scores of the fp16 sink tokens: one wave per sink token, two channels per lane (one thread per token walked the
      // 128 channels alone -- 128 dependent loads, which made these workgroups the prologue's critical path: 19.4 us
      // against 12.2 without sink tokens)
      static_assert(kHeadDim == 128, "two channels per lane");
      const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = blockDim.x >> 6;
      const float q0 = ld_act(P.q, h * kHeadDim + lane, P.q_is_half), q1 = ld_act(P.q, h * kHeadDim + lane + 64, P.q_is_half);
      for (int i = wave; i < P.n_sink; i += nw) {
        float acc = q0 * __half2float(P.k_sink[((int64_t)h * kHeadDim + lane) * P.n_sink + i]);
        acc = fmaf(q1, __half2float(P.k_sink[((int64_t)h * kHeadDim + lane + 64) * P.n_sink + i]), acc);
#pragma unroll
        for (int d = 32; d >= 1; d >>= 1) acc += __shfl_xor(acc, d);
        if (lane == 0) P.sink_scores[h * P.n_sink + i] = __float2half_rn(scaled(acc, P.sink_inv));
      }
    }
  }
}

static int check_append(bool is_v, const AppendArgs &a, int H, int hd) {
  // outlier destinations: rows (f32 + i32, or index array alone = packed entries) and / or, K, the mirror
  if ((a.outliers && !a.outlier_idx) || (!a.outlier_idx && (is_v || !a.outlier_idx_t))) return KVQ_EINVAL;
  if (!a.mat || !a.x || a.thr_k <= 0 || H <= 0 || hd <= 0 || hd % 32 ||
      a.col < 0 || a.col >= a.max_len || a.C > kMaxPerLane * kSelThreads || 2 * (a.thr_k + 1) > a.C ||
      a.C >= 65536)
    return KVQ_EINVAL;
  if (is_v ? (!a.lut_rows || !a.lut_sorted) : (!a.lut || !a.lut_off || !a.lo || !a.hi)) return KVQ_EINVAL;
  if (a.outliers_t != nullptr && a.outlier_idx_t == nullptr) return KVQ_EINVAL;   // (index alone: the compact mirror)
  return KVQ_OK;
}

template <bool IS_V>
static int launch_fused(int bits, const AppendArgs &a, int H, int hd, hipStream_t st) {
  int rc = check_append(IS_V, a, H, hd);
  if (rc) return rc;
  dim3 grid(1), block(kSelThreads);
  switch (bits) {
    case 4: fused_append_kernel<4, IS_V><<<grid, block, 0, st>>>(a); break;
    case 3: fused_append_kernel<3, IS_V><<<grid, block, 0, st>>>(a); break;
    case 2: fused_append_kernel<2, IS_V><<<grid, block, 0, st>>>(a); break;
    default: return KVQ_EINVAL;
  }
  return check_launch();
}

static AppendArgs k_args(int32_t *mat, const float *lut, const float *lut_off, const void *x, int x_is_half,
                         const float *lo, const float *hi, float *outliers, int32_t *idx, int thr_k, int H, int hd,
                         int64_t max_len, int64_t col, float *outliers_t = nullptr, int32_t *idx_t = nullptr) {
  AppendArgs a;
#if KVQ_TRACE
  a.trace = nullptr;
#endif
  a.outliers_t = outliers_t;
  a.outlier_idx_t = idx_t;
  a.x_stride = 1;
  a.codes_elsewhere = 0;
  a.lut_ends = nullptr;
  a.mat = reinterpret_cast<uint32_t *>(mat);
  a.lut = lut;
  a.lut_off = lut_off;
  a.lut_rows = nullptr;
  a.lut_sorted = nullptr;
  a.lut_rows2 = nullptr;
  a.normscale = 1.f;
  a.normoffset = 0.f;
  a.zp_from_rows2 = 0;
  a.tie_quirk = 0;
  a.x = x;
  a.x_is_half = x_is_half;
  a.lo = lo;
  a.hi = hi;
  a.outliers = outliers;
  a.outlier_idx = idx;
  a.thr_k = thr_k;
  a.C = H * hd;
  a.max_len = max_len;
  a.col = col;
  return a;
}

static AppendArgs v_args(int32_t *mat, float *lut_rows, const float *lut_sorted, const void *x, int x_is_half,
                         float *outliers, int32_t *idx, int thr_k, int H, int hd, int64_t max_len, int64_t col,
                         const kvq_vopts *norm) {
  AppendArgs a = k_args(mat, nullptr, nullptr, x, x_is_half, nullptr, nullptr, outliers, idx, thr_k, H, hd, max_len,
                        col);
  a.lut_rows = lut_rows;
  a.lut_sorted = lut_sorted;
  a.tie_quirk = 1;          // NULL options = the reference's own arithmetic (include/kvq.h: kvq_vopts)
  if (norm != nullptr) {
    a.tie_quirk = norm->reference_tie_quirk;
    if (norm->lut_rows2 != nullptr) {
      a.lut_rows2 = norm->lut_rows2;
      a.normscale = norm->normscale;
      a.normoffset = norm->normoffset;
      a.zp_from_rows2 = norm->zp_from_rows2;
    }
  }
  return a;
}

template <int NT, int E>
static int launch_pack_nt(bool is_v, int bits, const AppendArgs &a, int64_t S, hipStream_t st) {
  const int64_t nb = S < 64 ? S : (S + 7) / 8 * 8;     // (a multiple of 8: whole XCD ranges)
  dim3 grid((unsigned)nb), block(NT);
  if (is_v) {
    switch (bits) {
      case 4: fused_pack_kernel<4, true, NT, E><<<grid, block, 0, st>>>(a, S); break;
      case 3: fused_pack_kernel<3, true, NT, E><<<grid, block, 0, st>>>(a, S); break;
      case 2: fused_pack_kernel<2, true, NT, E><<<grid, block, 0, st>>>(a, S); break;
      default: return KVQ_EINVAL;
    }
  } else {
    switch (bits) {
      case 4: fused_pack_kernel<4, false, NT, E><<<grid, block, 0, st>>>(a, S); break;
      case 3: fused_pack_kernel<3, false, NT, E><<<grid, block, 0, st>>>(a, S); break;
      case 2: fused_pack_kernel<2, false, NT, E><<<grid, block, 0, st>>>(a, S); break;
      default: return KVQ_EINVAL;
    }
  }
  return check_launch();
}

static int launch_pack(bool is_v, int bits, AppendArgs a, int H, int hd, int64_t S, hipStream_t st) {
  if (S <= 0 || a.col + S > a.max_len) return KVQ_EINVAL;
  int rc = check_append(is_v, a, H, hd);
  if (rc) return rc;
  a.x_stride = S;
  if (KVQ_PACK_TILED && !a.codes_elsewhere && pack_tiled_ok(a, S)) return launch_pack_tiled(is_v, bits, a, S, st);
  if (a.C <= kPackThreads * kPackPerLane) return launch_pack_nt<kPackThreads, kPackPerLane>(is_v, bits, a, S, st);
  return launch_pack_nt<kSelThreads, kMaxPerLane>(is_v, bits, a, S, st);
}

}  // namespace kvq

using namespace kvq;

extern "C" {

int kvq_append_k_fused(int bits, int32_t *mat, const float *lut, const float *lut_off, const float *x,
                       const float *lo, const float *hi, float *outliers, int32_t *outlier_idx, int thr_k,
                       int H, int hd, int64_t max_len, int64_t col, float *outliers_t, int32_t *outlier_idx_t,
                       void *stream) {
  return launch_fused<false>(bits, k_args(mat, lut, lut_off, x, 0, lo, hi, outliers, outlier_idx, thr_k, H, hd,
                                          max_len, col, outliers_t, outlier_idx_t), H, hd, (hipStream_t)stream);
}

int kvq_append_v_fused(int bits, int32_t *mat, float *lut_rows, const float *lut_sorted, const float *x,
                       float *outliers, int32_t *outlier_idx, int thr_k, int H, int hd, int64_t max_len,
                       int64_t col, const kvq_vopts *norm, void *stream) {
  return launch_fused<true>(bits, v_args(mat, lut_rows, lut_sorted, x, 0, outliers, outlier_idx, thr_k, H, hd,
                                         max_len, col, norm), H, hd, (hipStream_t)stream);
}

int kvq_pack_k_fused(int bits, int32_t *mat, const float *lut, const float *lut_off, const float *x,
                     const float *lo, const float *hi, float *outliers, int32_t *outlier_idx, int thr_k, int H,
                     int hd, int64_t max_len, int64_t col0, int64_t S, float *outliers_t, int32_t *outlier_idx_t,
                     void *stream) {
  // two launches: outlier selection + rows (one workgroup per token, no codebook traffic) and the streaming
  // codes + pack kernel (codebook rows amortised over 256 tokens): 0.57 -> see DESIGN.md ms at S = 8192
  AppendArgs a = k_args(mat, lut, lut_off, x, 0, lo, hi, outliers, outlier_idx, thr_k, H, hd, max_len, col0,
                        outliers_t, outlier_idx_t);
  a.x_stride = S;
  if (KVQ_PACK_TILED && S > 0 && col0 + S <= max_len && check_append(false, a, H, hd) == KVQ_OK && pack_tiled_ok(a, S))
    return launch_pack_tiled(false, bits, a, S, (hipStream_t)stream);   // one pass: selection, rows AND codes
  a.codes_elsewhere = 1;
  int rc = launch_pack(false, bits, a, H, hd, S, (hipStream_t)stream);
  if (rc) return rc;
  return pack_k_codes(bits, mat, lut, x, lo, hi, H, hd, S, max_len, col0, (hipStream_t)stream);
}

int kvq_pack_v_fused(int bits, int32_t *mat, float *lut_rows, const float *lut_sorted, const float *x,
                     float *outliers, int32_t *outlier_idx, int thr_k, int H, int hd, int64_t max_len,
                     int64_t col0, int64_t S, const kvq_vopts *norm, void *stream) {
  return launch_pack(true, bits, v_args(mat, lut_rows, lut_sorted, x, 0, outliers, outlier_idx, thr_k, H, hd,
                                        max_len, col0, norm), H, hd, S, (hipStream_t)stream);
}

}  // extern "C"

namespace kvq {
// kvq_decode_prologue with the choice of score images: `pair_images` = 1: also the 3-bit fp16 pair-sum image (KTabPair3;
// 4096 gathered entries per head -- only the layers that score with it pay for it)
int decode_prologue(int bits, int32_t *kmat, const float *klut, const float *klut_off, const void *k,
                    const float *lo, const float *hi, float *koutliers, int32_t *kidx, int64_t kcol,
                    int32_t *vmat, float *vlut_rows, const float *vlut_sorted, const void *v,
                    float *voutliers, int32_t *vidx, int64_t vcol, const void *q, int acts_are_half,
                    int thr_k, int H, int hd, int64_t max_len, float *koutliers_t, int32_t *kidx_t,
                    const float *klut_ends, const float *klut_score, const kvq_vopts *vnorm,
                    const kvq_sinks *sinks, void *score_workspace, size_t score_workspace_bytes,
                    int pair_images, void *stream) {
  if (hd != kHeadDim || !q || !score_workspace || bits < 2 || bits > 4) return KVQ_EINVAL;
  if (sinks != nullptr && sinks->n_sink > 0 && (!sinks->k_sink || !sinks->sink_scores || sinks->n_sink > 1024))
    return KVQ_EINVAL;
  if (score_workspace_bytes < kvq_score_k_workspace_bytes(bits, 1, H) ||
      reinterpret_cast<uintptr_t>(score_workspace) % 16)
    return KVQ_EWORKSPACE;
  PrologueArgs P;
  P.k = k_args(kmat, klut, klut_off, k, acts_are_half, lo, hi, koutliers, kidx, thr_k, H, hd, max_len, kcol,
               koutliers_t, kidx_t);
  P.k.codes_elsewhere = 1;   // (hd == 128 here: one table workgroup per head covers every channel)
  P.k.lut_ends = klut_ends;
  P.v = v_args(vmat, vlut_rows, vlut_sorted, v, acts_are_half, voutliers, vidx, thr_k, H, hd, max_len, vcol,
               vnorm);
  int rc = check_append(false, P.k, H, hd);
  if (rc) return rc;
  rc = check_append(true, P.v, H, hd);
  if (rc) return rc;
  P.klut = klut_score ? klut_score : klut;   // table the score images are built from (Q-Norm 2 bit: ML:811-815)
  P.q = q;
  P.q_is_half = acts_are_half;
  P.tab = reinterpret_cast<unsigned char *>(score_workspace);
  const size_t tabb = bits == 4 ? KTab<4>::BUF_B : (bits == 3 ? KTab<3>::BUF_B : KTab<2>::BUF_B);
  P.q32 = reinterpret_cast<float *>(P.tab + (size_t)H * tabb);
  P.pair_tab = (bits == 3 && pair_images) ? P.tab + ktab_pair_offset<3>(1, H) : nullptr;
  P.pair_mode = pair_images;
  P.H = H;
  P.k_sink = nullptr;
  P.sink_scores = nullptr;
  P.n_sink = 0;
  P.sink_inv = 0.f;
  if (sinks != nullptr && sinks->n_sink > 0) {
    P.k_sink = reinterpret_cast<const __half *>(sinks->k_sink);
    P.sink_scores = reinterpret_cast<__half *>(sinks->sink_scores);
    P.n_sink = sinks->n_sink;
    P.sink_inv = sinks->inv_sqrt_hd;
  }
#if KVQ_TRACE
  {
    unsigned long long *tr = reinterpret_cast<unsigned long long *>(strtoull(getenv("KVQ_TRACE_PTR") ? getenv("KVQ_TRACE_PTR") : "0", nullptr, 0));
    P.k.trace = tr;
    P.v.trace = tr ? tr + 16 : nullptr;
  }
#endif
  dim3 grid(2 + H), block(kSelThreads);
  hipStream_t st = (hipStream_t)stream;
  switch (bits) {
    case 4: decode_prologue_kernel<4><<<grid, block, 0, st>>>(P); break;
    case 3: decode_prologue_kernel<3><<<grid, block, 0, st>>>(P); break;
    default: decode_prologue_kernel<2><<<grid, block, 0, st>>>(P); break;
  }
  return check_launch();
}
}  // namespace kvq

extern "C" {

// (the public entry builds every image kvq_score_k_prepared* may be asked to read)
int kvq_decode_prologue(int bits, int32_t *kmat, const float *klut, const float *klut_off, const void *k,
                        const float *lo, const float *hi, float *koutliers, int32_t *kidx, int64_t kcol,
                        int32_t *vmat, float *vlut_rows, const float *vlut_sorted, const void *v,
                        float *voutliers, int32_t *vidx, int64_t vcol, const void *q, int acts_are_half,
                        int thr_k, int H, int hd, int64_t max_len, float *koutliers_t, int32_t *kidx_t,
                        const float *klut_ends, const float *klut_score, const kvq_vopts *vnorm,
                        const kvq_sinks *sinks, void *score_workspace, size_t score_workspace_bytes,
                        void *stream) {
  return kvq::decode_prologue(bits, kmat, klut, klut_off, k, lo, hi, koutliers, kidx, kcol, vmat, vlut_rows, vlut_sorted, v,
                              voutliers, vidx, vcol, q, acts_are_half, thr_k, H, hd, max_len, koutliers_t, kidx_t, klut_ends,
                              klut_score, vnorm, sinks, score_workspace, score_workspace_bytes, true, stream);
}

/* K append | V append of one token as ONE launch (the two selection workgroups of kvq_decode_prologue without the table
 * roles): for callers that build the score tables elsewhere -- the head-sharded step appends the whole token into its
 * full-width staging column, then extracts (kvq_head_shard_step). */
int kvq_append_kv_fused(const kvq_layer *ly, int64_t col, const void *k, const void *v, int acts_are_half, void *stream) {
  if (!ly || !k || !v || col < 0 || col >= ly->max_len || ly->bits < 2 || ly->bits > 4) return KVQ_EINVAL;
  PrologueArgs P;
  P.k = k_args(ly->kmat, ly->klut, ly->klut_off, k, acts_are_half, ly->klo, ly->khi, ly->koutliers, ly->kidx, ly->thr_k,
               ly->H, ly->hd, ly->max_len, col, ly->koutliers_t, ly->kidx_t);
  P.k.lut_ends = ly->klut_ends;
  P.v = v_args(ly->vmat, ly->vlut_rows, ly->vlut_sorted, v, acts_are_half, ly->voutliers, ly->vidx, ly->thr_k, ly->H, ly->hd,
               ly->max_len, col, ly->vnorm);
  int rc = check_append(false, P.k, ly->H, ly->hd);
  if (rc) return rc;
  rc = check_append(true, P.v, ly->H, ly->hd);
  if (rc) return rc;
  P.klut = ly->klut;
  P.q = nullptr;
  P.q_is_half = 0;
  P.tab = nullptr;
  P.q32 = nullptr;
  P.pair_tab = nullptr;
  P.pair_mode = 0;
  P.H = ly->H;
  P.k_sink = nullptr;
  P.sink_scores = nullptr;
  P.n_sink = 0;
  P.sink_inv = 0.f;
#if KVQ_TRACE
  P.k.trace = nullptr;
  P.v.trace = nullptr;
#endif
  dim3 grid(2), block(kSelThreads);          // (blocks 0 and 1 of the prologue kernel: the two appends)
  hipStream_t st = (hipStream_t)stream;
  switch (ly->bits) {
    case 4: decode_prologue_kernel<4><<<grid, block, 0, st>>>(P); break;
    case 3: decode_prologue_kernel<3><<<grid, block, 0, st>>>(P); break;
    default: decode_prologue_kernel<2><<<grid, block, 0, st>>>(P); break;
  }
  return check_launch();
}

}  // extern "C"
