// Prefill pack, tiled: kTT prompt tokens per workgroup, the channel-major prompt [C][S] read ONCE with 8- or 16-byte loads.
// Included by kvq_fused_append.hip (uses AppendArgs, fkey, wave_incl_scan, nearest_code, pack32).
//
// The reference runs torch.topk over [S, C] plus ~10 elementwise launches around its pack kernel
// (modeling_llama.py:879-972 for K, 1294-1382 for V).  The first fused version here gave every prompt token its own
// workgroup that read its COLUMN of the channel-major input -- 4-byte loads S*4 bytes apart, every 64-byte sector
// fetched by 16 neighbouring tokens -- and (K) a second streaming kernel for the codes that read the prompt again:
// 7.5x the algorithmic bytes through the memory pipe, 5 % of the HBM rate at S = 8192.  Here:
//   phase 1 (1024 lanes, lane = channel): one float4 = the channel's values of the workgroup's four tokens (the eight
//     workgroups that share a 128-byte line are dispatched back to back on one XCD and meet in its L2), K: the
//     channel's codebook row -> four codes, thresholds -> four rescaled values; values / rescaled values and codes
//     go to LDS, transposed to [token][channel];
//   phase 2 (four groups of 256 lanes, group = token): exact top-k of both tails by radix select on the token's 4096
//     keys from LDS (the barriers are workgroup-wide, the four groups run the same steps), membership with ties
//     broken by lowest channel, V: per-token thresholds / codebook row / codes, compaction in channel order, outlier
//     row (+ K mirror); K residuals re-read the 42 selected values from the prompt (L2-hot);
//   phase 3: codes -> packed words, the four tokens of a row next to each other (16-byte pieces).
// Results are bit-identical to the per-token kernel (tests/test_atsize_gpu.py, test_decode_kv_gpu.py).
//
// Tokens per workgroup (round 5): the kernel is a chain of barrier-separated steps (four radix passes, scans, compaction),
// i.e. bound by latencies that only ANOTHER workgroup on the CU can fill.  Four tokens need 150 KB of LDS -- one workgroup per
// CU; TWO tokens with one histogram copy need 50 KB -- three workgroups of 512 lanes per CU: 8192 tokens nuq4 K 274 -> 211 us,
// V 246 -> 180 (two tokens with two copies, two workgroups per CU: 244 / 209; ONE token, 4-byte loads, 4 - 5 workgroups per CU:
// 305 / 255 -- profiles/r05_pack_tokens.txt).  The price is the K codebook row of a channel read per two tokens instead of
// four (L2).
#pragma once

namespace kvq {

#ifndef KVQ_PACK_TT
#define KVQ_PACK_TT 2
#endif
#ifndef KVQ_PACK_THC
#define KVQ_PACK_THC 1
#endif
constexpr int kTT = KVQ_PACK_TT;   // tokens per workgroup (4 or 2)
static_assert(kTT == 4 || kTT == 2 || kTT == 1, "the prompt is read as one 16-, 8- or 4-byte piece per channel");
constexpr int kTG = 256;        // lanes per token group
constexpr int kTNT = kTT * kTG;
constexpr int kTE = 16;         // channels per lane in phase 2: C <= 4096
constexpr int kTC = kTG * kTE;  // 4096

constexpr int kTHC = KVQ_PACK_THC;   // copies of a selection histogram, lanes spread over them (same-address LDS atomics
                                     // serialise: the first pass puts a token's 4096 keys into ~10 bins).  Four copies
                                     // changed nothing measurable at four tokens per workgroup; one copy is what lets three
                                     // workgroups share a CU at two tokens
struct TiledSel {
  uint32_t hist[2][2][kTHC][256];     // [pass parity][side][copy][digit]; the pruning select's candidate lists alias it
  FselCtl fctl;
  uint32_t prefix[2], krem[2];
  uint32_t scan[kTG / 64];
  float vrow[16], vrow2[16];
};

struct TiledShared {
  float ss[kTT][kTC + kTC / 32];          // K: rescaled values, V: values; channel c at c + c/32
  unsigned char cb[kTT][kTC + 16];        // codes (+16: the tokens of a 32-channel group in different banks for the pack's reads)
  TiledSel sel[kTT];
  int any_cut;
  uint32_t fallback;                      // a token group's candidate list overflowed: every group takes the radix select
};
static_assert(sizeof(FselShared) <= sizeof(uint32_t) * 2 * 2 * kTHC * 256, "the candidate lists alias the histograms");

// exclusive scan over the kTG lanes of a token group (whole-workgroup barriers: every group calls it together)
__device__ __forceinline__ uint32_t group_excl_scan(uint32_t v, uint32_t *ws, int tg, uint32_t &total) {
  const int lane = tg & 63, wave = tg >> 6;
  const uint32_t inc = wave_incl_scan(v);
  __syncthreads();
  if (lane == 63) ws[wave] = inc;
  __syncthreads();
  uint32_t base = 0, tot = 0;
#pragma unroll
  for (int w = 0; w < kTG / 64; w++) {
    const uint32_t s = ws[w];
    if (w < wave) base += s;
    tot += s;
  }
  total = tot;
  return base + inc - v;
}

// radix_select_both for one token group (see there): T[0] = key of the k-th largest, gt[0] = #keys > T[0]; T[1] / gt[1]
// for the smallest side.  Ends with the final histograms (parity 0) intact.
__device__ __forceinline__ void group_select_both(const uint32_t (&key)[kTE], const bool (&ok)[kTE], uint32_t k,
                                                  TiledSel &sh, int tg, uint32_t (&T)[2], uint32_t (&gt)[2]) {
  if (tg < 2) {
    sh.prefix[tg] = 0;
    sh.krem[tg] = k;
  }
  for (int i = tg; i < 512 * kTHC; i += kTG) (&sh.hist[1][0][0][0])[i] = 0;   // (the first pass is pass 3: parity 1)
  __syncthreads();
  for (int pass = 3; pass >= 0; pass--) {
    uint32_t (*hist)[kTHC][256] = sh.hist[pass & 1];
    const int cp = tg & (kTHC - 1);
    const uint32_t p0 = sh.prefix[0], p1 = sh.prefix[1];
#pragma unroll
    for (int e = 0; e < kTE; e++) {
      if (!ok[e]) continue;
      const uint32_t kk = key[e];
      const uint32_t hi = (pass == 3) ? 0u : (kk >> (8 * (pass + 1)));
      const uint32_t d = (kk >> (8 * pass)) & 0xffu;
      if (hi == p0) atomicAdd(&hist[0][cp][d], 1u);                   // (first pass: both sides count every element)
      if (pass != 3 && hi == p1) atomicAdd(&hist[1][cp][d], 1u);
    }
    __syncthreads();
    const int wave = tg >> 6, lane = tg & 63;
    if (wave < 2) {
      const int side = wave;
      uint32_t c[4];
      uint32_t s = 0;
#pragma unroll
      for (int j = 0; j < 4; j++) {
        const int pos = lane * 4 + j;
        const int bin = side == 0 ? 255 - pos : pos;
        c[j] = 0;
#pragma unroll
        for (int q = 0; q < kTHC; q++) c[j] += hist[pass == 3 ? 0 : side][q][bin];
        s += c[j];
      }
      const uint32_t inc = wave_incl_scan(s);
      const uint32_t before = inc - s;
      const uint32_t kr = sh.krem[side];
      if (before < kr && kr <= inc) {
        uint32_t run = before;
#pragma unroll
        for (int j = 0; j < 4; j++) {
          if (run < kr && kr <= run + c[j]) {
            const int pos = lane * 4 + j;
            const int bin = side == 0 ? 255 - pos : pos;
            sh.prefix[side] = (sh.prefix[side] << 8) | (uint32_t)bin;
            sh.krem[side] = kr - run;
          }
          run += c[j];
        }
      }
    } else if (pass > 0) {
      for (int i = tg - 128; i < 512 * kTHC; i += kTG - 128) (&sh.hist[(pass - 1) & 1][0][0][0])[i] = 0;
    }
    __syncthreads();
  }
  T[0] = sh.prefix[0];
  T[1] = sh.prefix[1];
  gt[0] = k - sh.krem[0];
  gt[1] = k - sh.krem[1];
}

template <int BITS, bool IS_V>
__global__ __launch_bounds__(kTNT) __attribute__((amdgpu_waves_per_eu(6, 6))) void pack_tiled_kernel(AppendArgs A, int64_t S) {
  constexpr int N = Fmt<BITS>::kN;
  __shared__ __attribute__((aligned(16))) TiledShared sh;
  const int tid = threadIdx.x;
  const int C = A.C, thr_k = A.thr_k;
  const int64_t max_len = A.max_len;
  // token groups are handed to the XCDs in contiguous ranges (workgroup b runs on XCD b % 8): the 32 / kTT groups that
  // share each 128-byte line of the channel-major prompt then go through one L2, dispatched back to back
  const int64_t nb = gridDim.x, nq = (S + kTT - 1) / kTT;
  const int64_t per_xcd = (nb + 7) / 8;
  int64_t quad = (int64_t)(blockIdx.x % 8) * per_xcd + blockIdx.x / 8;
  if (nb < 64) quad = blockIdx.x;
  if (quad >= nq) return;
  const int64_t s0 = quad * kTT;
  const float *xq = reinterpret_cast<const float *>(A.x) + s0;         // element (c, t): xq[c * S + t]

  // ---- phase 1: lane = channel -----------------------------------------------------------------------------------
  for (int c = tid; c < kTC; c += kTNT) {
    float xv[kTT];
#pragma unroll
    for (int t = 0; t < kTT; t++) xv[t] = 0.f;
    if (c < C) {
      if constexpr (kTT == 4) {
        const float4 v4 = *reinterpret_cast<const float4 *>(xq + (int64_t)c * S);   // (S % 4 == 0, 16-byte aligned base)
        xv[0] = v4.x; xv[1] = v4.y; xv[2] = v4.z; xv[3] = v4.w;
      } else if constexpr (kTT == 2) {
        const float2 v2 = *reinterpret_cast<const float2 *>(xq + (int64_t)c * S);
        xv[0] = v2.x; xv[1] = v2.y;
      } else {
        xv[0] = xq[(int64_t)c * S];
      }
    }
    const int cp = c + (c >> 5);
    if constexpr (!IS_V) {
      float row[N];
      float zp = 0.f, rg = 1.f;
      if (c < C) {
        const float *src = A.lut + (int64_t)c * N;
#pragma unroll
        for (int v = 0; v < N; v += 4) {
          const float4 t = *reinterpret_cast<const float4 *>(src + v);
          row[v] = t.x; row[v + 1] = t.y; row[v + 2] = t.z; row[v + 3] = t.w;
        }
        const float l = A.lo[c], h = A.hi[c];
        rg = (h - l) / 2;                       // KCU:1759-1764
        zp = (h + l) / 2;
      }
#pragma unroll
      for (int t = 0; t < kTT; t++) {
        sh.ss[t][cp] = (c < C) ? (xv[t] - zp) / rg : 0.f;
        sh.cb[t][c] = (c < C) ? (unsigned char)nearest_code<N>(row, xv[t]) : 0;
      }
    } else {
#pragma unroll
      for (int t = 0; t < kTT; t++) sh.ss[t][cp] = xv[t];
    }
  }
  if (tid == 0) {
    sh.any_cut = 0;
    sh.fallback = 0;
  }
  if (tid < 2 * kTT) sh.sel[tid >> 1].fctl.ncand[tid & 1] = 0;
  __syncthreads();

  // ---- phase 2: group = token ------------------------------------------------------------------------------------
  const int g = tid / kTG, tg = tid % kTG;
  // (a token past the end of the prompt -- S % 4 != 0 is not dispatched here, but keep the groups in step anyway --
  //  repeats the last real one: identical stores)
  const int64_t s = (s0 + g < S) ? s0 + g : S - 1;
  const int gt_ = (int)(s - s0);                  // LDS row of this group's token
  const int64_t col = A.col + s;
  TiledSel &ts = sh.sel[g];
  const int per = (C + kTG - 1) / kTG;            // <= kTE
  const int c0 = tg * per;
  float sel[kTE];
  bool ok[kTE];
#pragma unroll
  for (int e = 0; e < kTE; e++) {
    ok[e] = e < per && (c0 + e) < C;
    const int c = ok[e] ? c0 + e : 0;
    sel[e] = ok[e] ? sh.ss[gt_][c + (c >> 5)] : 0.f;
  }
  // The selection keys are recomputed from the values wherever they are needed (two instructions each) instead of living
  // next to them through the whole phase: 16 registers of a kernel that is at the limit for three workgroups per CU.  The
  // empty asm makes the values opaque between the uses, so that the compiler does not merge the recomputations again.
  auto forget_keys = [&]() {
#pragma unroll
    for (int e = 0; e < kTE; e++) asm volatile("" : "+v"(sel[e]));
  };
  uint32_t T[2], gtc[2];
  const uint32_t ksel = IS_V ? (uint32_t)(thr_k + 1) : (uint32_t)thr_k;
  uint32_t eq_hi = 0, eq_lo = 0;
  bool radix = !KVQ_FAST_SELECT;
  if constexpr (KVQ_FAST_SELECT != 0) {
    // pruning select (kvq_select.h; round 6): three barriers instead of the radix chain's nine, no histogram atomics
    FselShared &fs = reinterpret_cast<FselShared &>(ts.hist);
    {
      uint32_t key[kTE];
#pragma unroll
      for (int e = 0; e < kTE; e++) key[e] = fkey(sel[e]);
      fsel_bounds<kTE>(key, ok, ksel, kTG / 64, tg >> 6, ts.fctl);
      __syncthreads();
      fsel_collect<kTE>(key, ok, kTG / 64, fs, ts.fctl);
    }
    forget_keys();
    __syncthreads();
    if (tg < 128) {
      if (!fsel_resolve(tg >> 6, ksel, fs, ts.fctl)) sh.fallback = 1;
    }
    __syncthreads();
    radix = sh.fallback != 0;            // (workgroup-uniform: the radix select has barriers inside)
    if (!radix) {
      T[0] = ts.fctl.res[0][0]; gtc[0] = ts.fctl.res[0][1]; eq_hi = ts.fctl.res[0][2];
      T[1] = ts.fctl.res[1][0]; gtc[1] = ts.fctl.res[1][1]; eq_lo = ts.fctl.res[1][2];
    }
  }
  if (radix) {
    uint32_t key[kTE];
#pragma unroll
    for (int e = 0; e < kTE; e++) key[e] = fkey(sel[e]);
    group_select_both(key, ok, ksel, ts, tg, T, gtc);
    forget_keys();
    // membership: strictly beyond the threshold, plus the first ties in channel order (fused_append_body)
#pragma unroll
    for (int q = 0; q < kTHC; q++) {
      eq_hi += ts.hist[0][0][q][T[0] & 0xffu];
      eq_lo += ts.hist[0][1][q][T[1] & 0xffu];
    }
  }
  const uint32_t want_hi = (uint32_t)thr_k - gtc[0], want_lo = (uint32_t)thr_k - gtc[1];
  const bool cut = !((want_hi == 0 || eq_hi == want_hi) && (want_lo == 0 || eq_lo == want_lo));   // (group-uniform)
  if (cut && tg == 0) sh.any_cut = 1;
  __syncthreads();
  uint32_t rank_hi = 0, rank_lo = 0;
  if (sh.any_cut) {                               // (workgroup-uniform: the scan has barriers inside)
    uint32_t ntie_hi = 0, ntie_lo = 0;
#pragma unroll
    for (int e = 0; e < kTE; e++) {
      if (!ok[e]) continue;
      const uint32_t ke = fkey(sel[e]);
      ntie_hi += ke == T[0];
      ntie_lo += ke == T[1];
    }
    forget_keys();
    uint32_t tot;
    const uint32_t packed = group_excl_scan(ntie_hi | (ntie_lo << 16), ts.scan, tg, tot);
    rank_hi = packed & 0xffffu;
    rank_lo = packed >> 16;
  }
  // (membership as per-lane bit masks -- bit e = element e -- instead of 2 x 16 predicates: those live in scalar register
  //  pairs, 64 of the ~100 a wave has, and spilled into vector registers once the V code search wanted its midpoints there)
  uint32_t m_hi = 0, m_lo = 0;
  uint32_t nsel = 0;
#pragma unroll
  for (int e = 0; e < kTE; e++) {
    if (!ok[e]) continue;
    bool ih = false, il = false;
    const uint32_t ke = fkey(sel[e]);
    if (ke > T[0]) ih = true;
    else if (ke == T[0]) { ih = rank_hi < want_hi; rank_hi++; }
    if (ke < T[1]) il = true;
    else if (ke == T[1]) { il = rank_lo < want_lo; rank_lo++; }
    m_hi |= ih ? (1u << e) : 0u;
    m_lo |= il ? (1u << e) : 0u;
    nsel += (ih || il);
  }

  float vmin = 0.f, vmax = 0.f, zpv = 0.f;
  if constexpr (IS_V) {
    const uint32_t bh = T[0] ^ ((T[0] >> 31) ? 0x80000000u : 0xffffffffu);
    const uint32_t bl = T[1] ^ ((T[1] >> 31) ? 0x80000000u : 0xffffffffu);
    vmax = __uint_as_float(bh);
    vmin = __uint_as_float(bl);
    const float offset = (vmax + vmin) / 2;    // modeling_llama.py:1097-1098 (fp32)
    const float sf = (vmax - vmin) / 2;
    if (tg < N) {
      const float r = A.lut_sorted[tg] * sf + offset;   // two roundings (-ffp-contract=off), ML:1113
      ts.vrow[tg] = r;
      A.lut_rows[col * N + tg] = r;
      if (A.lut_rows2 != nullptr) {
        const float r2 = (A.lut_sorted[tg] * A.normscale + A.normoffset) * sf + offset;
        ts.vrow2[tg] = r2;
        A.lut_rows2[col * N + tg] = r2;
      }
    }
    __syncthreads();
    zpv = (A.lut_rows2 != nullptr && A.zp_from_rows2) ? ts.vrow2[Fmt<BITS>::kZeroCode] : ts.vrow[Fmt<BITS>::kZeroCode];
    float row[N];
#pragma unroll
    for (int v = 0; v < N; v++) row[v] = ts.vrow[v];
    // (the lane's 16 codes leave as four words when its channels are 16 whole ones -- C = 4096 --: byte stores of
    //  neighbouring lanes 16 bytes apart are a 4-way bank conflict each)
    uint32_t cw[kTE / 4];
#pragma unroll
    for (int i = 0; i < kTE / 4; i++) cw[i] = 0;
#pragma unroll
    for (int e = 0; e < kTE; e++) {
      if (!ok[e]) continue;
      const bool clip = A.tie_quirk ? (sel[e] < vmin || sel[e] > vmax) : ((((m_hi | m_lo) >> e) & 1u) != 0);
      const unsigned code = clip ? Fmt<BITS>::kZeroCode : nearest_code<N>(row, sel[e]);
      if (per == kTE) cw[e >> 2] |= code << (8 * (e & 3));
      else sh.cb[gt_][c0 + e] = (unsigned char)code;
    }
    if (per == kTE) {
      const bool full = c0 + kTE <= C;
      if (full) {
        *reinterpret_cast<uint4 *>(&sh.cb[gt_][c0]) = make_uint4(cw[0], cw[1], cw[2], cw[3]);
      } else {
#pragma unroll
        for (int e = 0; e < kTE; e++)
          if (ok[e]) sh.cb[gt_][c0 + e] = (unsigned char)(cw[e >> 2] >> (8 * (e & 3)));
      }
    }
  }

  // outlier row: compaction in channel order
  uint32_t tot2;
  uint32_t pos = group_excl_scan(nsel, ts.scan, tg, tot2);   // (its barriers also publish the V codes)
  const int n_out = 2 * thr_k;
  float *orow = A.outliers + col * n_out;
  int32_t *irow = A.outlier_idx + col * n_out;
  const float *xs_ = reinterpret_cast<const float *>(A.x) + s;
  int c0_late = c0;
  asm volatile("" : "+v"(c0_late));     // (the channel ids are re-derived here: kept from the top of the phase they were spilled)
#pragma unroll
  for (int e = 0; e < kTE; e++) {
    if (!ok[e] || !(((m_hi | m_lo) >> e) & 1u)) continue;
    const int c = c0_late + e;
    float val;
    if constexpr (IS_V) {
      val = sel[e] - zpv;                                  // modeling_llama.py:1169
    } else {
      // residual to the saturated end point; zero when the rescaled value is inside [-1, 1] (ML:729-747)
      if ((m_hi >> e) & 1u) val = (sel[e] <= 1.0f) ? 0.f : xs_[(int64_t)c * S] - A.lut_off[(int64_t)c * N + (N - 1)];
      else val = (sel[e] >= -1.0f) ? 0.f : xs_[(int64_t)c * S] - A.lut_off[(int64_t)c * N];
    }
    if ((int)pos < n_out) {
      if (A.outliers != nullptr) {
        orow[pos] = val;
        irow[pos] = c;
      } else if (A.outlier_idx != nullptr) {
        irow[pos] = pack_entry(val, c);                       // compact rows
      }
      if constexpr (!IS_V) {
        if (A.outliers_t != nullptr) {
          A.outliers_t[(int64_t)pos * max_len + col] = val;
          A.outlier_idx_t[(int64_t)pos * max_len + col] = c;
        } else if (A.outlier_idx_t != nullptr) {
          A.outlier_idx_t[(int64_t)pos * max_len + col] = pack_entry(val, c);   // compact mirror
        }
      }
    }
    pos++;
  }

  // ---- phase 3: pack.  Unit = (32-channel group, token), the four tokens of a group on neighbouring lanes -----------
  for (int u = tid; u < (C / 32) * kTT; u += kTNT) {
    const int grp = u / kTT, t = u % kTT;
    if (s0 + t >= S) continue;
    unsigned cd[32];
    const uint4 *src = reinterpret_cast<const uint4 *>(&sh.cb[t][grp * 32]);
    const uint4 a0 = src[0], a1 = src[1];
    const uint32_t wsrc[8] = {a0.x, a0.y, a0.z, a0.w, a1.x, a1.y, a1.z, a1.w};
#pragma unroll
    for (int i = 0; i < 32; i++) cd[i] = (wsrc[i >> 2] >> (8 * (i & 3))) & 0xffu;
    uint32_t w[BITS];
    pack32<BITS>(cd, w);
#pragma unroll
    for (int i = 0; i < BITS; i++) A.mat[((int64_t)grp * BITS + i) * max_len + A.col + s0 + t] = w[i];
  }
}

// shapes the tiled kernel takes: fp32 prompt, C <= 4096 and a multiple of 32, S a multiple of 4, 16-byte aligned columns
static bool pack_tiled_ok(const AppendArgs &a, int64_t S) {
  return !a.x_is_half && a.C <= kTC && a.C % 32 == 0 && S % kTT == 0 && reinterpret_cast<uintptr_t>(a.x) % 16 == 0 &&
         (a.lut_off != nullptr || a.lut_sorted != nullptr);
}

static int launch_pack_tiled(bool is_v, int bits, const AppendArgs &a, int64_t S, hipStream_t st) {
  const int64_t nq = (S + kTT - 1) / kTT;
  const int64_t nb = nq < 64 ? nq : (nq + 7) / 8 * 8;
  dim3 grid((unsigned)nb), block(kTNT);
  if (is_v) {
    switch (bits) {
      case 4: pack_tiled_kernel<4, true><<<grid, block, 0, st>>>(a, S); break;
      case 3: pack_tiled_kernel<3, true><<<grid, block, 0, st>>>(a, S); break;
      case 2: pack_tiled_kernel<2, true><<<grid, block, 0, st>>>(a, S); break;
      default: return KVQ_EINVAL;
    }
  } else {
    switch (bits) {
      case 4: pack_tiled_kernel<4, false><<<grid, block, 0, st>>>(a, S); break;
      case 3: pack_tiled_kernel<3, false><<<grid, block, 0, st>>>(a, S); break;
      case 2: pack_tiled_kernel<2, false><<<grid, block, 0, st>>>(a, S); break;
      default: return KVQ_EINVAL;
    }
  }
  return check_launch();
}

}  // namespace kvq
