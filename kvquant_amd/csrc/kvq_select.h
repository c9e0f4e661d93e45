// Exact k-th order statistics of both tails of a token's keys WITHOUT histograms (round 6).  Included by
// kvq_fused_append.hip (decode append: one 1024-lane group; per-token prefill pack: 256 lanes) and kvq_pack_tiled.h (256-lane
// token groups).  Same contract as radix_select_both / group_select_both -- T[0] = key of the k-th LARGEST element, gt[0] =
// #keys > T[0]; T[1] / gt[1] for the smallest side -- plus eq[] = #keys == T[], which the membership step needs and the
// radix select left in its last histogram.
//
// Why: the radix select is four passes of LDS atomics (the first one puts a token's 4096 keys into ~10 bins: same-address
// atomics serialise) with two workgroup barriers each -- 4.4 of the decode append's 9 - 10 us, and what bounds the prefill
// pack.  Here the k-th largest is found by PRUNING, then a search over the few keys that are left:
//   1. every lane takes the maximum of its own keys; every wave extracts the j = ceil(k / waves) largest of its 64 lane
//      maxima (j wave-wide maximum reductions on DPP row operations: j = 2 for the 16 waves of the decode append, 6 for the
//      4 waves of a prefill token) and publishes the j-th one.  b = the smallest of the waves' values: waves x j >= k keys
//      of the token are >= b, so the k-th largest key T is >= b, and every key >= T -- everything the selection and its tie
//      handling can ask about -- is >= b: a CANDIDATE.  Of a token's 4096 keys ~35 (prefill) to ~80 (decode) are left;
//      barrier;
//   2. the candidates are compacted into one LDS list per side (one atomic per wave and side for the base, a wave scan for
//      the slots); barrier;
//   3. one wave per side finds T by a bitwise search on wave ballots (16 steps of two bits: three v_cmp + scalar popcounts
//      per listed key and step, no LDS); gt / eq are two ballots over the list; a barrier publishes the result.
// Both sides run in the same instruction stream (the smallest side on ~key).  A list that overflows (a token with hundreds of
// keys tied at the bound -- e.g. all-equal padding rows --, or fewer than j lanes with keys in some wave) raises a
// workgroup-uniform flag and the caller runs the radix select instead: the result is the same by construction, only the time
// differs.  tests/test_ties_gpu.py, test_decode_kv_gpu.py and the prefill suites compare the outcome bit for bit with the
// reference's kernel + glue as before.
//
// (First version of the round: per-wave k-th largest lane maximum by the bitwise search, no bound exchange -- 16 x 36 scalar
//  instructions per wave, four waves per SIMD: 5.7 us for the slowest wave, ~420 candidates: slower than the radix select,
//  profiles/r06_v_select_trace.txt.)
#pragma once
#include "kvq_common.h"

namespace kvq {

constexpr int kFselCap = 256;      // candidate slots per side (uint32 each), aliased with the radix histograms
constexpr int kFselMaxWaves = 16;

struct FselShared {
  uint32_t cand[2][kFselCap];      // [side][slot]
};
struct FselCtl {
  uint32_t wbound[2][kFselMaxWaves];   // [side][wave of the group]: its j-th largest lane maximum
  uint32_t ncand[2];                   // candidates per side (may exceed the capacity: overflow)
  uint32_t res[2][3];                  // [side] = T (as a key of the caller's order), gt, eq
};

__device__ __forceinline__ uint32_t fsel_wave_incl_scan(uint32_t v) {   // (= wave_incl_scan of kvq_fused_append.hip)
  v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x111, 0xf, 0xf, false);
  v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x112, 0xf, 0xf, false);
  v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x114, 0xf, 0xf, false);
  v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x118, 0xf, 0xf, false);
  v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x142, 0xa, 0xf, false);
  v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x143, 0xc, 0xf, false);
  return v;
}

__device__ __forceinline__ uint32_t umax32(uint32_t a, uint32_t b) { return a > b ? a : b; }

// maximum over the 64 lanes (wave-uniform): the scan's DPP steps with max instead of add
__device__ __forceinline__ uint32_t fsel_wave_max(uint32_t v) {
  v = umax32(v, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x111, 0xf, 0xf, false));
  v = umax32(v, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x112, 0xf, 0xf, false));
  v = umax32(v, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x114, 0xf, 0xf, false));
  v = umax32(v, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x118, 0xf, 0xf, false));
  v = umax32(v, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x142, 0xa, 0xf, false));
  v = umax32(v, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x143, 0xc, 0xf, false));
  return (uint32_t)__builtin_amdgcn_readlane((int)v, 63);
}

// k-th largest of V values per lane (`val` marks the slots that hold one); 0 when there are fewer than k.  Wave-uniform.
template <int V>
__device__ __forceinline__ uint32_t wave_kth_largest(const uint32_t (&v)[V], const bool (&val)[V], uint32_t k) {
  unsigned long long okm[V];
#pragma unroll
  for (int i = 0; i < V; i++) okm[i] = __ballot(val[i]);
  uint32_t t = 0;
#pragma unroll 1
  for (int bit = 30; bit >= 0; bit -= 2) {
    const uint32_t c1 = t | (1u << bit), c2 = t | (2u << bit), c3 = t | (3u << bit);
    uint32_t n1 = 0, n2 = 0, n3 = 0;
#pragma unroll
    for (int i = 0; i < V; i++) {
      n1 += (uint32_t)__popcll(__ballot(v[i] >= c1) & okm[i]);
      n2 += (uint32_t)__popcll(__ballot(v[i] >= c2) & okm[i]);
      n3 += (uint32_t)__popcll(__ballot(v[i] >= c3) & okm[i]);
    }
    t = n3 >= k ? c3 : (n2 >= k ? c2 : (n1 >= k ? c1 : t));
  }
  return (uint32_t)__builtin_amdgcn_readfirstlane((int)t);
}

// Step 1 of a group of `nw` waves (`gwave` = this wave's index in the group): publishes the wave's bounds.  The caller zeroes
// ctl.ncand[] (and its overflow flag) before the barrier it puts behind this call.
template <int E>
__device__ __forceinline__ void fsel_bounds(const uint32_t (&key)[E], const bool (&ok)[E], uint32_t k, int nw, int gwave,
                                            FselCtl &ctl) {
  const int lane = threadIdx.x & 63;
  uint32_t mh = 0, ml = 0;
#pragma unroll
  for (int e = 0; e < E; e++) {
    if (!ok[e]) continue;
    mh = umax32(mh, key[e]);
    ml = umax32(ml, ~key[e]);
  }
  const int j = ((int)k + nw - 1) / nw;
  uint32_t bh = 0, bl = 0;
  if (j <= 64) {
    for (int it = 0; it < j; it++) {
      bh = fsel_wave_max(mh);
      bl = fsel_wave_max(ml);
      // one lane that holds the maximum leaves (its other keys, if any, are candidates anyway: they are compared with the
      // bound itself)
      const unsigned long long eh = __ballot(mh == bh), el = __ballot(ml == bl);
      if (lane == (int)__builtin_ctzll(eh)) mh = 0;
      if (lane == (int)__builtin_ctzll(el)) ml = 0;
    }
  }
  if (lane == 0) {
    ctl.wbound[0][gwave] = bh;       // (0: fewer than j lanes with keys -- this wave promises nothing, everything is a candidate)
    ctl.wbound[1][gwave] = bl;
  }
}

// Step 2 (after the barrier behind fsel_bounds): the candidates of this wave -> the group's lists.  Barrier behind it.
template <int E>
__device__ __forceinline__ void fsel_collect(const uint32_t (&key)[E], const bool (&ok)[E], int nw, FselShared &sh,
                                             FselCtl &ctl) {
  const int lane = threadIdx.x & 63;
  uint32_t xh = 0xffffffffu, xl = 0xffffffffu;
  if (lane < nw) {
    xh = ctl.wbound[0][lane];
    xl = ctl.wbound[1][lane];
  }
  const uint32_t bh = ~fsel_wave_max(~xh), bl = ~fsel_wave_max(~xl);     // the smallest of the waves' bounds
  uint32_t ch = 0, cl = 0;
#pragma unroll
  for (int e = 0; e < E; e++) {
    if (!ok[e]) continue;
    ch += key[e] >= bh;
    cl += ~key[e] >= bl;
  }
  const uint32_t packed = ch | (cl << 16);                 // (<= E per lane and side: no carry between the halves)
  if (__ballot(packed != 0) == 0) return;                  // (most waves of a decode append hold no candidate at all)
  const uint32_t inc = fsel_wave_incl_scan(packed);
  const uint32_t tot = (uint32_t)__builtin_amdgcn_readlane((int)inc, 63);
  uint32_t base_h = 0, base_l = 0;
  if (lane == 0) {
    base_h = (tot & 0xffffu) ? atomicAdd(&ctl.ncand[0], tot & 0xffffu) : 0u;
    base_l = (tot >> 16) ? atomicAdd(&ctl.ncand[1], tot >> 16) : 0u;
  }
  base_h = (uint32_t)__builtin_amdgcn_readfirstlane((int)base_h);
  base_l = (uint32_t)__builtin_amdgcn_readfirstlane((int)base_l);
  uint32_t oh = base_h + ((inc - packed) & 0xffffu), ol = base_l + ((inc - packed) >> 16);
#pragma unroll
  for (int e = 0; e < E; e++) {
    if (!ok[e]) continue;
    if (key[e] >= bh) {
      if (oh < (uint32_t)kFselCap) sh.cand[0][oh] = key[e];
      oh++;
    }
    if (~key[e] >= bl) {
      if (ol < (uint32_t)kFselCap) sh.cand[1][ol] = ~key[e];
      ol++;
    }
  }
}

template <int V>
__device__ __forceinline__ void fsel_resolve_v(int side, uint32_t k, uint32_t n, const FselShared &sh, FselCtl &ctl) {
  const int lane = threadIdx.x & 63;
  uint32_t v[V];
  bool val[V];
#pragma unroll
  for (int i = 0; i < V; i++) {
    const uint32_t jx = (uint32_t)lane + 64u * i;
    val[i] = jx < n;
    v[i] = val[i] ? sh.cand[side][jx] : 0u;
  }
  const uint32_t T = wave_kth_largest<V>(v, val, k);
  uint32_t gt = 0, eq = 0;
#pragma unroll
  for (int i = 0; i < V; i++) {
    gt += (uint32_t)__popcll(__ballot(val[i] && v[i] > T));
    eq += (uint32_t)__popcll(__ballot(val[i] && v[i] == T));
  }
  if (lane == 0) {
    ctl.res[side][0] = side ? ~T : T;
    ctl.res[side][1] = gt;
    ctl.res[side][2] = eq;
  }
}

// Step 3, executed by ONE wave for side `side` (after the barrier behind fsel_collect).  Returns false on overflow (nothing
// written); otherwise ctl.res[side] = {T, gt, eq} in terms of the ORIGINAL keys.  Barrier behind it.
__device__ __forceinline__ bool fsel_resolve(int side, uint32_t k, const FselShared &sh, FselCtl &ctl) {
  const uint32_t n = (uint32_t)__builtin_amdgcn_readfirstlane((int)ctl.ncand[side]);
  if (n > (uint32_t)kFselCap || n < k) return false;
  if (n <= 64u) fsel_resolve_v<1>(side, k, n, sh, ctl);
  else if (n <= 128u) fsel_resolve_v<2>(side, k, n, sh, ctl);
  else fsel_resolve_v<kFselCap / 64>(side, k, n, sh, ctl);
  return true;
}

}  // namespace kvq
