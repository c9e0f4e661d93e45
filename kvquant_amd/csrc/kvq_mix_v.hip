// softmax(p).V over the packed NUQ value cache (per-token codebooks), fused
// with the fixed-width sparse-outlier SpMV.  Reference semantics:
// KCU:3211-3433 (+4117-4491, 4998-5248), SPMV_ATOMIC_BALANCED KCU:437-470,
// launchers KCU:3491-3538, 3625-3690.
//
// CDNA4 design.  The reduction runs over tokens, and the cache is laid out
// [row][token] with the token axis contiguous.  The reference puts a token on
// every thread, transposes 32 KB of products through LDS per block and issues
// 128 atomics per block.  Here:
//   * HBM side: the packed rows are streamed with LDS-DMA (global_load_lds, 16 B
//     per lane): every wave instruction moves whole 64/128-byte row segments
//     straight into an LDS tile, no VGPR staging, one chunk ahead of the math
//     (double buffered, one barrier per chunk);
//   * LDS side: a LANE OWNS A ROW UNIT (one packed word-row = 8 / 16 channels for
//     4 / 2 bit, three rows = 32 channels for 3 bit) and reads its own row back
//     with ds_read_b128.  The tile is stored with the 16-byte quads of row r
//     rotated by f(r) (done on the DMA's per-lane SOURCE address, the LDS image
//     of a DMA instruction is linear by construction) so those reads are
//     bank-conflict free;
//   * all lanes of a wave decode the SAME token at the same time, so the
//     per-token codebook row is a broadcast, conflict-free ds_read_b32 per code
//     (2 address ops + 1 FMA per code), and the 8..32 channel sums of a unit
//     stay in VGPRs for the whole token range: no cross-lane reduction, no
//     transposes, no atomics;
//   * a workgroup = 512 lanes = UW units x SLOTS token slots over one token
//     range; slots are summed through LDS once at the end and every workgroup
//     writes its channel slice of one partial slab; a second small kernel adds
//     the slabs into `mul` in a fixed order (deterministic);
//   * the sparse residuals of the range that fall into the workgroup's channel
//     slice are scattered into an LDS accumulator (32.32 fixed-point ds_add_u64:
//     exact, order independent) before or after its dense loop and added to the
//     dense sums in registers: one partial slab per workgroup, nothing else.
// Needs max_len % 4 == 0 (16-byte DMA source alignment); other shapes take the
// row-per-lane fallback in kvq_mix_v_rows.h.
// Algorithmic HBM bytes per cached token: C*bits/8 + 4*2^bits (codebook row)
// + 4*H (probabilities) (+ 8*n_out sparse).
#include "kvq_common.h"
#include "kvq_host.h"
#include "kvq_mix_v_stage.h"
#include "kvq_mix_lut.h"

#include <hip/hip_fp16.h>

#include <cstdlib>
#define KVQ_V_WGS 512           // workgroups the plan aims at (2 per CU)
#define KVQ_V_WAVES 4           // waves per SIMD the register allocation aims at
#define KVQ_V_MERGE_PARTS 256   // fused softmax: up to this many score tiles (64K tokens) the p.V workgroups merge the partials themselves
#ifndef KVQ_V_WIDE_FROM
#define KVQ_V_WIDE_FROM 6144    // cached tokens from which the 1024-lane geometry (kvq_mix_v_wide.hip) takes a decode step's p.V
                                // (same box, profiles/r06_k_misc.txt: 4K 1.757 vs 1.794 ms/step for the wide one, 8K 1.955 vs 1.92, 12K 1.99 vs 1.90)
#endif
#ifndef KVQ_W_MERGE_PARTS
#define KVQ_W_MERGE_PARTS 256   // wide p.V: up to this many score tiles (64K tokens) the workgroups merge the softmax partials themselves
                                // (1024 measured: the merge in every workgroup's prologue costs what the launch costs -- 128K 5.56 vs 5.67 ms/step,
                                //  profiles/r06_n_merge.txt; with 16 or 32 partial loads per lane in one round trip: the same, r06_r_merge_batch.txt --
                                //  the merge runs behind the wait for the first tile, not beside it)
#endif
#ifndef KVQ_V_RB
#define KVQ_V_RB 24             // outlier phase: entries per lane and token block (one round of loads)
#endif
// (the experiment switches of rounds 2-3 -- wave priorities, 16-token chunks at 4 bit, the C++ look-up loops next to the
//  hand-scheduled ones, DMA in one burst, the development ablations -- were measured neutral or slower and are gone;
//  DESIGN.md 3 keeps the numbers)

namespace kvq {

// bytes / 4 of `ns` whole outlier rows
__device__ __forceinline__ unsigned nent_row(int ns, int n_out) { return (unsigned)ns * (unsigned)n_out; }
// buffer descriptor over [p, p + bytes) from values that ARE wave-uniform, made provably so (readfirstlane of the
// descriptor's inputs: otherwise hipcc wraps every buffer load in a waterfall loop -- cdna_hip_programming.md T8 / T20)
__device__ __forceinline__ __amdgpu_buffer_rsrc_t uniform_rsrc(const void *p, unsigned bytes) {
  const uint64_t u = reinterpret_cast<uint64_t>(p);
  const uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)u), hi = __builtin_amdgcn_readfirstlane((uint32_t)(u >> 32));
  return __builtin_amdgcn_make_buffer_rsrc(reinterpret_cast<void *>(((uint64_t)hi << 32) | lo), 0,
                                           (int)__builtin_amdgcn_readfirstlane(bytes), 0x00020000);
}

// FUSED: the second softmax pass (kvq_softmax_finish) happens inside this kernel -- the workgroups merge the
// (max, sum) partials of the score kernel themselves and convert raw scores to probabilities in LDS, one chunk
// ahead of the look-up loop; the [H][L] probabilities never exist in memory (saves a launch, 13 us and 2 x 4 bytes
// per token and head of traffic at 128K).
template <int BITS, bool FUSED>
__global__ __launch_bounds__(512, KVQ_V_WAVES) void mix_v_kernel(MixArgs a) {
  using Cfg = VCfg<BITS>;
  constexpr int N = Cfg::N, CH = Cfg::CH, WORDS = Cfg::WORDS, CT = Cfg::CT, CHL = Cfg::CHL;
  __shared__ __attribute__((aligned(16))) unsigned char smem[Cfg::SMEM_B + (FUSED ? Cfg::MZ_B : 0)];  // static: LDS offsets fold into ds immediates
  const int tid = threadIdx.x;
  const int ul = tid % Cfg::UW;          // unit within the workgroup
  const int hf = __builtin_amdgcn_readfirstlane((tid / Cfg::UW) % Cfg::HALVES);   // which half of the unit's channels
  const int lu = tid % (Cfg::UW * Cfg::HALVES);                                   // lane within the slot
  const int sl = tid / (Cfg::UW * Cfg::HALVES);   // token slot (wave-uniform)
  // block -> (range, unit group).  The workgroups of one range share its codebook rows and its outlier entries: they are
  // numbered 8 apart, so that they run on the same XCD (block b goes to XCD b % 8) back to back and the second one
  // finds those lines in that XCD's L2.  (Last, partial chunk of ranges: any bijection.)
  int g, range;
  {
    const int G = a.groups, n_ranges = (int)gridDim.x / G;
    const int chunk = (int)blockIdx.x / (8 * G), r = (int)blockIdx.x % (8 * G);
    const int nr = (n_ranges - 8 * chunk < 8) ? (n_ranges - 8 * chunk) : 8;
    g = r / nr;
    range = 8 * chunk + r % nr;
  }
  const int b = blockIdx.z;
  const int C = a.H * kHeadDim;
  const int u0 = g * Cfg::UW;
  const int row_base = u0 * WORDS;
  int n_units_valid = a.n_units - u0;
  if (n_units_valid > Cfg::UW) n_units_valid = Cfg::UW;
  const int n_rows_valid = n_units_valid * WORDS;
  const int h0 = u0 / Cfg::UPH;
  const int hl = ul / Cfg::UPH;
  const int64_t t0 = (int64_t)range * a.tr;
  const int64_t t1 = (t0 + a.tr < a.L) ? (t0 + a.tr) : a.L;
  const int n_chunks = (int)((t1 - t0 + CT - 1) / CT);

  const uint32_t lds0 = lds_addr(smem);
  DmaLane dl = make_dma_lane<BITS>();
  const bool sparse = (a.idx != nullptr) && (b == 0);        // reference: batch 0 only (KCU:3675)
  // COMPACT rows (opt-in format): no value array, `idx` holds packed entries fp16 residual << 16 | channel
  const bool compact = a.outliers == nullptr;
  // The outlier phase is bound by memory latency (VALU and LDS idle), the dense loop by the look-ups.  Half of the
  // workgroups run it BEFORE their dense loop, the other half after it: wherever two workgroups share a CU in either
  // order, one's latency-bound phase hides behind the other's look-ups (nuq3 p.V + reduce at 128K: 84 us alternating,
  // 91 us with every phase at the end).  The groups of a range (blocks 8 apart) have the same parity: they read the
  // range's entries at about the same time.
  // (round 6: long caches run kvq_mix_v_wide.hip, whose outlier entries ride in the loop; what is left for this kernel --
  //  short caches, q_len > 1 -- keeps the phase behind the loop in every workgroup: the alternating order bought nothing
  //  there and cost the non-fused 2 / 3-bit instantiations 8 spilled VGPRs)
  constexpr bool sparse_first = false;
  if (!sparse_first) issue_chunk<BITS>(a, dl, lds0, 0, t0, row_base, n_rows_valid, h0, b, !FUSED);
  const float2 *mz = reinterpret_cast<const float2 *>(smem + Cfg::SMEM_B);   // FUSED: (max, normaliser) per head
  float myM = 0.f, myZ = 1.f;                                                  // ... of the head this lane converts for
  if constexpr (FUSED) {
    // raw scores of the first two chunks -> p buffers 0 and 1 (chunk c lives in buffer c % 3)
    issue_p<BITS>(a.scores, a, dl, lds0 + Cfg::p_off(0), t0, h0, b);
    if (n_chunks > 1) issue_p<BITS>(a.scores, a, dl, lds0 + Cfg::p_off(1), t0 + CT, h0, b);
    // (max, normaliser) of every head -> LDS (the sparse phase needs all of them), this lane's head -> registers
    float2 *mzw = reinterpret_cast<float2 *>(smem + Cfg::SMEM_B);
    if (a.mz != nullptr) {
      for (int h = tid; h < a.H; h += Cfg::NT) {
        const float2 t = reinterpret_cast<const float2 *>(a.mz)[h];
        mzw[h] = make_float2(t.x, 1.0f / t.y);     // (max, 1 / normaliser)
      }
    } else {
      // The workgroup merges the partials of the heads it needs itself (+ the fp16 sink scores), the same way in every
      // workgroup: its own unit group's heads -- 32 lanes per head -- or, for the workgroup that writes the sink
      // probabilities, all of them.  The partials are read 16 per lane at a time, all loads of a batch in flight together
      // (one load per merge step made this 11 us per workgroup at 128K).
      const bool all_heads = blockIdx.x == 0 && a.n_sink > 0;
      const int hfirst = all_heads ? 0 : h0;
      int hcount = all_heads ? a.H : (n_units_valid + Cfg::UPH - 1) / Cfg::UPH;
      if (hfirst + hcount > a.H) hcount = a.H - hfirst;
      constexpr int LPH = 32;                                     // lanes per head: the same partition in every workgroup
      const int sub = tid & (LPH - 1);
      constexpr int MB = 16;                                      // partials per lane and batch
      for (int hb = 0; hb < hcount; hb += Cfg::NT / LPH) {
        const int hr = hb + tid / LPH;
        const bool hv = hr < hcount;
        const int h = hfirst + (hv ? hr : 0);
        const float2 *pr = reinterpret_cast<const float2 *>(a.parts) + (int64_t)h * a.n_parts;
        float M = -INFINITY, Z = 0.f;
        for (int i0 = sub; i0 < a.n_parts; i0 += LPH * MB) {
          float2 ms[MB];
#pragma unroll
          for (int k = 0; k < MB; k++) {
            const int i = i0 + LPH * k;
            ms[k] = (hv && i < a.n_parts) ? pr[i] : make_float2(-INFINITY, 0.f);
          }
#pragma unroll
          for (int k = 0; k < MB; k++)
            if (ms[k].x > -INFINITY) {
              const float mn = fmaxf(M, ms[k].x);
              Z = Z * mz_w(M - mn) + ms[k].y * mz_w(ms[k].x - mn);
              M = mn;
            }
        }
        if (hv)
          for (int i = sub; i < a.n_sink; i += LPH) {
            const float x = __half2float(a.sink[h * a.n_sink + i]);
            const float mn = fmaxf(M, x);
            Z = Z * expf(M - mn) + expf(x - mn);
            M = mn;
          }
        for (int d = LPH / 2; d >= 1; d >>= 1) {
          const float mo = __shfl_xor(M, d), zo = __shfl_xor(Z, d);
          const float mn = fmaxf(M, mo);
          Z = (mn == -INFINITY) ? 0.f : Z * mz_w(M - mn) + zo * mz_w(mo - mn);
          M = mn;
        }
        if (hv && sub == 0) mzw[h] = make_float2(M, 1.0f / Z);
      }
    }
    dma_wait_all();
    __syncthreads();          // mz visible, the scores of chunk 0 (and 1) landed
    {
      const int hr = tid / CT;                       // the element of a p buffer this lane converts: [hr][tid % CT]
      const int hc = h0 + hr < a.H ? h0 + hr : a.H - 1;
      myM = mz[hc].x;
      myZ = mz[hc].y;
    }
    if (a.mz == nullptr && blockIdx.x == 0 && a.n_sink > 0) {
      for (int i = tid; i < a.H * a.n_sink; i += Cfg::NT) {
        const float2 t = mz[i / a.n_sink];
        a.sink_probs[i] = __float2half_rn(prob_fp16(__half2float(a.sink[i]), t.x, t.y));
      }
      if (a.v_sink != nullptr)
        for (int i = tid; i < a.H * kHeadDim; i += Cfg::NT) {
          const float2 t = mz[i / kHeadDim];
          a.sink_out[i] = sink_output(a.sink, a.v_sink, a.n_sink, i / kHeadDim, i % kHeadDim, t.x, t.y);
        }
    }
  }
  // FUSED: raw score -> probability in place, one element per lane; tokens at or past the end of the range get 0
  auto convert_p = [&](int pbuf, int64_t c0) {
    float *pp = reinterpret_cast<float *>(smem + Cfg::p_off(pbuf)) + tid;
    const float x = *pp;
    *pp = (c0 + (tid % CT) < t1) ? prob_of(x, a.inv, myM, myZ) : 0.f;
  };
  if constexpr (FUSED) convert_p(0, t0);   // (visible after the first chunk's barrier)
  DmaFast df = make_dma_fast<BITS>(a, dl);
  // chunks whose DMA needs no clamps (see issue_fast): all that start at or before `fast_end`
  int64_t fast_end = (a.max_len < a.L ? a.max_len : a.L) - CT;
  if (n_units_valid != Cfg::UW) fast_end = -1;

  // ---- sparse residuals ------------------------------------------------------------------------------
  // Workgroup (range, group g) takes the g-th share of the range's TOKENS for ALL channels, so every
  // entry is touched once.  Sums are accumulated with 64-bit FIXED-POINT LDS atomics (2^-32 resolution,
  // exact and order-independent): ds_add_u64 runs at ~0.1 cycle/lane on gfx950, ds_add_f32 at ~2.6
  // (measured, tools/ubench/lds_atomic.hip).  The accumulator sits behind pipeline stage 0, idle until
  // the first chunk iteration; the sums go to this workgroup's own sparse slab, which the reduce kernel
  // adds like any other slab.
#if KVQ_TRACE
  unsigned tr_acc[12] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
  unsigned tr_prev, tr_passes = 0;
  unsigned long long tr_t0, tr_r0;
  {
    unsigned long long tt;
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(tt)::"memory");
    tr_prev = (unsigned)tt;
    tr_t0 = tt;
    asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(tr_r0)::"memory");
  }
#endif
  // Outlier entries.  Workgroup (range, group g) takes ALL tokens of its range and the entries whose channel belongs to
  // ITS units: the sums land in the channel slice its dense partial covers -- no separate outlier slabs (round 2: one
  // 16 KB slab per workgroup, written here and read back by the reduce kernel: 32 MB of the launch's traffic at 128K)
  // and only this group's heads of the probabilities to stage.  The entries are read once per group (from the L2: see
  // the block numbering above).  The phase runs after the dense loop (32.32 fixed-point LDS adds: exact, order
  // independent); nothing of it is live across the loop.
  auto sparse_phase_rows = [&]() {
    // LDS (aliases the pipeline stages): [0, SP_P_B) the probabilities of a block of the range's tokens for the group's
    // heads, [SP_P_B, ...) the fixed-point accumulators of the group's channels.  Staging p (coalesced, independent of the
    // entries) removes the second dependent memory round trip (entry -> head -> p gather): the phase is pure latency.
    float *pl = reinterpret_cast<float *>(smem);
    long long *sacc = reinterpret_cast<long long *>(smem + Cfg::SP_P_B);
    static_assert((Cfg::SMEM_B - Cfg::SP_P_B) / 8 >= Cfg::UW * CH, "accumulators of all the group's channels in one pass");
    const int c_lo = u0 * CH;                                  // the group's channels: [c_lo, c_lo + cn)
    const int cn = n_units_valid * CH;
    const int HWv = (n_units_valid + Cfg::UPH - 1) / Cfg::UPH; // its heads: [h0, h0 + HWv)
    constexpr int RB = 24;          // entries per lane per round, all loads of a round in flight together
    // token blocks: the entries of a block fit one round (24 x 512 / 42 = 292 tokens), its probabilities the stage
    int sb = a.n_out > 0 ? (RB * Cfg::NT) / a.n_out : 0;
    if (sb > 320) sb = 320;
    if (sb > Cfg::SP_P_B / (4 * HWv) - 1) sb = Cfg::SP_P_B / (4 * HWv) - 1;
    if (sb < 1) return;
#if KVQ_TRACE
    auto sstamp = [&](int k) {
      unsigned long long tt;
      asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(tt)::"memory");
      tr_acc[k] += (unsigned)tt - tr_prev;
      tr_prev = (unsigned)tt;
    };
#endif
    for (int64_t b0 = t0; b0 < t1; b0 += sb) {
      const int ns = (t1 - b0 < sb) ? (int)(t1 - b0) : sb;     // tokens of the block
      const unsigned nent = (unsigned)ns * (unsigned)a.n_out;  // <= RB * NT
      const float *ov = compact ? reinterpret_cast<const float *>(a.idx + b0 * a.n_out) : a.outliers + b0 * a.n_out;
      const int32_t *oi = a.idx + b0 * a.n_out;
      const float *p0 = (FUSED ? a.scores : a.p) + (int64_t)h0 * a.L + b0;   // FUSED: raw scores, converted on the way
      int row[RB];
      float val[RB];
      // (an opaque copy of the thread id per block: otherwise hipcc hoists the 24 entry -> token divisions out of the
      //  block loop and spills them)
      unsigned tid_b = tid;
      asm volatile("" : "+v"(tid_b));
#pragma unroll
      for (int j = 0; j < RB; j++) {
        const unsigned e = j * Cfg::NT + tid_b;
        const unsigned ec = e < nent ? e : nent - 1;
        // (branch-free on purpose: a branch in here makes hipcc lose the wave-uniformity of the DMA bases further down.
        //  Compact rows: `ov` aliases the packed array, the second load hits the line the first one fetched.)
        const uint32_t w = (uint32_t)oi[ec];
        const float fv = ov[ec];
        row[j] = compact ? (int)(w & 0xffffu) : (int)w;
        val[j] = compact ? __half2float(__ushort_as_half((unsigned short)(w >> 16))) : fv;
      }
      const int nsp = ns | 1;                     // odd row stride: consecutive heads fall into different LDS banks
      {
        // wave w stages heads w, w+8, ...; lanes along the tokens (coalesced); all of a lane's loads in flight together
        const int wv = tid >> 6, ln = tid & 63;
        for (int hb = 0; hb < HWv; hb += 32) {
          float v[4][5];
#pragma unroll
          for (int k = 0; k < 4; k++)
#pragma unroll
            for (int m = 0; m < 5; m++) {
              const int hh = hb + wv + 8 * k, tl = ln + 64 * m;
              v[k][m] = (hh < HWv && tl < ns) ? p0[(int64_t)hh * a.L + tl] : 0.f;
            }
#pragma unroll
          for (int k = 0; k < 4; k++)
#pragma unroll
            for (int m = 0; m < 5; m++) {
              const int hh = hb + wv + 8 * k, tl = ln + 64 * m;
              if (hh < HWv && tl < ns)
                pl[hh * nsp + tl] = FUSED ? prob_of(v[k][m], a.inv, mz[h0 + hh].x, mz[h0 + hh].y) : v[k][m];
            }
        }
      }
#if KVQ_TRACE
      sstamp(6);
#endif
      __syncthreads();                            // staged probabilities visible
#if KVQ_TRACE
      sstamp(7);
#endif
#pragma unroll
      for (int j = 0; j < RB; j++) {
        const unsigned e = j * Cfg::NT + tid_b;
        const unsigned ec = e < nent ? e : nent - 1;
        const unsigned tl = __umulhi(ec, a.n_out_magic);
        const unsigned rel = (unsigned)(row[j] - c_lo);          // channel within the group
        const bool mine = e < nent && rel < (unsigned)cn;
        const unsigned hh = mine ? (rel >> 7) : 0u;
        const float pt = pl[hh * nsp + tl];
        if (mine) {
          // x -> 32.32 fixed point without 64-bit float math: floor part + exact 32-bit fraction
          const float x = val[j] * pt;
          const float fl = floorf(x);
          const unsigned lo = (unsigned)((x - fl) * 4294967296.0f);
          const int hi = (int)fl;
          const unsigned long long fx = ((unsigned long long)(unsigned)hi << 32) | lo;
          atomicAdd(reinterpret_cast<unsigned long long *>(&sacc[rel]), fx);
        }
      }
#if KVQ_TRACE
      sstamp(8);
#endif
      __syncthreads();                            // the stage may be rewritten / the sums are complete
#if KVQ_TRACE
      sstamp(9);
#endif
    }
  };
  // Outlier entries, straight-line form (round 5): the workgroup takes the tokens [ts0, ts1) of its range, ALL their slots,
  // and adds into the accumulators of the channels [c_lo, c_lo + cn) (heads [hf0, hf0 + HWv)) -- every entry it reads is
  // its own.  All of a block's entries and the raw scores of its tokens are requested together through buffer
  // descriptors (one 32-bit lane offset per load, no address pairs), every entry is evaluated without control flow (a
  // padding entry adds 0 into a per-lane dummy accumulator) and nothing of the entry -> (token, head) arithmetic is kept
  // across the loads.  Two uses:
  //  * ONE unit group (2 / 3 bit at H = 32): the whole range, the group's channels.  Same box, 128K nuq3 + 5 sinks:
  //    p.V + reduce 93.6 -> 89.3 us (profiles/r05_pv_outlier_phase.txt).
  //  * several unit groups, SPLIT BY TOKENS (a.split; 4 bit at H = 32, long caches): group g takes the g-th share of the
  //    range's tokens for ALL channels, so that every entry is read once, by one workgroup, in one round of 21 loads per
  //    lane (sparse_phase_rows: every group reads every entry, two rounds of 24, and skips the other groups' with a branch
  //    per entry).  The sums for its own channels join its dense partial as before; those for the other groups' channels
  //    go to EXTRA slabs that the reduce kernel adds like any other (slab n_ranges + range * (G - 1) + j holds, for every
  //    group x, what group (x - 1 - j) mod G found for x's channels: each extra slab is complete).
  //    Windows of the channel-sorted rows instead (32 of 42 slots from the group's end, a second pass when some token
  //    needs it) were measured slower: one workgroup in four takes the second pass and the launch waits for it (112 us).
  auto sparse_phase_all = [&](int64_t ts0, int64_t ts1, int c_lo, int cn, int hf0, int HWv) {
    float *pl = reinterpret_cast<float *>(smem);
    long long *sacc = reinterpret_cast<long long *>(smem + Cfg::SP_P_B);
    static_assert((Cfg::SMEM_B - Cfg::SP_P_B) / 8 >= Cfg::UW * CH + 64, "accumulators of all the group's channels + dummies");
    constexpr int NACC = (Cfg::SMEM_B - Cfg::SP_P_B) / 8 - 64;      // accumulators in front of the 64 dummies
    constexpr int RB = KVQ_V_RB;    // entries per lane and block, all loads in flight together
    constexpr int PE = 16;          // staged probabilities per lane and batch
#if KVQ_TRACE
    auto sstamp = [&](int k) {
      unsigned long long tt;
      asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(tt)::"memory");
      tr_acc[k] += (unsigned)tt - tr_prev;
      tr_prev = (unsigned)tt;
    };
#endif
    const int w_n = a.n_out;
    const uint32_t w_magic = a.n_out_magic;
    // token blocks: the entries of a block fit one round, its probabilities the stage
    int sb = w_n > 0 ? (RB * Cfg::NT) / w_n : 0;
    if (sb > Cfg::NT) sb = Cfg::NT;
    if (sb > Cfg::SP_P_B / (4 * HWv) - 1) sb = Cfg::SP_P_B / (4 * HWv) - 1;
    if (sb < 1) return;
    for (int64_t b0 = ts0; b0 < ts1; b0 += sb) {
      const int ns = (ts1 - b0 < sb) ? (int)(ts1 - b0) : sb;     // tokens of the block
      const unsigned nent = (unsigned)ns * (unsigned)w_n;        // <= RB * NT
      // buffer descriptors over the block's entries and the score rows: every load below is a wave-uniform descriptor +
      // ONE 32-bit lane offset (no 64-bit address pairs: 40 loads in flight per lane)
      const __amdgpu_buffer_rsrc_t r_idx = uniform_rsrc(a.idx + b0 * a.n_out, nent_row(ns, a.n_out) * 4u);
      const __amdgpu_buffer_rsrc_t r_val = uniform_rsrc(
          compact ? reinterpret_cast<const float *>(a.idx + b0 * a.n_out) : a.outliers + b0 * a.n_out, nent_row(ns, a.n_out) * 4u);
      const __amdgpu_buffer_rsrc_t r_p = uniform_rsrc((FUSED ? a.scores : a.p) + (int64_t)hf0 * a.L + b0, 0x7ffffffcu);   // FUSED: raw scores, converted on the way
      int row[RB];
      float val[RB];
      // (an opaque copy of the thread id per block: otherwise hipcc hoists the entry -> token divisions out of the
      //  block loop and spills them)
      unsigned tid_b = tid;
      asm volatile("" : "+v"(tid_b));
#pragma unroll
      for (int j = 0; j < RB; j++) {
        const unsigned e = j * Cfg::NT + tid_b;
        const unsigned ec = e < nent ? e : nent - 1;
        // (Compact rows: `r_val` aliases the packed array, the second load hits the line the first one fetched.)
        const uint32_t w = (uint32_t)__builtin_amdgcn_raw_buffer_load_b32(r_idx, ec * 4u, 0, 0);
        const float fv = __uint_as_float((uint32_t)__builtin_amdgcn_raw_buffer_load_b32(r_val, ec * 4u, 0, 0));
        row[j] = compact ? (int)(w & 0xffffu) : (int)w;
        val[j] = compact ? __half2float(__ushort_as_half((unsigned short)(w >> 16))) : fv;
      }
      const int nsp = ns | 1;                     // odd row stride: consecutive heads fall into different LDS banks
      {
        // the block's probabilities of the heads: lanes along the tokens (coalesced), PE loads per lane in flight
        const float rns = 1.0f / (float)ns;
        const unsigned nel = (unsigned)HWv * (unsigned)ns;
        for (unsigned k0 = 0; k0 < nel; k0 += PE * Cfg::NT) {
          float v[PE];
#pragma unroll
          for (int k = 0; k < PE; k++) {
            const unsigned e = k0 + k * Cfg::NT + tid_b;
            const unsigned ec = e < nel ? e : nel - 1;
            const unsigned hh = (unsigned)(((float)ec + 0.5f) * rns);      // ec / ns (exact: the quotient is never within 0.5 / ns of an integer)
            const unsigned tl = ec - hh * (unsigned)ns;
            v[k] = __uint_as_float((uint32_t)__builtin_amdgcn_raw_buffer_load_b32(r_p, (hh * (unsigned)a.L + tl) * 4u, 0, 0));
          }
#pragma unroll
          for (int k = 0; k < PE; k++) {
            const unsigned e = k0 + k * Cfg::NT + tid_b;
            const unsigned ec = e < nel ? e : nel - 1;
            const unsigned hh = (unsigned)(((float)ec + 0.5f) * rns);
            const unsigned tl = ec - hh * (unsigned)ns;
            if (e < nel)
              pl[hh * nsp + tl] = FUSED ? prob_of(v[k], a.inv, mz[hf0 + hh].x, mz[hf0 + hh].y) : v[k];
          }
        }
      }
#if KVQ_TRACE
      sstamp(6);
#endif
      __syncthreads();                            // staged probabilities visible
#if KVQ_TRACE
      sstamp(7);
#endif
      // (a second opaque copy: the entry -> (token, slot) arithmetic is done again instead of being kept -- spilled --
      //  across the loads)
      unsigned tid_c = tid;
      asm volatile("" : "+v"(tid_c));
#pragma unroll
      for (int j = 0; j < RB; j++) {
        const unsigned e = j * Cfg::NT + tid_c;
        const unsigned ec = e < nent ? e : nent - 1;
        const unsigned tl = __umulhi(ec, w_magic);               // token within the block
        const unsigned rel = (unsigned)(row[j] - c_lo);          // channel within the span
        const bool mine = e < nent && rel < (unsigned)cn;
        const unsigned hh = mine ? (rel >> 7) : 0u;
        const float pt = pl[hh * nsp + tl];
        // x -> 32.32 fixed point without 64-bit float math: floor part + exact 32-bit fraction
        const float x = mine ? val[j] * pt : 0.f;
        const float fl = floorf(x);
        const unsigned lo = (unsigned)((x - fl) * 4294967296.0f);
        const int hi = (int)fl;
        const unsigned long long fx = ((unsigned long long)(unsigned)hi << 32) | lo;
        // (a padding entry: + 0 into this lane's dummy accumulator -- no branch)
        const unsigned slot = mine ? rel : (unsigned)NACC + (tid_c & 63u);
        atomicAdd(reinterpret_cast<unsigned long long *>(&sacc[slot]), fx);
      }
#if KVQ_TRACE
      sstamp(8);
#endif
      __syncthreads();                            // the stage may be rewritten / the sums are complete
#if KVQ_TRACE
      sstamp(9);
#endif
    }
  };
  // this workgroup's share of the range's tokens when the outlier entries are split by tokens (a.split)
  const int64_t split_share = ((t1 - t0) + a.groups - 1) / a.groups;
  const int64_t split_t0 = (t0 + g * split_share < t1) ? t0 + g * split_share : t1;
  const int64_t split_t1 = (split_t0 + split_share < t1) ? split_t0 + split_share : t1;
  auto sparse_phase = [&]() {
    if (a.groups == 1 || a.split) {
      // (one call site: the arguments are wave-uniform selects)
      const bool sp = a.split != 0;
      sparse_phase_all(sp ? split_t0 : t0, sp ? split_t1 : t1, sp ? 0 : u0 * CH, sp ? C : n_units_valid * CH, sp ? 0 : h0,
                       sp ? a.H : (n_units_valid + Cfg::UPH - 1) / Cfg::UPH);
    } else {
      sparse_phase_rows();
    }
  };
  // the lane that stores a unit half's channels at the end (slot 0), its slice of the partial slab and of the accumulators
  // (computed from an opaque copy of the thread id wherever it is needed, so that none of it stays live across the dense
  //  loop, which runs at the 128-VGPR limit: a spill reload inside the loop waits for vmcnt(0) and drains the DMAs)
  struct Slice { bool writer; float *dst; long long *sacc; };
  auto slice_of = [&]() {
    int t = threadIdx.x;
    asm volatile("" : "+v"(t));
    const int ul_ = t % Cfg::UW, hf_ = (t / Cfg::UW) % Cfg::HALVES, sl_ = t / (Cfg::UW * Cfg::HALVES);
    Slice r;
    r.writer = sl_ == 0 && ul_ < n_units_valid;
    r.dst = a.partial + ((int64_t)range * a.q_len + b) * C + (int64_t)(u0 + ul_) * CH + hf_ * CHL;
    r.sacc = reinterpret_cast<long long *>(smem + Cfg::SP_P_B) + (a.split ? u0 * CH : 0) + ul_ * CH + hf_ * CHL;   // (split: indexed by the layer's channel)
    return r;
  };
  // split by tokens: every accumulator of the layer is this workgroup's; after the phase the sums of the OTHER groups'
  // channels go to the extra slabs (see sparse_phase_all)
  auto zero_all_acc = [&]() {
    long long *sa = reinterpret_cast<long long *>(smem + Cfg::SP_P_B);
    for (int i = tid; i < C; i += Cfg::NT) sa[i] = 0;
  };
  auto write_foreign = [&]() {
    const long long *sa = reinterpret_cast<const long long *>(smem + Cfg::SP_P_B);
    const int G = a.groups, n_ranges = (int)gridDim.x / G;
    constexpr int GC = Cfg::UW * CH;                                   // channels of a full unit group
    for (int j = 0; j + 1 < G; j++) {
      const int x = (g + 1 + j) % G;
      const int gcx = (C - x * GC < GC) ? C - x * GC : GC;
      float *slab = a.partial + (int64_t)(n_ranges + range * (G - 1) + j) * C + x * GC;
      for (int c = tid * 4; c < gcx; c += Cfg::NT * 4) {
        const long long *q4 = sa + x * GC + c;
        *reinterpret_cast<float4 *>(slab + c) = make_float4((float)((double)q4[0] * (1.0 / 4294967296.0)), (float)((double)q4[1] * (1.0 / 4294967296.0)),
                                                             (float)((double)q4[2] * (1.0 / 4294967296.0)), (float)((double)q4[3] * (1.0 / 4294967296.0)));
      }
    }
  };
  if (sparse_first) {
    const Slice sc = slice_of();
    const bool writer = sc.writer;
    float *dst = sc.dst;
    long long *sacc_w = sc.sacc;
    // phase first: its sums wait in this workgroup's slab (global memory; the registers are needed by the loop) and are
    // picked up again by the same lanes at the end
    if (a.split) {
      zero_all_acc();
    } else if (writer) {
#pragma unroll
      for (int i = 0; i < CHL; i++) sacc_w[i] = 0;
    }
    __syncthreads();
    sparse_phase();
    if (a.split) write_foreign();
    if (writer) {
#pragma unroll
      for (int i = 0; i < CHL; i += 4) {
        float4 v;
        v.x = (float)((double)sacc_w[i] * (1.0 / 4294967296.0));
        v.y = (float)((double)sacc_w[i + 1] * (1.0 / 4294967296.0));
        v.z = (float)((double)sacc_w[i + 2] * (1.0 / 4294967296.0));
        v.w = (float)((double)sacc_w[i + 3] * (1.0 / 4294967296.0));
        *reinterpret_cast<float4 *>(dst + i) = v;
      }
    }
    __syncthreads();                      // the phase is done with the LDS: the first chunk may land
    // (the per-lane DMA constants are computed again here: their first definitions are then dead on this path and do
    //  not occupy registers during the phase)
    dl = make_dma_lane<BITS>();
    asm volatile("" : "+v"(dl.tile_row), "+v"(dl.tile_q4), "+v"(dl.lut_tok), "+v"(dl.lut_sub), "+v"(dl.p_head), "+v"(dl.p_tok));
    df = make_dma_fast<BITS>(a, dl);
    issue_chunk<BITS>(a, dl, lds0, 0, t0, row_base, n_rows_valid, h0, b, !FUSED);
  }
  float acc[CHL];
#pragma unroll
  for (int i = 0; i < CHL; i++) acc[i] = 0.f;


  // slot pattern OR-ed into the pre-masked nibble bytes (4-bit fast path): byte = slot*64 + code*4
  const uint32_t slotpat = (uint32_t)sl * 0x40404040u;

  // hand-scheduled loops (3 / 4 bit): the lane's LDS addresses inside a stage do not change with the chunk
  uint32_t taddr[Cfg::QPL][WORDS];   // its 16-byte quads of the tile
  uint32_t paddr = 0;                // the probabilities of its head for its slot's tokens
  if (lds0 != 0) __builtin_trap();   // (the instruction immediates below assume the one static LDS array at 0)
#pragma unroll
  for (int qq = 0; qq < Cfg::QPL; qq++)
#pragma unroll
    for (int wi = 0; wi < WORDS; wi++) {
      const int r = ul * WORDS + wi, rot = (r >> Cfg::SH) & (Cfg::QR - 1);     // (tile row, its quad rotation)
      taddr[qq][wi] = (uint32_t)(r * Cfg::ROWB + (((sl * Cfg::QPL + qq + rot) & (Cfg::QR - 1)) << 4));
    }
  paddr = (uint32_t)((hl * CT + sl * Cfg::QPL * 4) * 4);

  auto chunk = [&](auto STAGE, int ci) {
    constexpr int stage = decltype(STAGE)::value;
    const int64_t c0 = t0 + (int64_t)ci * CT;
#if KVQ_TRACE
    auto stamp = [&](int k) {
      unsigned long long tt;
      asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(tt)::"memory");
      tr_acc[k] += (unsigned)tt - tr_prev;
      tr_prev = (unsigned)tt;
    };
    stamp(0);
#endif
    dma_wait_all();                       // this wave's DMA pieces of chunk ci have landed
#if KVQ_TRACE
    stamp(1);
#endif
    __syncthreads();   // ... and everybody else's; the other stage is free again
#if KVQ_TRACE
    stamp(2);
#endif
    const int64_t cn0 = c0 + CT;                   // start of the next chunk
    const bool more = ci + 1 < n_chunks;
    // FUSED: the scores travel one chunk further ahead than the rows (chunk ci+2 -> p buffer (ci+2) % 3)
    const int pcur = FUSED ? ci % 3 : stage;
    const int pnext = FUSED ? (ci + 2) % 3 : 1 - stage;
    const int64_t pc0 = FUSED ? cn0 + CT : cn0;
    const bool spread = pc0 <= fast_end;   // (wave-uniform)
    if (more && !spread) {
      issue_chunk<BITS>(a, dl, lds0, 1 - stage, cn0, row_base, n_rows_valid, h0, b, !FUSED);
      if (FUSED && ci + 2 < n_chunks) issue_p<BITS>(a.scores, a, dl, lds0 + Cfg::p_off(pnext), pc0, h0, b);
    }
    // the next chunk's scores landed with this chunk's rows (issued one chunk earlier): convert them now -- in the
    // hand-scheduled loop the LDS read is issued here and the value is used two quads later, behind the look-ups
    constexpr bool CONV_IN_LOOP = FUSED && BITS == 4 && Cfg::QPL >= 3;
    float raw_next = 0.f;
    const uint32_t conv_addr = (uint32_t)(tid * 4 + ((ci + 1) % 3) * Cfg::P_B);
    if constexpr (CONV_IN_LOOP) {
      if (more) asm volatile("ds_read_b32 %0, %1 offset:%2" : "=v"(raw_next) : "v"(conv_addr), "n"(Cfg::p_off(0)) : "memory");
    } else if (FUSED && more) {
      convert_p((ci + 1) % 3, cn0);
    }
#if KVQ_TRACE
    stamp(3);
#endif
    float *pb = reinterpret_cast<float *>(smem + Cfg::p_off(pcur));
    const int rem = (int)(t1 - c0);       // tokens of this range left in the chunk
    if (!FUSED && rem < CT) {             // ragged last chunk: zero the probabilities past the end once
      for (int i = tid; i < Cfg::HW * CT; i += Cfg::NT)
        if (i % CT >= rem) pb[i] = 0.f;
      __syncthreads();
    }
    if constexpr (BITS == 4) {
      constexpr int S0 = Cfg::tile_off(stage);                  // tile
      constexpr int L0 = Cfg::lut_off(stage);                   // codebook rows (row (qq*4+e)*SLOTS + slot)
      constexpr int P0 = Cfg::p_off(0);                         // probabilities (buffer offset in the address register)
      const uint32_t paddr_c = paddr + (uint32_t)(pcur * Cfg::P_B);
      constexpr int TS = Cfg::SLOTS * N * 4;                    // bytes between the rows of consecutive tokens of a slot
      uint4 wq[2];
      float4 pq[2];
      lds_read16<S0>(wq[0], taddr[0][0]);
      lds_read16<P0>(pq[0], paddr_c);
      static_for<0, Cfg::QPL>([&](auto QQ) {
        constexpr int qq = decltype(QQ)::value;
        constexpr int cur = qq & 1;
        uint32_t we, wo, ua[8], ub[8];
        float va[8], vb[8];
        lds_wait<0>();                                          // this quad's words and probabilities
        if constexpr (CONV_IN_LOOP && qq == 1) asm volatile("" : "+v"(raw_next));   // (landed: everything older than this wait has)
        if constexpr (CONV_IN_LOOP && qq == 2) {
          if (more) {
            float pr = (cn0 + (tid % CT) < t1) ? prob_of(raw_next, a.inv, myM, myZ) : 0.f;
            asm volatile("ds_write_b32 %0, %1 offset:%2" ::"v"(conv_addr), "v"(pr), "n"(Cfg::p_off(0)) : "memory");
          }
        }
        nib_prep(we, wo, wq[cur].x, slotpat); nib_extract(ua, we, wo); lut_read8<L0 + (qq * 4 + 0) * TS>(va, ua);
        nib_prep(we, wo, wq[cur].y, slotpat); nib_extract(ub, we, wo); lut_read8<L0 + (qq * 4 + 1) * TS>(vb, ub);
        // a quad's share of the next chunk's DMA, behind 16 look-ups in flight (the other stage is free since the barrier)
        if (more && spread) issue_fast<BITS, qq>(a, df, lds0, 1 - stage, pnext, FUSED ? a.scores : a.p, cn0, pc0, row_base, h0, b);
        lds_wait<8>(); fmac8(acc, va, pq[cur].x);
        nib_prep(we, wo, wq[cur].z, slotpat); nib_extract(ua, we, wo); lut_read8<L0 + (qq * 4 + 2) * TS>(va, ua);
        lds_wait<8>(); fmac8(acc, vb, pq[cur].y);
        nib_prep(we, wo, wq[cur].w, slotpat); nib_extract(ub, we, wo); lut_read8<L0 + (qq * 4 + 3) * TS>(vb, ub);
        if constexpr (qq + 1 < Cfg::QPL) {                      // the next quad, behind the look-ups in flight
          lds_read16<S0>(wq[1 - cur], taddr[qq + 1][0]);
          lds_read16<P0 + (qq + 1) * 16>(pq[1 - cur], paddr_c);
          lds_wait<10>(); fmac8(acc, va, pq[cur].z);
          lds_wait<2>(); fmac8(acc, vb, pq[cur].w);
        } else {
          lds_wait<8>(); fmac8(acc, va, pq[cur].z);
          lds_wait<0>(); fmac8(acc, vb, pq[cur].w);
        }
      });
#if KVQ_TRACE
      stamp(4);
#endif
      return;
    }
    if constexpr (BITS == 3) {
      // per token: two groups of 8 look-ups (codes 0..7 and 8..15 of the lane's half); the groups of token t+1 are in
      // flight while token t is accumulated, as in the 4-bit loop
      constexpr int S0 = Cfg::tile_off(stage);
      constexpr int L0 = Cfg::lut_off(stage);
      constexpr int P0 = Cfg::p_off(0);
      constexpr int TS = Cfg::SLOTS * N * 4;                    // bytes between the rows of consecutive tokens of a slot
      const uint32_t paddr_c = paddr + (uint32_t)(pcur * Cfg::P_B);
      const uint32_t slot3 = (uint32_t)sl * 0x20820820u;        // the slot's row offset (N*4 = 32 bytes) in every 6-bit field
      uint4 wq[2][3];
      float4 pq[2];
#pragma unroll
      for (int wi = 0; wi < 3; wi++) lds_read16<S0>(wq[0][wi], taddr[0][wi]);
      lds_read16<P0>(pq[0], paddr_c);
      static_for<0, Cfg::QPL>([&](auto QQ) {
        constexpr int qq = decltype(QQ)::value;
        constexpr int cur = qq & 1;
        uint32_t s1, s2, e1, o1, e2, o2, ua[8];
        float va[8], vb[8];
        lds_wait<0>();                                          // this quad's words and probabilities
        // token 0
        tri_streams(s1, s2, wq[cur][0].x, wq[cur][1].x, wq[cur][2].x, hf);
        tri_prep(e1, o1, s1, slot3); tri_prep(e2, o2, s2, slot3);
        tri_extract_a(ua, e1, o1); lut_read8<L0 + (qq * 4 + 0) * TS>(va, ua);
        tri_extract_b(ua, e1, o1, e2, o2); lut_read8<L0 + (qq * 4 + 0) * TS>(vb, ua);
        if (more && spread) issue_fast<BITS, qq>(a, df, lds0, 1 - stage, pnext, FUSED ? a.scores : a.p, cn0, pc0, row_base, h0, b);
        // token 1
        tri_streams(s1, s2, wq[cur][0].y, wq[cur][1].y, wq[cur][2].y, hf);
        tri_prep(e1, o1, s1, slot3); tri_prep(e2, o2, s2, slot3);
        lds_wait<8>(); fmac8_at<0>(acc, va, pq[cur].x);
        tri_extract_a(ua, e1, o1); lut_read8<L0 + (qq * 4 + 1) * TS>(va, ua);
        lds_wait<8>(); fmac8_at<8>(acc, vb, pq[cur].x);
        tri_extract_b(ua, e1, o1, e2, o2); lut_read8<L0 + (qq * 4 + 1) * TS>(vb, ua);
        // token 2
        tri_streams(s1, s2, wq[cur][0].z, wq[cur][1].z, wq[cur][2].z, hf);
        tri_prep(e1, o1, s1, slot3); tri_prep(e2, o2, s2, slot3);
        lds_wait<8>(); fmac8_at<0>(acc, va, pq[cur].y);
        tri_extract_a(ua, e1, o1); lut_read8<L0 + (qq * 4 + 2) * TS>(va, ua);
        lds_wait<8>(); fmac8_at<8>(acc, vb, pq[cur].y);
        tri_extract_b(ua, e1, o1, e2, o2); lut_read8<L0 + (qq * 4 + 2) * TS>(vb, ua);
        // token 3
        tri_streams(s1, s2, wq[cur][0].w, wq[cur][1].w, wq[cur][2].w, hf);
        tri_prep(e1, o1, s1, slot3); tri_prep(e2, o2, s2, slot3);
        lds_wait<8>(); fmac8_at<0>(acc, va, pq[cur].z);
        tri_extract_a(ua, e1, o1); lut_read8<L0 + (qq * 4 + 3) * TS>(va, ua);
        lds_wait<8>(); fmac8_at<8>(acc, vb, pq[cur].z);
        tri_extract_b(ua, e1, o1, e2, o2); lut_read8<L0 + (qq * 4 + 3) * TS>(vb, ua);
        if constexpr (qq + 1 < Cfg::QPL) {                      // the next quad, behind the look-ups in flight
#pragma unroll
          for (int wi = 0; wi < 3; wi++) lds_read16<S0>(wq[1 - cur][wi], taddr[qq + 1][wi]);
          lds_read16<P0 + (qq + 1) * 16>(pq[1 - cur], paddr_c);
          lds_wait<12>(); fmac8_at<0>(acc, va, pq[cur].w);
          lds_wait<4>(); fmac8_at<8>(acc, vb, pq[cur].w);
        } else {
          lds_wait<8>(); fmac8_at<0>(acc, va, pq[cur].w);
          lds_wait<0>(); fmac8_at<8>(acc, vb, pq[cur].w);
        }
      });
#if KVQ_TRACE
      stamp(4);
#endif
      return;
    }
    if constexpr (BITS == 2) {
      // as the 3-bit loop: two groups of 8 look-ups per token (channels 0..7 and 8..15 of the word)
      constexpr int S0 = Cfg::tile_off(stage);
      constexpr int L0 = Cfg::lut_off(stage);
      constexpr int P0 = Cfg::p_off(0);
      constexpr int TS = Cfg::SLOTS * N * 4;
      const uint32_t paddr_c = paddr + (uint32_t)(pcur * Cfg::P_B);
      const uint32_t slot2 = (uint32_t)sl * 0x10101010u;        // the slot's row offset (N*4 = 16 bytes) in every byte
      uint4 wq[2];
      float4 pq[2];
      lds_read16<S0>(wq[0], taddr[0][0]);
      lds_read16<P0>(pq[0], paddr_c);
      static_for<0, Cfg::QPL>([&](auto QQ) {
        constexpr int qq = decltype(QQ)::value;
        constexpr int cur = qq & 1;
        uint32_t pk[4], ua[8];
        float va[8], vb[8];
        lds_wait<0>();
        duo_prep(pk, wq[cur].x, slot2);
        duo_extract_a(ua, pk); lut_read8<L0 + (qq * 4 + 0) * TS>(va, ua);
        duo_extract_b(ua, pk); lut_read8<L0 + (qq * 4 + 0) * TS>(vb, ua);
        if (more && spread) issue_fast<BITS, qq>(a, df, lds0, 1 - stage, pnext, FUSED ? a.scores : a.p, cn0, pc0, row_base, h0, b);
        duo_prep(pk, wq[cur].y, slot2);
        lds_wait<8>(); fmac8_at<0>(acc, va, pq[cur].x);
        duo_extract_a(ua, pk); lut_read8<L0 + (qq * 4 + 1) * TS>(va, ua);
        lds_wait<8>(); fmac8_at<8>(acc, vb, pq[cur].x);
        duo_extract_b(ua, pk); lut_read8<L0 + (qq * 4 + 1) * TS>(vb, ua);
        duo_prep(pk, wq[cur].z, slot2);
        lds_wait<8>(); fmac8_at<0>(acc, va, pq[cur].y);
        duo_extract_a(ua, pk); lut_read8<L0 + (qq * 4 + 2) * TS>(va, ua);
        lds_wait<8>(); fmac8_at<8>(acc, vb, pq[cur].y);
        duo_extract_b(ua, pk); lut_read8<L0 + (qq * 4 + 2) * TS>(vb, ua);
        duo_prep(pk, wq[cur].w, slot2);
        lds_wait<8>(); fmac8_at<0>(acc, va, pq[cur].z);
        duo_extract_a(ua, pk); lut_read8<L0 + (qq * 4 + 3) * TS>(va, ua);
        lds_wait<8>(); fmac8_at<8>(acc, vb, pq[cur].z);
        duo_extract_b(ua, pk); lut_read8<L0 + (qq * 4 + 3) * TS>(vb, ua);
        if constexpr (qq + 1 < Cfg::QPL) {
          lds_read16<S0>(wq[1 - cur], taddr[qq + 1][0]);
          lds_read16<P0 + (qq + 1) * 16>(pq[1 - cur], paddr_c);
          lds_wait<10>(); fmac8_at<0>(acc, va, pq[cur].w);
          lds_wait<2>(); fmac8_at<8>(acc, vb, pq[cur].w);
        } else {
          lds_wait<8>(); fmac8_at<0>(acc, va, pq[cur].w);
          lds_wait<0>(); fmac8_at<8>(acc, vb, pq[cur].w);
        }
      });
#if KVQ_TRACE
      stamp(4);
#endif
      return;
    }
  };
  for (int ci = 0; ci < n_chunks; ci += 2) {
    chunk(std::integral_constant<int, 0>{}, ci);
    if (ci + 1 < n_chunks) chunk(std::integral_constant<int, 1>{}, ci + 1);
  }

  // ---- sum the token slots through LDS (aliases the pipeline stages)
  __syncthreads();
  float *red = reinterpret_cast<float *>(smem);
#pragma unroll
  for (int i = 0; i < CHL; i++) red[(i * Cfg::SLOTS + sl) * (Cfg::UW * Cfg::HALVES) + lu] = acc[i];
  __syncthreads();
  static_assert(Cfg::RED_B <= Cfg::SP_P_B, "the slot sums are read while the outlier accumulators are zeroed");
  const Slice sc_end = slice_of();
  const bool writer = sc_end.writer;
  float *dst = sc_end.dst;
  long long *sacc_w = sc_end.sacc;
  float o[CHL];
#pragma unroll
  for (int i = 0; i < CHL; i++) o[i] = 0.f;
  if (writer) {
#pragma unroll
    for (int i = 0; i < CHL; i++) {
      float s = red[(i * Cfg::SLOTS) * (Cfg::UW * Cfg::HALVES) + lu];
#pragma unroll
      for (int k = 1; k < Cfg::SLOTS; k++) s += red[(i * Cfg::SLOTS + k) * (Cfg::UW * Cfg::HALVES) + lu];
      o[i] = s;
    }
    if (sparse && !sparse_first && !a.split) {
#pragma unroll
      for (int i = 0; i < CHL; i++) sacc_w[i] = 0;     // (the writers cover exactly the group's channels)
    }
  }
  if (sparse && !sparse_first && a.split) zero_all_acc();
#if KVQ_TRACE
  {
    unsigned long long tt;
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(tt)::"memory");
    tr_acc[5] += (unsigned)tt - tr_prev;   // slot reduce
    tr_prev = (unsigned)tt;
  }
#endif
  // dense sums (fp32, exact for a channel without entries) + outlier sums (exact in fixed point, rounded once)
  if (sparse_first) {
    if (writer) {
      // what this lane stored before the loop: asm loads, so that hipcc cannot forward the stored values through
      // registers (which would keep them live across the loop)
      typedef float f32x4_t __attribute__((ext_vector_type(4)));
      f32x4_t q[CHL / 4];
#pragma unroll
      for (int i = 0; i < CHL; i += 4) asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(q[i / 4]) : "v"(dst + i) : "memory");
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
      for (int i = 0; i < CHL; i += 4) {
        asm volatile("" : "+v"(q[i / 4]));
        o[i] += q[i / 4].x; o[i + 1] += q[i / 4].y; o[i + 2] += q[i / 4].z; o[i + 3] += q[i / 4].w;
      }
#pragma unroll
      for (int i = 0; i < CHL; i += 4) *reinterpret_cast<float4 *>(dst + i) = make_float4(o[i], o[i + 1], o[i + 2], o[i + 3]);
    }
  } else if (sparse) {
    // (the dense sums wait in the slab during the phase, like the outlier sums of the other order during the loop)
    if (writer) {
#pragma unroll
      for (int i = 0; i < CHL; i += 4) *reinterpret_cast<float4 *>(dst + i) = make_float4(o[i], o[i + 1], o[i + 2], o[i + 3]);
    }
    __syncthreads();   // the slot sums are read, the accumulators zeroed: the stage may be overwritten
    sparse_phase();    // (ends with a barrier: the sums are complete)
    if (a.split) write_foreign();
    const Slice sc2 = slice_of();
    if (sc2.writer) {
      typedef float f32x4_t __attribute__((ext_vector_type(4)));
      f32x4_t q[CHL / 4];
#pragma unroll
      for (int i = 0; i < CHL; i += 4) asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(q[i / 4]) : "v"(sc2.dst + i) : "memory");
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
      for (int i = 0; i < CHL; i += 4) {
        asm volatile("" : "+v"(q[i / 4]));
        const float4 v = make_float4(q[i / 4].x + (float)((double)sc2.sacc[i] * (1.0 / 4294967296.0)),
                                     q[i / 4].y + (float)((double)sc2.sacc[i + 1] * (1.0 / 4294967296.0)),
                                     q[i / 4].z + (float)((double)sc2.sacc[i + 2] * (1.0 / 4294967296.0)),
                                     q[i / 4].w + (float)((double)sc2.sacc[i + 3] * (1.0 / 4294967296.0)));
        *reinterpret_cast<float4 *>(sc2.dst + i) = v;
      }
    }
  }
  else if (writer) {      // (every path stores its own sums: nothing of `o` is live across the outlier phase)
#pragma unroll
    for (int i = 0; i < CHL; i += 4) *reinterpret_cast<float4 *>(dst + i) = make_float4(o[i], o[i + 1], o[i + 2], o[i + 3]);
  }
#if KVQ_TRACE
  {
    unsigned long long tt;
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(tt)::"memory");
    if ((tid & 63) == 0 && blockIdx.x < 512) {
      unsigned long long *tr = a.trace + ((int64_t)blockIdx.x * 8 + (tid >> 6)) * 16;
      for (int k = 0; k < 10; k++) tr[k] = tr_acc[k];
      tr[10] = (unsigned)tt - tr_prev;   // rest of the sparse phase (slab write)
      tr[11] = n_chunks;
      unsigned long long rr;
      asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(rr)::"memory");
      tr[12] = tt - tr_t0;               // shader clocks of the wave
      tr[13] = rr - tr_r0;               // 100 MHz ticks of the wave
      tr[14] = tr_passes;                // passes of the outlier phase
    }
  }
#endif
}

// mul[b][c] (+)= sum_r partial[r][b][c], fixed order.  16 channels x 64 slab lanes per block: 256 blocks at C = 4096
// (one per CU), 64-byte row segments, all of a lane's loads in flight (the pass is bound by memory latency:
// 456 slabs at 128K = 8 per lane; the outlier sums are part of the slabs).
__global__ __launch_bounds__(1024) void mix_v_reduce_kernel(const float *__restrict__ partial,
                                                            float *__restrict__ mul, int n_ranges,
                                                            int q_len, int C, int accumulate) {
  __shared__ float red[64][17];
  const int cl = threadIdx.x & 15, rg = threadIdx.x >> 4;
  const int c = blockIdx.x * 16 + cl;
  const int b = blockIdx.y;
  float s = 0.f;
  if (c < C) {
    // slabs [r][b][c], 8 loads in flight per lane
    auto run = [&](const float *src, int64_t stride, int n) {
      int r = rg;
      for (; r + 7 * 64 < n; r += 8 * 64) {
        float v[8];
#pragma unroll
        for (int k = 0; k < 8; k++) v[k] = src[(int64_t)(r + 64 * k) * stride];
#pragma unroll
        for (int k = 0; k < 8; k++) s += v[k];
      }
      float v[8];
#pragma unroll
      for (int k = 0; k < 8; k++) v[k] = (r + 64 * k < n) ? src[(int64_t)(r + 64 * k) * stride] : 0.f;
#pragma unroll
      for (int k = 0; k < 8; k++) s += v[k];
    };
    run(partial + (int64_t)b * C + c, (int64_t)q_len * C, n_ranges);
  }
  red[rg][cl] = s;
  __syncthreads();
  if (rg == 0 && c < C) {
    float t = 0.f;
#pragma unroll
    for (int k = 0; k < 64; k++) t += red[k][cl];
    if (accumulate) t += mul[(int64_t)b * C + c];
    mul[(int64_t)b * C + c] = t;
  }
}

// (max, normaliser) of every softmax row from the score kernel's per-(head, tile) partials and the fp16 sink scores,
// and the sink probabilities: the part of kvq_softmax_finish that is not per token.  One workgroup per head.
__global__ __launch_bounds__(256) void softmax_merge_kernel(const float *__restrict__ parts, int n_parts,
                                                            const __half *__restrict__ sink, __half *__restrict__ sink_probs,
                                                            int n_sink, float *__restrict__ mz,
                                                            const __half *__restrict__ v_sink, float *__restrict__ sink_out) {
  __shared__ float red[8];
  const int h = blockIdx.x, tid = threadIdx.x;
  const float2 *pr = reinterpret_cast<const float2 *>(parts) + (int64_t)h * n_parts;
  float M = -INFINITY, Z = 0.f;
  for (int i0 = 0; i0 < n_parts; i0 += 4 * 256) {
    float2 ms[4];
#pragma unroll
    for (int k = 0; k < 4; k++) {
      const int i = i0 + tid + 256 * k;
      ms[k] = i < n_parts ? pr[i] : make_float2(-INFINITY, 0.f);
    }
#pragma unroll
    for (int k = 0; k < 4; k++)
      if (ms[k].x > -INFINITY) {
        const float mn = fmaxf(M, ms[k].x);
        Z = Z * mz_w(M - mn) + ms[k].y * mz_w(ms[k].x - mn);
        M = mn;
      }
  }
  for (int i = tid; i < n_sink; i += 256) {
    const float x = __half2float(sink[h * n_sink + i]);
    const float mn = fmaxf(M, x);
    Z = Z * expf(M - mn) + expf(x - mn);
    M = mn;
  }
#pragma unroll
  for (int d = 32; d >= 1; d >>= 1) {
    const float mo = __shfl_xor(M, d), zo = __shfl_xor(Z, d);
    const float mn = fmaxf(M, mo);
    Z = (mn == -INFINITY) ? 0.f : Z * mz_w(M - mn) + zo * mz_w(mo - mn);
    M = mn;
  }
  if ((tid & 63) == 0) { red[tid >> 6] = M; red[4 + (tid >> 6)] = Z; }
  __syncthreads();
  float Mb = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
  float Zb = 0.f;
#pragma unroll
  for (int i = 0; i < 4; i++) if (red[i] > -INFINITY) Zb += red[4 + i] * mz_w(red[i] - Mb);
  if (tid == 0) { mz[2 * h] = Mb; mz[2 * h + 1] = Zb; }
  for (int i = tid; i < n_sink; i += 256)
    sink_probs[h * n_sink + i] = __float2half_rn(prob_fp16(__half2float(sink[h * n_sink + i]), Mb, 1.0f / Zb));
  if (v_sink != nullptr && n_sink > 0)
    for (int c = tid; c < kHeadDim; c += 256) sink_out[h * kHeadDim + c] = sink_output(sink, v_sink, n_sink, h, c, Mb, 1.0f / Zb);
}

struct Plan {
  int64_t tr;
  int n_ranges, groups, n_units;
  size_t bytes;          // (room for the extra slabs of the token-split outlier phase: n_ranges * groups slabs at q_len = 1)
};

template <int BITS>
static Plan plan_mix(int q_len, int H, int64_t L) {
  using Cfg = VCfg<BITS>;
  Plan pl;
  pl.n_units = H * Cfg::UPH;
  pl.groups = (pl.n_units + Cfg::UW - 1) / Cfg::UW;
  static const int wgs_rt = [] {
    const char *e = getenv("KVQ_V_WGS_RT");          // (A/B runs)
    return e ? atoi(e) : 0;
  }();
  // KVQ_V_WGS / 256 workgroups of 512 lanes per CU.  One unit group (3 / 2 bit: every workgroup covers all heads, 16-token
  // chunks) up to 48K tokens: one workgroup per CU -- its fixed work (the partials of 32 heads, the outlier phase, a 16 KB
  // slab) is then spread over 8 chunks instead of 4 (32K nuq3 + 5 sinks 3.24 -> 3.18 ms / step, nuq2 2.82 -> 2.73; at 128K
  // 512 workgroups win by 11 us per launch (64K: by 3.5; 48K: a tie; 16K: 256 win by 5), and two unit groups (4 bit) want 512 at every length:
  // profiles/r05_pv_plan.txt)
  const int wgs = wgs_rt > 0 ? wgs_rt : ((pl.groups == 1 && L <= 49152) ? KVQ_V_WGS / 2 : KVQ_V_WGS);
  int64_t want = wgs / pl.groups;
  if (want < 1) want = 1;
  int64_t tr = (L + want - 1) / want;
  tr = (tr + Cfg::CT - 1) / Cfg::CT * Cfg::CT;
  if (tr < 2 * Cfg::CT) tr = 2 * Cfg::CT;
  pl.tr = tr;
  pl.n_ranges = (int)((L + tr - 1) / tr);
  if (pl.n_ranges < 1) pl.n_ranges = 1;
  pl.bytes = (size_t)pl.n_ranges * (q_len == 1 ? pl.groups : q_len) * H * kHeadDim * sizeof(float) +
             (size_t)H * 2 * sizeof(float);   // + the (max, normaliser) pairs of the fused softmax
  return pl;
}

constexpr int kMergeInKernelParts = KVQ_V_MERGE_PARTS;   // score tiles (256 tokens each)
// KVQ_V_SPLIT=0: the per-group rows phase at every length (A/B runs)
static bool split_enabled() {
  static const bool on = [] {
    const char *e = getenv("KVQ_V_SPLIT");
    return !(e && e[0] == '0');
  }();
  return on;
}

// Up to how many score tiles the p.V workgroups merge the softmax partials themselves.  With sink tokens whose values are
// added here (v_sink) the merge kernel runs at every length: in the in-kernel variant workgroup 0 alone writes the sink
// probabilities and the sink tokens' 4096 outputs before its own range, and the launch waits for it (32K nuq4 + 5 sinks:
// p.V 51 us against 37 without sinks).  KVQ_V_MERGE_PARTS_RT overrides the threshold (A/B runs).
static int merge_in_kernel_parts(bool sink_outputs) {
  static const int rt = [] {
    const char *e = getenv("KVQ_V_MERGE_PARTS_RT");
    return e ? atoi(e) : -1;
  }();
  if (rt >= 0) return rt;
  return sink_outputs ? 0 : kMergeInKernelParts;
}

// kvq_mix_v_wide.hip: the one-workgroup-per-CU geometry with the outlier entries evaluated inside the dense loop
bool mix_wide_ok(int bits, const MixArgs &a);
int launch_mix_wide_bits(int bits, const MixArgs &a, float *mul, int accumulate, hipStream_t st, const FusedSoftmax *fs,
                         const float *mz, int *n_slabs_out);
// from how many cached tokens the wide geometry runs (KVQ_V_WIDE_FROM; KVQ_V_WIDE=0: never)
static int64_t wide_from() {
  static const int64_t v = [] {
    const char *off = getenv("KVQ_V_WIDE");
    if (off && off[0] == '0') return (int64_t)1 << 62;
    const char *e = getenv("KVQ_V_WIDE_FROM");
    return e ? (int64_t)atoll(e) : (int64_t)KVQ_V_WIDE_FROM;
  }();
  return v;
}

template <int BITS>
static int launch_mix(MixArgs a, float *mul, int accumulate, hipStream_t st, const FusedSoftmax *fs = nullptr) {
  using Cfg = VCfg<BITS>;
  Plan pl = plan_mix<BITS>(a.q_len, a.H, a.L);
  if (a.L >= wide_from() && mix_wide_ok(BITS, a)) {
    // (the workspace is the 512-lane plan's: the wide plan writes fewer slabs; the (max, normaliser) pairs keep their place)
    const float *mzp = nullptr;
    // the merge kernel: sink tokens (whose probabilities and outputs it writes), or very many score tiles.  Otherwise the 256
    // workgroups merge the partials themselves, 32 lanes per head behind their first requests (KVQ_W_MERGE_PARTS tiles =
    // 256K tokens: 16 loads per lane at 128K, from the L2 -- 32 MB of reads per launch against a 4.8 us launch)
    static const int wide_merge_parts = [] {
      const char *e = getenv("KVQ_W_MERGE_PARTS_RT");          // (A/B runs)
      return e ? atoi(e) : KVQ_W_MERGE_PARTS;
    }();
    if (fs && (fs->n_sink > 0 || fs->n_parts > wide_merge_parts)) {
      float *mz = a.partial + (size_t)pl.n_ranges * (a.q_len == 1 ? pl.groups : a.q_len) * a.H * kHeadDim;
      softmax_merge_kernel<<<a.H, 256, 0, st>>>(fs->parts, fs->n_parts, fs->sink, fs->sink_probs, fs->n_sink, mz, fs->v_sink, mul);
      int rc0 = check_launch();
      if (rc0) return rc0;
      if (fs->v_sink != nullptr) accumulate = 1;     // the reduce adds the slabs onto the sink tokens' output
      mzp = mz;
    }
    int n_slabs = 0;
    int rc = launch_mix_wide_bits(BITS, a, mul, accumulate, st, fs, mzp, &n_slabs);
    if (rc) return rc;
    const int C = a.H * kHeadDim;
    dim3 rgrid((C + 15) / 16, a.q_len);
    mix_v_reduce_kernel<<<rgrid, 1024, 0, st>>>(a.partial, mul, n_slabs, a.q_len, C, accumulate);
    return check_launch();
  }
  a.tr = pl.tr;
  a.groups = pl.groups;
  a.n_units = pl.n_units;
  dim3 grid(pl.n_ranges * pl.groups, 1, a.q_len), block(Cfg::NT);
  // Outlier entries split by tokens between the unit groups of a range (sparse_phase_all): long caches, where the phase is
  // bound by the bytes every group re-reads (128K nuq4: see DESIGN.md); short ones keep the per-group rows (the groups
  // merge the softmax partials of their own heads only, and the extra slabs would double the reduce's work)
  const int C_all = a.H * kHeadDim;
  const bool merge_first = fs && fs->n_parts > merge_in_kernel_parts(fs->n_sink > 0 && fs->v_sink != nullptr);
  a.split = (a.idx != nullptr && a.q_len == 1 && pl.groups >= 2 && C_all <= (Cfg::SMEM_B - Cfg::SP_P_B) / 8 - 64 &&
             a.H <= Cfg::SP_P_B / (4 * 65) && (fs ? (merge_first && fs->n_parts > kMergeInKernelParts) : a.L >= 65536) &&
             split_enabled()) ? 1 : 0;
  const int n_slabs = a.split ? pl.n_ranges * pl.groups : pl.n_ranges;
  if (fs) {
    a.scores = fs->scores;
    a.inv = fs->inv;
    a.parts = fs->parts;
    a.n_parts = fs->n_parts;
    a.sink = fs->sink;
    a.sink_probs = fs->sink_probs;
    a.n_sink = fs->n_sink;
    a.v_sink = fs->v_sink;
    a.sink_out = mul;
    if (fs->v_sink != nullptr) accumulate = 1;     // the reduce adds the slabs onto the sink tokens' output
    a.mz = nullptr;
    if (merge_first) {
      float *mz = a.partial + (size_t)pl.n_ranges * (a.q_len == 1 ? pl.groups : a.q_len) * a.H * kHeadDim;   // (tail of the workspace)
      softmax_merge_kernel<<<a.H, 256, 0, st>>>(fs->parts, fs->n_parts, fs->sink, fs->sink_probs, fs->n_sink, mz,
                                                fs->v_sink, mul);
      int rc0 = check_launch();
      if (rc0) return rc0;
      a.mz = mz;
    }
    kvq_step_mark_pv(st);      // (measurement hook of kvq_decode_step: the p.V kernel starts here)
    mix_v_kernel<BITS, true><<<grid, block, 0, st>>>(a);
  } else {
    mix_v_kernel<BITS, false><<<grid, block, 0, st>>>(a);
  }
  int rc = check_launch();
  if (rc) return rc;
  const int C = a.H * kHeadDim;
  dim3 rgrid((C + 15) / 16, a.q_len);
  mix_v_reduce_kernel<<<rgrid, 1024, 0, st>>>(a.partial, mul, n_slabs, a.q_len, C, accumulate);
  return check_launch();
}

int launch_mix_reduce(const float *partial, float *mul, int n_ranges, int q_len, int C, int accumulate, hipStream_t st) {
  dim3 rgrid((C + 15) / 16, q_len);
  mix_v_reduce_kernel<<<rgrid, 1024, 0, st>>>(partial, mul, n_ranges, q_len, C, accumulate);
  return check_launch();
}
int launch_softmax_merge(const float *parts, int n_parts, const void *sink, void *sink_probs, int n_sink, float *mz,
                         const void *v_sink, float *sink_out, int H, hipStream_t st) {
  softmax_merge_kernel<<<H, 256, 0, st>>>(parts, n_parts, reinterpret_cast<const __half *>(sink),
                                          reinterpret_cast<__half *>(sink_probs), n_sink, mz,
                                          reinterpret_cast<const __half *>(v_sink), sink_out);
  return check_launch();
}
int mix_merge_in_kernel_parts() { return kMergeInKernelParts; }

size_t mix_plan_bytes(int bits, int q_len, int H, int64_t L) {
  return bits == 4 ? plan_mix<4>(q_len, H, L).bytes : (bits == 3 ? plan_mix<3>(q_len, H, L).bytes : plan_mix<2>(q_len, H, L).bytes);
}
int launch_mix_bits(int bits, MixArgs a, float *mul, int accumulate, hipStream_t st, const FusedSoftmax *fs) {
  switch (bits) {
    case 4: return launch_mix<4>(a, mul, accumulate, st, fs);
    case 3: return launch_mix<3>(a, mul, accumulate, st, fs);
    default: return launch_mix<2>(a, mul, accumulate, st, fs);
  }
}

}  // namespace kvq
