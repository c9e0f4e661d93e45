// softmax(p).V over the packed NUQ value cache, long-cache decode form (round 6): ONE 1024-lane workgroup per CU, the
// sparse-outlier entries of a chunk evaluated INSIDE the dense loop.  Reference semantics: KCU:3211-3433 (+4117-4491,
// 4998-5248) fused with SPMV_ATOMIC_BALANCED KCU:437-470; launchers KCU:3491-3538, 3625-3690.
//
// Why a second geometry next to kvq_mix_v.hip.  That kernel (512 lanes, two workgroups per CU) streams its rows at the
// rate HBM delivers while its dense loop runs (two 37 KB stages per workgroup: 6.1 TB/s), but evaluates the outlier
// entries of its token range in a phase of its own behind the loop -- 24 us of a 78 us wave during which nothing streams
// (profiles/r05_pv_outlier_phase.txt) -- because a two-workgroup CU has no LDS left for accumulators while the stages are
// live.  One workgroup per CU has: the chunk's 42 entries per token (1 % of its codes) are loaded one chunk ahead into
// four VGPRs per lane, multiplied with the chunk's probabilities -- which are in LDS for the dense loop anyway -- and
// added into 64-bit fixed-point LDS accumulators (ds_add_u64: exact, order independent, 17 cycles per wave instruction;
// ds_add_f32 costs 171 and a global atomic 1500: profiles/r06_atomic_rate.txt) while the look-ups of the chunk run.
// No trailing phase, no second read of the scores, no extra slabs; every entry is read once.
//   * 4 bit: a chunk of 32 tokens is consumed as TWO sub-stages of 256 row units (a lane owns one unit of each), the
//     tile ring has THREE slots of 32 KB: the rows of sub-stage s + 2 are requested while s is decoded, i.e. a full
//     sub-stage of look-ahead more than double buffering gives (the 512-lane kernel waits 543 cycles per chunk for the
//     piece it issued last).  Counted waits (memory operations return in order): every wave issues the same number of
//     operations per sub-stage, so `s_waitcnt vmcnt(N)` with a constant N per (sub-stage, wave class) covers exactly the
//     rows, codebook rows, scores and entries the sub-stage needs.
//   * 3 / 2 bit: one sub-stage per chunk (all 4096 channels: 128 units x 2 halves / 256 units, four token slots), two
//     slots, 32-token chunks -- 128-byte row segments instead of the 64-byte ones of the 512-lane geometry, whose second
//     halves missed the L2 four times in ten (nuq3 p.V fetched 1.39x its algorithmic bytes, profiles/r05_z_cfg3_pmc_bench.txt).
// The look-up sequences are those of kvq_mix_lut.h; the slab reduce and the softmax merge are kvq_mix_v.hip's.
// Needs q_len = 1, max_len % 4 == 0, n_out <= 64, and (fused softmax) the merged (max, normaliser) pairs.
#include "kvq_common.h"
#include "kvq_host.h"
#include "kvq_mix_v_stage.h"
#include "kvq_mix_lut.h"

#include <hip/hip_fp16.h>

#include <cstdlib>

// development switches (tools/abl/build_var.sh; timing only unless noted)
#ifndef KVQ_W_BURST
#define KVQ_W_BURST 0     // 1: the next tile's pieces in one burst behind the barrier instead of a piece per quad
#endif
#ifndef KVQ_W_NT
#define KVQ_W_NT 0        // 1: the tile rows with the non-temporal load policy
#endif
#ifndef KVQ_W_PRIO
#define KVQ_W_PRIO 3      // 0: none, 1: wave priority = token slot (the youngest wave of a SIMD highest), 2: the reverse,
                          // 3 (default): slot in a sub-stage's first quad, the reverse in its second -- every wave is favoured half of
                          // the time, the waves of a SIMD reach the barrier together (barrier wait 1815 -> 1073 cycles per chunk; the loop is
                          // throughput-bound, so the kernel gains little: nuq3 73.3 -> 71.4 us, nuq4 75.3 -> 74.8, profiles/r06_o_prio3.txt)
#endif
#ifndef KVQ_W_ABL
#define KVQ_W_ABL 0       // 1: no outlier evaluation, 2: no look-up loop, 4: no tile DMA (results wrong)
#endif

namespace kvq {

template <int BITS>
struct WCfg {
  static constexpr int N = Fmt<BITS>::kN;
  static constexpr int WORDS = Unit<BITS>::kWords;
  static constexpr int CH = Unit<BITS>::kCh;
  static constexpr int UPH = kHeadDim / CH;              // units per head
  static constexpr int NT = 1024, NW = NT / 64;
  static constexpr int NU = BITS == 4 ? 2 : 1;           // sub-stages per chunk
  static constexpr int NS = NU + 1;                      // slots of the tile ring
  static constexpr int SU = BITS == 3 ? 128 : 256;       // row units per sub-stage
  static constexpr int HALVES = BITS == 3 ? 2 : 1;
  static constexpr int CHL = CH / HALVES;                // channels per lane and sub-stage
  static constexpr int LPS = SU * HALVES;                // lanes per token slot (256)
  static constexpr int SLOTS = NT / LPS;                 // token slots (4)
  static constexpr int CT = 32, QR = CT / 4, SH = 1;
  static constexpr int QPL = QR / SLOTS;                 // quads per lane and sub-stage (2)
  static constexpr int ROWS = SU * WORDS, ROWB = CT * 4;
  static constexpr int TILE_B = ROWS * ROWB;             // 32 / 48 / 32 KB
  static constexpr int GU = SU * NU;                     // units per unit group
  static constexpr int GC = GU * CH;                     // channels per unit group (4096)
  static constexpr int HW = GC / kHeadDim;               // heads per unit group (32)
  static constexpr int LUT_B = CT * N * 4;
  // probabilities of a chunk: [head][token], the rows of two consecutive heads adjacent (one 256-byte DMA piece), the
  // pieces P_PIECE = 288 bytes apart: the lanes of a ds_read_b128 group cover four consecutive heads, and with 256-byte
  // pieces heads h and h + 2 sat in the same banks (a two-way conflict on every probability read: 2.1 M of the kernel's
  // 23 M LDS cycles, profiles/r06_g_pmc_bench.txt)
  static constexpr int P_PIECE = 288;
  static constexpr int P_B = HW / 2 * P_PIECE;           // 4.5 KB; one element per lane
  static constexpr int p_byte(int h, int t) { return (h >> 1) * P_PIECE + (h & 1) * (CT * 4) + t * 4; }
  static constexpr int NPB = 3;
  static constexpr int lut_off(int b) { return b * LUT_B; }
  static constexpr int p_off(int b) { return 2 * LUT_B + b * P_B; }
  static constexpr int tile_off(int s) { return 2 * LUT_B + NPB * P_B + s * TILE_B; }
  static constexpr int ACC_OFF = tile_off(NS);           // 64-bit accumulators of the group's channels + 64 dummies
  static constexpr int SMEM_B = ACC_OFF + GC * 8 + 512;
  static constexpr int RED_B = NT * CHL * NU * 4;        // slot reduction (aliases the tile ring)
  // VMEM operations a wave issues per sub-stage (the counted waits rest on these)
  static constexpr int T_OPS = TILE_B / 1024 / NW;       // tile pieces per wave
  static constexpr int LUT_LANES = LUT_B / 16;           // lanes that fetch 16 B of the codebook rows
  static constexpr int E_R = 2;                          // entry rounds (n_out * CT <= E_R * NT)
  static constexpr int E_OPS = 2 * E_R;                  // entry loads per wave and chunk (index + value)
  static_assert(HW * CT == NT, "one probability per lane and chunk");
  static_assert(TILE_B % (1024 * NW) == 0, "every wave issues the same number of tile pieces");
  static_assert(RED_B <= NS * TILE_B, "slot sums alias the tile ring");
  static_assert(SMEM_B <= 160 * 1024, "one workgroup per CU");
  static_assert(SLOTS * N * 4 <= 256, "the slot's row offset rides in the look-up byte");
};

// per-lane constants of the DMA (kvq_mix_v_stage.h: DmaLane, here for 1024 lanes and per-sub-stage tiles)
template <int BITS>
__device__ __forceinline__ DmaLane wide_dma_lane() {
  using Cfg = WCfg<BITS>;
  DmaLane d;
  const int s = threadIdx.x;
  const int r = s / Cfg::QR, pos = s % Cfg::QR;
  d.tile_row = r;
  d.tile_q4 = 4 * ((pos - ((r >> Cfg::SH) & (Cfg::QR - 1))) & (Cfg::QR - 1));
  const int pidx = (s * 4) / Cfg::N;
  const int qe = pidx / Cfg::SLOTS, slp = pidx % Cfg::SLOTS;
  d.lut_tok = (slp * Cfg::QPL + qe / 4) * 4 + qe % 4;
  d.lut_sub = (s * 4) % Cfg::N;
  d.p_head = s / Cfg::CT;
  d.p_tok = s % Cfg::CT;
  return d;
}

struct WideArgs {
  MixArgs m;
  int n_chunks_all;     // chunks of CT tokens in [0, L)
  int n_ranges;
};

// piece K (of T_OPS per wave) of the tile of a sub-stage -- tokens [c0, c0 + CT), rows [row0, row0 + n_rows_valid) -- -> ring
// slot at LDS byte `dst`.  ONE instruction on every path (the counted waits and tools/check_isa.py rest on it): a chunk
// that needs clamps (ragged end of the rows, partial unit group) only changes the lane's source offset.
template <int BITS, int K>
__device__ __forceinline__ void wide_tile_piece(const MixArgs &a, const DmaLane &d, uint32_t f_tile, int64_t c0, int row0,
                                                int n_rows_valid, uint32_t dst, int wave, bool fast) {
  using Cfg = WCfg<BITS>;
  constexpr int RPI = 64 / Cfg::QR;
  uint32_t voff = f_tile + (uint32_t)(K * Cfg::NW * RPI) * (uint32_t)a.max_len * 4u;
  if (!fast) {   // (wave-uniform; VALU only)
    const int lim_len = (int)(a.max_len - c0);
    const int tq = (int)d.tile_q4;
    const uint32_t toff = (uint32_t)(tq + 4 > lim_len ? lim_len - 4 : tq);
    int r = d.tile_row + K * Cfg::NW * RPI;
    if (r >= n_rows_valid) r = n_rows_valid - 1;
    voff = ((uint32_t)r * (uint32_t)a.max_len + toff) * 4u;
  }
#if KVQ_W_NT
  dma16_nt(a.mat + (int64_t)row0 * a.max_len + c0, voff, dst + (wave + K * Cfg::NW) * 1024);
#else
  dma16(a.mat + (int64_t)row0 * a.max_len + c0, voff, dst + (wave + K * Cfg::NW) * 1024);
#endif
}
// codebook rows of chunk c0 -> LDS `dst` (waves below LUT_LANES / 64; clamped to the rows that exist)
template <int BITS>
__device__ __forceinline__ void wide_lut(const MixArgs &a, const DmaLane &d, int64_t c0, uint32_t dst, int wave, bool fast,
                                         uint32_t voff_fast) {
  using Cfg = WCfg<BITS>;
  if ((int)threadIdx.x < Cfg::LUT_LANES) {
    const int lim_len = (int)(a.max_len - c0);
    const int tr2 = (int)d.lut_tok < lim_len ? (int)d.lut_tok : lim_len - 1;
    const uint32_t voff = fast ? voff_fast : ((uint32_t)tr2 * Cfg::N + d.lut_sub) * 4u;
    dma16(a.lut_rows + c0 * Cfg::N, voff, dst + wave * 1024);
  }
}
// probabilities / raw scores of the group's heads for chunk c0 -> p buffer `dst`: one 256-byte piece per wave
template <int BITS>
__device__ __forceinline__ void wide_p(const float *src, const MixArgs &a, const DmaLane &d, int64_t c0, int h0, uint32_t dst,
                                       int wave) {
  const int lim_L = (int)(a.L - c0);
  const float *gbase = src + (int64_t)h0 * a.L + (lim_L <= 0 ? a.L - 1 : c0);
  const uint32_t toff = (uint32_t)(lim_L <= 0 ? 0 : ((int)d.p_tok < lim_L ? (int)d.p_tok : lim_L - 1));
  int hr = (int)d.p_head;
  if (h0 + hr >= a.H) hr = a.H - 1 - h0;
  dma4(gbase, ((uint32_t)hr * (uint32_t)a.L + toff) * 4u, dst + wave * WCfg<BITS>::P_PIECE);
}

// outlier entries of a chunk, in flight in registers for a whole chunk: hand-issued loads (hipcc must not put them on
// its scoreboard: it would wait for them -- and for the DMA pieces issued behind them -- at the loop's back edge);
// tools/check_isa.py verifies on the generated code that nothing touches the registers before the covering wait
__device__ __forceinline__ void entry_load(uint32_t &dst, const void *sbase, uint32_t voff) {
  asm volatile("global_load_dword %0, %1, %2" : "=v"(dst) : "v"(voff), "s"(sbase) : "memory");
}

template <int BITS, bool FUSED>
__global__ __launch_bounds__(1024) void mix_v_wide_kernel(WideArgs wa) {
  using Cfg = WCfg<BITS>;
  const MixArgs &a = wa.m;
  constexpr int N = Cfg::N, CH = Cfg::CH, WORDS = Cfg::WORDS, CT = Cfg::CT, CHL = Cfg::CHL, NU = Cfg::NU, NS = Cfg::NS;
  __shared__ __attribute__((aligned(16))) unsigned char smem[Cfg::SMEM_B];
  const int tid = threadIdx.x;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int ul = tid % Cfg::SU;
  const int hf = __builtin_amdgcn_readfirstlane((tid / Cfg::SU) % Cfg::HALVES);
  const int lu = tid % Cfg::LPS;
  const int sl = __builtin_amdgcn_readfirstlane(tid / Cfg::LPS);
  // block -> (range, unit group): the groups of a range 8 blocks apart (same XCD: they read the same entries and rows)
  int g, range;
  {
    const int G = a.groups, n_ranges = wa.n_ranges;
    const int chunk = (int)blockIdx.x / (8 * G), r = (int)blockIdx.x % (8 * G);
    const int nr = (n_ranges - 8 * chunk < 8) ? (n_ranges - 8 * chunk) : 8;
    g = r / nr;
    range = 8 * chunk + r % nr;
  }
  const int C = a.H * kHeadDim;
  const int u0 = g * Cfg::GU;
  int n_units_valid = a.n_units - u0;
  if (n_units_valid > Cfg::GU) n_units_valid = Cfg::GU;
  const int h0 = u0 / Cfg::UPH;
  // the range's chunks: the launch's chunks dealt out evenly (the first `rem` ranges take one more)
  int n_chunks;
  int64_t t0;
  {
    const int q = wa.n_chunks_all / wa.n_ranges, rem = wa.n_chunks_all % wa.n_ranges;
    const int first = range * q + (range < rem ? range : rem);
    n_chunks = q + (range < rem ? 1 : 0);
    t0 = (int64_t)first * CT;
  }
  const int64_t t1 = (t0 + (int64_t)n_chunks * CT < a.L) ? t0 + (int64_t)n_chunks * CT : a.L;
  if (lds_addr(smem) != 0) __builtin_trap();   // (the instruction immediates below assume the one static LDS array at 0)
#if KVQ_W_PRIO == 1 || KVQ_W_PRIO == 2
  {
    const int pr = KVQ_W_PRIO == 1 ? sl : 3 - sl;
    if (pr == 1) __builtin_amdgcn_s_setprio(1);
    else if (pr == 2) __builtin_amdgcn_s_setprio(2);
    else if (pr == 3) __builtin_amdgcn_s_setprio(3);
  }
#endif

#if KVQ_TRACE
  // development: cycles per wave in [0] data waits, [1] barriers, [2] entries + look-ahead issue + score conversion,
  // [3] look-up loop, [4] prologue, [5] epilogue (tools/dbg/trace_vw.py)
  unsigned tr_acc[6] = {0, 0, 0, 0, 0, 0};
  unsigned tr_prev;
  unsigned long long tr_t0, tr_r0;
  {
    unsigned long long tt;
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(tt)::"memory");
    tr_prev = (unsigned)tt;
    tr_t0 = tt;
    asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(tr_r0)::"memory");
  }
  auto stamp = [&](int k) {
    unsigned long long tt;
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(tt)::"memory");
    tr_acc[k] += (unsigned)tt - tr_prev;
    tr_prev = (unsigned)tt;
  };
#define KVQ_W_STAMP(k) stamp(k)
#else
#define KVQ_W_STAMP(k)
#endif
  const DmaLane dl = wide_dma_lane<BITS>();
  const uint32_t f_tile = (dl.tile_row * (uint32_t)a.max_len + dl.tile_q4) * 4u;
  const uint32_t f_lut = (dl.lut_tok * Cfg::N + dl.lut_sub) * 4u;
  const bool sparse = a.idx != nullptr;
  const bool compact = a.outliers == nullptr;
  // chunks whose DMA needs no clamps: all that start at or before `fast_end`, for a full unit group
  int64_t fast_end = (a.max_len < a.L ? a.max_len : a.L) - CT;
  if (n_units_valid != Cfg::GU) fast_end = -1;
  const float *psrc = FUSED ? a.scores : a.p;

  // ---- outlier entries of a chunk: entry e = round * NT + tid of its CT * n_out entries
  const int NE = sparse ? CT * a.n_out : 0;
  uint32_t e_idx[Cfg::E_R], e_val[Cfg::E_R];
  // round r of the chunk at c0 (clamped to the rows that exist; without sparse rows: the same loads from a valid address)
  auto issue_entry_round = [&](int r, int64_t c0) {
    const int64_t lim64 = (a.max_len - c0) * (int64_t)a.n_out;
    const unsigned lim = lim64 > 0x7fffffff ? 0x7fffffffu : (unsigned)lim64;        // (wave-uniform)
    const void *bi = sparse ? static_cast<const void *>(a.idx + c0 * a.n_out) : static_cast<const void *>(a.lut_rows);
    const void *bv = sparse && !compact ? static_cast<const void *>(a.outliers + c0 * a.n_out) : bi;
    unsigned e = (unsigned)(r * Cfg::NT + tid);
    if (!sparse) e = 0;
    else if (e >= lim) e = lim - 1;
    entry_load(e_idx[r], bi, e * 4u);
    entry_load(e_val[r], bv, e * 4u);
  };
  auto hold_entries = [&]() {
#pragma unroll
    for (int r = 0; r < Cfg::E_R; r++) asm volatile("" ::"v"(e_idx[r]), "v"(e_val[r]));
  };
  const int c_lo = u0 * CH, cn = n_units_valid * CH;
  // Evaluation of one round of entries, in two steps that ride between the look-up batches of the dense loop (every wave of
  // the workgroup is in the same phase at the same time: as a block of its own behind the barrier this arithmetic ran with
  // the LDS idle -- 2400 of 8750 cycles per chunk, profiles/r06_c_wide_trace.txt):
  //   ent_issue(r):  entry -> (token, channel, head), the LDS read of its probability (hand-issued: lands behind the batch
  //                  in flight, covered by the loop's next counted wait)
  //   ent_commit():  value * probability -> 32.32 fixed point -> ds_add_u64 into the channel's accumulator
  float ent_pt = 0.f;
  uint32_t ent_acc = 0;
  auto ent_issue = [&](int r, int pbuf, int64_t c0) {
    asm volatile("" : "+v"(e_idx[r]), "+v"(e_val[r]));       // (landed: the sub-stage's wait covers them)
    const unsigned e = (unsigned)(r * Cfg::NT + tid);
    const unsigned tl = __umulhi(e, a.n_out_magic);           // token within the chunk
    const uint32_t w = e_idx[r];
    const unsigned ch = compact ? (w & 0xffffu) : w;
    const unsigned rel = ch - (unsigned)c_lo;
    const unsigned rem = (unsigned)(t1 - c0);                  // tokens of the range left from the chunk's start (wave-uniform, >= 1)
    const bool mine = (int)e < NE && rel < (unsigned)cn && tl < rem;
    const unsigned hh = mine ? (rel >> 7) : 0u;
    const uint32_t p_addr = (uint32_t)(Cfg::p_off(0) + pbuf * Cfg::P_B) + (hh >> 1) * (unsigned)Cfg::P_PIECE + (hh & 1u) * (CT * 4u) +
                            (mine ? tl : 0u) * 4u;
    // (not mine -- another group's channel, a padding slot, a token past the range: its product goes to a dummy accumulator)
    ent_acc = (uint32_t)Cfg::ACC_OFF + (mine ? rel : (unsigned)Cfg::GC + (tid & 63u)) * 8u;
    asm volatile("ds_read_b32 %0, %1" : "=v"(ent_pt) : "v"(p_addr) : "memory");
  };
  auto ent_commit = [&](int r) {
    asm volatile("" : "+v"(ent_pt));                          // (landed: a counted wait of the loop has passed)
    const float v = compact ? __half2float(__ushort_as_half((unsigned short)(e_idx[r] >> 16))) : __uint_as_float(e_val[r]);
    // x -> 32.32 fixed point without 64-bit float math: floor part + exact 32-bit fraction
    const float x = v * ent_pt;
    const float fl = floorf(x);
    const unsigned lo = (unsigned)((x - fl) * 4294967296.0f);
    const int hi = (int)fl;
    const unsigned long long fx = ((unsigned long long)(unsigned)hi << 32) | lo;
    asm volatile("ds_add_u64 %0, %1" ::"v"(ent_acc), "v"(fx) : "memory");
  };
  static_assert(Cfg::E_R == 2 && Cfg::QPL == 2, "one round of entries per quad");

  // ---- fused softmax: the (max, 1 / normaliser) of the head whose scores this lane converts -- from the merge kernel's
  // pairs (a.mz), or, few score tiles and no sink tokens, merged here from the score kernel's partials by the 32 lanes
  // that convert the head's scores (one launch less: what a short cache's step is made of)
  float myM = 0.f, myZ = 1.f;
  auto load_mz = [&]() {
    if constexpr (FUSED) {
      int hc = h0 + tid / CT;
      if (hc >= a.H) hc = a.H - 1;
      if (a.mz != nullptr) {
        const float2 t = reinterpret_cast<const float2 *>(a.mz)[hc];
        myM = t.x;
        myZ = 1.0f / t.y;
      } else {
        static_assert(CT == 32, "a head's scores of a chunk are converted by half a wave");
        const float2 *pr = reinterpret_cast<const float2 *>(a.parts) + (int64_t)hc * a.n_parts;
        float M = -INFINITY, Z = 0.f;
        constexpr int MB = 8;
        for (int i0 = tid & 31; i0 < a.n_parts; i0 += 32 * MB) {
          float2 ms[MB];
#pragma unroll
          for (int k = 0; k < MB; k++) ms[k] = (i0 + 32 * k < a.n_parts) ? pr[i0 + 32 * k] : make_float2(-INFINITY, 0.f);
#pragma unroll
          for (int k = 0; k < MB; k++)
            if (ms[k].x > -INFINITY) {
              const float mn = fmaxf(M, ms[k].x);
              Z = Z * mz_w(M - mn) + ms[k].y * mz_w(ms[k].x - mn);
              M = mn;
            }
        }
#pragma unroll
        for (int d = 16; d >= 1; d >>= 1) {
          const float mo = __shfl_xor(M, d), zo = __shfl_xor(Z, d);
          const float mn = fmaxf(M, mo);
          Z = (mn == -INFINITY) ? 0.f : Z * mz_w(M - mn) + zo * mz_w(mo - mn);
          M = mn;
        }
        myM = M;
        myZ = 1.0f / Z;
      }
    }
  };
  // (element tid of a p buffer = what this lane's DMA piece brought: piece tid / 64, lane tid % 64)
  const uint32_t p_elem = (uint32_t)((tid >> 6) * Cfg::P_PIECE + (tid & 63) * 4);
  auto convert_p = [&](int pbuf, int64_t c0) {
    float *pp = reinterpret_cast<float *>(smem + Cfg::p_off(0) + pbuf * Cfg::P_B + p_elem);
    const float x = *pp;
    *pp = (c0 + (tid % CT) < t1) ? prob_of(x, a.inv, myM, myZ) : 0.f;
  };

  // the same conversion riding in the loop: the raw score is read behind one look-up batch, converted behind others
  float cv_raw = 0.f;
  auto cv_issue = [&](int pbuf) {
    const uint32_t addr = (uint32_t)(Cfg::p_off(0) + pbuf * Cfg::P_B) + p_elem;
    asm volatile("ds_read_b32 %0, %1" : "=v"(cv_raw) : "v"(addr) : "memory");
  };
  auto cv_commit = [&](int pbuf, int64_t c0) {
    asm volatile("" : "+v"(cv_raw));
    const uint32_t addr = (uint32_t)(Cfg::p_off(0) + pbuf * Cfg::P_B) + p_elem;
    const float pr = (c0 + (tid % CT) < t1) ? prob_of(cv_raw, a.inv, myM, myZ) : 0.f;
    asm volatile("ds_write_b32 %0, %1" ::"v"(addr), "v"(pr) : "memory");
  };

  // ---- issue of everything a chunk needs besides its tiles: codebook rows, probabilities / scores, entries
  auto issue_chunk_extras = [&](int cnext, int lut_buf, int pbuf_next) {
    const int64_t cn0 = t0 + (int64_t)cnext * CT;
    wide_lut<BITS>(a, dl, cn0, Cfg::lut_off(0) + lut_buf * Cfg::LUT_B, wave, cn0 <= fast_end, f_lut);
    // fused: the scores travel one chunk further ahead (converted during the chunk before they are used)
    const int64_t pc0 = FUSED ? cn0 + CT : cn0;
    wide_p<BITS>(psrc, a, dl, pc0, h0, Cfg::p_off(0) + pbuf_next * Cfg::P_B, wave);
  };
  // source of the tile of sub-stage u of the chunk at c0: first row, valid rows (a partial unit group; none: the group's
  // first rows, decoded by no writer), whether it needs no clamps
  struct TileSrc { int row0, nrv; bool fast; };
  auto tile_src = [&](int64_t c0, int u) {
    TileSrc t;
    t.row0 = (u0 + u * Cfg::SU) * WORDS;
    t.nrv = n_units_valid * WORDS - u * Cfg::ROWS;
    if (t.nrv < 1) {
      t.row0 = u0 * WORDS;
      t.nrv = n_units_valid * WORDS;
    }
    if (t.nrv > Cfg::ROWS) t.nrv = Cfg::ROWS;
    t.fast = c0 <= fast_end;
    return t;
  };
  auto issue_tile_all = [&](int cidx, int u, int slot) {
    const int64_t c0 = t0 + (int64_t)cidx * CT;
    const TileSrc ts = tile_src(c0, u);
    const uint32_t dst = Cfg::tile_off(0) + slot * Cfg::TILE_B;
    static_for<0, Cfg::T_OPS>([&](auto K) { wide_tile_piece<BITS, decltype(K)::value>(a, dl, f_tile, c0, ts.row0, ts.nrv, dst, wave, ts.fast); });
  };

  // ---- prologue: first tiles, codebook rows, scores, entries; behind the requests: the accumulators are zeroed
  issue_tile_all(0, 0, 0);
  wide_lut<BITS>(a, dl, t0, Cfg::lut_off(0), wave, t0 <= fast_end, f_lut);
  wide_p<BITS>(psrc, a, dl, t0, h0, Cfg::p_off(0), wave);
  if constexpr (FUSED) wide_p<BITS>(psrc, a, dl, t0 + CT, h0, Cfg::p_off(1), wave);
  issue_entry_round(0, t0);
  issue_entry_round(1, t0);
  if constexpr (NU == 2) issue_tile_all(0, 1, 1);          // (the last T_OPS operations: the loop's first wait leaves them in flight)
  {
    uint4 *z = reinterpret_cast<uint4 *>(smem + Cfg::ACC_OFF);
    for (int i = tid; i < (Cfg::GC * 8 + 512) / 16; i += Cfg::NT) z[i] = make_uint4(0u, 0u, 0u, 0u);
  }
  load_mz();                                               // (behind the first requests)
  if constexpr (NU == 2) vm_wait<Cfg::T_OPS>(); else vm_wait<0>();
  __syncthreads();
  if constexpr (FUSED) convert_p(0, t0);                // (visible after the first sub-stage's barrier)
  KVQ_W_STAMP(4);

  float acc[NU][CHL];
#pragma unroll
  for (int u = 0; u < NU; u++)
#pragma unroll
    for (int i = 0; i < CHL; i++) acc[u][i] = 0.f;

  // the lane's LDS addresses inside a tile slot / a p buffer
  uint32_t taddr[Cfg::QPL][WORDS];
#pragma unroll
  for (int qq = 0; qq < Cfg::QPL; qq++)
#pragma unroll
    for (int wi = 0; wi < WORDS; wi++) {
      const int r = ul * WORDS + wi, rot = (r >> Cfg::SH) & (Cfg::QR - 1);
      taddr[qq][wi] = (uint32_t)(r * Cfg::ROWB + (((sl * Cfg::QPL + qq + rot) & (Cfg::QR - 1)) << 4));
    }
  const uint32_t paddr = (uint32_t)(Cfg::p_byte(ul / Cfg::UPH, sl * Cfg::QPL * 4));     // (+ the sub-stage's heads, an immediate)

  int slot = 0;       // ring slot of the sub-stage being decoded
  int pcur = 0;       // p buffer of the chunk being decoded

  // one sub-stage.  LP: which codebook-row buffer (compile time: its offset is part of every look-up's immediate);
  // U: sub-stage within the chunk.
  // Every sub-stage issues its look-ahead UNCONDITIONALLY -- the last chunk of a range re-requests itself (L2 hits, landing in
  // ring slots and registers nobody reads) -- so that there is one path, on which the number of operations in flight at every
  // wait is a constant, and no control flow between a hand-issued load and the wait that covers it.
  auto substage = [&](auto LP_, auto U_, auto HI_, int c) {
    constexpr int LP = decltype(LP_)::value, U = decltype(U_)::value, HI = decltype(HI_)::value;
    const int64_t c0 = t0 + (int64_t)c * CT;
    const int cnx = (c + 1 < n_chunks) ? c + 1 : c;     // the chunk whose data this one requests
    // ---- the sub-stage's rows (and with U == 0 the chunk's codebook rows, probabilities, entries) have landed
    if constexpr (NS == 2) {
      vm_wait<0>();
    } else if constexpr (U == 0) {
      vm_wait<Cfg::T_OPS>();                          // in flight: the tile of sub-stage 1, issued during the previous sub-stage
    } else {
      // in flight: everything issued during sub-stage 0 of this chunk (entries, tile, codebook rows, scores)
      if (wave * 64 < Cfg::LUT_LANES) vm_wait<Cfg::E_OPS + Cfg::T_OPS + 2>(); else vm_wait<Cfg::E_OPS + Cfg::T_OPS + 1>();
    }
    KVQ_W_STAMP(0);
    __syncthreads();
    KVQ_W_STAMP(1);
    const int slot_next = (slot + NS - 1) % NS;        // ring slot of sub-stage s + NS - 1 (the one freed by the barrier)
    // Positions inside the look-up loop where the chunk's other work rides (K = 0: behind the quad's first wait, in front
    // of its first batch; K = 1..3: in front of the first batch of the quad's token K): each step issues at most one LDS
    // operation, immediately IN FRONT of a look-up batch -- operations return in order, so every counted wait that leaves
    // that batch in flight has waited for the step's operation too.
    //   4 bit (two sub-stages per chunk): sub-stage 0 evaluates entry round qq in quad qq, sub-stage 1 converts the next
    //   chunk's scores; 3 / 2 bit: everything in the one sub-stage.
    const bool conv_on = FUSED && c + 1 < n_chunks;
    auto hook = [&](auto QQ, auto K_) {
      constexpr int qq = decltype(QQ)::value, k = decltype(K_)::value;
#if KVQ_W_PRIO == 3
      if constexpr (k == 0) {
        const int pr = (qq & 1) ? 3 - sl : sl;
        if (pr == 0) __builtin_amdgcn_s_setprio(0);
        else if (pr == 1) __builtin_amdgcn_s_setprio(1);
        else if (pr == 2) __builtin_amdgcn_s_setprio(2);
        else __builtin_amdgcn_s_setprio(3);
      }
#endif
      constexpr int K_ENT = NU == 2 ? 2 : 1;                 // entries: issue at K_ENT, commit at K_ENT + 1
      if constexpr (NU == 1 || U == 0) {
        const bool on = sparse && qq * Cfg::NT + wave * 64 < NE && !(KVQ_W_ABL & 1);        // (this wave has entries in round qq)
        if constexpr (k == K_ENT) { if (on) ent_issue(qq, pcur, c0); }
        if constexpr (k == K_ENT + 1) {
          if (on) ent_commit(qq);
          // the round's registers are free: the same round of the next chunk (in flight until that chunk's first wait)
          issue_entry_round(qq, t0 + (int64_t)cnx * CT);
        }
      }
      if constexpr (FUSED && (NU == 1 || U == 1) && qq == Cfg::QPL - 1) {
        if constexpr (k == 2) { if (conv_on) cv_issue((pcur + 1) % 3); }
        if constexpr (k == 3) { if (conv_on) cv_commit((pcur + 1) % 3, c0 + CT); }
      }
    };
    constexpr std::integral_constant<int, 0> K0{};
    constexpr std::integral_constant<int, 1> K1{};
    constexpr std::integral_constant<int, 2> K2{};
    constexpr std::integral_constant<int, 3> K3{};
    if constexpr (U == 0) issue_chunk_extras(cnx, 1 - LP, FUSED ? (pcur + 2) % 3 : (pcur + 1) % 3);
    if (!FUSED && U == 0 && t1 - c0 < CT) {            // ragged last chunk: zero the probabilities past the end once
      const int rem = (int)(t1 - c0);
      if (tid % CT >= rem) *reinterpret_cast<float *>(smem + Cfg::p_off(0) + pcur * Cfg::P_B + p_elem) = 0.f;
      __syncthreads();
    }
    KVQ_W_STAMP(2);
    constexpr int S0 = Cfg::tile_off(0);
    constexpr int L0 = Cfg::lut_off(LP);
    static_assert((Cfg::SU / Cfg::UPH) % 2 == 0, "a sub-stage's heads are whole DMA pieces");
    constexpr int P0 = Cfg::p_off(0) + U * (Cfg::SU / Cfg::UPH / 2) * Cfg::P_PIECE;
    constexpr int TS = Cfg::SLOTS * N * 4;              // bytes between the rows of consecutive tokens of a slot
    const uint32_t tb = (uint32_t)(slot * Cfg::TILE_B);
    const uint32_t paddr_c = paddr + (uint32_t)(pcur * Cfg::P_B);
    // the tile of sub-stage s + NS - 1 = (chunk c + 1, U), a piece per quad between the look-ups
    const int64_t cn0 = t0 + (int64_t)cnx * CT;
    const TileSrc tn = tile_src(cn0, U);
    const uint32_t dst_next = Cfg::tile_off(0) + slot_next * Cfg::TILE_B;
    auto piece_now = [&](auto K) {
      if constexpr (!(KVQ_W_ABL & 4)) wide_tile_piece<BITS, decltype(K)::value>(a, dl, f_tile, cn0, tn.row0, tn.nrv, dst_next, wave, tn.fast);
    };
    if constexpr (KVQ_W_BURST || (KVQ_W_ABL & 2)) static_for<0, Cfg::T_OPS>(piece_now);
    auto piece = [&](auto K) {
      if constexpr (!KVQ_W_BURST && !(KVQ_W_ABL & 2)) piece_now(K);
    };
    if constexpr (BITS == 4 && !(KVQ_W_ABL & 2)) {
      const uint32_t slotpat = (uint32_t)sl * 0x40404040u;
      uint4 wq[2];
      float4 pq[2];
      lds_read16<S0>(wq[0], taddr[0][0] + tb);
      lds_read16<P0>(pq[0], paddr_c);
      static_for<0, Cfg::QPL>([&](auto QQ) {
        constexpr int qq = decltype(QQ)::value;
        constexpr int cur = qq & 1;
        uint32_t we, wo, ua[8], ub[8];
        float va[8], vb[8];
        lds_wait<0>();
        hook(QQ, K0);
        nib_prep(we, wo, wq[cur].x, slotpat); nib_extract(ua, we, wo); lut_read8<L0 + (qq * 4 + 0) * TS>(va, ua);
        nib_prep(we, wo, wq[cur].y, slotpat); nib_extract(ub, we, wo); lut_read8<L0 + (qq * 4 + 1) * TS>(vb, ub);
        // a quad's share of the next tile's DMA, behind 16 look-ups in flight
        piece(QQ);
        lds_wait<8>(); fmac8(acc[U], va, pq[cur].x);
        hook(QQ, K2);
        nib_prep(we, wo, wq[cur].z, slotpat); nib_extract(ua, we, wo); lut_read8<L0 + (qq * 4 + 2) * TS>(va, ua);
        lds_wait<8>(); fmac8(acc[U], vb, pq[cur].y);
        hook(QQ, K3);
        nib_prep(we, wo, wq[cur].w, slotpat); nib_extract(ub, we, wo); lut_read8<L0 + (qq * 4 + 3) * TS>(vb, ub);
        if constexpr (qq + 1 < Cfg::QPL) {
          lds_read16<S0>(wq[1 - cur], taddr[qq + 1][0] + tb);
          lds_read16<P0 + (qq + 1) * 16>(pq[1 - cur], paddr_c);
          lds_wait<10>(); fmac8(acc[U], va, pq[cur].z);
          lds_wait<2>(); fmac8(acc[U], vb, pq[cur].w);
        } else {
          lds_wait<8>(); fmac8(acc[U], va, pq[cur].z);
          lds_wait<0>(); fmac8(acc[U], vb, pq[cur].w);
        }
      });
      static_assert(Cfg::T_OPS == Cfg::QPL || BITS != 4, "one tile piece per quad");
    }
    if constexpr (BITS == 3 && !(KVQ_W_ABL & 2)) {
      // slots 2, 3: their codebook rows sit 2 * N * 4 bytes further (the 6-bit look-up fields carry one slot bit)
      {
        constexpr int L1 = L0 + HI * 2 * N * 4;
        const uint32_t slot3 = (uint32_t)(sl & 1) * 0x20820820u;
        uint4 wq[2][3];
        float4 pq[2];
#pragma unroll
        for (int wi = 0; wi < 3; wi++) lds_read16<S0>(wq[0][wi], taddr[0][wi] + tb);
        lds_read16<P0>(pq[0], paddr_c);
        static_for<0, Cfg::QPL>([&](auto QQ) {
          constexpr int qq = decltype(QQ)::value;
          constexpr int cur = qq & 1;
          uint32_t s1, s2, e1, o1, e2, o2, ua[8];
          float va[8], vb[8];
          lds_wait<0>();
          hook(QQ, K0);
          tri_streams(s1, s2, wq[cur][0].x, wq[cur][1].x, wq[cur][2].x, hf);
          tri_prep(e1, o1, s1, slot3); tri_prep(e2, o2, s2, slot3);
          tri_extract_a(ua, e1, o1); lut_read8<L1 + (qq * 4 + 0) * TS>(va, ua);
          tri_extract_b(ua, e1, o1, e2, o2); lut_read8<L1 + (qq * 4 + 0) * TS>(vb, ua);
          if constexpr (qq == 0) {
            piece(std::integral_constant<int, 0>{});
            piece(std::integral_constant<int, 1>{});
          } else {
            piece(std::integral_constant<int, 2>{});
          }
          tri_streams(s1, s2, wq[cur][0].y, wq[cur][1].y, wq[cur][2].y, hf);
          tri_prep(e1, o1, s1, slot3); tri_prep(e2, o2, s2, slot3);
          lds_wait<8>(); fmac8_at<0>(acc[U], va, pq[cur].x);
          hook(QQ, K1);
          tri_extract_a(ua, e1, o1); lut_read8<L1 + (qq * 4 + 1) * TS>(va, ua);
          lds_wait<8>(); fmac8_at<8>(acc[U], vb, pq[cur].x);
          tri_extract_b(ua, e1, o1, e2, o2); lut_read8<L1 + (qq * 4 + 1) * TS>(vb, ua);
          tri_streams(s1, s2, wq[cur][0].z, wq[cur][1].z, wq[cur][2].z, hf);
          tri_prep(e1, o1, s1, slot3); tri_prep(e2, o2, s2, slot3);
          lds_wait<8>(); fmac8_at<0>(acc[U], va, pq[cur].y);
          hook(QQ, K2);
          tri_extract_a(ua, e1, o1); lut_read8<L1 + (qq * 4 + 2) * TS>(va, ua);
          lds_wait<8>(); fmac8_at<8>(acc[U], vb, pq[cur].y);
          tri_extract_b(ua, e1, o1, e2, o2); lut_read8<L1 + (qq * 4 + 2) * TS>(vb, ua);
          tri_streams(s1, s2, wq[cur][0].w, wq[cur][1].w, wq[cur][2].w, hf);
          tri_prep(e1, o1, s1, slot3); tri_prep(e2, o2, s2, slot3);
          lds_wait<8>(); fmac8_at<0>(acc[U], va, pq[cur].z);
          hook(QQ, K3);
          tri_extract_a(ua, e1, o1); lut_read8<L1 + (qq * 4 + 3) * TS>(va, ua);
          lds_wait<8>(); fmac8_at<8>(acc[U], vb, pq[cur].z);
          tri_extract_b(ua, e1, o1, e2, o2); lut_read8<L1 + (qq * 4 + 3) * TS>(vb, ua);
          if constexpr (qq + 1 < Cfg::QPL) {
#pragma unroll
            for (int wi = 0; wi < 3; wi++) lds_read16<S0>(wq[1 - cur][wi], taddr[qq + 1][wi] + tb);
            lds_read16<P0 + (qq + 1) * 16>(pq[1 - cur], paddr_c);
            lds_wait<12>(); fmac8_at<0>(acc[U], va, pq[cur].w);
            lds_wait<4>(); fmac8_at<8>(acc[U], vb, pq[cur].w);
          } else {
            lds_wait<8>(); fmac8_at<0>(acc[U], va, pq[cur].w);
            lds_wait<0>(); fmac8_at<8>(acc[U], vb, pq[cur].w);
          }
        });
      }
    }
    if constexpr (BITS == 2 && !(KVQ_W_ABL & 2)) {
      const uint32_t slot2 = (uint32_t)sl * 0x10101010u;
      uint4 wq[2];
      float4 pq[2];
      lds_read16<S0>(wq[0], taddr[0][0] + tb);
      lds_read16<P0>(pq[0], paddr_c);
      static_for<0, Cfg::QPL>([&](auto QQ) {
        constexpr int qq = decltype(QQ)::value;
        constexpr int cur = qq & 1;
        uint32_t pk[4], ua[8];
        float va[8], vb[8];
        lds_wait<0>();
        hook(QQ, K0);
        duo_prep(pk, wq[cur].x, slot2);
        duo_extract_a(ua, pk); lut_read8<L0 + (qq * 4 + 0) * TS>(va, ua);
        duo_extract_b(ua, pk); lut_read8<L0 + (qq * 4 + 0) * TS>(vb, ua);
        piece(QQ);
        duo_prep(pk, wq[cur].y, slot2);
        lds_wait<8>(); fmac8_at<0>(acc[U], va, pq[cur].x);
        hook(QQ, K1);
        duo_extract_a(ua, pk); lut_read8<L0 + (qq * 4 + 1) * TS>(va, ua);
        lds_wait<8>(); fmac8_at<8>(acc[U], vb, pq[cur].x);
        duo_extract_b(ua, pk); lut_read8<L0 + (qq * 4 + 1) * TS>(vb, ua);
        duo_prep(pk, wq[cur].z, slot2);
        lds_wait<8>(); fmac8_at<0>(acc[U], va, pq[cur].y);
        hook(QQ, K2);
        duo_extract_a(ua, pk); lut_read8<L0 + (qq * 4 + 2) * TS>(va, ua);
        lds_wait<8>(); fmac8_at<8>(acc[U], vb, pq[cur].y);
        duo_extract_b(ua, pk); lut_read8<L0 + (qq * 4 + 2) * TS>(vb, ua);
        duo_prep(pk, wq[cur].w, slot2);
        lds_wait<8>(); fmac8_at<0>(acc[U], va, pq[cur].z);
        hook(QQ, K3);
        duo_extract_a(ua, pk); lut_read8<L0 + (qq * 4 + 3) * TS>(va, ua);
        lds_wait<8>(); fmac8_at<8>(acc[U], vb, pq[cur].z);
        duo_extract_b(ua, pk); lut_read8<L0 + (qq * 4 + 3) * TS>(vb, ua);
        if constexpr (qq + 1 < Cfg::QPL) {
          lds_read16<S0>(wq[1 - cur], taddr[qq + 1][0] + tb);
          lds_read16<P0 + (qq + 1) * 16>(pq[1 - cur], paddr_c);
          lds_wait<10>(); fmac8_at<0>(acc[U], va, pq[cur].w);
          lds_wait<2>(); fmac8_at<8>(acc[U], vb, pq[cur].w);
        } else {
          lds_wait<8>(); fmac8_at<0>(acc[U], va, pq[cur].w);
          lds_wait<0>(); fmac8_at<8>(acc[U], vb, pq[cur].w);
        }
      });
    }
    KVQ_W_STAMP(3);
    slot = (slot + 1) % NS;
  };
  // HI: the second bit of the token slot (3 bit only: the 6-bit look-up fields carry one slot bit, the other one selects a
  // different immediate in every look-up -- slots 2 and 3 run their own copy of the WHOLE loop: no control flow inside it)
  auto run = [&](auto HI_) {
    auto chunk = [&](auto LP_, int c) {
      substage(LP_, std::integral_constant<int, 0>{}, HI_, c);
      if constexpr (NU == 2) substage(LP_, std::integral_constant<int, 1>{}, HI_, c);
      pcur = (pcur + 1) % 3;
    };
    for (int c = 0; c < n_chunks; c += 2) {
      chunk(std::integral_constant<int, 0>{}, c);
      if (c + 1 < n_chunks) chunk(std::integral_constant<int, 1>{}, c + 1);
    }
    // the last chunk's look-ahead has landed (the ring is about to be overwritten by the slot sums); its entry registers stay
    // reserved until here
    vm_wait<0>();
    hold_entries();
  };
  if (BITS == 3 && (sl >> 1)) run(std::integral_constant<int, BITS == 3 ? 1 : 0>{}); else run(std::integral_constant<int, 0>{});

  // ---- sum the token slots through LDS (aliases the tile ring), add the outlier sums, one slab per workgroup
  __syncthreads();
  float *red = reinterpret_cast<float *>(smem + Cfg::tile_off(0));
#pragma unroll
  for (int u = 0; u < NU; u++)
#pragma unroll
    for (int i = 0; i < CHL; i++) red[((u * CHL + i) * Cfg::SLOTS + sl) * Cfg::LPS + lu] = acc[u][i];
  __syncthreads();
  {
    // lane -> 4 consecutive channels of the group
    const int c4 = tid * 4;
    const int gu = c4 / CH, chn = c4 % CH;
    const int u = gu / Cfg::SU, ul_ = gu % Cfg::SU, hf_ = chn / CHL, i0 = chn % CHL;
    const int lu_ = hf_ * Cfg::SU + ul_;
    if (gu < n_units_valid) {
      const long long *sacc = reinterpret_cast<const long long *>(smem + Cfg::ACC_OFF) + c4;
      float o[4];
#pragma unroll
      for (int j = 0; j < 4; j++) {
        float s = 0.f;
#pragma unroll
        for (int k = 0; k < Cfg::SLOTS; k++) s += red[((u * CHL + i0 + j) * Cfg::SLOTS + k) * Cfg::LPS + lu_];
        o[j] = s + (float)((double)sacc[j] * (1.0 / 4294967296.0));
      }
      float *dst = a.partial + (int64_t)range * C + c_lo + c4;      // slab `range`, the group's channel slice
      *reinterpret_cast<float4 *>(dst) = make_float4(o[0], o[1], o[2], o[3]);
    }
  }
#if KVQ_TRACE
  {
    stamp(5);
    unsigned long long tt, rr;
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(tt)::"memory");
    asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(rr)::"memory");
    if ((tid & 63) == 0 && blockIdx.x < 256) {
      unsigned long long *tr = a.trace + ((int64_t)blockIdx.x * 16 + (tid >> 6)) * 16;
      for (int k = 0; k < 6; k++) tr[k] = tr_acc[k];
      tr[11] = n_chunks;
      tr[12] = tt - tr_t0;
      tr[13] = rr - tr_r0;
    }
  }
#endif
}

// ---- plan + launch -------------------------------------------------------------------------------------------------
template <int BITS>
static bool wide_shape_ok(const MixArgs &a) {
  using Cfg = WCfg<BITS>;
  return a.q_len == 1 && (a.idx == nullptr || (a.n_out > 0 && a.n_out * Cfg::CT <= Cfg::E_R * Cfg::NT)) &&
         a.max_len * (int64_t)(a.n_out > 0 ? a.n_out : 1) < (1ll << 30) && a.L * (int64_t)a.H < (1ll << 30);
}

template <int BITS>
int launch_mix_wide(MixArgs a, float *mul, int accumulate, hipStream_t st, const FusedSoftmax *fs, const float *mz,
                    int *n_slabs_out) {
  using Cfg = WCfg<BITS>;
  WideArgs wa;
  a.n_units = a.H * Cfg::UPH;
  a.groups = (a.n_units + Cfg::GU - 1) / Cfg::GU;
  wa.n_chunks_all = (int)((a.L + Cfg::CT - 1) / Cfg::CT);
  int want = 256 / a.groups;
  if (want < 1) want = 1;
  // at least two chunks per range
  int n_ranges = wa.n_chunks_all / 2 < want ? wa.n_chunks_all / 2 : want;
  if (n_ranges < 1) n_ranges = 1;
  wa.n_ranges = n_ranges;
  if (fs) {
    a.scores = fs->scores;
    a.inv = fs->inv;
    a.mz = mz;            // (null: the workgroups merge the partials themselves)
    a.parts = fs->parts;
    a.n_parts = fs->n_parts;
  }
  wa.m = a;
  dim3 grid(n_ranges * a.groups), block(Cfg::NT);
  if (fs) {
    kvq_step_mark_pv(st);
    mix_v_wide_kernel<BITS, true><<<grid, block, 0, st>>>(wa);
  } else {
    mix_v_wide_kernel<BITS, false><<<grid, block, 0, st>>>(wa);
  }
  *n_slabs_out = n_ranges;
  return check_launch();
}

bool mix_wide_ok(int bits, const MixArgs &a) {
  return bits == 4 ? wide_shape_ok<4>(a) : (bits == 3 ? wide_shape_ok<3>(a) : wide_shape_ok<2>(a));
}
int launch_mix_wide_bits(int bits, const MixArgs &a, float *mul, int accumulate, hipStream_t st, const FusedSoftmax *fs,
                         const float *mz, int *n_slabs_out) {
  switch (bits) {
    case 4: return launch_mix_wide<4>(a, mul, accumulate, st, fs, mz, n_slabs_out);
    case 3: return launch_mix_wide<3>(a, mul, accumulate, st, fs, mz, n_slabs_out);
    default: return launch_mix_wide<2>(a, mul, accumulate, st, fs, mz, n_slabs_out);
  }
}

}  // namespace kvq
