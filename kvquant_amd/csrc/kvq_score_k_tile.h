// The tile body of the q.K^T kernel (design notes: kvq_score_k.hip), shared with kvq_fused_decode.hip, which runs the
// same K phase in front of its softmax and p.V phases.
#pragma once
#include "kvq_common.h"
#include "kvq_host.h"
#include "kvq_ktab.h"

#ifndef KVQ_ABL
#define KVQ_ABL 0      // ablation builds (tools/abl): timing experiments, results are wrong by construction
#endif
#ifndef KVQ_TRACE
#define KVQ_TRACE 0     // development: per-phase s_memtime stamps of the head loop (tools/dbg/trace_k.py)
#endif
#ifndef KVQ_QL_PAIR
#define KVQ_QL_PAIR 1   // 0: q of the outlier step as two ds_read_b32 (round 5; A/B runs)
#endif
// (the experiment switches of rounds 2-3 -- look-up software pipeline, woven outlier step, wave priorities, outlier
//  entries after the head loop, a third table buffer -- were measured neutral or slower and are gone; DESIGN.md 3 keeps
//  the numbers)
#include <cmath>
#include <cstdlib>

namespace kvq {

struct ScoreKArgs {
  const float *q;          // [q_len][H][128]
  const uint32_t *mat;     // [H][WPH][max_len]
  float *mul;              // [q_len][H][L]
  const unsigned char *tab;  // [q_len][H][TAB_B] pre-multiplied codebook images (workspace)
  const unsigned char *tab_pair;  // 3 bit, PAIR variants: [H][KTabPair3::BUF_B] fp16 pair-sum images (kvq_ktab.h)
  const float *outliers;   // [max_len][n_out] or null
  const int32_t *idx;
  const float *out_t;      // token-contiguous mirror [n_out][max_len] (TRANSPOSED variant) or null
  const int32_t *idx_t;
  int H;
  int hpg;                 // heads per workgroup (full tiles)
  int groups;              // head groups per full tile
  int full_blocks;         // full tiles * groups
  int hpg_tail;            // heads per workgroup of the ragged last tile
  int64_t L;
  int64_t max_len;
  int pos_offset;
  int n_out;
  uint32_t n_out_magic;    // ceil(2^32 / n_out): e / n_out == umulhi(e, magic) for e < 2^32 / n_out
  int accumulate;
  int pair;                // 3 bit + mirror: read the fp16 pair-sum tables (PAIR variants)
  // optional fusion of the first softmax pass (sparse variant, q_len = 1, accumulate = 0): per (head, tile)
  // max and sum of exp of the SCALED scores, [H][sm_nparts][2]
  float *sm_parts;
  float sm_inv;
  int sm_nparts;
  float rope_theta;
#if KVQ_TRACE
  unsigned long long *trace;   // development: [block][wave][head][8] cycle stamps
#endif
};

// BITS consecutive word-rows starting at uniform row `row0`, each read at the lane's byte offset `voff`:
// wave-uniform 64-bit row base in SGPRs + one 32-bit VGPR offset (no per-lane 64-bit address math).
// Issued from inline asm: the loads are NOT on hipcc's vmcnt scoreboard, so nothing may touch the
// destination registers until the explicit vm_wait<0>() at the top of the next head (the two word sets
// ping-pong, there is no register copy).
template <int BITS>
__device__ __forceinline__ void load_words(uint32_t (&w)[BITS], const uint32_t *__restrict__ base,
                                           int64_t max_len, uint32_t voff) {
#pragma unroll
  for (int i = 0; i < BITS; i++) {
    asm volatile("global_load_dword %0, %1, %2 nt" : "=v"(w[i]) : "v"(voff), "s"(base) : "memory");
    base += max_len;     // (one scalar 64-bit add per row instead of a 64-bit multiply: the kernel issues ~75
                         //  SALU instructions per head iteration, most of them address arithmetic)
  }
}

// 4-bit field extraction of the score kernel: nibble n of a packed word -> the variable part of its look-up address,
// code * 8 + role * 128, in 1.25 VALU instructions per code (2 masks + 8 cuts per word; round 3: 4 pre-masks + 8 byte cuts).
// The word is split once into its low nibbles, with the role bit planted at bit 4 of every byte, and its high nibbles,
// with the role bit at bit 0 of every byte; a cut then lifts 8 bits that start 3 below the nibble -- three zeroed
// bits, the nibble, the neighbouring role bit:
//   low nibble of byte m >= 1: v_bfe_u32(lo, 8m - 3, 8);  m = 0: byte 0 shifted left by 3 (SDWA byte select);
//   high nibble of byte m <= 2: v_bfe_u32(hi, 8m + 1, 8);  m = 3: v_alignbit_b32(role, hi, 25).
struct NibSplit {
  uint32_t lo, hi;
};
__device__ __forceinline__ NibSplit nib_split(uint32_t w, uint32_t role_lo, uint32_t role_hi) {
  NibSplit r;
  asm("v_and_or_b32 %0, %1, %2, %3" : "=v"(r.lo) : "v"(w), "s"(0x0f0f0f0fu), "v"(role_lo));
  asm("v_and_or_b32 %0, %1, %2, %3" : "=v"(r.hi) : "v"(w), "s"(0xf0f0f0f0u), "v"(role_hi));
  return r;
}
template <int NIB>
__device__ __forceinline__ uint32_t nib_field(const NibSplit &x, uint32_t role, uint32_t three) {
  constexpr int m = NIB / 2;
  uint32_t r;
  if constexpr (NIB == 0)
    asm("v_lshlrev_b32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_0" : "=v"(r) : "s"(three), "v"(x.lo));
  else if constexpr ((NIB & 1) == 0) asm("v_bfe_u32 %0, %1, %2, 8" : "=v"(r) : "v"(x.lo), "n"(8 * m - 3));
  else if constexpr (NIB == 7) asm("v_alignbit_b32 %0, %1, %2, 25" : "=v"(r) : "v"(role), "v"(x.hi));
  else asm("v_bfe_u32 %0, %1, %2, 8" : "=v"(r) : "v"(x.hi), "n"(8 * m + 1));
  return r;
}

typedef _Float16 h16x2 __attribute__((ext_vector_type(2)));
// (a & ma) | (b & mb) | role: the pair-index register of the 3-bit PAIR variants
__device__ __forceinline__ uint32_t pair_merge(uint32_t a, uint32_t b, uint32_t ma, uint32_t mb, uint32_t rolepat) {
  uint32_t t, r;
  asm("v_and_or_b32 %0, %1, %2, %3" : "=v"(t) : "v"(a), "s"(ma), "v"(rolepat));
  asm("v_and_or_b32 %0, %1, %2, %3" : "=v"(r) : "v"(b), "s"(mb), "v"(t));
  return r;
}

constexpr int kSparseHpg = 32;   // heads per workgroup cap of the sparse variant (LDS budget: 32 KB score tile)

// SPARSE: fused outlier SpMV.  TRANSPOSED (implies SPARSE): the outliers come from the token-contiguous
// mirror [n_out][max_len] that kvquant_amd's own cache keeps next to the reference's [max_len][n_out]
// rows: a lane then owns ITS token's entries (coalesced loads, no segmented scan, no index division).
// COMPACT (implies TRANSPOSED): the mirror holds packed entries -- fp16 residual << 16 | channel, 4 bytes instead of 8 --
// in idx_t (opt-in format of kvquant_amd's own cache); one load per entry.
// LDS geometry of a score workgroup (shared with kvq_fused_decode.hip, which runs the same tile body)
// PAIR: 0 = one table entry per code, 1 = fp16 pair-sum tables (KTabPair3), 2 = fp32 pair-sum tables (KTabPair32)
template <int BITS, bool SPARSE, int NWAVES, bool TRANSPOSED, int PAIR = 0>
struct KGeom {
  static constexpr int T = NWAVES * 32;
  static constexpr int NT = NWAVES * 64;
  static constexpr int TAB_B = PAIR == 2 ? KTabPair32::BUF_B : (PAIR == 1 ? KTabPair3::BUF_B : KTab<BITS>::BUF_B);
  static constexpr int SCS = kSparseHpg;
  static constexpr int SC_B = SPARSE ? T * SCS * 4 : 16;
  static constexpr int PF = SPARSE ? 2 : 3;
  static constexpr int QL_B = SPARSE ? kSparseHpg * kHeadDim * 4 : 0;
  static constexpr int SMEM_B = PF * TAB_B + SC_B + QL_B;
  static constexpr int SC_OFF = PF * TAB_B;            // the [T][SCS] score tile
};

// what the tile body leaves behind for its epilogue: the workgroup's tile and head group, this lane's token
struct KTile {
  int tile_i, h0, nh, ntok, tl, role, b;
  int64_t tile0, t;
  bool valid;
};

// One workgroup's tile: the scores of `nh` heads x T tokens (dense + outlier entries).  SPARSE variants collect them in
// the LDS tile at KGeom::SC_OFF (complete for the lanes' own rows when this returns; a barrier publishes them to the other
// waves); the dense variant writes them out itself.
// (tile_i, h0_i, nh_i): the tile and the head group; score_k_tile below derives them from the block index
// PAIR (3 bit, implies TRANSPOSED): fp16 pair-sum tables (KTabPair3) -- one ds_read_b32 + one v_dot2_f32_f16 per TWO codes,
// half2 (cos, sin) per rotation pair (32 instead of 64 trig registers).
template <int BITS, bool SPARSE, int NWAVES, bool TRANSPOSED, bool COMPACT, int PAIR = 0>
__device__ __forceinline__ KTile score_k_tile_at(const ScoreKArgs &a, unsigned char *smem, int tile_i, int h0_i, int nh_i) {
  static_assert(!TRANSPOSED || SPARSE, "the transposed mirror is a sparse variant");
  static_assert(!COMPACT || TRANSPOSED, "packed entries live in the mirror");
  static_assert(PAIR == 0 || (BITS == 3 && TRANSPOSED), "pair-sum tables: 3 bit, decode (mirror) variants");
  constexpr int N = Fmt<BITS>::kN;
  constexpr int WPH = Fmt<BITS>::kWordsPerHead;
  constexpr int T = NWAVES * 32;
  constexpr int NT = NWAVES * 64;
  constexpr int TAB_B = PAIR == 2 ? KTabPair32::BUF_B : (PAIR == 1 ? KTabPair3::BUF_B : KTab<BITS>::BUF_B);
  constexpr int SCS = kSparseHpg;       // score-tile row stride (token-major; the column is rotated by the
                                        // token so that neither the per-token nor the per-head access conflicts)
  constexpr int SC_B = SPARSE ? T * SCS * 4 : 16;
  constexpr int TAB_DMA = TAB_B / 1024;                        // 16-byte-per-lane DMA instructions per table
  constexpr int TAB_DMA_W = (TAB_DMA + NWAVES - 1) / NWAVES;   // ... per wave
  // look-ahead depth in heads: words and table of head hh+PF-1 are requested at the top of head hh.  The
  // kernel is bound by bytes in flight (Little's law at ~2 us loaded HBM latency), not by issue: the dense
  // variant has the registers and the LDS for two heads of look-ahead, the sparse one for one.
  constexpr int PF = SPARSE ? 2 : 3;
  // JIT (mirror variant): two register sets, but a set is re-loaded for head h+2 while head h is still being decoded
  // -- each pair of word registers right after the batch that consumed it -- and the outlier entry of head h+2 right
  // after the one of head h has been used.  The loads of a head are then in flight for one to two head iterations
  // (what a third register set would buy) and leave the wave spread over the look-ups instead of in one burst behind
  // the barrier.  Memory operations return in order, so at the top of head h+1 `s_waitcnt vmcnt(JIT_OPS)` -- all but
  // the JIT_OPS operations issued during head h -- covers exactly what head h+1 needs: its table (issued at the top
  // of head h), its words and its outlier entry (issued during head h-1).
  constexpr bool JIT = TRANSPOSED;
  constexpr int JIT_OPS = 2 * BITS + (COMPACT ? 1 : 2) - ((KVQ_ABL & 1024) ? BITS : 0);
  // VMEM operations of one look-ahead step that EVERY wave issues (waves with an extra table piece wait
  // for one more than they need to)
  constexpr int STEP_OPS = 2 * BITS + TAB_DMA / NWAVES;

  // static LDS: every table offset below is a compile-time constant that folds into ds immediates
  // q of the group's heads for the sparse phase: 16 KB.  With 4-bit tables it aliases table buffer 1, which
  // is first written (by the DMA for the second head) after the sparse phase; smaller tables leave room.
  constexpr int QL_B = SPARSE ? kSparseHpg * kHeadDim * 4 : 0;
  static_assert(PF * TAB_B + SC_B + QL_B == KGeom<BITS, SPARSE, NWAVES, TRANSPOSED, PAIR>::SMEM_B, "KGeom");
  unsigned char *lutq = smem;                                                    // [PF][TAB_B]
  float *sc = reinterpret_cast<float *>(smem + PF * TAB_B);                      // [T][SCS]
  float *ql = reinterpret_cast<float *>(smem + PF * TAB_B + SC_B);   // [hpg][128]
  const uint32_t lds0 = lds_addr(smem);

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
#if KVQ_TRACE
  // kernel-level timeline of every wave (100 MHz wall clock, comparable across the chip): entry, loop start, loop
  // end, exit, + where it ran (HW_ID, XCC_ID); behind the per-head stamps
  unsigned long long *tl8 = a.trace + (int64_t)1024 * 8 * 32 * 8 + ((int64_t)blockIdx.x * NWAVES + wave) * 8;
  const bool tlw = lane == 0 && blockIdx.x < 1024;
  auto tstamp = [&](int k) {
    unsigned long long rr;
    asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(rr)::"memory");
    if (tlw) tl8[k] = rr;
  };
  tstamp(0);
  if (tlw) {
    tl8[4] = __builtin_amdgcn_s_getreg(63492);   // HW_REG_HW_ID
    tl8[5] = __builtin_amdgcn_s_getreg(63508);   // HW_REG_XCC_ID
  }
#endif
  const int role = lane >> 5;
  const int tl = wave * 32 + (lane & 31);
  // blocks [0, full_blocks): (full tile, head group) pairs; the ragged last tile -- L % T tokens, present on
  // almost every decode step -- is cut into single-head-group blocks of hpg_tail heads so that it cannot
  // become a whole extra round of a grid that otherwise fills the chip exactly once
  const int nh = __builtin_amdgcn_readfirstlane(nh_i);   // heads of this workgroup
  const int64_t tile0 = (int64_t)tile_i * T;
  const int64_t t = tile0 + tl;
  const bool valid = t < a.L;
  const int64_t tc = valid ? t : a.L - 1;
  const int h0 = __builtin_amdgcn_readfirstlane(h0_i);
  const int b = blockIdx.z;
  const float *qb = a.q + (int64_t)b * a.H * kHeadDim;
  const unsigned char *tabb = (PAIR ? a.tab_pair : a.tab) + ((int64_t)b * a.H + h0) * TAB_B;

  // table of head `hh` -> LDS buffer `buf` (linear copy, lane*16 bytes per instruction)
  auto issue_table = [&](int hh, int buf) {
#pragma unroll
    for (int k = 0; k < TAB_DMA_W; k++) {
      const int j = wave + k * NWAVES;
      if (j < TAB_DMA)
        dma16(tabb + (int64_t)hh * TAB_B, (uint32_t)(j * 1024 + lane * 16), lds0 + buf * TAB_B + j * 1024);
    }
  };

  // ---- sparse entries of the tile.  Wave w owns the 32 tokens it also decodes densely: a contiguous run
  // of 32*n_out entries, walked in 64-lane chunks (fully coalesced, all lanes busy), ONE CHUNK PER HEAD
  // ITERATION of the dense loop, fetched one iteration ahead: the sparse work hides in the dense loop's
  // memory waits instead of being a serial, latency-bound prologue in every workgroup at once.
  const bool do_sparse = SPARSE && b == 0 && (TRANSPOSED ? a.idx_t != nullptr : a.outliers != nullptr);   // reference: batch 0 only (KCU:3605)
  const int ntok = (a.L - tile0 < T) ? (int)(a.L - tile0) : T;
  const unsigned nent = do_sparse ? (unsigned)ntok * (unsigned)a.n_out : 0u;   // entries of the tile
  const unsigned wbase = (unsigned)wave * 32u * (unsigned)a.n_out;             // this wave's first entry
  const unsigned wcnt = 32u * (unsigned)a.n_out;                               // ... and how many
  const unsigned wavail = nent > wbase ? nent - wbase : 0u;                      // ... that exist (ragged tile)
  const int nchunks = (do_sparse && !TRANSPOSED) ? (int)(((wavail < wcnt ? wavail : wcnt) + 63) / 64) : 0;
  const bool wact = wave * 32 < ntok;   // this wave has at least one real token (ragged last tile)
  const float *ov = a.outliers + tile0 * a.n_out;
  const int32_t *oi = a.idx + tile0 * a.n_out;
  // chunk j of this wave -> (val, col) registers (clamped index)
  auto sparse_fetch = [&](int j, float &val, int &col) {
    const unsigned e = wbase + (unsigned)j * 64 + lane;
    const unsigned ec = (e < nent ? e : (nent ? nent - 1 : 0)) * 4u;
    asm volatile("global_load_dword %0, %1, %2" : "=v"(val) : "v"(ec), "s"(ov) : "memory");
    asm volatile("global_load_dword %0, %1, %2" : "=v"(col) : "v"(ec), "s"(oi) : "memory");
  };
  // transposed mirror: role r of token t takes entries [first, first + per) of the token's n_out, one entry
  // per head iteration, fetched one iteration ahead (64 consecutive tokens per row: 256 contiguous bytes)
  const int per_t = (a.n_out + 1) >> 1;
  const int first_t = role ? a.n_out - per_t : 0;       // (odd n_out: role 1's first entry is role 0's last)
  const int nsteps = (do_sparse && TRANSPOSED && wact) ? per_t : 0;
  const uint32_t toff_t = (uint32_t)(((int64_t)first_t * a.max_len + tc) * 4);   // host checks the 32-bit range
  auto sparse_fetch_t = [&](int s2, float &val, int &col) {
    const float *bv = a.out_t + (int64_t)s2 * a.max_len;
    const int32_t *bi = a.idx_t + (int64_t)s2 * a.max_len;
#if KVQ_ABL & 512
    asm volatile("v_mov_b32 %0, 1.0\n\tv_and_b32 %1, 0xfff, %2" : "=v"(val), "=v"(col) : "v"(toff_t));
    return;
#endif
    if constexpr (COMPACT) asm volatile("" : "=v"(val));      // (one packed word per entry: defined in place, no load)
    else asm volatile("global_load_dword %0, %1, %2" : "=v"(val) : "v"(toff_t), "s"(bv) : "memory");
    asm volatile("global_load_dword %0, %1, %2" : "=v"(col) : "v"(toff_t), "s"(bi) : "memory");
  };
  // packed entry -> (residual, global channel)
  auto entry_of = [&](float &val, int &col) {
    if constexpr (COMPACT) {
      val = __half2float(__ushort_as_half((unsigned short)((uint32_t)col >> 16)));
      col = col & 0xffff;
    }
  };
  // The look-ahead (val, col) registers are a two-set ring like the word sets below: a set is written by
  // asm loads (outside hipcc's vmcnt scoreboard) after the dense section of one unrolled copy of the head
  // loop and read in the other copy, after that head's explicit vm_wait.  hipcc does not know the loads
  // are in flight, so nothing may copy or spill a set between its load and that wait: the window is kept
  // short (the sparse work of the head and the loop back-edge), and every path through a head defines the
  // other set in place so that the sets are plain loop-carried values without merge copies.  After the wait the
  // values are ordinary: hipcc may spill them across the dense section as it likes.  tools/check_isa.py
  // (run by build() and tests/test_isa_cpu.py) verifies the property on the generated code for every asm
  // load of every variant of this kernel.
  float spv_all[2] = {0.f, 0.f};
  int spc_all[2] = {0, 0};

  // RoPE frequency j lives in lane j of one VGPR (64 lanes = 64 frequencies); theta_of(j) is a wave shuffle
  const float th_reg = rope_freq(a.rope_theta, lane);
  auto theta_of = [&](int j) { return __shfl(th_reg, j); };
  if constexpr (SPARSE) {
    for (int i = tid; i < T * SCS; i += NT) sc[i] = 0.f;
    // q of the group's heads, the two channels of a rotation pair adjacent -- (q[j], q[j + 64]) -- so that an outlier entry
    // reads both with ONE ds_read_b64 (the entries' channels are data: these reads are where the kernel's bank conflicts are)
    for (int i = tid; i < nh * kHeadDim; i += NT) {
      const int c = i & (kHeadDim - 1);
      ql[KVQ_QL_PAIR ? (i - c) + ((c & 63) << 1) + (c >> 6) : i] = qb[h0 * kHeadDim + i];
    }
  }

  // packed words (role r: channel groups r and 2+r) of PF heads rotate through PF register sets
  uint32_t wlo_all[PF][BITS], whi_all[PF][BITS];
  float oldv[PF];   // dense + accumulate: the score's previous value travels with the head's words
  // lane offset inside a head's rows: role r starts BITS rows further down (host checks it fits 32 bits)
  const uint32_t woff = (uint32_t)(((int64_t)role * BITS * a.max_len + tc) * 4);
  const bool acc_dense = !SPARSE && a.accumulate;
  const uint32_t toff = (uint32_t)tc * 4u;
  auto load_old = [&](float &dst, int hh) {
    const float *base = a.mul + ((int64_t)b * a.H + h0 + hh) * a.L;
    asm volatile("global_load_dword %0, %1, %2" : "=v"(dst) : "v"(toff), "s"(base) : "memory");
  };
  const uint32_t *mat_h0 = a.mat + (int64_t)h0 * WPH * a.max_len;   // first row of the group's first head
  const int64_t head_words = (int64_t)WPH * a.max_len;               // words between consecutive heads
  const int64_t hi_words = (int64_t)2 * BITS * a.max_len;            // ... between the lo and hi channel halves
  auto fetch_head = [&](int hh, auto SET) {   // everything head hh needs from memory -> set SET / table SET
    constexpr int set = decltype(SET)::value;
#if KVQ_ABL & 4
    if (hh < 2) issue_table(hh, set);
#else
    issue_table(hh, set);
#endif
    const uint32_t *hb = mat_h0 + (int64_t)hh * head_words;
    load_words<BITS>(wlo_all[set], hb, a.max_len, woff);
    load_words<BITS>(whi_all[set], hb + hi_words, a.max_len, woff);
    if (acc_dense) load_old(oldv[set], hh);
  };
  if constexpr (JIT) {
    // table 0, then the words and the outlier entry of head 0 (set 0) and -- the JIT_OPS operations that may still be
    // in flight at the top of head 0 -- those of head 1 (set 1)
    issue_table(0, 0);
    load_words<BITS>(wlo_all[0], mat_h0, a.max_len, woff);
    load_words<BITS>(whi_all[0], mat_h0 + hi_words, a.max_len, woff);
    sparse_fetch_t(0, spv_all[0], spc_all[0]);
    {
      const uint32_t *h1 = mat_h0 + (nh > 1 ? head_words : 0);     // (a single head: the same rows again, never used)
      load_words<BITS>(wlo_all[1], h1, a.max_len, woff);
      load_words<BITS>(whi_all[1], h1 + hi_words, a.max_len, woff);
      sparse_fetch_t(per_t > 1 ? 1 : 0, spv_all[1], spc_all[1]);
    }
  } else {
    static_for<0, PF - 1>([&](auto U) {
      if (decltype(U)::value < nh) fetch_head(decltype(U)::value, U);
    });
  }

  __syncthreads();   // sc / ql visible

  // One 64-entry chunk of the wave's sparse run.  The entries of a token are sorted by channel, so equal
  // (token, head) keys are contiguous in the flat entry stream: every lane evaluates its entry, a 6-step
  // segmented inclusive scan (wave shuffles) sums each run, and the run's LAST lane adds the sum into the
  // LDS score tile with a plain read-modify-write.  Keys are distinct within a chunk and a wave owns its
  // tokens' tile rows (the dense epilogue of the same wave adds into them too), so program order is
  // enough: no atomics (ds_add_f32 costs ~2.6 cycles per LANE on gfx950, measured).
  const int pos0 = (int)tile0 + a.pos_offset;
  auto sparse_chunk = [&](int j, float val, int col) {
    const unsigned el = (unsigned)j * 64 + lane;         // entry within the wave's run
    const unsigned e = wbase + el;
    const bool in = (el < wcnt) && (e < nent);
    const int hh = (col >> 7) - h0;
    // capped-away slot (zero, modeling_llama.py:745-747) or another head group: contributes nothing
    const bool use = in && (val != 0.f) && ((unsigned)hh < (unsigned)nh);
    const unsigned tle = __umulhi(e, a.n_out_magic);     // token within the tile
    const int ch = col & 127;
    const float ang = theta_of(ch & 63) * (float)(pos0 + (int)tle);
    float sn, c;
    sincos_rev(ang, sn, c);
    const int hq = use ? hh : 0;
#if KVQ_QL_PAIR
    const f32x2 qp = *reinterpret_cast<const f32x2 *>(ql + hq * kHeadDim + ((ch & 63) << 1));
    const float q1 = (ch < 64) ? qp.x : qp.y;
    const float q2 = (ch < 64) ? qp.y : qp.x;
#else
    const float q1 = ql[hq * kHeadDim + ch];
    const float q2 = ql[hq * kHeadDim + (ch ^ 64)];
#endif
    const float sg = (ch < 64) ? sn : -sn;
    float sum = use ? val * fmaf(c, q1, sg * q2) : 0.f;
    // run key = (token, TRUE head): a zeroed or foreign-group entry keeps its own head, it just carries 0
    const int key = in ? (int)(tle * 1024 + (unsigned)((col >> 7) & 1023)) : -1 - lane;
    // The scan below needs equal (token, head) keys to be contiguous, i.e. a token's entries in ascending
    // channel order -- which is how the reference's glue stores them (modeling_llama.py:742, 1171) but not
    // something its kernel (one atomic per entry) depends on.  Any other order: one lane at a time.
    {
      const int kp = __shfl_up(key, 1);
      if (__any(in && lane > 0 && kp >= 0 && kp > key)) {
        for (int i = 0; i < 64; i++)
          if (lane == i && use && sum != 0.f) sc[tle * SCS + ((hh + tle) & (SCS - 1))] += sum;
        return;
      }
    }
#if !(KVQ_ABL & 8)
    // runs are short (42 entries of a token over 32 heads: ~1.8 on average), so the scan stops as soon as
    // no lane has a run-mate d lanes below (keys are sorted inside a token: if nobody matches at distance
    // d, nobody matches further away either)
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
      const int ku = __shfl_up(key, d);
      const bool mate = lane >= d && ku == key;
      if (!__any(mate)) break;
      const float vu = __shfl_up(sum, d);
      if (mate) sum += vu;
    }
#endif
#if KVQ_ABL & 8
    const bool tail = true;
#else
    const int kn = __shfl_down(key, 1);
    const bool tail = (lane == 63) || (kn != key);
#endif
#if KVQ_ABL & 32
    if (in && sum != 0.f && (unsigned)hh < (unsigned)nh) __hip_atomic_fetch_add(&sc[tle * SCS + ((hh + tle) & (SCS - 1))], sum, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
#else
    if (tail && in && sum != 0.f && (unsigned)hh < (unsigned)nh) sc[tle * SCS + ((hh + tle) & (SCS - 1))] += sum;
#endif
  };

  // first look-ahead set: after the barrier above (its latency hides behind the trig below), checked like
  // every other asm load by tools/check_isa.py
  if constexpr (JIT) {
    // (both sets were requested above)
  } else if constexpr (SPARSE) {
    if (nchunks > 0) sparse_fetch(0, spv_all[0], spc_all[0]);
  }

  // RoPE angles of this lane's token for its 32 rotation pairs (KCU:3083, 3122-3123)
  f32x2 cs[PAIR == 1 ? 1 : 32];   // (cos, sin)
  h16x2 csh[PAIR == 1 ? 32 : 1];  // PAIR == 1: the same as half2
  const float posf = (float)((int)tc + a.pos_offset);
  static_for<0, 32>([&](auto I) {
    constexpr int i = decltype(I)::value;
    const float ang = theta_of(role * 32 + i) * posf;
    float sn, c;
    sincos_rev(ang, sn, c);
    if constexpr (PAIR == 1) {
      csh[i] = h16x2{(_Float16)c, (_Float16)sn};
    } else {
      cs[i].x = c;
      cs[i].y = sn;
    }
  });

  // One entry of this lane's token (transposed mirror).  The lane owns its token's row of the score tile;
  // the only other lane that can touch the same (token, head) cell in the same instruction is the other
  // role half's lane of the same token, so the pair is merged into the role-0 lane first.
  // (two halves: sparse_eval is the latency chain -- angle shuffle, sincos, two q look-ups, pair merge -- and touches
  //  nothing another entry depends on; sparse_commit is the read-modify-write of the score tile, which has to stay in
  //  program order.  The tail below evaluates a whole batch before it commits, so that the chains overlap.)
  struct SpEntry {
    float x;
    int hh;
    bool use;
  };
  auto sparse_eval = [&](int s2, float val, int col) {
    entry_of(val, col);
    const bool s_ok = !(role == 1 && (a.n_out & 1) && s2 == 0);
    const int hhE = (col >> 7) - h0;
    const int ch = col & 127;
    bool use = valid && s_ok && (val != 0.f) && ((unsigned)hhE < (unsigned)nh);
    const float ang = theta_of(ch & 63) * posf;
    float sn, c;
    sincos_rev(ang, sn, c);
    const int hq = use ? hhE : 0;
#if KVQ_QL_PAIR
    const f32x2 qp = *reinterpret_cast<const f32x2 *>(ql + hq * kHeadDim + ((ch & 63) << 1));
    const float q1 = (ch < 64) ? qp.x : qp.y;
    const float q2 = (ch < 64) ? qp.y : qp.x;
#else
    const float q1 = ql[hq * kHeadDim + ch];
    const float q2 = ql[hq * kHeadDim + (ch ^ 64)];
#endif
    const float sg = (ch < 64) ? sn : -sn;
    float x = use ? val * fmaf(c, q1, sg * q2) : 0.f;
    const int hk = use ? hhE : (-1 - role);
    const int ho = __shfl_xor(hk, 32);
    const float xo = __shfl_xor(x, 32);
    if (ho == hk) {
      if (role == 0) x += xo;
      else use = false;
    }
    return SpEntry{x, hhE, use};
  };
  auto sparse_commit = [&](const SpEntry &e) {
    if (e.use) sc[tl * SCS + ((e.hh + tl) & (SCS - 1))] += e.x;
  };
  auto sparse_step_t = [&](int s2, float val, int col) { sparse_commit(sparse_eval(s2, val, col)); };

  // per-lane constant part of every look-up address
  const uint32_t role_lo = role ? 0x10101010u : 0u, role_hi = role ? 0x01010100u : 0u, role_u = (uint32_t)role;   // 4 bit: nib_split
  const uint32_t three = __builtin_amdgcn_readfirstlane(3);
  const uint32_t rolebytes = (uint32_t)role * N * 8;    // generic
  const uint32_t role_pa = role ? 0x00100100u : 0u, role_pb = role ? 0x04004000u : 0u, role_8 = role ? 0x100u : 0u;   // PAIR

  auto head = [&](auto BUF, int hh) {
    constexpr int buf = decltype(BUF)::value;          // register set and table buffer of this head
    constexpr int nxt = (buf + PF - 1) % PF;           // ... of head hh+PF-1, free since head hh-1
    const int h = h0 + hh;
    uint32_t (&wlo)[BITS] = wlo_all[buf];
    uint32_t (&whi)[BITS] = whi_all[buf];
#if KVQ_TRACE
    unsigned long long *tr = a.trace + (((int64_t)blockIdx.x * NWAVES + wave) * 32 + hh) * 8;
    const bool trw = lane == 0 && blockIdx.x < 1024 && hh < 32;
    auto stamp = [&](int k) { __builtin_amdgcn_sched_barrier(0); const unsigned long long tt = __builtin_amdgcn_s_memtime(); asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); __builtin_amdgcn_sched_barrier(0); if (trw) tr[k] = tt; };
    stamp(0);
#endif
    // this head's table and words were requested PF-1 heads ago; younger requests may stay in flight
    if constexpr (JIT) {
      vm_wait<JIT_OPS>();   // (what head hh-1 issued for head hh+1 may stay in flight)
      // the set this head consumes exists from HERE on: an empty read-write asm makes the registers opaque at this
      // point, so that no use of them can be scheduled above the wait (a "memory" clobber orders memory operations,
      // not register-only instructions -- cdna_hip_programming.md 5.7 item 3)
#pragma unroll
      for (int i = 0; i < BITS; i++) asm volatile("" : "+v"(wlo_all[buf][i]), "+v"(whi_all[buf][i]));
      asm volatile("" : "+v"(spv_all[buf & 1]), "+v"(spc_all[buf & 1]));
    } else if (PF > 2 && hh + 1 < nh) {
      if (acc_dense) vm_wait<(PF - 2) * (STEP_OPS + 1)>();
      else vm_wait<(PF - 2) * STEP_OPS>();
    } else {
      vm_wait<0>();
    }
#if KVQ_TRACE
    stamp(1);
#endif
#if !(KVQ_ABL & 2)
    __syncthreads();  // ... landed for all waves (and sc complete); table buffer `nxt` is free
#endif
#if KVQ_TRACE
    stamp(2);
#endif
    // JIT: this head re-loads its own register set for head hh+2, row by row.  EVERY head issues the same JIT_OPS
    // operations (no conditional definitions of the in-flight registers, constant wait counts): the last two heads
    // read their own rows once more (never used; just consumed, so the lines are still in the L2)
    const uint32_t *jit_row = mat_h0 + (int64_t)(hh + 2 < nh ? hh + 2 : hh) * head_words;
    if constexpr (JIT) {
      if (hh + 1 < nh) issue_table(hh + 1, nxt);
    } else {
      if (hh + PF - 1 < nh) fetch_head(hh + PF - 1, std::integral_constant<int, nxt>{});
    }
#if KVQ_TRACE
    stamp(3);
#endif
    const unsigned char *tlo = lutq + buf * TAB_B;
    const unsigned char *thi = lutq + buf * TAB_B + KTab<BITS>::HALF_B;
    // 16 look-ups (8 pairs) are issued back to back before their 16 packed FMAs, into 4 independent
    // accumulators: the LDS pipe needs >= 16 reads in flight per wave to run at rate, and a single
    // accumulator would serialise the FMAs
    f32x2 acc4[4] = {{0.f, 0.f}, {0.f, 0.f}, {0.f, 0.f}, {0.f, 0.f}};
    // (JIT: a wave without tokens -- ragged last tile -- decodes its clamped token like the others, so that every wave
    //  issues the same operations per head; its results are dropped below)
    float accp[4] = {0.f, 0.f, 0.f, 0.f};          // PAIR: fp32 accumulators of the v_dot2 chain
    if ((wact || JIT) && !(KVQ_ABL & 64)) {
    if constexpr (PAIR != 0) {
      // The 3-bit fields of the two 96-bit streams (channels 32r.. in wlo, 64+32r.. in whi) are merged pairwise into 6-bit
      // indices c_lo | c_hi << 3, planted at bits 2 + 6k of a register with the role bit above every second one, so that a
      // single cut of 9 bits is the variable part of the look-up address, index * 4 + role * 256 (KTabPair3).  Word w holds
      // fields g = 0..9 of the stream from bit w on (pair i = 11w + g); the two straddling fields, i = 10 and 21, are
      // assembled from two words.  PAIR == 2 (fp32 sums, KTabPair32): the same index, entries of 8 bytes -- the cut shifted left
      // by one is index * 8 + role * 512 --, one ds_read_b64 and one packed FMA against the lane's fp32 (cos, sin).
      const unsigned char *tp = lutq + buf * TAB_B;
      constexpr int PB = PAIR == 2 ? KTabPair32::PAIR_B : KTabPair3::PAIR_B;
      static_for<0, 3>([&](auto WI) {
        constexpr int w = decltype(WI)::value;
        static_for<0, 2>([&](auto PI) {
          constexpr int par = decltype(PI)::value;          // 0: fields g = 0, 2, .., 8;  1: g = 1, 3, .., 9
          constexpr int sa = 2 - w - 3 * par;               // net left shift that puts c_lo of field g = 2k + par at bit 2 + 6k
          constexpr int sb = 5 - w - 3 * par;               // ... and c_hi at bit 5 + 6k
          uint32_t A2;
          if constexpr (sa >= 0) A2 = wlo[w] << sa;
          else A2 = wlo[w] >> (-sa);
          const uint32_t B5 = whi[w] << sb;
          const uint32_t xa = pair_merge(A2, B5, 0x1C01C01Cu, 0xE00E00E0u, role_pa);   // k = 0, 2, 4 (+ role bits 8, 20)
          const uint32_t xb = pair_merge(A2, B5, 0x00700700u, 0x03803800u, role_pb);   // k = 1, 3 (+ role bits 14, 26)
          uint32_t ad[5];
          ad[0] = xa & 0x1ffu;
          asm("v_bfe_u32 %0, %1, 6, 9" : "=v"(ad[1]) : "v"(xb));
          asm("v_bfe_u32 %0, %1, 12, 9" : "=v"(ad[2]) : "v"(xa));
          asm("v_bfe_u32 %0, %1, 18, 9" : "=v"(ad[3]) : "v"(xb));
          asm("v_alignbit_b32 %0, %1, %2, 24" : "=v"(ad[4]) : "v"(role_u), "v"(xa));
          if constexpr (PAIR == 2) {
            f32x2 v[5];
            static_for<0, 5>([&](auto KI) {
              constexpr int k = decltype(KI)::value;
              constexpr int i = 11 * w + 2 * k + par;
              v[k] = *reinterpret_cast<const f32x2 *>(tp + i * 2 * PB + (ad[k] << 1));
            });
            static_for<0, 5>([&](auto KI) {
              constexpr int k = decltype(KI)::value;
              constexpr int i = 11 * w + 2 * k + par;
              acc4[k & 3] = __builtin_elementwise_fma(cs[i], v[k], acc4[k & 3]);
            });
          } else {
            h16x2 v[5];
            static_for<0, 5>([&](auto KI) {
              constexpr int k = decltype(KI)::value;
              constexpr int i = 11 * w + 2 * k + par;
              v[k] = *reinterpret_cast<const h16x2 *>(tp + i * 2 * PB + ad[k]);
            });
            static_for<0, 5>([&](auto KI) {
              constexpr int k = decltype(KI)::value;
              constexpr int i = 11 * w + 2 * k + par;
              accp[k & 3] = __builtin_amdgcn_fdot2(v[k], csh[i], accp[k & 3], false);
            });
          }
        });
      });
      // the straddling fields: i = 10 (bits 30, 31 of word 0 + bit 0 of word 1), i = 21 (bit 31 of word 1 + bits 0, 1 of word 2)
      {
        uint32_t x10, y10, x21, y21;
        asm("v_alignbit_b32 %0, %1, %2, 30" : "=v"(x10) : "v"(wlo[1]), "v"(wlo[0]));
        asm("v_alignbit_b32 %0, %1, %2, 30" : "=v"(y10) : "v"(whi[1]), "v"(whi[0]));
        asm("v_alignbit_b32 %0, %1, %2, 31" : "=v"(x21) : "v"(wlo[2]), "v"(wlo[1]));
        asm("v_alignbit_b32 %0, %1, %2, 31" : "=v"(y21) : "v"(whi[2]), "v"(whi[1]));
        const uint32_t a10 = ((((y10 & 7u) << 3) | (x10 & 7u)) << 2) | role_8;
        const uint32_t a21 = ((((y21 & 7u) << 3) | (x21 & 7u)) << 2) | role_8;
        if constexpr (PAIR == 2) {
          const f32x2 v10 = *reinterpret_cast<const f32x2 *>(tp + 10 * 2 * PB + (a10 << 1));
          const f32x2 v21 = *reinterpret_cast<const f32x2 *>(tp + 21 * 2 * PB + (a21 << 1));
          acc4[1] = __builtin_elementwise_fma(cs[10], v10, acc4[1]);
          acc4[2] = __builtin_elementwise_fma(cs[21], v21, acc4[2]);
        } else {
          const h16x2 v10 = *reinterpret_cast<const h16x2 *>(tp + 10 * 2 * PB + a10);
          const h16x2 v21 = *reinterpret_cast<const h16x2 *>(tp + 21 * 2 * PB + a21);
          accp[1] = __builtin_amdgcn_fdot2(v10, csh[10], accp[1], false);
          accp[2] = __builtin_amdgcn_fdot2(v21, csh[21], accp[2], false);
        }
      }
    } else if constexpr (BITS == 4) {
      static_for<0, 4>([&](auto J) {
        constexpr int j = decltype(J)::value;
        // the words split into low / high nibbles with the role bit next to each (nib_field cuts code*8 + role*128 out)
        const NibSplit xl = nib_split(wlo[j], role_lo, role_hi), xh = nib_split(whi[j], role_lo, role_hi);
        if constexpr (JIT) {
          // rows j of the lo / hi halves are consumed: their registers take the same rows of head hh+2
          asm volatile("" ::"v"(xl.lo), "v"(xl.hi), "v"(xh.lo), "v"(xh.hi));
#if KVQ_ABL & 256
          asm volatile("v_mov_b32 %0, %1" : "=v"(wlo[j]) : "v"(woff));
          asm volatile("v_mov_b32 %0, %1" : "=v"(whi[j]) : "v"(woff));
#elif KVQ_ABL & 1024
          asm volatile("global_load_dword %0, %1, %2 nt" : "=v"(wlo[j]) : "v"(woff), "s"(jit_row + j * a.max_len) : "memory");
          asm volatile("v_mov_b32 %0, %1" : "=v"(whi[j]) : "v"(woff));
#else
          asm volatile("global_load_dword %0, %1, %2 nt" : "=v"(wlo[j]) : "v"(woff), "s"(jit_row + j * a.max_len) : "memory");
          asm volatile("global_load_dword %0, %1, %2 nt" : "=v"(whi[j]) : "v"(woff), "s"(jit_row + hi_words + j * a.max_len) : "memory");
#endif
        }
        constexpr int LKB = 4, NA = 2;   // rotation pairs per look-up batch (2 * LKB ds_read_b64 in flight), packed accumulators
        static_for<0, 8 / LKB>([&](auto HH) {
          constexpr int hf = decltype(HH)::value;   // batches of LKB pairs = 2 * LKB look-ups each
          f32x2 vl[4], vh[4];
          static_for<0, LKB>([&](auto NN) {
            constexpr int n = LKB * hf + decltype(NN)::value;
            constexpr int i = 8 * j + n;
            const uint32_t fl = nib_field<n>(xl, role_u, three);
            const uint32_t fh = nib_field<n>(xh, role_u, three);
#if KVQ_ABL & 1
            vl[n & 3] = f32x2{__uint_as_float(fl), 1.f};
            vh[n & 3] = f32x2{__uint_as_float(fh), 1.f};
#else
            vl[n & 3] = *reinterpret_cast<const f32x2 *>(tlo + i * 2 * N * 8 + fl);
            vh[n & 3] = *reinterpret_cast<const f32x2 *>(thi + i * 2 * N * 8 + fh);
#endif
          });
          static_for<0, LKB>([&](auto NN) {
            constexpr int n = LKB * hf + decltype(NN)::value;
            constexpr int i = 8 * j + n;
            acc4[n & (NA - 1)] = __builtin_elementwise_fma(cs[i], vl[n & 3], acc4[n & (NA - 1)]);
            acc4[(n + NA / 2) & (NA - 1)] = __builtin_elementwise_fma(cs[i], vh[n & 3], acc4[(n + NA / 2) & (NA - 1)]);
          });
          __builtin_amdgcn_sched_group_barrier(0x002, 2 * LKB, 0);    // field extraction (VALU)
          __builtin_amdgcn_sched_group_barrier(0x100, 2 * LKB, 0);    // ds_read_b64
          __builtin_amdgcn_sched_group_barrier(0x002, 2 * LKB, 0);    // v_pk_fma_f32
        });
      });
    } else if constexpr (BITS == 3) {
      // 3 bit, fp32 tables (round 5): 1.3 instead of 2 VALU operations per code for the look-up address code * 8 + role * 64.
      // Word w of a 96-bit stream holds the fields g = 0..9 from bit w on (channel i = 11w + g); taking every THIRD
      // field (class cl = g % 3) leaves six zero bits between two fields, so with the role bit planted right above each
      // field ONE cut of 7 bits that starts 3 below the field is the address part (the 4-bit path's nib_split / nib_field
      // idea; with every second field the role bit of one field would be the lowest bit of the next one's cut).
      // The straddling fields i = 10 and 21 are assembled from two words.
      const uint32_t role_all = role ? 0xffffffffu : 0u;
      static_for<0, 3>([&](auto WI) {
        constexpr int w = decltype(WI)::value;
        static_for<0, 3>([&](auto CI) {
          constexpr int cl = decltype(CI)::value;
          constexpr int NF = cl == 0 ? 4 : 3;                    // fields g = cl, cl + 3, ... < 10
          constexpr int b0 = w + 3 * cl;                         // bit of the class's first field; the others 9 bits apart
          constexpr uint32_t fm = (7u << b0) | (7u << (b0 + 9)) | (7u << (b0 + 18)) | (NF == 4 ? (7u << (b0 + 27)) : 0u);
          constexpr uint32_t rb = (uint32_t)((((1ull << (b0 + 3)) | (1ull << (b0 + 12)) | (1ull << (b0 + 21)) |
                                               (NF == 4 ? (1ull << (b0 + 30)) : 0ull))) & 0xffffffffull);
          const uint32_t rpat = role_all & rb;
          uint32_t xl, xh;
          asm("v_and_or_b32 %0, %1, %2, %3" : "=v"(xl) : "v"(wlo[w]), "s"(fm), "v"(rpat));
          asm("v_and_or_b32 %0, %1, %2, %3" : "=v"(xh) : "v"(whi[w]), "s"(fm), "v"(rpat));
          // (the two streams one after the other: NF look-ups in flight at a time -- both at once spill four registers
          //  of the 8-wave mirror variant, and a spilled look-ahead register there is not merely slow but wrong)
          static_for<0, 2>([&](auto HI) {
            constexpr int hi = decltype(HI)::value;
            const uint32_t x = hi ? xh : xl;
            const unsigned char *tb = hi ? thi : tlo;
            f32x2 v[NF];
            static_for<0, NF>([&](auto KI) {
              constexpr int k = decltype(KI)::value;
              constexpr int i = 11 * w + 3 * k + cl;
              constexpr int b = b0 + 9 * k;
              uint32_t f;
              if constexpr (b < 3) {
                f = (x << (3 - b)) & 0x7fu;
              } else if constexpr (b + 3 >= 32) {
                asm("v_alignbit_b32 %0, %1, %2, %3" : "=v"(f) : "v"(role_u), "v"(x), "n"(b - 3));
              } else {
                asm("v_bfe_u32 %0, %1, %2, 7" : "=v"(f) : "v"(x), "n"(b - 3));
              }
              v[k] = *reinterpret_cast<const f32x2 *>(tb + i * 2 * N * 8 + f);
            });
            static_for<0, NF>([&](auto KI) {
              constexpr int k = decltype(KI)::value;
              constexpr int i = 11 * w + 3 * k + cl;
              acc4[(k + 2 * hi) & 3] = __builtin_elementwise_fma(cs[i], v[k], acc4[(k + 2 * hi) & 3]);
            });
          });
        });
      });
      static_for<0, 2>([&](auto SI) {
        constexpr int i = decltype(SI)::value == 0 ? 10 : 21;
        const uint32_t fl = (code_of<BITS, i>(wlo) << 3) + rolebytes;
        const uint32_t fh = (code_of<BITS, i>(whi) << 3) + rolebytes;
        const f32x2 vl = *reinterpret_cast<const f32x2 *>(tlo + i * 2 * N * 8 + fl);
        const f32x2 vh = *reinterpret_cast<const f32x2 *>(thi + i * 2 * N * 8 + fh);
        acc4[0] = __builtin_elementwise_fma(cs[i], vl, acc4[0]);
        acc4[2] = __builtin_elementwise_fma(cs[i], vh, acc4[2]);
      });
    } else {
      // generic (2 bit) decode: batches of GB pairs; the sparse variants are at the VGPR limit, and a
      // spilled look-ahead register is not merely slow but wrong (see the look-ahead sets above)
      constexpr int GB = SPARSE ? 4 : 8;
      static_for<0, 32 / GB>([&](auto J) {
        constexpr int j = decltype(J)::value;
        f32x2 vl[GB], vh[GB];
        static_for<0, GB>([&](auto NN) {
          constexpr int n = decltype(NN)::value;
          constexpr int i = GB * j + n;
          const uint32_t fl = (code_of<BITS, i>(wlo) << 3) + rolebytes;
          const uint32_t fh = (code_of<BITS, i>(whi) << 3) + rolebytes;
          vl[n] = *reinterpret_cast<const f32x2 *>(tlo + i * 2 * N * 8 + fl);
          vh[n] = *reinterpret_cast<const f32x2 *>(thi + i * 2 * N * 8 + fh);
        });
        static_for<0, GB>([&](auto NN) {
          constexpr int n = decltype(NN)::value;
          constexpr int i = GB * j + n;
          acc4[n & 3] = __builtin_elementwise_fma(cs[i], vl[n], acc4[n & 3]);
          acc4[(n + 2) & 3] = __builtin_elementwise_fma(cs[i], vh[n], acc4[(n + 2) & 3]);
        });
      });
    }
      if constexpr (JIT && BITS != 4) {
        load_words<BITS>(wlo, jit_row, a.max_len, woff);
        load_words<BITS>(whi, jit_row + hi_words, a.max_len, woff);
      }
    }   // wact
    const f32x2 acc = (acc4[0] + acc4[1]) + (acc4[2] + acc4[3]);
    float res = PAIR == 1 ? (accp[0] + accp[1]) + (accp[2] + accp[3]) : acc.x + acc.y;
    res += __shfl_xor(res, 32);
#if KVQ_TRACE
    asm volatile("" :: "v"(res));
    stamp(4);
#endif
    if constexpr (SPARSE) {
      // scores of the tile collect in LDS (dense part here, sparse runs whenever their chunk comes up) and
      // are written out once after the last head
      if (role == 0 && wact) sc[tl * SCS + ((hh + tl) & (SCS - 1))] += res;
      __builtin_amdgcn_sched_barrier(0);   // keep the sparse chunk's temporaries out of the dense section
      static_assert(!SPARSE || PF == 2, "the sparse look-ahead registers are a two-set ring");
      // (when there is nothing left to fetch the set is "defined" by an empty asm instead: both paths then
      // define it in place and hipcc needs no merge copy -- which it would place inside the in-flight window)
      if constexpr (JIT) {
        // entry hh of this lane's token (landed: the wait at the top of this head), then its registers take entry hh+2
        if (hh < nsteps && !(KVQ_ABL & 128)) sparse_step_t(hh, spv_all[buf & 1], spc_all[buf & 1]);
        sparse_fetch_t(hh + 2 < per_t ? hh + 2 : per_t - 1, spv_all[buf & 1], spc_all[buf & 1]);
      } else if (nchunks > 0 && !(KVQ_ABL & 16)) {
        if (hh + 1 < nchunks) sparse_fetch(hh + 1, spv_all[1 - (buf & 1)], spc_all[1 - (buf & 1)]);
        else asm volatile("" : "=v"(spv_all[1 - (buf & 1)]), "=v"(spc_all[1 - (buf & 1)]));
        if (hh < nchunks) sparse_chunk(hh, spv_all[buf & 1], spc_all[buf & 1]);
      }
    } else {
      if (role == 0 && valid) {
        float *dst = a.mul + ((int64_t)b * a.H + h) * a.L + t;
        if (acc_dense) res += oldv[buf];
        __builtin_nontemporal_store(res, dst);
      }
    }
#if KVQ_TRACE
    stamp(5);
#endif
  };
#if KVQ_TRACE
  tstamp(1);
#endif
  if constexpr (JIT) {
    // pairs of heads, then the odd one: no path through the loop skips a head's loads, so the constant wait counts
    // hold on every control-flow path (which is what tools/check_isa.py verifies on the generated code)
    int hb = 0;
    for (; hb + 1 < nh; hb += 2) {
      head(std::integral_constant<int, 0>{}, hb);
      head(std::integral_constant<int, 1>{}, hb + 1);
    }
    // The last two heads' re-loads are never used, but they ARE in flight: their registers must stay allocated
    // until they have landed (a dead asm output is a register hipcc re-uses at once).  Wait + keep-alive on each
    // exit path separately, so that no merge copy of an in-flight register can precede the wait.
    auto drain = [&]() {
      vm_wait<0>();
#pragma unroll
      for (int i = 0; i < BITS; i++)
        asm volatile("" ::"v"(wlo_all[0][i]), "v"(whi_all[0][i]), "v"(wlo_all[1][i]), "v"(whi_all[1][i]));
      asm volatile("" ::"v"(spv_all[0]), "v"(spc_all[0]), "v"(spv_all[1]), "v"(spc_all[1]));
    };
    if (hb < nh) {
      head(std::integral_constant<int, 0>{}, hb);
      drain();
    } else {
      drain();
    }
  } else {
    for (int hb = 0; hb < nh; hb += PF) {
      static_for<0, PF>([&](auto U) {
        if (hb + decltype(U)::value < nh) head(U, hb + decltype(U)::value);
      });
    }
  }
#if KVQ_TRACE
  tstamp(2);
#endif
  if constexpr (SPARSE) {
    // chunks beyond the number of heads of this workgroup (small head groups / wide rows): serial tail
    for (int j = nh; j < nchunks; j++) {
      float v;
      int cidx;
      sparse_fetch(j, v, cidx);
      vm_wait<0>();
      asm volatile("" : "+v"(v), "+v"(cidx));   // (the values exist from here on)
      sparse_chunk(j, v, cidx);
    }
    if constexpr (TRANSPOSED) {
      // entries beyond the number of heads of this workgroup (ragged-tile / small-group blocks): batches of
      // TB, so that the memory latency is paid per batch (the registers of the dense loop are free here;
      // fetch and wait are back to back, nothing can touch the destinations in between)
      // (evaluated as a batch, committed in order -- sparse_eval / sparse_commit; same box, profiles/r04_tail_batch.txt:
      //  4K 27.3 -> 25.2 us, 32K 42.0 -> 40.4; the 4-wave tiles of short caches have the registers for a token's whole
      //  half of 21 entries in one batch, the 8-wave tiles run at the 128-VGPR limit: 7)
      constexpr int TB = NWAVES == 4 ? 21 : 7;
      for (int s0 = nh; s0 < nsteps; s0 += TB) {
        float v[TB];
        int ci2[TB];
#pragma unroll
        for (int k = 0; k < TB; k++) {
          v[k] = 0.f;
          ci2[k] = 0;
          if (s0 + k < nsteps) sparse_fetch_t(s0 + k, v[k], ci2[k]);
        }
        vm_wait<0>();
#pragma unroll
        for (int k = 0; k < TB; k++) asm volatile("" : "+v"(v[k]), "+v"(ci2[k]));   // (the values exist from here on)
        // (entries past the end carry value 0: evaluated like the others, never committed)
        SpEntry ev[TB];
#pragma unroll
        for (int k = 0; k < TB; k++) ev[k] = sparse_eval(s0 + k, v[k], ci2[k]);
#pragma unroll
        for (int k = 0; k < TB; k++) sparse_commit(ev[k]);
      }
    }
  }
#if KVQ_TRACE
  tstamp(6);       // (the outlier tails are done)
#endif
  KTile kt;
  kt.tile_i = tile_i; kt.h0 = h0; kt.nh = nh; kt.ntok = ntok; kt.tl = tl; kt.role = role; kt.b = b;
  kt.tile0 = tile0; kt.t = t; kt.valid = valid;
  return kt;
}

// blocks [0, full_blocks): (full tile, head group) pairs; the ragged last tile -- L % T tokens, present on almost every
// decode step -- is cut into single-head-group blocks of hpg_tail heads so that it cannot become a whole extra round of a
// grid that otherwise fills the chip exactly once
__device__ __forceinline__ void score_k_block_map(const ScoreKArgs &a, int &tile_i, int &h0_i, int &nh_i) {
  if ((int)blockIdx.x < a.full_blocks) {
    tile_i = blockIdx.x / a.groups;
    h0_i = (blockIdx.x % a.groups) * a.hpg;
    nh_i = a.hpg;
  } else {
    tile_i = a.full_blocks / a.groups;
    h0_i = ((int)blockIdx.x - a.full_blocks) * a.hpg_tail;
    nh_i = (a.H - h0_i < a.hpg_tail) ? (a.H - h0_i) : a.hpg_tail;
  }
}

template <int BITS, bool SPARSE, int NWAVES, bool TRANSPOSED, bool COMPACT, int PAIR = 0>
__device__ __forceinline__ KTile score_k_tile(const ScoreKArgs &a, unsigned char *smem) {
  int tile_i, h0_i, nh_i;
  score_k_block_map(a, tile_i, h0_i, nh_i);
  return score_k_tile_at<BITS, SPARSE, NWAVES, TRANSPOSED, COMPACT, PAIR>(a, smem, tile_i, h0_i, nh_i);
}

}  // namespace kvq
