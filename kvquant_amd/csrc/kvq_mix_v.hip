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
#include "kvq_mix_v_rows.h"
#include "kvq_mix_lut.h"

#include <hip/hip_fp16.h>

#include <cstdlib>
#ifndef KVQ_V_WGS
#define KVQ_V_WGS 512
#endif
#ifndef KVQ_V_NOSP
#define KVQ_V_NOSP 0   // experiment: LDS without the sparse-phase buffers (compile-only occupancy probes)
#endif
#ifndef KVQ_V_CT16
#define KVQ_V_CT16 0
#endif
#ifndef KVQ_V_WAVES
#define KVQ_V_WAVES 4    // waves per SIMD the register allocation aims at
#endif
#ifndef KVQ_V_MERGE_PARTS
#define KVQ_V_MERGE_PARTS 256   // fused softmax: up to this many score tiles (64K tokens) the p.V workgroups merge the softmax partials themselves (beyond: 6.09 vs 6.01 ms/step at 128K)
#endif
#ifndef KVQ_V_RB
#define KVQ_V_RB 24
#endif
#ifndef KVQ_V_PRIO
#define KVQ_V_PRIO 0    // experiment (measured neutral): wave priority (s_setprio): 1 = high outside the look-up loop (chunk hand-over, sparse phase), low inside;
#endif                  //  2 = the two workgroups of a CU take turns by chunk parity; 3 = both
#ifndef KVQ_TRACE
#define KVQ_TRACE 0     // development: per-phase s_memtime stamps of the chunk loop (tools/dbg/trace_v.py)
#endif
#ifndef KVQ_PAD_LDS
#define KVQ_PAD_LDS 0   // development: extra LDS per workgroup (occupancy experiments)
#endif
#ifndef KVQ_V_DBG
#define KVQ_V_DBG 0   // development ablations (-DKVQ_V_DBG=n): 1 skip the math, 2 skip the DMA, 4 DMA from an L2-resident source
#endif

namespace kvq {

template <int BITS>
struct VCfg {
  static constexpr int N = Fmt<BITS>::kN;
  static constexpr int WORDS = Unit<BITS>::kWords;     // word-rows per unit
  static constexpr int CH = Unit<BITS>::kCh;           // channels per unit
  static constexpr int UPH = kHeadDim / CH;            // units per head
  static constexpr int NT = 512;                       // threads per workgroup
  static constexpr int UW = BITS == 3 ? 128 : 256;     // units per workgroup
  // 3 bit: a unit is 32 channels in 3 word-rows; two lanes ("halves", wave-uniform: lanes [0,UW) / [UW,2UW) of a
  // slot) read the same three rows and decode 16 channels each -- 32 accumulators in one lane do not fit the
  // 128-VGPR budget next to the look-ups in flight, and a spill inside the chunk loop makes hipcc drain the
  // DMAs just issued
  static constexpr int HALVES = BITS == 3 ? 2 : 1;
  static constexpr int CHL = CH / HALVES;              // channels per lane
  static constexpr int SLOTS = NT / (UW * HALVES);     // token slots
  static constexpr int CT = (BITS == 4 && !KVQ_V_CT16) ? 32 : 16;       // tokens per chunk (2 bit: 32 needs ~200 VGPRs as unrolled)
  static constexpr int QR = CT / 4;                    // 16-byte quads per tile row
  static constexpr int SH = CT == 32 ? 1 : 2;          // log2(tile rows per 256 B)
  static constexpr int ROWS = UW * WORDS;              // tile rows
  static constexpr int ROWB = CT * 4;                  // tile row bytes
  static constexpr int TILE_B = ROWS * ROWB;
  static constexpr int HW = UW / UPH;                  // heads per workgroup
  static constexpr int LUT_B = CT * N * 4;
  static constexpr int P_B = HW * CT * 4;
  static constexpr int QPL = QR / SLOTS;               // quads per lane per chunk
  // LDS layout of the two pipeline stages: the small arrays first, so that every look-up address of either stage is
  // an instruction immediate (16 bits) on top of a byte-sized register value: [rows 0][rows 1][p 0][p 1][p 2][tile 0][tile 1]
  // (p 2: the fused-softmax mode converts scores one chunk ahead of the one being read)
  static constexpr int NPB = 3;
  static constexpr int lut_off(int st) { return st * LUT_B; }
  static constexpr int p_off(int st) { return 2 * LUT_B + st * P_B; }
  static constexpr int tile_off(int st) { return 2 * LUT_B + NPB * P_B + st * TILE_B; }
  static constexpr int STAGES_B = 2 * LUT_B + NPB * P_B + 2 * TILE_B;
  static constexpr int RED_B = NT * CHL * 4;           // slot reduction (aliases the stages)
  // sparse phase after the loop (aliases the stages): staged probabilities of the workgroup's token share
  // (37 KB: 288 tokens x 32 heads, odd row stride) + 32 KB of 64-bit accumulators (4096 channels in one pass)
  static constexpr int SP_P_B = 37888;
  static constexpr int SP_B = SP_P_B + 32768;
  static constexpr int SMEM_0 = (STAGES_B > RED_B ? STAGES_B : RED_B);
  static constexpr int SMEM_B = KVQ_V_NOSP ? SMEM_0 : (SMEM_0 > SP_B ? SMEM_0 : SP_B);
  static constexpr int MZ_HEADS = 128;                 // fused softmax: (max, normaliser) of every head, behind everything else
  static constexpr int MZ_B = MZ_HEADS * 8;
  static_assert(P_B / 4 == NT, "the fused softmax converts one score per lane per chunk");
  static_assert(QR % SLOTS == 0, "slots must split the chunk's quads");
};

struct MixArgs {
  const float *p;          // [q_len][H][L]
  const uint32_t *mat;     // [rows][max_len]
  const float *lut_rows;   // [max_len][N]
  const float *outliers;
  const int32_t *idx;
  float *partial;          // [n_ranges][q_len][C]
  int H;
  int q_len;
  int64_t L;
  int64_t max_len;
  int64_t tr;              // tokens per range (multiple of CT)
  int groups;              // unit groups (workgroups per range)
  int n_units;
  int n_out;
  uint32_t n_out_magic;    // ceil(2^32 / n_out)
  // FUSED softmax (decode, q_len = 1): `p` is unused; the kernel reads the RAW scores and converts them to
  // probabilities on the way (exactly the arithmetic of kvq_softmax_finish: half(expf(half(half(s) * inv) - M) / Z))
  const float *scores;     // [H][L]
  const float *mz;         // [H][2]: (max, normaliser) of every row, merged from the partials by softmax_merge_kernel,
                           // or null: few partials (short caches), every workgroup merges them itself (one launch less)
  const float *parts;      // [H][n_parts][2]
  int n_parts;
  const __half *sink;      // [H][n_sink] or null
  __half *sink_probs;
  int n_sink;
  const __half *v_sink;    // [H][n_sink][128] or null: the sink tokens' output goes to sink_out (= mul, which the reduce accumulates onto)
  float *sink_out;
  float inv;
#if KVQ_TRACE
  unsigned long long *trace;   // development: [block][wave][chunk][8]
#endif
};

// Per-lane constants of the chunk DMA.  Every tile DMA instruction of a wave moves 64/QR consecutive
// tile rows; instruction k of a wave is rows 64/QR*8*k further down, which is a wave-uniform pointer
// increment, so ONE 32-bit lane offset serves all of a wave's tile instructions.
struct DmaLane {
  uint32_t tile_row;   // first tile row of this lane (instruction k adds k * rows_per_round)
  uint32_t tile_q4;    // token offset (4*source quad) inside the chunk
  uint32_t lut_tok;    // token (within chunk) whose codebook row this lane fetches
  uint32_t lut_sub;    // float offset inside that row
  uint32_t p_head;     // head (relative to h0) / token (within chunk) of the probability this lane fetches
  uint32_t p_tok;
};

template <int BITS>
__device__ __forceinline__ DmaLane make_dma_lane() {
  using Cfg = VCfg<BITS>;
  const int lane = threadIdx.x & 63;
  const int wave = threadIdx.x >> 6;
  DmaLane d;
  const int s = wave * 64 + lane;              // slot of the wave's first tile instruction
  const int r = s / Cfg::QR;
  const int pos = s % Cfg::QR;
  d.tile_row = r;
  d.tile_q4 = 4 * ((pos - ((r >> Cfg::SH) & (Cfg::QR - 1))) & (Cfg::QR - 1));
  // codebook rows: LDS row index pidx holds token tl with pidx = (qq*4+e)*SLOTS + slot,
  // tl = (slot*QPL+qq)*4+e: the rows the slots decode in one step sit next to each other
  const int pidx = (s * 4) / Cfg::N;
  const int qe = pidx / Cfg::SLOTS, slp = pidx % Cfg::SLOTS;
  d.lut_tok = (slp * Cfg::QPL + qe / 4) * 4 + qe % 4;
  d.lut_sub = (s * 4) % Cfg::N;
  d.p_head = s / Cfg::CT;
  d.p_tok = s % Cfg::CT;
  return d;
}

// probabilities (or, fused softmax: raw scores) of the workgroup's heads for tokens [c0, c0+CT) -> LDS `pbuf`
// ([HW][CT] floats), 4 B per lane
template <int BITS>
__device__ __forceinline__ void issue_p(const float *src, const MixArgs &a, const DmaLane &d, uint32_t pbuf, int64_t c0,
                                        int h0, int b) {
  using Cfg = VCfg<BITS>;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  constexpr int NW = Cfg::NT / 64;
  constexpr int N_P = Cfg::P_B / 256;
  constexpr int K_P = (N_P + NW - 1) / NW;
  const int lim_L = (int)(a.L - c0);           // tokens from the chunk start to the end of the cache
  const float *gbase = src + ((int64_t)b * a.H + h0) * a.L + c0;
  const uint32_t toff = (uint32_t)(lim_L <= 0 ? 0 : ((int)d.p_tok < lim_L ? (int)d.p_tok : lim_L - 1));
  if (lim_L <= 0) gbase = src + ((int64_t)b * a.H + h0) * a.L + a.L - 1;
#pragma unroll
  for (int k = 0; k < K_P; k++) {
    const int j = wave + k * NW;
    if (j < N_P) {
      int hr = d.p_head + k * NW * (64 / Cfg::CT);
      if (h0 + hr >= a.H) hr = a.H - 1 - h0;
      const uint32_t voff = ((uint32_t)hr * (uint32_t)a.L + toff) * 4u;
      dma4(gbase, voff, pbuf + j * 256);
    }
  }
}

// issue the DMA of one chunk (tokens [c0, c0+CT)) into stage `buf`.  (Issued in one burst right after the chunk
// barrier, the 42 pieces of the workgroup's 8 waves queue up in the CU's memory front end and cost every wave ~1300
// cycles per chunk by s_memtime -- but spreading them over the math was measured neutral, 88 vs 88 us: the queueing
// is hidden by the other waves.)
template <int BITS>
__device__ __forceinline__ void issue_chunk(const MixArgs &a, const DmaLane &d, uint32_t lds0, int stage, int64_t c0,
                                            int row_base, int n_rows_valid, int h0, int b, bool with_p = true) {
  using Cfg = VCfg<BITS>;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  constexpr int NW = Cfg::NT / 64;
  constexpr int RPI = 64 / Cfg::QR;                  // tile rows per DMA instruction
  constexpr int N_TILE = Cfg::TILE_B / 1024;
  constexpr int K_TILE = (N_TILE + NW - 1) / NW;     // tile instructions per wave
  constexpr int LUT_SLOTS = Cfg::LUT_B / 16;
  // ---- packed rows: uniform base per instruction, ONE 32-bit lane offset (bytes) for all of them
  // (token clamps in 32-bit arithmetic relative to the chunk: wave-uniform 64-bit limits, 32-bit lane values --
  // 64-bit per-lane compares cost VGPR pairs, and a spill in here makes hipcc drain the DMAs just issued)
  const int lim_len = (int)(a.max_len - c0);   // tokens from the chunk start to the end of the rows (multiple of 4, >= 4)
  {
    const uint32_t *gbase = a.mat + (int64_t)row_base * a.max_len + c0;
    const int tq = (int)d.tile_q4;
    const uint32_t toff = (uint32_t)(tq + 4 > lim_len ? lim_len - 4 : tq);
#pragma unroll
    for (int k = 0; k < K_TILE; k++) {
      const int j = wave + k * NW;
      if (j < N_TILE) {
        int r = d.tile_row + k * NW * RPI;
        if (r >= n_rows_valid) r = n_rows_valid - 1;
        const uint32_t voff = ((uint32_t)r * (uint32_t)a.max_len + toff) * 4u;   // < 2^32 (checked by the host)
        dma16(gbase, voff, lds0 + Cfg::tile_off(stage) + j * 1024);
      }
    }
  }
  // ---- codebook rows of the chunk
  if ((int)threadIdx.x < LUT_SLOTS) {   // wave-granular: LUT_SLOTS is a multiple of 64 or < 64
    const int tr2 = (int)d.lut_tok < lim_len ? (int)d.lut_tok : lim_len - 1;
    const float *gbase = a.lut_rows + c0 * Cfg::N;
    const uint32_t voff = ((uint32_t)tr2 * Cfg::N + d.lut_sub) * 4u;
    dma16(gbase, voff, lds0 + Cfg::lut_off(stage) + wave * 1024);
  }
  if (with_p) issue_p<BITS>(a.p, a, d, lds0 + Cfg::p_off(stage), c0, h0, b);
}


// The common case of issue_chunk -- a chunk that lies entirely inside the rows and the cache, a full unit group --
// needs no clamps: the per-lane part of every source address is a constant of the kernel (three VGPRs), everything
// that changes with the chunk or the piece is in the wave-uniform base.  PART q in [0, QPL): the tile pieces k with
// k % QPL == q, and with q == 0 the codebook rows and the probabilities: the hand-scheduled loop issues one part per
// quad, between its look-ups.  (Issued in one burst after the chunk barrier, the 42 pieces of a workgroup queue up in
// the CU's memory front end -- 64 B/clk -- and every wave sits ~1500 cycles per chunk in the issue, 20 % of the
// kernel by s_memtime once the look-up loop itself is fast.)
struct DmaFast {
  uint32_t tile, lut, p;   // byte offsets
};

template <int BITS>
__device__ __forceinline__ DmaFast make_dma_fast(const MixArgs &a, const DmaLane &d) {
  using Cfg = VCfg<BITS>;
  DmaFast f;
  f.tile = (d.tile_row * (uint32_t)a.max_len + d.tile_q4) * 4u;
  f.lut = (d.lut_tok * Cfg::N + d.lut_sub) * 4u;
  f.p = (d.p_head * (uint32_t)a.L + d.p_tok) * 4u;
  return f;
}

template <int BITS, int PART>
__device__ __forceinline__ void issue_fast(const MixArgs &a, const DmaFast &f, uint32_t lds0, int stage, int pstage,
                                           const float *psrc, int64_t c0, int64_t pc0, int row_base, int h0, int b) {
  using Cfg = VCfg<BITS>;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  constexpr int NW = Cfg::NT / 64;
  constexpr int RPI = 64 / Cfg::QR;
  constexpr int N_TILE = Cfg::TILE_B / 1024;
  constexpr int K_TILE = (N_TILE + NW - 1) / NW;
  constexpr int LUT_SLOTS = Cfg::LUT_B / 16;
  constexpr int N_P = Cfg::P_B / 256;
  constexpr int K_P = (N_P + NW - 1) / NW;
#pragma unroll
  for (int k = 0; k < K_TILE; k++) {
    if (k % Cfg::QPL != PART) continue;
    const int j = wave + k * NW;
    if (j < N_TILE)
      dma16(a.mat + (int64_t)(row_base + k * NW * RPI) * a.max_len + c0, f.tile, lds0 + Cfg::tile_off(stage) + j * 1024);
  }
  if (PART != 0) return;
  if ((int)threadIdx.x < LUT_SLOTS) dma16(a.lut_rows + c0 * Cfg::N, f.lut, lds0 + Cfg::lut_off(stage) + wave * 1024);
#pragma unroll
  for (int k = 0; k < K_P; k++) {
    const int j = wave + k * NW;
    if (j < N_P)
      dma4(psrc + ((int64_t)b * a.H + h0 + k * NW * (64 / Cfg::CT)) * a.L + pc0, f.p, lds0 + Cfg::p_off(pstage) + j * 256);
  }
}

#ifndef KVQ_V_SPREAD
#define KVQ_V_SPREAD 1   // hand-scheduled loop: issue the next chunk's DMA a quad's share at a time (0: one burst after the barrier)
#endif
#ifndef KVQ_V_ASM
#define KVQ_V_ASM 1   // 4 bit: the chunk's look-up loop as hand-scheduled instruction groups (0: what hipcc makes of the C++)
#endif

template <int BITS, int I, int WORDS>
__device__ __forceinline__ unsigned vcode(const uint32_t (&w)[WORDS]) {
  if constexpr (BITS == 3) return code_of<3, I>(w);
  else if constexpr (BITS == 4) return (w[0] >> (4 * I)) & 0xfu;
  else return (w[0] >> (2 * I)) & 0x3u;
}

// 2^(d log2 e): the weights that merge the (max, sum) partials of a softmax row (as kvq_softmax.hip does)
__device__ __forceinline__ float mz_w(float d) { return __builtin_amdgcn_exp2f(d * 1.4426950408889634f); }
// raw score -> fp16-rounded probability, the arithmetic of kvq_softmax_finish (modeling_llama.py:873-874, 1972-1976)
__device__ __forceinline__ float prob_of(float raw, float inv, float M, float rZ) { return prob_fp16(scaled(raw, inv), M, rZ); }

// FUSED: the second softmax pass (kvq_softmax_finish) happens inside this kernel -- the workgroups merge the
// (max, sum) partials of the score kernel themselves and convert raw scores to probabilities in LDS, one chunk
// ahead of the look-up loop; the [H][L] probabilities never exist in memory (saves a launch, 13 us and 2 x 4 bytes
// per token and head of traffic at 128K).
template <int BITS, bool FUSED>
__global__ __launch_bounds__(512, KVQ_V_WAVES) void mix_v_kernel(MixArgs a) {
  using Cfg = VCfg<BITS>;
  constexpr int N = Cfg::N, CH = Cfg::CH, WORDS = Cfg::WORDS, CT = Cfg::CT, CHL = Cfg::CHL;
  __shared__ __attribute__((aligned(16))) unsigned char smem[Cfg::SMEM_B + (FUSED ? Cfg::MZ_B : 0) + KVQ_PAD_LDS];  // static: LDS offsets fold into ds immediates
  unsigned char *stage0 = smem;

#if KVQ_V_PRIO & 1
  __builtin_amdgcn_s_setprio(3);         // (everything outside the look-up loop is a latency chain)
#endif
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
#ifndef KVQ_V_SPFIRST
#define KVQ_V_SPFIRST 1   // 0: always after the loop; 1: odd workgroups first
#endif
  const bool sparse_first = !FUSED && sparse && KVQ_V_SPFIRST && (blockIdx.x & 1);
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
  unsigned tr_prev;
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
  auto sparse_phase = [&]() {
    // LDS (aliases the pipeline stages): [0, SP_P_B) the probabilities of a block of the range's tokens for the group's
    // heads, [SP_P_B, ...) the fixed-point accumulators of the group's channels.  Staging p (coalesced, independent of the
    // entries) removes the second dependent memory round trip (entry -> head -> p gather): the phase is pure latency.
    float *pl = reinterpret_cast<float *>(smem);
    long long *sacc = reinterpret_cast<long long *>(smem + Cfg::SP_P_B);
    static_assert((Cfg::SMEM_B - Cfg::SP_P_B) / 8 >= Cfg::UW * CH, "accumulators of all the group's channels in one pass");
    const int c_lo = u0 * CH;                                  // the group's channels: [c_lo, c_lo + cn)
    const int cn = n_units_valid * CH;
    const int HWv = (n_units_valid + Cfg::UPH - 1) / Cfg::UPH; // its heads: [h0, h0 + HWv)
    constexpr int RB = KVQ_V_RB;    // entries per lane per round, all loads of a round in flight together
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
                pl[hh * nsp + tl] = (FUSED && !(KVQ_V_DBG & 128)) ? prob_of(v[k][m], a.inv, mz[h0 + hh].x, mz[h0 + hh].y) : v[k][m];
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
    r.sacc = reinterpret_cast<long long *>(smem + Cfg::SP_P_B) + ul_ * CH + hf_ * CHL;
    return r;
  };
  if (sparse_first) {
    const Slice sc = slice_of();
    const bool writer = sc.writer;
    float *dst = sc.dst;
    long long *sacc_w = sc.sacc;
    // phase first: its sums wait in this workgroup's slab (global memory; the registers are needed by the loop) and are
    // picked up again by the same lanes at the end
    if (writer) {
#pragma unroll
      for (int i = 0; i < CHL; i++) sacc_w[i] = 0;
    }
    __syncthreads();
    sparse_phase();
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

  // per-lane constant pieces of the LDS addresses
  int rowoff[WORDS], rot[WORDS];
#pragma unroll
  for (int wi = 0; wi < WORDS; wi++) {
    const int r = ul * WORDS + wi;
    rowoff[wi] = r * Cfg::ROWB;
    rot[wi] = (r >> Cfg::SH) & (Cfg::QR - 1);
  }

  // slot pattern OR-ed into the pre-masked nibble bytes (4-bit fast path): byte = slot*64 + code*4
  const uint32_t slotpat = (uint32_t)sl * 0x40404040u;

  // hand-scheduled loops (3 / 4 bit): the lane's LDS addresses inside a stage do not change with the chunk
  constexpr bool ASM_LOOP = KVQ_V_ASM;
  uint32_t taddr[Cfg::QPL][WORDS];   // its 16-byte quads of the tile
  uint32_t paddr = 0;                // the probabilities of its head for its slot's tokens
  if constexpr (ASM_LOOP) {
    if (lds0 != 0) __builtin_trap();   // (the instruction immediates below assume the one static LDS array at 0)
#pragma unroll
    for (int qq = 0; qq < Cfg::QPL; qq++)
#pragma unroll
      for (int wi = 0; wi < WORDS; wi++)
        taddr[qq][wi] = (uint32_t)(rowoff[wi] + (((sl * Cfg::QPL + qq + rot[wi]) & (Cfg::QR - 1)) << 4));
    paddr = (uint32_t)((hl * CT + sl * Cfg::QPL * 4) * 4);
  }

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
    if (!(KVQ_V_DBG & 32)) __syncthreads();   // ... and everybody else's; the other stage is free again
#if KVQ_TRACE
    stamp(2);
#endif
    const int64_t cn0 = (KVQ_V_DBG & 4) ? t0 : c0 + CT;                   // start of the next chunk
    const bool more = ci + 1 < n_chunks && (!(KVQ_V_DBG & 2) || a.q_len > 1000);
    // FUSED: the scores travel one chunk further ahead than the rows (chunk ci+2 -> p buffer (ci+2) % 3)
    const int pcur = FUSED ? ci % 3 : stage;
    const int pnext = FUSED ? (ci + 2) % 3 : 1 - stage;
    const int64_t pc0 = FUSED ? cn0 + CT : cn0;
    const bool spread = ASM_LOOP && KVQ_V_SPREAD && pc0 <= fast_end;   // (wave-uniform)
    if (more && !spread) {
      issue_chunk<BITS>(a, dl, lds0, 1 - stage, cn0, row_base, n_rows_valid, h0, b, !FUSED);
      if (FUSED && ci + 2 < n_chunks) issue_p<BITS>(a.scores, a, dl, lds0 + Cfg::p_off(pnext), pc0, h0, b);
    }
    // the next chunk's scores landed with this chunk's rows (issued one chunk earlier): convert them now -- in the
    // hand-scheduled loop the LDS read is issued here and the value is used two quads later, behind the look-ups
    constexpr bool CONV_IN_LOOP = FUSED && BITS == 4 && KVQ_V_ASM && Cfg::QPL >= 3;
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
    const unsigned char *tile = smem + Cfg::tile_off(stage);
    const unsigned char *lutb = smem + Cfg::lut_off(stage);
    float *pb = reinterpret_cast<float *>(smem + Cfg::p_off(pcur));
    const int rem = (int)(t1 - c0);       // tokens of this range left in the chunk
    if (!FUSED && rem < CT) {             // ragged last chunk: zero the probabilities past the end once
      for (int i = tid; i < Cfg::HW * CT; i += Cfg::NT)
        if (i % CT >= rem) pb[i] = 0.f;
      __syncthreads();
    }
    if (KVQ_V_DBG & 1) return;
#if KVQ_V_PRIO
    {
      const int turn = (KVQ_V_PRIO & 2) ? ((ci + ((int)blockIdx.x >= (int)gridDim.x / 2 ? 1 : 0)) & 1) : 0;
      __builtin_amdgcn_sched_barrier(0);
      if (turn) __builtin_amdgcn_s_setprio(1);
      else __builtin_amdgcn_s_setprio(0);
      __builtin_amdgcn_sched_barrier(0);
    }
#endif
    if constexpr (BITS == 4 && KVQ_V_ASM) {
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
            float pr = (KVQ_V_DBG & 64) ? raw_next : ((cn0 + (tid % CT) < t1) ? prob_of(raw_next, a.inv, myM, myZ) : 0.f);
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
    if constexpr (BITS == 3 && KVQ_V_ASM) {
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
    if constexpr (BITS == 2 && KVQ_V_ASM) {
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
    int rotv[WORDS];
#pragma unroll
    for (int wi = 0; wi < WORDS; wi++) {
      rotv[wi] = rot[wi];
      asm volatile("" : "+v"(rotv[wi]));   // recompute the tile addresses per chunk (else hoisted and spilled)
    }
    static_for<0, Cfg::QPL>([&](auto QQ) {
      constexpr int qq = decltype(QQ)::value;
      const int q = sl * Cfg::QPL + qq;
      uint4 wq[WORDS];
#pragma unroll
      for (int wi = 0; wi < WORDS; wi++) {
        if (KVQ_V_DBG & 16) {
          wq[wi] = make_uint4(rotv[wi] * 0x9E3779B1u, rowoff[wi] * 0x85EBCA6Bu, tid * 0xC2B2AE35u, (tid + q) * 0x27D4EB2Fu);
          asm volatile("" : "+v"(wq[wi].x), "+v"(wq[wi].y), "+v"(wq[wi].z), "+v"(wq[wi].w));
        } else {
          wq[wi] = *reinterpret_cast<const uint4 *>(tile + rowoff[wi] + (((q + rotv[wi]) & (Cfg::QR - 1)) << 4));
        }
      }
      float4 p4;
      if (KVQ_V_DBG & 8) {
        p4 = make_float4(0.5f, 0.25f, 0.125f, 0.0625f);
        asm volatile("" : "+v"(p4.x), "+v"(p4.y), "+v"(p4.z), "+v"(p4.w));
      } else {
        p4 = *reinterpret_cast<const float4 *>(pb + hl * CT + q * 4);
      }
      static_for<0, 4>([&](auto E) {
        constexpr int e = decltype(E)::value;
        uint32_t w[WORDS];
#pragma unroll
        for (int wi = 0; wi < WORDS; wi++)
          w[wi] = e == 0 ? wq[wi].x : (e == 1 ? wq[wi].y : (e == 2 ? wq[wi].z : wq[wi].w));
        const float pt = e == 0 ? p4.x : (e == 1 ? p4.y : (e == 2 ? p4.z : p4.w));
        if constexpr (BITS == 4) {
          // even / odd nibbles as bytes holding code*4 (+ slot*64): one v_bfe_u32 per code gives the
          // complete variable part of the LDS address, everything else is an instruction immediate
          const uint32_t we = ((w[0] << 2) & 0x3C3C3C3Cu) | slotpat;
          const uint32_t wo = ((w[0] >> 2) & 0x3C3C3C3Cu) | slotpat;
          const unsigned char *row = lutb + (qq * 4 + e) * Cfg::SLOTS * N * 4;
          static_for<0, 8>([&](auto I) {
            constexpr int i = decltype(I)::value;
            const uint32_t src = (i & 1) ? wo : we;
            const uint32_t off = (src >> (8 * (i / 2))) & 0xffu;
            acc[i] = fmaf(*reinterpret_cast<const float *>(row + off), pt, acc[i]);
          });
        } else if constexpr (BITS == 2) {
          // same idea at 2 bit: byte b of pre-masked word k holds code*4 (+ slot*16) of channel 4b + k
          const uint32_t sp2 = (uint32_t)sl * 0x10101010u;
          const uint32_t pk[4] = {((w[0] << 2) & 0x0C0C0C0Cu) | sp2, (w[0] & 0x0C0C0C0Cu) | sp2,
                                  ((w[0] >> 2) & 0x0C0C0C0Cu) | sp2, ((w[0] >> 4) & 0x0C0C0C0Cu) | sp2};
          const unsigned char *row = lutb + (qq * 4 + e) * Cfg::SLOTS * N * 4;
          static_for<0, 2>([&](auto HF) {          // 8 look-ups in flight at a time
            static_for<0, 8>([&](auto I) {
              constexpr int i = decltype(HF)::value * 8 + decltype(I)::value;
              const uint32_t off = (pk[i & 3] >> (8 * (i / 4))) & 0xffu;
              acc[i] = fmaf(*reinterpret_cast<const float *>(row + off), pt, acc[i]);
            });
            __builtin_amdgcn_sched_barrier(0);
          });
        } else {
          const float *tab = reinterpret_cast<const float *>(lutb) + ((qq * 4 + e) * Cfg::SLOTS + sl) * N;
          static_for<0, Cfg::HALVES>([&](auto HF) {      // (wave-uniform branch: the code positions are compile-time)
            if (hf == decltype(HF)::value)
              static_for<0, CHL>([&](auto I) {
                constexpr int i = decltype(I)::value;
                acc[i] = fmaf(tab[vcode<BITS, decltype(HF)::value * CHL + i, WORDS>(w)], pt, acc[i]);
              });
          });
        }
        if constexpr (CHL > 8 || (e & 1))
          __builtin_amdgcn_sched_barrier(0);   // 16 look-ups (two 4-bit tokens / one 2-bit token) or one 3-bit token's 32 in
                                               // flight at a time (VGPR budget)
      });
    });
#if KVQ_TRACE
    asm volatile("" :: "v"(acc[0]));
    stamp(4);
#endif
  };
  auto hand_over = [&]() {
#if KVQ_V_PRIO & 1
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_setprio(3);       // chunk hand-over (wait, barrier, DMA issue): latency, not throughput
    __builtin_amdgcn_sched_barrier(0);
#endif
  };
  hand_over();
  for (int ci = 0; ci < n_chunks; ci += 2) {
    chunk(std::integral_constant<int, 0>{}, ci);
    hand_over();
    if (ci + 1 < n_chunks) {
      chunk(std::integral_constant<int, 1>{}, ci + 1);
      hand_over();
    }
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
    if (sparse && !sparse_first) {
#pragma unroll
      for (int i = 0; i < CHL; i++) sacc_w[i] = 0;     // (the writers cover exactly the group's channels)
    }
  }
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
    }
  } else if (sparse) {
    // (the dense sums wait in the slab during the phase, like the outlier sums of the other order during the loop)
    if (writer) {
#pragma unroll
      for (int i = 0; i < CHL; i += 4) *reinterpret_cast<float4 *>(dst + i) = make_float4(o[i], o[i + 1], o[i + 2], o[i + 3]);
    }
    __syncthreads();   // the slot sums are read, the accumulators zeroed: the stage may be overwritten
    sparse_phase();    // (ends with a barrier: the sums are complete)
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
  if (writer && (sparse_first || !sparse)) {
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
  size_t bytes;
};

template <int BITS>
static Plan plan_mix(int q_len, int H, int64_t L) {
  using Cfg = VCfg<BITS>;
  Plan pl;
  pl.n_units = H * Cfg::UPH;
  pl.groups = (pl.n_units + Cfg::UW - 1) / Cfg::UW;
  int64_t want = KVQ_V_WGS / pl.groups;   // KVQ_V_WGS / 256 workgroups of 512 lanes per CU
  if (want < 1) want = 1;
  int64_t tr = (L + want - 1) / want;
  tr = (tr + Cfg::CT - 1) / Cfg::CT * Cfg::CT;
  if (tr < 2 * Cfg::CT) tr = 2 * Cfg::CT;
  pl.tr = tr;
  pl.n_ranges = (int)((L + tr - 1) / tr);
  if (pl.n_ranges < 1) pl.n_ranges = 1;
  pl.bytes = (size_t)pl.n_ranges * q_len * H * kHeadDim * sizeof(float) +
             (size_t)H * 2 * sizeof(float);   // + the (max, normaliser) pairs of the fused softmax
  return pl;
}

constexpr int kMergeInKernelParts = KVQ_V_MERGE_PARTS;   // up to this many score tiles (256 tokens each): the p.V workgroups merge the softmax partials themselves

struct FusedSoftmax {
  const float *scores, *parts;
  int n_parts;
  float inv;
  const __half *sink;
  __half *sink_probs;
  int n_sink;
  const __half *v_sink;
};

template <int BITS>
static int launch_mix(MixArgs a, float *mul, int accumulate, hipStream_t st, const FusedSoftmax *fs = nullptr) {
  using Cfg = VCfg<BITS>;
  Plan pl = plan_mix<BITS>(a.q_len, a.H, a.L);
  a.tr = pl.tr;
  a.groups = pl.groups;
  a.n_units = pl.n_units;
  dim3 grid(pl.n_ranges * pl.groups, 1, a.q_len), block(Cfg::NT);
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
    if (fs->n_parts > kMergeInKernelParts) {
      float *mz = a.partial + (size_t)pl.n_ranges * a.q_len * a.H * kHeadDim;   // (tail of the workspace)
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
  mix_v_reduce_kernel<<<rgrid, 1024, 0, st>>>(a.partial, mul, pl.n_ranges, a.q_len, C, accumulate);
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

static size_t ws_bytes(int bits, int q_len, int H, int64_t L) {
  size_t a = bits == 4 ? plan_mix<4>(q_len, H, L).bytes : (bits == 3 ? plan_mix<3>(q_len, H, L).bytes : plan_mix<2>(q_len, H, L).bytes);
  size_t b = plan_mix_rows(bits, q_len, H, L, true).bytes;
  return a > b ? a : b;
}

}  // namespace kvq

using namespace kvq;

extern "C" {

size_t kvq_mix_v_workspace_bytes(int bits, int q_len, int H, int hd, int64_t L) {
  if (bits < 2 || bits > 4 || q_len <= 0 || H <= 0 || hd != kHeadDim || L < 0) return 0;
  return ws_bytes(bits, q_len, H, L > 0 ? L : 1);
}

static bool mix_fast_shape(const int32_t *mat, const float *lut_rows, int H, int hd, int64_t L, int64_t max_len, int bits) {
  return (max_len % 4 == 0) && (max_len >= 4) && ((int64_t)H * (hd / 32 * bits) * max_len < (1ll << 31)) &&
         ((reinterpret_cast<uintptr_t>(mat) | reinterpret_cast<uintptr_t>(lut_rows)) % 16 == 0);
}

// p: the probabilities, or with `fs` the raw scores (fast shapes only)
static int mix_v_any(int bits, const float *p, const FusedSoftmax *fs, const int32_t *mat, float *mul, const float *lut_rows,
                     int q_len, int H, int hd, int64_t L, int64_t max_len, const float *outliers,
                     const int32_t *outlier_idx, int n_out, int accumulate, void *workspace, size_t workspace_bytes,
                     void *stream) {
  if (!p || !mat || !mul || !lut_rows || q_len <= 0 || H <= 0 || hd != kHeadDim || L < 0 || L > max_len ||
      bits < 2 || bits > 4)
    return KVQ_EINVAL;
  const bool sparse = outlier_idx != nullptr;     // (without `outliers`: compact rows, packed entries in outlier_idx)
  if ((outliers && !outlier_idx) || (sparse && n_out <= 0)) return KVQ_EINVAL;
  hipStream_t st = (hipStream_t)stream;
  if (L == 0) {
    if (!accumulate) {
      if (hipMemsetAsync(mul, 0, (size_t)q_len * H * hd * sizeof(float), st) != hipSuccess) return KVQ_ELAUNCH;
    }
    return KVQ_OK;
  }
  if (!workspace || workspace_bytes < ws_bytes(bits, q_len, H, L)) return KVQ_EWORKSPACE;
  const bool fast = mix_fast_shape(mat, lut_rows, H, hd, L, max_len, bits);
  if (!fast) {
    if (fs || (sparse && !outliers)) return KVQ_EINVAL;      // (the row-per-lane fallback reads the reference format only)
    MixPlan pl = plan_mix_rows(bits, q_len, H, L, sparse);
    MixVArgs a;
    a.p = p;
    a.mat = reinterpret_cast<const uint32_t *>(mat);
    a.lut_rows = lut_rows;
    a.outliers = outliers;
    a.idx = outlier_idx;
    a.partial = reinterpret_cast<float *>(workspace);
    a.H = H;
    a.q_len = q_len;
    a.L = L;
    a.max_len = max_len;
    a.tr = pl.tr;
    a.n_ranges = pl.n_ranges;
    a.ubg = pl.ubg;
    a.n_units = pl.n_units;
    a.n_out = n_out;
    switch (bits) {
      case 4: return launch_mix_rows<4>(a, pl, mul, accumulate, st);
      case 3: return launch_mix_rows<3>(a, pl, mul, accumulate, st);
      default: return launch_mix_rows<2>(a, pl, mul, accumulate, st);
    }
  }
  MixArgs a;
  a.p = p;
  a.mat = reinterpret_cast<const uint32_t *>(mat);
  a.lut_rows = lut_rows;
  a.outliers = outliers;
  a.idx = outlier_idx;
  a.partial = reinterpret_cast<float *>(workspace);
  a.H = H;
  a.q_len = q_len;
  a.L = L;
  a.max_len = max_len;
  a.tr = 0;
  a.groups = 1;
  a.n_units = 0;
  a.n_out = n_out;
  a.n_out_magic = sparse ? (uint32_t)(((1ull << 32) + (uint64_t)n_out - 1) / (uint64_t)n_out) : 0u;
  a.scores = nullptr;
  a.mz = nullptr;
  a.parts = nullptr;
  a.n_parts = 0;
  a.sink = nullptr;
  a.sink_probs = nullptr;
  a.n_sink = 0;
  a.v_sink = nullptr;
  a.sink_out = nullptr;
  a.inv = 0.f;
#if KVQ_TRACE
  a.trace = reinterpret_cast<unsigned long long *>(strtoull(getenv("KVQ_TRACE_PTR") ? getenv("KVQ_TRACE_PTR") : "0", nullptr, 0));
  if (!a.trace) return KVQ_EINVAL;
#endif
  switch (bits) {
    case 4: return launch_mix<4>(a, mul, accumulate, st, fs);
    case 3: return launch_mix<3>(a, mul, accumulate, st, fs);
    default: return launch_mix<2>(a, mul, accumulate, st, fs);
  }
}

int kvq_mix_v(int bits, const float *p, const int32_t *mat, float *mul, const float *lut_rows, int q_len,
              int H, int hd, int64_t L, int64_t max_len, const float *outliers, const int32_t *outlier_idx,
              int n_out, int accumulate, void *workspace, size_t workspace_bytes, void *stream) {
  return mix_v_any(bits, p, nullptr, mat, mul, lut_rows, q_len, H, hd, L, max_len, outliers, outlier_idx, n_out, accumulate,
                   workspace, workspace_bytes, stream);
}

int kvq_mix_v_softmax(int bits, const float *scores, const float *parts, int n_parts, float inv_sqrt_hd,
                      const uint16_t *sink_scores, uint16_t *sink_probs, int n_sink, const uint16_t *v_sink,
                      float *probs, const int32_t *mat, float *mul, const float *lut_rows, int H, int hd, int64_t L,
                      int64_t max_len, const float *outliers, const int32_t *outlier_idx, int n_out,
                      int accumulate, void *workspace, size_t workspace_bytes, void *stream) {
  if (!scores || !parts || n_parts <= 0 || n_sink < 0 || H <= 0 || L <= 0) return KVQ_EINVAL;
  if (n_sink > 0 && (!sink_scores || !sink_probs)) return KVQ_EINVAL;
  if (v_sink != nullptr && (n_sink <= 0 || accumulate)) return KVQ_EINVAL;
  const bool fast = mix_fast_shape(mat, lut_rows, H, hd, L, max_len, bits) && H <= VCfg<4>::MZ_HEADS;
  if (!fast) {
    // shapes the streaming kernel does not take: the two passes separately
    if (!probs) return KVQ_EINVAL;
    int rc = kvq_softmax_finish(scores, sink_scores, parts, n_parts, probs, sink_probs, H, L, n_sink, inv_sqrt_hd, v_sink,
                                mul, stream);
    if (rc) return rc;
    return mix_v_any(bits, probs, nullptr, mat, mul, lut_rows, 1, H, hd, L, max_len, outliers, outlier_idx, n_out,
                     v_sink ? 1 : accumulate, workspace, workspace_bytes, stream);
  }
  FusedSoftmax f;
  f.scores = scores;
  f.parts = parts;
  f.n_parts = n_parts;
  f.inv = inv_sqrt_hd;
  f.sink = reinterpret_cast<const __half *>(sink_scores);
  f.sink_probs = reinterpret_cast<__half *>(sink_probs);
  f.n_sink = n_sink;
  f.v_sink = reinterpret_cast<const __half *>(v_sink);
  return mix_v_any(bits, scores, &f, mat, mul, lut_rows, 1, H, hd, L, max_len, outliers, outlier_idx, n_out, accumulate,
                   workspace, workspace_bytes, stream);
}

}  // extern "C"
