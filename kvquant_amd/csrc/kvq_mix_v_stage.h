// Geometry, arguments and the LDS-DMA staging of the streaming p.V kernel (kvq_mix_v.hip; design notes there).
#pragma once
#include "kvq_common.h"
#include "kvq_host.h"

#include <hip/hip_fp16.h>

#ifndef KVQ_TRACE
#define KVQ_TRACE 0     // development: per-phase s_memtime stamps of the chunk loop (tools/dbg/trace_v.py)
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
  static constexpr int CT = BITS == 4 ? 32 : 16;       // tokens per chunk (2 bit: 32 needs ~200 VGPRs as unrolled)
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
  // (37 KB: 512 tokens x 16 heads / 256 x 32, odd row stride) + 32 KB of 64-bit accumulators (4096 channels in one pass)
  // + 64 dummy accumulators (one per lane of a wave: where the entries of other groups go, branch-free)
  static constexpr int SP_P_B = 37888;
  static constexpr int SP_B = SP_P_B + 32768 + 512;
  static constexpr int SMEM_0 = (STAGES_B > RED_B ? STAGES_B : RED_B);
  static constexpr int SMEM_B = SMEM_0 > SP_B ? SMEM_0 : SP_B;
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
  int split;               // outlier entries split by TOKENS between the unit groups of a range (kvq_mix_v.hip: sparse_phase_all)
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

// 2^(d log2 e): the weights that merge the (max, sum) partials of a softmax row (as kvq_softmax.hip does)
__device__ __forceinline__ float mz_w(float d) { return __builtin_amdgcn_exp2f(d * 1.4426950408889634f); }
// raw score -> fp16-rounded probability, the arithmetic of kvq_softmax_finish (modeling_llama.py:873-874, 1972-1976)
__device__ __forceinline__ float prob_of(float raw, float inv, float M, float rZ) { return prob_fp16(scaled(raw, inv), M, rZ); }

// the fused-softmax inputs of a launch (kvq_mix_v_softmax)
struct FusedSoftmax {
  const float *scores, *parts;
  int n_parts;
  float inv;
  const __half *sink;
  __half *sink_probs;
  int n_sink;
  const __half *v_sink;
};

// kvq_mix_v.hip: plan + launch of the streaming kernel and its slab reduce (kvq_mix_v_api.hip validates and dispatches)
size_t mix_plan_bytes(int bits, int q_len, int H, int64_t L);
int launch_mix_bits(int bits, MixArgs a, float *mul, int accumulate, hipStream_t st, const FusedSoftmax *fs);

}  // namespace kvq
