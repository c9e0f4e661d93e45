// softmax(p).V over the packed NUQ value cache on the AFFINE structure of the per-token codebooks, fused with the
// second softmax pass and the fixed-width sparse-outlier SpMV: the decode-path form of kvq_mix_v_softmax.
// Reference semantics: KCU:3211-3433 (+4117-4491), SPMV_ATOMIC_BALANCED KCU:437-470, launchers KCU:3625-3690;
// the codebook rows it reads are lookup_table[t] = lut * sf_t + off_t (modeling_llama.py:1113, kvq_fused_append.hip).
//
// Every row of the per-token table is an affine image of ONE sorted codebook T[0..n):  row_t[c] = T[c]*sf_t + off_t
// (two roundings).  So   sum_t p_t * row_t[code]  =  sum_t (p_t*sf_t) * T[code]  +  sum_t p_t*off_t :
//   * the look-up table does not change with the token: ONE table in LDS for the whole kernel instead of a 64-byte
//     row per token -- no row DMA, no row base in the look-up address;
//   * a constant table can be indexed by TWO codes at once: a byte of a 4-bit row (a 6-bit field of a 3-bit row)
//     addresses a 256- (64-) entry table of float2 = (T'[code0], T'[code1]), T' = T - T[0]: one address operation pair
//     + one ds_read_b64 + two FMAs per TWO codes (the per-token-row form: 7 VALU + 2 LDS per two codes).  64 lanes
//     hitting arbitrary entries would serialise on the LDS banks, so the table is replicated R times with the copies
//     interleaved (entry stride R*8 bytes, lane l reads copy l % R): R = 32 gives every lane of a 32-lane LDS group
//     its own bank pair (3 bit: 16 KB); at 4 bit the budget of two workgroups per CU allows R = 8 (16 KB), 2.2 LDS
//     cycles per group instead of 1 (tools/ubench/pair_table2.hip);
//   * the weight of a token is w = p_t * sf_t, formed where the raw score becomes a probability (one element per
//     lane, a chunk ahead of the look-ups, as in kvq_mix_v.hip's fused mode); sum_t p_t * row_t[0] is a per-head
//     scalar accumulated by the same lanes and added to the head's 128 channels at the end.
// sf_t and row_t[0] come from the stored rows themselves (first and last entry: sf_t = (row[n-1] - row[0]) /
// (T[n-1] - T[0])), so the cache format does not change and any caller's rows work (8 of a row's 64 bytes are used;
// the line is still fetched: algorithmic bytes as before).  The result differs from the per-row form by the
// rounding of the rows (<= 2e-7 relative per term): p.V is held to 1e-3 (north_star), not to bit equality.
// Everything around the look-up loop is kvq_mix_v.hip's design: LDS-DMA row tiles (16-token chunks, three stages = two
// chunks in flight), lane-owns-row-unit, deterministic slab reduce, outlier phase with 32.32 fixed-point LDS adds
// before (odd workgroups) or after the dense loop.
#include "kvq_common.h"
#include "kvq_host.h"

#include <hip/hip_fp16.h>

#include <cstdlib>

namespace kvq {
namespace va {

// A variant of the kernel's geometry: lanes per workgroup, tokens per chunk, tile stages, table copies, workgroups per CU
template <int NT_, int CT_, int NS_, int R_, int WGPC_>
struct Var {
  static constexpr int NT = NT_, CT = CT_, NS = NS_, R = R_, WGPC = WGPC_;
};

template <int BITS, typename V>
struct ACfg {
  static constexpr int N = Fmt<BITS>::kN;
  static constexpr int WORDS = BITS == 3 ? 3 : 1;      // word-rows per unit
  static constexpr int CH = BITS == 4 ? 8 : (BITS == 3 ? 32 : 16);   // channels per unit
  static constexpr int UPH = kHeadDim / CH;            // units per head
  static constexpr int NT = V::NT;
  static constexpr int NW = NT / 64;
  static constexpr int WGPC = V::WGPC;
  static constexpr int HALVES = BITS == 3 ? 2 : 1;     // 3 bit: two lanes share a unit (16 channels each)
  static constexpr int UW = 256 / HALVES;              // units per workgroup
  static constexpr int CHL = CH / HALVES;              // channels per lane
  static constexpr int SLOTS = NT / (UW * HALVES);     // token slots
  static constexpr int CT = V::CT;                     // tokens per chunk
  static constexpr int QR = CT / 4;                    // 16-byte quads per tile row
  static constexpr int SH = CT == 32 ? 1 : 2;          // log2(tile rows per 256 B)
  static constexpr int QPL = QR / SLOTS;               // quads per lane per chunk
  static constexpr int ROWS = UW * WORDS;
  static constexpr int ROWB = CT * 4;
  static constexpr int TILE_B = ROWS * ROWB;
  static constexpr int NS = V::NS;                     // tile stages
  static constexpr int HW = UW / UPH;                  // heads per workgroup
  static constexpr int P_B = HW * CT * 4;              // weights (raw scores before their conversion) of a chunk
  static constexpr int NPB = NS + 1;                   // the scores travel one chunk further ahead than the rows
  static constexpr int E_B = 256;                      // (row[0], row[n-1]) of a chunk's tokens: [CT][2] floats, padded to one DMA piece
  static constexpr int PBITS = BITS == 4 ? 8 : 6;      // index bits of a code pair
  static constexpr int R = V::R;                       // table copies
  static constexpr int ESH = R == 32 ? 8 : (R == 16 ? 7 : 6);   // log2(entry stride)
  static constexpr int PAIR_B = (1 << PBITS) * R * 8;
  static constexpr int e_off(int pb) { return pb * E_B; }
  static constexpr int p_off(int pb) { return NPB * E_B + pb * P_B; }
  static constexpr int PAIR_OFF = NPB * E_B + NPB * P_B;
  static constexpr int TILE_OFF = PAIR_OFF + PAIR_B;
  static constexpr int STAGES_B = TILE_OFF + NS * TILE_B;
  static constexpr int RED_B = NT * CHL * 4;           // slot reduction (aliases the stages)
  static constexpr int SPR_P_B = 32768;                // outlier workgroups: staged probabilities (halfs) ...
  static constexpr int SPR_CH = 4096;                  // ... + 64-bit accumulators of a window of channels
  static constexpr int SP_B = SPR_P_B + SPR_CH * 8;
  static constexpr int SMEM_0 = STAGES_B > RED_B ? STAGES_B : RED_B;
  static constexpr int SMEM_B = SMEM_0 > SP_B ? SMEM_0 : SP_B;
  static constexpr int MZ_HEADS = 128;
  static constexpr int MZ_B = MZ_HEADS * 8;
  static constexpr int OFF_B = 128 * 4;                // per-head offset sums, behind the (max, normaliser) pairs
  static constexpr int TILE_PIECES = TILE_B / 1024;
  static constexpr int K_TILE = (TILE_PIECES + NW - 1) / NW;
  static constexpr int P_PIECES = P_B / 256;
  static constexpr int K_P = (P_PIECES + NW - 1) / NW;
  static constexpr int E_WAVE = P_PIECES % NW;         // the wave that fetches the row ends (the first without a score piece, or 0)
  static constexpr int RB = NT == 512 ? 24 : 48;       // outlier phase: entries per lane and round
  static_assert(SLOTS >= 1 && QR % SLOTS == 0, "slots must split the chunk's quads");
  static_assert(TILE_PIECES % NW == 0, "every wave issues the same number of tile pieces");
  static_assert(HW * CT <= NT && (HW * CT) % 64 == 0, "one score per lane per chunk, whole waves");
  static_assert(SMEM_B + MZ_B + OFF_B <= 163840 / WGPC, "LDS budget of WGPC workgroups per CU");
  static_assert(K_TILE + 2 <= 8, "vm_wait_dyn");
};

struct Args {
  const uint32_t *mat;     // [rows][max_len]
  const float *lut_rows;   // [max_len][N]
  const float *table;      // [N] sorted codebook the rows are affine images of
  const float *outliers;
  const int32_t *idx;
  float *partial;          // [n_ranges][C]
  int H;
  int64_t L;
  int64_t max_len;
  int64_t tr;              // tokens per range (multiple of CT)
  int groups;
  int n_units;
  int n_out;
  uint32_t n_out_magic;
  const float *scores;     // [H][L] raw
  const float *mz;         // [H][2] or null (the workgroups merge `parts` themselves)
  const float *parts;      // [H][n_parts][2]
  int n_parts;
  const __half *sink;
  __half *sink_probs;
  int n_sink;
  const __half *v_sink;
  float *sink_out;
  float inv;
  int n_dense;             // workgroups [0, n_dense): (range, unit group) of the dense stream; [n_dense, gridDim.x): outlier entries
  int64_t str;             // tokens per outlier workgroup
};

__device__ __forceinline__ float mz_w(float d) { return __builtin_amdgcn_exp2f(d * 1.4426950408889634f); }
__device__ __forceinline__ float prob_of(float raw, float inv, float M, float rZ) { return prob_fp16(scaled(raw, inv), M, rZ); }

template <int OFF>
__device__ __forceinline__ void lds_read16(uint4 &w, uint32_t addr) {
  asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(w) : "v"(addr), "n"(OFF) : "memory");
}
template <int OFF>
__device__ __forceinline__ void lds_read16(float4 &w, uint32_t addr) {
  asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(w) : "v"(addr), "n"(OFF) : "memory");
}
template <int N>
__device__ __forceinline__ void lds_wait() {
  asm volatile("s_waitcnt lgkmcnt(%0)" ::"n"(N) : "memory");
}
// wait until at most n (wave-uniform, 0..8) VMEM operations of this wave are outstanding
__device__ __forceinline__ void vm_wait_dyn(int n) {
  switch (n) {
    case 1: vm_wait<1>(); break;
    case 2: vm_wait<2>(); break;
    case 3: vm_wait<3>(); break;
    case 4: vm_wait<4>(); break;
    case 5: vm_wait<5>(); break;
    case 6: vm_wait<6>(); break;
    case 7: vm_wait<7>(); break;
    case 8: vm_wait<8>(); break;
    default: vm_wait<0>(); break;
  }
}

typedef float f32x2 __attribute__((ext_vector_type(2)));

// ---- 4 bit: the four bytes of a packed word (two codes each) -> look-up addresses (byte << SH) | lo
template <int SH>
__device__ __forceinline__ void pair_addr4(uint32_t (&u)[4], uint32_t w, uint32_t lo) {
  if constexpr (SH == 8) {
    // entry stride 256 bytes: the byte lands in byte 1 of the address, the lane's copy offset (< 256) in byte 0 --
    // ONE v_perm_b32 per pair (selector bytes: 0 = lo.byte0, 4 + m = w.byte m, 0x0c = zero)
    asm volatile("v_perm_b32 %0, %4, %5, %6\n\tv_perm_b32 %1, %4, %5, %7\n\tv_perm_b32 %2, %4, %5, %8\n\tv_perm_b32 %3, %4, %5, %9"
                 : "=&v"(u[0]), "=&v"(u[1]), "=&v"(u[2]), "=&v"(u[3])
                 : "v"(w), "v"(lo), "s"(0x0C0C0400u), "s"(0x0C0C0500u), "s"(0x0C0C0600u), "s"(0x0C0C0700u));
    return;
  }
  uint32_t b0, b1, b2, b3;
  asm volatile("v_and_b32 %0, 0xff, %4\n\tv_bfe_u32 %1, %4, 8, 8\n\tv_bfe_u32 %2, %4, 16, 8\n\tv_lshrrev_b32 %3, 24, %4"
               : "=&v"(b0), "=&v"(b1), "=&v"(b2), "=&v"(b3) : "v"(w));
  asm volatile("v_lshl_or_b32 %0, %4, %9, %8\n\tv_lshl_or_b32 %1, %5, %9, %8\n\tv_lshl_or_b32 %2, %6, %9, %8\n\tv_lshl_or_b32 %3, %7, %9, %8"
               : "=&v"(u[0]), "=&v"(u[1]), "=&v"(u[2]), "=&v"(u[3]) : "v"(b0), "v"(b1), "v"(b2), "v"(b3), "v"(lo), "n"(SH));
}
// ---- 3 bit: NF 6-bit fields of a stream word from bit 0 -> addresses (field << 8) | lo
template <int I, int NF>
__device__ __forceinline__ void field_addr_i(uint32_t (&u)[NF], uint32_t s, uint32_t lo) {
  if constexpr (I < NF) {
    uint32_t f;
    if constexpr (I == 0) asm volatile("v_and_b32 %0, 63, %1" : "=v"(f) : "v"(s));
    else asm volatile("v_bfe_u32 %0, %1, %2, 6" : "=v"(f) : "v"(s), "n"(6 * I));
    asm volatile("v_lshl_or_b32 %0, %1, 8, %2" : "=v"(u[I]) : "v"(f), "v"(lo));
    field_addr_i<I + 1, NF>(u, s, lo);
  }
}
template <int NF>
__device__ __forceinline__ void field_addr(uint32_t (&u)[NF], uint32_t s, uint32_t lo) {
  field_addr_i<0, NF>(u, s, lo);
}
template <int OFF>
__device__ __forceinline__ void pair_read4(f32x2 (&v)[4], const uint32_t (&u)[4]) {
  asm volatile("ds_read_b64 %0, %4 offset:%8\n\tds_read_b64 %1, %5 offset:%8\n\tds_read_b64 %2, %6 offset:%8\n\tds_read_b64 %3, %7 offset:%8"
               : "=&v"(v[0]), "=&v"(v[1]), "=&v"(v[2]), "=&v"(v[3]) : "v"(u[0]), "v"(u[1]), "v"(u[2]), "v"(u[3]), "n"(OFF)
               : "memory");
}
// 4 bit: addresses + the four look-ups of a packed word as ONE instruction block (between separate asm statements
// hipcc inserts conservative s_nop hazard slots: it cannot see that no transcendental / op_sel result is consumed)
template <int SH, int OFF>
__device__ __forceinline__ void pair_lookup4(f32x2 (&v)[4], uint32_t w, uint32_t lo) {
  uint32_t t0, t1, t2, t3;
  if constexpr (SH == 8) {
    asm volatile("v_perm_b32 %4, %8, %9, %10\n\tv_perm_b32 %5, %8, %9, %11\n\tv_perm_b32 %6, %8, %9, %12\n\tv_perm_b32 %7, %8, %9, %13\n\t"
                 "ds_read_b64 %0, %4 offset:%14\n\tds_read_b64 %1, %5 offset:%14\n\tds_read_b64 %2, %6 offset:%14\n\tds_read_b64 %3, %7 offset:%14"
                 : "=&v"(v[0]), "=&v"(v[1]), "=&v"(v[2]), "=&v"(v[3]), "=&v"(t0), "=&v"(t1), "=&v"(t2), "=&v"(t3)
                 : "v"(w), "v"(lo), "s"(0x0C0C0400u), "s"(0x0C0C0500u), "s"(0x0C0C0600u), "s"(0x0C0C0700u), "n"(OFF)
                 : "memory");
  } else {
    asm volatile("v_and_b32 %4, 0xff, %8\n\tv_bfe_u32 %5, %8, 8, 8\n\tv_bfe_u32 %6, %8, 16, 8\n\tv_lshrrev_b32 %7, 24, %8\n\t"
                 "v_lshl_or_b32 %4, %4, %10, %9\n\tv_lshl_or_b32 %5, %5, %10, %9\n\tv_lshl_or_b32 %6, %6, %10, %9\n\tv_lshl_or_b32 %7, %7, %10, %9\n\t"
                 "ds_read_b64 %0, %4 offset:%11\n\tds_read_b64 %1, %5 offset:%11\n\tds_read_b64 %2, %6 offset:%11\n\tds_read_b64 %3, %7 offset:%11"
                 : "=&v"(v[0]), "=&v"(v[1]), "=&v"(v[2]), "=&v"(v[3]), "=&v"(t0), "=&v"(t1), "=&v"(t2), "=&v"(t3)
                 : "v"(w), "v"(lo), "n"(SH), "n"(OFF)
                 : "memory");
  }
}
template <int OFF>
__device__ __forceinline__ void pair_read1(f32x2 &v, uint32_t u) {
  asm volatile("ds_read_b64 %0, %1 offset:%2" : "=v"(v) : "v"(u), "n"(OFF) : "memory");
}
// acc[O .. O+8) += v[0..4) (pairs) * p
template <int O, int NA>
__device__ __forceinline__ void fmac8(float (&a)[NA], const f32x2 (&v)[4], float p) {
  asm volatile("v_fmac_f32 %0, %8, %16\n\tv_fmac_f32 %1, %9, %16\n\tv_fmac_f32 %2, %10, %16\n\tv_fmac_f32 %3, %11, %16\n\t"
               "v_fmac_f32 %4, %12, %16\n\tv_fmac_f32 %5, %13, %16\n\tv_fmac_f32 %6, %14, %16\n\tv_fmac_f32 %7, %15, %16"
               : "+v"(a[O]), "+v"(a[O + 1]), "+v"(a[O + 2]), "+v"(a[O + 3]), "+v"(a[O + 4]), "+v"(a[O + 5]), "+v"(a[O + 6]), "+v"(a[O + 7])
               : "v"(v[0].x), "v"(v[0].y), "v"(v[1].x), "v"(v[1].y), "v"(v[2].x), "v"(v[2].y), "v"(v[3].x), "v"(v[3].y), "v"(p));
}
template <int O, int NA>
__device__ __forceinline__ void fmac2(float (&a)[NA], const f32x2 &v, float p) {
  asm volatile("v_fmac_f32 %0, %2, %4\n\tv_fmac_f32 %1, %3, %4" : "+v"(a[O]), "+v"(a[O + 1]) : "v"(v.x), "v"(v.y), "v"(p));
}

// LDS-DMA pieces of the steady-state loop: M0 is set, not saved and restored (nothing else in this kernel keeps a value in it)
__device__ __forceinline__ void dma16_m0(const void *sbase, uint32_t voff, uint32_t lds_dst) {
  asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" ::"v"(voff), "s"(sbase), "s"(lds_dst) : "memory", "m0");
}
__device__ __forceinline__ void dma4_m0(const void *sbase, uint32_t voff, uint32_t lds_dst) {
  asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dword %0, %1" ::"v"(voff), "s"(sbase), "s"(lds_dst) : "memory", "m0");
}

// Per-lane constants of the chunk DMA (as in kvq_mix_v.hip: every tile piece of a wave moves 64/QR consecutive tile
// rows; piece k of a wave is NW*64/QR rows further down -- a wave-uniform increment -- so one lane offset serves all)
struct DmaLane {
  uint32_t tile_row, tile_q4, p_head, p_tok;
};
template <typename Cfg>
__device__ __forceinline__ DmaLane make_dma_lane() {
  const int s = threadIdx.x;
  DmaLane d;
  const int r = s / Cfg::QR, pos = s % Cfg::QR;
  d.tile_row = r;
  d.tile_q4 = 4 * ((pos - ((r >> Cfg::SH) & (Cfg::QR - 1))) & (Cfg::QR - 1));
  d.p_head = s / Cfg::CT;
  d.p_tok = s % Cfg::CT;
  return d;
}

// tile of chunk [c0, c0 + CT) -> stage `stage`; returns nothing, every wave issues K_TILE pieces
template <typename Cfg>
__device__ __forceinline__ void issue_tile(const Args &a, const DmaLane &d, int stage, int64_t c0, int row_base, int n_rows_valid) {
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  constexpr int RPI = 64 / Cfg::QR;
  const int lim_len = (int)(a.max_len - c0);     // tokens from the chunk start to the end of the rows (multiple of 4, >= 4)
  const uint32_t *gbase = a.mat + (int64_t)row_base * a.max_len + c0;
  const int tq = (int)d.tile_q4;
  const uint32_t toff = (uint32_t)(tq + 4 > lim_len ? lim_len - 4 : tq);
  const uint32_t dst = (uint32_t)(Cfg::TILE_OFF + stage * Cfg::TILE_B);
#pragma unroll
  for (int k = 0; k < Cfg::K_TILE; k++) {
    const int j = wave + k * Cfg::NW;
    int r = d.tile_row + k * Cfg::NW * RPI;
    if (r >= n_rows_valid) r = n_rows_valid - 1;
    const uint32_t voff = ((uint32_t)r * (uint32_t)a.max_len + toff) * 4u;   // < 2^32 (checked by the host)
    dma16(gbase, voff, dst + j * 1024);
  }
}
// raw scores of the workgroup's heads and the row ends of chunk [c0, c0 + CT) -> buffers `pb`; returns the number of
// pieces this wave issued (wave-uniform)
template <typename Cfg>
__device__ __forceinline__ int issue_scores(const Args &a, const DmaLane &d, int pb, int64_t c0, int h0) {
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  int n = 0;
  const int lim_L = (int)(a.L - c0);             // > 0
  const float *gbase = a.scores + (int64_t)h0 * a.L + c0;
  const uint32_t toff = (uint32_t)((int)d.p_tok < lim_L ? (int)d.p_tok : lim_L - 1);
#pragma unroll
  for (int k = 0; k < Cfg::K_P; k++) {
    const int j = wave + k * Cfg::NW;
    if (j < Cfg::P_PIECES) {
      int hr = d.p_head + k * Cfg::NW * (64 / Cfg::CT);
      if (h0 + hr >= a.H) hr = a.H - 1 - h0;
      const uint32_t voff = ((uint32_t)hr * (uint32_t)a.L + toff) * 4u;
      dma4(gbase, voff, (uint32_t)(Cfg::p_off(pb) + j * 256));
      n++;
    }
  }
  if (wave == Cfg::E_WAVE) {
    const int lim_len = (int)(a.max_len - c0);
    const int tok = (lane >> 1) & (Cfg::CT - 1);
    const int tc = tok < lim_len ? tok : lim_len - 1;
    const uint32_t voff = (uint32_t)(tc * Cfg::N + (lane & 1) * (Cfg::N - 1)) * 4u;
    dma4(a.lut_rows + c0 * Cfg::N, voff, (uint32_t)Cfg::e_off(pb));
    n++;
  }
  return n;
}

// The common case -- a chunk that lies entirely inside the rows and the cache, a full unit group -- needs no clamps: the
// per-lane part of every source address is a constant of the kernel (three VGPRs), everything that changes with the
// chunk or the piece is in the wave-uniform base.
struct DmaFast {
  uint32_t tile, p, e;
};
template <typename Cfg>
__device__ __forceinline__ DmaFast make_dma_fast(const Args &a, const DmaLane &d) {
  DmaFast f;
  const int lane = threadIdx.x & 63;
  f.tile = (d.tile_row * (uint32_t)a.max_len + d.tile_q4) * 4u;
  f.p = (d.p_head * (uint32_t)a.L + d.p_tok) * 4u;
  f.e = (uint32_t)(((lane >> 1) & (Cfg::CT - 1)) * Cfg::N + (lane & 1) * (Cfg::N - 1)) * 4u;
  return f;
}
template <typename Cfg>
__device__ __forceinline__ void issue_tile_fast(const Args &a, const DmaFast &f, int stage, int64_t c0, int row_base) {
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  constexpr int RPI = 64 / Cfg::QR;
  const uint32_t dst = (uint32_t)(Cfg::TILE_OFF + stage * Cfg::TILE_B);
#pragma unroll
  for (int k = 0; k < Cfg::K_TILE; k++) {
    const int j = wave + k * Cfg::NW;
    dma16(a.mat + (int64_t)(row_base + k * Cfg::NW * RPI) * a.max_len + c0, f.tile, dst + j * 1024);
  }
}
template <typename Cfg>
__device__ __forceinline__ int issue_scores_fast(const Args &a, const DmaFast &f, int pb, int64_t c0, int h0) {
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  int n = 0;
#pragma unroll
  for (int k = 0; k < Cfg::K_P; k++) {
    const int j = wave + k * Cfg::NW;
    if (j < Cfg::P_PIECES) {
      dma4(a.scores + (int64_t)(h0 + k * Cfg::NW * (64 / Cfg::CT)) * a.L + c0, f.p, (uint32_t)(Cfg::p_off(pb) + j * 256));
      n++;
    }
  }
  if (wave == Cfg::E_WAVE) {
    dma4(a.lut_rows + c0 * Cfg::N, f.e, (uint32_t)Cfg::e_off(pb));
    n++;
  }
  return n;
}
// component E of a quad
template <int E>
__device__ __forceinline__ uint32_t comp(const uint4 &v) { return E == 0 ? v.x : (E == 1 ? v.y : (E == 2 ? v.z : v.w)); }
template <int E>
__device__ __forceinline__ float comp(const float4 &v) { return E == 0 ? v.x : (E == 1 ? v.y : (E == 2 ? v.z : v.w)); }

// (max, 1 / normaliser) of heads [hfirst, hfirst + hcount) -> mzw[h]: from the merged pairs of softmax_merge_kernel, or
// merged here from the score kernel's per-tile partials (+ the fp16 sink scores), the same way in every workgroup
template <typename Cfg>
__device__ __forceinline__ void load_mz(const Args &a, float2 *mzw, int hfirst, int hcount) {
  const int tid = threadIdx.x;
  if (a.mz != nullptr) {
    for (int h = tid; h < a.H; h += Cfg::NT) {
      const float2 t = reinterpret_cast<const float2 *>(a.mz)[h];
      mzw[h] = make_float2(t.x, 1.0f / t.y);
    }
    return;
  }
  if (hfirst + hcount > a.H) hcount = a.H - hfirst;
  constexpr int LPH = 32;
  const int sub = tid & (LPH - 1);
  constexpr int MB = 16;
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

// ---- outlier entries: a workgroup of the launch's trailing blocks takes `str` tokens, ALL heads and channels.  The
// (fp16-valued) probabilities of its tokens are staged in LDS as halfs, every entry adds val * p to a 32.32 fixed-point
// accumulator of its channel (ds_add_u64: exact, order independent), the sums go out as one more slab of the reduce.
// These workgroups start in the slots the first dense workgroups leave behind: the phase is latency-bound (one memory
// round trip for the entries), and inside the dense workgroups it cost its full duration (kvq_mix_v.hip runs it
// there; measured here: 21 us of a 93 us launch at 128K).
template <typename Cfg>
__device__ __forceinline__ void outlier_role(const Args &a, unsigned char *smem, int sr, float *slab) {
  constexpr int RB = Cfg::RB;
  const int tid = threadIdx.x;
  const int C = a.H * kHeadDim;
  float2 *mzw = reinterpret_cast<float2 *>(smem + Cfg::SMEM_B);
  const int64_t t0 = (int64_t)sr * a.str;
  const int64_t t1 = (t0 + a.str < a.L) ? (t0 + a.str) : a.L;
  __half *pl = reinterpret_cast<__half *>(smem);                                 // [H][nsp]
  long long *sacc = reinterpret_cast<long long *>(smem + Cfg::SPR_P_B);          // [SPR_CH]: a window of channels
  const bool compact = a.outliers == nullptr;
  int sb = a.n_out > 0 ? (RB * Cfg::NT) / a.n_out : 0;
  if (sb > Cfg::SPR_P_B / (2 * a.H) - 2) sb = Cfg::SPR_P_B / (2 * a.H) - 2;
  // the first round of entries travels while the row statistics are merged
  load_mz<Cfg>(a, mzw, 0, a.H);
  for (int c_lo = 0; c_lo < C; c_lo += Cfg::SPR_CH) {       // (one window at the 7B shape)
  const int cn = C - c_lo < Cfg::SPR_CH ? C - c_lo : Cfg::SPR_CH;
  for (int i = tid; i < cn; i += Cfg::NT) sacc[i] = 0;
  __syncthreads();
  for (int64_t b0 = t0; b0 < t1 && sb >= 1; b0 += sb) {
    const int ns = (t1 - b0 < sb) ? (int)(t1 - b0) : sb;
    const unsigned nent = (unsigned)ns * (unsigned)a.n_out;
    const float *ov = compact ? reinterpret_cast<const float *>(a.idx + b0 * a.n_out) : a.outliers + b0 * a.n_out;
    const int32_t *oi = a.idx + b0 * a.n_out;
    const float *p0 = a.scores + b0;
    int row[RB];
    float val[RB];
#pragma unroll
    for (int j = 0; j < RB; j++) {
      const unsigned e = j * Cfg::NT + tid;
      const unsigned ec = e < nent ? e : nent - 1;
      const uint32_t w = (uint32_t)oi[ec];
      const float fv = ov[ec];
      row[j] = compact ? (int)(w & 0xffffu) : (int)w;
      val[j] = compact ? __half2float(__ushort_as_half((unsigned short)(w >> 16))) : fv;
    }
    const int nsp = ns | 1;
    {
      const int wv = tid >> 6, ln = tid & 63;
      for (int hb = 0; hb < a.H; hb += 4 * Cfg::NW) {
        float v[4][5];
#pragma unroll
        for (int k = 0; k < 4; k++)
#pragma unroll
          for (int m = 0; m < 5; m++) {
            const int hh = hb + wv + Cfg::NW * k, tl = ln + 64 * m;
            v[k][m] = (hh < a.H && tl < ns) ? p0[(int64_t)hh * a.L + tl] : 0.f;
          }
#pragma unroll
        for (int k = 0; k < 4; k++)
#pragma unroll
          for (int m = 0; m < 5; m++) {
            const int hh = hb + wv + Cfg::NW * k, tl = ln + 64 * m;
            if (hh < a.H && tl < ns) pl[hh * nsp + tl] = __float2half_rn(prob_of(v[k][m], a.inv, mzw[hh].x, mzw[hh].y));
          }
      }
    }
    __syncthreads();
#pragma unroll
    for (int j = 0; j < RB; j++) {
      const unsigned e = j * Cfg::NT + tid;
      const unsigned ec = e < nent ? e : nent - 1;
      const unsigned tl = __umulhi(ec, a.n_out_magic);
      const unsigned rel = (unsigned)(row[j] - c_lo);
      const bool mine = e < nent && rel < (unsigned)cn;
      const unsigned hh = mine ? ((unsigned)row[j] >> 7) : 0u;
      const float pt = __half2float(pl[hh * nsp + tl]);
      if (mine) {
        const float x = val[j] * pt;
        const float fl = floorf(x);
        const unsigned lo = (unsigned)((x - fl) * 4294967296.0f);
        const int hi = (int)fl;
        const unsigned long long fx = ((unsigned long long)(unsigned)hi << 32) | lo;
        atomicAdd(reinterpret_cast<unsigned long long *>(&sacc[rel]), fx);
      }
    }
    __syncthreads();
  }
  for (int i = tid; i < cn; i += Cfg::NT) slab[c_lo + i] = (float)((double)sacc[i] * (1.0 / 4294967296.0));
  __syncthreads();
  }
}

template <int BITS, typename V>
__global__ __launch_bounds__(V::NT, V::NT / 64 * V::WGPC / 4) void mix_va_kernel(Args a) {
  using Cfg = ACfg<BITS, V>;
  constexpr int CH = Cfg::CH, WORDS = Cfg::WORDS, CT = Cfg::CT, CHL = Cfg::CHL, NS = Cfg::NS, NPB = Cfg::NPB;
  __shared__ __attribute__((aligned(16))) unsigned char smem[Cfg::SMEM_B + Cfg::MZ_B + Cfg::OFF_B];
  const int tid = threadIdx.x;
  const int ul = tid % Cfg::UW;
  const int hf = __builtin_amdgcn_readfirstlane((tid / Cfg::UW) % Cfg::HALVES);
  const int lu = tid % (Cfg::UW * Cfg::HALVES);
  const int sl = tid / (Cfg::UW * Cfg::HALVES);
  if (lds_addr(smem) != 0) __builtin_trap();   // (the instruction immediates below assume the one static LDS array at 0)
  // block -> (range, unit group): the groups of a range are numbered 8 apart (same XCD, back to back: the second one
  // finds the range's outlier entries and scores in that L2)
  if ((int)blockIdx.x >= a.n_dense) {     // trailing blocks: the outlier entries (one slab each, behind the dense ranges' slabs)
    const int sr = (int)blockIdx.x - a.n_dense;
    outlier_role<Cfg>(a, smem, sr, a.partial + ((int64_t)(a.n_dense / a.groups) + sr) * a.H * kHeadDim);
    return;
  }
  int g, range;
  {
    const int G = a.groups, n_ranges = a.n_dense / G;
    const int chunk = (int)blockIdx.x / (8 * G), r = (int)blockIdx.x % (8 * G);
    const int nr = (n_ranges - 8 * chunk < 8) ? (n_ranges - 8 * chunk) : 8;
    g = r / nr;
    range = 8 * chunk + r % nr;
  }
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

  DmaLane dl = make_dma_lane<Cfg>();
  float2 *mzw = reinterpret_cast<float2 *>(smem + Cfg::SMEM_B);
  const float2 *mz = mzw;
  float *offs = reinterpret_cast<float *>(smem + Cfg::SMEM_B + Cfg::MZ_B);

  // the first chunks of the stream
  int n_prev = 0;     // VMEM pieces this wave issued in the previous step
  auto prime = [&]() {
#pragma unroll
    for (int s = 0; s < NS - 1; s++)
      if (s < n_chunks) issue_tile<Cfg>(a, dl, s, t0 + (int64_t)s * CT, row_base, n_rows_valid);
#pragma unroll
    for (int s = 0; s < NS; s++)
      if (s < n_chunks) issue_scores<Cfg>(a, dl, s, t0 + (int64_t)s * CT, h0);
  };
  prime();

  // ---- (max, 1 / normaliser) of every head this workgroup needs: its unit group's, or (the workgroup that writes the sink
  // probabilities) all of them
  {
    const bool all_heads = blockIdx.x == 0 && a.n_sink > 0;
    load_mz<Cfg>(a, mzw, all_heads ? 0 : h0, all_heads ? a.H : (n_units_valid + Cfg::UPH - 1) / Cfg::UPH);
  }
  // the sorted codebook: T' = T - T[0] and 1 / (T[n-1] - T[0])  (uniform loads)
  float tprime[Cfg::N];
#pragma unroll
  for (int i = 0; i < Cfg::N; i++) tprime[i] = a.table[i] - a.table[0];
  const float inv_dt = tprime[Cfg::N - 1] != 0.f ? 1.0f / tprime[Cfg::N - 1] : 0.f;
  __syncthreads();          // mz visible
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

  struct Slice { bool writer; float *dst; int head; };
  auto slice_of = [&]() {
    int t = threadIdx.x;
    asm volatile("" : "+v"(t));
    const int ul_ = t % Cfg::UW, hf_ = (t / Cfg::UW) % Cfg::HALVES, sl_ = t / (Cfg::UW * Cfg::HALVES);
    Slice r;
    r.writer = sl_ == 0 && ul_ < n_units_valid;
    r.dst = a.partial + (int64_t)range * C + (int64_t)(u0 + ul_) * CH + hf_ * CHL;
    r.head = ul_ / Cfg::UPH;
    return r;
  };
  // ---- the pair table: entry b = (T'[low code], T'[high code]), R interleaved copies -----------------------------------
  {
    f32x2 *tab = reinterpret_cast<f32x2 *>(smem + Cfg::PAIR_OFF);
    constexpr int NE = 1 << Cfg::PBITS;
    for (int i = tid; i < NE * Cfg::R; i += Cfg::NT) {
      const int b = i / Cfg::R;
      const int c0 = b & (Cfg::N - 1), c1 = b >> BITS;
      float x0 = 0.f, x1 = 0.f;
#pragma unroll
      for (int k = 0; k < Cfg::N; k++) {
        x0 = (c0 == k) ? tprime[k] : x0;
        x1 = (c1 == k) ? tprime[k] : x1;
      }
      f32x2 t = {x0, x1};
      tab[i] = t;
    }
  }
  // this lane's element of the weight buffers: head hr, token tid % CT
  const bool conv_lane = tid < Cfg::HW * CT;             // (wave-uniform: HW*CT is a multiple of 64)
  float myM = 0.f, myZ = 1.f;
  {
    const int hr = tid / CT;
    const int hc = (conv_lane && h0 + hr < a.H) ? h0 + hr : (h0 < a.H ? h0 : a.H - 1);
    myM = mz[hc].x;
    myZ = mz[hc].y;
  }
  float offsum = 0.f;
  // raw score -> weight p * sf in place; tokens at or past the end of the range get 0
  auto convert = [&](int pb, int64_t c0) {
    float *pp = reinterpret_cast<float *>(smem + Cfg::p_off(pb)) + tid;
    const float2 e = *reinterpret_cast<const float2 *>(smem + Cfg::e_off(pb) + (tid % CT) * 8);
    const float x = *pp;
    const bool in = c0 + (tid % CT) < t1;
    const float p = prob_of(x, a.inv, myM, myZ);
    const float sf = (e.y - e.x) * inv_dt;
    *pp = in ? p * sf : 0.f;
    offsum += in ? p * e.x : 0.f;
  };
  dma_wait_all();
  __syncthreads();                        // first chunks landed for everybody, pair table written
  if (conv_lane) convert(0, t0);          // (visible after the first chunk's barrier)

  float acc[CHL];
#pragma unroll
  for (int i = 0; i < CHL; i++) acc[i] = 0.f;
  // per-lane LDS offsets inside a stage: its quads of its unit's rows, its head's weights for its slot's tokens
  uint32_t taddr[Cfg::QPL][WORDS];
#pragma unroll
  for (int qq = 0; qq < Cfg::QPL; qq++)
#pragma unroll
    for (int wi = 0; wi < WORDS; wi++) {
      const int r = ul * WORDS + wi;
      const int rot = (r >> Cfg::SH) & (Cfg::QR - 1);
      taddr[qq][wi] = (uint32_t)(r * Cfg::ROWB + (((sl * Cfg::QPL + qq + rot) & (Cfg::QR - 1)) << 4));
    }
  const uint32_t paddr = (uint32_t)((hl * CT + sl * Cfg::QPL * 4) * 4);
  const uint32_t lo = (uint32_t)((tid % Cfg::R) * 8);

  const DmaFast df = make_dma_fast<Cfg>(a, dl);
  // chunks whose DMA needs no clamps: all that start at or before `fast_end`
  int64_t fast_end = (a.max_len < a.L ? a.max_len : a.L) - CT;
  if (n_units_valid != Cfg::UW || h0 + Cfg::HW > a.H) fast_end = -1;

  // the look-ups of one chunk.  TB >= 0: the tile's LDS offset as an instruction immediate (steady-state loop, one
  // unrolled copy per stage); TB < 0: `tbase` at run time
  auto dense = [&](auto TB, uint32_t tbase, uint32_t pbase, bool conv, int pb_next, int64_t c_next) {
    constexpr int TBV = decltype(TB)::value;
    constexpr int TBS = (TBV < 0 || TBV > 65000) ? 0 : TBV;           // (ds offsets are 16 bits)
    const uint32_t tbd = TBV < 0 ? tbase : (TBV > 65000 ? (uint32_t)TBV : 0u);
    if constexpr (BITS == 4) {
      // per token: 4 address operations (pairs) -> 4 ds_read_b64 -> 8 v_fmac; the look-ups of tokens i+1 (and i+2) are in
      // flight while token i is accumulated (LDS operations return in order: lgkmcnt(4) = all but the last 4 landed)
      constexpr int QPL = Cfg::QPL, NTOK = 4 * QPL, PO = Cfg::PAIR_OFF;
      uint4 wq[QPL];
      float4 pq[QPL];
      static_for<0, QPL>([&](auto QQ) {
        constexpr int qq = decltype(QQ)::value;
        lds_read16<TBS>(wq[qq], taddr[qq][0] + tbd);
        lds_read16<qq * 16>(pq[qq], pbase);
      });
      lds_wait<0>();
      f32x2 va[4], vb[4];
      pair_lookup4<Cfg::ESH, PO>(va, wq[0].x, lo);
      pair_lookup4<Cfg::ESH, PO>(vb, wq[0].y, lo);
      static_for<0, NTOK>([&](auto I) {
        constexpr int i = decltype(I)::value;
        if constexpr (i + 1 < NTOK) lds_wait<4>(); else lds_wait<0>();
        const float w = comp<i % 4>(pq[i / 4]);
        if constexpr (i % 2 == 0) fmac8<0>(acc, va, w); else fmac8<0>(acc, vb, w);
        if constexpr (i + 2 < NTOK) {
          const uint32_t wn = comp<(i + 2) % 4>(wq[(i + 2) / 4]);
          if constexpr (i % 2 == 0) pair_lookup4<Cfg::ESH, PO>(va, wn, lo);
          else pair_lookup4<Cfg::ESH, PO>(vb, wn, lo);
        }
        if constexpr (i == 2) {
          if (conv) convert(pb_next, c_next);       // (its own LDS traffic drains the look-ups in flight once per chunk)
        }
      });
    } else {
      // 3 bit: a unit's 32 codes are one 96-bit stream over three words; this lane decodes 16 of them (8 pairs) per
      // token: a run of 5 fields in one 32-bit window (X) and a run of 3 in another (Y).
      //   half 0 (channels 0..15):  X = w0 (pairs 0..4), Y = bits 30.. of (w1:w0) (pairs 5..7)   -> acc order = channel order
      //   half 1 (channels 16..31): X = w2 >> 2 (pairs 11..15), Y = bits 16.. of (w2:w1) (pairs 8..10)
      //                             -> acc[0..10) = channels 22..31, acc[10..16) = channels 16..21 (undone at the end)
      constexpr int PO = Cfg::PAIR_OFF;
      if (conv) convert(pb_next, c_next);
      static_for<0, Cfg::QPL>([&](auto QQ) {
        constexpr int qq = decltype(QQ)::value;
        uint4 wq[3];
        float4 pq;
#pragma unroll
        for (int wi = 0; wi < 3; wi++) lds_read16<TBS>(wq[wi], taddr[qq][wi] + tbd);
        lds_read16<qq * 16>(pq, pbase);
        lds_wait<0>();
        uint32_t xs[4], ys[4];
        auto streams = [&](int e, uint32_t w0, uint32_t w1, uint32_t w2) {
          if (hf == 0) {
            xs[e] = w0;
            asm volatile("v_alignbit_b32 %0, %1, %2, 30" : "=v"(ys[e]) : "v"(w1), "v"(w0));
          } else {
            asm volatile("v_lshrrev_b32 %0, 2, %1" : "=v"(xs[e]) : "v"(w2));
            asm volatile("v_alignbit_b32 %0, %1, %2, 16" : "=v"(ys[e]) : "v"(w2), "v"(w1));
          }
        };
        streams(0, wq[0].x, wq[1].x, wq[2].x);
        streams(1, wq[0].y, wq[1].y, wq[2].y);
        streams(2, wq[0].z, wq[1].z, wq[2].z);
        streams(3, wq[0].w, wq[1].w, wq[2].w);
        const float pe[4] = {pq.x, pq.y, pq.z, pq.w};
        uint32_t ua[4], ub[4];
        f32x2 va[4], vb[4];
        // token e: group A = X fields 0..3, group B = X field 4 + Y fields 0..2; the groups of token e+1 are in flight
        // while token e is accumulated
        auto groupA = [&](int e) { field_addr<4>(ua, xs[e], lo); pair_read4<PO>(va, ua); };
        auto groupB = [&](int e) {
          uint32_t f;
          asm volatile("v_bfe_u32 %0, %1, 24, 6" : "=v"(f) : "v"(xs[e]));
          asm volatile("v_lshl_or_b32 %0, %1, 8, %2" : "=v"(ub[0]) : "v"(f), "v"(lo));
          uint32_t u3[3];
          field_addr<3>(u3, ys[e], lo);
          ub[1] = u3[0]; ub[2] = u3[1]; ub[3] = u3[2];
          pair_read4<PO>(vb, ub);
        };
        groupA(0); groupB(0);
#pragma unroll
        for (int e = 0; e < 4; e++) {
          lds_wait<4>(); fmac8<0>(acc, va, pe[e]);
          if (e + 1 < 4) groupA(e + 1);
          if (e + 1 < 4) lds_wait<4>(); else lds_wait<0>();
          fmac8<8>(acc, vb, pe[e]);
          if (e + 1 < 4) groupB(e + 1);
        }
      });
    }
  };

  // ---- steady state: chunks whose prefetches (tile NS-1 ahead, scores and row ends NS ahead) exist and need no clamps.
  // Unrolled by the stage count (tile offsets are immediates), source pointers advance by constants, every wave issues
  // a constant number of DMA pieces per chunk (constant wait counts), M0 is not saved around a piece.
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  static_assert(Cfg::K_P == 1, "one score piece per wave at most");
  const bool has_p = wave < Cfg::P_PIECES, is_e = wave == Cfg::E_WAVE;
  const bool extra2 = has_p && is_e, extra1 = has_p != is_e;
  int ci = 0;
  if constexpr (NS >= 3) {
    int64_t hi = (fast_end >= t0) ? (fast_end - t0) / CT : -1;
    if (hi > n_chunks - 1) hi = n_chunks - 1;
    int n_hot = (int)(hi - NS + 1);
    n_hot = n_hot > 0 ? n_hot / NS * NS : 0;
    if (n_hot > 0) {
      constexpr int RPI = 64 / Cfg::QR;
      const int64_t pstride = (int64_t)Cfg::NW * RPI * a.max_len;                       // words between a wave's tile pieces
      const uint32_t *tp = a.mat + (int64_t)row_base * a.max_len + t0 + (int64_t)(NS - 1) * CT;   // tile of chunk ci + NS - 1
      const float *sp = a.scores + (int64_t)h0 * a.L + t0 + (int64_t)NS * CT;           // scores of chunk ci + NS
      const float *ep = a.lut_rows + (t0 + (int64_t)NS * CT) * Cfg::N;                  // row ends of chunk ci + NS
      const uint32_t wd16 = (uint32_t)wave * 1024u, wd4 = (uint32_t)wave * 256u;
      auto hot = [&](auto ST, int c) {
        constexpr int st = decltype(ST)::value;
        constexpr int nst = (st + NS - 1) % NS;
        if (extra2) vm_wait<Cfg::K_TILE + 2>();
        else if (extra1) vm_wait<Cfg::K_TILE + 1>();
        else vm_wait<Cfg::K_TILE>();
        __syncthreads();
#pragma unroll
        for (int k = 0; k < Cfg::K_TILE; k++)
          dma16_m0(tp + k * pstride, df.tile, (uint32_t)(Cfg::TILE_OFF + nst * Cfg::TILE_B + k * Cfg::NW * 1024) + wd16);
        tp += CT;
        const int pbn = (c + NS) % NPB;
        if (has_p) dma4_m0(sp, df.p, (uint32_t)Cfg::p_off(pbn) + wd4);
        if (is_e) dma4_m0(ep, df.e, (uint32_t)Cfg::e_off(pbn));
        sp += CT;
        ep += CT * Cfg::N;
        dense(std::integral_constant<int, Cfg::TILE_OFF + st * Cfg::TILE_B>{}, 0u,
              paddr + (uint32_t)(Cfg::p_off(0) + (c % NPB) * Cfg::P_B), conv_lane, (c + 1) % NPB, t0 + (int64_t)(c + 1) * CT);
      };
      for (; ci < n_hot; ci += NS) static_for<0, NS>([&](auto ST) { hot(ST, ci + decltype(ST)::value); });
      n_prev = Cfg::K_TILE + (extra2 ? 2 : (extra1 ? 1 : 0));
    }
  }
  // ---- the other chunks (range ends, ragged shapes): clamped sources, counted pieces
  for (; ci < n_chunks; ci++) {
    const int64_t c0 = t0 + (int64_t)ci * CT;
    if (ci > 0) {
      if (NS >= 3) vm_wait_dyn(n_prev);
      else dma_wait_all();
    }
    __syncthreads();                      // chunk ci landed for everybody; stage (ci - 1) % NS is free; weights of ci visible
    const int stage = ci % NS, pcur = ci % NPB;
    n_prev = 0;
    if (ci + NS - 1 < n_chunks) {
      const int64_t ct = c0 + (int64_t)(NS - 1) * CT;
      if (ct <= fast_end) issue_tile_fast<Cfg>(a, df, (ci + NS - 1) % NS, ct, row_base);
      else issue_tile<Cfg>(a, dl, (ci + NS - 1) % NS, ct, row_base, n_rows_valid);
      n_prev += Cfg::K_TILE;
    }
    if (ci + NS < n_chunks) {
      const int64_t cs = c0 + (int64_t)NS * CT;
      if (cs <= fast_end) n_prev += issue_scores_fast<Cfg>(a, df, (ci + NS) % NPB, cs, h0);
      else n_prev += issue_scores<Cfg>(a, dl, (ci + NS) % NPB, cs, h0);
    }
    dense(std::integral_constant<int, -1>{}, (uint32_t)(Cfg::TILE_OFF + stage * Cfg::TILE_B),
          paddr + (uint32_t)(Cfg::p_off(0) + pcur * Cfg::P_B), conv_lane && ci + 1 < n_chunks, (ci + 1) % NPB, c0 + CT);
  }

  // ---- per-head offset sums: the CT lanes of a head hold its partial sums
  if (conv_lane) {
    float s = offsum;
#pragma unroll
    for (int d = CT / 2; d >= 1; d >>= 1) s += __shfl_xor(s, d);
    if (tid % CT == 0) offs[tid / CT] = s;
  }
  // ---- sum the token slots through LDS (aliases the stages)
  __syncthreads();
  float *red = reinterpret_cast<float *>(smem);
#pragma unroll
  for (int i = 0; i < CHL; i++) red[(i * Cfg::SLOTS + sl) * (Cfg::UW * Cfg::HALVES) + lu] = acc[i];
  __syncthreads();
  const Slice sc_end = slice_of();
  const bool writer = sc_end.writer;
  float *dst = sc_end.dst;
  float o[CHL];
#pragma unroll
  for (int i = 0; i < CHL; i++) o[i] = 0.f;
  if (writer) {
    const float ho = offs[sc_end.head];
#pragma unroll
    for (int i = 0; i < CHL; i++) {
      float s = red[(i * Cfg::SLOTS) * (Cfg::UW * Cfg::HALVES) + lu];
#pragma unroll
      for (int k = 1; k < Cfg::SLOTS; k++) s += red[(i * Cfg::SLOTS + k) * (Cfg::UW * Cfg::HALVES) + lu];
      // 3 bit, half 1: accumulator i holds channel (i + 6) % 16 of the half
      const int ch = (BITS == 3) ? ((i + (hf ? 6 : 0)) & 15) : i;
      static_for<0, CHL>([&](auto J) {
        if (decltype(J)::value == ch) o[decltype(J)::value] = s + ho;
      });
    }
  }
  if (writer) {
#pragma unroll
    for (int i = 0; i < CHL; i += 4) *reinterpret_cast<float4 *>(dst + i) = make_float4(o[i], o[i + 1], o[i + 2], o[i + 3]);
  }
}

struct Plan {
  int64_t tr, str;
  int n_ranges, groups, n_units, n_sranges;
  size_t bytes;
};

#ifndef KVQ_VA_STR
#define KVQ_VA_STR 256     // tokens per outlier workgroup: its 256 x 42 entries are one round of loads (21 per lane)
#endif

template <typename Cfg>
static Plan plan(int H, int64_t L) {
  Plan pl;
  pl.n_units = H * Cfg::UPH;
  pl.groups = (pl.n_units + Cfg::UW - 1) / Cfg::UW;
  int64_t want = 256 * Cfg::WGPC / pl.groups;      // one generation of workgroups
  if (want < 1) want = 1;
  int64_t tr = (L + want - 1) / want;
  tr = (tr + Cfg::CT - 1) / Cfg::CT * Cfg::CT;
  if (tr < 4 * Cfg::CT) tr = 4 * Cfg::CT;
  pl.tr = tr;
  pl.n_ranges = (int)((L + tr - 1) / tr);
  if (pl.n_ranges < 1) pl.n_ranges = 1;
  pl.str = KVQ_VA_STR;
  pl.n_sranges = (int)((L + pl.str - 1) / pl.str);
  // slabs of the dense ranges and of the outlier workgroups, then the merged (max, normaliser) pairs
  pl.bytes = (size_t)(pl.n_ranges + pl.n_sranges) * H * kHeadDim * sizeof(float) + (size_t)H * 2 * sizeof(float);
  return pl;
}

template <int BITS, typename V>
static int launch(Args a, float *mul, int accumulate, hipStream_t st) {
  using Cfg = ACfg<BITS, V>;
  const Plan pl = plan<Cfg>(a.H, a.L);
  a.tr = pl.tr;
  a.groups = pl.groups;
  a.n_units = pl.n_units;
  a.sink_out = mul;
  if (a.v_sink != nullptr) accumulate = 1;     // the reduce adds the slabs onto the sink tokens' output
  const bool sparse = a.idx != nullptr;
  const int n_sr = sparse ? pl.n_sranges : 0;
  a.n_dense = pl.n_ranges * pl.groups;
  a.str = pl.str;
  a.mz = nullptr;
  if (a.n_parts > mix_merge_in_kernel_parts()) {
    float *mz = a.partial + (size_t)(pl.n_ranges + pl.n_sranges) * a.H * kHeadDim;   // (tail of the workspace)
    int rc0 = launch_softmax_merge(a.parts, a.n_parts, a.sink, a.sink_probs, a.n_sink, mz, a.v_sink, mul, a.H, st);
    if (rc0) return rc0;
    a.mz = mz;
  }
  kvq_step_mark_pv(st);
  mix_va_kernel<BITS, V><<<dim3(a.n_dense + n_sr), dim3(Cfg::NT), 0, st>>>(a);
  int rc = check_launch();
  if (rc) return rc;
  return launch_mix_reduce(a.partial, mul, pl.n_ranges + n_sr, 1, a.H * kHeadDim, accumulate, st);
}

// geometry (lanes, tokens per chunk, stages, table copies, workgroups per CU).  Measured and dropped
// (profiles/r04_affine_pv.txt): 256 lanes (111.6 us at 128K), 32-token chunks with 16 copies on one workgroup per CU
// (100.5 - 106.2), two stages with 32 copies (99.7 - 100.1); this one: 91.3 - 97.4
typedef Var<512, 16, 3, 8, 2> V4A;
typedef Var<512, 16, 2, 32, 2> V3A;

static size_t plan_bytes(int bits, int H, int64_t L) {
  return bits == 3 ? plan<ACfg<3, V3A>>(H, L).bytes : plan<ACfg<4, V4A>>(H, L).bytes;
}

}  // namespace va
}  // namespace kvq

using namespace kvq;

extern "C" {

int kvq_mix_v_affine_supported(int bits, int H, int hd, int64_t L, int64_t max_len) {
  return (bits == 3 || bits == 4) && hd == kHeadDim && H > 0 && H <= 128 && L > 0 && max_len % 4 == 0 &&
         max_len >= 4 && (int64_t)H * (hd / 32 * bits) * max_len < (1ll << 31);
}

size_t kvq_mix_v_affine_workspace_bytes(int bits, int H, int hd, int64_t L) {
  if (bits < 2 || bits > 4 || H <= 0 || hd != kHeadDim || L <= 0) return 0;
  size_t b = kvq_mix_v_workspace_bytes(bits, 1, H, hd, L);
  size_t a = (bits == 3 || bits == 4) ? va::plan_bytes(bits, H, L) : 0;
  return a > b ? a : b;
}

int kvq_mix_v_softmax_affine(int bits, const float *scores, const float *parts, int n_parts, float inv_sqrt_hd,
                             const uint16_t *sink_scores, uint16_t *sink_probs, int n_sink, const uint16_t *v_sink,
                             float *probs, const int32_t *mat, float *mul, const float *lut_rows, const float *table,
                             int H, int hd, int64_t L, int64_t max_len, const float *outliers,
                             const int32_t *outlier_idx, int n_out, int accumulate, void *workspace,
                             size_t workspace_bytes, void *stream) {
  if (!scores || !parts || n_parts <= 0 || n_sink < 0 || H <= 0 || L <= 0 || !mat || !mul || !lut_rows) return KVQ_EINVAL;
  if (n_sink > 0 && (!sink_scores || !sink_probs)) return KVQ_EINVAL;
  if (v_sink != nullptr && (n_sink <= 0 || accumulate)) return KVQ_EINVAL;
  const bool ok = table != nullptr && kvq_mix_v_affine_supported(bits, H, hd, L, max_len) && L <= max_len &&
                  (reinterpret_cast<uintptr_t>(mat) | reinterpret_cast<uintptr_t>(lut_rows)) % 16 == 0;
  if (!ok)   // shapes / widths the affine kernel does not take: the per-row form
    return kvq_mix_v_softmax(bits, scores, parts, n_parts, inv_sqrt_hd, sink_scores, sink_probs, n_sink, v_sink, probs, mat,
                             mul, lut_rows, H, hd, L, max_len, outliers, outlier_idx, n_out, accumulate, workspace,
                             workspace_bytes, stream);
  const bool sparse = outlier_idx != nullptr;
  if ((outliers && !outlier_idx) || (sparse && n_out <= 0)) return KVQ_EINVAL;
  if (!workspace || workspace_bytes < kvq_mix_v_affine_workspace_bytes(bits, H, hd, L)) return KVQ_EWORKSPACE;
  va::Args a;
  a.mat = reinterpret_cast<const uint32_t *>(mat);
  a.lut_rows = lut_rows;
  a.table = table;
  a.outliers = outliers;
  a.idx = outlier_idx;
  a.partial = reinterpret_cast<float *>(workspace);
  a.H = H;
  a.L = L;
  a.max_len = max_len;
  a.tr = 0;
  a.groups = 1;
  a.n_units = 0;
  a.n_out = n_out;
  a.n_out_magic = sparse ? (uint32_t)(((1ull << 32) + (uint64_t)n_out - 1) / (uint64_t)n_out) : 0u;
  a.scores = scores;
  a.mz = nullptr;
  a.parts = parts;
  a.n_parts = n_parts;
  a.sink = reinterpret_cast<const __half *>(sink_scores);
  a.sink_probs = reinterpret_cast<__half *>(sink_probs);
  a.n_sink = n_sink;
  a.v_sink = reinterpret_cast<const __half *>(v_sink);
  a.sink_out = nullptr;
  a.inv = inv_sqrt_hd;
  hipStream_t st = (hipStream_t)stream;
  if (bits == 3) return va::launch<3, va::V3A>(a, mul, accumulate, st);
  return va::launch<4, va::V4A>(a, mul, accumulate, st);
}

}  // extern "C"
