// One decode token's attention over the compressed cache of a layer as ONE streaming kernel + a small merge: the
// split-L ("flash-decoding") form of the reference's decode branch (modeling_llama.py:1948-2000 around
// QuantK / QuantV.forward_fused_sparse; kernels KCU:3040-3209 + 473-521 and 3211-3433 + 437-470).
//
// A workgroup owns a TILE of 256 cached tokens (and, for the ragged last tile, a group of heads):
//   1. K phase  -- exactly the score kernel's tile body (kvq_score_k.hip: score_k_tile): q.K^T with RoPE on the
//      dequantised keys + the outlier entries of the tile's tokens, scores of all heads collected in LDS;
//   2. softmax, tile-local -- x = half(half(score) * 1/sqrt(d)) as the reference forms it (ML:873-874, 1972-1973), per
//      head m = max_t x, ptilde_t = half(exp(x_t - m)), l = sum_t ptilde_t; (m, l) go to memory, ptilde stays in LDS;
//   3. V phase  -- the per-row look-up loop of kvq_mix_v.hip over the SAME tokens: sum_t ptilde_t * Vhat_t for the
//      workgroup's heads (LDS-DMA row tiles, 16-token chunks, three stages, lane owns a row unit), the V outlier entries
//      of the tile (32.32 fixed-point LDS adds against the LDS-resident ptilde -- no probability staging), one slab of
//      partial outputs per tile.
// kvq_fused_merge then forms, per head, M = max(m_tile, sinks), Z = sum l_tile e^(m_tile - M) (+ sinks) and
// out = sum_tile slab_tile * e^(m_tile - M) / Z (+ the fp16 sink tokens' share).
// What it buys over the separate kernels: the [H][L] scores never travel through memory (2 x 16 MB at 128K), one
// launch boundary and the softmax merge launch are gone, the V outlier phase needs no staging -- and the K phase (bound by
// VALU / LDS look-ups) of one workgroup runs beside the V phase (bound by the HBM stream) of the other workgroup of
// its CU.
// Numerics: a probability is rounded to fp16 relative to its TILE's maximum, not to the row's normaliser as in the
// reference (ML:1976): the same relative rounding (2^-11) at a different scale, and no flush of probabilities below
// 2^-24.  q.K^T / p.V are held to the north-star 1e-3 against the reference pipeline (tests/test_fused_decode_gpu.py).
#include "kvq_common.h"
#include "kvq_host.h"
#include "kvq_ktab.h"
#include "kvq_mix_lut.h"
#include "kvq_score_k_tile.h"

#include <hip/hip_fp16.h>

namespace kvq {
namespace fused {

constexpr int NW = 8, NT = 512, T = 256;

template <int BITS>
struct FCfg {
  using G = KGeom<BITS, true, NW, true>;
  static constexpr int N = Fmt<BITS>::kN;
  static constexpr int WORDS = BITS == 3 ? 3 : 1;
  static constexpr int CH = BITS == 4 ? 8 : (BITS == 3 ? 32 : 16);   // channels per row unit
  static constexpr int UPH = kHeadDim / CH;
  static constexpr int HALVES = BITS == 3 ? 2 : 1;
  static constexpr int UW = 256 / HALVES;              // units per pass
  static constexpr int CHL = CH / HALVES;
  static constexpr int SLOTS = NT / (UW * HALVES);     // 2
  static constexpr int CT = 16, QR = 4, SH = 2, QPL = QR / SLOTS;
  static constexpr int NCH = T / CT;                   // chunks per pass
  static constexpr int ROWS = UW * WORDS, ROWB = CT * 4, TILE_B = ROWS * ROWB;   // 16 KB / 24 KB
  static constexpr int NS = BITS == 4 ? 3 : 2;
  static constexpr int LUT_B = CT * N * 4;             // codebook rows of a chunk
  static constexpr int LUT_SLOTS = LUT_B / 16;
  static constexpr int TILE_PIECES = TILE_B / 1024, K_TILE = TILE_PIECES / NW;
  // LDS of the V phase.  ptilde sits where the K phase kept q (free once the scores are complete); the tile stages from 0
  // (table buffers, then the score tile); multi-pass widths (4 bit) keep the slot reduction in its own 8 KB, single-pass
  // ones put it over stage 0 at the end; the chunks' codebook rows in front of ptilde when they fit, else behind it
  static constexpr int PT_OFF = G::SC_OFF + G::SC_B;                 // half [32][T]: 16 KB
  static constexpr int PT_B = kSparseHpg * T * 2;
  static constexpr int ACC_OFF = G::SC_OFF;                          // outlier accumulators: 4096 x 8 B over the score tile
  static constexpr int MAX_PASS = (kSparseHpg * UPH + UW - 1) / UW;  // passes of a 32-head workgroup
  static constexpr int RED_B = UW * HALVES * CHL * 4 * (SLOTS - 1);
  static constexpr int RED_OFF = MAX_PASS > 1 ? NS * TILE_B : 0;     // (single pass: over stage 0, after the loop)
  static constexpr int RED_DED = MAX_PASS > 1 ? RED_B : 0;
  static constexpr bool LUT_FRONT = NS * TILE_B + RED_DED + NS * LUT_B <= PT_OFF;
  static constexpr int LUT_OFF = LUT_FRONT ? NS * TILE_B + RED_DED : PT_OFF + PT_B;
  static constexpr int SMEM_B = LUT_FRONT ? PT_OFF + PT_B : PT_OFF + PT_B + NS * LUT_B;
  // the first tile may be requested before the tile-local softmax has read the scores if it lands in the table buffers
  static constexpr bool EARLY_TILE = TILE_B <= G::PF * G::TAB_B;
  static constexpr int tile_off(int st) { return st * TILE_B; }
  static_assert(NS * TILE_B + RED_DED <= PT_OFF, "the tile stages end below ptilde");
  static_assert(G::SMEM_B <= 81920 && SMEM_B <= 81920, "two workgroups per CU");
  static_assert(TILE_PIECES % NW == 0, "tile pieces per wave");
  static_assert(G::QL_B >= PT_B, "ptilde takes over the q buffer");
  static constexpr int SMEM_ALL = G::SMEM_B > SMEM_B ? G::SMEM_B : SMEM_B;
};

struct FusedArgs {
  ScoreKArgs k;
  const uint32_t *vmat;    // [rows][max_len]
  const float *vrows;      // [max_len][N]
  const float *vout;       // V outlier values [max_len][n_out] or null (compact: not supported here)
  const int32_t *vidx;
  float *slabs;            // [n_tiles][C]
  float *stats;            // [H][n_tiles][2]
  int n_tiles;
  float inv;
};

// code of channel I of a row unit held in WORDS words
template <int BITS, int I, int WORDS>
__device__ __forceinline__ unsigned unit_code(const uint32_t (&w)[WORDS]) {
  if constexpr (BITS == 3) return code_of<3, I>(w);
  else if constexpr (BITS == 4) return (w[0] >> (4 * I)) & 0xfu;
  else return (w[0] >> (2 * I)) & 0x3u;
}

// wait until at most n (wave-uniform) VMEM operations of this wave are outstanding
__device__ __forceinline__ void vm_wait_dyn(int n) {
  switch (n) {
    case 1: vm_wait<1>(); break;
    case 2: vm_wait<2>(); break;
    case 3: vm_wait<3>(); break;
    case 4: vm_wait<4>(); break;
    default: vm_wait<0>(); break;
  }
}

// per-lane constants of the V tile DMA (kvq_mix_v.hip's scheme: piece k of a wave is NW * 64/QR rows further down)
struct VDma {
  uint32_t tile_row, tile_q4;
};

// DMA of one 16-token chunk of the V phase into stage `stage`: the packed rows (`tile`) and / or the tokens' codebook rows
// (`rows`); returns the number of pieces this wave issued (wave-uniform)
template <int BITS>
__device__ __forceinline__ int issue_v_chunk(const FusedArgs &f, const VDma &d, int stage, int64_t c0, int row_base,
                                             int n_rows_valid, bool tile, bool rows) {
  using Cfg = FCfg<BITS>;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  constexpr int RPI = 64 / Cfg::QR;
  const int64_t max_len = f.k.max_len;
  const int lim_len = (int)(max_len - c0 < (1 << 30) ? max_len - c0 : (1 << 30));   // tokens to the end of the rows (>= 4, multiple of 4)
  int n = 0;
  if (tile) {
    const uint32_t *gbase = f.vmat + (int64_t)row_base * max_len + c0;
    const int tq = (int)d.tile_q4;
    const uint32_t toff = (uint32_t)(tq + 4 > lim_len ? lim_len - 4 : tq);
    const uint32_t dst = (uint32_t)(stage * Cfg::TILE_B);
#pragma unroll
    for (int k = 0; k < Cfg::K_TILE; k++) {
      const int j = wave + k * NW;
      int r = d.tile_row + k * NW * RPI;
      if (r >= n_rows_valid) r = n_rows_valid - 1;
      const uint32_t voff = ((uint32_t)r * (uint32_t)max_len + toff) * 4u;
      dma16(gbase, voff, dst + j * 1024);
    }
    n += Cfg::K_TILE;
  }
  // the chunk's codebook rows: LDS row (qq*4+e)*SLOTS + slot holds token (slot*QPL+qq)*4+e (the rows the slots decode in one
  // step sit next to each other)
  if (rows && wave == 0) {
    if ((int)threadIdx.x < Cfg::LUT_SLOTS) {
      const int s = threadIdx.x;
      const int pidx = (s * 4) / Cfg::N;
      const int qe = pidx / Cfg::SLOTS, slp = pidx % Cfg::SLOTS;
      const int tok = (slp * Cfg::QPL + qe / 4) * 4 + qe % 4;
      const int tc = tok < lim_len ? tok : lim_len - 1;
      const uint32_t voff = (uint32_t)(tc * Cfg::N + (s * 4) % Cfg::N) * 4u;
      dma16(f.vrows + c0 * Cfg::N, voff, (uint32_t)(Cfg::LUT_OFF + stage * Cfg::LUT_B));
    }
    n += 1;
  }
  return n;
}

template <int BITS>
__global__ __launch_bounds__(NT, 4) void fused_decode_kernel(FusedArgs f) {
  using Cfg = FCfg<BITS>;
  using G = typename Cfg::G;
  constexpr int N = Cfg::N, CH = Cfg::CH, WORDS = Cfg::WORDS, CT = Cfg::CT, CHL = Cfg::CHL, NS = Cfg::NS, SCS = G::SCS;
  __shared__ __attribute__((aligned(16))) unsigned char smem[Cfg::SMEM_ALL];
  if (lds_addr(smem) != 0) __builtin_trap();

  // ---- 1. K phase: the scores of the tile, all heads of the group, in LDS
  const KTile kt = score_k_tile<BITS, true, NW, true, false>(f.k, smem);
  // (the tile body has drained its hand-issued loads on every path it can take; the explicit wait makes that a property
  //  of the control-flow graph, which is what tools/check_isa.py verifies: nothing below may meet a register in flight)
  vm_wait<0>();
  const int tid = threadIdx.x;
  const int nh = kt.nh, h0 = kt.h0, ntok = kt.ntok, tile_i = kt.tile_i;
  const int64_t tile0 = kt.tile0;
  const int C = f.k.H * kHeadDim;
  const int64_t max_len = f.k.max_len;
  __syncthreads();                      // a wave wrote only its own tokens' rows; the table buffers and q are free

  // V geometry: passes over groups of UW row units of this workgroup's heads
  const int n_units = nh * Cfg::UPH;
  const int n_pass = (n_units + Cfg::UW - 1) / Cfg::UW;
  const int u_base = h0 * Cfg::UPH;
  const int n_chunks = (ntok + CT - 1) / CT;
  const int n_steps = n_pass * n_chunks;
  VDma dl;
  dl.tile_row = tid / Cfg::QR;
  dl.tile_q4 = 4 * (((tid % Cfg::QR) - (((tid / Cfg::QR) >> Cfg::SH) & (Cfg::QR - 1))) & (Cfg::QR - 1));
  auto issue_step = [&](int s, bool tile, bool rows) -> int {
    const int pass = s / n_chunks;
    int nu = n_units - pass * Cfg::UW;
    if (nu > Cfg::UW) nu = Cfg::UW;
    return issue_v_chunk<BITS>(f, dl, s % NS, tile0 + (int64_t)(s % n_chunks) * CT, (u_base + pass * Cfg::UW) * WORDS,
                               nu * WORDS, tile, rows);
  };
  // the first tile goes into the table buffers right away (4 bit); the V outlier entries of the tile travel with it
  if (Cfg::EARLY_TILE) issue_step(0, true, false);
  constexpr int RB = 24;
  const bool v_sparse = f.vidx != nullptr;
  const unsigned nent = v_sparse ? (unsigned)ntok * (unsigned)f.k.n_out : 0u;
  int orow[RB];
  float oval[RB];
  if (v_sparse) {
    const float *ov = f.vout + tile0 * f.k.n_out;
    const int32_t *oi = f.vidx + tile0 * f.k.n_out;
#pragma unroll
    for (int j = 0; j < RB; j++) {
      const unsigned e = j * NT + tid;
      const unsigned ec = e < nent ? e : nent - 1;
      orow[j] = oi[ec];
      oval[j] = ov[ec];
    }
  }

  // ---- 2. tile-local softmax: NT/32 lanes per head read the head's column of the score tile
  {
    const float *sc = reinterpret_cast<const float *>(smem + G::SC_OFF);
    __half *pt = reinterpret_cast<__half *>(smem + Cfg::PT_OFF);
    constexpr int TPH = NT / 32;
    const int hh = tid / TPH, r = tid % TPH;
    float x[T / TPH];
    float m = -INFINITY;
#pragma unroll
    for (int k = 0; k < T / TPH; k++) {
      const int j = r + k * TPH;
      x[k] = (j < ntok && hh < nh) ? scaled(sc[j * SCS + ((hh + j) & (SCS - 1))], f.inv) : -INFINITY;
      m = fmaxf(m, x[k]);
    }
#pragma unroll
    for (int d = TPH / 2; d >= 1; d >>= 1) m = fmaxf(m, __shfl_xor(m, d));
    float sm = 0.f;
    __half ph[T / TPH];
#pragma unroll
    for (int k = 0; k < T / TPH; k++) {
      const float p = (x[k] > -INFINITY) ? exp_neg(x[k] - m) : 0.f;
      ph[k] = __float2half_rn(p);
      sm += __half2float(ph[k]);
    }
#pragma unroll
    for (int d = TPH / 2; d >= 1; d >>= 1) sm += __shfl_xor(sm, d);
    if (hh < nh) {
#pragma unroll
      for (int k = 0; k < T / TPH; k++) pt[hh * T + r + k * TPH] = ph[k];
      if (r == 0) {
        float *dst = f.stats + ((int64_t)(h0 + hh) * f.n_tiles + tile_i) * 2;
        dst[0] = m;
        dst[1] = sm;
      }
    }
  }
  __syncthreads();                      // every lane has read its scores (the accumulators / later stages overwrite them); ptilde visible
  // ---- 3a. V outlier entries of the tile: val * ptilde into 32.32 fixed-point accumulators of the workgroup's channels
  // (over the score tile); the sums wait in the slab for the dense sums of the same lanes
  const int ul = tid % Cfg::UW;
  const int hf = __builtin_amdgcn_readfirstlane((tid / Cfg::UW) % Cfg::HALVES);
  const int lu = tid % (Cfg::UW * Cfg::HALVES);
  const int sl = tid / (Cfg::UW * Cfg::HALVES);
  float *slab = f.slabs + (int64_t)tile_i * C + (int64_t)h0 * kHeadDim;
  if (v_sparse) {
    long long *sacc = reinterpret_cast<long long *>(smem + Cfg::ACC_OFF);
    const int cn = nh * kHeadDim;
    for (int i = tid; i < cn; i += NT) sacc[i] = 0;
    __syncthreads();
    const __half *pt = reinterpret_cast<const __half *>(smem + Cfg::PT_OFF);
#pragma unroll
    for (int j = 0; j < RB; j++) {
      const unsigned e = j * NT + tid;
      const unsigned ec = e < nent ? e : nent - 1;
      const unsigned tl = __umulhi(ec, f.k.n_out_magic);
      const unsigned rel = (unsigned)(orow[j] - h0 * kHeadDim);
      const bool mine = e < nent && rel < (unsigned)cn;
      const unsigned hh = mine ? (rel >> 7) : 0u;
      const float p = __half2float(pt[hh * T + tl]);
      if (mine) {
        const float x = oval[j] * p;
        const float fl = floorf(x);
        const unsigned lo = (unsigned)((x - fl) * 4294967296.0f);
        const int hi = (int)fl;
        atomicAdd(reinterpret_cast<unsigned long long *>(&sacc[rel]), ((unsigned long long)(unsigned)hi << 32) | lo);
      }
    }
    __syncthreads();
    if (sl == 0) {
      for (int pass = 0; pass < n_pass; pass++) {
        const int u = pass * Cfg::UW + ul;
        if (u < n_units) {
#pragma unroll
          for (int i = 0; i < CHL; i += 4) {
            const long long *a4 = sacc + u * CH + hf * CHL + i;
            *reinterpret_cast<float4 *>(slab + u * CH + hf * CHL + i) =
                make_float4((float)((double)a4[0] * (1.0 / 4294967296.0)), (float)((double)a4[1] * (1.0 / 4294967296.0)),
                            (float)((double)a4[2] * (1.0 / 4294967296.0)), (float)((double)a4[3] * (1.0 / 4294967296.0)));
          }
        }
      }
    }
    __syncthreads();                    // the accumulators are read: the stages over them may land
  }
  // the rest of the first NS-1 chunks
  issue_step(0, !Cfg::EARLY_TILE, true);
  if (NS > 2 && n_steps > 1) issue_step(1, true, true);

  // ---- 3b. dense p.V of the tile, pass by pass
  float acc[CHL];
#pragma unroll
  for (int i = 0; i < CHL; i++) acc[i] = 0.f;
  uint32_t taddr[Cfg::QPL][WORDS];
#pragma unroll
  for (int qq = 0; qq < Cfg::QPL; qq++)
#pragma unroll
    for (int wi = 0; wi < WORDS; wi++) {
      const int r = ul * WORDS + wi;
      const int rot = (r >> Cfg::SH) & (Cfg::QR - 1);
      taddr[qq][wi] = (uint32_t)(r * Cfg::ROWB + (((sl * Cfg::QPL + qq + rot) & (Cfg::QR - 1)) << 4));
    }
  const uint32_t slotpat = (uint32_t)sl * 0x40404040u;      // (4 bit: slot*64 in every byte)
  float *red = reinterpret_cast<float *>(smem + Cfg::RED_OFF);
  typedef float f32x4_t __attribute__((ext_vector_type(4)));
  // the sums of a finished pass: slot 1's lanes park theirs in LDS (before a step barrier), slot 0's lanes add them and the
  // outlier sums waiting in the slab (after it) and write the slab
  auto finish_pass = [&](int pass, const float (&o0)[CHL]) {
    const int u = pass * Cfg::UW + ul;
    if (sl == 0 && u < n_units) {
      float *dst = slab + u * CH + hf * CHL;
      f32x4_t q[CHL / 4];
      if (v_sparse) {
#pragma unroll
        for (int i = 0; i < CHL; i += 4) asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(q[i / 4]) : "v"(dst + i) : "memory");
      }
      float o[CHL];
#pragma unroll
      for (int i = 0; i < CHL; i++) {
        float sum = o0[i];
#pragma unroll
        for (int k = 1; k < Cfg::SLOTS; k++) sum += red[((k - 1) * CHL + i) * (Cfg::UW * Cfg::HALVES) + lu];
        o[i] = sum;
      }
      if (v_sparse) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
        for (int i = 0; i < CHL; i += 4) {
          asm volatile("" : "+v"(q[i / 4]));
          o[i] += q[i / 4].x; o[i + 1] += q[i / 4].y; o[i + 2] += q[i / 4].z; o[i + 3] += q[i / 4].w;
        }
      }
#pragma unroll
      for (int i = 0; i < CHL; i += 4) *reinterpret_cast<float4 *>(dst + i) = make_float4(o[i], o[i + 1], o[i + 2], o[i + 3]);
    }
  };
  // one chunk's look-ups; ST: its stage (tile and codebook-row offsets are instruction immediates)
  auto dense = [&](auto ST, int pass, int ci) {
    constexpr int st = decltype(ST)::value;
    constexpr int TB = st * Cfg::TILE_B, LB = Cfg::LUT_OFF + st * Cfg::LUT_B;
    const int u = pass * Cfg::UW + ul;
    const int hh = (u < n_units ? u : 0) / Cfg::UPH;
    const __half *prow = reinterpret_cast<const __half *>(smem + Cfg::PT_OFF) + hh * T + ci * CT + sl * Cfg::QPL * 4;
    if constexpr (BITS == 4) {
      constexpr int TS = Cfg::SLOTS * N * 4;               // bytes between the rows of consecutive tokens of a slot
      static_for<0, Cfg::QPL>([&](auto QQ) {
        constexpr int qq = decltype(QQ)::value;
        uint4 wq;
        lds_read16<TB>(wq, taddr[qq][0]);
        const uint2 ph = *reinterpret_cast<const uint2 *>(prow + qq * 4);
        lds_wait<0>();
        const float p0 = __half2float(__ushort_as_half((unsigned short)(ph.x & 0xffffu)));
        const float p1 = __half2float(__ushort_as_half((unsigned short)(ph.x >> 16)));
        const float p2 = __half2float(__ushort_as_half((unsigned short)(ph.y & 0xffffu)));
        const float p3 = __half2float(__ushort_as_half((unsigned short)(ph.y >> 16)));
        uint32_t we, wo, ua[8], ub[8];
        float va[8], vb[8];
        nib_prep(we, wo, wq.x, slotpat); nib_extract(ua, we, wo); lut_read8<LB + (qq * 4 + 0) * TS>(va, ua);
        nib_prep(we, wo, wq.y, slotpat); nib_extract(ub, we, wo); lut_read8<LB + (qq * 4 + 1) * TS>(vb, ub);
        lds_wait<8>(); fmac8(acc, va, p0);
        nib_prep(we, wo, wq.z, slotpat); nib_extract(ua, we, wo); lut_read8<LB + (qq * 4 + 2) * TS>(va, ua);
        lds_wait<8>(); fmac8(acc, vb, p1);
        nib_prep(we, wo, wq.w, slotpat); nib_extract(ub, we, wo); lut_read8<LB + (qq * 4 + 3) * TS>(vb, ub);
        lds_wait<8>(); fmac8(acc, va, p2);
        lds_wait<0>(); fmac8(acc, vb, p3);
      });
    } else {
      // 3 / 2 bit (first version: the compiler's schedule of the generic decode)
      const float *tab0 = reinterpret_cast<const float *>(smem + LB);
      static_for<0, Cfg::QPL>([&](auto QQ) {
        constexpr int qq = decltype(QQ)::value;
        uint4 wq[WORDS];
#pragma unroll
        for (int wi = 0; wi < WORDS; wi++) wq[wi] = *reinterpret_cast<const uint4 *>(smem + TB + taddr[qq][wi]);
        const uint2 ph = *reinterpret_cast<const uint2 *>(prow + qq * 4);
        const float pe[4] = {__half2float(__ushort_as_half((unsigned short)(ph.x & 0xffffu))),
                             __half2float(__ushort_as_half((unsigned short)(ph.x >> 16))),
                             __half2float(__ushort_as_half((unsigned short)(ph.y & 0xffffu))),
                             __half2float(__ushort_as_half((unsigned short)(ph.y >> 16)))};
        static_for<0, 4>([&](auto E) {
          constexpr int e = decltype(E)::value;
          uint32_t w[WORDS];
#pragma unroll
          for (int wi = 0; wi < WORDS; wi++) w[wi] = e == 0 ? wq[wi].x : (e == 1 ? wq[wi].y : (e == 2 ? wq[wi].z : wq[wi].w));
          const float *tab = tab0 + ((qq * 4 + e) * Cfg::SLOTS + sl) * N;
          static_for<0, Cfg::HALVES>([&](auto HF) {
            if (hf == decltype(HF)::value)
              static_for<0, CHL>([&](auto I) {
                constexpr int i = decltype(I)::value;
                acc[i] = fmaf(tab[unit_code<BITS, decltype(HF)::value * CHL + i, WORDS>(w)], pe[e], acc[i]);
              });
          });
        });
      });
    }
  };
  float done[CHL];           // slot 0: the finished pass's own sums, until the next step's barrier has published slot 1's
#pragma unroll
  for (int i = 0; i < CHL; i++) done[i] = 0.f;
  int pending = -1;
  int n_prev = 0;            // DMA pieces this wave issued in the previous step (they may stay in flight)
  for (int s = 0; s < n_steps; s++) {
    const int pass = s / n_chunks, ci = s % n_chunks;
    // chunk s was requested NS-1 steps ago; what the previous step requested may stay in flight
    if (NS > 2 && s > 0) vm_wait_dyn(n_prev); else dma_wait_all();
    __syncthreads();
    if (pending >= 0) {
      finish_pass(pending, done);
      pending = -1;
    }
    n_prev = 0;
    if (s + NS - 1 < n_steps) n_prev = issue_step(s + NS - 1, true, true);
    switch (s % NS) {
      case 0: dense(std::integral_constant<int, 0>{}, pass, ci); break;
      case 1: dense(std::integral_constant<int, 1 % NS>{}, pass, ci); break;
      default: dense(std::integral_constant<int, 2 % NS>{}, pass, ci); break;
    }
    // end of a pass: slot >= 1 parks its sums, slot 0 keeps them until the next barrier
    if (ci == n_chunks - 1) {
      if (Cfg::MAX_PASS == 1) __syncthreads();       // (the parking space is over stage 0: everybody is done reading it)
      if (sl > 0) {
#pragma unroll
        for (int i = 0; i < CHL; i++) red[((sl - 1) * CHL + i) * (Cfg::UW * Cfg::HALVES) + lu] = acc[i];
      } else {
#pragma unroll
        for (int i = 0; i < CHL; i++) done[i] = acc[i];
      }
      pending = pass;
#pragma unroll
      for (int i = 0; i < CHL; i++) acc[i] = 0.f;
    }
  }
  if (pending >= 0) {
    __syncthreads();
    finish_pass(pending, done);
  }
}

// z * e^d for d <= 0, with 0 * e^(-inf) = 0 (an empty partial carries (-inf, 0))
__device__ __forceinline__ float wexp(float z, float d) { return d > -INFINITY ? z * exp_neg(d) : 0.f; }

// out[h][c] = sum_tile slab[tile][h][c] * e^(m_tile - M) / Z  (+ the fp16 sink tokens' share), M / Z merged from the tiles'
// (m, l) and the sink scores; one block per 16 channels, 64 slab lanes per channel (the layout of mix_v_reduce_kernel)
__global__ __launch_bounds__(1024) void fused_merge_kernel(const float *__restrict__ slabs, const float *__restrict__ stats,
                                                           int n_tiles, int H, const __half *__restrict__ sink,
                                                           __half *__restrict__ sink_probs, int n_sink,
                                                           const __half *__restrict__ v_sink, float *__restrict__ out) {
  __shared__ float red[64][17];
  __shared__ float rm[16], rz[16];
  const int C = H * kHeadDim;
  const int tid = threadIdx.x, cl = tid & 15, rg = tid >> 4;
  const int c = blockIdx.x * 16 + cl;
  const int h = (blockIdx.x * 16) / kHeadDim;
  const float2 *st = reinterpret_cast<const float2 *>(stats) + (int64_t)h * n_tiles;
  // row statistics of the head: every lane merges a strided share, then the block
  float M = -INFINITY, Z = 0.f;
  for (int i = tid; i < n_tiles; i += 1024) {
    const float2 t = st[i];
    if (t.x > -INFINITY) {
      const float mn = fmaxf(M, t.x);
      Z = wexp(Z, M - mn) + t.y * exp_neg(t.x - mn);
      M = mn;
    }
  }
  for (int i = tid; i < n_sink; i += 1024) {
    const float x = __half2float(sink[h * n_sink + i]);
    const float mn = fmaxf(M, x);
    Z = wexp(Z, M - mn) + exp_neg(x - mn);
    M = mn;
  }
#pragma unroll
  for (int d = 32; d >= 1; d >>= 1) {
    const float mo = __shfl_xor(M, d), zo = __shfl_xor(Z, d);
    const float mn = fmaxf(M, mo);
    Z = (mn == -INFINITY) ? 0.f : wexp(Z, M - mn) + wexp(zo, mo - mn);
    M = mn;
  }
  if ((tid & 63) == 0) { rm[tid >> 6] = M; rz[tid >> 6] = Z; }
  __syncthreads();
  float Mb = -INFINITY;
#pragma unroll
  for (int i = 0; i < 16; i++) Mb = fmaxf(Mb, rm[i]);
  float Zb = 0.f;
#pragma unroll
  for (int i = 0; i < 16; i++) if (rm[i] > -INFINITY) Zb += rz[i] * exp_neg(rm[i] - Mb);
  const float rZ = 1.0f / Zb;
  // weighted slab sum: 8 loads in flight per lane
  float s = 0.f;
  if (c < C) {
    const float *src = slabs + c;
    int r = rg;
    for (; r < n_tiles; r += 8 * 64) {
      float v[8], w[8];
#pragma unroll
      for (int k = 0; k < 8; k++) {
        const int rr = r + 64 * k;
        const bool in = rr < n_tiles;
        v[k] = in ? src[(int64_t)rr * C] : 0.f;
        w[k] = in ? st[rr].x : -INFINITY;
      }
#pragma unroll
      for (int k = 0; k < 8; k++) s += (w[k] > -INFINITY) ? v[k] * exp_neg(w[k] - Mb) : 0.f;
    }
  }
  red[rg][cl] = s;
  __syncthreads();
  if (rg == 0 && c < C) {
    float t = 0.f;
#pragma unroll
    for (int k = 0; k < 64; k++) t += red[k][cl];
    t *= rZ;
    if (v_sink != nullptr && n_sink > 0) t += sink_output(sink, v_sink, n_sink, h, c % kHeadDim, Mb, rZ);
    out[c] = t;
  }
  if (sink_probs != nullptr && (blockIdx.x * 16) % kHeadDim == 0)
    for (int i = tid; i < n_sink; i += 1024)
      sink_probs[h * n_sink + i] = __float2half_rn(prob_fp16(__half2float(sink[h * n_sink + i]), Mb, rZ));
}

template <int BITS>
static int launch(FusedArgs f, const __half *sink, __half *sink_probs, int n_sink, const __half *v_sink, float *out,
                  hipStream_t st) {
  const int H = f.k.H;
  const int64_t L = f.k.L;
  const int64_t full = L / T;
  const int rem = (int)(L % T);
#ifndef KVQ_FUSED_HPG_TAIL
#define KVQ_FUSED_HPG_TAIL 4     // the ragged last tile: head groups of this size (short workgroups that start in freed slots)
#endif
  f.k.groups = 1;
  f.k.hpg = H;
  f.k.full_blocks = (int)full;
  f.k.hpg_tail = full ? KVQ_FUSED_HPG_TAIL : H / (H >= 8 ? 8 : 1);
  if (f.k.hpg_tail > H) f.k.hpg_tail = H;
  if (f.k.hpg_tail < 1) f.k.hpg_tail = 1;
  const int tail_blocks = rem ? (H + f.k.hpg_tail - 1) / f.k.hpg_tail : 0;
  f.n_tiles = (int)full + (rem ? 1 : 0);
  fused_decode_kernel<BITS><<<dim3((unsigned)(full + tail_blocks)), dim3(NT), 0, st>>>(f);
  int rc = check_launch();
  if (rc) return rc;
  kvq_step_mark_fused(st);       // (measurement hook of kvq_decode_step)
  const int C = H * kHeadDim;
  fused_merge_kernel<<<dim3((C + 15) / 16), dim3(1024), 0, st>>>(f.slabs, f.stats, f.n_tiles, H, sink, sink_probs, n_sink, v_sink, out);
  return check_launch();
}

}  // namespace fused
}  // namespace kvq

using namespace kvq;

extern "C" {

int kvq_fused_attend_supported(int bits, int H, int hd, int64_t L, int64_t max_len, int n_out) {
  return bits >= 2 && bits <= 4 && hd == kHeadDim && H > 0 && H <= kSparseHpg && L > 0 && L <= max_len && max_len % 4 == 0 &&
         n_out > 0 && n_out * fused::T <= 24 * fused::NT && (int64_t)H * (hd / 32 * bits) * max_len < (1ll << 31) &&
         (int64_t)n_out * max_len * 4 < (1ll << 32);
}

size_t kvq_fused_attend_workspace_bytes(int bits, int H, int hd, int64_t L) {
  if (bits < 2 || bits > 4 || H <= 0 || hd != kHeadDim || L <= 0) return 0;
  const size_t n_tiles = (size_t)((L + fused::T - 1) / fused::T);
  return ((n_tiles * H * hd * sizeof(float) + 255) & ~(size_t)255) + n_tiles * H * 2 * sizeof(float);
}

int kvq_fused_attend(int bits, const int32_t *kmat, const float *klut, const void *score_workspace, const int32_t *vmat,
                     const float *vlut_rows, int H, int hd, int64_t L, int64_t max_len, float rope_theta, int pos_offset,
                     const float *koutliers_t, const int32_t *kidx_t, const float *voutliers, const int32_t *vidx, int n_out,
                     float inv_sqrt_hd, const uint16_t *sink_scores, uint16_t *sink_probs, int n_sink,
                     const uint16_t *v_sink, float *out, void *workspace, size_t workspace_bytes, void *stream) {
  if (!kmat || !klut || !score_workspace || !vmat || !vlut_rows || !koutliers_t || !kidx_t || !voutliers || !vidx || !out)
    return KVQ_EINVAL;
  if (!kvq_fused_attend_supported(bits, H, hd, L, max_len, n_out)) return KVQ_EINVAL;
  if (n_sink < 0 || (n_sink > 0 && (!sink_scores || !sink_probs))) return KVQ_EINVAL;
  if (v_sink && n_sink <= 0) return KVQ_EINVAL;
  if (!workspace || workspace_bytes < kvq_fused_attend_workspace_bytes(bits, H, hd, L) ||
      reinterpret_cast<uintptr_t>(workspace) % 16 || reinterpret_cast<uintptr_t>(score_workspace) % 16 ||
      (reinterpret_cast<uintptr_t>(vmat) | reinterpret_cast<uintptr_t>(vlut_rows)) % 16)
    return KVQ_EWORKSPACE;
  const size_t tab_b = bits == 4 ? KTab<4>::BUF_B : (bits == 3 ? KTab<3>::BUF_B : KTab<2>::BUF_B);
  fused::FusedArgs f;
  ScoreKArgs &a = f.k;
  a.tab = reinterpret_cast<const unsigned char *>(score_workspace);
  a.q = reinterpret_cast<const float *>(a.tab + (size_t)H * tab_b);      // fp32 copy of q behind the tables (kvq_ktab.h)
  a.mat = reinterpret_cast<const uint32_t *>(kmat);
  a.mul = nullptr;
  a.outliers = nullptr;
  a.idx = nullptr;
  a.out_t = koutliers_t;
  a.idx_t = kidx_t;
  a.H = H;
  a.hpg = H;
  a.groups = 1;
  a.full_blocks = 0;
  a.hpg_tail = H;
  a.L = L;
  a.max_len = max_len;
  a.pos_offset = pos_offset;
  a.n_out = n_out;
  a.n_out_magic = (uint32_t)(((1ull << 32) + (uint64_t)n_out - 1) / (uint64_t)n_out);
  a.accumulate = 0;
  a.sm_parts = nullptr;
  a.sm_inv = inv_sqrt_hd;
  a.sm_nparts = 0;
  a.rope_theta = rope_theta;
  f.vmat = reinterpret_cast<const uint32_t *>(vmat);
  f.vrows = vlut_rows;
  f.vout = voutliers;
  f.vidx = vidx;
  const size_t n_tiles = (size_t)((L + fused::T - 1) / fused::T);
  f.slabs = reinterpret_cast<float *>(workspace);
  f.stats = reinterpret_cast<float *>(reinterpret_cast<unsigned char *>(workspace) +
                                      ((n_tiles * H * hd * sizeof(float) + 255) & ~(size_t)255));
  f.n_tiles = (int)n_tiles;
  f.inv = inv_sqrt_hd;
  hipStream_t st = (hipStream_t)stream;
  const __half *sk = reinterpret_cast<const __half *>(sink_scores);
  __half *sp = reinterpret_cast<__half *>(sink_probs);
  const __half *vs = reinterpret_cast<const __half *>(v_sink);
  (void)klut;
  switch (bits) {
    case 4: return fused::launch<4>(f, sk, sp, n_sink, vs, out, st);
    case 3: return fused::launch<3>(f, sk, sp, n_sink, vs, out, st);
    default: return fused::launch<2>(f, sk, sp, n_sink, vs, out, st);
  }
}

}  // extern "C"
