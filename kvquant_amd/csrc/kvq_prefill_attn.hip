// Causal self-attention of the prompt (prefill) on the matrix cores: the one true dense contraction of the
// hot path (BASELINE config 4).  The reference calls flash-attn here (ML:1861-1874 -> 2013-2070, third party);
// this is a flash-style kernel written for gfx950: fp16 inputs, fp32 accumulation on
// v_mfma_f32_32x32x16_f16, online softmax, no S x S matrix in memory.
//
// Decomposition: workgroup = 4 waves = 128 query rows of one head; wave = 32 query rows (its Q fragments live
// in registers for the whole kernel); keys / values arrive in 64-key tiles shared by the four waves through LDS
// (double buffered, one barrier per tile).  Both products are computed TRANSPOSED so that a lane owns ONE
// query row end to end (column = lane & 31 of every accumulator):
//   S^T[key][q] = K[key][:] . Q[q][:]      A = K tile (LDS, 16-byte rows chunks XOR-swizzled by the key: conflict
//                                          free ds_read_b128), B = Q (registers)
//   O^T[d][q]  += V^T[d][key] . P^T[key][q]  A = V^T tile (LDS, transposed while staging, row stride 136 B:
//                                          conflict free ds_read_b64), B = P packed to fp16 IN PLACE: the MFMA
//                                          summation index is just a label, so the keys are fed in the order the
//                                          S^T accumulator holds them (rows (r&3) + 8(r>>2) + 4(lane>>5)) and the
//                                          V^T fragment is read in the same order -- no cross-lane exchange.
// Row max / sum: in-lane over the lane's 32 scores of the tile + one exchange with lane ^ 32.
// Causal mask only on the tiles that cross the diagonal; heavy (late) query blocks are scheduled first.
#include "kvq_common.h"
#include "kvq_host.h"

namespace kvq {

typedef _Float16 f16;
typedef f16 f16x8 __attribute__((ext_vector_type(8)));
typedef f16 f16x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int kD = 128;          // head_dim
constexpr int kQB = 32;          // query rows per wave
#ifndef KVQ_ATTN_WAVES
#define KVQ_ATTN_WAVES 8
#endif
constexpr int kAW = KVQ_ATTN_WAVES;           // waves per workgroup
constexpr int kQW = kQB * kAW;   // query rows per workgroup
constexpr int kKV = 64;          // keys per tile
constexpr int kKRow = kD * 2;    // bytes of a K row in LDS
constexpr int kVRow = 136;       // bytes of a V^T row in LDS (64 keys x 2 B + 8 pad)
constexpr int kKTile = kKV * kKRow;      // 16384
constexpr int kVTile = kD * kVRow;       // 17408
constexpr int kStage = kKTile + kVTile;  // 33792

struct AttnArgs {
  const f16 *q, *k, *v;   // element [h][s][d] at h * stride_h + s * stride_s + d
  f16 *o;
  int64_t qsh, qss, ksh, kss, vsh, vss, osh, oss;
  int S, H;
  float scale_log2e;      // softmax scale * log2(e)
};

// two floats -> packed fp16, ROUND TO NEAREST (gfx950's v_cvt_pk_f16_f32; cvt_pkrtz truncates, which biases P -- and
// with it every output -- down by 2^-12 .. 2^-11 relative to the unrounded row sum it is normalised with)
__device__ __forceinline__ uint32_t pack_f16(float a, float b) {
  uint32_t r;
  asm("v_cvt_pk_f16_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
  return r;
}

__global__ __launch_bounds__(kAW * 64, kAW == 8 ? 2 : 2) void prefill_attn_kernel(AttnArgs a) {
  __shared__ __attribute__((aligned(16))) unsigned char smem[2 * kStage];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int hi = lane >> 5, ql = lane & 31;
  // heavy (late, long-row) query blocks of ALL heads first: the grid is walked in dispatch order
  const int nqb = (a.S + kQW - 1) / kQW;
  const int h = (int)blockIdx.x % a.H;
  const int qb = nqb - 1 - (int)blockIdx.x / a.H;
  const int q0w = qb * kQW;                            // first query row of the workgroup
  const int q0 = q0w + wave * kQB;                     // ... of this wave
  const int qrow = q0 + ql;                            // this lane's query row
  const int qrc = qrow < a.S ? qrow : a.S - 1;
  const f16 *kh = a.k + (int64_t)h * a.ksh;
  const f16 *vh = a.v + (int64_t)h * a.vsh;

  // ---- Q fragments (B operand of S^T): lane (q, hi) holds Q[q][ks*16 + hi*8 .. +7], ks = 0..7
  f16x8 qf[8];
  {
    const f16 *qp = a.q + (int64_t)h * a.qsh + (int64_t)qrc * a.qss + hi * 8;
#pragma unroll
    for (int ks = 0; ks < 8; ks++) qf[ks] = *reinterpret_cast<const f16x8 *>(qp + ks * 16);
  }

  f32x16 oacc[4];
#pragma unroll
  for (int i = 0; i < 4; i++)
#pragma unroll
    for (int r = 0; r < 16; r++) oacc[i][r] = 0.f;
  float m_run = -INFINITY, l_run = 0.f;

  const int kv_end = (q0w + kQW < a.S ? q0w + kQW : a.S);     // keys [0, kv_end) matter to this workgroup
  const int n_tiles = (kv_end + kKV - 1) / kKV;

  // ---- staging: thread -> (key = tid & 63, 16-byte chunks c = (tid >> 6) + kAW j, j < 16 / kAW) of K and V
  constexpr int NCH = 16 / kAW;
  const int skey = tid & 63, sc0 = tid >> 6;
  typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));     // (a native vector: HIP's uint4 class ends up on the stack here)
  u32x4 kreg[NCH], vreg[NCH];
  auto load_tile = [&](int t) __attribute__((always_inline)) {
    int key = t * kKV + skey;
    if (key >= a.S) key = a.S - 1;
    const f16 *kp = kh + (int64_t)key * a.kss;
    const f16 *vp = vh + (int64_t)key * a.vss;
#pragma unroll
    for (int j = 0; j < NCH; j++) {
      kreg[j] = *reinterpret_cast<const u32x4 *>(kp + (sc0 + kAW * j) * 8);
      vreg[j] = *reinterpret_cast<const u32x4 *>(vp + (sc0 + kAW * j) * 8);
    }
  };
  auto store_tile = [&](int stage) __attribute__((always_inline)) {
    unsigned char *ks = smem + stage * kStage;
    unsigned char *vs = ks + kKTile;
#pragma unroll
    for (int j = 0; j < NCH; j++) {
      const int c = sc0 + kAW * j;                                      // 16-byte chunk (8 d values)
      *reinterpret_cast<u32x4 *>(ks + skey * kKRow + ((c ^ (skey & 15)) << 4)) = kreg[j];
      // (the eight halves cut out of the four words: no pointer into the staging registers)
      const uint32_t wv[4] = {vreg[j].x, vreg[j].y, vreg[j].z, vreg[j].w};
#pragma unroll
      for (int i = 0; i < 8; i++)
        *reinterpret_cast<unsigned short *>(vs + (c * 8 + i) * kVRow + skey * 2) = (unsigned short)(wv[i >> 1] >> (16 * (i & 1)));
    }
  };

  load_tile(0);
  store_tile(0);
  __syncthreads();

  for (int t = 0; t < n_tiles; t++) {
    const int stage = t & 1;
    if (t + 1 < n_tiles) load_tile(t + 1);               // in flight during the math
    const int key0 = t * kKV;
    // a wave whose rows all lie before this tile has nothing to add (causal); it still takes part in the staging
    const bool active = key0 <= q0 + kQB - 1;
    if (active) {
      const unsigned char *ks = smem + stage * kStage;
      const unsigned char *vs = ks + kKTile;
      // ---- S^T = K . Q^T: two 32-key blocks
      f32x16 sacc[2];
#pragma unroll
      for (int kb = 0; kb < 2; kb++) {
#pragma unroll
        for (int r = 0; r < 16; r++) sacc[kb][r] = 0.f;
        const int key = kb * 32 + ql;                      // A operand row of this lane
#pragma unroll
        for (int kk = 0; kk < 8; kk++) {
          const int c = kk * 2 + hi;                       // 16-byte chunk: d = kk*16 + hi*8 .. +7
          const f16x8 kf = *reinterpret_cast<const f16x8 *>(ks + key * kKRow + ((c ^ (key & 15)) << 4));
          sacc[kb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(kf, qf[kk], sacc[kb], 0, 0, 0);
        }
      }
      // ---- mask + scale + online softmax.  Accumulator reg r of block kb: key = key0 + kb*32 + (r&3) + 8(r>>2) + 4hi
      const bool diag = key0 + kKV - 1 > q0 || key0 + kKV > a.S;       // tile crosses the diagonal / the end
      // (running max m_run is kept in RAW score units; the softmax scale (> 0) and log2(e) are folded into one FMA
      //  per exponential: p = 2^(s*c - m*c))
      float mx = -INFINITY;
#pragma unroll
      for (int kb = 0; kb < 2; kb++)
#pragma unroll
        for (int r = 0; r < 16; r++) {
          float s = sacc[kb][r];
          if (diag) {
            const int key = key0 + kb * 32 + (r & 3) + 8 * (r >> 2) + 4 * hi;
            if (key > qrow || key >= a.S) s = -INFINITY;
            sacc[kb][r] = s;
          }
          mx = fmaxf(mx, s);
        }
      mx = fmaxf(mx, __shfl_xor(mx, 32));
      const float m_new = fmaxf(m_run, mx);
      const float m_use = m_new == -INFINITY ? 0.f : m_new;            // (a fully masked row so far)
      const float mc = -m_use * a.scale_log2e;
      float ps = 0.f;
      uint32_t pp[2][8];                                                // P^T packed: regs (2i, 2i+1)
#pragma unroll
      for (int kb = 0; kb < 2; kb++)
#pragma unroll
        for (int r = 0; r < 16; r += 2) {
          const float p0 = __builtin_amdgcn_exp2f(fmaf(sacc[kb][r], a.scale_log2e, mc));
          const float p1 = __builtin_amdgcn_exp2f(fmaf(sacc[kb][r + 1], a.scale_log2e, mc));
          ps += p0 + p1;
          pp[kb][r >> 1] = pack_f16(p0, p1);
        }
      ps += __shfl_xor(ps, 32);
      // the running sums only need rescaling when some row's maximum moved (wave-uniform test: rare after the first
      // tiles on real data); 2^-inf = 0 on the first tile
      if (__any(m_new != m_run)) {
        const float alpha = __builtin_amdgcn_exp2f((m_run - m_use) * a.scale_log2e);
        l_run *= alpha;
#pragma unroll
        for (int i = 0; i < 4; i++)
#pragma unroll
          for (int r = 0; r < 16; r++) oacc[i][r] *= alpha;
      }
      l_run += ps;
      m_run = m_new;
      // ---- O^T += V^T . P^T.  k index (hi*8 + i) of step (kb, ks) stands for key kb*32 + ks*16 + (i>>2)*8 + 4hi + (i&3)
#pragma unroll
      for (int kb = 0; kb < 2; kb++)
#pragma unroll
        for (int ks2 = 0; ks2 < 2; ks2++) {
          typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
          const u32x4 pw = {pp[kb][ks2 * 4], pp[kb][ks2 * 4 + 1], pp[kb][ks2 * 4 + 2], pp[kb][ks2 * 4 + 3]};
          const f16x8 pf = __builtin_bit_cast(f16x8, pw);
          const int kofs = (kb * 32 + ks2 * 16 + 4 * hi) * 2;          // byte offset of the first 4 keys in a V^T row
#pragma unroll
          for (int db = 0; db < 4; db++) {
            const unsigned char *row = vs + (db * 32 + ql) * kVRow + kofs;
            const f16x4 v0 = *reinterpret_cast<const f16x4 *>(row);
            const f16x4 v1 = *reinterpret_cast<const f16x4 *>(row + 16);
            const f16x8 vf = {v0[0], v0[1], v0[2], v0[3], v1[0], v1[1], v1[2], v1[3]};
            oacc[db] = __builtin_amdgcn_mfma_f32_32x32x16_f16(vf, pf, oacc[db], 0, 0, 0);
          }
        }
    }
    if (t + 1 < n_tiles) store_tile(1 - stage);     // the other stage: free since the barrier of the previous tile
    __syncthreads();
  }

  // ---- normalise and write O[q][d] (fp16): accumulator reg r of block db is d = db*32 + (r&3) + 8(r>>2) + 4hi
  if (qrow < a.S) {
    const float inv = l_run > 0.f ? 1.0f / l_run : 0.f;
    f16 *op = a.o + (int64_t)h * a.osh + (int64_t)qrow * a.oss;
#pragma unroll
    for (int db = 0; db < 4; db++)
#pragma unroll
      for (int g = 0; g < 4; g++) {
        const int d = db * 32 + 8 * g + 4 * hi;
        uint2 w;
        w.x = pack_f16(oacc[db][4 * g] * inv, oacc[db][4 * g + 1] * inv);
        w.y = pack_f16(oacc[db][4 * g + 2] * inv, oacc[db][4 * g + 3] * inv);
        *reinterpret_cast<uint2 *>(op + d) = w;
      }
  }
}

}  // namespace kvq

using namespace kvq;

extern "C" {

int kvq_prefill_attention(const void *q, const void *k, const void *v, void *out, int H, int S, int hd,
                          int64_t q_stride_h, int64_t q_stride_s, int64_t k_stride_h, int64_t k_stride_s,
                          int64_t v_stride_h, int64_t v_stride_s, int64_t o_stride_h, int64_t o_stride_s,
                          float softmax_scale, void *stream) {
  if (!q || !k || !v || !out || H <= 0 || S <= 0 || hd != kD) return KVQ_EINVAL;
  // 16-byte row segments are loaded / stored as vectors
  if ((q_stride_h | q_stride_s | k_stride_h | k_stride_s | v_stride_h | v_stride_s) % 8 || (o_stride_h | o_stride_s) % 4 ||
      (reinterpret_cast<uintptr_t>(q) | reinterpret_cast<uintptr_t>(k) | reinterpret_cast<uintptr_t>(v)) % 16 ||
      reinterpret_cast<uintptr_t>(out) % 8)
    return KVQ_EINVAL;
  AttnArgs a;
  a.q = reinterpret_cast<const f16 *>(q);
  a.k = reinterpret_cast<const f16 *>(k);
  a.v = reinterpret_cast<const f16 *>(v);
  a.o = reinterpret_cast<f16 *>(out);
  a.qsh = q_stride_h; a.qss = q_stride_s; a.ksh = k_stride_h; a.kss = k_stride_s;
  a.vsh = v_stride_h; a.vss = v_stride_s; a.osh = o_stride_h; a.oss = o_stride_s;
  a.S = S;
  a.H = H;
  a.scale_log2e = softmax_scale * 1.4426950408889634f;
  dim3 grid((unsigned)((S + kQW - 1) / kQW) * (unsigned)H), block(kAW * 64);
  prefill_attn_kernel<<<grid, block, 0, (hipStream_t)stream>>>(a);
  return check_launch();
}

}  // extern "C"
