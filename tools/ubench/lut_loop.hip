// microbenchmark 4 (development tool): the p.V inner loop at the granularity the kernel runs it -- one iteration =
// one 16-byte quad of a lane's packed row (4 tokens x 8 nibbles): ds_read_b128 of the tile word, per token the
// nibble -> byte-offset preparation (4 VALU), 8 extractions, 8 ds_read_b32 look-ups into that token's table, 8 FMAs.
// Variants differ only in how the look-ups are batched / software-pipelined.  Reports ns per code-step per SIMD.
//   MODE 0: per token: prep, extract, 8 reads, wait(0), 8 fmac                       (8 in flight)
//   MODE 1: per 2 tokens: prep x2, extract x2, 16 reads... (lgkmcnt saturates at 15) (16 in flight)
//   MODE 2: pipelined: token t+1's prep/extract/reads are issued before token t's FMAs (wait lgkmcnt(8))
//   MODE 3: as 2 + the probabilities of the quad from LDS (ds_read_b128, broadcast) instead of a register
//   MODE 4: as 2, the quad as one asm block with the look-up results in pinned register pairs and v_pk_fma_f32
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>

#define PREP(we, wo, w, slot)                                                                                  \
  asm volatile("v_lshlrev_b32 %0, 2, %2\n v_lshrrev_b32 %1, 2, %2\n v_and_or_b32 %0, %0, %4, %3\n v_and_or_b32 %1, %1, %4, %3" \
               : "=&v"(we), "=&v"(wo) : "v"(w), "v"(slot), "s"(0x3C3C3C3Cu))
#define EXTRACT(u, we, wo)                                                                                      \
  asm volatile("v_and_b32 %0, 0xff, %8\n v_and_b32 %1, 0xff, %9\n v_bfe_u32 %2, %8, 8, 8\n v_bfe_u32 %3, %9, 8, 8\n" \
               "v_bfe_u32 %4, %8, 16, 8\n v_bfe_u32 %5, %9, 16, 8\n v_lshrrev_b32 %6, 24, %8\n v_lshrrev_b32 %7, 24, %9\n" \
               : "=&v"(u[0]), "=&v"(u[1]), "=&v"(u[2]), "=&v"(u[3]), "=&v"(u[4]), "=&v"(u[5]), "=&v"(u[6]), "=&v"(u[7]) \
               : "v"(we), "v"(wo))
#define READ8(v, u, OFF)                                                                                         \
  asm volatile("ds_read_b32 %0, %8 offset:" #OFF "\n ds_read_b32 %1, %9 offset:" #OFF "\n ds_read_b32 %2, %10 offset:" #OFF "\n" \
               "ds_read_b32 %3, %11 offset:" #OFF "\n ds_read_b32 %4, %12 offset:" #OFF "\n ds_read_b32 %5, %13 offset:" #OFF "\n" \
               "ds_read_b32 %6, %14 offset:" #OFF "\n ds_read_b32 %7, %15 offset:" #OFF "\n"                      \
               : "=&v"(v[0]), "=&v"(v[1]), "=&v"(v[2]), "=&v"(v[3]), "=&v"(v[4]), "=&v"(v[5]), "=&v"(v[6]), "=&v"(v[7]) \
               : "v"(u[0]), "v"(u[1]), "v"(u[2]), "v"(u[3]), "v"(u[4]), "v"(u[5]), "v"(u[6]), "v"(u[7]) : "memory")
#define FMA8(a, v, p)                                                                                            \
  asm volatile("v_fmac_f32 %0, %8, %16\n v_fmac_f32 %1, %9, %16\n v_fmac_f32 %2, %10, %16\n v_fmac_f32 %3, %11, %16\n" \
               "v_fmac_f32 %4, %12, %16\n v_fmac_f32 %5, %13, %16\n v_fmac_f32 %6, %14, %16\n v_fmac_f32 %7, %15, %16\n" \
               : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(a[4]), "+v"(a[5]), "+v"(a[6]), "+v"(a[7])       \
               : "v"(v[0]), "v"(v[1]), "v"(v[2]), "v"(v[3]), "v"(v[4]), "v"(v[5]), "v"(v[6]), "v"(v[7]), "v"(p))
typedef float f32x2 __attribute__((ext_vector_type(2)));
#define QUAD_ASM "s_waitcnt lgkmcnt(0)\n\tv_lshlrev_b32 v100, 2, %[wx]\n\tv_lshrrev_b32 v101, 2, %[wx]\n\tv_and_or_b32 v100, v100, %[mask], %[slot]\n\tv_and_or_b32 v101, v101, %[mask], %[slot]\n\tv_and_b32 v102, 0xff, v100\n\tv_and_b32 v103, 0xff, v101\n\tv_bfe_u32 v104, v100, 8, 8\n\tv_bfe_u32 v105, v101, 8, 8\n\tv_bfe_u32 v106, v100, 16, 8\n\tv_bfe_u32 v107, v101, 16, 8\n\tv_lshrrev_b32 v108, 24, v100\n\tv_lshrrev_b32 v109, 24, v101\n\tds_read_b32 v110, v102 offset:0\n\tds_read_b32 v111, v103 offset:0\n\tds_read_b32 v112, v104 offset:0\n\tds_read_b32 v113, v105 offset:0\n\tds_read_b32 v114, v106 offset:0\n\tds_read_b32 v115, v107 offset:0\n\tds_read_b32 v116, v108 offset:0\n\tds_read_b32 v117, v109 offset:0\n\tv_lshlrev_b32 v100, 2, %[wy]\n\tv_lshrrev_b32 v101, 2, %[wy]\n\tv_and_or_b32 v100, v100, %[mask], %[slot]\n\tv_and_or_b32 v101, v101, %[mask], %[slot]\n\tv_and_b32 v102, 0xff, v100\n\tv_and_b32 v103, 0xff, v101\n\tv_bfe_u32 v104, v100, 8, 8\n\tv_bfe_u32 v105, v101, 8, 8\n\tv_bfe_u32 v106, v100, 16, 8\n\tv_bfe_u32 v107, v101, 16, 8\n\tv_lshrrev_b32 v108, 24, v100\n\tv_lshrrev_b32 v109, 24, v101\n\tds_read_b32 v118, v102 offset:128\n\tds_read_b32 v119, v103 offset:128\n\tds_read_b32 v120, v104 offset:128\n\tds_read_b32 v121, v105 offset:128\n\tds_read_b32 v122, v106 offset:128\n\tds_read_b32 v123, v107 offset:128\n\tds_read_b32 v124, v108 offset:128\n\tds_read_b32 v125, v109 offset:128\n\ts_waitcnt lgkmcnt(8)\n\tv_pk_fma_f32 %[a0], v[110:111], %[pxy], %[a0] op_sel_hi:[1,0,1]\n\tv_pk_fma_f32 %[a1], v[112:113], %[pxy], %[a1] op_sel_hi:[1,0,1]\n\tv_pk_fma_f32 %[a2], v[114:115], %[pxy], %[a2] op_sel_hi:[1,0,1]\n\tv_pk_fma_f32 %[a3], v[116:117], %[pxy], %[a3] op_sel_hi:[1,0,1]\n\tv_lshlrev_b32 v100, 2, %[wz]\n\tv_lshrrev_b32 v101, 2, %[wz]\n\tv_and_or_b32 v100, v100, %[mask], %[slot]\n\tv_and_or_b32 v101, v101, %[mask], %[slot]\n\tv_and_b32 v102, 0xff, v100\n\tv_and_b32 v103, 0xff, v101\n\tv_bfe_u32 v104, v100, 8, 8\n\tv_bfe_u32 v105, v101, 8, 8\n\tv_bfe_u32 v106, v100, 16, 8\n\tv_bfe_u32 v107, v101, 16, 8\n\tv_lshrrev_b32 v108, 24, v100\n\tv_lshrrev_b32 v109, 24, v101\n\tds_read_b32 v110, v102 offset:256\n\tds_read_b32 v111, v103 offset:256\n\tds_read_b32 v112, v104 offset:256\n\tds_read_b32 v113, v105 offset:256\n\tds_read_b32 v114, v106 offset:256\n\tds_read_b32 v115, v107 offset:256\n\tds_read_b32 v116, v108 offset:256\n\tds_read_b32 v117, v109 offset:256\n\ts_waitcnt lgkmcnt(8)\n\tv_pk_fma_f32 %[a0], v[118:119], %[pxy], %[a0] op_sel:[0,1,0]\n\tv_pk_fma_f32 %[a1], v[120:121], %[pxy], %[a1] op_sel:[0,1,0]\n\tv_pk_fma_f32 %[a2], v[122:123], %[pxy], %[a2] op_sel:[0,1,0]\n\tv_pk_fma_f32 %[a3], v[124:125], %[pxy], %[a3] op_sel:[0,1,0]\n\tv_lshlrev_b32 v100, 2, %[ww]\n\tv_lshrrev_b32 v101, 2, %[ww]\n\tv_and_or_b32 v100, v100, %[mask], %[slot]\n\tv_and_or_b32 v101, v101, %[mask], %[slot]\n\tv_and_b32 v102, 0xff, v100\n\tv_and_b32 v103, 0xff, v101\n\tv_bfe_u32 v104, v100, 8, 8\n\tv_bfe_u32 v105, v101, 8, 8\n\tv_bfe_u32 v106, v100, 16, 8\n\tv_bfe_u32 v107, v101, 16, 8\n\tv_lshrrev_b32 v108, 24, v100\n\tv_lshrrev_b32 v109, 24, v101\n\tds_read_b32 v118, v102 offset:384\n\tds_read_b32 v119, v103 offset:384\n\tds_read_b32 v120, v104 offset:384\n\tds_read_b32 v121, v105 offset:384\n\tds_read_b32 v122, v106 offset:384\n\tds_read_b32 v123, v107 offset:384\n\tds_read_b32 v124, v108 offset:384\n\tds_read_b32 v125, v109 offset:384\n\ts_waitcnt lgkmcnt(8)\n\tv_pk_fma_f32 %[a0], v[110:111], %[pzw], %[a0] op_sel_hi:[1,0,1]\n\tv_pk_fma_f32 %[a1], v[112:113], %[pzw], %[a1] op_sel_hi:[1,0,1]\n\tv_pk_fma_f32 %[a2], v[114:115], %[pzw], %[a2] op_sel_hi:[1,0,1]\n\tv_pk_fma_f32 %[a3], v[116:117], %[pzw], %[a3] op_sel_hi:[1,0,1]\n\ts_waitcnt lgkmcnt(0)\n\tv_pk_fma_f32 %[a0], v[118:119], %[pzw], %[a0] op_sel:[0,1,0]\n\tv_pk_fma_f32 %[a1], v[120:121], %[pzw], %[a1] op_sel:[0,1,0]\n\tv_pk_fma_f32 %[a2], v[122:123], %[pzw], %[a2] op_sel:[0,1,0]\n\tv_pk_fma_f32 %[a3], v[124:125], %[pzw], %[a3] op_sel:[0,1,0]"
#define WAIT(N) asm volatile("s_waitcnt lgkmcnt(" #N ")" ::: "memory")

template <int MODE>
__global__ __launch_bounds__(512, 2) void k(float *out, int iters, unsigned seed) {
  __shared__ __attribute__((aligned(16))) unsigned tile[8192];   // 32 KB "tile": a lane reads its own 16-byte quads
  __shared__ __attribute__((aligned(16))) float tab[1024];       // tables of the chunk's tokens (64 B each)
  __shared__ __attribute__((aligned(16))) float pbuf[512];
  for (int i = threadIdx.x; i < 8192; i += 512) tile[i] = (i * 2654435761u) ^ seed;
  for (int i = threadIdx.x; i < 1024; i += 512) tab[i] = (float)i * 0.001f;
  if (threadIdx.x < 512) pbuf[threadIdx.x] = 0.001f * threadIdx.x;
  __syncthreads();
  float a[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  f32x2 a2[4] = {{0, 0}, {0, 0}, {0, 0}, {0, 0}};
  const unsigned slot = ((threadIdx.x >> 8) & 1) * 0x40404040u;
  const unsigned tbase = (unsigned)(uintptr_t)tile + (threadIdx.x & 255) * 128;   // 8 quads per lane
  const unsigned pbase = (unsigned)(uintptr_t)pbuf + ((threadIdx.x >> 4) & 15) * 128;
  float pt = 0.37f;
  asm volatile("" : "+v"(pt));
  for (int it = 0; it < iters; it++) {
    uint4 w;
    float4 p4 = make_float4(pt, pt, pt, pt);
    const unsigned qa = tbase + ((it & 7) << 4);
    asm volatile("ds_read_b128 %0, %1" : "=v"(w) : "v"(qa) : "memory");
    if (MODE == 3) {
      const unsigned pa = pbase + ((it & 7) << 4);
      asm volatile("ds_read_b128 %0, %1" : "=v"(p4) : "v"(pa) : "memory");
    }
    WAIT(0);
    unsigned we, wo, u[8], u2[8];
    float v[8], v2[8];
    if (MODE == 0) {
      PREP(we, wo, w.x, slot); EXTRACT(u, we, wo); READ8(v, u, 0);   WAIT(0); FMA8(a, v, p4.x);
      PREP(we, wo, w.y, slot); EXTRACT(u, we, wo); READ8(v, u, 128); WAIT(0); FMA8(a, v, p4.y);
      PREP(we, wo, w.z, slot); EXTRACT(u, we, wo); READ8(v, u, 256); WAIT(0); FMA8(a, v, p4.z);
      PREP(we, wo, w.w, slot); EXTRACT(u, we, wo); READ8(v, u, 384); WAIT(0); FMA8(a, v, p4.w);
    } else if (MODE == 1) {
      PREP(we, wo, w.x, slot); EXTRACT(u, we, wo); READ8(v, u, 0);
      PREP(we, wo, w.y, slot); EXTRACT(u2, we, wo); READ8(v2, u2, 128);
      WAIT(8); FMA8(a, v, p4.x); WAIT(0); FMA8(a, v2, p4.y);
      PREP(we, wo, w.z, slot); EXTRACT(u, we, wo); READ8(v, u, 256);
      PREP(we, wo, w.w, slot); EXTRACT(u2, we, wo); READ8(v2, u2, 384);
      WAIT(8); FMA8(a, v, p4.z); WAIT(0); FMA8(a, v2, p4.w);
    } else if (MODE == 4) {
      // the whole quad as ONE instruction block: pinned temporaries (v100-v125), look-up results in even-aligned register
      // pairs, v_pk_fma_f32 (two channels per instruction, full rate) instead of v_fmac
      f32x2 pxy = {p4.x, p4.y}, pzw = {p4.z, p4.w};
      asm volatile(QUAD_ASM
                   : [a0] "+v"(a2[0]), [a1] "+v"(a2[1]), [a2] "+v"(a2[2]), [a3] "+v"(a2[3])
                   : [wx] "v"(w.x), [wy] "v"(w.y), [wz] "v"(w.z), [ww] "v"(w.w), [pxy] "v"(pxy), [pzw] "v"(pzw), [slot] "v"(slot),
                     [mask] "s"(0x3C3C3C3Cu)
                   : "memory", "v100", "v101", "v102", "v103", "v104", "v105", "v106", "v107", "v108", "v109", "v110", "v111",
                     "v112", "v113", "v114", "v115", "v116", "v117", "v118", "v119", "v120", "v121", "v122", "v123", "v124", "v125");
    } else {
      PREP(we, wo, w.x, slot); EXTRACT(u, we, wo); READ8(v, u, 0);
      PREP(we, wo, w.y, slot); EXTRACT(u2, we, wo); READ8(v2, u2, 128);
      WAIT(8); FMA8(a, v, p4.x);
      PREP(we, wo, w.z, slot); EXTRACT(u, we, wo); READ8(v, u, 256);
      WAIT(8); FMA8(a, v2, p4.y);
      PREP(we, wo, w.w, slot); EXTRACT(u2, we, wo); READ8(v2, u2, 384);
      WAIT(8); FMA8(a, v, p4.z);
      WAIT(0); FMA8(a, v2, p4.w);
    }
  }
  float s = a[0] + a[1] + a[2] + a[3] + a[4] + a[5] + a[6] + a[7] + a2[0].x + a2[0].y + a2[1].x + a2[1].y + a2[2].x + a2[2].y + a2[3].x + a2[3].y;
  if (s == 12345.678f) out[0] = s;
}

template <int MODE>
static void run(const char *name, float *d, int blocks) {
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  int iters = 5000;
  float ms = 0;
  for (int rep = 0; rep < 2; rep++) {
    (void)hipEventRecord(e0);
    k<MODE><<<blocks, 512>>>(d, iters, 12345u);
    (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    (void)hipEventElapsedTime(&ms, e0, e1);
  }
  double steps_per_simd = (blocks / 256.0) * 2.0 * iters * 32.0;
  printf("%-60s %d waves/SIMD: %8.3f ms -> %6.2f ns per code-step per SIMD\n", name, blocks / 128, ms, ms * 1e6 / steps_per_simd);
}
#define RUN(M, NAME) run<M>(NAME, d, 512); run<M>(NAME, d, 256);
int main() {
  float *d;
  (void)hipMalloc(&d, 4096);
  RUN(0, "quad loop, 8 look-ups in flight, full waits")
  RUN(1, "quad loop, 16 in flight")
  RUN(2, "quad loop, software pipelined (reads of t+1 before FMAs of t)")
  RUN(3, "as above + probabilities from LDS")
  RUN(4, "one asm block per quad, v_pk_fma_f32 on pinned pairs")
  return 0;
}
