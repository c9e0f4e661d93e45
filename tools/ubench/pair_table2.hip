// microbenchmark 6 (development tool): pair-table p.V inner loop, replication R in {32, 16, 8}, address forms,
// v_fmac x2 vs v_pk_fma_f32, at 512 / 1024 lanes per CU.  Prints ns per code-step per SIMD (64 codes of one lane row).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
typedef float f32x2 __attribute__((ext_vector_type(2)));

#define READ4(v, u)                                                                                             \
  asm volatile("ds_read_b64 %0, %4\n ds_read_b64 %1, %5\n ds_read_b64 %2, %6\n ds_read_b64 %3, %7"               \
               : "=&v"(v[0]), "=&v"(v[1]), "=&v"(v[2]), "=&v"(v[3]) : "v"(u[0]), "v"(u[1]), "v"(u[2]), "v"(u[3]) : "memory")
#define FMA8(a, v, p)                                                                                           \
  asm volatile("v_fmac_f32 %0, %8, %16\n v_fmac_f32 %1, %9, %16\n v_fmac_f32 %2, %10, %16\n v_fmac_f32 %3, %11, %16\n" \
               "v_fmac_f32 %4, %12, %16\n v_fmac_f32 %5, %13, %16\n v_fmac_f32 %6, %14, %16\n v_fmac_f32 %7, %15, %16\n" \
               : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(a[4]), "+v"(a[5]), "+v"(a[6]), "+v"(a[7])      \
               : "v"(v[0].x), "v"(v[0].y), "v"(v[1].x), "v"(v[1].y), "v"(v[2].x), "v"(v[2].y), "v"(v[3].x), "v"(v[3].y), "v"(p))
#define PK4(a, v, p)                                                                                            \
  asm volatile("v_pk_fma_f32 %0, %4, %8, %0 op_sel_hi:[1,0,1]\n v_pk_fma_f32 %1, %5, %8, %1 op_sel_hi:[1,0,1]\n"    \
               "v_pk_fma_f32 %2, %6, %8, %2 op_sel_hi:[1,0,1]\n v_pk_fma_f32 %3, %7, %8, %3 op_sel_hi:[1,0,1]\n"    \
               : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]) : "v"(v[0]), "v"(v[1]), "v"(v[2]), "v"(v[3]), "v"(p))
#define WAIT(N) asm volatile("s_waitcnt lgkmcnt(" #N ")" ::: "memory")

template <int R>
__device__ __forceinline__ void addr4(unsigned (&u)[4], unsigned w, unsigned lo) {
  if constexpr (R == 32) {
    asm volatile("v_perm_b32 %0, %4, %5, %6\n v_perm_b32 %1, %4, %5, %7\n v_perm_b32 %2, %4, %5, %8\n v_perm_b32 %3, %4, %5, %9"
                 : "=&v"(u[0]), "=&v"(u[1]), "=&v"(u[2]), "=&v"(u[3])
                 : "v"(w), "v"(lo), "s"(0x0C0C0400u), "s"(0x0C0C0500u), "s"(0x0C0C0600u), "s"(0x0C0C0700u));
  } else {
    // entry stride R*8 = 128 / 64 bytes: byte m << SH | lo.  Byte 0: and + lshl_or; bytes 1..3: bfe + lshl_or
    constexpr int SH = R == 16 ? 7 : 6;
    unsigned b0, b1, b2, b3;
    asm volatile("v_and_b32 %0, 0xff, %4\n v_bfe_u32 %1, %4, 8, 8\n v_bfe_u32 %2, %4, 16, 8\n v_lshrrev_b32 %3, 24, %4"
                 : "=&v"(b0), "=&v"(b1), "=&v"(b2), "=&v"(b3) : "v"(w));
    asm volatile("v_lshl_or_b32 %0, %4, %9, %8\n v_lshl_or_b32 %1, %5, %9, %8\n v_lshl_or_b32 %2, %6, %9, %8\n v_lshl_or_b32 %3, %7, %9, %8"
                 : "=&v"(u[0]), "=&v"(u[1]), "=&v"(u[2]), "=&v"(u[3]) : "v"(b0), "v"(b1), "v"(b2), "v"(b3), "v"(lo), "n"(SH));
  }
}

template <int R, int NT, bool PK>
__global__ __launch_bounds__(NT, 1) void k(float *out, int iters, unsigned seed) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  f32x2 *tab = reinterpret_cast<f32x2 *>(smem);                       // [256][R]
  unsigned *tile = reinterpret_cast<unsigned *>(smem + 256 * R * 8);  // 32 KB
  for (int i = threadIdx.x; i < 256 * R; i += NT) {
    const int b = i / R;
    f32x2 t = {(float)(b & 15) * 0.01f, (float)(b >> 4) * 0.01f};
    tab[i] = t;
  }
  for (int i = threadIdx.x; i < 8192; i += NT) {
    unsigned x = (i * 2654435761u) ^ seed;
    x ^= x >> 15; x *= 2246822519u; x ^= x >> 13;
    tile[i] = x;
  }
  __syncthreads();
  float a[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  f32x2 a2[4] = {{0, 0}, {0, 0}, {0, 0}, {0, 0}};
  const unsigned lo = (threadIdx.x % R) * 8;
  const unsigned tbase = 256 * R * 8 + (threadIdx.x & 255) * 128;
  f32x2 pt = {0.37f, 0.37f};
  asm volatile("" : "+v"(pt));
  for (int it = 0; it < iters; it++) {
    uint4 w;
    const unsigned qa = tbase + ((it & 7) << 4);
    asm volatile("ds_read_b128 %0, %1" : "=v"(w) : "v"(qa) : "memory");
    WAIT(0);
    unsigned u[4], u2[4];
    f32x2 v[4], v2[4];
    if constexpr (PK) {
      addr4<R>(u, w.x, lo); READ4(v, u);
      addr4<R>(u2, w.y, lo); READ4(v2, u2);
      WAIT(4); PK4(a2, v, pt);
      addr4<R>(u, w.z, lo); READ4(v, u);
      WAIT(4); PK4(a2, v2, pt);
      addr4<R>(u2, w.w, lo); READ4(v2, u2);
      WAIT(4); PK4(a2, v, pt);
      WAIT(0); PK4(a2, v2, pt);
    } else {
      addr4<R>(u, w.x, lo); READ4(v, u);
      addr4<R>(u2, w.y, lo); READ4(v2, u2);
      WAIT(4); FMA8(a, v, pt.x);
      addr4<R>(u, w.z, lo); READ4(v, u);
      WAIT(4); FMA8(a, v2, pt.x);
      addr4<R>(u2, w.w, lo); READ4(v2, u2);
      WAIT(4); FMA8(a, v, pt.x);
      WAIT(0); FMA8(a, v2, pt.x);
    }
  }
  float s = a[0] + a[1] + a[2] + a[3] + a[4] + a[5] + a[6] + a[7] + a2[0].x + a2[0].y + a2[1].x + a2[1].y + a2[2].x + a2[2].y + a2[3].x + a2[3].y;
  if (s == 12345.678f) out[0] = s;
}

template <int R, int NT, bool PK>
static void run(float *d) {
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  int iters = 5000;
  float ms = 0;
  const size_t lds = 256 * R * 8 + 32768;
  (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&k<R, NT, PK>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  for (int rep = 0; rep < 2; rep++) {
    (void)hipEventRecord(e0);
    k<R, NT, PK><<<256, NT, lds>>>(d, iters, 12345u);
    (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    (void)hipEventElapsedTime(&ms, e0, e1);
  }
  hipError_t err = hipGetLastError();
  double waves_per_simd = NT / 256.0;
  double steps_per_simd = waves_per_simd * iters * 32.0;   // 32 codes per lane per iteration
  printf("pair table R=%2d %4d lanes/CU (%g waves/SIMD) %s: %8.3f ms -> %6.2f ns per code-step per SIMD  %s\n", R, NT,
         waves_per_simd, PK ? "pk_fma" : "fmac x2", ms, ms * 1e6 / steps_per_simd, err == hipSuccess ? "" : hipGetErrorString(err));
}
int main() {
  float *d;
  (void)hipMalloc(&d, 4096);
  run<32, 1024, false>(d); run<32, 1024, true>(d);
  run<32, 512, false>(d);  run<32, 512, true>(d);
  run<32, 256, false>(d);
  run<16, 1024, false>(d); run<16, 1024, true>(d);
  run<16, 512, false>(d);
  run<8, 1024, false>(d);  run<8, 1024, true>(d);
  run<8, 512, false>(d);
  return 0;
}
