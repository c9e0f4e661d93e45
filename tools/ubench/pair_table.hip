// microbenchmark 5 (development tool): the "pair table" form of the p.V inner loop.  When every token's codebook is
// an affine image of ONE sorted codebook (row_t[j] = sorted[j]*sf_t + off_t -- what the V append writes), the
// look-up table does not change with the token, and a BYTE of the packed row (two 4-bit codes) can index a constant
// 256-entry table of float2 (sorted[lo], sorted[hi]): one v_perm_b32 (byte -> LDS address) + one ds_read_b64 + two
// v_fmac per two codes.  64 lanes hitting random entries would conflict, so the table is replicated R times
// (entry stride R*8 bytes, lane l uses replica l % R): with R = 32 every lane of a half-wave has its own bank pair.
//   per iteration: one ds_read_b128 of "tile" data (4 words = 32 codes), 16 v_perm, 16 ds_read_b64, 32 v_fmac
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
typedef float f32x2 __attribute__((ext_vector_type(2)));

#define ADDR4(u, w, lo)                                                                                        \
  asm volatile("v_perm_b32 %0, %4, %5, %6\n v_perm_b32 %1, %4, %5, %7\n v_perm_b32 %2, %4, %5, %8\n v_perm_b32 %3, %4, %5, %9" \
               : "=&v"(u[0]), "=&v"(u[1]), "=&v"(u[2]), "=&v"(u[3])                                             \
               : "v"(w), "v"(lo), "s"(0x0C0C0400u), "s"(0x0C0C0500u), "s"(0x0C0C0600u), "s"(0x0C0C0700u))
#define READ4(v, u)                                                                                             \
  asm volatile("ds_read_b64 %0, %4\n ds_read_b64 %1, %5\n ds_read_b64 %2, %6\n ds_read_b64 %3, %7"               \
               : "=&v"(v[0]), "=&v"(v[1]), "=&v"(v[2]), "=&v"(v[3]) : "v"(u[0]), "v"(u[1]), "v"(u[2]), "v"(u[3]) : "memory")
#define FMA8(a, v, p)                                                                                           \
  asm volatile("v_fmac_f32 %0, %8, %16\n v_fmac_f32 %1, %9, %16\n v_fmac_f32 %2, %10, %16\n v_fmac_f32 %3, %11, %16\n" \
               "v_fmac_f32 %4, %12, %16\n v_fmac_f32 %5, %13, %16\n v_fmac_f32 %6, %14, %16\n v_fmac_f32 %7, %15, %16\n" \
               : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(a[4]), "+v"(a[5]), "+v"(a[6]), "+v"(a[7])      \
               : "v"(v[0].x), "v"(v[0].y), "v"(v[1].x), "v"(v[1].y), "v"(v[2].x), "v"(v[2].y), "v"(v[3].x), "v"(v[3].y), "v"(p))
#define WAIT(N) asm volatile("s_waitcnt lgkmcnt(" #N ")" ::: "memory")

template <int R, int NT>
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
  const unsigned lo = (threadIdx.x % R) * 8;
  const unsigned tbase = 256 * R * 8 + (threadIdx.x & 255) * 128;
  float pt = 0.37f;
  asm volatile("" : "+v"(pt));
  for (int it = 0; it < iters; it++) {
    uint4 w;
    const unsigned qa = tbase + ((it & 7) << 4);
    asm volatile("ds_read_b128 %0, %1" : "=v"(w) : "v"(qa) : "memory");
    WAIT(0);
    unsigned u[4], u2[4];
    f32x2 v[4], v2[4];
    if (R == 32) {
      ADDR4(u, w.x, lo); READ4(v, u);
      ADDR4(u2, w.y, lo); READ4(v2, u2);
      WAIT(4); FMA8(a, v, pt);
      ADDR4(u, w.z, lo); READ4(v, u);
      WAIT(4); FMA8(a, v2, pt);
      ADDR4(u2, w.w, lo); READ4(v2, u2);
      WAIT(4); FMA8(a, v, pt);
      WAIT(0); FMA8(a, v2, pt);
    } else {
      // entry stride R*8 < 256: the byte has to be scaled with shifts instead of v_perm
      const unsigned sh = R == 16 ? 7 : 6;
      auto addr4 = [&](unsigned (&uu)[4], unsigned ww) {
        uu[0] = ((ww & 0xffu) << sh) | lo;
        uu[1] = (((ww >> 8) & 0xffu) << sh) | lo;
        uu[2] = (((ww >> 16) & 0xffu) << sh) | lo;
        uu[3] = ((ww >> 24) << sh) | lo;
      };
      addr4(u, w.x); READ4(v, u);
      addr4(u2, w.y); READ4(v2, u2);
      WAIT(4); FMA8(a, v, pt);
      addr4(u, w.z); READ4(v, u);
      WAIT(4); FMA8(a, v2, pt);
      addr4(u2, w.w); READ4(v2, u2);
      WAIT(4); FMA8(a, v, pt);
      WAIT(0); FMA8(a, v2, pt);
    }
  }
  float s = a[0] + a[1] + a[2] + a[3] + a[4] + a[5] + a[6] + a[7];
  if (s == 12345.678f) out[0] = s;
}

template <int R, int NT>
static void run(const char *name, float *d) {
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  int iters = 5000;
  float ms = 0;
  const size_t lds = 256 * R * 8 + 32768;
  (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&k<R, NT>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  for (int rep = 0; rep < 2; rep++) {
    (void)hipEventRecord(e0);
    k<R, NT><<<256, NT, lds>>>(d, iters, 12345u);
    (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    (void)hipEventElapsedTime(&ms, e0, e1);
  }
  hipError_t err = hipGetLastError();
  double waves_per_simd = NT / 256.0;
  double steps_per_simd = waves_per_simd * iters * 32.0;
  printf("%-40s R=%2d %4d lanes/CU (%g waves/SIMD): %8.3f ms -> %6.2f ns per code-step per SIMD  %s\n", name, R, NT,
         waves_per_simd, ms, ms * 1e6 / steps_per_simd, err == hipSuccess ? "" : hipGetErrorString(err));
}
int main() {
  float *d;
  (void)hipMalloc(&d, 4096);
  run<32, 1024>("pair table", d);
  run<32, 512>("pair table", d);
  run<16, 1024>("pair table", d);
  run<8, 1024>("pair table", d);
  return 0;
}
