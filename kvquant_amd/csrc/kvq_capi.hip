// Library-level entry points of libkvq.so (version, error strings).
#include "kvq_common.h"
#include "kvq_host.h"

#include <hip/hip_fp16.h>

namespace kvq {
__global__ void rope_freqs_kernel(float rope_theta, float *out) {
  if (threadIdx.x < kHeadDim / 2) out[threadIdx.x] = rope_freq(rope_theta, threadIdx.x);
}

// q * cos + rotate_half(q) * sin in fp16 with torch's roundings: every product and the sum are formed in fp32 and
// rounded to fp16 (what three elementwise fp16 torch kernels do); rotate_half(q)[d] = -q[d + 64] for d < 64, q[d - 64] else
__global__ __launch_bounds__(128) void rope_q_f16_kernel(const __half *__restrict__ q, const __half *__restrict__ cosv,
                                                         const __half *__restrict__ sinv, __half *__restrict__ out) {
  const int h = blockIdx.x, d = threadIdx.x;
  const __half *qh = q + (int64_t)h * kHeadDim;
  const float x = __half2float(qh[d]);
  const float r = d < kHeadDim / 2 ? -__half2float(qh[d + kHeadDim / 2]) : __half2float(qh[d - kHeadDim / 2]);
  const __half a = __float2half_rn(x * __half2float(cosv[d]));
  const __half b = __float2half_rn(r * __half2float(sinv[d]));
  out[(int64_t)h * kHeadDim + d] = __float2half_rn(__half2float(a) + __half2float(b));
}

int &last_hip_error_ref() {
  static thread_local int e = 0;
  return e;
}
}  // namespace kvq

extern "C" {

int kvq_version(void) { return 402; }    // history: include/kvq.h

const char *kvq_strerror(int code) {
  switch (code) {
    case KVQ_OK: return "ok";
    case KVQ_EINVAL: return "invalid argument (null pointer, unsupported bits/head_dim, or size out of range)";
    case KVQ_ELAUNCH: return "HIP launch/runtime error (see kvq_last_hip_error)";
    case KVQ_EWORKSPACE: return "workspace missing or too small";
    default: return "unknown kvq error";
  }
}

int kvq_last_hip_error(void) { return kvq::last_hip_error_ref(); }

int kvq_rope_freqs(float rope_theta, float *out, void *stream) {
  if (!out || !(rope_theta > 0.f)) return KVQ_EINVAL;
  kvq::rope_freqs_kernel<<<1, 64, 0, (hipStream_t)stream>>>(rope_theta, out);
  return kvq::check_launch();
}

int kvq_rope_q_f16(const uint16_t *q, const uint16_t *cosv, const uint16_t *sinv, uint16_t *out, int H, int hd, void *stream) {
  if (!q || !cosv || !sinv || !out || H <= 0 || hd != kvq::kHeadDim) return KVQ_EINVAL;
  kvq::rope_q_f16_kernel<<<H, 128, 0, (hipStream_t)stream>>>(reinterpret_cast<const __half *>(q), reinterpret_cast<const __half *>(cosv),
                                                             reinterpret_cast<const __half *>(sinv), reinterpret_cast<__half *>(out));
  return kvq::check_launch();
}

}  // extern "C"
