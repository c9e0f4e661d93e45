// Library-level entry points of libkvq.so (version, error strings).
#include "kvq_common.h"
#include "kvq_host.h"

namespace kvq {
__global__ void rope_freqs_kernel(float rope_theta, float *out) {
  if (threadIdx.x < kHeadDim / 2) out[threadIdx.x] = rope_freq(rope_theta, threadIdx.x);
}

int &last_hip_error_ref() {
  static thread_local int e = 0;
  return e;
}
}  // namespace kvq

extern "C" {

int kvq_version(void) { return 300; }    // history: include/kvq.h

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

}  // extern "C"
