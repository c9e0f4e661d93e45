// Library-level entry points of libkvq.so (version, error strings).
#include "kvq_host.h"

namespace kvq {
int &last_hip_error_ref() {
  static thread_local int e = 0;
  return e;
}
}  // namespace kvq

extern "C" {

int kvq_version(void) { return 100; }

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

}  // extern "C"
