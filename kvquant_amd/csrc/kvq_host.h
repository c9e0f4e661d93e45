// Host-side helpers shared by the C-ABI translation units.
#pragma once
#include <hip/hip_runtime.h>

#include "../../include/kvq.h"

namespace kvq {

int &last_hip_error_ref();

// Launch errors are sticky per thread until the next successful check.
inline int check_launch() {
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) {
    last_hip_error_ref() = (int)e;
    return KVQ_ELAUNCH;
  }
  return KVQ_OK;
}

}  // namespace kvq
