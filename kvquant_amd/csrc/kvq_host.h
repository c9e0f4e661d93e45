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

// kvq_append.hip: packed codes of S prompt tokens of K (no rescaled output)
int pack_k_codes(int bits, int32_t *mat, const float *lut, const float *x, const float *lo, const float *hi, int H,
                 int hd, int64_t S, int64_t max_len, int64_t col0, hipStream_t st);

}  // namespace kvq
