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

// kvq_fused_append.hip: kvq_decode_prologue with the choice of score images (pair_images: also the 3-bit fp16 pair sums)
int decode_prologue(int bits, int32_t *kmat, const float *klut, const float *klut_off, const void *k, const float *lo,
                    const float *hi, float *koutliers, int32_t *kidx, int64_t kcol, int32_t *vmat, float *vlut_rows,
                    const float *vlut_sorted, const void *v, float *voutliers, int32_t *vidx, int64_t vcol, const void *q,
                    int acts_are_half, int thr_k, int H, int hd, int64_t max_len, float *koutliers_t, int32_t *kidx_t,
                    const float *klut_ends, const float *klut_score, const kvq_vopts *vnorm, const kvq_sinks *sinks,
                    void *score_workspace, size_t score_workspace_bytes, int pair_images, void *stream);

// kvq_mix_v.hip: the launches around the p.V kernels (shared with tools/experiments/kvq_mix_va.hip)
int launch_mix_reduce(const float *partial, float *mul, int n_ranges, int q_len, int C, int accumulate, hipStream_t st);
int launch_softmax_merge(const float *parts, int n_parts, const void *sink, void *sink_probs, int n_sink, float *mz,
                         const void *v_sink, float *sink_out, int H, hipStream_t st);
int mix_merge_in_kernel_parts();

}  // namespace kvq
