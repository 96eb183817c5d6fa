// SoP path (global operators) and feature-layout helpers, gfx950.
#include "s3grl_internal.hpp"

namespace s3grl {
namespace {

// X [N, F] with arbitrary ld -> [N, ldy] with ldy % 4 == 0, padding columns zeroed.
__global__ void copy_pad_kernel(const float* __restrict__ X, int64_t ldx, int64_t N, int64_t F,
                                float* __restrict__ Y, int64_t ldy) {
  const int64_t total = N * ldy;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total;
       i += (int64_t)gridDim.x * blockDim.x) {
    const int64_t r = i / ldy, c = i - r * ldy;
    Y[i] = c < F ? X[r * ldx + c] : 0.f;
  }
}

}  // namespace

s3grl_status launch_copy_pad(s3grl_context* ctx, const float* X, int64_t ldx, int64_t N, int64_t F,
                             float* Y, int64_t ldy) {
  const int64_t total = N * ldy;
  if (total == 0) return S3GRL_OK;
  const unsigned grid = (unsigned)std::min<int64_t>((total + 255) / 256, 256 * 32);
  hipLaunchKernelGGL(copy_pad_kernel, dim3(grid), dim3(256), 0, ctx->stream, X, ldx, N, F, Y, ldy);
  S3GRL_HIP_TRY(hipGetLastError());
  return S3GRL_OK;
}

}  // namespace s3grl

extern "C" {

s3grl_status s3grl_sop_create(s3grl_context*, const s3grl_graph*, const float*, int64_t, int64_t,
                              int32_t, s3grl_sop**) {
  s3grl::set_last_error("SoP not built yet");
  return S3GRL_ERR_NOT_IMPLEMENTED;
}
s3grl_status s3grl_sop_destroy(s3grl_sop*) { return S3GRL_OK; }
s3grl_status s3grl_sop_run(s3grl_context*, const s3grl_sop*, const int64_t*, int64_t, float*) {
  s3grl::set_last_error("SoP not built yet");
  return S3GRL_ERR_NOT_IMPLEMENTED;
}
}
