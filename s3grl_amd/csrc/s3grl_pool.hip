// Centre / common-neighbour pooling of the consumer, gfx950.
//
// Reference: SIGNNet._centre_pool_helper (models.py:339-369), applied to h = operator_diff(x):
//   h_a = h[c] ⊙ h[c+1]                      c = first row of every link (its src; dst follows)
//   k_heuristic: the remaining rows of the link (common neighbours) are mean / sum pooled with
//   `size=B`, so links without any pool to zeros, and concatenated:  out = [h_a | pool].
// The reference finds c with np.unique on the host every batch (models.py:341: a device->host
// sync) and builds its mask on the CPU; here the rows of link b are row_ptr[b]..row_ptr[b+1]
// (the engine's own output layout) and nothing leaves the device.
// One wavefront per link; lanes own feature columns (float4), rows are walked serially, so sums
// have a fixed order.  Backward is the exact adjoint.
#include "s3grl_internal.hpp"

namespace s3grl {
namespace {

typedef float float4_t __attribute__((ext_vector_type(4)));

enum { kPoolNone = 0, kPoolMean = 1, kPoolSum = 2 };

__global__ __launch_bounds__(256) void pool_fwd_kernel(const float* __restrict__ h,
                                                       const int64_t* __restrict__ row_ptr, int64_t B,
                                                       int H, int mode, float* __restrict__ out) {
  const int lane = threadIdx.x & 63;
  const int64_t b = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (b >= B) return;
  const int64_t r0 = row_ptr[b], r1 = row_ptr[b + 1];
  const int OW = mode == kPoolNone ? H : 2 * H;
  float* __restrict__ o = out + b * OW;
  const float* __restrict__ hs = h + r0 * H;
  const float* __restrict__ hd = hs + H;
  const int extra = (int)(r1 - r0 - 2);
  const float scale = (mode == kPoolMean && extra > 0) ? 1.0f / (float)extra : 1.0f;
  for (int c = lane * 4; c < H; c += 256) {
    if (c + 4 <= H && (H & 3) == 0) {
      const float4_t a = *reinterpret_cast<const float4_t*>(hs + c);
      const float4_t d = *reinterpret_cast<const float4_t*>(hd + c);
      *reinterpret_cast<float4_t*>(o + c) = a * d;
      if (mode != kPoolNone) {
        float4_t s = (float4_t)(0.f);
        for (int e = 0; e < extra; ++e)
          s += *reinterpret_cast<const float4_t*>(hs + (int64_t)(2 + e) * H + c);
        *reinterpret_cast<float4_t*>(o + H + c) = s * scale;
      }
    } else {
      for (int k = c; k < min(c + 4, H); ++k) {
        o[k] = hs[k] * hd[k];
        if (mode != kPoolNone) {
          float s = 0.f;
          for (int e = 0; e < extra; ++e) s += hs[(int64_t)(2 + e) * H + k];
          o[H + k] = s * scale;
        }
      }
    }
  }
}

__global__ __launch_bounds__(256) void pool_bwd_kernel(const float* __restrict__ h,
                                                       const int64_t* __restrict__ row_ptr, int64_t B,
                                                       int H, int mode, const float* __restrict__ gout,
                                                       float* __restrict__ gh) {
  const int lane = threadIdx.x & 63;
  const int64_t b = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (b >= B) return;
  const int64_t r0 = row_ptr[b], r1 = row_ptr[b + 1];
  const int OW = mode == kPoolNone ? H : 2 * H;
  const float* __restrict__ g = gout + b * OW;
  const float* __restrict__ hs = h + r0 * H;
  const float* __restrict__ hd = hs + H;
  float* __restrict__ gs = gh + r0 * H;
  const int extra = (int)(r1 - r0 - 2);
  const float scale = (mode == kPoolMean && extra > 0) ? 1.0f / (float)extra : 1.0f;
  for (int k = lane; k < H; k += 64) {
    const float ga = g[k];
    gs[k] = ga * hd[k];
    gs[H + k] = ga * hs[k];
    const float gp = mode == kPoolNone ? 0.f : g[H + k] * scale;
    for (int e = 0; e < extra; ++e) gs[(int64_t)(2 + e) * H + k] = gp;
  }
}

}  // namespace
}  // namespace s3grl

using namespace s3grl;

extern "C" {

static s3grl_status pool_args(s3grl_context* ctx, const void* h, const void* row_ptr, int64_t B,
                              int64_t H, int32_t mode, const void* a, const void* b) {
  if (!ctx || B < 0 || H <= 0 || H > (1 << 20)) return S3GRL_ERR_INVALID_ARGUMENT;
  if (B > 0 && (!h || !row_ptr || !a || !b)) return S3GRL_ERR_INVALID_ARGUMENT;
  if (mode < 0 || mode > 2) {
    set_last_error("Check pool strat: only none / mean / sum are implemented");
    return S3GRL_ERR_NOT_IMPLEMENTED;
  }
  return S3GRL_OK;
}

s3grl_status s3grl_centre_pool_forward(s3grl_context* ctx, const float* h, const int64_t* row_ptr,
                                       int64_t B, int64_t H, int32_t mode, float* out) {
  S3GRL_TRY(pool_args(ctx, h, row_ptr, B, H, mode, out, out));
  if (B == 0) return S3GRL_OK;
  S3GRL_HIP_TRY(hipSetDevice(ctx->device));
  hipLaunchKernelGGL(pool_fwd_kernel, dim3((unsigned)((B + 3) / 4)), dim3(256), 0, ctx->stream, h,
                     row_ptr, B, (int)H, mode, out);
  S3GRL_HIP_TRY(hipGetLastError());
  return S3GRL_OK;
}

s3grl_status s3grl_centre_pool_backward(s3grl_context* ctx, const float* h, const int64_t* row_ptr,
                                        int64_t B, int64_t H, int32_t mode, const float* grad_out,
                                        float* grad_h) {
  S3GRL_TRY(pool_args(ctx, h, row_ptr, B, H, mode, grad_out, grad_h));
  if (B == 0) return S3GRL_OK;
  S3GRL_HIP_TRY(hipSetDevice(ctx->device));
  hipLaunchKernelGGL(pool_bwd_kernel, dim3((unsigned)((B + 3) / 4)), dim3(256), 0, ctx->stream, h,
                     row_ptr, B, (int)H, mode, grad_out, grad_h);
  S3GRL_HIP_TRY(hipGetLastError());
  return S3GRL_OK;
}

}  // extern "C"

S3GRL_DEFINE_TOUCH(pool)
