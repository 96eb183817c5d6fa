// Measurement aid: reads of a KNOWN byte count in the access shapes of the engine's kernels, to
// calibrate rocprofv3's FETCH_SIZE on gfx950 (MI355X_MICROARCH.md, HBM section: the counter tallies
// fabric requests at 64 B each; 16-byte-per-lane coalesced streams come out at exactly 1/2, "other
// access widths are uncalibrated: calibrate on a known byte count in your own access pattern").
// Not on the product path; driven by tools/pmc_calibrate.py under rocprofv3.
#include "s3grl_internal.hpp"

namespace s3grl {
namespace {

typedef float float4_t __attribute__((ext_vector_type(4)));
typedef float float2_t __attribute__((ext_vector_type(2)));

// pattern 0 / 1 / 2: one pass over the buffer, 16 / 8 / 4 bytes per lane, consecutive lanes
// consecutive addresses (a wave-instruction = 1024 / 512 / 256 contiguous bytes)
template <int W>
__global__ __launch_bounds__(256) void calib_stream_kernel(const char* __restrict__ buf, int64_t bytes,
                                                           float* __restrict__ sink) {
  const int64_t n = bytes / W;
  float acc = 0.f;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    if constexpr (W == 16) {
      const float4_t v = *reinterpret_cast<const float4_t*>(buf + i * 16);
      acc += v.x + v.y + v.z + v.w;
    } else if constexpr (W == 8) {
      const float2_t v = *reinterpret_cast<const float2_t*>(buf + i * 8);
      acc += v.x + v.y;
    } else {
      acc += *reinterpret_cast<const float*>(buf + i * 4);
    }
  }
  if (acc == 123.456f) sink[0] = acc;   // keeps the loads alive
}

// pattern 3: `rows` rows of row_bytes (a multiple of 16) at pseudo-random 16-byte-aligned places
// of the buffer, one wavefront per row, 16 bytes per lane, lanes beyond the row idle — the shape of
// the gather kernels' feature loads (packed rows ~0.7 KB, dense rows 2 KB)
__global__ __launch_bounds__(256) void calib_rows_kernel(const char* __restrict__ buf, int64_t bytes,
                                                         int64_t rows, int row_bytes, float* __restrict__ sink) {
  const int lane = threadIdx.x & 63;
  const int64_t slots = (bytes - row_bytes) / 16;
  float acc = 0.f;
  for (int64_t r = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6); r < rows; r += (int64_t)gridDim.x * 4) {
    uint64_t x = (uint64_t)r * 0x9E3779B97F4A7C15ull;
    x ^= x >> 29;
    x *= 0xBF58476D1CE4E5B9ull;
    x ^= x >> 32;
    const char* __restrict__ row = buf + (int64_t)(x % (uint64_t)slots) * 16;
    for (int o = lane * 16; o < row_bytes; o += 1024) {
      const float4_t v = *reinterpret_cast<const float4_t*>(row + o);
      acc += v.x + v.y + v.z + v.w;
    }
  }
  if (acc == 123.456f) sink[0] = acc;
}

}  // namespace
}  // namespace s3grl

using namespace s3grl;

extern "C" s3grl_status s3grl_calibration_read(s3grl_context* ctx, const void* buf, int64_t bytes,
                                               int32_t pattern, int64_t rows, int32_t row_bytes,
                                               int64_t* requested_bytes) {
  if (!ctx || !buf || bytes < 4096 || !requested_bytes) return S3GRL_ERR_INVALID_ARGUMENT;
  S3GRL_HIP_TRY(hipSetDevice(ctx->device));
  float* sink = reinterpret_cast<float*>(ctx->d_scalars + 60);
  const char* b = static_cast<const char*>(buf);
  const unsigned grid = 256 * 16;
  switch (pattern) {
    case 0:
      hipLaunchKernelGGL(calib_stream_kernel<16>, dim3(grid), dim3(256), 0, ctx->stream, b, bytes, sink);
      *requested_bytes = bytes / 16 * 16;
      break;
    case 1:
      hipLaunchKernelGGL(calib_stream_kernel<8>, dim3(grid), dim3(256), 0, ctx->stream, b, bytes, sink);
      *requested_bytes = bytes / 8 * 8;
      break;
    case 2:
      hipLaunchKernelGGL(calib_stream_kernel<4>, dim3(grid), dim3(256), 0, ctx->stream, b, bytes, sink);
      *requested_bytes = bytes / 4 * 4;
      break;
    case 3:
      if (rows <= 0 || row_bytes < 16 || row_bytes % 16 || row_bytes >= bytes) return S3GRL_ERR_INVALID_ARGUMENT;
      hipLaunchKernelGGL(calib_rows_kernel, dim3(grid), dim3(256), 0, ctx->stream, b, bytes, rows, (int)row_bytes,
                         sink);
      *requested_bytes = rows * row_bytes;
      break;
    default:
      return S3GRL_ERR_INVALID_ARGUMENT;
  }
  S3GRL_HIP_TRY(hipGetLastError());
  S3GRL_HIP_TRY(hipStreamSynchronize(ctx->stream));
  return S3GRL_OK;
}
