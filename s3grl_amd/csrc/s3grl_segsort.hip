// rocPRIM's segmented radix sort in a translation unit of its own: only s3grl_plan_export_subgraphs on graphs
// beyond 524 288 nodes uses it, and HIP loads a unit's code object whole at the first launch of any of its
// kernels — inside s3grl_relabel.hip it was part of what every process's first s3grl_graph_create loaded.
#include <rocprim/device/device_segmented_radix_sort.hpp>

#include "s3grl_internal.hpp"

namespace s3grl {

// keys of segment k are keys_in[seg[k] .. seg[k + 1]); tmp == nullptr: the size query
s3grl_status segmented_sort_i32(s3grl_context* ctx, void* tmp, size_t* bytes, const int32_t* keys_in, int32_t* keys_out,
                                size_t n, unsigned segments, const int64_t* seg) {
  S3GRL_HIP_TRY(rocprim::segmented_radix_sort_keys(tmp, *bytes, keys_in, keys_out, n, segments, seg, seg + 1, 0, 32,
                                                   ctx->stream));
  return S3GRL_OK;
}

}  // namespace s3grl
