// Internal node order of the multi-hop kernels: ids in DESCENDING DEGREE order.
//
// The row walker gives every CSR row G lanes; rows longer than 2G neighbours keep their lanes in a
// tail loop while the other rows of the wavefront have finished, and on a citation graph the
// longest row of 16 sets the length of nearly every trip (the link kernels issue at under half of
// their lanes).  The node lists of a link are hop-major and ascending in id inside a hop, so with
// ids sorted by degree a wavefront's rows have nearly the same length — no relabelling work per
// link, one CSR permutation per graph.  Measured on PubMed (ids permuted outside the engine): link
// kernels 5.46 -> 4.17 ms (sign_k = 3), 13.4 -> 10.2 ms (sign_k = 5).
//
// Built once per graph (s3grl_graph_create): new id = rank of (degree descending, id ascending);
// the permuted CSR has its rows ascending in NEW ids (the kernels bisect rows).  Everything a plan
// hands out stays in the caller's ids: link_kernel translates the node lists, row nodes and job
// endpoints on the way out, common-neighbour rows are ordered by the caller's ids, and
// s3grl_plan_export_subgraphs restores "ascending id inside a hop".  One-hop plans on big graphs
// (s3grl_onehop.inl) walk the same order through oriented rows built from the permuted CSR
// (collab-scale link kernels 46.9 -> 42.1 ms).  SoP, sampled and random-walk plans keep the
// original order (their random draws are keyed by the caller's ids).
#include <algorithm>
#include <cstring>

#include <rocprim/device/device_radix_sort.hpp>

#include "s3grl_internal.hpp"
#include "s3grl_device.hpp"

namespace s3grl {
namespace {

__global__ void node_keys_kernel(const int32_t* __restrict__ indptr, int64_t N, uint32_t max_degree,
                                 uint64_t* __restrict__ keys) {
  const int64_t v = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (v >= N) return;
  const uint32_t d = (uint32_t)(indptr[v + 1] - indptr[v]);
  keys[v] = ((uint64_t)(max_degree - d) << 32) | (uint64_t)(uint32_t)v;
}

__global__ void perm_kernel(const uint64_t* __restrict__ sorted, const int32_t* __restrict__ indptr,
                            int64_t N, int32_t* __restrict__ old_of_new,
                            int32_t* __restrict__ new_of_old, int32_t* __restrict__ deg_new) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= N) return;
  const int32_t old = (int32_t)(sorted[i] & 0xffffffffull);
  old_of_new[i] = old;
  new_of_old[old] = (int32_t)i;
  deg_new[i] = indptr[old + 1] - indptr[old];
}

// one thread per stored entry of the ORIGINAL CSR, found by bisection of indptr
__global__ void edge_keys_kernel(const int32_t* __restrict__ indptr, const int32_t* __restrict__ indices,
                                 int64_t N, int64_t nnz, const int32_t* __restrict__ new_of_old,
                                 uint64_t* __restrict__ keys) {
  const int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= nnz) return;
  int64_t lo = 0, hi = N;   // last row with indptr[row] <= e
  while (hi - lo > 1) {
    const int64_t mid = (lo + hi) >> 1;
    if (indptr[mid] <= e) lo = mid; else hi = mid;
  }
  keys[e] = ((uint64_t)(uint32_t)new_of_old[lo] << 32) | (uint64_t)(uint32_t)new_of_old[indices[e]];
}

// first position of the sorted node keys whose degree is <= d (keys ascend in max_degree - degree)
__global__ void first_degree_le_kernel(const uint64_t* __restrict__ sorted, int64_t N, uint32_t max_degree,
                                       uint32_t d, int64_t* __restrict__ out) {
  if (threadIdx.x || blockIdx.x) return;
  const uint64_t want = (uint64_t)(max_degree >= d ? max_degree - d : 0) << 32;
  int64_t lo = 0, hi = N;
  while (lo < hi) {
    const int64_t mid = (lo + hi) >> 1;
    if (sorted[mid] < want) lo = mid + 1; else hi = mid;
  }
  out[0] = lo;
}

__global__ void low_words_kernel(const uint64_t* __restrict__ keys, int64_t n, int32_t* __restrict__ out) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) out[i] = (int32_t)(keys[i] & 0xffffffffull);
}

__global__ void offsets_to_i32_kernel(const int64_t* __restrict__ in, int64_t n, int32_t* __restrict__ out) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) out[i] = (int32_t)in[i];
}

__global__ void translate_links_kernel(const int64_t* __restrict__ links, int64_t L, int64_t N,
                                       const int32_t* __restrict__ new_of_old, int64_t* __restrict__ out) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= 2 * L) return;
  const int64_t v = links[i];
  out[i] = (v >= 0 && v < N) ? (int64_t)new_of_old[v] : v;   // an invalid id stays invalid (count_kernel reports it)
}

// s3grl_plan_export_subgraphs: every hop segment of every node list back to ascending id.  One
// workgroup per link, the segment as an N-bit LDS bitmap, enumerated like count_kernel does.
constexpr int kSortBlock = 256;
__global__ __launch_bounds__(kSortBlock) void sort_hops_kernel(const int64_t* __restrict__ node_off,
                                                               const int32_t* __restrict__ lvl, int W,
                                                               int32_t* __restrict__ nodes) {
  extern __shared__ uint32_t bm[];
  __shared__ int sh[kSortBlock / 64];
  const int64_t l = blockIdx.x;
  const int64_t o = node_off[l];
  const int n = (int)(node_off[l + 1] - o);
  const int tid = threadIdx.x;
  const int C = (W + kSortBlock - 1) / kSortBlock;
  const int w0 = min(tid * C, W), w1 = min(w0 + C, W);
  int begin = 0;
  for (int d = 0; d < kMaxLevels && begin < n; ++d) {
    const int end = min(lvl[l * kMaxLevels + d], n);
    if (end <= begin) continue;
    for (int t = tid; t < W; t += kSortBlock) bm[t] = 0;
    __syncthreads();
    for (int t = begin + tid; t < end; t += kSortBlock) {
      const int v = nodes[o + t];
      atomicOr(&bm[v >> 5], 1u << (v & 31));
    }
    __syncthreads();
    int mine = 0;
    for (int t = w0; t < w1; ++t) mine += __popc(bm[t]);
    int total;
    int pos = begin + block_excl_scan<kSortBlock>(mine, sh, total);
    for (int t = w0; t < w1; ++t) {
      uint32_t w = bm[t];
      while (w) {
        const int b = __ffs(w) - 1;
        w &= w - 1;
        nodes[o + pos++] = t * 32 + b;
      }
    }
    __syncthreads();
    begin = end;
  }
}

// Processing order of a plan's links on big graphs: links that share their higher-degree endpoint next
// to each other (key = (hub, other) in the ids the kernels walk), so that the workgroups running at
// the same time read the same hub rows — of the CSR in the sizing pass and the link kernels, of X in
// the gather.  Only the ORDER in which links are worked on changes; every output stays where the
// caller's list puts it.  Collab-scale graph (235 000 nodes, X = 120 MB, 1 M links in random order):
// gather 10.2 -> 7.6 ms, sizing pass 2.3 -> 1.7 ms; nothing on PubMed (everything cache-resident).
__global__ void link_order_keys_kernel(const int64_t* __restrict__ links, int64_t L,
                                       const int32_t* __restrict__ indptr, uint64_t* __restrict__ keys,
                                       int32_t* __restrict__ vals) {
  const int64_t l = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (l >= L) return;
  const int64_t s = links[2 * l], d = links[2 * l + 1];
  vals[l] = (int32_t)l;
  // (invalid endpoints sort to the end; the sizing pass reports them)
  uint64_t key = ~0ull;
  if (s >= 0 && d >= 0 && s < (int64_t)INT32_MAX && d < (int64_t)INT32_MAX && indptr) {
    const int ds = indptr[s + 1] - indptr[s], dd = indptr[d + 1] - indptr[d];
    const bool s_hub = ds > dd || (ds == dd && s < d);
    key = ((uint64_t)(uint32_t)(s_hub ? s : d) << 32) | (uint32_t)(s_hub ? d : s);
  }
  keys[l] = key;
}

// segment bounds of the (link, hop) pieces of the exported node lists, for the segmented sort below:
// seg[l * kMaxLevels + d] = where hop d of link l starts, seg[L * kMaxLevels] = Σn
__global__ void hop_segments_kernel(const int64_t* __restrict__ node_off, const int32_t* __restrict__ lvl,
                                    int64_t L, int64_t* __restrict__ seg) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i > L * kMaxLevels) return;
  if (i == L * kMaxLevels) {
    seg[i] = node_off[L];
    return;
  }
  const int64_t l = i / kMaxLevels;
  const int d = (int)(i - l * kMaxLevels);
  const int64_t o = node_off[l];
  const int n = (int)(node_off[l + 1] - o);
  seg[i] = o + (d == 0 ? 0 : min(lvl[l * kMaxLevels + d - 1], n));
}

// union of two ascending rows per node: sizes, then entries (one thread per node, two-pointer merge)
__global__ void union_count_kernel(const int32_t* __restrict__ ap, const int32_t* __restrict__ ai,
                                   const int32_t* __restrict__ bp, const int32_t* __restrict__ bi, int64_t N,
                                   int32_t* __restrict__ cnt) {
  const int64_t v = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (v >= N) return;
  int i = ap[v], j = bp[v], c = 0;
  const int ie = ap[v + 1], je = bp[v + 1];
  while (i < ie || j < je) {
    const int x = i < ie ? ai[i] : 0x7fffffff, y = j < je ? bi[j] : 0x7fffffff;
    i += x <= y ? 1 : 0;
    j += y <= x ? 1 : 0;
    ++c;
  }
  cnt[v] = c;
}

__global__ void union_fill_kernel(const int32_t* __restrict__ ap, const int32_t* __restrict__ ai,
                                  const int32_t* __restrict__ bp, const int32_t* __restrict__ bi, int64_t N,
                                  const int64_t* __restrict__ off64, int32_t* __restrict__ up,
                                  int32_t* __restrict__ ui) {
  const int64_t v = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (v > N) return;
  up[v] = (int32_t)off64[v];
  if (v == N) return;
  int i = ap[v], j = bp[v];
  const int ie = ap[v + 1], je = bp[v + 1];
  int64_t o = off64[v];
  while (i < ie || j < je) {
    const int x = i < ie ? ai[i] : 0x7fffffff, y = j < je ? bi[j] : 0x7fffffff;
    ui[o++] = min(x, y);
    i += x <= y ? 1 : 0;
    j += y <= x ? 1 : 0;
  }
}

}  // namespace

s3grl_status build_union_graph(s3grl_context* ctx, int64_t N, const int32_t* a_indptr, const int32_t* a_indices,
                               const int32_t* b_indptr, const int32_t* b_indices, int32_t* u_indptr,
                               int32_t** u_indices, int64_t* u_nnz) {
  Transient tmp{ctx, {}};
  void *cnt = nullptr, *off = nullptr, *ws = nullptr;
  S3GRL_TRY(ctx->arena.alloc((size_t)N * 4, &cnt));
  tmp.ptrs.push_back(cnt);
  S3GRL_TRY(ctx->arena.alloc((size_t)(N + 1) * 8, &off));
  tmp.ptrs.push_back(off);
  S3GRL_TRY(ctx->arena.alloc((size_t)scan_workspace_elems(N) * 8, &ws));
  tmp.ptrs.push_back(ws);
  const unsigned grid = (unsigned)((N + 1 + 255) / 256);
  hipLaunchKernelGGL(union_count_kernel, dim3(grid), dim3(256), 0, ctx->stream, a_indptr, a_indices, b_indptr,
                     b_indices, N, static_cast<int32_t*>(cnt));
  S3GRL_HIP_TRY(hipGetLastError());
  S3GRL_TRY(launch_scan_i32_to_i64(ctx, static_cast<int32_t*>(cnt), N, static_cast<int64_t*>(off),
                                   static_cast<int64_t*>(ws)));
  S3GRL_HIP_TRY(hipMemcpyAsync(ctx->h_scalars, static_cast<int64_t*>(off) + N, 8, hipMemcpyDeviceToHost, ctx->stream));
  S3GRL_HIP_TRY(hipStreamSynchronize(ctx->stream));
  *u_nnz = ctx->h_scalars[0];
  if (*u_nnz >= (int64_t)INT32_MAX) {
    set_last_error("the union of successors and predecessors needs 64-bit offsets");
    return S3GRL_ERR_GRAPH_TOO_LARGE;
  }
  void* ui = nullptr;
  S3GRL_TRY(ctx->arena.alloc((size_t)std::max<int64_t>(*u_nnz, 1) * 4, &ui));
  *u_indices = static_cast<int32_t*>(ui);
  hipLaunchKernelGGL(union_fill_kernel, dim3(grid), dim3(256), 0, ctx->stream, a_indptr, a_indices, b_indptr,
                     b_indices, N, static_cast<int64_t*>(off), u_indptr, *u_indices);
  S3GRL_HIP_TRY(hipGetLastError());
  S3GRL_HIP_TRY(hipStreamSynchronize(ctx->stream));   // tmp is released on return
  return S3GRL_OK;
}

s3grl_status build_degree_order(s3grl_context* ctx, s3grl_graph* g) {
  const int64_t N = g->num_nodes, nnz = g->nnz;
  Transient tmp{ctx, {}};
  auto talloc = [&](size_t bytes, void** out) -> s3grl_status {
    S3GRL_TRY(ctx->arena.alloc(std::max<size_t>(bytes, 16), out));
    tmp.ptrs.push_back(*out);
    return S3GRL_OK;
  };
  const int64_t big = std::max<int64_t>(N, std::max<int64_t>(nnz, 1));
  void *ka = nullptr, *kb = nullptr, *dn = nullptr, *off = nullptr, *ws = nullptr, *rt = nullptr;
  S3GRL_TRY(talloc((size_t)big * 8, &ka));
  S3GRL_TRY(talloc((size_t)big * 8, &kb));
  S3GRL_TRY(talloc((size_t)N * 4, &dn));
  S3GRL_TRY(talloc((size_t)(N + 1) * 8, &off));
  S3GRL_TRY(talloc((size_t)scan_workspace_elems(N) * 8, &ws));
  size_t rt_bytes = 0;
  uint64_t* keys_a = static_cast<uint64_t*>(ka);
  uint64_t* keys_b = static_cast<uint64_t*>(kb);
  S3GRL_HIP_TRY(rocprim::radix_sort_keys(nullptr, rt_bytes, keys_a, keys_b, (size_t)big, 0, 64, ctx->stream));
  S3GRL_TRY(talloc(rt_bytes, &rt));

  void* q = nullptr;
  S3GRL_TRY(ctx->arena.alloc((size_t)N * 4, &q));
  g->old_of_new = static_cast<int32_t*>(q);
  S3GRL_TRY(ctx->arena.alloc((size_t)N * 4, &q));
  g->new_of_old = static_cast<int32_t*>(q);
  S3GRL_TRY(ctx->arena.alloc((size_t)(N + 1) * 4, &q));
  g->r_indptr = static_cast<int32_t*>(q);
  S3GRL_TRY(ctx->arena.alloc((size_t)std::max<int64_t>(nnz, 1) * 4, &q));
  g->r_indices = static_cast<int32_t*>(q);

  const unsigned gn = (unsigned)((N + 255) / 256);
  hipLaunchKernelGGL(node_keys_kernel, dim3(gn), dim3(256), 0, ctx->stream, g->indptr, N,
                     (uint32_t)g->max_degree, keys_a);
  S3GRL_HIP_TRY(hipGetLastError());
  size_t bytes = rt_bytes;
  S3GRL_HIP_TRY(rocprim::radix_sort_keys(rt, bytes, keys_a, keys_b, (size_t)N, 0, 64, ctx->stream));
  hipLaunchKernelGGL(perm_kernel, dim3(gn), dim3(256), 0, ctx->stream, keys_b, g->indptr, N, g->old_of_new,
                     g->new_of_old, static_cast<int32_t*>(dn));
  S3GRL_HIP_TRY(hipGetLastError());
  hipLaunchKernelGGL(first_degree_le_kernel, dim3(1), dim3(1), 0, ctx->stream, keys_b, N, (uint32_t)g->max_degree, 2u,
                     ctx->d_scalars);
  S3GRL_HIP_TRY(hipGetLastError());
  S3GRL_HIP_TRY(hipMemcpyAsync(ctx->h_scalars, ctx->d_scalars, 8, hipMemcpyDeviceToHost, ctx->stream));
  S3GRL_HIP_TRY(hipStreamSynchronize(ctx->stream));
  g->deg_le2_from = (int32_t)ctx->h_scalars[0];
  S3GRL_TRY(launch_scan_i32_to_i64(ctx, static_cast<int32_t*>(dn), N, static_cast<int64_t*>(off),
                                   static_cast<int64_t*>(ws)));
  hipLaunchKernelGGL(offsets_to_i32_kernel, dim3((unsigned)((N + 1 + 255) / 256)), dim3(256), 0, ctx->stream,
                     static_cast<int64_t*>(off), N + 1, g->r_indptr);
  S3GRL_HIP_TRY(hipGetLastError());
  if (nnz > 0) {
    const unsigned ge = (unsigned)((nnz + 255) / 256);
    hipLaunchKernelGGL(edge_keys_kernel, dim3(ge), dim3(256), 0, ctx->stream, g->indptr, g->indices, N, nnz,
                       g->new_of_old, keys_a);
    S3GRL_HIP_TRY(hipGetLastError());
    bytes = rt_bytes;
    // sorted by (new row, new column): the rows of the permuted CSR, each ascending
    S3GRL_HIP_TRY(rocprim::radix_sort_keys(rt, bytes, keys_a, keys_b, (size_t)nnz, 0, 64, ctx->stream));
    hipLaunchKernelGGL(low_words_kernel, dim3(ge), dim3(256), 0, ctx->stream, keys_b, nnz, g->r_indices);
    S3GRL_HIP_TRY(hipGetLastError());
  }
  S3GRL_HIP_TRY(hipStreamSynchronize(ctx->stream));   // tmp is released on return
  return S3GRL_OK;
}

s3grl_status launch_link_order(s3grl_context* ctx, const int64_t* links, int64_t L, int64_t N,
                               const int32_t* indptr, int32_t* perm) {
  if (L == 0) return S3GRL_OK;
  Transient tmp{ctx, {}};
  void *ka = nullptr, *kb = nullptr, *va = nullptr, *rt = nullptr;
  S3GRL_TRY(ctx->arena.alloc((size_t)L * 8, &ka));
  tmp.ptrs.push_back(ka);
  S3GRL_TRY(ctx->arena.alloc((size_t)L * 8, &kb));
  tmp.ptrs.push_back(kb);
  S3GRL_TRY(ctx->arena.alloc((size_t)L * 4, &va));
  tmp.ptrs.push_back(va);
  hipLaunchKernelGGL(link_order_keys_kernel, dim3((unsigned)((L + 255) / 256)), dim3(256), 0, ctx->stream, links,
                     L, indptr, static_cast<uint64_t*>(ka), static_cast<int32_t*>(va));
  S3GRL_HIP_TRY(hipGetLastError());
  (void)N;
  size_t bytes = 0;
  S3GRL_HIP_TRY(rocprim::radix_sort_pairs(nullptr, bytes, static_cast<uint64_t*>(ka), static_cast<uint64_t*>(kb),
                                          static_cast<int32_t*>(va), perm, (size_t)L, 0, 64, ctx->stream));
  S3GRL_TRY(ctx->arena.alloc(std::max<size_t>(bytes, 16), &rt));
  tmp.ptrs.push_back(rt);
  S3GRL_HIP_TRY(rocprim::radix_sort_pairs(rt, bytes, static_cast<uint64_t*>(ka), static_cast<uint64_t*>(kb),
                                          static_cast<int32_t*>(va), perm, (size_t)L, 0, 64, ctx->stream));
  return S3GRL_OK;   // (tmp is released on return: stream-ordered reuse)
}

// (u64 key, i32 value) radix sort for the other translation units: rocPRIM's kernels are templates, and a
// second unit instantiating the same sort makes the host-side launch stubs of this one resolve to WHICHEVER
// unit's code object the linker kept — s3grl_graph_create then loaded the SoP unit's 2 MB to sort its node keys
s3grl_status sort_pairs_u64_i32_bytes(s3grl_context* ctx, size_t n, size_t* bytes) {
  *bytes = 0;
  S3GRL_HIP_TRY(rocprim::radix_sort_pairs(nullptr, *bytes, static_cast<uint64_t*>(nullptr), static_cast<uint64_t*>(nullptr),
                                          static_cast<int32_t*>(nullptr), static_cast<int32_t*>(nullptr), n, 0, 64,
                                          ctx->stream));
  return S3GRL_OK;
}

s3grl_status sort_pairs_u64_i32(s3grl_context* ctx, void* tmp, size_t bytes, uint64_t* keys_in, uint64_t* keys_out,
                                int32_t* vals_in, int32_t* vals_out, size_t n) {
  S3GRL_HIP_TRY(rocprim::radix_sort_pairs(tmp, bytes, keys_in, keys_out, vals_in, vals_out, n, 0, 64, ctx->stream));
  return S3GRL_OK;
}

s3grl_status launch_translate_links(s3grl_context* ctx, const s3grl_graph* g, const int64_t* links, int64_t L,
                                    int64_t* out) {
  if (L == 0) return S3GRL_OK;
  hipLaunchKernelGGL(translate_links_kernel, dim3((unsigned)((2 * L + 255) / 256)), dim3(256), 0, ctx->stream,
                     links, L, g->num_nodes, g->new_of_old, out);
  S3GRL_HIP_TRY(hipGetLastError());
  return S3GRL_OK;
}

s3grl_status launch_sort_hops(s3grl_context* ctx, const s3grl_plan* p, int32_t* nodes) {
  if (p->L == 0) return S3GRL_OK;
  const int W = (int)((p->graph->num_nodes + 31) / 32);
  const bool force_seg = getenv("S3GRL_FORCE_SEGSORT") != nullptr;   // test hook
  if ((size_t)W * 4 <= 65536 && !force_seg) {
    hipLaunchKernelGGL(sort_hops_kernel, dim3((unsigned)p->L), dim3(kSortBlock), (size_t)W * 4, ctx->stream,
                       p->node_off, p->lvl, W, nodes);
    S3GRL_HIP_TRY(hipGetLastError());
    return S3GRL_OK;
  }
  // Graphs whose N-bit bitmap passes the default 64 KiB of dynamic LDS (num_nodes > 524 288; one-hop
  // plans have no node limit): a segmented radix sort over the (link, hop) pieces, independent of N.
  const int64_t n = p->stats.extracted_nodes;
  if (n == 0) return S3GRL_OK;
  const int64_t nseg = p->L * kMaxLevels;
  Transient tmp{ctx, {}};
  void *seg = nullptr, *sorted = nullptr, *rt = nullptr;
  S3GRL_TRY(ctx->arena.alloc((size_t)(nseg + 1) * 8, &seg));
  tmp.ptrs.push_back(seg);
  S3GRL_TRY(ctx->arena.alloc((size_t)n * 4, &sorted));
  tmp.ptrs.push_back(sorted);
  int64_t* so = static_cast<int64_t*>(seg);
  hipLaunchKernelGGL(hop_segments_kernel, dim3((unsigned)((nseg + 1 + 255) / 256)), dim3(256), 0, ctx->stream,
                     p->node_off, p->lvl, p->L, so);
  S3GRL_HIP_TRY(hipGetLastError());
  size_t bytes = 0;
  S3GRL_TRY(segmented_sort_i32(ctx, nullptr, &bytes, nodes, static_cast<int32_t*>(sorted), (size_t)n, (unsigned)nseg, so));
  S3GRL_TRY(ctx->arena.alloc(std::max<size_t>(bytes, 16), &rt));
  tmp.ptrs.push_back(rt);
  S3GRL_TRY(segmented_sort_i32(ctx, rt, &bytes, nodes, static_cast<int32_t*>(sorted), (size_t)n, (unsigned)nseg, so));
  S3GRL_HIP_TRY(hipMemcpyAsync(nodes, sorted, (size_t)n * 4, hipMemcpyDeviceToDevice, ctx->stream));
  S3GRL_HIP_TRY(hipStreamSynchronize(ctx->stream));   // tmp is released on return
  return S3GRL_OK;
}

}  // namespace s3grl

S3GRL_DEFINE_TOUCH(relabel)
