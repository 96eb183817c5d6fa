// extern "C" entry points of libs3grl_hip.so (see include/s3grl.h) and host orchestration.
#include <algorithm>
#include <chrono>
#include <cstdlib>
#include <cstring>
#include <memory>

#include "s3grl_internal.hpp"

namespace s3grl {

static thread_local std::string g_last_error;
void set_last_error(const std::string& msg) { g_last_error = msg; }

Arena::~Arena() {
  for (auto& kv : free_)
    if (!carved_.count(kv.second)) (void)hipFree(kv.second);
  for (auto& kv : live_)
    if (!carved_.count(kv.first)) (void)hipFree(kv.first);
  for (auto& sl : slabs_)
    if (sl.base) (void)hipFree(sl.base);
}

s3grl_status Arena::reserve(size_t bytes) {
  if (bytes < (1u << 20) || cached_ * 2 >= bytes) return S3GRL_OK;   // small, or the cache will serve most of it
  if (cur_slab_ >= 0 && slabs_[cur_slab_].size - slabs_[cur_slab_].used >= bytes) return S3GRL_OK;
  bytes = (bytes + (2u << 20) - 1) / (2u << 20) * (2u << 20);
  void* p = nullptr;
  if (hipMalloc(&p, bytes) != hipSuccess) {   // no slab: the allocations go one by one (and report for themselves)
    (void)hipGetLastError();
    cur_slab_ = -1;
    return S3GRL_OK;
  }
  held_ += bytes;
  slabs_.push_back(Slab{static_cast<char*>(p), bytes, 0, 0});
  cur_slab_ = (int)slabs_.size() - 1;
  return S3GRL_OK;
}

s3grl_status Arena::alloc(size_t bytes, void** out) {
  if (bytes == 0) bytes = 256;
  // 2 MiB granularity for big blocks so that sizes that wobble between steps hit the cache
  const size_t gran = bytes >= (1u << 20) ? (2u << 20) : 256;
  bytes = (bytes + gran - 1) / gran * gran;
  auto it = free_.lower_bound(bytes);
  if (it != free_.end() && it->first <= bytes * 2 + (4u << 20)) {
    *out = it->second;
    live_[it->second] = it->first;
    cached_ -= it->first;
    free_.erase(it);
    return S3GRL_OK;
  }
  if (cur_slab_ >= 0) {   // carve it out of the reserved block
    Slab& sl = slabs_[cur_slab_];
    if (sl.size - sl.used >= bytes) {
      void* p = sl.base + sl.used;
      sl.used += bytes;
      sl.blocks += 1;
      carved_[p] = cur_slab_;
      live_[p] = bytes;
      *out = p;
      return S3GRL_OK;
    }
  }
  void* p = nullptr;
  hipError_t e = hipMalloc(&p, bytes);
  if (e != hipSuccess) {
    // give cached blocks back and retry once
    (void)hipGetLastError();
    trim();
    e = hipMalloc(&p, bytes);
    if (e != hipSuccess) {
      set_last_error("hipMalloc(" + std::to_string(bytes) + "): " + hipGetErrorString(e));
      return S3GRL_ERR_OUT_OF_MEMORY;
    }
  }
  held_ += bytes;
  live_[p] = bytes;
  *out = p;
  return S3GRL_OK;
}

size_t Arena::trim() {
  size_t freed = 0;
  for (auto& kv : free_) {
    auto c = carved_.find(kv.second);
    if (c == carved_.end()) {
      (void)hipFree(kv.second);
      freed += kv.first;
    } else {   // part of a slab: forgotten; the slab goes when its last block has
      const int si = c->second;
      Slab& sl = slabs_[si];
      carved_.erase(c);
      if (--sl.blocks == 0 && sl.base) {
        if (si == cur_slab_) cur_slab_ = -1;
        (void)hipFree(sl.base);
        freed += sl.size;
        sl.base = nullptr;
      }
    }
  }
  free_.clear();
  cached_ = 0;
  held_ -= freed;
  return freed;
}

void Arena::release(void* p) {
  if (!p) return;
  auto it = live_.find(p);
  if (it == live_.end()) return;
  free_.emplace(it->second, p);
  cached_ += it->second;
  live_.erase(it);
}

namespace {

__global__ void indptr_to_i32_kernel(const int64_t* __restrict__ in, int64_t n,
                                     int32_t* __restrict__ out) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) out[i] = (int32_t)in[i];
}

// structure check of a caller's CSR: bit 0 indptr not monotone / not ending at nnz / not starting
// at 0, bit 1 a column id outside [0,N), bit 2 a row not strictly ascending (unsorted or duplicate)
__global__ void validate_csr_kernel(const int32_t* __restrict__ indptr,
                                    const int32_t* __restrict__ indices, int64_t n, int64_t nnz,
                                    int64_t* __restrict__ flags) {
  int bad = 0;
  for (int64_t v = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; v < n;
       v += (int64_t)gridDim.x * blockDim.x) {
    const int64_t b = indptr[v], e = indptr[v + 1];
    if (b > e || b < 0 || e > nnz || (v == 0 && b != 0) || (v == n - 1 && e != nnz)) {
      bad |= 1;
      continue;
    }
    int prev = -1;
    for (int64_t c = b; c < e; ++c) {
      const int u = indices[c];
      if (u < 0 || u >= n) bad |= 2;
      if (u <= prev) bad |= 4;
      prev = u;
    }
  }
  if (bad) atomicOr(reinterpret_cast<unsigned long long*>(flags), (unsigned long long)bad);
}

__global__ void max_degree_kernel(const int32_t* __restrict__ indptr, int64_t n, int64_t* out) {
  int m = 0;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n;
       i += (int64_t)gridDim.x * blockDim.x)
    m = max(m, indptr[i + 1] - indptr[i]);
  for (int o = 32; o > 0; o >>= 1) m = max(m, __shfl_xor(m, o));
  if ((threadIdx.x & 63) == 0) atomicMax(reinterpret_cast<long long*>(out), (long long)m);
}

__global__ void max_i32_kernel(const int32_t* __restrict__ in, int64_t n, int64_t* out) {
  int m = 0;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n;
       i += (int64_t)gridDim.x * blockDim.x)
    m = max(m, in[i]);
  for (int o = 32; o > 0; o >>= 1) m = max(m, __shfl_xor(m, o));
  if ((threadIdx.x & 63) == 0) atomicMax(reinterpret_cast<long long*>(out), (long long)m);
}

constexpr float kHubCostSlope = 12.1f;   // cost of a link at a cached hub per node of its subgraph (S3GRL_HUB_COST_SLOPE: fitting hook)

__global__ void link_cost_kernel(const int32_t* __restrict__ n_nodes, const int32_t* __restrict__ e_cap,
                                 const int64_t* __restrict__ x_cap, const int64_t* __restrict__ row_ptr, int64_t L,
                                 float hub_slope, float* __restrict__ cost) {
  const int64_t l = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (l >= L) return;
  const int n = n_nodes[l];
  // a reversed duplicate folded into its primary is not extracted (n == 0): it costs its output rows
  // and its share of the per-link bookkeeping
  if (n == 0) {
    cost[l] = 250.f;
    return;
  }
  // every row pair beyond the first (PoS Plus: the common-neighbour rows) is another K passes over the
  // subgraph and another gather job over its list
  const float pairs = (float)((row_ptr[l + 1] - row_ptr[l] + 1) / 2);
  // one-hop plans on big graphs: a link served from a cached hub neighbourhood (link_hub_kernel) costs
  // its pulls and its gather, both ~ n; the others the probes of their oriented rows
  if (x_cap && x_cap[l] >= 0) cost[l] = hub_slope * (float)n + 1380.f + (pairs - 1.f) * (float)n;
  else cost[l] = e_cap ? (float)e_cap[l] + 220.f + (pairs - 1.f) * (float)n
                       : pairs * (float)n + 400.f;
}

// job_n[job_off[l] + p] = n_nodes[l]: every row pair of a link gets a list of the link's size
__global__ void expand_job_n_kernel(const int32_t* __restrict__ n_nodes,
                                    const int64_t* __restrict__ job_off, int64_t L,
                                    int32_t* __restrict__ job_n) {
  const int64_t l = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (l >= L) return;
  const int n = n_nodes[l];
  for (int64_t j = job_off[l]; j < job_off[l + 1]; ++j) job_n[j] = n;
}

template <typename T>
s3grl_status arena_alloc(s3grl_context* ctx, size_t count, T** out, std::vector<void*>* owned) {
  void* p = nullptr;
  S3GRL_TRY(ctx->arena.alloc(count * sizeof(T), &p));
  *out = static_cast<T*>(p);
  if (owned) owned->push_back(p);
  return S3GRL_OK;
}

}  // namespace

s3grl_status ensure_side_streams(s3grl_context* ctx) {
  for (int i = 0; i < s3grl_context::kSide; ++i)
    if (!ctx->side[i]) S3GRL_HIP_TRY(hipStreamCreateWithFlags(&ctx->side[i], hipStreamNonBlocking));
  for (int i = 0; i <= s3grl_context::kSide; ++i)
    if (!ctx->side_ev[i]) S3GRL_HIP_TRY(hipEventCreateWithFlags(&ctx->side_ev[i], hipEventDisableTiming));
  return S3GRL_OK;
}

namespace {

s3grl_status record(s3grl_context* ctx, int idx) {
  if (!ctx->profiling) return S3GRL_OK;
  S3GRL_HIP_TRY(hipEventRecord(ctx->ev[idx], ctx->stream));
  return S3GRL_OK;
}

}  // namespace
}  // namespace s3grl

using namespace s3grl;

extern "C" {

int32_t s3grl_abi_version(void) { return S3GRL_ABI_VERSION; }

const char* s3grl_status_string(s3grl_status s) {
  switch (s) {
    case S3GRL_OK: return "ok";
    case S3GRL_ERR_INVALID_ARGUMENT: return "invalid argument";
    case S3GRL_ERR_NOT_IMPLEMENTED: return "not implemented";
    case S3GRL_ERR_NO_FEATURES: return "node features are required";
    case S3GRL_ERR_OUT_OF_MEMORY: return "out of device memory";
    case S3GRL_ERR_HIP: return "HIP runtime error";
    case S3GRL_ERR_NO_DEVICE: return "no gfx950 device";
    case S3GRL_ERR_GRAPH_TOO_LARGE: return "graph or subgraph too large for this build";
    case S3GRL_ERR_SELF_LINK: return "src == dst";
  }
  return "unknown status";
}

const char* s3grl_last_error(void) { return g_last_error.c_str(); }

s3grl_status s3grl_context_create(int32_t device, void* stream, s3grl_context** out) {
  if (!out) return S3GRL_ERR_INVALID_ARGUMENT;
  int count = 0;
  if (hipGetDeviceCount(&count) != hipSuccess || count <= 0 || device < 0 || device >= count) {
    set_last_error("no HIP device " + std::to_string(device));
    return S3GRL_ERR_NO_DEVICE;
  }
  S3GRL_HIP_TRY(hipSetDevice(device));
  auto* ctx = new s3grl_context();
  ctx->device = device;
  ctx->stream = static_cast<hipStream_t>(stream);
  for (auto& e : ctx->ev) S3GRL_HIP_TRY(hipEventCreate(&e));
  S3GRL_HIP_TRY(hipMalloc(&ctx->d_scalars, 64 * sizeof(int64_t)));
  S3GRL_HIP_TRY(hipHostMalloc(&ctx->h_scalars, 64 * sizeof(int64_t)));
  S3GRL_HIP_TRY(hipMalloc(&ctx->d_stats, kStatRows * kStatShards * kStatStride * sizeof(int64_t)));
  S3GRL_HIP_TRY(hipHostMalloc(&ctx->h_stats, kStatRows * kStatShards * kStatStride * sizeof(int64_t)));
  *out = ctx;
  return S3GRL_OK;
}

s3grl_status s3grl_context_preload(s3grl_context* ctx, uint32_t units, double* ms) {
  if (!ctx) return S3GRL_ERR_INVALID_ARGUMENT;
  S3GRL_HIP_TRY(hipSetDevice(ctx->device));
  struct Unit {
    void (*touch)();
    uint32_t group;
  };
  // group 1: what every PoS / PoS Plus plan at sign_k 3, 4 touches; 2: the other sign_k; 4: SoP and the pooling
  static const Unit units_all[] = {{touch_api, 1},      {touch_relabel, 1}, {touch_structure, 1}, {touch_balls, 1},
                                   {touch_features, 1}, {touch_packed, 1},  {touch_gather, 1},    {touch_csr, 1},
                                   {touch_hub, 1},      {touch_links_a, 2}, {touch_links_b, 2},   {touch_links_c, 2},
                                   {touch_sop, 4},      {touch_pool, 4}};
  int k = 0;
  for (const Unit& u : units_all) {
    const auto t0 = std::chrono::steady_clock::now();
    if (units & u.group) u.touch();
    if (ms) ms[k] = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
    ++k;
  }
  (void)hipGetLastError();
  return S3GRL_OK;
}

s3grl_status s3grl_context_destroy(s3grl_context* ctx) {
  if (!ctx) return S3GRL_OK;
  (void)hipSetDevice(ctx->device);
  (void)hipStreamSynchronize(ctx->stream);
  for (auto& e : ctx->ev)
    if (e) (void)hipEventDestroy(e);
  for (auto& st : ctx->side)
    if (st) {
      (void)hipStreamSynchronize(st);
      (void)hipStreamDestroy(st);
    }
  for (auto& e : ctx->side_ev)
    if (e) (void)hipEventDestroy(e);
  if (ctx->d_scalars) (void)hipFree(ctx->d_scalars);
  if (ctx->h_scalars) (void)hipHostFree(ctx->h_scalars);
  if (ctx->d_stats) (void)hipFree(ctx->d_stats);
  if (ctx->h_stats) (void)hipHostFree(ctx->h_stats);
  delete ctx;
  return S3GRL_OK;
}

// The gather launch is asynchronous: its two events are resolved lazily (next profiled call or
// s3grl_context_timings), so profiling adds no synchronisation of its own to a step.
static s3grl_status resolve_pending_gather(s3grl_context* ctx) {
  if (!ctx->gather_pending) return S3GRL_OK;
  S3GRL_HIP_TRY(hipEventSynchronize(ctx->ev[4]));
  float ms = 0;
  S3GRL_HIP_TRY(hipEventElapsedTime(&ms, ctx->ev[3], ctx->ev[4]));
  ctx->timings[2] += ms;
  ctx->timings[5] += 1.0;
  ctx->gather_pending = false;
  return S3GRL_OK;
}

// Totals the link kernels of the last plan summed (see s3grl_context::stats_owner)
static s3grl_status resolve_plan_stats(s3grl_context* ctx) {
  s3grl_plan* p = ctx->stats_owner;
  if (!p) return S3GRL_OK;
  ctx->stats_owner = nullptr;
  S3GRL_HIP_TRY(hipStreamSynchronize(ctx->stream));
  constexpr size_t kStatRow = (size_t)kStatShards * kStatStride;
  auto total = [&](int row) {
    int64_t t = 0;
    for (int k = 0; k < kStatShards; ++k) t += ctx->h_stats[row * kStatRow + (size_t)k * kStatStride];
    return t;
  };
  p->stats.total_sub_edges = total(0);
  p->stats.total_support = total(1);
  p->stats.total_volume = total(2);
  p->stats.oriented_entries = total(4);   // (link_hub_kernel takes its links' share back)
  p->stats.hub_links = total(5);
  p->stats.hub_read_bytes = total(6);
  p->stats.hub_endpoint_entries = total(7);
  p->stats.hub_nodes = total(8);
  p->stats_pending = false;
  if (ctx->profiling) {
    float ms = 0;
    S3GRL_HIP_TRY(hipEventElapsedTime(&ms, ctx->ev[0], ctx->ev[1]));
    ctx->timings[0] += ms;
    S3GRL_HIP_TRY(hipEventElapsedTime(&ms, ctx->ev[1], ctx->ev[2]));
    ctx->timings[1] += ms;
    ctx->timings[6] += 1.0;
  }
  return S3GRL_OK;
}

s3grl_status s3grl_context_set_profiling(s3grl_context* ctx, int32_t enabled) {
  if (!ctx) return S3GRL_ERR_INVALID_ARGUMENT;
  S3GRL_TRY(resolve_plan_stats(ctx));
  S3GRL_TRY(resolve_pending_gather(ctx));
  ctx->profiling = enabled != 0;
  for (double& t : ctx->timings) t = 0.0;   // (re)start accumulation
  return S3GRL_OK;
}

s3grl_status s3grl_context_timings(s3grl_context* ctx, double* what) {
  if (!ctx || !what) return S3GRL_ERR_INVALID_ARGUMENT;
  S3GRL_TRY(resolve_plan_stats(ctx));
  S3GRL_TRY(resolve_pending_gather(ctx));
  std::memcpy(what, ctx->timings, sizeof(ctx->timings));
  return S3GRL_OK;
}

// a caller's CSR (int64 indptr, int32 indices, device) copied into arena arrays with int32 offsets and
// validated on the device: indptr[0] == 0, monotone, ending at nnz; ids in [0,N); rows strictly ascending
static s3grl_status upload_csr(s3grl_context* ctx, int64_t num_nodes, const int64_t* indptr, const int32_t* indices,
                               int64_t nnz, const char* what, int32_t** d_indptr, int32_t** d_indices) {
  void* p = nullptr;
  S3GRL_TRY(ctx->arena.alloc((size_t)(num_nodes + 1) * 4, &p));
  *d_indptr = static_cast<int32_t*>(p);
  S3GRL_TRY(ctx->arena.alloc((size_t)std::max<int64_t>(nnz, 1) * 4, &p));
  *d_indices = static_cast<int32_t*>(p);
  const int64_t n1 = num_nodes + 1;
  hipLaunchKernelGGL(indptr_to_i32_kernel, dim3((unsigned)((n1 + 255) / 256)), dim3(256), 0,
                     ctx->stream, indptr, n1, *d_indptr);
  S3GRL_HIP_TRY(hipGetLastError());
  if (nnz)
    S3GRL_HIP_TRY(hipMemcpyAsync(*d_indices, indices, (size_t)nnz * 4, hipMemcpyDeviceToDevice, ctx->stream));
  S3GRL_HIP_TRY(hipMemsetAsync(ctx->d_scalars, 0, 16, ctx->stream));
  // the kernels index LDS bitmaps with the column ids and search rows by bisection: a malformed
  // CSR would corrupt memory or give silently wrong rows, so it is rejected here
  hipLaunchKernelGGL(validate_csr_kernel, dim3(512), dim3(256), 0, ctx->stream, *d_indptr, *d_indices,
                     num_nodes, nnz, ctx->d_scalars + 1);
  S3GRL_HIP_TRY(hipGetLastError());
  S3GRL_HIP_TRY(hipMemcpyAsync(ctx->h_scalars + 1, ctx->d_scalars + 1, 8, hipMemcpyDeviceToHost, ctx->stream));
  S3GRL_HIP_TRY(hipStreamSynchronize(ctx->stream));
  if (const int64_t bad = ctx->h_scalars[1]) {
    set_last_error(std::string("malformed ") + what + ":" + ((bad & 1) ? " indptr is not monotone from 0 to nnz;" : "") +
                   ((bad & 2) ? " a column id is outside [0, num_nodes);" : "") +
                   ((bad & 4) ? " a row is not strictly ascending (unsorted or duplicate entries);" : ""));
    return S3GRL_ERR_INVALID_ARGUMENT;
  }
  return S3GRL_OK;
}

static s3grl_status graph_max_degree(s3grl_context* ctx, s3grl_graph* g) {
  S3GRL_HIP_TRY(hipMemsetAsync(ctx->d_scalars, 0, 8, ctx->stream));
  hipLaunchKernelGGL(max_degree_kernel, dim3(256), dim3(256), 0, ctx->stream, g->indptr, g->num_nodes,
                     ctx->d_scalars);
  S3GRL_HIP_TRY(hipGetLastError());
  S3GRL_HIP_TRY(hipMemcpyAsync(ctx->h_scalars, ctx->d_scalars, 8, hipMemcpyDeviceToHost, ctx->stream));
  S3GRL_HIP_TRY(hipStreamSynchronize(ctx->stream));
  g->max_degree = (int32_t)ctx->h_scalars[0];
  return S3GRL_OK;
}

s3grl_status s3grl_graph_create(s3grl_context* ctx, int64_t num_nodes, const int64_t* indptr,
                                const int32_t* indices, int64_t nnz, s3grl_graph** out) {
  if (!ctx || !out || !indptr || num_nodes <= 0 || nnz < 0 || (nnz > 0 && !indices))
    return S3GRL_ERR_INVALID_ARGUMENT;
  if (nnz >= (int64_t)INT32_MAX || num_nodes >= (int64_t)INT32_MAX) {
    set_last_error("graph needs 64-bit edge offsets, not supported");
    return S3GRL_ERR_GRAPH_TOO_LARGE;
  }
  S3GRL_HIP_TRY(hipSetDevice(ctx->device));
  // everything this call allocates (two copies of the CSR, the permutations, the sort keys and rocPRIM's
  // temporaries), out of one block when the arena is cold
  S3GRL_TRY(ctx->arena.reserve((size_t)64 * (size_t)std::max<int64_t>(num_nodes, nnz) + (size_t)96 * num_nodes + (4u << 20)));
  auto* g = new s3grl_graph();
  g->ctx = ctx;
  g->num_nodes = num_nodes;
  g->nnz = nnz;
  s3grl_status st = upload_csr(ctx, num_nodes, indptr, indices, nnz, "CSR", &g->indptr, &g->indices);
  if (st == S3GRL_OK) st = graph_max_degree(ctx, g);
  // the same graph in descending degree order: what the link kernels walk
  if (st == S3GRL_OK) st = build_degree_order(ctx, g);
  if (st == S3GRL_OK && onehop_mode_for(g)) {
    st = build_forward_rows(ctx, g);
    if (st == S3GRL_OK) {
      s3grl_graph t = *g;   // and the oriented rows of the degree order
      t.indptr = g->r_indptr;
      t.indices = g->r_indices;
      t.fwd_indptr = nullptr;
      t.fwd_indices = nullptr;
      t.fwd_deg = nullptr;
      st = build_forward_rows(ctx, &t);
      g->r_fwd_indptr = t.fwd_indptr;
      g->r_fwd_indices = t.fwd_indices;
      g->r_fwd_deg = t.fwd_deg;
    }
    // and the neighbourhoods of its hubs, shared by all their links (s3grl_hub.hip)
    if (st == S3GRL_OK && g->r_fwd_indptr) st = build_hub_cache(ctx, g);
  }
  if (st != S3GRL_OK) {
    s3grl_graph_destroy(g);
    return st;
  }
  *out = g;
  return S3GRL_OK;
}

s3grl_status s3grl_graph_create_directed(s3grl_context* ctx, int64_t num_nodes, const int64_t* csr_indptr,
                                         const int32_t* csr_indices, const int64_t* csc_indptr,
                                         const int32_t* csc_indices, int64_t nnz, s3grl_graph** out) {
  if (!ctx || !out || !csr_indptr || !csc_indptr || num_nodes <= 0 || nnz < 0 ||
      (nnz > 0 && (!csr_indices || !csc_indices)))
    return S3GRL_ERR_INVALID_ARGUMENT;
  if (2 * nnz >= (int64_t)INT32_MAX || num_nodes >= (int64_t)INT32_MAX) {
    set_last_error("graph needs 64-bit edge offsets, not supported");
    return S3GRL_ERR_GRAPH_TOO_LARGE;
  }
  S3GRL_HIP_TRY(hipSetDevice(ctx->device));
  auto* g = new s3grl_graph();
  g->ctx = ctx;
  g->num_nodes = num_nodes;
  g->directed = true;
  g->arcs = nnz;
  s3grl_status st = upload_csr(ctx, num_nodes, csr_indptr, csr_indices, nnz, "CSR", &g->out_indptr, &g->out_indices);
  if (st == S3GRL_OK)
    st = upload_csr(ctx, num_nodes, csc_indptr, csc_indices, nnz, "CSC", &g->in_indptr, &g->in_indices);
  if (st == S3GRL_OK) {   // what the BFS follows: successors and predecessors together (utils.py:60-63)
    void* p = nullptr;
    st = ctx->arena.alloc((size_t)(num_nodes + 1) * 4, &p);
    g->indptr = static_cast<int32_t*>(p);
    if (st == S3GRL_OK)
      st = build_union_graph(ctx, num_nodes, g->out_indptr, g->out_indices, g->in_indptr, g->in_indices,
                             g->indptr, &g->indices, &g->nnz);
  }
  if (st == S3GRL_OK) st = graph_max_degree(ctx, g);
  // (no degree order, no oriented rows: directed plans walk the caller's ids on the bitmap flavour)
  if (st != S3GRL_OK) {
    s3grl_graph_destroy(g);
    return st;
  }
  *out = g;
  return S3GRL_OK;
}

s3grl_status s3grl_graph_destroy(s3grl_graph* g) {
  if (!g) return S3GRL_OK;
  g->ctx->arena.release(g->indptr);
  g->ctx->arena.release(g->indices);
  g->ctx->arena.release(g->fwd_indptr);
  g->ctx->arena.release(g->fwd_indices);
  g->ctx->arena.release(g->fwd_deg);
  g->ctx->arena.release(g->r_indptr);
  g->ctx->arena.release(g->r_indices);
  g->ctx->arena.release(g->new_of_old);
  g->ctx->arena.release(g->old_of_new);
  g->ctx->arena.release(g->r_fwd_indptr);
  g->ctx->arena.release(g->r_fwd_indices);
  g->ctx->arena.release(g->r_fwd_deg);
  g->ctx->arena.release(g->out_indptr);
  g->ctx->arena.release(g->out_indices);
  g->ctx->arena.release(g->in_indptr);
  g->ctx->arena.release(g->in_indices);
  release_hub_cache(g);
  release_ball_cache(g);
  delete g;
  return S3GRL_OK;
}

s3grl_status s3grl_plan_destroy(s3grl_plan* p) {
  if (!p) return S3GRL_OK;
  if (p->ctx->stats_owner == p) {   // nobody asked for the totals; the events of a profiled plan still count
    if (p->ctx->profiling) (void)resolve_plan_stats(p->ctx);
    p->ctx->stats_owner = nullptr;
  }
  for (void* q : p->owned) p->ctx->arena.release(q);
  delete p;
  return S3GRL_OK;
}

static s3grl_status plan_create_impl(s3grl_context* ctx, const s3grl_graph* g, const int64_t* links,
                                     int64_t L, const s3grl_cfg* cfg, const s3grl_node_sets* sets,
                                     s3grl_plan** out);

s3grl_status s3grl_plan_create(s3grl_context* ctx, const s3grl_graph* g, const int64_t* links,
                               int64_t L, const s3grl_cfg* cfg, s3grl_plan** out) {
  return plan_create_impl(ctx, g, links, L, cfg, nullptr, out);
}

s3grl_status s3grl_plan_create_sets(s3grl_context* ctx, const s3grl_graph* g, const int64_t* links,
                                    int64_t L, const s3grl_cfg* cfg, const s3grl_node_sets* sets,
                                    s3grl_plan** out) {
  if (!sets || !sets->set_ptr || sets->num_sets < 0 || sets->reserved != 0 ||
      (sets->per_link != 0 && sets->per_link != 1)) {
    set_last_error("s3grl_node_sets: set_ptr is required, per_link is 0 or 1, reserved 0");
    return S3GRL_ERR_INVALID_ARGUMENT;
  }
  return plan_create_impl(ctx, g, links, L, cfg, sets, out);
}

s3grl_status s3grl_walk_sets(s3grl_context* ctx, const s3grl_graph* g, const int64_t* starts,
                             int64_t num_starts, int32_t rw_m, int32_t rw_M, uint32_t seed,
                             int64_t* set_ptr, int32_t* set_nodes) {
  if (!ctx || !g || num_starts < 0 || (num_starts > 0 && (!starts || !set_nodes)) || !set_ptr)
    return S3GRL_ERR_INVALID_ARGUMENT;
  if (rw_m < 1 || rw_M < 1 || rw_m > 65535 || rw_M > 65535) {
    set_last_error("rw_m and rw_M must be in 1..65535");
    return S3GRL_ERR_INVALID_ARGUMENT;
  }
  S3GRL_HIP_TRY(hipSetDevice(ctx->device));
  return launch_walk_sets(ctx, g, starts, num_starts, rw_m, rw_M, seed, set_ptr, set_nodes);
}

static s3grl_status plan_create_impl(s3grl_context* ctx, const s3grl_graph* g, const int64_t* links,
                                     int64_t L, const s3grl_cfg* cfg, const s3grl_node_sets* sets,
                                     s3grl_plan** out) {
  if (!ctx || !g || !cfg || !out || L < 0 || (L > 0 && !links)) return S3GRL_ERR_INVALID_ARGUMENT;
  if (cfg->sign_k < 1 || cfg->sign_k > kMaxSignK || cfg->num_hops < 0 ||
      cfg->num_hops > kMaxLevels - 2) {
    set_last_error("sign_k must be in 1..8 and num_hops in 0..30");
    return S3GRL_ERR_INVALID_ARGUMENT;
  }
  if (cfg->mode != S3GRL_MODE_POS && cfg->mode != S3GRL_MODE_POS_PLUS && cfg->mode != S3GRL_MODE_SOP_RESTRICTED) {
    set_last_error("s3grl_plan_create handles PoS / PoS Plus / the num_hops-restricted SoP; use s3grl_sop_* for SoP");
    return S3GRL_ERR_INVALID_ARGUMENT;
  }
  const bool sop2 = cfg->mode == S3GRL_MODE_SOP_RESTRICTED;
  if (sop2 && (g->directed || sets || cfg->rw_m > 0 || cfg->max_nodes_per_hop > 0 ||
               (cfg->ratio_per_hop > 0.0 && cfg->ratio_per_hop < 1.0) || cfg->num_hops < 1 ||
               cfg->sign_k - 1 > cfg->num_hops)) {
    set_last_error("the num_hops-restricted SoP needs an undirected graph, plain k-hop balls and sign_k - 1 <= num_hops "
                   "(the operators' rows must stay inside the ball until the last step)");
    return S3GRL_ERR_NOT_IMPLEMENTED;
  }
  if ((cfg->directed != 0) != g->directed) {
    set_last_error(g->directed ? "the graph was created with s3grl_graph_create_directed: s3grl_cfg.directed must be 1"
                               : "s3grl_cfg.directed = 1 needs a graph made by s3grl_graph_create_directed (A and A_csc)");
    return S3GRL_ERR_INVALID_ARGUMENT;
  }
  const bool plus = cfg->mode == S3GRL_MODE_POS_PLUS;
  if (plus && cfg->strategy != S3GRL_STRATEGY_INTERSECTION) {
    set_last_error("check strat: only k_node_set_strategy='intersection' is usable");
    return S3GRL_ERR_NOT_IMPLEMENTED;
  }
  if (L >= (int64_t)INT32_MAX / 16) return S3GRL_ERR_INVALID_ARGUMENT;
  if (cfg->rw_m < 0 || cfg->rw_M < 0 || cfg->rw_m > 65535 || cfg->rw_M > 65535 ||
      (cfg->rw_m > 0) != (cfg->rw_M > 0)) {
    set_last_error("rw_m and rw_M must both be 0 or both be in 1..65535");
    return S3GRL_ERR_INVALID_ARGUMENT;
  }
  if (cfg->max_nodes_per_hop < 0 || !(cfg->ratio_per_hop >= 0.0)) {
    set_last_error("max_nodes_per_hop must be >= 0 and ratio_per_hop >= 0 (0 or >= 1: keep all)");
    return S3GRL_ERR_INVALID_ARGUMENT;
  }
  for (int r : cfg->reserved)
    if (r != 0) {
      set_last_error("s3grl_cfg.reserved must be zero");
      return S3GRL_ERR_INVALID_ARGUMENT;
    }
  if (sets) {
    if (cfg->rw_m > 0) {
      set_last_error("node sets replace the engine's own walks: rw_m / rw_M must be 0 with s3grl_plan_create_sets");
      return S3GRL_ERR_INVALID_ARGUMENT;
    }
    if (sets->num_sets != (sets->per_link ? L : g->num_nodes)) {
      set_last_error("s3grl_node_sets.num_sets must be num_links (per_link) or the graph's num_nodes");
      return S3GRL_ERR_INVALID_ARGUMENT;
    }
    if (sets->num_set_nodes < 0 || (sets->num_set_nodes > 0 && !sets->set_nodes)) return S3GRL_ERR_INVALID_ARGUMENT;
  }
  S3GRL_HIP_TRY(hipSetDevice(ctx->device));
  S3GRL_TRY(resolve_plan_stats(ctx));   // the statistics buffers are about to be reused
  const int K = cfg->sign_k;

  std::unique_ptr<s3grl_plan, s3grl_status (*)(s3grl_plan*)> plan(new s3grl_plan(),
                                                                  s3grl_plan_destroy);
  plan->ctx = ctx;
  plan->graph = g;
  plan->cfg = *cfg;
  plan->L = L;
  plan->stats = s3grl_plan_stats{};
  plan->stats.num_links = L;
  Transient tmp{ctx, {}};
  auto* own = &plan->owned;
  auto* tr = &tmp.ptrs;

  S3GRL_TRY(arena_alloc(ctx, (size_t)(L + 1), &plan->row_ptr, own));
  S3GRL_TRY(arena_alloc(ctx, (size_t)(L + 1), &plan->node_off, own));
  S3GRL_TRY(arena_alloc(ctx, (size_t)(L + 1), &plan->job_off, own));
  if (L == 0) {
    S3GRL_HIP_TRY(hipMemsetAsync(plan->row_ptr, 0, 8, ctx->stream));
    S3GRL_HIP_TRY(hipMemsetAsync(plan->node_off, 0, 8, ctx->stream));
    S3GRL_HIP_TRY(hipMemsetAsync(plan->job_off, 0, 8, ctx->stream));
    *out = plan.release();
    return S3GRL_OK;
  }
  S3GRL_TRY(record(ctx, 0));
  // the per-link arrays of the sizing pass (a cold arena: one block instead of two dozen)
  S3GRL_TRY(ctx->arena.reserve((size_t)L * (size_t)(160 + 4 * num_class_lists() + 4 * kMaxLevels) +
                               (size_t)mirror_table_slots(L) * 12 + (2u << 20)));
  S3GRL_TRY(arena_alloc(ctx, (size_t)L * 2, &plan->links, own));
  S3GRL_HIP_TRY(hipMemcpyAsync(plan->links, links, (size_t)L * 16, hipMemcpyDeviceToDevice,
                               ctx->stream));
  int32_t *n_rows, *n_jobs, *p_nodes, *lvl_max, *class_list, *class_count;
  int64_t* scan_ws;
  S3GRL_TRY(arena_alloc(ctx, (size_t)L, &plan->n_nodes, own));
  S3GRL_TRY(arena_alloc(ctx, (size_t)L, &n_rows, tr));
  S3GRL_TRY(arena_alloc(ctx, (size_t)L, &p_nodes, tr));
  S3GRL_TRY(arena_alloc(ctx, (size_t)L, &lvl_max, tr));
  S3GRL_TRY(arena_alloc(ctx, (size_t)L, &n_jobs, tr));
  S3GRL_TRY(arena_alloc(ctx, (size_t)L * num_class_lists(), &class_list, tr));
  S3GRL_TRY(arena_alloc(ctx, (size_t)3 * scan_workspace_elems(L), &scan_ws, tr));

  // d_scalars (int64 x 64): [0] err flag, [1] max n, [2] Σ edges, [3] Σ support, [4] Σ vol,
  // [5] max R, [6] Σ n counting folded links twice, [7] folded links, [8] node-set flags, [9..11] Σ n of the
  // extracted links, Σ R, row pairs, [16..23] debug stamps,
  // [32..47] class counts (int32 each, see classify_kernel)
  int64_t* ds = ctx->d_scalars;
  int64_t* hs = ctx->h_scalars;
  class_count = reinterpret_cast<int32_t*>(ds + 32);
  S3GRL_HIP_TRY(hipMemsetAsync(ds, 0, 64 * sizeof(int64_t), ctx->stream));
  constexpr size_t kStatRow = (size_t)kStatShards * kStatStride;   // int64 per total
  int64_t* st = ctx->d_stats;   // rows: 0 Σ edges, 1 Σ support, 2 Σ vol, 3 Σ n counting folded links twice
  S3GRL_HIP_TRY(hipMemsetAsync(st, 0, kStatRows * kStatRow * sizeof(int64_t), ctx->stream));
  auto stat_total = [&](int row) {
    int64_t t = 0;
    for (int k = 0; k < kStatShards; ++k) t += ctx->h_stats[row * kStatRow + (size_t)k * kStatStride];
    return t;
  };
  // ScaLed: the walk nodes of src and dst replace the BFS — the engine's own per-node walks, or the
  // node sets the caller cached (reference utils.py:94-104)
  const int rw_m = cfg->rw_m, rw_M = cfg->rw_M;
  WalkSets ws{};
  ws.num_nodes = (int)g->num_nodes;
  if (rw_m * rw_M > 0) {
    int32_t* rw_raw = nullptr;
    S3GRL_TRY(arena_alloc(ctx, (size_t)g->num_nodes * rw_m * rw_M, &rw_raw, tr));
    S3GRL_TRY(launch_random_walks(ctx, g, rw_m, rw_M, cfg->seed, rw_raw));
    ws.raw = rw_raw;
    ws.len = rw_m * rw_M;
  } else if (sets) {
    // checked BEFORE any kernel walks them (the kernels index LDS bitmaps with these ids and the
    // set bounds come from device memory): one extra round trip for set plans
    S3GRL_TRY(launch_validate_sets(ctx, sets->set_ptr, sets->set_nodes, sets->num_sets, sets->num_set_nodes,
                                   g->num_nodes, ds + 8));
    S3GRL_HIP_TRY(hipMemcpyAsync(hs + 8, ds + 8, 8, hipMemcpyDeviceToHost, ctx->stream));
    S3GRL_HIP_TRY(hipStreamSynchronize(ctx->stream));
    if (hs[8]) {
      set_last_error(std::string("malformed node sets:") +
                     ((hs[8] & 1) ? " set_ptr is not monotone from 0 to num_set_nodes;" : "") +
                     ((hs[8] & 2) ? " a node id is outside [0, num_nodes);" : ""));
      return S3GRL_ERR_INVALID_ARGUMENT;
    }
    ws.ptr = sets->set_ptr;
    ws.nodes = sets->set_nodes;
    ws.per_link = sets->per_link;
  }
  const bool walks = walks_on(ws);
  plan->walk_plan = walks;
  // reversed duplicates (both directions of a train edge) are folded into one extraction
  int32_t *partner = nullptr, *mirror_of = nullptr;
  // (a set per LINK need not be the set of the reversed link)
  const bool fold = !(cfg->flags & (S3GRL_FLAG_FULL_STATS | S3GRL_FLAG_NO_FOLD)) &&
                    !(sets && sets->per_link) && !getenv("S3GRL_NO_MIRROR");
  // per-hop sampling (utils.py:66-70; the reference's rw branch ignores it).  Its BFS keeps a
  // fourth bitmap, so the plan stays on the bitmap flavour of the visited set.
  HopSampling smp{walks ? 1.0 : cfg->ratio_per_hop, walks ? 0 : cfg->max_nodes_per_hop, cfg->seed};
  const bool sampling = hop_sampling_on(smp);
  if (fold) {
    uint64_t* keys;
    int32_t* vals;
    const int64_t slots = mirror_table_slots(L);
    S3GRL_TRY(arena_alloc(ctx, (size_t)slots, &keys, tr));
    S3GRL_TRY(arena_alloc(ctx, (size_t)slots, &vals, tr));
    S3GRL_TRY(arena_alloc(ctx, (size_t)L, &partner, tr));
    S3GRL_TRY(arena_alloc(ctx, (size_t)L, &mirror_of, tr));
    S3GRL_TRY(launch_find_mirrors(ctx, plan->links, L, g->num_nodes, keys, vals, slots, partner,
                                  mirror_of, ds + 7));
  }
  // count_kernel leaves every link's node list in HBM for link_kernel (one slot per link: lists
  // longer than the slot are walked again there).  4096 entries cover 97 % of PubMed's 3-hop
  // subgraphs; the slot shrinks when L slots would pass 6 GB.
  S3GRL_TRY(arena_alloc(ctx, (size_t)L * kMaxLevels, &plan->lvl, own));
  // one-hop plans on big graphs: sizes by intersecting the two sorted rows (no bitmaps, no limit on
  // num_nodes, no node-list hand-over: link_full_kernel merges the rows again)
  const bool onehop = cfg->num_hops == 1 && !walks && !sampling && onehop_mode_for(g) && g->fwd_indptr && !sop2;
  int32_t* stash = nullptr;
  int slot = 4096;
  if (const char* e = getenv("S3GRL_STASH_SLOT")) slot = std::max(0, atoi(e));   // test hook; 0 = off
  // Which flavour of the link kernel a link takes must not depend on the rest of the list (a link gives the same
  // bits in a sharded and an unsharded run), and the induced-CSR flavour needs the list in the stash: plans that
  // may use it keep the full slot up to 32 GiB of stash (2 M links) instead of shrinking it from 6 GiB on
  const bool csr_candidate = !sop2 && !onehop && !walks && !sampling && !g->directed && g->r_indptr && !getenv("S3GRL_NO_RELABEL") &&
                             csr_mode_for(g, cfg->num_hops, K, true, true);
  const int64_t stash_cap = csr_candidate ? ((int64_t)32 << 30) : ((int64_t)6 << 30);
  while (slot > 256 && (int64_t)L * slot * 4 > stash_cap) slot >>= 1;
  if (!onehop && slot > 0 && (int64_t)L * slot * 4 <= stash_cap)
    S3GRL_TRY(arena_alloc(ctx, (size_t)L * slot, &stash, tr));
  // Plain plans walk the graph in its degree order (s3grl_relabel.hip): links translated on the way
  // in, everything the plan hands out translated back by the link kernels.  Sampled and random-walk
  // plans draw by the caller's ids and stay on the original order.
  const bool relabel = !walks && !sampling && g->r_indptr && (!onehop || g->r_fwd_indptr) &&
                       !getenv("S3GRL_NO_RELABEL");
  s3grl_graph g_walk = *g;   // what count_kernel / link_kernel see
  const int64_t* links_walk = plan->links;
  if (relabel) {
    g_walk.indptr = g->r_indptr;
    g_walk.indices = g->r_indices;
    g_walk.fwd_indptr = g->r_fwd_indptr;
    g_walk.fwd_indices = g->r_fwd_indices;
    g_walk.fwd_deg = g->r_fwd_deg;
    int64_t* lr = nullptr;
    S3GRL_TRY(arena_alloc(ctx, (size_t)std::max<int64_t>(2 * L, 1), &lr, tr));
    S3GRL_TRY(launch_translate_links(ctx, g, plan->links, L, lr));
    links_walk = lr;
  }
  if (!relabel) g_walk.hub = HubCache{};   // the cache is the degree-ordered graph's
  plan->relabelled = relabel;
  // Long lists are worked on in the order of their links' higher-degree endpoint (launch_link_order):
  // the sizing pass and the link kernels (through the class lists) on any graph — PubMed's sizing pass
  // 1.32 -> 1.24 ms, the sort included — and on big graphs the gather too; small graphs sit in the
  // caches whatever the order and keep the longest-first gather order (PubMed: 7.07 vs 7.21 ms).
  int32_t* perm = nullptr;
  bool hub_order = L >= kHubOrderMinLinks;
  bool hub_gather = hub_order && g->num_nodes > kHubOrderMinNodes;
  if (const char* e = getenv("S3GRL_HUB_ORDER")) {   // test hook: 0 off, 1 on (gather too), 2 sizing pass + link kernels only
    hub_order = atoi(e) != 0;
    hub_gather = atoi(e) == 1;
  }
  if (hub_order) {
    S3GRL_TRY(arena_alloc(ctx, (size_t)L, &perm, tr));
    S3GRL_TRY(launch_link_order(ctx, links_walk, L, g->num_nodes, g_walk.indptr, perm));
  }
  int32_t* e_cap = nullptr;
  int64_t* x_cap = nullptr;   // one-hop plans: (LDS need << 32 | bound of the edges outside the hub's cache), -1: no hub
  bool balls = false;   // the sizing pass ran on the graph's cached balls (levels 1 .. num_hops are there)
  if (onehop) {
    S3GRL_TRY(arena_alloc(ctx, (size_t)L, &e_cap, own));
    plan->e_cap = e_cap;
    if (g_walk.hub.nh > 0) {
      S3GRL_TRY(arena_alloc(ctx, (size_t)L, &x_cap, own));
      plan->x_cap = x_cap;
    }
    S3GRL_TRY(launch_count1(ctx, &g_walk, links_walk, L, plus ? 1 : 0, K, partner, mirror_of, plan->n_nodes, p_nodes,
                            n_rows, n_jobs, lvl_max, e_cap, reinterpret_cast<int32_t*>(ds), st + 3 * kStatRow,
                            st + 4 * kStatRow, perm, x_cap));
  } else {
    // plain plans on graphs whose node balls fit: the sizing pass is bitmap arithmetic on the cached balls
    // of the two endpoints (s3grl_balls.hip) instead of a BFS per link
    if (relabel && !walks && !sampling)
      S3GRL_TRY(ensure_ball_cache(ctx, const_cast<s3grl_graph*>(g), cfg->num_hops, &balls));
    if (balls) {
      g_walk.balls = g->balls;
      S3GRL_TRY(launch_count_balls(ctx, &g_walk, links_walk, L, cfg->num_hops, plus ? 1 : 0, K, partner, mirror_of,
                                   plan->n_nodes, p_nodes, n_rows, n_jobs, lvl_max, reinterpret_cast<int32_t*>(ds),
                                   st + 3 * kStatRow, stash, slot, plan->lvl, perm));
    } else {
      S3GRL_TRY(launch_count(ctx, &g_walk, links_walk, L, cfg->num_hops, plus ? 1 : 0, K, ws,
                             partner, mirror_of,
                             plan->n_nodes, p_nodes, n_rows, n_jobs, lvl_max,
                             reinterpret_cast<int32_t*>(ds), st + 3 * kStatRow, smp, stash, slot, plan->lvl, perm));
    }
  }
  if (fold) S3GRL_TRY(launch_mirror_rows(ctx, partner, L, n_rows));
  // offsets of nodes / rows / row pairs, their maxima (ds[1], ds[5]) and totals (ds[9..11]) in one go
  S3GRL_TRY(launch_scan3(ctx, plan->n_nodes, n_rows, n_jobs, L, plan->node_off, plan->row_ptr, plan->job_off,
                         scan_ws, ds + 1, ds + 5, ds + 9));
  // Plans whose every operator reaches the whole subgraph, on graphs of the bitmap flavour with cached
  // balls: the links run on their induced LDS CSR (s3grl_csr.hip); its classes are cut by the exact entry
  // count, which a sizing kernel of its own produces once the node offsets are known
  const bool csr_plan = !sop2 && !(cfg->flags & S3GRL_FLAG_COUNT_ONLY) && stash != nullptr && balls &&
                        csr_mode_for(g, cfg->num_hops, K, balls, relabel && !walks && !sampling && !onehop);
  // PoS has no common-neighbour rows: the LDS classes are known without a round trip
  if (!plus && !csr_plan)
    S3GRL_TRY(launch_classify(ctx, g, 1, K, plan->n_nodes, p_nodes, lvl_max, L, class_count, class_list,
                              !sampling && !g->directed, e_cap, stash ? slot : 0, perm, x_cap, nullptr, true));
  S3GRL_HIP_TRY(hipMemcpyAsync(hs, ds, 56 * 8, hipMemcpyDeviceToHost, ctx->stream));   // scalars + class counts
  S3GRL_HIP_TRY(hipMemcpyAsync(ctx->h_stats + 3 * kStatRow, st + 3 * kStatRow, 2 * kStatRow * sizeof(int64_t),
                               hipMemcpyDeviceToHost, ctx->stream));
  S3GRL_HIP_TRY(hipStreamSynchronize(ctx->stream));
  hs[16] = hs[9];    // Σ n over the extracted links, Σ R, row pairs
  hs[17] = hs[10];
  hs[18] = hs[11];
  const int err = (int)(hs[0] & 0xffffffff);
  if (err == 2) {
    set_last_error("a link has src == dst");
    return S3GRL_ERR_SELF_LINK;
  }
  if (err == 1) {
    set_last_error("a link endpoint is outside [0, num_nodes)");
    return S3GRL_ERR_INVALID_ARGUMENT;
  }
  const int64_t max_n = hs[1], max_R = hs[5];
  const int64_t tot_n = hs[16], tot_rows = hs[17], njobs = hs[18];
  // (a relabelled PoS Plus plan sorts its common neighbours by the caller's ids inside cn[]: 3x)
  const int cn_cap = ((int)std::max<int64_t>(max_R - 2, 0) + 1) * ((relabel && plus) ? 3 : 1);
  uint16_t* csr_cnt = nullptr;
  int32_t* csr_e = nullptr;
  if (csr_plan) {
    S3GRL_TRY(arena_alloc(ctx, (size_t)std::max<int64_t>(tot_n, 1), &csr_cnt, tr));
    S3GRL_TRY(arena_alloc(ctx, (size_t)L, &csr_e, tr));
    S3GRL_TRY(launch_csr_count(ctx, &g_walk, links_walk, L, cfg->num_hops, plan->n_nodes, plan->node_off, plan->lvl,
                               stash, slot, perm, csr_cnt, csr_e));
  }
  if (plus || csr_plan) {
    S3GRL_TRY(launch_classify(ctx, g, plus ? cn_cap : 1, K, plan->n_nodes, p_nodes, lvl_max, L, class_count,
                              class_list, !sampling && !g->directed, e_cap, stash ? slot : 0, perm, x_cap, csr_e,
                              !plus));
    S3GRL_HIP_TRY(hipMemcpyAsync(hs + 32, ds + 32, 24 * 8, hipMemcpyDeviceToHost, ctx->stream));
    S3GRL_HIP_TRY(hipStreamSynchronize(ctx->stream));
  }
  int32_t class_count_host[48];   // [0..5] bitmap classes, [6] HBM-scratch class, [7] its max need,
                                  // [8..13] hash classes, [14..19] one-hop classes, [20] one-hop
                                  // class with its bit matrix in HBM, [21..25] cached-hub classes, [29..31] maxima,
                                  // [32..45] induced-CSR classes (see classify_kernel)
  std::memcpy(class_count_host, hs + 32, sizeof(class_count_host));
  if (getenv("S3GRL_DEBUG")) {
    fprintf(stderr, "[s3grl] L=%lld max_n=%lld classes:", (long long)L, (long long)max_n);
    for (int c = 0; c < num_class_lists(); ++c) fprintf(stderr, " %d", class_count_host[c]);
    fprintf(stderr, "\n");
  }
  if (cfg->flags & S3GRL_FLAG_COUNT_ONLY) {   // sizing pass: sizes and offsets only
    plan->stats.total_nodes = stat_total(3);
    plan->stats.folded_links = hs[7];
    plan->stats.extracted_nodes = tot_n;
    plan->stats.max_nodes = max_n;
    plan->stats.total_rows = tot_rows;
    plan->stats.num_row_pairs = 0;
    plan->njobs = 0;
    plan->stats.workspace_bytes = (int64_t)ctx->arena.bytes_held();
    *out = plan.release();
    return S3GRL_OK;
  }
  plan->stats.total_nodes = stat_total(3);  // algorithmic: a folded link counts like any other
  plan->stats.oriented_entries = stat_total(4);
  plan->stats.folded_links = hs[7];
  plan->stats.extracted_nodes = tot_n;
  plan->stats.max_nodes = max_n;
  plan->stats.total_rows = tot_rows;
  plan->stats.num_row_pairs = njobs;
  plan->njobs = njobs;
  {
    int32_t* hist;
    S3GRL_TRY(arena_alloc(ctx, (size_t)256, &hist, tr));
    S3GRL_TRY(arena_alloc(ctx, (size_t)std::max<int64_t>(njobs, 1), &plan->job_order, own));
    int32_t* pc = nullptr;
    int64_t* po = nullptr;
    if (perm && hub_gather) {
      S3GRL_TRY(arena_alloc(ctx, (size_t)L, &pc, tr));
      S3GRL_TRY(arena_alloc(ctx, (size_t)(L + 1), &po, tr));
    }
    S3GRL_TRY(launch_job_order(ctx, plan->n_nodes, n_jobs, plan->job_off, L, hist, plan->job_order,
                               hub_gather ? perm : nullptr, pc, po, scan_ws));
  }
  plan->hub_order = perm != nullptr && hub_gather;

  // coefficient lists: one per row pair, sized by the link's node count
  const int64_t* coef_off = nullptr;          // PoS: one pair per link, list at node_off[link]
  int64_t tot_coef = tot_n;
  if (plus) {
    int32_t* job_n;
    int64_t* co;
    S3GRL_TRY(arena_alloc(ctx, (size_t)scan_workspace_elems(njobs), &scan_ws, tr));
    S3GRL_TRY(arena_alloc(ctx, (size_t)njobs, &job_n, tr));
    S3GRL_TRY(arena_alloc(ctx, (size_t)(njobs + 1), &co, tr));
    hipLaunchKernelGGL(expand_job_n_kernel, dim3((unsigned)((L + 255) / 256)), dim3(256), 0,
                       ctx->stream, plan->n_nodes, plan->job_off, L, job_n);
    S3GRL_HIP_TRY(hipGetLastError());
    S3GRL_TRY(launch_scan_i32_to_i64(ctx, job_n, njobs, co, scan_ws));
    S3GRL_HIP_TRY(hipMemcpyAsync(hs + 20, co + njobs, 8, hipMemcpyDeviceToHost, ctx->stream));
    S3GRL_HIP_TRY(hipStreamSynchronize(ctx->stream));
    tot_coef = hs[20];
    coef_off = co;
  }
  // ids, coefficients, jobs, rows: sizes known now
  S3GRL_TRY(ctx->arena.reserve((size_t)std::max<int64_t>(tot_n, 1) * 4 + (size_t)std::max<int64_t>(tot_coef, 1) * 8 * K +
                               (size_t)njobs * (sizeof(Job) + 12 * K + 8) + (size_t)tot_rows * 8 + (16u << 20)));
  S3GRL_TRY(arena_alloc(ctx, (size_t)std::max<int64_t>(tot_n, 1), &plan->c_ids, own));
  S3GRL_TRY(arena_alloc(ctx, (size_t)std::max<int64_t>(tot_coef, 1) * 2 * K, &plan->c_coef, own));
  S3GRL_TRY(arena_alloc(ctx, (size_t)njobs, &plan->jobs, own));
  S3GRL_TRY(arena_alloc(ctx, (size_t)njobs * K * 2, &plan->job_z, own));
  S3GRL_TRY(arena_alloc(ctx, (size_t)std::max<int64_t>(njobs * K, 1), &plan->job_lim, own));
  S3GRL_TRY(arena_alloc(ctx, (size_t)tot_rows, &plan->row_nodes, own));
  // lists longer than split_t entries are gathered in pieces of 2^seg_shift (kSplitThreshold)
  int seg_shift = kSplitSegShift;
  if (const char* e = getenv("S3GRL_SPLIT_SEG_SHIFT")) seg_shift = std::min(20, std::max(4, atoi(e)));
  int split_t = kSplitThreshold;
  if (const char* e = getenv("S3GRL_SPLIT_T")) split_t = std::max(0, atoi(e));
  if (split_t > 0) split_t = std::max(split_t, 1 << seg_shift);   // a split job has at least two pieces
  if (max_n <= split_t) split_t = 0;                               // nothing to split in this plan
  plan->split_t = split_t;
  plan->seg_shift = seg_shift;
  S3GRL_TRY(record(ctx, 1));
  S3GRL_TRY(launch_links(ctx, &g_walk, links_walk, L, class_list,
                         class_count_host, cfg->num_hops,
                         plus ? 1 : 0, cn_cap, (cfg->flags & S3GRL_FLAG_FULL_STATS) ? 1 : 0, K, ws, p_nodes,
                         plan->node_off,
                         plan->row_ptr, plan->job_off, coef_off, mirror_of, plan->c_ids,
                         plan->c_coef, plan->jobs, plan->job_z, plan->job_lim, plan->row_nodes, plan->lvl, st,
                         st + kStatRow, st + 2 * kStatRow, smp, stash, slot, e_cap, max_n,
                         relabel ? g->old_of_new : nullptr, relabel ? g->new_of_old : nullptr, split_t, seg_shift,
                         x_cap, csr_cnt, csr_e, sop2 ? 1 : 0));
  if (split_t > 0) {   // pieces per job and their total (read with the statistics below)
    int32_t* pcnt;
    S3GRL_TRY(arena_alloc(ctx, (size_t)njobs, &pcnt, tr));
    S3GRL_TRY(arena_alloc(ctx, (size_t)(njobs + 1), &plan->piece_off, own));
    S3GRL_TRY(arena_alloc(ctx, (size_t)scan_workspace_elems(njobs), &scan_ws, tr));
    S3GRL_TRY(launch_split_count(ctx, plan->jobs, njobs, seg_shift, pcnt, plan->piece_off, scan_ws));
    S3GRL_HIP_TRY(hipMemcpyAsync(hs + 21, plan->piece_off + njobs, 8, hipMemcpyDeviceToHost, ctx->stream));
  }
  S3GRL_TRY(record(ctx, 2));
  S3GRL_HIP_TRY(hipMemcpyAsync(ctx->h_stats, st, 3 * kStatRow * sizeof(int64_t), hipMemcpyDeviceToHost,
                               ctx->stream));
  S3GRL_HIP_TRY(hipMemcpyAsync(ctx->h_stats + 4 * kStatRow, st + 4 * kStatRow, 5 * kStatRow * sizeof(int64_t),
                               hipMemcpyDeviceToHost, ctx->stream));
  plan->stats.workspace_bytes = (int64_t)ctx->arena.bytes_held();
  // The totals (Σ edges / support / vol) are read back when somebody asks for them: no wait for the
  // link kernels here, so the gather can be queued right behind them.  Plans with split jobs need
  // the piece count on the host now.
  plan->stats_pending = true;
  ctx->stats_owner = plan.get();
  if (split_t > 0 || getenv("S3GRL_DEBUG_STAMPS")) {
    S3GRL_TRY(resolve_plan_stats(ctx));
    if (split_t > 0 && hs[21] > 0) {
      const int64_t np = hs[21];
      plan->npieces = np;
      const size_t units = (size_t)(njobs + np);   // the jobs, then their pieces: one array of gather units
      S3GRL_TRY(arena_alloc(ctx, units, &plan->gjobs, own));
      S3GRL_TRY(arena_alloc(ctx, units * K * 2, &plan->g_z, own));
      S3GRL_TRY(arena_alloc(ctx, units * K, &plan->g_lim, own));
      S3GRL_TRY(arena_alloc(ctx, units, &plan->g_order, own));
      S3GRL_TRY(arena_alloc(ctx, (size_t)np, &plan->piece_job, own));
      S3GRL_HIP_TRY(hipMemcpyAsync(plan->g_z, plan->job_z, (size_t)njobs * K * 2 * sizeof(float),
                                   hipMemcpyDeviceToDevice, ctx->stream));
      S3GRL_HIP_TRY(hipMemsetAsync(plan->g_z + (size_t)njobs * K * 2, 0, (size_t)np * K * 2 * sizeof(float), ctx->stream));
      S3GRL_TRY(launch_split_fill(ctx, plan->jobs, plan->job_lim, njobs, plan->job_order, K, seg_shift,
                                  plan->piece_off, np, plan->gjobs, plan->g_lim, plan->g_order, plan->piece_job));
    }
    if (getenv("S3GRL_DEBUG_STAMPS")) {   // diagnostic build aid: cycles per link_kernel phase
      S3GRL_HIP_TRY(hipMemcpy(hs + 16, ds + 16, 8 * 8, hipMemcpyDeviceToHost));
      fprintf(stderr, "[s3grl] link_kernel phase cycles (sum over workgroups): bfs %lld  P/rank %lld  "
                      "deg %lld  ops<K %lld  last op %lld  tail %lld\n",
              (long long)hs[16], (long long)hs[17], (long long)hs[18], (long long)hs[19],
              (long long)hs[20], (long long)hs[21]);
      if (csr_plan)
        fprintf(stderr, "[s3grl] link_csr_kernel phase cycles (same slots): bitmap+ranks+offsets %lld  ids+rows %lld  "
                        "columns %lld  passes %lld   links %lld, mean LDS %lld B\n",
                (long long)hs[16], (long long)hs[17], (long long)hs[18], (long long)hs[19], (long long)hs[20],
                (long long)(hs[21] / std::max<int64_t>(hs[20], 1)));
      if (x_cap)
        fprintf(stderr, "[s3grl] link_hub_kernel phase cycles (same slots): nodes %lld  ids+rows %lld  walk %lld  "
                        "small csr %lld  passes %lld  tail %lld\n",
                (long long)hs[16], (long long)hs[17], (long long)hs[18], (long long)hs[19], (long long)hs[20],
                (long long)hs[21]);
      S3GRL_HIP_TRY(hipMemcpy(hs + 24, ds + 24, 8 * 8, hipMemcpyDeviceToHost));
      fprintf(stderr, "[s3grl] link_full_kernel, big class: merge %lld  hash+ids %lld  probes %lld  "
                      "csr+sort %lld  passes %lld   links %lld, columns in HBM for %lld\n",
              (long long)hs[24], (long long)hs[25], (long long)hs[26], (long long)hs[27], (long long)hs[28],
              (long long)hs[30], (long long)hs[31]);
    }
  }
  *out = plan.release();
  return S3GRL_OK;
}

s3grl_status s3grl_context_trim(s3grl_context* ctx, int64_t* released) {
  if (!ctx) return S3GRL_ERR_INVALID_ARGUMENT;
  S3GRL_HIP_TRY(hipSetDevice(ctx->device));
  S3GRL_HIP_TRY(hipStreamSynchronize(ctx->stream));   // cached blocks may still be read by queued work
  for (auto& st : ctx->side)
    if (st) S3GRL_HIP_TRY(hipStreamSynchronize(st));
  const size_t freed = ctx->arena.trim();
  if (released) *released = (int64_t)freed;
  return S3GRL_OK;
}

s3grl_status s3grl_plan_gather_traffic(s3grl_context* ctx, const s3grl_plan* p,
                                       const s3grl_features* f, int64_t* what) {
  if (!ctx || !p || !f || !what) return S3GRL_ERR_INVALID_ARGUMENT;
  S3GRL_HIP_TRY(hipSetDevice(ctx->device));
  for (int i = 0; i < 8; ++i) what[i] = 0;
  if (p->njobs == 0) return S3GRL_OK;
  if (f->N != p->graph->num_nodes) return S3GRL_ERR_INVALID_ARGUMENT;
  S3GRL_HIP_TRY(hipMemsetAsync(ctx->d_scalars, 0, 8 * sizeof(int64_t), ctx->stream));
  S3GRL_TRY(launch_gather_traffic(ctx, p, f, reinterpret_cast<unsigned long long*>(ctx->d_scalars)));
  S3GRL_HIP_TRY(hipMemcpyAsync(ctx->h_scalars, ctx->d_scalars, 8 * sizeof(int64_t), hipMemcpyDeviceToHost,
                               ctx->stream));
  S3GRL_HIP_TRY(hipStreamSynchronize(ctx->stream));
  for (int i = 0; i < 8; ++i) what[i] = ctx->h_scalars[i];
  return S3GRL_OK;
}

s3grl_status s3grl_plan_get_stats(const s3grl_plan* p, s3grl_plan_stats* out) {
  if (!p || !out) return S3GRL_ERR_INVALID_ARGUMENT;
  if (p->stats_pending && p->ctx->stats_owner == p) S3GRL_TRY(resolve_plan_stats(p->ctx));
  *out = p->stats;
  return S3GRL_OK;
}

s3grl_status s3grl_plan_total_rows(const s3grl_plan* p, int64_t* total_rows) {
  if (!p || !total_rows) return S3GRL_ERR_INVALID_ARGUMENT;
  *total_rows = p->stats.total_rows;
  return S3GRL_OK;
}

s3grl_status s3grl_plan_counts(const s3grl_plan* p, int64_t* what) {
  if (!p || !what) return S3GRL_ERR_INVALID_ARGUMENT;
  what[0] = p->L;
  what[1] = p->stats.total_rows;
  what[2] = p->stats.folded_links;
  what[3] = p->njobs;
  return S3GRL_OK;
}

s3grl_status s3grl_plan_link_cost(const s3grl_plan* p, float* cost) {
  if (!p || (!cost && p->L)) return S3GRL_ERR_INVALID_ARGUMENT;
  if (p->L == 0) return S3GRL_OK;
  hipLaunchKernelGGL(link_cost_kernel, dim3((unsigned)((p->L + 255) / 256)), dim3(256), 0, p->ctx->stream,
                     p->n_nodes, p->e_cap, p->x_cap, p->row_ptr, p->L,
                     getenv("S3GRL_HUB_COST_SLOPE") ? (float)atof(getenv("S3GRL_HUB_COST_SLOPE")) : kHubCostSlope, cost);
  S3GRL_HIP_TRY(hipGetLastError());
  return S3GRL_OK;
}

// measurement hook (tools/xcd_locality_probe.py; not in include/s3grl.h): replaces the order in which the
// gather takes the plan's jobs
s3grl_status s3grl_debug_set_job_order(s3grl_plan* p, const int32_t* order, int64_t n) {
  if (!p || !order || n != p->njobs) return S3GRL_ERR_INVALID_ARGUMENT;
  S3GRL_HIP_TRY(hipMemcpyAsync(p->job_order, order, (size_t)n * 4, hipMemcpyDeviceToDevice, p->ctx->stream));
  if (p->npieces)
    S3GRL_HIP_TRY(hipMemcpyAsync(p->g_order + p->npieces, order, (size_t)n * 4, hipMemcpyDeviceToDevice,
                                 p->ctx->stream));
  return S3GRL_OK;
}

s3grl_status s3grl_plan_row_ptr(const s3grl_plan* p, int64_t* row_ptr) {
  if (!p || !row_ptr) return S3GRL_ERR_INVALID_ARGUMENT;
  S3GRL_HIP_TRY(hipMemcpyAsync(row_ptr, p->row_ptr, (size_t)(p->L + 1) * 8, hipMemcpyDeviceToDevice,
                               p->ctx->stream));
  return S3GRL_OK;
}

s3grl_status s3grl_plan_row_nodes(const s3grl_plan* p, int64_t* row_nodes) {
  if (!p || (!row_nodes && p->stats.total_rows)) return S3GRL_ERR_INVALID_ARGUMENT;
  if (p->stats.total_rows)
    S3GRL_HIP_TRY(hipMemcpyAsync(row_nodes, p->row_nodes, (size_t)p->stats.total_rows * 8,
                                 hipMemcpyDeviceToDevice, p->ctx->stream));
  return S3GRL_OK;
}

s3grl_status s3grl_plan_export_subgraphs(const s3grl_plan* p, int64_t* node_ptr, int32_t* nodes,
                                         int8_t* dists) {
  if (!p || !node_ptr) return S3GRL_ERR_INVALID_ARGUMENT;
  hipStream_t st = p->ctx->stream;
  S3GRL_HIP_TRY(hipMemcpyAsync(node_ptr, p->node_off, (size_t)(p->L + 1) * 8,
                               hipMemcpyDeviceToDevice, st));
  const size_t n = (size_t)p->stats.extracted_nodes;
  if ((nodes || dists) && (p->cfg.flags & S3GRL_FLAG_COUNT_ONLY)) {
    set_last_error("a count-only plan holds sizes only (pass NULL for nodes and dists)");
    return S3GRL_ERR_INVALID_ARGUMENT;
  }
  if (nodes && n) S3GRL_HIP_TRY(hipMemcpyAsync(nodes, p->c_ids, n * 4, hipMemcpyDeviceToDevice, st));
  // the lists of a plan that walked the degree order are hop-major but not ascending inside a hop
  if (nodes && n && p->relabelled) S3GRL_TRY(launch_sort_hops(p->ctx, p, nodes));
  if (dists && n) S3GRL_TRY(launch_dists(p->ctx, p->node_off, p->lvl, p->L, dists));
  return S3GRL_OK;
}

static s3grl_status run_with(s3grl_context* ctx, const s3grl_plan* p, const s3grl_features* f,
                             float* rows) {
  if (p->cfg.flags & S3GRL_FLAG_COUNT_ONLY) {
    set_last_error("a count-only plan cannot be run");
    return S3GRL_ERR_INVALID_ARGUMENT;
  }
  if (p->njobs == 0) return S3GRL_OK;
  if (f->N != p->graph->num_nodes) {
    set_last_error("features have " + std::to_string(f->N) + " rows, the graph " +
                   std::to_string(p->graph->num_nodes) + " nodes");
    return S3GRL_ERR_INVALID_ARGUMENT;
  }
  if (ctx->profiling) S3GRL_TRY(resolve_pending_gather(ctx));
  S3GRL_TRY(record(ctx, 3));
  auto gather = [&](const GatherView& v) -> s3grl_status {
    if (f->sparse) return launch_gather_sparse(ctx, p, v, f, rows);
    if (f->packed) return launch_gather_packed(ctx, p, v, f, rows);
    return launch_gather(ctx, v, p->c_ids, p->c_coef, p->cfg.sign_k, f->dense, f->ld, f->F, rows, p->hub_order);
  };
  if (p->npieces == 0) {
    S3GRL_TRY(gather(GatherView{p->jobs, p->njobs, p->job_z, p->job_lim, p->job_order, nullptr}));
  } else {
    // ONE launch over the jobs and the pieces of the split ones (the pieces first: they belong to the
    // longest lists); the pieces write partial rows, from which the split jobs' rows are then added up
    Transient tmp{ctx, {}};
    void* q = nullptr;
    S3GRL_TRY(ctx->arena.alloc((size_t)p->npieces * 2 * (p->cfg.sign_k + 1) * (f->F + 1) * sizeof(float), &q));
    tmp.ptrs.push_back(q);
    float* prows = static_cast<float*>(q);
    S3GRL_TRY(gather(GatherView{p->gjobs, p->njobs + p->npieces, p->g_z, p->g_lim, p->g_order, prows}));
    S3GRL_TRY(launch_combine(ctx, p, prows, f->dense, f->ld, f->F, rows));
  }
  S3GRL_TRY(record(ctx, 4));
  if (ctx->profiling) ctx->gather_pending = true;
  return S3GRL_OK;
}

s3grl_status s3grl_run_features(s3grl_context* ctx, const s3grl_plan* p, const s3grl_features* f,
                                float* rows) {
  if (!ctx || !p || !f || (p->stats.total_rows && !rows)) return S3GRL_ERR_INVALID_ARGUMENT;
  S3GRL_HIP_TRY(hipSetDevice(ctx->device));
  return run_with(ctx, p, f, rows);
}

s3grl_status s3grl_run(s3grl_context* ctx, const s3grl_plan* p, const float* X, int64_t ldx,
                       int64_t F, float* rows) {
  if (!ctx || !p) return S3GRL_ERR_INVALID_ARGUMENT;
  if (!X) {
    set_last_error("node features are None");
    return S3GRL_ERR_NO_FEATURES;
  }
  if (F <= 0 || ldx < F || (p->stats.total_rows && !rows)) return S3GRL_ERR_INVALID_ARGUMENT;
  if (p->njobs == 0) return S3GRL_OK;
  // plain dense operand, no density analysis (that costs a host round trip): use
  // s3grl_features_create + s3grl_run_features to let the engine pick sparse rows
  s3grl_features* f = nullptr;
  S3GRL_TRY(s3grl_features_create(ctx, X, ldx, p->graph->num_nodes, F, 1, &f));
  const s3grl_status st = run_with(ctx, p, f, rows);
  s3grl_features_destroy(f);   // stream-ordered: a padded copy is only reused by later launches
  return st;
}

}  // extern "C"

S3GRL_DEFINE_TOUCH(api)
