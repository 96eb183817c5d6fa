// Feature operand of the gather: layout preparation and the sparse-row gather kernel, gfx950.
//
// The reference hands its operators a DENSE fp32 X (sgrl_link_pred.py:196, utils.py:83), but the
// feature matrices of the datasets it is run on are bag-of-words / TF-IDF / one-hot-degree rows
// (reference sgrl_link_pred.py:851,961-963; its own SoP flow calls `x.to_sparse()`,
// tuned_SIGN.py:93).  On request (flags = 2) `s3grl_features_create` also builds a
// per-column-tile CSR of X in HBM: (column-in-tile, value) pairs, 8 bytes each.  The sparse
// gather then moves 8·nnz(row) bytes per subgraph node instead of 4·F and accumulates into LDS;
// results are the same sums in the same node order (zeros contribute nothing), so parity is
// unaffected.  It is OPT-IN: at 10 % density it measured slower than the dense
// register-accumulator kernel of s3grl_gather.hip (see s3grl_features_create).
#include <algorithm>
#include <cstdlib>
#include <memory>

#include "s3grl_internal.hpp"
#include "s3grl_device.hpp"

namespace s3grl {
namespace {

constexpr int kTileCols = 512;   // feature columns one wave accumulates in LDS

struct Entry {
  int32_t col;   // column inside the tile
  float val;
};

// one wave per (row, tile): number of non-zeros
__global__ __launch_bounds__(256) void sp_count_kernel(const float* __restrict__ X, int64_t ldx,
                                                       int64_t N, int F, int tiles,
                                                       int32_t* __restrict__ cnt) {
  const int lane = threadIdx.x & 63;
  const int64_t item = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (item >= N * tiles) return;
  const int tile = (int)(item / N);
  const int64_t row = item - (int64_t)tile * N;
  const int c0 = tile * kTileCols, c1 = min(F, c0 + kTileCols);
  int n = 0;
  for (int c = c0 + lane; c - lane < c1; c += 64) {
    const bool nz = c < c1 && X[row * ldx + c] != 0.f;
    n += __popcll(__ballot(nz));
  }
  if (lane == 0) cnt[item] = n;
}

// one wave per (row, tile): ordered compaction of the non-zeros
__global__ __launch_bounds__(256) void sp_fill_kernel(const float* __restrict__ X, int64_t ldx,
                                                      int64_t N, int F, int tiles,
                                                      const int64_t* __restrict__ ptr,
                                                      Entry* __restrict__ ent) {
  const int lane = threadIdx.x & 63;
  const int64_t item = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (item >= N * tiles) return;
  const int tile = (int)(item / N);
  const int64_t row = item - (int64_t)tile * N;
  const int c0 = tile * kTileCols, c1 = min(F, c0 + kTileCols);
  int64_t o = ptr[item];
  for (int c = c0 + lane; c - lane < c1; c += 64) {
    const float v = c < c1 ? X[row * ldx + c] : 0.f;
    const bool nz = v != 0.f;
    const unsigned long long m = __ballot(nz);
    if (nz) {
      Entry e;
      e.col = c - c0;
      e.val = v;
      ent[o + __popcll(m & ((1ull << lane) - 1ull))] = e;
    }
    o += __popcll(m);
  }
}

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

// One wavefront owns one row pair and one column tile.  acc[kTileCols][K] float2 (rows a, b)
// lives in LDS; every non-zero (col, val) of a support row is one lane of one wave-instruction:
// K float2 read-modify-writes with wave-uniform (SGPR) coefficients.  Columns inside one row are
// distinct, so the lanes of one instruction touch distinct accumulators (plain ds_read/ds_write,
// no atomics: `ds_add_f32` measured 7x slower here), and the LDS executes one wave's
// instructions in order, so every accumulator sees its addends in support order:
// bit-reproducible.
template <int K>
__device__ __forceinline__ void scatter_entry(float2* acc, int col, float val, const float2* q) {
#pragma unroll
  for (int i = 0; i < K; ++i) {
    float2 a = acc[col * K + i];
    a.x += q[i].x * val;
    a.y += q[i].y * val;
    acc[col * K + i] = a;
  }
}

template <int K>
__global__ __launch_bounds__(256) void gather_sparse_kernel(
    const Job* __restrict__ jobs, int njobs, const int32_t* __restrict__ c_ids,
    const float* __restrict__ c_coef, const float* __restrict__ job_z,
    const int64_t* __restrict__ xptr, const Entry* __restrict__ ent, const float* __restrict__ X,
    int64_t ldx, int64_t N, int F, float* __restrict__ rows_out, float* __restrict__ prows) {
  extern __shared__ float2 lds2[];
  const int lane = threadIdx.x & 63;
  const int wave = threadIdx.x >> 6;
  const int jid = __builtin_amdgcn_readfirstlane(blockIdx.x * 4 + wave);
  if (jid >= njobs) return;
  float2* acc = lds2 + wave * (K * kTileCols);
  const int tile = blockIdx.y;
  const int col0 = tile * kTileCols;
  const int width = min(kTileCols, F - col0);
  const Job job = jobs[jid];
  if (job.split == 1) return;   // gathered piece by piece (the entries with split == 2)
  float* __restrict__ rows = job.split == 2 ? prows : rows_out;   // a piece writes partial rows
  const int cnt = __builtin_amdgcn_readfirstlane(job.support);
  const int32_t* __restrict__ ids = c_ids + job.ids_off;
  const float2* __restrict__ cf = reinterpret_cast<const float2*>(c_coef) + job.coef_off;
  const int64_t* __restrict__ xp = xptr + (int64_t)tile * N;

  {
    float4* a4 = reinterpret_cast<float4*>(acc);
    for (int k = lane; k < K * kTileCols / 2; k += 64) a4[k] = make_float4(0.f, 0.f, 0.f, 0.f);
  }

  constexpr int U = 4;
  int j = 0;
  for (; j + U <= cnt; j += U) {
    int64_t beg[U], end[U];
    Entry e[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int id = ids[j + u];
      beg[u] = xp[id];
      end[u] = xp[id + 1];
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      e[u].col = 0;
      e[u].val = 0.f;
      if (beg[u] + lane < end[u]) e[u] = ent[beg[u] + lane];
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      float2 q[K];
#pragma unroll
      for (int i = 0; i < K; ++i) q[i] = cf[(int64_t)i * cnt + j + u];
      if (beg[u] + lane < end[u]) scatter_entry<K>(acc, e[u].col, e[u].val, q);
      for (int64_t p = beg[u] + 64 + lane; p - lane < end[u]; p += 64) {  // rows with > 64 non-zeros
        if (p < end[u]) {
          const Entry t = ent[p];
          scatter_entry<K>(acc, t.col, t.val, q);
        }
      }
    }
  }
  for (; j < cnt; ++j) {
    const int id = ids[j];
    const int64_t b = xp[id], en = xp[id + 1];
    float2 q[K];
#pragma unroll
    for (int i = 0; i < K; ++i) q[i] = cf[(int64_t)i * cnt + j];
    for (int64_t p = b + lane; p - lane < en; p += 64) {
      if (p < en) {
        const Entry t = ent[p];
        scatter_entry<K>(acc, t.col, t.val, q);
      }
    }
  }

  // epilogue: same output contract as the dense kernel (incl. the folded reversed duplicate)
  const int Fp = F + 1;
  const int64_t rstride = (int64_t)(K + 1) * Fp;
  const int nrow = job.node_b >= 0 ? 2 : 1;
  const int ncopy = job.mirror_row >= 0 ? 2 : 1;
  for (int r = 0; r < nrow; ++r) {
    const int node = r == 0 ? job.node_a : job.node_b;
    const float* __restrict__ xr = X + (int64_t)node * ldx + col0;
    for (int m = 0; m < ncopy; ++m) {
      const int64_t orow = m == 0 ? job.out_row + r
                                  : job.mirror_row + (job.mirror_swap ? 1 - r : r);
      float* __restrict__ out = rows + orow * rstride;
      for (int c = lane; c < width; c += 64) {
        out[1 + col0 + c] = xr[c];
#pragma unroll
        for (int i = 0; i < K; ++i) {
          const float2 a = acc[c * K + i];
          out[(int64_t)(i + 1) * Fp + 1 + col0 + c] = r == 0 ? a.x : a.y;
        }
      }
      if (tile == 0 && lane <= K) {
        const float z = lane == 0 ? (float)(r == 0 ? job.z_a : job.z_b)
                                  : job_z[((int64_t)jid * K + (lane - 1)) * 2 + r];
        out[(int64_t)lane * Fp] = z;
      }
    }
  }
}

template <int K>
s3grl_status launch_sparse_k(s3grl_context* ctx, const s3grl_plan* p, const GatherView& v,
                             const s3grl_features* f, float* rows) {
  const unsigned gx = (unsigned)((v.njobs + 3) / 4);
  const size_t lds = (size_t)4 * 2 * K * kTileCols * sizeof(float);
  auto kern = gather_sparse_kernel<K>;
  S3GRL_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(kern),
                                    hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  hipLaunchKernelGGL(kern, dim3(gx, (unsigned)f->tiles), dim3(256), lds, ctx->stream, v.jobs,
                     (int)v.njobs, p->c_ids, p->c_coef, v.job_z, f->sp_ptr,
                     reinterpret_cast<const Entry*>(f->sp_ent), f->dense, f->ld, f->N, (int)f->F, rows, v.prows);
  S3GRL_HIP_TRY(hipGetLastError());
  return S3GRL_OK;
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

s3grl_status launch_gather_sparse(s3grl_context* ctx, const s3grl_plan* p, const GatherView& v,
                                  const s3grl_features* f, float* rows) {
  if (v.njobs == 0) return S3GRL_OK;
  switch (p->cfg.sign_k) {
    case 1: return launch_sparse_k<1>(ctx, p, v, f, rows);
    case 2: return launch_sparse_k<2>(ctx, p, v, f, rows);
    case 3: return launch_sparse_k<3>(ctx, p, v, f, rows);
    case 4: return launch_sparse_k<4>(ctx, p, v, f, rows);
    case 5: return launch_sparse_k<5>(ctx, p, v, f, rows);
    case 6: return launch_sparse_k<6>(ctx, p, v, f, rows);
    case 7: return launch_sparse_k<7>(ctx, p, v, f, rows);
    case 8: return launch_sparse_k<8>(ctx, p, v, f, rows);
    default:
      set_last_error("sign_k must be in 1..8");
      return S3GRL_ERR_INVALID_ARGUMENT;
  }
}

}  // namespace s3grl

using namespace s3grl;

extern "C" {

s3grl_status s3grl_features_destroy(s3grl_features* f) {
  if (!f) return S3GRL_OK;
  for (void* q : f->owned) f->ctx->arena.release(q);
  delete f;
  return S3GRL_OK;
}

s3grl_status s3grl_features_create(s3grl_context* ctx, const float* X, int64_t ldx, int64_t N,
                                   int64_t F, int32_t flags, s3grl_features** out) {
  if (!ctx || !out) return S3GRL_ERR_INVALID_ARGUMENT;
  if (!X) {
    set_last_error("node features are None");
    return S3GRL_ERR_NO_FEATURES;
  }
  if (N <= 0 || F <= 0 || ldx < F || F >= (int64_t)INT32_MAX / 2) return S3GRL_ERR_INVALID_ARGUMENT;
  S3GRL_HIP_TRY(hipSetDevice(ctx->device));
  std::unique_ptr<s3grl_features, s3grl_status (*)(s3grl_features*)> f(new s3grl_features(),
                                                                       s3grl_features_destroy);
  f->ctx = ctx;
  f->N = N;
  f->F = F;
  // dense operand: 16-byte aligned rows, ld a multiple of 4 floats >= F
  const int64_t need = (F + 3) / 4 * 4;
  if ((reinterpret_cast<uintptr_t>(X) & 15) == 0 && ldx % 4 == 0 && ldx >= need) {
    f->dense = X;   // borrowed: the caller keeps X alive while the handle lives
    f->ld = ldx;
  } else {
    void* p = nullptr;
    S3GRL_TRY(ctx->arena.alloc((size_t)N * need * 4, &p));
    f->owned.push_back(p);
    S3GRL_TRY(launch_copy_pad(ctx, X, ldx, N, F, static_cast<float*>(p), need));
    f->dense = static_cast<float*>(p);
    f->ld = need;
  }
  f->tiles = (int)((F + kTileCols - 1) / kTileCols);
  // Measured on MI355X (PubMed PoS K=3, 10 % dense X): sparse rows 52 ms vs dense rows 34 ms per
  // gather — the LDS read-modify-write chain at 12 waves/CU loses to the register-accumulator
  // kernel fed from the Infinity Cache.  Dense is therefore the default; sparse rows are opt-in.
  if (flags != 2 && !getenv("S3GRL_SPARSE_FEATURES")) {
    // flags 0 (auto): packed rows when at most half of the 16-byte chunks of X are non-zero
    // (PubMed TF-IDF: 33 %, Cora bag-of-words: 5 %); flags 4: always; flags 1: never
    const char* env = getenv("S3GRL_PACKED_FEATURES");
    const bool never = flags == 1 || (env && atoi(env) == 0);
    const bool always = flags == 4 || (env && atoi(env) == 1);
    if (!never) S3GRL_TRY(build_packed_rows(ctx, f.get(), always ? 2.0 : 0.5));
    *out = f.release();
    return S3GRL_OK;
  }
  // density
  const int64_t items = N * f->tiles;
  int32_t* cnt = nullptr;
  int64_t *ptr = nullptr, *ws = nullptr;
  void* p = nullptr;
  S3GRL_TRY(ctx->arena.alloc((size_t)items * 4, &p));
  cnt = static_cast<int32_t*>(p);
  std::vector<void*> tmp{p};
  struct Rel {
    s3grl_context* c;
    std::vector<void*>* v;
    ~Rel() {
      for (void* q : *v) c->arena.release(q);
    }
  } rel{ctx, &tmp};
  S3GRL_TRY(ctx->arena.alloc((size_t)(items + 1) * 8, &p));
  ptr = static_cast<int64_t*>(p);
  tmp.push_back(ptr);
  S3GRL_TRY(ctx->arena.alloc((size_t)scan_workspace_elems(items) * 8, &p));
  ws = static_cast<int64_t*>(p);
  tmp.push_back(ws);
  hipLaunchKernelGGL(sp_count_kernel, dim3((unsigned)((items + 3) / 4)), dim3(256), 0, ctx->stream,
                     f->dense, f->ld, N, (int)F, f->tiles, cnt);
  S3GRL_HIP_TRY(hipGetLastError());
  S3GRL_TRY(launch_scan_i32_to_i64(ctx, cnt, items, ptr, ws));
  S3GRL_HIP_TRY(hipMemcpyAsync(ctx->h_scalars, ptr + items, 8, hipMemcpyDeviceToHost, ctx->stream));
  S3GRL_HIP_TRY(hipStreamSynchronize(ctx->stream));
  f->nnz = ctx->h_scalars[0];
  const double density = (double)f->nnz / ((double)N * (double)F);
  (void)density;
  tmp.erase(std::find(tmp.begin(), tmp.end(), static_cast<void*>(ptr)));
  f->owned.push_back(ptr);
  f->sp_ptr = ptr;
  S3GRL_TRY(ctx->arena.alloc((size_t)std::max<int64_t>(f->nnz, 1) * sizeof(Entry), &p));
  f->owned.push_back(p);
  f->sp_ent = p;
  hipLaunchKernelGGL(sp_fill_kernel, dim3((unsigned)((items + 3) / 4)), dim3(256), 0, ctx->stream,
                     f->dense, f->ld, N, (int)F, f->tiles, ptr, static_cast<Entry*>(p));
  S3GRL_HIP_TRY(hipGetLastError());
  f->sparse = true;
  *out = f.release();
  return S3GRL_OK;
}

s3grl_status s3grl_features_info(const s3grl_features* f, int64_t* nnz, int32_t* is_sparse) {
  if (!f) return S3GRL_ERR_INVALID_ARGUMENT;
  if (nnz) *nnz = f->packed || !f->sparse ? f->pk_chunks : f->nnz;
  if (is_sparse) *is_sparse = f->sparse ? 1 : (f->packed ? 2 : 0);
  return S3GRL_OK;
}

}  // extern "C"

S3GRL_DEFINE_TOUCH(features)
