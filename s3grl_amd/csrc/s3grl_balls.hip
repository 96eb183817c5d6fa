// Sizing pass from cached node balls (plain multi-hop plans on graphs whose balls fit: PubMed, Cora, USAir).
//
// The subgraph of a link is the union of the BFS balls of its two endpoints on the UNMASKED graph
// (reference utils.py:53-74: the target link is removed afterwards, utils.py:76-80), level d of the link =
// (ball_d(src) ∪ ball_d(dst)) minus (ball_{d-1}(src) ∪ ball_{d-1}(dst)).  A node is an endpoint of ~16
// links of a split, and count_kernel (s3grl_structure.hip) walks its ball again for every one of them —
// level-synchronous, with atomics into LDS bitmaps.  Here the balls ball_1 .. ball_h of EVERY node are
// built once per graph as N-bit bitmaps (level d = OR of level d-1 over the node's row: nnz * N/32 word
// operations per level; PubMed, 3 hops: 146 MB, 0.3 ms) and the sizing pass of a plan becomes word-parallel
// bitmap arithmetic: two rows of N/32 words per level, read once, a popcount scan, and the level's nodes
// written out in ascending id order — the same n, level ends, node lists (the hand-over to link_kernel),
// |P| and PoS Plus row counts as count_kernel's, bit for bit.  Sampled, random-walk, directed and
// one-hop-on-big-graph plans, and graphs whose balls would take more than 1 GiB, keep their own sizing pass.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdlib>

#include "s3grl_device.hpp"

namespace s3grl {
namespace {

constexpr int kBallWordsPerThread = 16;  // bitmap words a thread of count_balls_kernel keeps in registers, at most
constexpr int kBallWordsTarget = 5;      // ... and what the launcher aims for (PubMed, 617 words: 128 threads x 5
                                         // words 1.01 ms of sizing pass; 256 x 3: 1.13; 64 x 10: 1.25; 1024 x 1: 4.0)

// level 1: x and its stored neighbours (one thread per node, then one per arc)
__global__ void ball_self_kernel(int64_t N, int W, uint32_t* __restrict__ bits) {
  const int64_t x = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (x < N) atomicOr(&bits[x * W + (x >> 5)], 1u << (x & 31));
}

__global__ void ball_arcs_kernel(const int32_t* __restrict__ indptr, const int32_t* __restrict__ indices, int64_t N,
                                 int W, uint32_t* __restrict__ bits) {
  // eight lanes per node stride its row
  const int64_t gid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int64_t x = gid >> 3;
  if (x >= N) return;
  const int e1 = indptr[x + 1];
  for (int e = indptr[x] + (int)(gid & 7); e < e1; e += 8) {
    const int u = indices[e];
    atomicOr(&bits[x * W + (u >> 5)], 1u << (u & 31));
  }
}

// level d from level d-1: ball_d(x) = ball_{d-1}(x) ∪ ⋃_{u ∈ N(x)} ball_{d-1}(u); one workgroup per node
__global__ __launch_bounds__(256) void ball_next_kernel(const int32_t* __restrict__ indptr,
                                                        const int32_t* __restrict__ indices, int W,
                                                        const uint32_t* __restrict__ prev, uint32_t* __restrict__ cur) {
  const int64_t x = blockIdx.x;
  const int e0 = indptr[x], e1 = indptr[x + 1];
  for (int w = threadIdx.x; w < W; w += 256) {
    uint32_t acc = prev[x * W + w];
    for (int e = e0; e < e1; ++e) acc |= prev[(int64_t)indices[e] * W + w];
    cur[x * W + w] = acc;
  }
}

// The sizing pass of one link from the cached balls; outputs exactly those of count_kernel.  Every thread
// owns a contiguous run of at most CW bitmap words (ascending ids: one block scan per level places the
// level's nodes in order).  What is left of its time is the node-by-node write of the lists that
// link_kernel takes over (requesting all levels' words up front changed nothing: 1.00 ms either way).
template <int T, int CW>
__global__ __launch_bounds__(T) void count_balls_kernel(
    const int32_t* __restrict__ indptr, const int32_t* __restrict__ indices, int N, int W,
    const uint32_t* __restrict__ balls, int64_t level_stride, const int64_t* __restrict__ links, int hops, int plus,
    int K, const int32_t* __restrict__ partner, const int32_t* __restrict__ mirror_of, int32_t* __restrict__ n_nodes,
    int32_t* __restrict__ p_nodes, int32_t* __restrict__ n_rows, int32_t* __restrict__ n_jobs,
    int32_t* __restrict__ lvl_max, int32_t* __restrict__ err_flag, unsigned long long* __restrict__ tot_nodes_alg,
    int32_t* __restrict__ stash, int slot, int32_t* __restrict__ lvl_stash, const int32_t* __restrict__ perm) {
  __shared__ int sh[32];
  const int tid = threadIdx.x;
  const int l = perm ? perm[blockIdx.x] : (int)blockIdx.x;
  const int64_t s64 = links[2 * (int64_t)l], d64 = links[2 * (int64_t)l + 1];
  if (s64 < 0 || s64 >= N || d64 < 0 || d64 >= N || s64 == d64) {
    if (tid == 0) {
      atomicMax(err_flag, s64 == d64 ? 2 : 1);
      n_nodes[l] = 0;
      p_nodes[l] = 0;
      n_rows[l] = 0;
      n_jobs[l] = 0;
      lvl_max[l] = 0;
    }
    return;
  }
  if (partner && partner[l] >= 0) {  // reversed duplicate: the primary link does the work
    if (tid == 0) {
      n_nodes[l] = 0;
      p_nodes[l] = 0;
      n_rows[l] = 0;  // copied from the primary by mirror_rows_kernel
      n_jobs[l] = 0;
      lvl_max[l] = 0;
    }
    return;
  }
  const int src = (int)s64, dst = (int)d64;
  int32_t* stash_l = stash ? stash + (int64_t)l * slot : nullptr;
  int32_t* lvl_l = lvl_stash ? lvl_stash + (int64_t)l * kMaxLevels : nullptr;
  if (lvl_l && tid == 0) lvl_l[0] = 2;
  const int C = (W + T - 1) / T;   // <= CW (the launcher's choice)
  const int w0 = min(tid * C, W), w1 = min(w0 + C, W);
  uint32_t prev[CW];   // the link's nodes so far, as bits
#pragma unroll
  for (int c = 0; c < CW; ++c) {
    const int w = w0 + c;
    prev[c] = (w == (src >> 5) ? 1u << (src & 31) : 0u) | (w == (dst >> 5) ? 1u << (dst & 31) : 0u);
  }
  int n = 2, cum_a = 2, cum_b = 2, biggest = 2, nlev_seen = 1;
  for (int d = 1; d <= hops; ++d) {
    const uint32_t* __restrict__ bs = balls + (int64_t)(d - 1) * level_stride + (int64_t)src * W;
    const uint32_t* __restrict__ bd = balls + (int64_t)(d - 1) * level_stride + (int64_t)dst * W;
    uint32_t fresh[CW];
    int mine = 0;
#pragma unroll
    for (int c = 0; c < CW; ++c) {
      const int w = w0 + c;
      const uint32_t cur = w < w1 ? (bs[w] | bd[w]) : 0u;
      fresh[c] = cur & ~prev[c];
      prev[c] |= cur;
      mine += __popc(fresh[c]);
    }
    int added;
    int pos = n + block_excl_scan<T>(mine, sh, added);
    if (added == 0) break;   // (uniform: a block-wide total)
    if (stash_l) {           // the level in ascending id order, hop 1 onwards
      int sp = pos - 2;
#pragma unroll
      for (int c = 0; c < CW; ++c) {
        uint32_t w = fresh[c];
        while (w) {
          const int b = __ffs(w) - 1;
          w &= w - 1;
          if (sp < slot) stash_l[sp] = (w0 + c) * 32 + b;
          ++sp;
        }
      }
    }
    if (lvl_l && tid == 0) lvl_l[d] = n + added;
    n += added;
    nlev_seen = d + 1;
    biggest = max(biggest, added);
    if (d <= K - 1) cum_a = n;
    if (d <= K) cum_b = n;
  }
  int R = 2;
  if (plus && (tid >> 6) == 0) {
    const uint32_t* __restrict__ bs = balls + (int64_t)(hops - 1) * level_stride + (int64_t)src * W;
    const uint32_t* __restrict__ bd = balls + (int64_t)(hops - 1) * level_stride + (int64_t)dst * W;
    R = 2 + common_neighbours(
                indptr, indices, [&](int x) { return (((bs[x >> 5] | bd[x >> 5]) >> (x & 31)) & 1u) != 0u || x == src || x == dst; },
                src, dst, nullptr);
  }
  if (tid == 0) {
    if (lvl_l) lvl_l[kMaxLevels - 1] = nlev_seen;   // levels 0 .. nlev_seen-1 are complete
    n_nodes[l] = n;
    p_nodes[l] = R > 2 ? cum_b : cum_a;
    n_rows[l] = R;
    n_jobs[l] = (R + 1) / 2;
    lvl_max[l] = biggest;
    const unsigned long long mult = (mirror_of && mirror_of[l] >= 0) ? 2ull : 1ull;
    atomicAdd(stat_slot(tot_nodes_alg), mult * (unsigned long long)n);
  }
}

}  // namespace

void release_ball_cache(s3grl_graph* g) {
  g->ctx->arena.release(g->balls.bits);
  g->balls = BallCache{};
}

// balls 1 .. hops of every node of the degree-ordered graph, built level by level on first use
s3grl_status ensure_ball_cache(s3grl_context* ctx, s3grl_graph* g, int hops, bool* usable) {
  *usable = false;
  if (!g->r_indptr || g->directed || hops < 1 || getenv("S3GRL_NO_BALL_CACHE")) return S3GRL_OK;
  const int64_t N = g->num_nodes;
  const int W = (int)((N + 31) / 32);
  if ((W + 1023) / 1024 > kBallWordsPerThread) return S3GRL_OK;
  const int64_t bytes = (int64_t)hops * N * W * 4;
  int64_t cap = (int64_t)1 << 30;
  if (const char* e = getenv("S3GRL_BALL_CACHE_BYTES")) cap = atoll(e);   // test hook
  if (bytes > cap) return S3GRL_OK;
  BallCache& bc = g->balls;
  if (bc.hops < hops) {
    void* q = nullptr;
    S3GRL_TRY(ctx->arena.alloc((size_t)bytes, &q));
    uint32_t* bits = static_cast<uint32_t*>(q);
    const int64_t stride = N * W;
    int have = 0;
    if (bc.bits) {   // keep the levels already built
      S3GRL_HIP_TRY(hipMemcpyAsync(bits, bc.bits, (size_t)bc.hops * stride * 4, hipMemcpyDeviceToDevice, ctx->stream));
      have = bc.hops;
    }
    if (have == 0) {
      S3GRL_HIP_TRY(hipMemsetAsync(bits, 0, (size_t)stride * 4, ctx->stream));
      hipLaunchKernelGGL(ball_self_kernel, dim3((unsigned)((N + 255) / 256)), dim3(256), 0, ctx->stream, N, W, bits);
      hipLaunchKernelGGL(ball_arcs_kernel, dim3((unsigned)((N * 8 + 255) / 256)), dim3(256), 0, ctx->stream,
                         g->r_indptr, g->r_indices, N, W, bits);
      have = 1;
    }
    for (int d = have; d < hops; ++d)
      hipLaunchKernelGGL(ball_next_kernel, dim3((unsigned)N), dim3(256), 0, ctx->stream, g->r_indptr, g->r_indices, W,
                         bits + (int64_t)(d - 1) * stride, bits + (int64_t)d * stride);
    S3GRL_HIP_TRY(hipGetLastError());
    S3GRL_HIP_TRY(hipStreamSynchronize(ctx->stream));   // (the old block is released below)
    if (bc.bits) ctx->arena.release(bc.bits);
    bc.bits = bits;
    bc.hops = hops;
    bc.level_stride = stride;
  }
  *usable = true;
  return S3GRL_OK;
}

s3grl_status launch_count_balls(s3grl_context* ctx, const s3grl_graph* g, const int64_t* links, int64_t L, int hops,
                                int plus, int K, const int32_t* partner, const int32_t* mirror_of, int32_t* n_nodes,
                                int32_t* p_nodes, int32_t* n_rows, int32_t* n_jobs, int32_t* lvl_max,
                                int32_t* err_flag, int64_t* tot_nodes_alg, int32_t* stash, int slot,
                                int32_t* lvl_stash, const int32_t* perm) {
  if (L == 0) return S3GRL_OK;
  const int N = (int)g->num_nodes;
  const int W = (N + 31) / 32;
  const BallCache& bc = g->balls;
  auto launch = [&](auto kern, int T) {
    hipLaunchKernelGGL(kern, dim3((unsigned)L), dim3(T), 0, ctx->stream, g->indptr, g->indices, N, W, bc.bits,
                       bc.level_stride, links, hops, plus, K, partner, mirror_of, n_nodes, p_nodes, n_rows, n_jobs,
                       lvl_max, err_flag, reinterpret_cast<unsigned long long*>(tot_nodes_alg), stash, slot, lvl_stash,
                       perm);
  };
  // threads per link: the smallest workgroup that gives a thread at most `target` words (a thread keeps
  // its words in registers and writes their nodes out one by one)
  int target = kBallWordsTarget;
  if (const char* e = getenv("S3GRL_BALL_WORDS")) target = std::min(kBallWordsPerThread, std::max(1, atoi(e)));   // tuning hook
  int T = 64;
  while (T < 1024 && W > T * target) T *= 2;
  const int C = (W + T - 1) / T;   // <= kBallWordsPerThread (ensure_ball_cache checked W <= 1024 * that)
#define S3GRL_BALLS_CW(TT, CW) launch(count_balls_kernel<TT, CW>, TT)
#define S3GRL_BALLS_T(TT)                                                                                      \
  (C <= 1 ? S3GRL_BALLS_CW(TT, 1) : C <= 2 ? S3GRL_BALLS_CW(TT, 2) : C <= 3 ? S3GRL_BALLS_CW(TT, 3)             \
   : C <= 4 ? S3GRL_BALLS_CW(TT, 4) : C <= 5 ? S3GRL_BALLS_CW(TT, 5) : C <= 6 ? S3GRL_BALLS_CW(TT, 6)           \
   : C <= 8 ? S3GRL_BALLS_CW(TT, 8) : C <= 10 ? S3GRL_BALLS_CW(TT, 10) : C <= 12 ? S3GRL_BALLS_CW(TT, 12)       \
                                                                                  : S3GRL_BALLS_CW(TT, 16))
  switch (T) {
    case 64: S3GRL_BALLS_T(64); break;
    case 128: S3GRL_BALLS_T(128); break;
    case 256: S3GRL_BALLS_T(256); break;
    case 512: S3GRL_BALLS_T(512); break;
    default: S3GRL_BALLS_T(1024); break;
  }
#undef S3GRL_BALLS_T
#undef S3GRL_BALLS_CW
  S3GRL_HIP_TRY(hipGetLastError());
  return S3GRL_OK;
}

}  // namespace s3grl

S3GRL_DEFINE_TOUCH(balls)
