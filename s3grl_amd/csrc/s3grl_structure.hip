// Structure kernels of the PoS / PoS Plus path (feature independent), gfx950.
//
//   count_kernel     BFS to num_hops from {src,dst} on the unmasked graph -> n, vol(S), R
//   scan_*           multi-block exclusive scan int32 -> int64 offsets
//   classify_kernel  bins links by subgraph size (one launch of link_kernel per LDS class)
//   link_kernel      ONE workgroup per link, everything on-chip: BFS (N-bit LDS bitmaps),
//                    local ids = popcount rank, degrees of the masked induced subgraph,
//                    D^-1/2, common neighbours, and rows {a,b} of Â^1..Â^K by K pull steps
//                    r_i = r_{i-1}·Â over the GLOBAL CSR rows filtered through the bitmap.
//                    No induced sub-CSR is ever materialised, nothing but the final
//                    (node id, coefficient) lists leaves the CU.
//
// Restates (not translates) reference utils.py:47-85 (k_hop_subgraph), utils.py:33-44
// (neighbors) and tuned_SIGN.py:151-175 / :206-240: the reference materialises Â², …, Â^K of
// the whole n×n subgraph by SpGEMM and keeps R rows; here only those R rows are ever formed.
#include <cstdlib>

#include "s3grl_internal.hpp"
#include "s3grl_device.hpp"

namespace s3grl {
namespace {

// ---------------------------------------------------------------------------------------
// count: level-synchronous BFS on LDS bitmaps, one thread per frontier word (reference
// utils.py:53-74: `fringe = neighbors(fringe, A) - visited`, early break on an empty fringe).
// Outputs per link: n = |S|, R rows, and p = |P|, the hop-major prefix of S that can carry a
// non-zero entry of r_{K-1}: the only nodes link_kernel keeps propagation state for.
// Frontier list of count_kernel and its threads per link.  Most links are small (PubMed: two thirds
// have under 250 nodes) and pay for the uniform part of the kernel once per wavefront: two waves
// and a 1 024-entry list (12 KB of LDS per link on PubMed, 12 links per CU) measured 1.34 ms
// against 1.70 for four waves and 3 072 entries, 1.60 for one wave; graphs whose bitmaps leave
// room for only a few workgroups per CU get four waves per link.
constexpr int kCountList = 1024;

// EXT: the N-bit bitmaps live in an HBM slice per workgroup (`ext`, `ext_stride` words) instead of
// LDS — graphs of more than kMaxNodesLds nodes; same code, slower memory.  Launched over chunks of the
// link list (`link_base`), so that a bounded number of slices serves any number of links.
template <int G, int kCountBlock, bool EXT = false>
__global__ __launch_bounds__(kCountBlock) void count_kernel(
    const int32_t* __restrict__ indptr, const int32_t* __restrict__ indices, int N, int W,
    const int64_t* __restrict__ links, int hops, int plus, int K, int hubs,
    const WalkSets ws, const int32_t* __restrict__ partner, const int32_t* __restrict__ mirror_of,
    int32_t* __restrict__ n_nodes, int32_t* __restrict__ p_nodes, int32_t* __restrict__ n_rows,
    int32_t* __restrict__ n_jobs, int32_t* __restrict__ lvl_max, int32_t* __restrict__ err_flag,
    unsigned long long* __restrict__ tot_nodes_alg, HopSampling smp, int32_t* __restrict__ stash,
    int slot, int32_t* __restrict__ lvl_stash, const int32_t* __restrict__ cn_indptr,
    const int32_t* __restrict__ cn_indices, int full_p, uint32_t* __restrict__ ext, int64_t ext_stride,
    int link_base, const int32_t* __restrict__ perm) {
  extern __shared__ uint32_t smem[];
  const bool walks = walks_on(ws);
  // `cur` (the frontier as a bitmap, for levels too big for the frontier list) is only needed when
  // a level below `hops` is expanded again: with one hop (and for ScaLed walks) the launcher
  // leaves it out — a third of the LDS of a big graph
  const bool one_hop = hops <= 1 || walks;
  const int nbm = one_hop ? 2 : 3;
  // per-hop sampling: `vis` also holds discovered-but-dropped nodes, so the members of S get a
  // bitmap of their own (the launcher adds W words behind the list when sampling is on)
  const bool sampling = !walks && sampling_on(smp);
  uint32_t *vis, *cur, *nxt, *mem;
  int* sh;
  if constexpr (EXT) {
    uint32_t* bm = ext + (int64_t)blockIdx.x * ext_stride;
    vis = bm;
    cur = one_hop ? nullptr : bm + W;
    nxt = bm + (nbm - 1) * W;
    mem = sampling ? bm + nbm * W : nullptr;
    sh = reinterpret_cast<int*>(smem);
  } else {
    vis = smem;
    cur = one_hop ? nullptr : smem + W;
    nxt = smem + (nbm - 1) * W;
    sh = reinterpret_cast<int*>(smem + nbm * W);
    mem = sampling ? smem + nbm * W + 8 + kHubWords + kCountList : nullptr;
  }
  const int tid = threadIdx.x;
  const int l = perm ? perm[link_base + blockIdx.x] : link_base + blockIdx.x;
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
  int* hub = hubs ? sh + 8 : nullptr;
  int32_t* list = reinterpret_cast<int32_t*>(sh + 8 + kHubWords);   // frontier nodes of the levels < hops
  // The node list found here is left in HBM for link_kernel (hop 1 onwards, `slot` entries per
  // link, level ends in lvl_stash): it then rebuilds its LDS state from the list instead of
  // repeating the BFS.  A list longer than the slot is simply not used (link_kernel walks again).
  int32_t* stash_l = stash ? stash + (int64_t)l * slot : nullptr;
  int32_t* lvl_l = stash ? lvl_stash + (int64_t)l * kMaxLevels : nullptr;
  if (lvl_l && tid == 0) lvl_l[0] = 2;
  for (int t = tid; t < W; t += kCountBlock) {
    vis[t] = 0;
    if (cur) cur[t] = 0;
    nxt[t] = 0;
    if (mem) mem[t] = 0;
  }
  __syncthreads();
  if (tid == 0) {
    atomicOr(&vis[src >> 5], 1u << (src & 31));
    atomicOr(&vis[dst >> 5], 1u << (dst & 31));
    if (mem) {
      atomicOr(&mem[src >> 5], 1u << (src & 31));
      atomicOr(&mem[dst >> 5], 1u << (dst & 31));
    }
    list[0] = src;
    list[1] = dst;
  }
  hub_rows_clear<kCountBlock>(hub);
  __syncthreads();
  // The frontier is a node list (G lanes per node: no serial row walks) as long as the levels
  // below `hops` fit kCountList entries; beyond that it degrades to a bitmap walked one thread
  // per word.  cum_a / cum_b: nodes within K-1 / K hops (P for a row at hop 0 / hop 1).
  int n = 2, cum_a = 2, cum_b = 2, f0 = 0, f1 = 2, biggest = 2, nlev_seen = 1;
  bool use_list = true;
  if (walks) hops = 1;  // ScaLed: the "hop" is what the cached random walks of src and dst visited
  for (int d = 1; d <= hops; ++d) {
    if (walks) {
      for_each_walk_node(ws, src, dst, l, tid, kCountBlock, [&](int u) {
        const uint32_t m = 1u << (u & 31);
        const uint32_t old = atomicOr(&vis[u >> 5], m);
        if (!(old & m)) atomicOr(&nxt[u >> 5], m);
      });
    } else if (use_list) {
      walk_rows<kCountBlock, G, 2>(
          f0, f1, list, indptr, indices, hub,
          [&](RowAcc&, int, int u, bool valid) {
            if (valid) {
              const uint32_t m = 1u << (u & 31);
              const uint32_t old = atomicOr(&vis[u >> 5], m);
              if (!(old & m)) atomicOr(&nxt[u >> 5], m);
            }
          },
          [](RowAcc&, int, int) {});
    } else {
      for (int t = tid; t < W; t += kCountBlock) {
        uint32_t w = cur[t];
        while (w) {
          const int b = __ffs(w) - 1;
          w &= w - 1;
          const int v = t * 32 + b;
          const int e1 = indptr[v + 1];
          for (int e = indptr[v]; e < e1; ++e) {
            const int u = indices[e];
            const uint32_t m = 1u << (u & 31);
            const uint32_t old = atomicOr(&vis[u >> 5], m);
            if (!(old & m)) atomicOr(&nxt[u >> 5], m);
          }
        }
      }
    }
    __syncthreads();
    // every thread owns a contiguous run of bitmap words: one block scan per level
    const int C = (W + kCountBlock - 1) / kCountBlock;
    const int w0 = min(tid * C, W), w1 = min(w0 + C, W);
    int mine = 0;
    for (int t = w0; t < w1; ++t) mine += __popc(nxt[t]);
    int added;
    int pos = f1 + block_excl_scan<kCountBlock>(mine, sh, added);
    if (added == 0) break;
    if (sampling) {  // utils.py:66-70, the same draw link_kernel's BFS makes
      const int keep = hop_keep(smp, added);
      if (keep == 0) break;                     // utils.py:71-72
      if (keep < added) {
        sample_level<kCountBlock>(nxt, W, keep, smp.seed, min(src, dst), max(src, dst), sh);
        mine = 0;
        for (int t = w0; t < w1; ++t) mine += __popc(nxt[t]);
        pos = f1 + block_excl_scan<kCountBlock>(mine, sh, added);
      }
      for (int t = w0; t < w1; ++t) mem[t] |= nxt[t];
    }
    // enumerate the new level in ascending id order: into the LDS frontier list (levels below
    // `hops`, while they fit) and into the HBM stash
    const bool to_list = d < hops && use_list && f1 + added <= kCountList;
    if (to_list || stash_l) {
      int lp = pos;
      int sp = n - 2 + (pos - f1);
      for (int t = w0; t < w1; ++t) {
        uint32_t w = nxt[t];
        while (w) {
          const int b = __ffs(w) - 1;
          w &= w - 1;
          if (to_list) list[lp++] = t * 32 + b;
          if (stash_l && sp < slot) stash_l[sp] = t * 32 + b;
          ++sp;
        }
      }
    }
    if (lvl_l && tid == 0) lvl_l[d] = n + added;
    if (d < hops) {  // the new level is the next frontier
      if (to_list) {
        for (int t = w0; t < w1; ++t) nxt[t] = 0;
        f0 = f1;
        f1 += added;
      } else {
        use_list = false;
        for (int t = w0; t < w1; ++t) {
          cur[t] = nxt[t];
          nxt[t] = 0;
        }
      }
      __syncthreads();
    }
    n += added;
    nlev_seen = d + 1;
    biggest = max(biggest, added);
    if (d <= K - 1) cum_a = n;
    if (d <= K) cum_b = n;
  }
  int R = 2;
  if (sampling) __syncthreads();   // mem is complete
  const uint32_t* member = sampling ? mem : vis;
  if (plus && wave_id() == 0)
    R = 2 + common_neighbours(cn_indptr, cn_indices, [&](int x) { return test_bit(member, x); }, src, dst, nullptr);
  if (tid == 0) {
    if (lvl_l) lvl_l[kMaxLevels - 1] = nlev_seen;   // levels 0 .. nlev_seen-1 are complete
    n_nodes[l] = n;
    // (directed plans keep D^-1/2 — an OUT-degree — and state for every node of S: full_p)
    p_nodes[l] = full_p ? n : (R > 2 ? cum_b : cum_a);
    n_rows[l] = R;
    n_jobs[l] = (R + 1) / 2;
    lvl_max[l] = biggest;
    // algorithmic totals count a folded link as if it had been extracted on its own
    const unsigned long long mult = (mirror_of && mirror_of[l] >= 0) ? 2ull : 1ull;
    atomicAdd(stat_slot(tot_nodes_alg), mult * (unsigned long long)n);
  }
}

// ---------------------------------------------------------------------------------------
// ScaLed subgraphs (reference utils.py:86-150 with sign=True, cache built by create_rw_cache
// utils.py:425-443): M random walks of length m from every node, cached per NODE; the subgraph of
// a link is {src,dst} plus everything the walks of src and of dst visited.  torch_cluster's
// uniform walk is restated with a counter-based generator keyed by (seed, node, walk, step), so a
// node's walks are the same in every link and every kernel that re-derives the subgraph.
__device__ __forceinline__ uint32_t rw_random(uint32_t seed, uint32_t node, uint32_t walk,
                                              uint32_t step) {
  uint64_t x = ((uint64_t)seed << 32) ^ ((uint64_t)node * 0x9E3779B97F4A7C15ull) ^
               ((uint64_t)walk << 20) ^ step;
  x ^= x >> 33;
  x *= 0xff51afd7ed558ccdull;
  x ^= x >> 33;
  x *= 0xc4ceb9fe1a85ec53ull;
  x ^= x >> 33;
  return (uint32_t)(x >> 16);
}

__global__ void random_walks_kernel(const int32_t* __restrict__ indptr,
                                    const int32_t* __restrict__ indices, int64_t N, int m, int M,
                                    uint32_t seed, int32_t* __restrict__ raw /* [N, M*m] */) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= N * M) return;
  const int64_t v = i / M;
  const int w = (int)(i - v * M);
  int cur = (int)v;
  for (int s = 0; s < m; ++s) {
    const int b = indptr[cur], deg = indptr[cur + 1] - b;
    if (deg > 0) cur = indices[b + rw_random(seed, (uint32_t)v, (uint32_t)w, (uint32_t)s) % deg];
    raw[(v * M + w) * m + s] = cur;   // an isolated node stays where it is
  }
}

#ifndef S3GRL_LINKS_PART
// The cache reference utils.create_rw_cache builds (utils.py:425-443): for every start node the
// sorted unique nodes of its M walks of length m, the start itself included (torch_cluster's walk
// tensor begins with the start; torch.unique sorts).  Same walks as random_walks_kernel for the
// same (seed, node): a plan built on these sets equals the plan that draws the walks itself.
// One workgroup per start node: walks into LDS, bitonic sort, unique -> padded scratch + count.
constexpr int kWalkSetBlock = 256;
constexpr int kWalkSetMax = 8192;   // M * m + 1 rounded up to a power of two: 32 KiB of LDS

__global__ __launch_bounds__(kWalkSetBlock) void walk_sets_kernel(
    const int32_t* __restrict__ indptr, const int32_t* __restrict__ indices, int64_t N,
    const int64_t* __restrict__ starts, int m, int M, uint32_t seed, int P, int cap,
    int32_t* __restrict__ scratch /* [num_starts, cap] */, int32_t* __restrict__ count,
    int32_t* __restrict__ err_flag) {
  extern __shared__ uint32_t smem[];
  int32_t* e = reinterpret_cast<int32_t*>(smem);   // [P]
  __shared__ int sh[kWalkSetBlock / 64];
  const int tid = threadIdx.x;
  const int64_t i = blockIdx.x;
  const int64_t v64 = starts[i];
  if (v64 < 0 || v64 >= N) {
    if (tid == 0) {
      atomicMax(err_flag, 1);
      count[i] = 0;
    }
    return;
  }
  const int v = (int)v64;
  for (int t = tid; t < P; t += kWalkSetBlock) e[t] = t == 0 ? v : 0x7fffffff;
  __syncthreads();
  for (int w = tid; w < M; w += kWalkSetBlock) {
    int cur = v;
    for (int st = 0; st < m; ++st) {
      const int b = indptr[cur], deg = indptr[cur + 1] - b;
      if (deg > 0) cur = indices[b + rw_random(seed, (uint32_t)v, (uint32_t)w, (uint32_t)st) % deg];
      e[1 + w * m + st] = cur;
    }
  }
  __syncthreads();
  for (int k = 2; k <= P; k <<= 1)
    for (int j = k >> 1; j > 0; j >>= 1) {
      for (int t = tid; t < P; t += kWalkSetBlock) {
        const int u = t ^ j;
        if (u > t) {
          const int a = e[t], b = e[u];
          if ((a > b) == ((t & k) == 0)) {
            e[t] = b;
            e[u] = a;
          }
        }
      }
      __syncthreads();
    }
  // unique: thread-contiguous runs, one block scan
  const int C = (P + kWalkSetBlock - 1) / kWalkSetBlock;
  const int t0 = min(tid * C, P), t1 = min(t0 + C, P);
  int mine = 0;
  for (int t = t0; t < t1; ++t) mine += (e[t] != 0x7fffffff && (t == 0 || e[t] != e[t - 1])) ? 1 : 0;
  int total;
  int pos = block_excl_scan<kWalkSetBlock>(mine, sh, total);
  for (int t = t0; t < t1; ++t)
    if (e[t] != 0x7fffffff && (t == 0 || e[t] != e[t - 1])) scratch[i * cap + pos++] = e[t];
  if (tid == 0) count[i] = total;
}

__global__ void walk_sets_compact_kernel(const int32_t* __restrict__ scratch, int cap,
                                         const int64_t* __restrict__ set_ptr, int32_t* __restrict__ set_nodes) {
  const int64_t i = blockIdx.x;
  const int64_t o = set_ptr[i];
  const int n = (int)(set_ptr[i + 1] - o);
  for (int t = threadIdx.x; t < n; t += blockDim.x) set_nodes[o + t] = scratch[i * cap + t];
}

__global__ void validate_sets_kernel(const int64_t* __restrict__ set_ptr, const int32_t* __restrict__ set_nodes,
                                     int64_t num_sets, int64_t total, int64_t N, int64_t* __restrict__ flags) {
  int bad = 0;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < num_sets;
       i += (int64_t)gridDim.x * blockDim.x) {
    const int64_t b = set_ptr[i], e = set_ptr[i + 1];
    if (b > e || b < 0 || e > total || (i == 0 && b != 0) || (i == num_sets - 1 && e != total)) {
      bad |= 1;
      continue;
    }
    for (int64_t c = b; c < e; ++c) {
      const int u = set_nodes[c];
      if (u < 0 || u >= N) bad |= 2;
    }
  }
  if (bad) atomicOr(reinterpret_cast<unsigned long long*>(flags), (unsigned long long)bad);
}

// ---- split jobs (Job::split, kSplitThreshold) ---------------------------------------------------
// pieces per job: ceil(support / 2^seg_shift) for a split job, none otherwise
__global__ void split_count_kernel(const Job* __restrict__ jobs, int64_t njobs, int seg_shift,
                                   int32_t* __restrict__ cnt) {
  const int64_t j = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= njobs) return;
  const Job job = jobs[j];
  cnt[j] = job.split ? (job.support + (1 << seg_shift) - 1) >> seg_shift : 0;
}

// piece s of job j as a gather unit of its own: list entries [s·SEG, min((s+1)·SEG, support)), its
// coefficient block (laid out by the link kernel), operator reach clamped to the piece, and two
// rows of the partial-row scratch as output.  No mirror, no label column: combine_kernel does those.
// The pieces sit behind the plan's jobs in ONE array of gather units (gjobs[njobs + q]); the launch
// order starts with the pieces (they belong to the longest lists of the plan) and goes on with the
// plan's own largest-first order.
__global__ void split_fill_kernel(const Job* __restrict__ jobs, const int32_t* __restrict__ job_lim,
                                  const int32_t* __restrict__ job_order, int64_t njobs, int K, int seg_shift,
                                  const int64_t* __restrict__ piece_off, int64_t npieces, Job* __restrict__ gjobs,
                                  int32_t* __restrict__ g_lim, int32_t* __restrict__ g_order,
                                  int32_t* __restrict__ piece_job) {
  const int64_t j = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= njobs) return;
  const Job job = jobs[j];
  gjobs[j] = job;
  for (int i = 0; i < K; ++i) g_lim[j * K + i] = job_lim[j * K + i];
  g_order[npieces + j] = job_order[j];
  const int64_t p0 = piece_off[j], p1 = piece_off[j + 1];
  for (int64_t q = p0; q < p1; ++q) {
    const int s0 = (int)(q - p0) << seg_shift;
    const int len = min(1 << seg_shift, job.support - s0);
    Job g = job;
    g.coef_off = job.coef_off + (int64_t)s0 * K;
    g.ids_off = job.ids_off + s0;
    g.out_row = 2 * q;
    g.support = len;
    g.mirror_row = -1;
    g.mirror_swap = 0;
    g.split = 2;
    gjobs[njobs + q] = g;
    for (int i = 0; i < K; ++i) g_lim[(njobs + q) * K + i] = min(max(job_lim[j * K + i] - s0, 0), len);
    g_order[q] = (int32_t)(njobs + q);
    piece_job[q] = (int32_t)j;
  }
}

// Rows of a split job: operator i+1 = Σ over its pieces (ascending, in f64) of the pieces' partial
// rows; operator 0 = [z | X[node]]; label column from job_z; the folded reversed link gets the same
// rows swapped.  Launched over the pieces: the workgroup of a job's FIRST piece does the job's row.
__global__ __launch_bounds__(256) void combine_kernel(
    const Job* __restrict__ jobs, const int64_t* __restrict__ piece_off, const int32_t* __restrict__ piece_job,
    const float* __restrict__ job_z, int K, const float* __restrict__ prows, const float* __restrict__ X,
    int64_t ldx, int F, float* __restrict__ rows) {
  const int64_t j = piece_job[blockIdx.x];
  const int r = blockIdx.y;
  const int64_t p0 = piece_off[j], p1 = piece_off[j + 1];
  if (p0 != (int64_t)blockIdx.x) return;
  const Job job = jobs[j];
  if (r == 1 && job.node_b < 0) return;
  const int Fp = F + 1;
  const int64_t rstride = (int64_t)(K + 1) * Fp;
  const int node = r == 0 ? job.node_a : job.node_b;
  float* __restrict__ out = rows + (job.out_row + r) * rstride;
  float* __restrict__ mir = job.mirror_row >= 0
                                ? rows + (job.mirror_row + (job.mirror_swap ? 1 - r : r)) * rstride : nullptr;
  for (int e = threadIdx.x; e < (K + 1) * Fp; e += blockDim.x) {
    const int i = e / Fp, c = e - i * Fp;
    float v;
    if (c == 0) {
      v = i == 0 ? (float)(r == 0 ? job.z_a : job.z_b) : job_z[(j * K + (i - 1)) * 2 + r];
    } else if (i == 0) {
      v = X[(int64_t)node * ldx + (c - 1)];
    } else {
      double acc = 0.0;
      for (int64_t q = p0; q < p1; ++q) acc += (double)prows[(2 * q + r) * rstride + e];
      v = (float)acc;
    }
    out[e] = v;
    if (mir) mir[e] = v;
  }
}
#endif  // !S3GRL_LINKS_PART

// ---------------------------------------------------------------------------------------
// Reversed duplicates.  The reference's train split holds BOTH directions of every train edge
// (PyG to_undirected, consumed at utils.py:628), and the link (d,s) has exactly the subgraph,
// the masked operator and the rows of (s,d) with src/dst swapped.  One open-addressing table
// over the links with src < dst, one lookup per link with src > dst; at most one fold per link.
constexpr uint64_t kEmptyKey = ~0ull;

__device__ __forceinline__ uint64_t mix64(uint64_t x) {
  x ^= x >> 33;
  x *= 0xff51afd7ed558ccdull;
  x ^= x >> 33;
  x *= 0xc4ceb9fe1a85ec53ull;
  x ^= x >> 33;
  return x;
}

__global__ void mirror_insert_kernel(const int64_t* __restrict__ links, int64_t L, int64_t N,
                                     uint64_t* __restrict__ keys, int32_t* __restrict__ vals,
                                     uint64_t mask) {
  const int64_t l = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (l >= L) return;
  const int64_t s = links[2 * l], d = links[2 * l + 1];
  if (s < 0 || d < 0 || s >= N || d >= N || s >= d) return;
  const uint64_t key = ((uint64_t)s << 32) | (uint64_t)d;
  uint64_t slot = mix64(key) & mask;
  for (;;) {
    const uint64_t old = atomicCAS(reinterpret_cast<unsigned long long*>(&keys[slot]), kEmptyKey, key);
    if (old == kEmptyKey || old == key) {
      atomicMin(&vals[slot], (int32_t)l);  // the lowest link index owns the key: deterministic
      return;
    }
    slot = (slot + 1) & mask;  // the table has >= 2L slots: a free one exists
  }
}

__global__ void mirror_lookup_kernel(const int64_t* __restrict__ links, int64_t L, int64_t N,
                                     const uint64_t* __restrict__ keys,
                                     const int32_t* __restrict__ vals, uint64_t mask,
                                     int32_t* __restrict__ partner, int32_t* __restrict__ mirror_of,
                                     unsigned long long* __restrict__ n_mirrored) {
  const int64_t l = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (l >= L) return;
  const int64_t s = links[2 * l], d = links[2 * l + 1];
  if (s < 0 || d < 0 || s >= N || d >= N || s <= d) return;
  const uint64_t key = ((uint64_t)d << 32) | (uint64_t)s;
  uint64_t slot = mix64(key) & mask;
  for (;;) {
    const uint64_t k = keys[slot];
    if (k == kEmptyKey) return;
    if (k == key) {
      const int32_t P = vals[slot];
      if (atomicCAS(&mirror_of[P], -1, (int32_t)l) == -1) {
        partner[l] = P;
        atomicAdd(n_mirrored, 1ull);
      }
      return;
    }
    slot = (slot + 1) & mask;
  }
}

__global__ void mirror_rows_kernel(const int32_t* __restrict__ partner, int64_t L,
                                   int32_t* __restrict__ n_rows) {
  const int64_t l = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (l >= L) return;
  const int32_t P = partner[l];
  if (P >= 0) n_rows[l] = n_rows[P];
}

// ---------------------------------------------------------------------------------------
// exclusive scan int32[n] -> int64[n+1], three small launches (tile = 1024 elements)
constexpr int kScanTile = 1024;

__global__ __launch_bounds__(256) void scan_partials_kernel(const int32_t* __restrict__ in,
                                                            int64_t n, int64_t* __restrict__ part) {
  __shared__ int sh[4];
  const int64_t base = (int64_t)blockIdx.x * kScanTile;
  int s = 0;
  for (int k = threadIdx.x; k < kScanTile; k += 256)
    if (base + k < n) s += in[base + k];
  s = block_sum<256>(s, sh);
  if (threadIdx.x == 0) part[blockIdx.x] = s;
}

__global__ __launch_bounds__(1024) void scan_top_kernel(int64_t* __restrict__ part, int64_t nb,
                                                        int64_t* __restrict__ total_out) {
  __shared__ int64_t sh[1024];
  const int tid = threadIdx.x;
  const int64_t chunk = (nb + 1023) / 1024;
  const int64_t b = tid * chunk, e = min(nb, b + chunk);
  int64_t s = 0;
  for (int64_t i = b; i < e; ++i) s += part[i];
  sh[tid] = s;
  __syncthreads();
  for (int o = 1; o < 1024; o <<= 1) {
    const int64_t t = tid >= o ? sh[tid - o] : 0;
    __syncthreads();
    sh[tid] += t;
    __syncthreads();
  }
  int64_t run = tid ? sh[tid - 1] : 0;
  for (int64_t i = b; i < e; ++i) {
    const int64_t v = part[i];
    part[i] = run;
    run += v;
  }
  if (tid == 1023) *total_out = sh[1023];
}

__global__ __launch_bounds__(256) void scan_apply_kernel(const int32_t* __restrict__ in, int64_t n,
                                                         const int64_t* __restrict__ part,
                                                         int64_t* __restrict__ out) {
  __shared__ int sh[4];
  const int64_t base = (int64_t)blockIdx.x * kScanTile;
  const int64_t i0 = base + threadIdx.x * 4;
  int v[4], s = 0;
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    v[k] = i0 + k < n ? in[i0 + k] : 0;
    s += v[k];
  }
  int total;
  int64_t run = part[blockIdx.x] + block_excl_scan<256>(s, sh, total);
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    if (i0 + k < n) out[i0 + k] = run;
    run += v[k];
  }
}

// The three offset arrays of a plan (nodes, rows, row pairs per link) in ONE set of launches
// (blockIdx.y picks the array) instead of three, with their maxima and totals written next to the
// plan's other scalars: a plan is a few dozen small launches, and on a sharded list their fixed cost
// is paid once per piece (0.5 ms per plan before this).
struct Scan3 {
  const int32_t* in[3];
  int64_t* out[3];
  int64_t* part[3];
  long long* max_out[3];   // may be null
  int64_t* total_out[3];   // may be null
};

__global__ __launch_bounds__(256) void scan3_partials_kernel(Scan3 a, int64_t n) {
  __shared__ int sh[4];
  const int y = blockIdx.y;
  const int32_t* __restrict__ in = a.in[y];
  const int64_t base = (int64_t)blockIdx.x * kScanTile;
  int s = 0, m = 0;
  for (int k = threadIdx.x; k < kScanTile; k += 256)
    if (base + k < n) {
      const int v = in[base + k];
      s += v;
      m = max(m, v);
    }
  s = block_sum<256>(s, sh);
  if (threadIdx.x == 0) a.part[y][blockIdx.x] = s;
  if (a.max_out[y]) {
    for (int o = 32; o > 0; o >>= 1) m = max(m, __shfl_xor(m, o));
    if ((threadIdx.x & 63) == 0 && m > 0) atomicMax(a.max_out[y], (long long)m);
  }
}

__global__ __launch_bounds__(1024) void scan3_top_kernel(Scan3 a, int64_t nb, int64_t n) {
  __shared__ int64_t sh[1024];
  const int y = blockIdx.x;
  int64_t* __restrict__ part = a.part[y];
  const int tid = threadIdx.x;
  const int64_t chunk = (nb + 1023) / 1024;
  const int64_t b = tid * chunk, e = min(nb, b + chunk);
  int64_t s = 0;
  for (int64_t i = b; i < e; ++i) s += part[i];
  sh[tid] = s;
  __syncthreads();
  for (int o = 1; o < 1024; o <<= 1) {
    const int64_t t = tid >= o ? sh[tid - o] : 0;
    __syncthreads();
    sh[tid] += t;
    __syncthreads();
  }
  int64_t run = tid ? sh[tid - 1] : 0;
  for (int64_t i = b; i < e; ++i) {
    const int64_t v = part[i];
    part[i] = run;
    run += v;
  }
  if (tid == 1023) {
    a.out[y][n] = sh[1023];
    if (a.total_out[y]) *a.total_out[y] = sh[1023];
  }
}

__global__ __launch_bounds__(256) void scan3_apply_kernel(Scan3 a, int64_t n) {
  __shared__ int sh[4];
  const int y = blockIdx.y;
  const int32_t* __restrict__ in = a.in[y];
  int64_t* __restrict__ out = a.out[y];
  const int64_t base = (int64_t)blockIdx.x * kScanTile;
  const int64_t i0 = base + threadIdx.x * 4;
  int v[4], s = 0;
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    v[k] = i0 + k < n ? in[i0 + k] : 0;
    s += v[k];
  }
  int total;
  int64_t run = a.part[y][blockIdx.x] + block_excl_scan<256>(s, sh, total);
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    if (i0 + k < n) out[i0 + k] = run;
    run += v[k];
  }
}

#include "s3grl_onehop.inl"

// ---------------------------------------------------------------------------------------
// LDS bytes link_kernel needs beyond its fixed part: list[n] + dinvP[p] + two float2 state
// arrays [p] (+ alignment slack); the hash flavour adds its keys/vals tables.
// When every operator reaches the whole subgraph (p == n) and the subgraph is small, link_kernel
// also keeps its adjacency as an n x n bit matrix (+ two index maps), see there.
constexpr int kBmMaxNodes = 512;
__host__ __device__ __forceinline__ int link_bm_bytes(int n, int p) {
  return (p == n && n <= kBmMaxNodes) ? 4 * n * ((n + 31) >> 5) + 4 * n + 8 : 0;
}
__host__ __device__ __forceinline__ int link_lds_need(int n, int p) { return 4 * n + 20 * p + 16; }
__host__ __device__ __forceinline__ int link_lds_need_sparse(int n, int p) {
  int C = 64;
  while (C < 2 * n) C <<= 1;
  return 8 * C + link_lds_need(n, p) + link_bm_bytes(n, p);
}

struct ClassBounds {
  int b[kNumClasses];
};

// class ids: 0..kNumClasses-1 bitmap flavour by LDS need, kNumClasses = HBM-scratch flavour,
// kSparseBase.. = hash flavour by LDS need.  class_count[kNumClasses + 1] = max need of the
// HBM-scratch class.
constexpr int kSparseBase = kNumClasses + 2;
// kFullBase.. = one-hop full-reach links for link_full_kernel by LDS need (bit matrix in LDS),
// kFullBig = the same with the bit matrix in an HBM slice (subgraphs of more than ~700 nodes)
constexpr int kFullBase = kSparseBase + kNumClasses;
constexpr int kFullBig = kFullBase + kNumClasses;
// kHubBase.. = one-hop links with a cached hub neighbourhood, link_hub_kernel (s3grl_hub.hip) by LDS need
constexpr int kHubBase = kFullBig + 1;
constexpr int kNumLists = kHubBase + kHubClasses + 1;   // (+ the class with its found edges in HBM slices)
constexpr int kTinyList = kNumLists;   // one-hop PoS links of at most kTinyNodes nodes: link_tiny_kernel (s3grl_hub.hip)
static_assert(kTinyList + 1 < 29, "class_count[29..31] carry maxima");
// kCsrBase.. (s3grl_internal.hpp) = full-reach links on their induced LDS CSR, link_csr_kernel (s3grl_csr.hip)
constexpr int kNumListsAll = kCsrBase + kCsrClasses;

__global__ void classify_kernel(const int32_t* __restrict__ n_nodes,
                                const int32_t* __restrict__ p_nodes,
                                const int32_t* __restrict__ lvl_max, int64_t L, ClassBounds bound,
                                int sparse_mode, ClassBounds sbound, const int32_t* __restrict__ e_cap,
                                ClassBounds fbound, int bm_limit, int dm_max_n, int dm_class_mask,
                                int32_t* __restrict__ class_count, int32_t* __restrict__ class_list,
                                const int32_t* __restrict__ perm, const int64_t* __restrict__ x_cap,
                                ClassBounds hbound, const int32_t* __restrict__ csr_e, CsrBounds cbound, int W,
                                int csr_pct, int tiny_max_n) {
  const int64_t li = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  // (perm: the class lists come out in the plan's processing order, up to the order of the atomics)
  const int64_t l = li < L ? (perm ? (int64_t)perm[li] : li) : L;
  const int n = l < L ? n_nodes[l] : 0;
  const int p = l < L ? p_nodes[l] : 0;
  int need = link_lds_need(n, p);
  int c = 0;
  bool sparse = false;
  // one-hop plan (e_cap is only produced for those), every operator reaches all of S, local ids fit
  // 16 bits: link_full_kernel, by its LDS need with or without the bit matrix on chip
  // every operator reaches all of S and the sizing kernel of the induced-CSR flavour counted the link's
  // entries (s3grl_csr.hip): link_csr_kernel, by its exact LDS need
  if (csr_e && n > 0 && p == n && csr_e[l] >= 0) {
    const int need_c = (int)((int64_t)csr_lds_need(n, csr_e[l], W) * csr_pct / 100);   // (csr_pct: measurement hook)
    if (need_c <= cbound.b[kCsrClasses - 1]) {
      sparse = true;
#pragma unroll
      for (int k = 0; k < kCsrClasses; ++k) c += need_c > cbound.b[k] ? 1 : 0;
      c += kCsrBase;
    }
  }
  if (!sparse && x_cap && n > 0 && p == n && x_cap[l] >= 0) {
    const int64_t stage = x_cap[l] >> 32, xb = x_cap[l] & 0xffffffffll;
    const int64_t need_h = hub_lds_need(n, xb) + stage;
    if (!hbound.b[kHubClasses] && need_h <= hbound.b[kHubClasses - 1]) {   // (b[kHubClasses]: test hook)
      sparse = true;
#pragma unroll
      for (int k = 0; k < kHubClasses; ++k) c += need_h > hbound.b[k] ? 1 : 0;
      c += kHubBase;
    } else if (hub_lds_need(n, 0) + stage <= hbound.b[kHubClasses - 1]) {
      sparse = true;
      c = kHubBase + kHubClasses;
      atomicMax(&class_count[29], (int)xb);   // sizes the HBM slices of that class
    }
  }
  if (!sparse && e_cap && n > 0 && p == n && n <= tiny_max_n) {   // (tiny_max_n = 0: plans with common-neighbour rows)
    sparse = true;
    c = kTinyList + (n > 32 ? 1 : 0);
  }
  if (!sparse && e_cap && n > 0 && p == n && n <= 65535) {
    const int ec = e_cap[l];
    const int need_a = full_lds_need(n, ec, true), need_b = full_lds_need(n, ec, false);
    if (need_a <= min(fbound.b[kNumClasses - 1], bm_limit)) {
      sparse = true;
#pragma unroll
      for (int k = 0; k < kNumClasses; ++k) c += need_a > fbound.b[k] ? 1 : 0;
      c += kFullBase;
    } else if (need_b <= fbound.b[kNumClasses - 1]) {
      sparse = true;
      c = kFullBig;
      atomicMax(&class_count[31], ec);   // sizes the HBM slices of that class
      atomicMax(&class_count[30], need_b);   // and its LDS
    }
  }
  if (!sparse && sparse_mode == 2 && n > 0 && n <= dm_max_n && need <= sbound.b[kNumClasses - 1]) {
    // direct-map flavour: sbound = what the map leaves of the LDS; the list must be in the stash;
    // class by class where the host found the map to pay (dm_class_mask)
    int k_dm = 0;
#pragma unroll
    for (int k = 0; k < kNumClasses; ++k) k_dm += need > sbound.b[k] ? 1 : 0;
    if ((dm_class_mask >> k_dm) & 1) {
      sparse = true;
      c = k_dm + kSparseBase;
    }
  }
  if (!sparse && sparse_mode == 1 && n > 0 && lvl_max[l] <= kSparseLevelMax) {
    const int sneed = link_lds_need_sparse(n, p);
    if (sneed <= sbound.b[kNumClasses - 1]) {
      sparse = true;
#pragma unroll
      for (int k = 0; k < kNumClasses; ++k) c += sneed > sbound.b[k] ? 1 : 0;
      c += kSparseBase;
    }
  }
  if (!sparse) {
#pragma unroll
    for (int k = 0; k < kNumClasses; ++k) c += need > bound.b[k] ? 1 : 0;
  }
  if (n == 0) c = -1;
  // one atomic per (wave, class) instead of one per link
  // (only the classes present in the wavefront are visited: two or three of kNumLists)
  const int lane = threadIdx.x & 63;
  unsigned long long todo = __ballot(c >= 0);
  while (todo) {
    const int k = __shfl(c, __ffsll((long long)todo) - 1);
    const unsigned long long m = __ballot(c == k);
    const int leader = __ffsll((long long)m) - 1;
    int base = 0;
    if (lane == leader) base = atomicAdd(&class_count[k], __popcll(m));
    base = __shfl(base, leader);
    if (c == k) class_list[(int64_t)k * L + base + __popcll(m & ((1ull << lane) - 1ull))] = (int32_t)l;
    todo &= ~m;
  }
  // class kNumClasses = links that do not fit LDS: they run with their lists in HBM scratch
  if (c == kNumClasses) atomicMax(&class_count[kNumClasses + 1], need);
}

// ---------------------------------------------------------------------------------------
// The fused per-link kernel.  LDS layout (dynamic, 16-byte aligned base), n = |S|, p = |P|:
//   vis[W]                 S as a bitmap over global ids
//   inP[W]                 P as a bitmap (the BFS's next-frontier bitmap until the BFS is done)
//   wpreP[W]               word-level popcount prefix of inP: local id of u ∈ P = rank in P
//   cn[cn_cap] lvl_end[kMaxLevels] zbuf[4K] sh[32]
//   list[n]                S in hop-major order, ascending id inside a hop (global ids)
//   dinvP[p]               D^-1/2 of the masked induced subgraph for the nodes of P
//   cur[p], nxs[p]         float2 propagation state s_i = dinv·r_i (rows a, b of the pair)
// P = the hop-major prefix of S that r_{K-1} can reach; only the LAST operator touches the
// rest of S, and it needs no state there: its degree and its sum come out of the same pass.
// GS = true: list / dinvP / state live in a per-workgroup HBM scratch slice instead of LDS (links
// whose subgraph does not fit what the bitmaps leave of 160 KiB); same code, slower memory.
// HS = true: the visited set is a hash table sized by the subgraph (keys/vals of C = pow2 >= 2n
// slots) instead of three N-bit bitmaps, local id = position in the hop-major list: for graphs
// whose bitmaps alone would take tens of KB of LDS per workgroup.  Same node lists, rows and
// statistics bit for bit; the sums agree to fp32 round-off (small fully-reached subgraphs are
// propagated through an LDS adjacency bit matrix in this flavour: another summation order).
// DM = true (with HS): the visited set is a direct map, one uint16 per node of the GRAPH holding the
// node's position in the hop-major list (0xFFFF = not in S): a neighbour visit is one LDS read instead
// of two bitmap words + a rank prefix + a popcount.  For graphs whose map (2N bytes) leaves most of
// the LDS free; the list comes from count_kernel's stash (the host sends no other link here).  Rows
// are walked in the same order by the same lanes as in the bitmap flavour: same sums bit for bit.
// DIRECTED (bitmap flavour only): the BFS above ran on the union of successors and predecessors
// (utils.py:60-63); the operator is D^-1/2 A D^-1/2 of the directed induced matrix with D = OUT-degrees
// (row counts, tuned_SIGN.py:158-161), so r_i = r_{i-1} A_hat pulls over a node's PREDECESSORS (dg.in_*)
// and the degrees are counted over its successors (dg.out_*), for every node of S (p == n).
#ifndef S3GRL_LINK_MINW
#define S3GRL_LINK_MINW(T) 1
#endif
template <int T, int K, int G, bool GS, bool HS, bool DM = false, bool DIRECTED = false>
__global__ __launch_bounds__(T, S3GRL_LINK_MINW(T)) void link_kernel(
    const int32_t* __restrict__ indptr, const int32_t* __restrict__ indices, int W,
    const int64_t* __restrict__ links, const int32_t* __restrict__ class_list, int hops, int plus,
    int cn_cap, int full_stats, int hubs, const WalkSets ws,
    const int32_t* __restrict__ p_nodes,
    const int64_t* __restrict__ node_off, const int64_t* __restrict__ row_ptr,
    const int64_t* __restrict__ job_off, const int64_t* __restrict__ coef_off,
    const int32_t* __restrict__ mirror_of, int32_t* __restrict__ c_ids, float* __restrict__ c_coef,
    Job* __restrict__ jobs, float* __restrict__ job_z, int32_t* __restrict__ job_lim,
    int64_t* __restrict__ row_nodes,
    int32_t* __restrict__ lvl_out, unsigned long long* __restrict__ tot_edges,
    unsigned long long* __restrict__ tot_support, unsigned long long* __restrict__ tot_vol,
    char* __restrict__ scratch, int64_t scratch_stride, int bm_ext_words, unsigned long long* __restrict__ dbg,
    HopSampling smp, const int32_t* __restrict__ stash, int slot,
    const int32_t* __restrict__ old_of_new, const int32_t* __restrict__ new_of_old, int lo_id,
    int split_t, int seg_shift, const DirGraph dg, int sop2) {
  static_assert(!DIRECTED || (!HS && !DM), "directed plans run on the bitmap flavour");
  // sop2 (S3GRL_MODE_SOP_RESTRICTED): the rows of the GLOBAL operator restricted to the subgraph — D^-1/2 from
  // the global degrees, the target link NOT removed, the partner's column zeroed in the features and the
  // label column = the diagonal entry (reference tuned_SIGN.py:71-78,102-113 on the ball instead of all of V)
  extern __shared__ uint32_t smem[];
  // rows walked by the operator passes (pull), by the degree count and by the common-neighbour test
  const int32_t* __restrict__ w_indptr = DIRECTED ? dg.in_indptr : indptr;
  const int32_t* __restrict__ w_indices = DIRECTED ? dg.in_indices : indices;
  const int32_t* __restrict__ o_indptr = DIRECTED ? dg.out_indptr : indptr;
  const int32_t* __restrict__ o_indices = DIRECTED ? dg.out_indices : indices;
  // the caller's id of an internal id (the graph is walked in its degree order, s3grl_relabel.hip)
  auto ext = [&](int v) -> int { return old_of_new ? old_of_new[v] : v; };
  // diagnostic only (S3GRL_DEBUG_STAMPS): cycles per phase, summed over workgroups; the extra
  // barriers change the timing of the build they run in — read shares, not totals
  unsigned long long t_prev = dbg ? __builtin_amdgcn_s_memtime() : 0ull;
#define S3GRL_STAMP(idx)                                                              \
  if (dbg) {                                                                          \
    __syncthreads();                                                                  \
    const unsigned long long t_now = __builtin_amdgcn_s_memtime();                    \
    if (threadIdx.x == 0) atomicAdd(&dbg[idx], t_now - t_prev);                       \
    t_prev = t_now;                                                                   \
  }
  const int tid = threadIdx.x;
  const int l = class_list[blockIdx.x];
  const int64_t noff = node_off[l];
  const int n_alloc = (int)(node_off[l + 1] - noff);
  const int p_alloc = p_nodes[l];
  const int mirror = mirror_of ? mirror_of[l] : -1;          // reversed duplicate folded into l
  const int64_t mrp = mirror >= 0 ? row_ptr[mirror] : -1;

  // visited set: three bitmaps of W words, or (HS) keys + vals of C words each
  uint32_t hmask = 0;
  int set_words = 3 * W;
  if constexpr (DM) {
    set_words = 16 * W;   // 32 W uint16 entries
  } else if constexpr (HS) {
    int C = 64;
    while (C < 2 * n_alloc) C <<= 1;
    hmask = (uint32_t)(C - 1);
    set_words = 2 * C;
  }
  // GS on a graph whose three N-bit bitmaps do not fit LDS (num_nodes > ~327 680): they sit at the head
  // of the workgroup's HBM slice (bm_ext_words > 0) and LDS holds the small fixed part only
  const bool bm_ext = GS && bm_ext_words > 0;
  if (bm_ext) set_words = 0;
  uint16_t* dmap = reinterpret_cast<uint16_t*>(smem);
  uint32_t* vis = smem;
  uint32_t* inP = smem + W;
  uint32_t* wpreP = smem + 2 * W;
  int32_t* hkeys = reinterpret_cast<int32_t*>(smem);
  int32_t* hvals = hkeys + (hmask + 1);
  int32_t* cn = reinterpret_cast<int32_t*>(smem + set_words);
  int* lvl_end = cn + cn_cap;
  float* zbuf = reinterpret_cast<float*>(lvl_end + kMaxLevels);  // [2 (src,dst)][K][2 (rows)]
  int* sh = reinterpret_cast<int*>(zbuf + 4 * K);
  int* hub = hubs ? sh + 32 : nullptr;
  int32_t* list;
  float* dinvP;
  float2* cur;
  if constexpr (GS) {
    char* base = scratch + (int64_t)blockIdx.x * scratch_stride;   // 256-byte aligned slices
    if (bm_ext) {
      vis = reinterpret_cast<uint32_t*>(base);
      inP = vis + W;
      wpreP = vis + 2 * W;
      base += (size_t)bm_ext_words * 4;
    }
    list = reinterpret_cast<int32_t*>(base);
    dinvP = reinterpret_cast<float*>(list + n_alloc);
    cur = reinterpret_cast<float2*>(base + (((size_t)(n_alloc + p_alloc) * 4 + 7) & ~(size_t)7));
  } else {
    list = sh + 32 + kHubWords;
    dinvP = reinterpret_cast<float*>(list + n_alloc);
    const int fixed_words = set_words + cn_cap + kMaxLevels + 4 * K + 32 + kHubWords;
    cur = reinterpret_cast<float2*>(smem + ((fixed_words + n_alloc + p_alloc + 1) & ~1));
  }
  float2* nxs = cur + p_alloc;

  const int src = (int)links[2 * (int64_t)l], dst = (int)links[2 * (int64_t)l + 1];
  const int msrc = sop2 ? -2 : src, mdst = sop2 ? -3 : dst;   // the endpoints as far as the MASKING is concerned
  auto gdinv = [&](int v) -> float {                          // sop2: D^-1/2 of the global degree
    const int d = indptr[v + 1] - indptr[v];
    return d > 0 ? 1.0f / sqrtf((float)d) : 0.0f;
  };

  // ---- BFS on the unmasked graph (reference utils.py:53-74) --------------------------------
  int nlev;
  int n;
  if (DM || (stash && n_alloc - 2 <= slot)) {
    // count_kernel left this link's node list (hop-major, ascending id inside a hop) and its
    // level ends in HBM: rebuild the LDS state from them instead of walking the graph again
    const int32_t* __restrict__ st = stash + (int64_t)l * slot;
    const int32_t* lv = lvl_out + (int64_t)l * kMaxLevels;   // rewritten below, after the barriers
    if constexpr (DM) {
      for (int t = tid; t < 16 * W; t += T) smem[t] = 0xffffffffu;
    } else if constexpr (HS) {
      for (uint32_t t = tid; t <= hmask; t += T) hkeys[t] = -1;
    } else {
      for (int t = tid; t < W; t += T) {
        vis[t] = 0;
        inP[t] = 0;
      }
    }
    nlev = lv[kMaxLevels - 1];
    if (tid < nlev) lvl_end[tid] = lv[tid];
    if (tid == 0) {
      list[0] = min(src, dst);
      list[1] = max(src, dst);
    }
    hub_rows_clear<T>(hub);
    __syncthreads();
    for (int t = tid; t < n_alloc; t += T) {
      const int v = t < 2 ? list[t] : st[t - 2];
      if (t >= 2) list[t] = v;
      if constexpr (DM) {
        dmap[v] = (uint16_t)t;
      } else if constexpr (HS) {
        hs_insert(hkeys, hmask, v);
        hvals[hs_find(hkeys, hmask, v)] = t;
      } else {
        atomicOr(&vis[v >> 5], 1u << (v & 31));
      }
    }
    __syncthreads();
    n = n_alloc;
  } else if constexpr (HS) {
    n = bfs_hash<T, G>(indptr, indices, src, dst, hops, hkeys, hvals, hmask, list, n_alloc, lvl_end, sh + 31,
                       hub, nlev, ws, l);
  } else {
    n = bfs_list<T, G>(indptr, indices, W, src, dst, hops, vis, inP, list, n_alloc, lvl_end, sh, hub, nlev,
                       ws, l, smp);
  }
  // set queries of the passes below: membership in S; index into the P-state arrays (+ is it in P);
  // the P-state index of list entry t (= node v)
  auto in_s = [&](int u) -> bool {
    if constexpr (DM) return dmap[u] != 0xffffu;
    else if constexpr (HS) return hs_find(hkeys, hmask, u) >= 0;
    else return test_bit(vis, u);
  };
  auto p_index_of_row = [&](int t, int v) -> int {
    if constexpr (HS) { (void)v; return t; }
    else { (void)t; return rank_of(inP, wpreP, v); }
  };

  S3GRL_STAMP(0)
  // ---- rows of this link ----------------------------------------------------------------
  const int64_t rp = row_ptr[l];
  const int R = (int)(row_ptr[l + 1] - rp);
  const int max_row_hop = R > 2 ? 1 : 0;  // common neighbours sit at hop 1
  const int p = DIRECTED ? n : lvl_end[min(K - 1 + max_row_hop, nlev - 1)];

  // ---- P as bitmap + rank prefix; node list out -------------------------------------------
  if constexpr (!HS) {
    for (int t = tid; t < p; t += T) {
      const int v = list[t];
      atomicOr(&inP[v >> 5], 1u << (v & 31));
    }
  }
  int vol_local = 0;   // vol(S) = Σ global degrees, the 4·vol(S) term of the algorithmic bytes
  for (int t = tid; t < n; t += T) {
    const int v = list[t];
    c_ids[noff + t] = ext(v);
    vol_local += indptr[v + 1] - indptr[v];
  }
  if (plus && wave_id() == 0) {
    const int c = common_neighbours(o_indptr, o_indices, in_s, src, dst, cn);
    if (old_of_new && c > 1) {
      // the common-neighbour rows go out in ascending order of the CALLER's ids: rank sort by one
      // wavefront; the host sized cn[] three times over for it (keys and the sorted copy behind the list)
      int* key = cn + c;
      int* tmp = cn + 2 * c;
      const int lane = lane_id();
      for (int i = lane; i < c; i += 64) key[i] = old_of_new[cn[i]];
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
      __builtin_amdgcn_wave_barrier();
      for (int i = lane; i < c; i += 64) {
        const int k = key[i];
        int r = 0;
        for (int j = 0; j < c; ++j) r += key[j] < k ? 1 : 0;
        tmp[r] = cn[i];
      }
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
      __builtin_amdgcn_wave_barrier();
      for (int i = lane; i < c; i += 64) cn[i] = tmp[i];
    }
  }
  __syncthreads();
  if constexpr (!HS) rank_prefix<T>(inP, wpreP, W, sh);
  __syncthreads();
  if (tid == 0)
    for (int d = 0; d < kMaxLevels; ++d)
      lvl_out[(int64_t)l * kMaxLevels + d] = d < nlev ? lvl_end[d] : n;
  for (int r = tid; r < R; r += T) {
    row_nodes[rp + r] = ext(r == 0 ? src : (r == 1 ? dst : cn[r - 2]));
    if (mirror >= 0) row_nodes[mrp + r] = ext(r == 0 ? dst : (r == 1 ? src : cn[r - 2]));
  }

  S3GRL_STAMP(1)
  // ---- D^-1/2 on P (inf -> 0) -------------------------------------------------------------
  // reference tuned_SIGN.py:153-161: structure only, target link removed, no self-loops added
  // Only for the hops the row nodes themselves sit in (src/dst; the common neighbours at hop 1):
  // every later pass derives the D^-1/2 of the list rows it reaches for the first time from its
  // own walk of those rows (dinv_rows = how far that has got), so no row of P is walked for its
  // degree alone.
  int edges_local = 0;
  int edges_exact = -1;   // set when a pass of pair 0 walked every row of S
  int dinv_rows = DIRECTED ? n : lvl_end[min(max_row_hop, nlev - 1)];
  if (sop2) {
    for (int t = tid; t < dinv_rows; t += T) {
      const int v = list[t];
      dinvP[p_index_of_row(t, v)] = gdinv(v);
      edges_local += indptr[v + 1] - indptr[v];
    }
  } else if (!DIRECTED && !walks_on(ws) && !sampling_on(smp) && hops > max_row_hop) {
    // A plain BFS to `hops` holds every neighbour of a node that sits below hop `hops`: the
    // subgraph degree of such a row is its global degree, minus the masked target link at src and
    // dst (utils.py:79-80).  No walk.
    for (int t = tid; t < dinv_rows; t += T) {
      const int v = list[t];
      const int b = indptr[v], e = indptr[v + 1];
      int d = e - b;
      if (v == src || v == dst) d -= sorted_contains(indices + b, d, v == src ? dst : src) ? 1 : 0;
      dinvP[p_index_of_row(t, v)] = d > 0 ? 1.0f / sqrtf((float)d) : 0.0f;
      edges_local += d;
    }
  } else {
    walk_rows<T, G, 2>(
        0, dinv_rows, list, o_indptr, o_indices, hub,
        [&](RowAcc& a, int v, int u, bool valid) {
          // the target link is masked (utils.py:79-80): one compare per neighbour against the
          // row's partner (-1 for every row but src and dst; hoisted out of the neighbour loop)
          const int mp = v == msrc ? dst : (v == mdst ? src : -1);
          a.n += (valid && in_s(u) && u != mp) ? 1 : 0;
        },
        [&](RowAcc& a, int t, int v) {
          dinvP[p_index_of_row(t, v)] = a.n > 0 ? 1.0f / sqrtf((float)a.n) : 0.0f;
          edges_local += a.n;
        });
  }
  __syncthreads();

  S3GRL_STAMP(2)
  // ---- per row pair: K pull steps --------------------------------------------------------
  // State s_i[u] = dinv[u]·r_i[u] for u ∈ P (float2: rows a and b of the pair):
  //   r_i[w] = dinv[w] · Σ_{u ∈ N_S(w)} s_{i-1}[u]            (Â symmetric: pull == r_{i-1}·Â)
  // Each r_i[w] is summed in the stored order of w's row and reduced over G lanes by a fixed
  // xor tree: bit-reproducible.  All terms are >= 0: no cancellation.  A walk of length i
  // from a row at hop h_r stays within hop h_r + i, so step i only visits that list prefix;
  // the last step visits everything it can reach and derives dinv[w] from the same pass.
  // Small subgraphs that every operator reaches entirely (p == n: sign_k - 1 >= the BFS depth):
  // the first pass over all rows also records the masked induced adjacency as an n x n bit matrix
  // in LDS (row = list position, column = list position), and every later full pass — the
  // remaining operators, the last one, the passes of the common-neighbour pairs — sums over the
  // set bits of a row instead of walking its global CSR row through the bitmaps / the hash
  // (a 1-hop subgraph of a power-law graph has a few hundred induced edges and ~13 000 stored
  // neighbours).  Columns are list positions in both flavours of the visited set, so both sum in
  // the same order.
  const int WB = (n + 31) >> 5;
  // Hash flavour only (big graphs, where a visit costs a hash probe).  In the bitmap flavour it
  // measured a loss: USAir's 1-hop subgraphs are nearly as dense as their global rows (+20 % on
  // the link kernel), and on PubMed K=5 the matrix of a 300-500-node subgraph pushes the link
  // into a bigger LDS class (+10 %); the collab-scale config gains 12 %.
  const bool use_bm = HS && !DM && !GS && K >= 2 && p == n && p_alloc == n_alloc && n <= kBmMaxNodes && !sop2;
  uint32_t* bm = reinterpret_cast<uint32_t*>(nxs + p_alloc);              // [n][WB]
  uint16_t* pos_of_rank = reinterpret_cast<uint16_t*>(bm + (use_bm ? n * WB : 0));   // bitmap flavour
  uint16_t* rank_of_pos = pos_of_rank + n;
  bool bm_ready = false;
  if (use_bm) {
    for (int i = tid; i < n * WB; i += T) bm[i] = 0;
    if constexpr (!HS) {
      for (int t = tid; t < n; t += T) {
        const int r = rank_of(inP, wpreP, list[t]);
        pos_of_rank[r] = (uint16_t)t;
        rank_of_pos[t] = (uint16_t)r;
      }
    }
    __syncthreads();
  }
  // The graph is walked in descending degree order (lo_id >= 0: ids >= lo_id have at most two stored
  // neighbours), so the rows of a hop are sorted by length and its leaves form its tail.  The tail of
  // the LAST hop — a fifth of a PubMed subgraph's rows — is walked with one lane per row (both
  // neighbours in the lane's two slots, no tail loop) instead of G; two terms add up to the same
  // bits either way.
  int lo_begin = n;
  if (lo_id >= 0 && nlev >= 2 && !walks_on(ws)) {
    int lo = lvl_end[nlev - 2], hi = n;   // ascending ids inside the hop
    while (lo < hi) {
      const int mid = (lo + hi) >> 1;
      if (list[mid] < lo_id) lo = mid + 1; else hi = mid;
    }
    lo_begin = lo;
  }
  auto walk_split = [&](int limit, auto visit, auto commit) __attribute__((always_inline)) {
    const int a_end = min(limit, lo_begin);
    walk_rows<T, G, 2>(0, a_end, list, w_indptr, w_indices, hub, visit, commit);
    if (limit > a_end) walk_rows<T, 1, 2>(a_end, limit, list, w_indptr, w_indices, nullptr, visit, commit);
  };
  const int npairs = (R + 1) / 2;
  for (int pr = 0; pr < npairs; ++pr) {
    const int64_t jid = job_off[l] + pr;
    // PoS: one pair per link, its list sits at the link's node offset; PoS Plus: per-pair offsets
    const int64_t coff = coef_off ? coef_off[jid] : noff;
    const int node_a = pr == 0 ? src : cn[2 * pr - 2];
    const int node_b = pr == 0 ? dst : (2 * pr + 1 < R ? cn[2 * pr - 1] : -1);
    const int row_hop = pr == 0 ? 0 : 1;
    const int support = lvl_end[min(K + row_hop, nlev - 1)];
    // with full_stats the last pass also walks the rows beyond its reach, to count edges
    const int last_rows = (pr == 0 && full_stats) ? n : support;

    for (int w = tid; w < p; w += T) {
      cur[w] = make_float2(0.f, 0.f);
      nxs[w] = make_float2(0.f, 0.f);
    }
    if (tid < 4 * K) zbuf[tid] = 0.f;
    __syncthreads();
    if (tid == 0) {
      auto p_index = [&](int v) -> int {
        if constexpr (DM) return (int)dmap[v];
        else if constexpr (HS) return hvals[hs_find(hkeys, hmask, v)];
        else return rank_of(inP, wpreP, v);
      };
      const int la = p_index(node_a);
      cur[la].x = dinvP[la];
      if (node_b >= 0) {
        const int lb = p_index(node_b);
        cur[lb].y = dinvP[lb];
      }
    }
    __syncthreads();

    float2* s_in = cur;
    float2* s_out = nxs;
    float2* coef = reinterpret_cast<float2*>(c_coef) + coff * K;  // [K][support] float2
    // A list longer than split_t entries is cut into pieces of 2^seg_shift entries that the gather
    // treats as jobs of their own (s3grl_internal.hpp, kSplitThreshold): their coefficients are laid
    // out piece by piece, [K][piece length] each.  All pieces before the last are full, so piece s
    // starts at s * 2^seg_shift * K.
    const bool split = split_t > 0 && support > split_t;
    auto cidx = [&](int i, int t) -> int64_t {
      if (!split) return (int64_t)i * support + t;
      const int s0 = (t >> seg_shift) << seg_shift;
      const int len = min(1 << seg_shift, support - s0);
      return (int64_t)s0 * K + (int64_t)i * len + (t - s0);
    };
    // one operator step over ALL rows through the bit matrix: WL lanes per row (one word of the
    // row each, WL = the row's word count rounded up to a power of two, at most 16), the set
    // bits of a word in ascending position, then a fixed xor tree over the WL lanes:
    // bit-reproducible, and the same order in both flavours of the visited set
    int edges_bm = 0;
    const int wl_shift = WB <= 1 ? 0 : (WB <= 2 ? 1 : (WB <= 4 ? 2 : (WB <= 8 ? 3 : 4)));
    auto bm_pass = [&](int i, bool last) {
      const int WL = 1 << wl_shift;
      const int j0 = tid & (WL - 1);
      for (int base = 0; base < n; base += T >> wl_shift) {
        const int t = base + (tid >> wl_shift);
        float ax = 0.f, ay = 0.f;
        int deg = 0;
        if (t < n) {
          for (int j = j0; j < WB; j += WL) {
            uint32_t w32 = bm[t * WB + j];
            deg += __popc(w32);
            while (w32) {
              const int c = j * 32 + __ffs(w32) - 1;
              w32 &= w32 - 1;
              int idx = c;
              if constexpr (!HS) idx = rank_of_pos[c];
              const float2 sv = s_in[idx];
              ax += sv.x;
              ay += sv.y;
            }
          }
        }
        for (int o = WL >> 1; o > 0; o >>= 1) {   // WL divides 64: the partners are in this wave
          ax += __shfl_xor(ax, o);
          ay += __shfl_xor(ay, o);
          deg += __shfl_xor(deg, o);
        }
        if (t < n && j0 == 0) {
          edges_bm += deg;
          const int v = list[t];
          int w = t;
          if constexpr (!HS) w = rank_of_pos[t];
          const float dw = dinvP[w];
          const float rx = dw * ax, ry = dw * ay;
          if (!last) s_out[w] = make_float2(dw * rx, dw * ry);
          coef[cidx(i, t)] = make_float2(rx, ry);
          if (v == src) { zbuf[(0 * K + i) * 2] = rx; zbuf[(0 * K + i) * 2 + 1] = ry; }
          if (v == dst) { zbuf[(1 * K + i) * 2] = rx; zbuf[(1 * K + i) * 2 + 1] = ry; }
        }
      }
    };
#pragma unroll 1
    for (int i = 0; i < K - 1; ++i) {
      const int limit = lvl_end[min(i + 1 + row_hop, nlev - 1)];  // <= p
      if (bm_ready && limit == n) {
        bm_pass(i, false);
        __syncthreads();
        float2* tmp = s_in;
        s_in = s_out;
        s_out = tmp;
        continue;
      }
      const bool build_bm = use_bm && !bm_ready && limit == n;   // first pass over all rows
      walk_split(
          limit,
          [&](RowAcc& a, int v, int u, bool valid) {
            bool on;
            float2 sv;
            bool member;
            int col;   // list position of u (meaningful for members)
            if constexpr (DM) {
              const int r = dmap[u];   // list position; 0xFFFF (>= p) = not in S
              sv = s_in[min(r, p - 1)];
              member = valid && r != 0xffff;
              on = valid && r < p;
              col = r;
            } else if constexpr (HS) {
              const int slot = hs_find(hkeys, hmask, u);
              const int r = hvals[max(slot, 0)];
              sv = s_in[min(max(r, 0), p - 1)];
              member = valid && slot >= 0;
              on = member && r < p;
              col = r;
            } else {
              // all LDS reads unconditional, count and contribution selected afterwards
              const uint32_t bit = 1u << (u & 31);
              const uint32_t wv = vis[u >> 5], wp = inP[u >> 5];
              const int r = (int)wpreP[u >> 5] + __popc(wp & (bit - 1u));
              sv = s_in[min(r, p - 1)];
              member = valid && (wv & bit);
              on = member && (wp & bit);
              col = r;
            }
            const int mp = v == msrc ? dst : (v == mdst ? src : -1);
            member = member && u != mp;
            on = on && u != mp;
            if (build_bm && member) {
              if constexpr (!HS) col = pos_of_rank[min(col, n - 1)];
              atomicOr(&bm[a.row * WB + (col >> 5)], 1u << (col & 31));
            }
            a.n += member ? 1 : 0;
            a.x += on ? sv.x : 0.f;
            a.y += on ? sv.y : 0.f;
          },
          [&](RowAcc& a, int t, int v) {
            const int w = p_index_of_row(t, v);
            float dw;
            if (t >= dinv_rows) {   // first pass to reach this row: its degree comes from this walk
              dw = sop2 ? gdinv(v) : (a.n > 0 ? 1.0f / sqrtf((float)a.n) : 0.0f);
              dinvP[w] = dw;
              edges_local += a.n;
            } else {
              dw = dinvP[w];
            }
            const float rx = dw * a.x, ry = dw * a.y;
            s_out[w] = make_float2(dw * rx, dw * ry);
            // (sop2: the partner's column is zeroed in the product with X — tuned_SIGN.py:73-76)
            coef[cidx(i, t)] = make_float2((sop2 && v == dst) ? 0.f : rx, (sop2 && v == src) ? 0.f : ry);
            // label column of operator i+1: Σ_w r[w] z_w = r[src] + r[dst]  (tuned_SIGN.py:177-185)
            if (v == src) { zbuf[(0 * K + i) * 2] = rx; zbuf[(0 * K + i) * 2 + 1] = ry; }
            if (v == dst) { zbuf[(1 * K + i) * 2] = rx; zbuf[(1 * K + i) * 2 + 1] = ry; }
          });
      for (int t = limit + tid; t < support; t += T) coef[cidx(i, t)] = make_float2(0.f, 0.f);
      dinv_rows = max(dinv_rows, limit);
      if (build_bm) bm_ready = true;
      __syncthreads();
      float2* tmp = s_in;
      s_in = s_out;
      s_out = tmp;
    }
    S3GRL_STAMP(3)
    if (bm_ready && last_rows == n && support == n) {   // last operator through the bit matrix
      edges_bm = 0;
      bm_pass(K - 1, true);
      if (pr == 0) edges_exact = edges_bm;
      __syncthreads();
    } else {  // last operator: degree and sum of every reachable row in one pass over its CSR row
      const int i = K - 1;
      int edges_pass = 0;
      walk_split(
          last_rows,
          [&](RowAcc& a, int v, int u, bool valid) {
            bool member, on;
            float2 sv;
            if constexpr (DM) {
              const int r = dmap[u];
              sv = s_in[min(r, p - 1)];
              member = valid && r != 0xffff;
              on = valid && r < p;
            } else if constexpr (HS) {
              const int slot = hs_find(hkeys, hmask, u);
              const int r = hvals[max(slot, 0)];
              sv = s_in[min(max(r, 0), p - 1)];
              member = valid && slot >= 0;
              on = member && r < p;
            } else {
              // all LDS reads unconditional, count and contribution selected afterwards
              const uint32_t bit = 1u << (u & 31);
              const uint32_t wv = vis[u >> 5], wp = inP[u >> 5];
              const int r = (int)wpreP[u >> 5] + __popc(wp & (bit - 1u));
              sv = s_in[min(r, p - 1)];
              member = valid && (wv & bit);
              on = member && (wp & bit);
            }
            const int mp = v == msrc ? dst : (v == mdst ? src : -1);
            const bool masked = u == mp;
            member = member && !masked;
            on = on && !masked;
            a.n += member ? 1 : 0;
            a.x += on ? sv.x : 0.f;
            a.y += on ? sv.y : 0.f;
          },
          [&](RowAcc& a, int t, int v) {
            edges_pass += a.n;
            if (t < support) {
              // (directed: a.n counted predecessors; D^-1/2 is the out-degree's, known for all of S)
              const float dw = DIRECTED ? dinvP[p_index_of_row(t, v)]
                                        : (sop2 ? gdinv(v) : (a.n > 0 ? 1.0f / sqrtf((float)a.n) : 0.0f));
              const float rx = dw * a.x, ry = dw * a.y;
              coef[cidx(i, t)] = make_float2((sop2 && v == dst) ? 0.f : rx, (sop2 && v == src) ? 0.f : ry);
              if (v == src) { zbuf[(0 * K + i) * 2] = rx; zbuf[(0 * K + i) * 2 + 1] = ry; }
              if (v == dst) { zbuf[(1 * K + i) * 2] = rx; zbuf[(1 * K + i) * 2 + 1] = ry; }
            }
          });
      if (pr == 0 && last_rows == n) edges_exact = edges_pass;
      __syncthreads();
    }
    S3GRL_STAMP(4)
    if (tid < 2 * K) {
      const int i = tid >> 1, r = tid & 1;
      // label column of operator i+1: r[src] + r[dst]; sop2: the diagonal entry — r_a[src] for row a, r_b[dst] for b
      job_z[(jid * K + i) * 2 + r] = sop2 ? zbuf[(r * K + i) * 2 + r]
                                          : zbuf[(0 * K + i) * 2 + r] + zbuf[(1 * K + i) * 2 + r];
    }
    // operator i+1 reaches the list prefix within i+1 hops of the row (the limits of the passes
    // above): the gather skips its multiply-adds beyond that
    if (tid < K) job_lim[jid * K + tid] = tid == K - 1 ? support : lvl_end[min(tid + 1 + row_hop, nlev - 1)];
    if (tid == 0) {
      Job j;
      j.coef_off = coff * K;
      j.ids_off = noff;
      j.out_row = rp + 2 * pr;
      j.link = l;
      j.support = support;
      j.node_a = ext(node_a);
      j.node_b = node_b >= 0 ? ext(node_b) : -1;
      j.z_a = (node_a == src || node_a == dst) ? 1 : 0;
      j.z_b = (node_b == src || node_b == dst) ? 1 : 0;
      j.mirror_row = mirror >= 0 ? mrp + 2 * pr : -1;
      j.mirror_swap = pr == 0 ? 1 : 0;
      j.split = split ? 1 : 0;
      jobs[jid] = j;
      atomicAdd(stat_slot(tot_support), (unsigned long long)support * (mirror >= 0 ? 2ull : 1ull));
    }
    __syncthreads();
  }
  // edges of the masked induced subgraph: exact when the last pass of pair 0 covered all of S
  // (always with full_stats; otherwise whenever K >= num_hops), else the edges of P's rows
  S3GRL_STAMP(5)
  edges_local = block_sum<T>(edges_exact >= 0 ? edges_exact : edges_local, sh);
  vol_local = block_sum<T>(vol_local, sh);
  if (tid == 0) {
    atomicAdd(stat_slot(tot_edges), (unsigned long long)edges_local * (mirror >= 0 ? 2ull : 1ull));
    atomicAdd(stat_slot(tot_vol), (unsigned long long)vol_local * (mirror >= 0 ? 2ull : 1ull));
  }
}

// hop distance of every exported node from the per-link level ends
__global__ void dists_kernel(const int64_t* __restrict__ node_off, const int32_t* __restrict__ lvl,
                             int64_t L, int8_t* __restrict__ dists) {
  const int64_t l = blockIdx.x;
  const int64_t o = node_off[l];
  const int n = (int)(node_off[l + 1] - o);
  const int32_t* lv = lvl + l * kMaxLevels;
  for (int t = threadIdx.x; t < n; t += blockDim.x) {
    int d = 0;
    while (d < kMaxLevels - 1 && t >= lv[d]) ++d;
    dists[o + t] = (int8_t)d;
  }
}

}  // namespace

// ---------------------------------------------------------------------------------------
static inline int words_for(int64_t N) { return (int)((N + 31) / 32); }

#ifndef S3GRL_LINKS_PART   // (the link-kernel translation units compile only the launch code, see the end of the file)
int64_t mirror_table_slots(int64_t L) {
  int64_t s = 1024;
  while (s < 2 * L) s <<= 1;
  return s;
}

s3grl_status launch_find_mirrors(s3grl_context* ctx, const int64_t* links, int64_t L, int64_t N,
                                 uint64_t* keys, int32_t* vals, int64_t slots, int32_t* partner,
                                 int32_t* mirror_of, int64_t* n_mirrored) {
  if (L == 0) return S3GRL_OK;
  S3GRL_HIP_TRY(hipMemsetAsync(keys, 0xff, (size_t)slots * 8, ctx->stream));
  S3GRL_HIP_TRY(hipMemsetAsync(vals, 0x7f, (size_t)slots * 4, ctx->stream));
  S3GRL_HIP_TRY(hipMemsetAsync(partner, 0xff, (size_t)L * 4, ctx->stream));
  S3GRL_HIP_TRY(hipMemsetAsync(mirror_of, 0xff, (size_t)L * 4, ctx->stream));
  const unsigned grid = (unsigned)((L + 255) / 256);
  hipLaunchKernelGGL(mirror_insert_kernel, dim3(grid), dim3(256), 0, ctx->stream, links, L, N, keys,
                     vals, (uint64_t)(slots - 1));
  hipLaunchKernelGGL(mirror_lookup_kernel, dim3(grid), dim3(256), 0, ctx->stream, links, L, N, keys,
                     vals, (uint64_t)(slots - 1), partner, mirror_of,
                     reinterpret_cast<unsigned long long*>(n_mirrored));
  S3GRL_HIP_TRY(hipGetLastError());
  return S3GRL_OK;
}

s3grl_status launch_random_walks(s3grl_context* ctx, const s3grl_graph* g, int m, int M,
                                 uint32_t seed, int32_t* raw) {
  const int64_t total = g->num_nodes * M;
  // (a directed graph is walked along its arcs, like torch_cluster walks the directed edge_index)
  hipLaunchKernelGGL(random_walks_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0,
                     ctx->stream, g->directed ? g->out_indptr : g->indptr, g->directed ? g->out_indices : g->indices,
                     g->num_nodes, m, M, seed, raw);
  S3GRL_HIP_TRY(hipGetLastError());
  return S3GRL_OK;
}

s3grl_status launch_validate_sets(s3grl_context* ctx, const int64_t* set_ptr, const int32_t* set_nodes,
                                  int64_t num_sets, int64_t total, int64_t num_nodes, int64_t* flags) {
  if (num_sets == 0) return S3GRL_OK;
  hipLaunchKernelGGL(validate_sets_kernel, dim3(512), dim3(256), 0, ctx->stream, set_ptr, set_nodes, num_sets,
                     total, num_nodes, flags);
  S3GRL_HIP_TRY(hipGetLastError());
  return S3GRL_OK;
}

s3grl_status launch_walk_sets(s3grl_context* ctx, const s3grl_graph* g, const int64_t* starts, int64_t num_starts,
                              int m, int M, uint32_t seed, int64_t* set_ptr, int32_t* set_nodes) {
  if (num_starts == 0) {
    S3GRL_HIP_TRY(hipMemsetAsync(set_ptr, 0, 8, ctx->stream));
    return S3GRL_OK;
  }
  const int64_t cap64 = (int64_t)m * M + 1;
  int P = 2;
  while (P < cap64 && P < kWalkSetMax) P <<= 1;
  if (cap64 > P) {
    set_last_error("rw_m * rw_M + 1 = " + std::to_string(cap64) + " entries per node exceed " +
                   std::to_string(kWalkSetMax));
    return S3GRL_ERR_GRAPH_TOO_LARGE;
  }
  const int cap = (int)cap64;
  Transient tmp{ctx, {}};
  void *scratch = nullptr, *cnt = nullptr, *ws = nullptr;
  S3GRL_TRY(ctx->arena.alloc((size_t)num_starts * cap * 4, &scratch));
  tmp.ptrs.push_back(scratch);
  S3GRL_TRY(ctx->arena.alloc((size_t)num_starts * 4, &cnt));
  tmp.ptrs.push_back(cnt);
  S3GRL_TRY(ctx->arena.alloc((size_t)scan_workspace_elems(num_starts) * 8, &ws));
  tmp.ptrs.push_back(ws);
  S3GRL_HIP_TRY(hipMemsetAsync(ctx->d_scalars, 0, 8, ctx->stream));
  hipLaunchKernelGGL(walk_sets_kernel, dim3((unsigned)num_starts), dim3(kWalkSetBlock), (size_t)P * 4, ctx->stream,
                     g->directed ? g->out_indptr : g->indptr, g->directed ? g->out_indices : g->indices,
                     g->num_nodes, starts, m, M, seed, P, cap, static_cast<int32_t*>(scratch),
                     static_cast<int32_t*>(cnt), reinterpret_cast<int32_t*>(ctx->d_scalars));
  S3GRL_HIP_TRY(hipGetLastError());
  S3GRL_TRY(launch_scan_i32_to_i64(ctx, static_cast<int32_t*>(cnt), num_starts, set_ptr,
                                   static_cast<int64_t*>(ws)));
  hipLaunchKernelGGL(walk_sets_compact_kernel, dim3((unsigned)num_starts), dim3(64), 0, ctx->stream,
                     static_cast<int32_t*>(scratch), cap, set_ptr, set_nodes);
  S3GRL_HIP_TRY(hipGetLastError());
  S3GRL_HIP_TRY(hipMemcpyAsync(ctx->h_scalars, ctx->d_scalars, 8, hipMemcpyDeviceToHost, ctx->stream));
  S3GRL_HIP_TRY(hipStreamSynchronize(ctx->stream));   // tmp is released on return
  if (ctx->h_scalars[0] & 0xffffffff) {
    set_last_error("a start node is outside [0, num_nodes)");
    return S3GRL_ERR_INVALID_ARGUMENT;
  }
  return S3GRL_OK;
}

s3grl_status launch_split_count(s3grl_context* ctx, const Job* jobs, int64_t njobs, int seg_shift,
                                int32_t* cnt, int64_t* piece_off, int64_t* scan_ws) {
  if (njobs == 0) return S3GRL_OK;
  hipLaunchKernelGGL(split_count_kernel, dim3((unsigned)((njobs + 255) / 256)), dim3(256), 0, ctx->stream, jobs,
                     njobs, seg_shift, cnt);
  S3GRL_HIP_TRY(hipGetLastError());
  return launch_scan_i32_to_i64(ctx, cnt, njobs, piece_off, scan_ws);
}

s3grl_status launch_split_fill(s3grl_context* ctx, const Job* jobs, const int32_t* job_lim, int64_t njobs,
                               const int32_t* job_order, int K, int seg_shift, const int64_t* piece_off,
                               int64_t npieces, Job* gjobs, int32_t* g_lim, int32_t* g_order, int32_t* piece_job) {
  if (njobs == 0) return S3GRL_OK;
  hipLaunchKernelGGL(split_fill_kernel, dim3((unsigned)((njobs + 255) / 256)), dim3(256), 0, ctx->stream, jobs,
                     job_lim, job_order, njobs, K, seg_shift, piece_off, npieces, gjobs, g_lim, g_order, piece_job);
  S3GRL_HIP_TRY(hipGetLastError());
  return S3GRL_OK;
}

s3grl_status launch_combine(s3grl_context* ctx, const s3grl_plan* p, const float* prows, const float* X,
                            int64_t ldx, int64_t F, float* rows) {
  if (p->njobs == 0 || p->npieces == 0) return S3GRL_OK;
  hipLaunchKernelGGL(combine_kernel, dim3((unsigned)p->npieces, 2), dim3(256), 0, ctx->stream, p->jobs,
                     p->piece_off, p->piece_job, p->job_z, p->cfg.sign_k, prows, X, ldx, (int)F, rows);
  S3GRL_HIP_TRY(hipGetLastError());
  return S3GRL_OK;
}

s3grl_status launch_mirror_rows(s3grl_context* ctx, const int32_t* partner, int64_t L,
                                int32_t* n_rows) {
  if (L == 0) return S3GRL_OK;
  hipLaunchKernelGGL(mirror_rows_kernel, dim3((unsigned)((L + 255) / 256)), dim3(256), 0, ctx->stream,
                     partner, L, n_rows);
  S3GRL_HIP_TRY(hipGetLastError());
  return S3GRL_OK;
}

s3grl_status launch_count(s3grl_context* ctx, const s3grl_graph* g, const int64_t* links, int64_t L,
                          int hops, int plus, int K, WalkSets ws,
                          const int32_t* partner, const int32_t* mirror_of, int32_t* n_nodes, int32_t* p_nodes,
                          int32_t* n_rows, int32_t* n_jobs, int32_t* lvl_max, int32_t* err_flag,
                          int64_t* tot_nodes_alg, HopSampling smp, int32_t* stash, int slot,
                          int32_t* lvl_stash, const int32_t* perm) {
  if (L == 0) return S3GRL_OK;
  const int W = words_for(g->num_nodes);
  const int nbm = (hops <= 1 || walks_on(ws)) ? 2 : 3;   // see count_kernel: no frontier bitmap for one hop
  const int nsets = nbm + (hop_sampling_on(smp) ? 1 : 0);
  const size_t lds = (size_t)(nsets * W + 8 + kHubWords + kCountList) * 4;
  const bool sparse = (double)g->nnz / (double)std::max<int64_t>(g->num_nodes, 1) <= 6.0;
  const int32_t* cn_ip = g->directed ? g->out_indptr : g->indptr;
  const int32_t* cn_ix = g->directed ? g->out_indices : g->indices;
  if (lds > 163840 || getenv("S3GRL_FORCE_EXT_BITMAPS")) {
    // The N-bit bitmaps do not fit a CU's LDS (num_nodes > ~327 680): they live in HBM, one slice per
    // workgroup of a launch, and the list is processed in chunks that share the slices.
    const int64_t stride = ((int64_t)nsets * W + 63) / 64 * 64;
    const int chunk = (int)std::min<int64_t>(L, std::max<int64_t>(256, ((int64_t)1 << 29) / (stride * 4)));   // <= 512 MiB of slices
    Transient tmp{ctx, {}};
    void* q = nullptr;
    S3GRL_TRY(ctx->arena.alloc((size_t)stride * 4 * chunk, &q));
    tmp.ptrs.push_back(q);
    const size_t lds_ext = (size_t)(8 + kHubWords + kCountList) * 4;
    auto kern = sparse ? count_kernel<4, 256, true> : count_kernel<8, 256, true>;
    for (int64_t base = 0; base < L; base += chunk) {
      const int64_t cnt = std::min<int64_t>(chunk, L - base);
      hipLaunchKernelGGL(kern, dim3((unsigned)cnt), dim3(256), lds_ext, ctx->stream, g->indptr, g->indices,
                         (int)g->num_nodes, W, links, hops, plus, K, g->max_degree > kHubArmDegree ? 1 : 0, ws,
                         partner, mirror_of, n_nodes, p_nodes, n_rows, n_jobs, lvl_max, err_flag,
                         reinterpret_cast<unsigned long long*>(tot_nodes_alg), smp, stash, slot, lvl_stash, cn_ip,
                         cn_ix, g->directed ? 1 : 0, static_cast<uint32_t*>(q), stride, (int)base, perm);
      S3GRL_HIP_TRY(hipGetLastError());
    }
    S3GRL_HIP_TRY(hipStreamSynchronize(ctx->stream));   // the slices are released on return
    return S3GRL_OK;
  }
  const bool small = lds <= 24 * 1024;
  auto kern = small ? (sparse ? count_kernel<4, 128> : count_kernel<8, 128>)
                    : (sparse ? count_kernel<4, 256> : count_kernel<8, 256>);
  S3GRL_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(kern),
                                    hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  hipLaunchKernelGGL(kern, dim3((unsigned)L), dim3(small ? 128 : 256), lds, ctx->stream, g->indptr,
                     g->indices, (int)g->num_nodes, W, links, hops, plus, K,
                     g->max_degree > kHubArmDegree ? 1 : 0, ws, partner, mirror_of, n_nodes,
                     p_nodes, n_rows, n_jobs, lvl_max, err_flag,
                     reinterpret_cast<unsigned long long*>(tot_nodes_alg), smp, stash, slot, lvl_stash,
                     cn_ip, cn_ix, g->directed ? 1 : 0, nullptr, 0, 0, perm);
  S3GRL_HIP_TRY(hipGetLastError());
  return S3GRL_OK;
}

// ---------------------------------------------------------------------------------------
// Gather order: one wavefront per job, and the jobs differ in length by two orders of magnitude.
// Started in list order, a long job that begins late is the tail of the launch; started longest
// first (LPT) the tail is made of short ones (PubMed: 10.7 -> 9.7 ms).  A counting sort of the
// jobs by their link's node count in kOrderBuckets descending buckets; inside a bucket the order
// is whatever the atomics give (results do not depend on it).
constexpr int kOrderBuckets = 256;
constexpr int kOrderShift = 5;   // 32 nodes per bucket, everything >= 8160 nodes in the first

__device__ __forceinline__ int order_bucket(int n) {
  return kOrderBuckets - 1 - min(n >> kOrderShift, kOrderBuckets - 1);
}

// Both passes keep a workgroup-local histogram in LDS and touch the global one once per
// (workgroup, bucket): 164 000 global atomics on 256 counters cost 0.2 ms per pass otherwise.
constexpr int kOrderThreads = 1024;

__global__ __launch_bounds__(kOrderThreads) void order_hist_kernel(
    const int32_t* __restrict__ n_nodes, const int32_t* __restrict__ n_jobs, int64_t L, int64_t per_block,
    int32_t* __restrict__ hist) {
  __shared__ int h[kOrderBuckets];
  const int t = threadIdx.x;
  if (t < kOrderBuckets) h[t] = 0;
  __syncthreads();
  const int64_t l0 = (int64_t)blockIdx.x * per_block, l1 = min(l0 + per_block, L);
  for (int64_t l = l0 + t; l < l1; l += kOrderThreads) {
    const int nj = n_jobs[l];
    if (nj > 0) atomicAdd(&h[order_bucket(n_nodes[l])], nj);
  }
  __syncthreads();
  if (t < kOrderBuckets && h[t]) atomicAdd(&hist[t], h[t]);
}

__global__ void order_scan_kernel(int32_t* __restrict__ hist /* in: counts, out: cursors */) {
  __shared__ int sh[kOrderBuckets];
  const int t = threadIdx.x;
  sh[t] = hist[t];
  __syncthreads();
  if (t == 0) {
    int run = 0;
    for (int b = 0; b < kOrderBuckets; ++b) {
      const int c = sh[b];
      sh[b] = run;
      run += c;
    }
  }
  __syncthreads();
  hist[t] = sh[t];
}

__global__ __launch_bounds__(kOrderThreads) void order_fill_kernel(
    const int32_t* __restrict__ n_nodes, const int32_t* __restrict__ n_jobs,
    const int64_t* __restrict__ job_off, int64_t L, int64_t per_block, int32_t* __restrict__ cursor,
    int32_t* __restrict__ job_order) {
  __shared__ int h[kOrderBuckets];     // this workgroup's jobs per bucket, then its running cursor
  __shared__ int base[kOrderBuckets];  // where its share of the bucket starts
  const int t = threadIdx.x;
  if (t < kOrderBuckets) h[t] = 0;
  __syncthreads();
  const int64_t l0 = (int64_t)blockIdx.x * per_block, l1 = min(l0 + per_block, L);
  for (int64_t l = l0 + t; l < l1; l += kOrderThreads) {
    const int nj = n_jobs[l];
    if (nj > 0) atomicAdd(&h[order_bucket(n_nodes[l])], nj);
  }
  __syncthreads();
  if (t < kOrderBuckets) {
    base[t] = h[t] ? atomicAdd(&cursor[t], h[t]) : 0;
    h[t] = 0;
  }
  __syncthreads();
  for (int64_t l = l0 + t; l < l1; l += kOrderThreads) {
    const int nj = n_jobs[l];
    if (nj <= 0) continue;
    const int b = order_bucket(n_nodes[l]);
    const int at = base[b] + atomicAdd(&h[b], nj);
    const int j0 = (int)job_off[l];
    for (int j = 0; j < nj; ++j) job_order[at + j] = j0 + j;
  }
}

// Plans with a processing order (launch_link_order): the gather takes the jobs link by link in that
// order — neighbours in the order read the same rows of X — instead of longest first.
__global__ void perm_jobs_kernel(const int32_t* __restrict__ perm, const int32_t* __restrict__ n_jobs, int64_t L,
                                 int32_t* __restrict__ cnt) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < L) cnt[i] = n_jobs[perm[i]];
}

__global__ void perm_fill_kernel(const int32_t* __restrict__ perm, const int32_t* __restrict__ n_jobs,
                                 const int64_t* __restrict__ job_off, const int64_t* __restrict__ at, int64_t L,
                                 int32_t* __restrict__ job_order) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= L) return;
  const int l = perm[i];
  const int nj = n_jobs[l];
  const int j0 = (int)job_off[l];
  const int64_t a = at[i];
  for (int j = 0; j < nj; ++j) job_order[a + j] = j0 + j;
}

s3grl_status launch_job_order(s3grl_context* ctx, const int32_t* n_nodes, const int32_t* n_jobs,
                              const int64_t* job_off, int64_t L, int32_t* hist /* [256] scratch */,
                              int32_t* job_order, const int32_t* perm, int32_t* scratch_cnt,
                              int64_t* scratch_off, int64_t* scan_ws) {
  if (L == 0) return S3GRL_OK;
  if (perm) {
    const unsigned grid = (unsigned)((L + 255) / 256);
    hipLaunchKernelGGL(perm_jobs_kernel, dim3(grid), dim3(256), 0, ctx->stream, perm, n_jobs, L, scratch_cnt);
    S3GRL_HIP_TRY(hipGetLastError());
    S3GRL_TRY(launch_scan_i32_to_i64(ctx, scratch_cnt, L, scratch_off, scan_ws));
    hipLaunchKernelGGL(perm_fill_kernel, dim3(grid), dim3(256), 0, ctx->stream, perm, n_jobs, job_off, scratch_off,
                       L, job_order);
    S3GRL_HIP_TRY(hipGetLastError());
    return S3GRL_OK;
  }
  S3GRL_HIP_TRY(hipMemsetAsync(hist, 0, kOrderBuckets * sizeof(int32_t), ctx->stream));
  const int64_t per_block = std::max<int64_t>((L + 255) / 256, 4 * kOrderThreads);
  const unsigned grid = (unsigned)((L + per_block - 1) / per_block);
  hipLaunchKernelGGL(order_hist_kernel, dim3(grid), dim3(kOrderThreads), 0, ctx->stream, n_nodes, n_jobs, L,
                     per_block, hist);
  hipLaunchKernelGGL(order_scan_kernel, dim3(1), dim3(kOrderBuckets), 0, ctx->stream, hist);
  hipLaunchKernelGGL(order_fill_kernel, dim3(grid), dim3(kOrderThreads), 0, ctx->stream, n_nodes, n_jobs, job_off,
                     L, per_block, hist, job_order);
  S3GRL_HIP_TRY(hipGetLastError());
  return S3GRL_OK;
}

int64_t scan_workspace_elems(int64_t n) { return (n + kScanTile - 1) / kScanTile + 1; }

s3grl_status launch_scan3(s3grl_context* ctx, const int32_t* in0, const int32_t* in1, const int32_t* in2,
                          int64_t n, int64_t* out0, int64_t* out1, int64_t* out2, int64_t* workspace3,
                          int64_t* max0, int64_t* max1, int64_t* totals) {
  const int64_t nb = (n + kScanTile - 1) / kScanTile;
  const int64_t wsz = scan_workspace_elems(n);
  Scan3 a;
  a.in[0] = in0; a.in[1] = in1; a.in[2] = in2;
  a.out[0] = out0; a.out[1] = out1; a.out[2] = out2;
  for (int y = 0; y < 3; ++y) {
    a.part[y] = workspace3 + y * wsz;
    a.total_out[y] = totals ? totals + y : nullptr;
  }
  a.max_out[0] = reinterpret_cast<long long*>(max0);
  a.max_out[1] = reinterpret_cast<long long*>(max1);
  a.max_out[2] = nullptr;
  hipLaunchKernelGGL(scan3_partials_kernel, dim3((unsigned)nb, 3), dim3(256), 0, ctx->stream, a, n);
  hipLaunchKernelGGL(scan3_top_kernel, dim3(3), dim3(1024), 0, ctx->stream, a, nb, n);
  hipLaunchKernelGGL(scan3_apply_kernel, dim3((unsigned)nb, 3), dim3(256), 0, ctx->stream, a, n);
  S3GRL_HIP_TRY(hipGetLastError());
  return S3GRL_OK;
}

s3grl_status launch_scan_i32_to_i64(s3grl_context* ctx, const int32_t* in, int64_t n, int64_t* out,
                                    int64_t* workspace) {
  if (n == 0) {
    S3GRL_HIP_TRY(hipMemsetAsync(out, 0, 8, ctx->stream));
    return S3GRL_OK;
  }
  const int64_t nb = (n + kScanTile - 1) / kScanTile;
  hipLaunchKernelGGL(scan_partials_kernel, dim3((unsigned)nb), dim3(256), 0, ctx->stream, in, n,
                     workspace);
  hipLaunchKernelGGL(scan_top_kernel, dim3(1), dim3(1024), 0, ctx->stream, workspace, nb, out + n);
  hipLaunchKernelGGL(scan_apply_kernel, dim3((unsigned)nb), dim3(256), 0, ctx->stream, in, n,
                     workspace, out);
  S3GRL_HIP_TRY(hipGetLastError());
  return S3GRL_OK;
}

#endif  // !S3GRL_LINKS_PART
// fixed part of link_kernel's LDS: 3 bitmaps + cn + lvl_end + zbuf + scan scratch + hub list
static inline int link_fixed_words(int64_t num_nodes, int cn_cap, int K) {
  return 3 * words_for(num_nodes) + cn_cap + kMaxLevels + 4 * K + 32 + kHubWords;
}
static inline int link_fixed_words_sparse(int cn_cap, int K) {
  return cn_cap + kMaxLevels + 4 * K + 32 + kHubWords;
}

// nominal class bounds (S3GRL_BOUNDS = "b0,b1,..." is a tuning hook)
static const int* nominal_bounds() {
  static int b[kNumClasses] = S3GRL_CLASS_BOUNDS;
  static bool parsed = false;
  if (!parsed) {
    parsed = true;
    if (const char* e = getenv("S3GRL_BOUNDS")) {
      int k = 0;
      for (const char* q = e; *q && k < kNumClasses; ++k) {
        b[k] = atoi(q);
        while (*q && *q != ',') ++q;
        if (*q == ',') ++q;
      }
    }
  }
  return b;
}

// class c holds the links whose variable LDS need is <= bound[c] bytes; the last bound is
// whatever the 160 KiB of a CU leave after the fixed part
static ClassBounds class_bounds(int64_t num_nodes, int cn_cap, int K) {
  const int* nominal = nominal_bounds();
  int avail = 163840 - 4 * link_fixed_words(num_nodes, cn_cap, K);
  if (const char* e = getenv("S3GRL_LDS_BUDGET")) avail = std::min(avail, atoi(e));  // test hook
  ClassBounds cb;
  for (int c = 0; c < kNumClasses; ++c) cb.b[c] = std::min(nominal[c], avail);
  cb.b[kNumClasses - 1] = avail;
  return cb;
}
static ClassBounds class_bounds_sparse(int cn_cap, int K) {
  static const int nominal[kNumClasses] = {4096, 8192, 16384, 32768, 65536, 131072};
  const int avail = 163840 - 4 * link_fixed_words_sparse(cn_cap, K);
  ClassBounds cb;
  for (int c = 0; c < kNumClasses; ++c) cb.b[c] = std::min(nominal[c], avail);
  return cb;
}

// link_full_kernel: fixed LDS = cn + cnpos + lvl_end[2] + zbuf + scan scratch + long-row list
static inline int full_fixed_words(int cn_cap, int K) { return 2 * cn_cap + 2 + 4 * K + 32 + kLongCap / 2; }
static ClassBounds class_bounds_full(int cn_cap, int K) {
  static const int nominal[kNumClasses] = {3072, 6144, 12288, 24576, 65536, 160000};
  const int avail = 163840 - 4 * full_fixed_words(cn_cap, K);
  ClassBounds cb;
  for (int c = 0; c < kNumClasses; ++c) cb.b[c] = std::min(nominal[c], avail);
  return cb;
}

// threads per link of an LDS class (S3GRL_T_CLASS<c> = tuning hook)
static int threads_for_class(size_t lds, int c) {
  char name[32];
  snprintf(name, sizeof(name), "S3GRL_T_CLASS%d", c);
  if (const char* e = getenv(name)) return atoi(e);
  // the smallest subgraphs (a few hundred nodes at most): two wavefronts per link — the uniform part
  // of the kernel is most of their cost, and ten such links fit a CU either way
  if (c == 0 && lds <= 40 * 1024) return 128;
  return lds <= 40 * 1024 ? 256 : (lds <= 80 * 1024 ? 512 : 1024);
}

static inline int link_fixed_words_dm(int64_t num_nodes, int cn_cap, int K) {
  return 16 * words_for(num_nodes) + cn_cap + kMaxLevels + 4 * K + 32 + kHubWords;
}
static ClassBounds class_bounds_dm(int64_t num_nodes, int cn_cap, int K) {
  const int* nominal = nominal_bounds();
  const int avail = 163840 - 4 * link_fixed_words_dm(num_nodes, cn_cap, K);
  ClassBounds cb;
  for (int c = 0; c < kNumClasses; ++c) cb.b[c] = std::min(nominal[c], avail);
  cb.b[kNumClasses - 1] = avail;
  return cb;
}

// The hash flavour pays off when the bitmaps alone would hold a CU to a few workgroups.
#ifndef S3GRL_LINKS_PART
bool sparse_mode_for(const s3grl_graph* g) {
  if (getenv("S3GRL_FORCE_HASH")) return true;   // test hook
  if (getenv("S3GRL_NO_HASH")) return false;
  return 3 * (size_t)words_for(g->num_nodes) * 4 > 24 * 1024;
}
#endif

// The direct-map flavour: graphs whose 2N-byte map leaves nearly all of a CU's LDS to the lists.
static bool dm_mode_for(const s3grl_graph* g) {
  if (sparse_mode_for(g) || getenv("S3GRL_NO_DM")) return false;
  if (getenv("S3GRL_FORCE_DM")) return g->num_nodes <= 65535;   // test hook
  // measured after the degree order: USAir (332 nodes) link kernels 0.077 -> 0.058 ms, Cora (2 708)
  // 0.35 -> 0.34, PubMed (19 717: 39 KB of map per link) 3.94 -> 4.13 — the map has to be small
  return g->num_nodes <= 8192;
}

// The map costs LDS, i.e. resident wavefronts: class by class, the direct-map flavour is used where
// it fits at least 4/5 of the waves the bitmap flavour fits on a CU (measured: a loss of up to 1/5
// is paid back by the cheaper visits; PubMed, 39 KB of map: every class but the smallest).
static int waves_per_cu(size_t lds, int c) {
  return std::min<int>(32, (int)(163840 / std::max<size_t>(lds, 1)) * (threads_for_class(lds, c) / 64));
}
static int dm_class_mask_for(const s3grl_graph* g, int cn_cap, int K) {
  if (const char* e = getenv("S3GRL_DM_CLASS_MASK")) return atoi(e);   // tuning hook
  const ClassBounds bb = class_bounds(g->num_nodes, cn_cap, K), bd = class_bounds_dm(g->num_nodes, cn_cap, K);
  int mask = 0;
  for (int c = 0; c < kNumClasses; ++c) {
    if (bd.b[c] <= 0) continue;
    if (bb.b[c] <= 0) { mask |= 1 << c; continue; }
    const int wb = waves_per_cu((size_t)4 * link_fixed_words(g->num_nodes, cn_cap, K) + bb.b[c], c);
    const int wd = waves_per_cu((size_t)4 * link_fixed_words_dm(g->num_nodes, cn_cap, K) + bd.b[c], c);
    // the smallest class is bound by links in flight, not by waves: at least half as many must fit
    const size_t lb = (size_t)4 * link_fixed_words(g->num_nodes, cn_cap, K) + bb.b[c];
    const size_t ld = (size_t)4 * link_fixed_words_dm(g->num_nodes, cn_cap, K) + bd.b[c];
    if (c == 0 && 2 * std::min<size_t>(163840 / ld, 16) < std::min<size_t>(163840 / lb, 16)) continue;
    if (5 * wd >= 4 * wb) mask |= 1 << c;
  }
  return mask;
}

#ifndef S3GRL_LINKS_PART
int num_class_lists() { return kNumListsAll; }

// One-hop plans take the row-intersection path on graphs where the hash flavour is in use anyway.
bool onehop_mode_for(const s3grl_graph* g) {
  if (getenv("S3GRL_NO_ONEHOP")) return false;
  if (getenv("S3GRL_FORCE_ONEHOP")) return true;   // test hook
  return sparse_mode_for(g);
}

s3grl_status build_forward_rows(s3grl_context* ctx, s3grl_graph* g) {
  const int64_t N = g->num_nodes;
  Transient tmp{ctx, {}};
  void* q = nullptr;
  S3GRL_TRY(ctx->arena.alloc((size_t)N * 4, &q));
  tmp.ptrs.push_back(q);
  int32_t* cnt = static_cast<int32_t*>(q);
  S3GRL_TRY(ctx->arena.alloc((size_t)(N + 1) * 8, &q));
  tmp.ptrs.push_back(q);
  int64_t* off64 = static_cast<int64_t*>(q);
  S3GRL_TRY(ctx->arena.alloc((size_t)scan_workspace_elems(N) * 8, &q));
  tmp.ptrs.push_back(q);
  int64_t* ws = static_cast<int64_t*>(q);
  const unsigned grid = (unsigned)((N + 255) / 256);
  hipLaunchKernelGGL(fwd_count_kernel, dim3(grid), dim3(256), 0, ctx->stream, g->indptr, g->indices, N, cnt);
  S3GRL_HIP_TRY(hipGetLastError());
  S3GRL_TRY(launch_scan_i32_to_i64(ctx, cnt, N, off64, ws));
  S3GRL_HIP_TRY(hipMemcpyAsync(ctx->h_scalars, off64 + N, 8, hipMemcpyDeviceToHost, ctx->stream));
  S3GRL_HIP_TRY(hipStreamSynchronize(ctx->stream));
  const int64_t fnnz = ctx->h_scalars[0];
  S3GRL_TRY(ctx->arena.alloc((size_t)(N + 1) * 4, &q));
  g->fwd_indptr = static_cast<int32_t*>(q);
  S3GRL_TRY(ctx->arena.alloc((size_t)std::max<int64_t>(fnnz, 1) * 4, &q));
  g->fwd_indices = static_cast<int32_t*>(q);
  S3GRL_TRY(ctx->arena.alloc((size_t)N * 2, &q));
  g->fwd_deg = static_cast<uint16_t*>(q);
  hipLaunchKernelGGL(fwd_fill_kernel, dim3((unsigned)((N + 1 + 255) / 256)), dim3(256), 0, ctx->stream,
                     g->indptr, g->indices, N, off64, g->fwd_indptr, g->fwd_indices, g->fwd_deg);
  S3GRL_HIP_TRY(hipGetLastError());
  S3GRL_HIP_TRY(hipStreamSynchronize(ctx->stream));   // tmp is released on return
  return S3GRL_OK;
}

s3grl_status launch_count1(s3grl_context* ctx, const s3grl_graph* g, const int64_t* links, int64_t L,
                           int plus, int K, const int32_t* partner, const int32_t* mirror_of,
                           int32_t* n_nodes, int32_t* p_nodes, int32_t* n_rows, int32_t* n_jobs,
                           int32_t* lvl_max, int32_t* e_cap, int32_t* err_flag, int64_t* tot_nodes_alg,
                           int64_t* tot_oriented, const int32_t* perm, int64_t* x_cap) {
  if (L == 0) return S3GRL_OK;
  hipLaunchKernelGGL(count1_kernel, dim3((unsigned)((L + kCount1Waves - 1) / kCount1Waves)),
                     dim3(64 * kCount1Waves), 0, ctx->stream, g->indptr, g->indices, g->fwd_deg,
                     (int)g->num_nodes, links, L, plus, K, partner, mirror_of, n_nodes, p_nodes, n_rows,
                     n_jobs, lvl_max, e_cap, err_flag, reinterpret_cast<unsigned long long*>(tot_nodes_alg),
                     reinterpret_cast<unsigned long long*>(tot_oriented), perm, g->hub,
                     (x_cap && g->hub.nh > 0) ? x_cap : nullptr);
  S3GRL_HIP_TRY(hipGetLastError());
  return S3GRL_OK;
}

s3grl_status launch_classify(s3grl_context* ctx, const s3grl_graph* g, int cn_cap, int K,
                             const int32_t* n_nodes, const int32_t* p_nodes,
                             const int32_t* lvl_max, int64_t L, int32_t* class_count,
                             int32_t* class_list, bool allow_hash, const int32_t* e_cap, int stash_slot,
                             const int32_t* perm, const int64_t* x_cap, const int32_t* csr_e, bool tiny_ok) {
  if (L == 0) return S3GRL_OK;
  // link_tiny_kernel takes the smallest one-hop links of plans without common-neighbour rows (never a list that
  // a test hook would split: S3GRL_SPLIT_SEG_SHIFT below 6)
  const char* segs = getenv("S3GRL_SPLIT_SEG_SHIFT");
  const int tiny_max_n = (tiny_ok && !getenv("S3GRL_NO_TINY") && !(segs && atoi(segs) < 6)) ? kTinyNodes : 0;
  ClassBounds cb = class_bounds(g->num_nodes, cn_cap, K);
  const bool dm = allow_hash && stash_slot > 0 && dm_mode_for(g);
  if (cb.b[kNumClasses - 1] < 0 || getenv("S3GRL_FORCE_EXT_BITMAPS")) {
    // the N-bit bitmaps of the bitmap flavour do not fit LDS: apart from the hash / one-hop classes
    // there is only the HBM-scratch class, with its bitmaps in the slice too
    for (int c = 0; c < kNumClasses; ++c) cb.b[c] = -1;   // every link "overflows" the LDS classes
  }
  ClassBounds hb{};
  static_assert(kHubClasses <= kNumClasses, "hub class bounds travel in a ClassBounds");
  static_assert(kHubClasses < kNumClasses, "one more entry for the test hook");
  for (int c = 0; c < kHubClasses; ++c) hb.b[c] = hub_class_bound(c, cn_cap, K);
  hb.b[kHubClasses] = getenv("S3GRL_FORCE_HUB_SLICES") ? 1 : 0;   // test hook: found edges in HBM slices
  CsrBounds csrb{};
  for (int c = 0; c < kCsrClasses; ++c) csrb.b[c] = csr_class_bound(c, cn_cap, K);
  hipLaunchKernelGGL(classify_kernel, dim3((unsigned)((L + 255) / 256)), dim3(256), 0, ctx->stream,
                     n_nodes, p_nodes, lvl_max, L, cb, dm ? 2 : ((allow_hash && sparse_mode_for(g)) ? 1 : 0),
                     dm ? class_bounds_dm(g->num_nodes, cn_cap, K) : class_bounds_sparse(cn_cap, K), e_cap,
                     class_bounds_full(cn_cap, K),
                     getenv("S3GRL_FORCE_BM_HBM") ? 0 : (1 << 30),   // test hook: bit matrices in HBM
                     dm ? std::min(stash_slot + 2, 65535) : 0, dm ? dm_class_mask_for(g, cn_cap, K) : 0,
                     class_count, class_list, perm, x_cap, hb, csr_e, csrb, words_for(g->num_nodes),
                     getenv("S3GRL_CSR_LDS_PCT") ? std::max(100, atoi(getenv("S3GRL_CSR_LDS_PCT"))) : 100, tiny_max_n);
  S3GRL_HIP_TRY(hipGetLastError());
  return S3GRL_OK;
}

#endif  // !S3GRL_LINKS_PART

namespace {

struct LinkArgs {
  const s3grl_graph* g;
  const int64_t* links;
  const int32_t* class_list;
  int hops, plus, cn_cap, full_stats;
  WalkSets ws;
  const int32_t* p_nodes;
  const int64_t *node_off, *row_ptr, *job_off, *coef_off;
  const int32_t* mirror_of;
  int32_t* c_ids;
  float* c_coef;
  Job* jobs;
  float* job_z;
  int32_t* job_lim;
  int64_t* row_nodes;
  int32_t* lvl;
  int64_t *tot_edges, *tot_support, *tot_vol;
  char* scratch;
  int64_t scratch_stride;
  unsigned long long* dbg;
  HopSampling smp;
  const int32_t* stash;
  int slot;
  const int32_t* e_cap;
  uint32_t* bm_scratch;
  int64_t bm_stride_words;
  int bm_grid;
  int big_need;   // LDS need of the biggest link of the class whose matrix / columns sit in HBM
  const int32_t *old_of_new, *new_of_old;   // non-null: the graph is walked in its degree order
  int lo_id;                                // then: ids >= lo_id have at most two stored neighbours (else -1)
  int split_t, seg_shift;                   // lists longer than split_t are laid out in pieces of 2^seg_shift
  DirGraph dg;                              // arcs of a directed graph (null otherwise)
  int bm_ext_words;                         // HBM-scratch class: words of the bitmaps at the head of a slice (0: LDS)
  int gs_chunk;                             // ... and how many slices there are (the class runs in chunks)
  int64_t list_offset;                      // first entry of the class list a launch works on
  const int64_t* x_cap;                     // one-hop plans: bound of the edges outside the hub's cache (-1: no hub)
  uint32_t* hub_slices;                     // link_hub_kernel's overflow class: found-edge list + columns per workgroup
  int64_t hub_slice_words;
  int hub_slice_grid;
  const uint16_t* csr_cnt;                  // induced-CSR flavour (s3grl_csr.hip): members per list entry,
  const int32_t* csr_e;                     // ... and per link
  int sop2;                                 // S3GRL_MODE_SOP_RESTRICTED: global normalisation, nothing masked (link_kernel)
};

// One-hop full-reach classes (link_full_kernel).  Small classes run one wavefront per link (no
// cross-wave barriers to pay for 25-node subgraphs), the others four; the class whose bit matrix
// lives in HBM runs a persistent grid, one matrix slice per resident workgroup.
template <int T, int K, bool BMG>
s3grl_status launch_full_class(s3grl_context* ctx, const LinkArgs& a, int64_t L, int cls, int count,
                               hipStream_t stream, uint32_t* bm_scratch, int64_t bm_stride_words, int grid) {
  const ClassBounds fb = class_bounds_full(a.cn_cap, K);
  const size_t lds = (size_t)4 * full_fixed_words(a.cn_cap, K) +
                     (size_t)(cls == kFullBig ? a.big_need : fb.b[cls - kFullBase]);
  auto kern = link_full_kernel<T, K, BMG>;
  S3GRL_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(kern),
                                    hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  hipLaunchKernelGGL(kern, dim3((unsigned)grid), dim3(T), lds, stream, a.g->indptr, a.g->indices,
                     a.g->fwd_indptr, a.g->fwd_indices, a.links, a.class_list + (int64_t)cls * L, count,
                     a.plus, a.cn_cap, a.e_cap, a.node_off, a.row_ptr, a.job_off, a.coef_off, a.mirror_of,
                     a.c_ids, a.c_coef, a.jobs, a.job_z, a.job_lim, a.row_nodes, a.lvl,
                     reinterpret_cast<unsigned long long*>(a.tot_edges),
                     reinterpret_cast<unsigned long long*>(a.tot_support),
                     reinterpret_cast<unsigned long long*>(a.tot_vol), bm_scratch, bm_stride_words,
                     getenv("S3GRL_BIG_COLS_HBM") ? 0 : (int)lds,   // test hook: big class, columns in HBM
                     (BMG && a.dbg) ? a.dbg : nullptr, a.old_of_new, a.split_t, a.seg_shift);
  S3GRL_HIP_TRY(hipGetLastError());
  return S3GRL_OK;
}

template <int T, int K, int G, bool GS, bool HS, bool DM = false, bool DIRECTED = false>
s3grl_status launch_link_class_g(s3grl_context* ctx, const LinkArgs& a, int64_t L, int cls, int count,
                                 hipStream_t stream) {
  const int W = words_for(a.g->num_nodes);
  size_t lds;
  if (DM)
    lds = (size_t)4 * link_fixed_words_dm(a.g->num_nodes, a.cn_cap, K) +
          class_bounds_dm(a.g->num_nodes, a.cn_cap, K).b[cls - kSparseBase];
  else if (HS)
    lds = (size_t)4 * link_fixed_words_sparse(a.cn_cap, K) +
          class_bounds_sparse(a.cn_cap, K).b[cls - kSparseBase];
  else if (GS && a.bm_ext_words > 0)
    lds = (size_t)4 * link_fixed_words_sparse(a.cn_cap, K);
  else
    lds = (size_t)4 * link_fixed_words(a.g->num_nodes, a.cn_cap, K) +
          (GS ? 0 : (size_t)class_bounds(a.g->num_nodes, a.cn_cap, K).b[cls]);
  auto kern = link_kernel<T, K, G, GS, HS, DM, DIRECTED>;
  S3GRL_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(kern),
                                    hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  hipLaunchKernelGGL(kern, dim3((unsigned)count), dim3(T), lds, stream, a.g->indptr,
                     a.g->indices, W, a.links, a.class_list + (int64_t)cls * L + a.list_offset, a.hops, a.plus,
                     a.cn_cap, a.full_stats, a.g->max_degree > kHubArmDegree ? 1 : 0, a.ws,
                     a.p_nodes, a.node_off, a.row_ptr, a.job_off, a.coef_off,
                     a.mirror_of, a.c_ids, a.c_coef, a.jobs, a.job_z, a.job_lim, a.row_nodes, a.lvl,
                     reinterpret_cast<unsigned long long*>(a.tot_edges),
                     reinterpret_cast<unsigned long long*>(a.tot_support),
                     reinterpret_cast<unsigned long long*>(a.tot_vol), a.scratch, a.scratch_stride,
                     GS ? a.bm_ext_words : 0, a.dbg, a.smp, a.stash, a.slot, a.old_of_new, a.new_of_old, a.lo_id, a.split_t, a.seg_shift,
                     a.dg, a.sop2);
  S3GRL_HIP_TRY(hipGetLastError());
  return S3GRL_OK;
}

// lanes per CSR row: 4 for sparse graphs (PubMed/Cora: mean degree ~4), 8 otherwise
template <int T, int K>
s3grl_status launch_link_class(s3grl_context* ctx, const LinkArgs& a, int64_t L, int cls, int count,
                               hipStream_t stream) {
  const double mean_deg = (double)a.g->nnz / (double)std::max<int64_t>(a.g->num_nodes, 1);
  static const int force_g = getenv("S3GRL_LANES_PER_ROW") ? atoi(getenv("S3GRL_LANES_PER_ROW")) : 0;
  const int gsel = force_g ? force_g : (mean_deg <= 6.0 ? 4 : 8);
  if (a.dg.out_indptr) {   // directed plans: bitmap flavour, four lanes per row (two instantiations per sign_k)
    if (cls == kNumClasses) return launch_link_class_g<1024, K, 4, true, false, false, true>(ctx, a, L, cls, count, stream);
    return launch_link_class_g<256, K, 4, false, false, false, true>(ctx, a, L, cls, count, stream);
  }
  if (cls == kNumClasses) {   // HBM-scratch overflow class
    if (gsel <= 4) return launch_link_class_g<1024, K, 4, true, false>(ctx, a, L, cls, count, stream);
    return launch_link_class_g<1024, K, 8, true, false>(ctx, a, L, cls, count, stream);
  }
  if (cls >= kSparseBase && dm_mode_for(a.g)) {   // direct-map flavour
    if (gsel <= 4) return launch_link_class_g<T, K, 4, false, true, true>(ctx, a, L, cls, count, stream);
    return launch_link_class_g<T, K, 8, false, true, true>(ctx, a, L, cls, count, stream);
  }
  if (cls >= kSparseBase) {   // hash flavour
    if (gsel <= 4) return launch_link_class_g<256, K, 4, false, true>(ctx, a, L, cls, count, stream);
    return launch_link_class_g<256, K, 8, false, true>(ctx, a, L, cls, count, stream);
  }
  if (gsel <= 4) return launch_link_class_g<T, K, 4, false, false>(ctx, a, L, cls, count, stream);
  return launch_link_class_g<T, K, 8, false, false>(ctx, a, L, cls, count, stream);
}

// The launches of the LDS classes do not depend on each other: they go round-robin onto the
// context's stream and its side streams (forked and joined with events), so that the tail of one
// class overlaps the start of the next instead of draining the chip five times per plan.
static s3grl_status side_streams(s3grl_context* ctx) { return ensure_side_streams(ctx); }

template <int K>
s3grl_status launch_links_k(s3grl_context* ctx, const LinkArgs& a, int64_t L,
                            const int32_t* class_count_in) {
  static const bool serial = getenv("S3GRL_SERIAL_CLASSES") != nullptr;
  // diagnostic only (with S3GRL_DEBUG_STAMPS): launch one class list, the results are incomplete
  static const int only = getenv("S3GRL_ONLY_CLASS") ? atoi(getenv("S3GRL_ONLY_CLASS")) : -1;
  int32_t class_count_host[kNumListsAll];
  for (int c = 0; c < kNumListsAll; ++c)
    class_count_host[c] = (only < 0 || c == only || (c > kTinyList + 1 && c < kCsrBase)) ? class_count_in[c] : 0;
  int launches = 0;
  for (int c = 0; c <= kTinyList + 1; ++c) launches += class_count_host[c] > 0 && c != kNumClasses + 1;
  for (int c = kCsrBase; c < kNumListsAll; ++c) launches += class_count_host[c] > 0;
  const bool fork = !serial && launches > 1;
  // the side streams rejoin the context's stream on EVERY way out: a launch that fails half-way must not leave
  // kernels of this plan running beside whatever the caller queues next (its buffers go back to the arena)
  struct SideJoin {
    s3grl_context* ctx;
    bool armed = false;
    hipError_t join() {
      hipError_t first = hipSuccess;
      if (armed)
        for (int i = 0; i < s3grl_context::kSide; ++i) {
          hipError_t e = hipEventRecord(ctx->side_ev[i], ctx->side[i]);
          if (e == hipSuccess) e = hipStreamWaitEvent(ctx->stream, ctx->side_ev[i], 0);
          if (e != hipSuccess && first == hipSuccess) first = e;
        }
      armed = false;
      return first;
    }
    ~SideJoin() { (void)join(); }
  } side_join{ctx};
  if (fork) {
    S3GRL_TRY(side_streams(ctx));
    S3GRL_HIP_TRY(hipEventRecord(ctx->side_ev[s3grl_context::kSide], ctx->stream));
    side_join.armed = true;
    for (int i = 0; i < s3grl_context::kSide; ++i)
      S3GRL_HIP_TRY(hipStreamWaitEvent(ctx->side[i], ctx->side_ev[s3grl_context::kSide], 0));
  }
  int turn = 0;
  auto next_stream = [&]() -> hipStream_t {
    if (!fork) return ctx->stream;
    const int k = turn++ % (s3grl_context::kSide + 1);
    return k == 0 ? ctx->stream : ctx->side[k - 1];
  };
  // largest subgraphs first: they are the long poles of the tail
  if (class_count_host[kNumClasses] > 0) {
    if (a.bm_ext_words > 0) {   // bounded number of slices: the class runs in chunks, one after the other
      hipStream_t st = next_stream();
      for (int off = 0; off < class_count_host[kNumClasses]; off += a.gs_chunk) {
        LinkArgs b = a;
        b.list_offset = off;
        S3GRL_TRY((launch_link_class<1024, K>(ctx, b, L, kNumClasses,
                                               std::min(a.gs_chunk, class_count_host[kNumClasses] - off), st)));
      }
    } else {
      S3GRL_TRY((launch_link_class<1024, K>(ctx, a, L, kNumClasses, class_count_host[kNumClasses], next_stream())));
    }
  }
  for (int c = kNumListsAll - 1; c >= kCsrBase; --c) {   // full-reach links on their induced LDS CSR (s3grl_csr.hip)
    if (class_count_host[c] == 0) continue;
    CsrLinkArgs h{a.g->indptr, a.g->indices, words_for(a.g->num_nodes), a.hops,
                  a.g->balls.bits + (int64_t)(a.hops - 1) * a.g->balls.level_stride, a.links, a.plus, a.cn_cap,
                  a.csr_cnt, a.csr_e, a.node_off, a.row_ptr, a.job_off, a.coef_off, a.mirror_of, a.c_ids, a.c_coef,
                  a.jobs, a.job_z, a.job_lim, a.row_nodes, a.lvl, reinterpret_cast<unsigned long long*>(a.tot_edges),
                  reinterpret_cast<unsigned long long*>(a.tot_support),
                  reinterpret_cast<unsigned long long*>(a.tot_vol), a.stash, a.slot, a.old_of_new, a.split_t,
                  a.seg_shift, a.dbg};
    S3GRL_TRY(launch_csr_class(ctx, h, K, c - kCsrBase, a.class_list + (int64_t)c * L, class_count_host[c],
                               next_stream()));
  }
  if (class_count_host[kFullBig] > 0)
    S3GRL_TRY((launch_full_class<1024, K, true>(ctx, a, L, kFullBig, class_count_host[kFullBig], next_stream(),
                                                  a.bm_scratch, a.bm_stride_words, a.bm_grid)));
  for (int c = kHubBase + kHubClasses; c >= kHubBase; --c) {   // cached hub neighbourhoods (s3grl_hub.hip)
    if (class_count_host[c] == 0) continue;
    HubLinkArgs h{a.g->indptr, a.g->indices, a.g->hub, a.links, a.plus, a.cn_cap, a.x_cap, a.node_off, a.row_ptr,
                  a.job_off, a.coef_off, a.mirror_of, a.c_ids, a.c_coef, a.jobs, a.job_z, a.job_lim, a.row_nodes,
                  a.lvl, reinterpret_cast<unsigned long long*>(a.tot_edges),
                  reinterpret_cast<unsigned long long*>(a.tot_support),
                  reinterpret_cast<unsigned long long*>(a.tot_vol),
                  reinterpret_cast<unsigned long long*>(a.tot_vol) + 2 * (size_t)kStatShards * kStatStride,   // rows 4..8 of d_stats
                  reinterpret_cast<unsigned long long*>(a.tot_vol) + 3 * (size_t)kStatShards * kStatStride,
                  reinterpret_cast<unsigned long long*>(a.tot_vol) + 4 * (size_t)kStatShards * kStatStride,
                  reinterpret_cast<unsigned long long*>(a.tot_vol) + 5 * (size_t)kStatShards * kStatStride,
                  reinterpret_cast<unsigned long long*>(a.tot_vol) + 6 * (size_t)kStatShards * kStatStride,
                  a.e_cap, a.old_of_new, a.split_t, a.seg_shift, a.dbg,
                  a.hub_slices, a.hub_slice_words, a.hub_slice_grid};
    S3GRL_TRY(launch_hub_class(ctx, h, K, c - kHubBase, a.class_list + (int64_t)c * L, class_count_host[c],
                               next_stream()));
  }
  for (int w = 0; w < 2; ++w) {   // the smallest one-hop links, half a wavefront / a wavefront each (s3grl_hub.hip)
    if (class_count_host[kTinyList + w] == 0) continue;
    TinyLinkArgs t{a.g->indptr, a.g->indices, a.g->fwd_indptr, a.g->fwd_indices, a.links, a.node_off, a.row_ptr,
                   a.job_off, a.coef_off, a.mirror_of, a.c_ids, a.c_coef, a.jobs, a.job_z, a.job_lim, a.row_nodes,
                   a.lvl, reinterpret_cast<unsigned long long*>(a.tot_edges),
                   reinterpret_cast<unsigned long long*>(a.tot_support),
                   reinterpret_cast<unsigned long long*>(a.tot_vol), a.old_of_new};
    S3GRL_TRY(launch_tiny_class(ctx, t, K, w == 0 ? 32 : 64, a.class_list + (int64_t)(kTinyList + w) * L,
                                class_count_host[kTinyList + w], next_stream()));
  }
  for (int c = kFullBig - 1; c >= kFullBase; --c) {
    const int count = class_count_host[c];
    if (count == 0) continue;
    // threads per link by class: a wavefront for the smallest subgraphs; the classes whose LDS
    // leaves one or two workgroups per CU get 1024 / 512 threads (their probing trips are chains
    // of dependent loads: more rows per trip, more loads in flight)
    const int fc = c - kFullBase;
    // (class 2 at 128 threads since the links of at most 64 nodes left for link_tiny_kernel: 16.55 -> 16.3 ms on
    // config 5; 64: 16.75, 256: 16.55, 512: 18.2)
    int t = fc <= 1 ? 64 : (fc == 2 ? 128 : (fc == 3 ? 256 : (fc == 4 ? 512 : 1024)));
    {
      char name[32];   // tuning hook
      snprintf(name, sizeof(name), "S3GRL_TF_CLASS%d", fc);
      if (const char* e = getenv(name)) t = atoi(e);
    }
    if (t <= 64)
      S3GRL_TRY((launch_full_class<64, K, false>(ctx, a, L, c, count, next_stream(), nullptr, 0, count)));
    else if (t <= 128)
      S3GRL_TRY((launch_full_class<128, K, false>(ctx, a, L, c, count, next_stream(), nullptr, 0, count)));
    else if (t <= 256)
      S3GRL_TRY((launch_full_class<256, K, false>(ctx, a, L, c, count, next_stream(), nullptr, 0, count)));
    else if (t <= 512)
      S3GRL_TRY((launch_full_class<512, K, false>(ctx, a, L, c, count, next_stream(), nullptr, 0, count)));
    else
      S3GRL_TRY((launch_full_class<1024, K, false>(ctx, a, L, c, count, next_stream(), nullptr, 0, count)));
  }
  for (int c = kFullBase - 1; c >= kSparseBase; --c) {
    if (class_count_host[c] == 0) continue;
    if (dm_mode_for(a.g)) {
      const size_t lds = (size_t)4 * link_fixed_words_dm(a.g->num_nodes, a.cn_cap, K) +
                         class_bounds_dm(a.g->num_nodes, a.cn_cap, K).b[c - kSparseBase];
      const int t = threads_for_class(lds, c - kSparseBase);
      if (t <= 128) S3GRL_TRY((launch_link_class<128, K>(ctx, a, L, c, class_count_host[c], next_stream())));
      else if (t <= 256) S3GRL_TRY((launch_link_class<256, K>(ctx, a, L, c, class_count_host[c], next_stream())));
      else if (t <= 512) S3GRL_TRY((launch_link_class<512, K>(ctx, a, L, c, class_count_host[c], next_stream())));
      else S3GRL_TRY((launch_link_class<1024, K>(ctx, a, L, c, class_count_host[c], next_stream())));
    } else {
      S3GRL_TRY((launch_link_class<256, K>(ctx, a, L, c, class_count_host[c], next_stream())));
    }
  }
  for (int c = kNumClasses - 1; c >= 0; --c) {
    const int count = class_count_host[c];
    if (count == 0) continue;
    // Fewer workgroups fit a CU as the LDS per workgroup grows (bigger subgraph class, or a big
    // graph whose three N-bit bitmaps alone take tens of KB): give each more waves then.
    const size_t lds = (size_t)4 * link_fixed_words(a.g->num_nodes, a.cn_cap, K) +
                       class_bounds(a.g->num_nodes, a.cn_cap, K).b[c];
    const int t = threads_for_class(lds, c);
    if (t <= 128) S3GRL_TRY((launch_link_class<128, K>(ctx, a, L, c, count, next_stream())));
    else if (t <= 256) S3GRL_TRY((launch_link_class<256, K>(ctx, a, L, c, count, next_stream())));
    else if (t <= 512) S3GRL_TRY((launch_link_class<512, K>(ctx, a, L, c, count, next_stream())));
    else S3GRL_TRY((launch_link_class<1024, K>(ctx, a, L, c, count, next_stream())));
  }
  S3GRL_HIP_TRY(side_join.join());
  return S3GRL_OK;
}

}  // namespace

// The link kernels are instantiated per sign_k, thread count, lanes per row and flavour — most of
// this file's compile time.  They are spread over four translation units that compile in parallel:
// this file itself (sign_k 3, 4) and s3grl_links_{a,b,c}.hip, which include it with
// S3GRL_LINKS_PART defined and export one launcher each for their sign_k values.
#ifdef S3GRL_LINKS_PART
}  // namespace s3grl
extern "C" s3grl_status S3GRL_LINKS_PART(s3grl_context* ctx, const void* args, int64_t L,
                                          const int32_t* class_count_host, int K) {
  const s3grl::LinkArgs& a = *static_cast<const s3grl::LinkArgs*>(args);
  switch (K) {
    case S3GRL_LINKS_K0: return s3grl::launch_links_k<S3GRL_LINKS_K0>(ctx, a, L, class_count_host);
    case S3GRL_LINKS_K1: return s3grl::launch_links_k<S3GRL_LINKS_K1>(ctx, a, L, class_count_host);
    default: return S3GRL_ERR_INVALID_ARGUMENT;
  }
}
namespace s3grl {
#else
}  // namespace s3grl
extern "C" {
s3grl_status s3grl_links_part_a(s3grl_context*, const void*, int64_t, const int32_t*, int);   // sign_k 1, 2
s3grl_status s3grl_links_part_b(s3grl_context*, const void*, int64_t, const int32_t*, int);   // sign_k 5, 6
s3grl_status s3grl_links_part_c(s3grl_context*, const void*, int64_t, const int32_t*, int);   // sign_k 7, 8
}
namespace s3grl {

s3grl_status launch_links(s3grl_context* ctx, const s3grl_graph* g, const int64_t* links, int64_t L,
                          const int32_t* class_list, const int32_t* class_count_host, int hops,
                          int plus, int cn_cap, int full_stats, int K, WalkSets ws, const int32_t* p_nodes,
                          const int64_t* node_off, const int64_t* row_ptr, const int64_t* job_off,
                          const int64_t* coef_off, const int32_t* mirror_of, int32_t* c_ids,
                          float* c_coef, Job* jobs, float* job_z, int32_t* job_lim, int64_t* row_nodes,
                          int32_t* lvl,
                          int64_t* tot_edges, int64_t* tot_support, int64_t* tot_vol,
                          HopSampling smp, const int32_t* stash, int slot, const int32_t* e_cap,
                          int64_t max_nodes, const int32_t* old_of_new, const int32_t* new_of_old,
                          int split_t, int seg_shift, const int64_t* x_cap, const uint16_t* csr_cnt,
                          const int32_t* csr_e, int sop2) {
  if (L == 0) return S3GRL_OK;
  // links too large for LDS keep their lists in HBM scratch: one 256-byte aligned slice each
  Transient scratch_owner{ctx, {}};
  char* scratch = nullptr;
  int64_t scratch_stride = 0;
  int bm_ext_words = 0, gs_chunk = 0;
  if (class_count_host[kNumClasses] > 0) {
    scratch_stride = ((int64_t)class_count_host[kNumClasses + 1] + 255) / 256 * 256;
    int64_t slices = class_count_host[kNumClasses];
    if (4 * (int64_t)link_fixed_words(g->num_nodes, cn_cap, K) > 163840 || getenv("S3GRL_FORCE_EXT_BITMAPS")) {
      // the graph's bitmaps do not fit LDS: they go to the head of every slice, and the class runs in
      // chunks over at most 512 MiB of slices
      bm_ext_words = (3 * words_for(g->num_nodes) + 63) / 64 * 64;
      scratch_stride += (int64_t)bm_ext_words * 4;
      gs_chunk = (int)std::min<int64_t>(slices, std::max<int64_t>(64, ((int64_t)1 << 29) / scratch_stride));
      slices = gs_chunk;
    }
    void* q = nullptr;
    S3GRL_TRY(ctx->arena.alloc((size_t)scratch_stride * slices, &q));
    scratch_owner.ptrs.push_back(q);
    scratch = static_cast<char*>(q);
  }
  LinkArgs a{g, links, class_list, hops, plus, cn_cap, full_stats, ws, p_nodes, node_off,
             row_ptr,
             job_off, coef_off, mirror_of, c_ids, c_coef, jobs, job_z, job_lim, row_nodes, lvl, tot_edges,
             tot_support, tot_vol, scratch, scratch_stride,
             getenv("S3GRL_DEBUG_STAMPS") ? reinterpret_cast<unsigned long long*>(ctx->d_scalars + 16) : nullptr,
             smp, stash, slot, e_cap, nullptr, 0, 0, 0, old_of_new, new_of_old,
             (old_of_new && !getenv("S3GRL_NO_LEAF_WALK")) ? g->deg_le2_from : -1, split_t, seg_shift,
             DirGraph{g->out_indptr, g->out_indices, g->in_indptr, g->in_indices}, bm_ext_words, gs_chunk, 0,
             x_cap, nullptr, 0, 0, csr_cnt, csr_e, sop2};
  if (class_count_host[kHubBase + kHubClasses] > 0) {   // list of found edges (uint32) + columns (2 x uint16) per slice
    const int64_t xmax = ((int64_t)class_count_host[29] + 63) / 64 * 64;
    a.hub_slice_words = 2 * xmax;
    a.hub_slice_grid = (int)std::min<int64_t>(class_count_host[kHubBase + kHubClasses], 256);
    void* q = nullptr;
    S3GRL_TRY(ctx->arena.alloc((size_t)a.hub_slice_words * 4 * a.hub_slice_grid, &q));
    scratch_owner.ptrs.push_back(q);
    a.hub_slices = static_cast<uint32_t*>(q);
  }
  // the class whose bit matrix does not fit LDS: one slice per resident workgroup of a persistent grid
  if (class_count_host[kFullBig] > 0) {
    // slice = list of found edges (uint32, at most ecap / 2) + CSR columns (uint16 x ecap) of the
    // link with the largest bound
    a.bm_stride_words = ((int64_t)class_count_host[31] / 2 + 2 + (class_count_host[31] + 2) / 2 + 63) / 64 * 64;
    (void)max_nodes;
    // all of a CU's LDS for one 1024-thread workgroup: whatever the hash and the per-node arrays
    // leave holds the CSR columns whenever the exact entry count allows (see link_full_kernel)
    a.big_need = 163840 - 4 * full_fixed_words(cn_cap, K);
    (void)class_count_host[30];
    a.bm_grid = (int)std::min<int64_t>(class_count_host[kFullBig], 256);
    void* q = nullptr;
    S3GRL_TRY(ctx->arena.alloc((size_t)a.bm_stride_words * 4 * a.bm_grid, &q));
    scratch_owner.ptrs.push_back(q);
    a.bm_scratch = static_cast<uint32_t*>(q);
  }
  switch (K) {
    case 1:
    case 2: return s3grl_links_part_a(ctx, &a, L, class_count_host, K);
    case 3: return launch_links_k<3>(ctx, a, L, class_count_host);
    case 4: return launch_links_k<4>(ctx, a, L, class_count_host);
    case 5:
    case 6: return s3grl_links_part_b(ctx, &a, L, class_count_host, K);
    case 7:
    case 8: return s3grl_links_part_c(ctx, &a, L, class_count_host, K);
    default:
      set_last_error("sign_k must be in 1..8");
      return S3GRL_ERR_INVALID_ARGUMENT;
  }
}

s3grl_status launch_dists(s3grl_context* ctx, const int64_t* node_off, const int32_t* lvl, int64_t L,
                          int8_t* dists) {
  if (L == 0) return S3GRL_OK;
  hipLaunchKernelGGL(dists_kernel, dim3((unsigned)L), dim3(256), 0, ctx->stream, node_off, lvl, L,
                     dists);
  S3GRL_HIP_TRY(hipGetLastError());
  return S3GRL_OK;
}
#endif  // S3GRL_LINKS_PART

}  // namespace s3grl

#ifndef S3GRL_TOUCH_UNIT
#define S3GRL_TOUCH_UNIT structure
#endif
S3GRL_DEFINE_TOUCH(S3GRL_TOUCH_UNIT)
