// Device-side helpers shared by the structure and SoP kernels (gfx950, wave64).
#pragma once

#include <hip/hip_runtime.h>

#include <cstdint>

#include "s3grl_internal.hpp"

namespace s3grl {
namespace {

__device__ __forceinline__ int lane_id() { return threadIdx.x & 63; }
__device__ __forceinline__ int wave_id() { return threadIdx.x >> 6; }

// Exclusive scan of one int per thread over a T-thread block; `sh` holds >= T/64 ints.
template <int T>
__device__ __forceinline__ int block_excl_scan(int v, int* sh, int& total) {
  const int lane = lane_id(), wid = wave_id();
  int inc = v;
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) {
    int t = __shfl_up(inc, o);
    if (lane >= o) inc += t;
  }
  if (lane == 63) sh[wid] = inc;
  __syncthreads();
  int woff = 0, tot = 0;
#pragma unroll
  for (int i = 0; i < T / 64; ++i) {
    int s = sh[i];
    if (i < wid) woff += s;
    tot += s;
  }
  __syncthreads();
  total = tot;
  return woff + inc - v;
}

template <int T>
__device__ __forceinline__ int block_sum(int v, int* sh) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
  if (lane_id() == 0) sh[wave_id()] = v;
  __syncthreads();
  int tot = 0;
#pragma unroll
  for (int i = 0; i < T / 64; ++i) tot += sh[i];
  __syncthreads();
  return tot;
}

__device__ __forceinline__ bool test_bit(const uint32_t* bm, int v) {
  return (bm[v >> 5] >> (v & 31)) & 1u;
}

__device__ __forceinline__ int rank_of(const uint32_t* vis, const uint32_t* wpre, int v) {
  return (int)wpre[v >> 5] + __popc(vis[v >> 5] & ((1u << (v & 31)) - 1u));
}

// membership in an ascending int list (global CSR row)
__device__ __forceinline__ bool sorted_contains(const int32_t* a, int n, int x) {
  int lo = 0, hi = n;
  while (lo < hi) {
    const int mid = (lo + hi) >> 1;
    if (a[mid] < x) lo = mid + 1; else hi = mid;
  }
  return lo < n && a[lo] == x;
}

// Walks the CSR rows of list[r0..r1) with G lanes per row.  The passes built on this are bound
// by dependent-load latency (list -> indptr -> indices -> LDS), not by lanes or bandwidth, so
// UN row groups are in flight per wave: their id, row-bounds and first-neighbour loads are
// issued back to back before anything is consumed.  visit(acc, v, u, valid) is called for every
// stored neighbour u of row v (and with valid = false for padding slots); commit(acc, t, v) once per row, by one lane, with acc summed over the
// row (fixed reduction trees: bit-reproducible).
// Hub rows (more than kHubFactor·G stored neighbours: a power-law hub would pin its G lanes for
// hundreds of trips while the rest of the workgroup waits at the next barrier) are deferred to
// `hub` = {count, row positions…} in LDS and then walked one at a time by the WHOLE workgroup.
// Must be called by every thread of the workgroup (contains barriers).
struct RowAcc {
  float x, y;
  int n;
};

constexpr int kHubFactor = 64;
constexpr int kHubCap = 255;     // deferred hub rows per call; further ones are walked in place
constexpr int kHubWords = kHubCap + 1 + 64;   // hub list + block-reduction scratch

template <int T, int G, int UN, typename Visit, typename Commit>
__device__ __forceinline__ void walk_rows(int r0, int r1, const int32_t* list,
                                          const int32_t* __restrict__ indptr,
                                          const int32_t* __restrict__ indices, int* hub,
                                          Visit visit, Commit commit) {
  const int tid = threadIdx.x;
  const int g = tid & (G - 1);
  constexpr int RPI = T / G;  // rows per wave-iteration slice
  for (int base = r0; base < r1; base += UN * RPI) {
    // all loads of a stage are unconditional (clamped to a valid location, results selected
    // afterwards): predicated loads make hipcc wait after each one
    int t[UN], v[UN], c0[UN], e1[UN], first[UN];
    int2 be[UN];
#pragma unroll
    for (int u = 0; u < UN; ++u) {
      t[u] = base + u * RPI + tid / G;
      v[u] = list[min(t[u], r1 - 1)];
    }
#pragma unroll
    for (int u = 0; u < UN; ++u) {
      be[u].x = indptr[v[u]];
      be[u].y = indptr[v[u] + 1];
    }
#pragma unroll
    for (int u = 0; u < UN; ++u) {
      bool ok = t[u] < r1;
      if (hub && ok && be[u].y - be[u].x > kHubFactor * G) {   // uniform over the row's G lanes
        int slot = kHubCap;
        if (g == 0) slot = atomicAdd(&hub[0], 1);
        slot = __shfl(slot, (tid & 63) & ~(G - 1));
        if (slot < kHubCap) {
          if (g == 0) hub[1 + slot] = t[u];
          ok = false;
        }
      }
      c0[u] = ok ? be[u].x + g : 0;
      e1[u] = ok ? be[u].y : 0;
      if (!ok) v[u] = -1;
    }
    int second[UN];
#pragma unroll
    for (int u = 0; u < UN; ++u) {
      first[u] = indices[c0[u] < e1[u] ? c0[u] : 0];
      second[u] = indices[c0[u] + G < e1[u] ? c0[u] + G : 0];
    }
#pragma unroll
    for (int u = 0; u < UN; ++u) {
      if (c0[u] >= e1[u]) first[u] = -1;
      if (c0[u] + G >= e1[u]) second[u] = -1;
    }
    // visits are branch-free (a `valid` flag instead of a skipped call) and the first two
    // neighbours of all UN rows are visited in one straight-line block, so that their LDS reads
    // interleave instead of forming one long dependent chain per row
    RowAcc acc[UN];
#pragma unroll
    for (int u = 0; u < UN; ++u) {
      acc[u] = RowAcc{0.f, 0.f, 0};
      const int vv = max(v[u], 0);
      visit(acc[u], vv, max(first[u], 0), first[u] >= 0);
      visit(acc[u], vv, max(second[u], 0), second[u] >= 0);
    }
#pragma unroll
    for (int u = 0; u < UN; ++u) {
      // longer rows: two neighbours per lane in flight per trip
      for (int c = c0[u] + 2 * G; c < e1[u]; c += 2 * G) {
        const bool vb = c + G < e1[u];
        const int ua = indices[c];
        const int ub = indices[vb ? c + G : c];
        visit(acc[u], v[u], ua, true);
        visit(acc[u], v[u], ub, vb);
      }
    }
#pragma unroll
    for (int u = 0; u < UN; ++u) {
      if (v[u] >= 0) {
#pragma unroll
        for (int o = G / 2; o > 0; o >>= 1) {
          acc[u].x += __shfl_xor(acc[u].x, o);
          acc[u].y += __shfl_xor(acc[u].y, o);
          acc[u].n += __shfl_xor(acc[u].n, o);
        }
        if (g == 0) commit(acc[u], t[u], v[u]);
      }
    }
  }
  // deferred hub rows: the whole workgroup walks one row at a time.  hub == nullptr (uniform:
  // the graph's maximum degree is below the hub threshold) skips the phase and its barriers.
  if (!hub) return;
  __syncthreads();
  const int nh = min(hub[0], kHubCap);
  float* red = reinterpret_cast<float*>(hub + 1 + kHubCap);   // [T/64][3] <= 48 words
  for (int h = 0; h < nh; ++h) {
    const int tt = hub[1 + h];
    const int v = list[tt];
    const int e1 = indptr[v + 1];
    RowAcc acc{0.f, 0.f, 0};
    for (int c = indptr[v] + tid; c < e1; c += T) visit(acc, v, indices[c], true);
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
      acc.x += __shfl_xor(acc.x, o);
      acc.y += __shfl_xor(acc.y, o);
      acc.n += __shfl_xor(acc.n, o);
    }
    if ((tid & 63) == 0) {
      red[(tid >> 6) * 3 + 0] = acc.x;
      red[(tid >> 6) * 3 + 1] = acc.y;
      red[(tid >> 6) * 3 + 2] = __int_as_float(acc.n);
    }
    __syncthreads();
    if (tid == 0) {
      RowAcc tot{0.f, 0.f, 0};
      for (int w = 0; w < T / 64; ++w) {   // fixed order
        tot.x += red[w * 3 + 0];
        tot.y += red[w * 3 + 1];
        tot.n += __float_as_int(red[w * 3 + 2]);
      }
      commit(tot, tt, v);
    }
    __syncthreads();
  }
  if (tid == 0) hub[0] = 0;
  __syncthreads();
}

// Word-level popcount prefix of a bitmap (local id = rank): thread-contiguous runs, one block scan.
template <int T>
__device__ __forceinline__ void rank_prefix(const uint32_t* bm, uint32_t* wpre, int W, int* sh) {
  const int tid = threadIdx.x;
  const int C = (W + T - 1) / T;
  const int w0 = min(tid * C, W), w1 = min(w0 + C, W);
  int mine = 0;
  for (int t = w0; t < w1; ++t) mine += __popc(bm[t]);
  int total;
  int run = block_excl_scan<T>(mine, sh, total);
  for (int t = w0; t < w1; ++t) {
    wpre[t] = run;
    run += __popc(bm[t]);
  }
}

// Level-synchronous BFS from {src,dst} to depth `hops` on the UNMASKED graph (reference
// utils.py:53-74: `fringe = neighbors(fringe, A) - visited`, early break on an empty fringe).
// The frontier is a segment of `list`, G lanes per frontier node; every finished level is
// appended to `list` in ascending id order, so the list is hop-major and deterministic.
// On return: vis = S as a bitmap, list[0..n) = S, lvl_end[d] = nodes within d hops for
// d < nlev, `nxt` is all zero again.  Returns n = |S|.  Needs hops <= kMaxLevels - 2.
template <int T, int G>
__device__ __forceinline__ int bfs_list(const int32_t* __restrict__ indptr,
                                        const int32_t* __restrict__ indices, int W, int src,
                                        int dst, int hops, uint32_t* vis, uint32_t* nxt,
                                        int32_t* list, int* lvl_end, int* sh, int* hub,
                                        int& nlev_out) {
  const int tid = threadIdx.x;
  for (int t = tid; t < W; t += T) {
    vis[t] = 0;
    nxt[t] = 0;
  }
  __syncthreads();
  if (tid == 0) {
    atomicOr(&vis[src >> 5], 1u << (src & 31));
    atomicOr(&vis[dst >> 5], 1u << (dst & 31));
    list[0] = min(src, dst);
    list[1] = max(src, dst);
    lvl_end[0] = 2;
    if (hub) hub[0] = 0;
  }
  __syncthreads();
  int n = 2, nlev = 1;  // levels 0..nlev-1 are complete
  for (int d = 1; d <= hops; ++d) {
    const int f0 = d >= 2 ? lvl_end[d - 2] : 0, f1 = n;
    walk_rows<T, G, 2>(
        f0, f1, list, indptr, indices, hub,
        [&](RowAcc&, int, int u, bool valid) {
          if (valid) {
            const uint32_t m = 1u << (u & 31);
            const uint32_t old = atomicOr(&vis[u >> 5], m);
            if (!(old & m)) atomicOr(&nxt[u >> 5], m);
          }
        },
        [](RowAcc&, int, int) {});
    __syncthreads();
    // append the new level in ascending id order: every thread owns a contiguous run of
    // bitmap words, one block scan over the per-thread popcounts (2 barriers per level)
    const int C = (W + T - 1) / T;
    const int w0 = min(tid * C, W), w1 = min(w0 + C, W);
    int mine = 0;
    for (int t = w0; t < w1; ++t) mine += __popc(nxt[t]);
    int added;
    int pos = n + block_excl_scan<T>(mine, sh, added);
    for (int t = w0; t < w1; ++t) {
      uint32_t w = nxt[t];
      nxt[t] = 0;
      while (w) {
        const int b = __ffs(w) - 1;
        w &= w - 1;
        list[pos++] = t * 32 + b;
      }
    }
    if (added == 0) break;  // uniform: `added` is a block-wide total
    n += added;
    if (tid == 0) lvl_end[d] = n;
    nlev = d + 1;
    __syncthreads();
  }
  __syncthreads();
  nlev_out = nlev;
  return n;
}

}  // namespace
}  // namespace s3grl
