// Device-side helpers shared by the structure and SoP kernels (gfx950, wave64).
#pragma once

#include <hip/hip_runtime.h>

#include <cstdint>

#include "s3grl_internal.hpp"

namespace s3grl {
namespace {

// Plan statistics are summed with one atomic per link: on ONE address a million of them serialise at
// the L2 (that alone was half of the sizing pass), so every total is kept in kStatShards counters
// on lines of their own, picked by the workgroup index, and added up on the host.
__device__ __forceinline__ unsigned long long* stat_slot(unsigned long long* base) {
  return base + (blockIdx.x & (kStatShards - 1)) * kStatStride;
}

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

// PoS Plus row selection, reference tuned_SIGN.py:233 evaluated on the MASKED sub-CSR whose
// `.indices` still hold the explicit zeros of the masking:
//   N'(0) = (N_G(src) ∩ S) \ {dst} ∪ {dst-as-explicit-zero};  N'(1) likewise;  CN = N'(0) ∩ N'(1).
// For x ∉ {src,dst}: x ∈ N(src) ∩ N(dst) ∩ S.  src itself is selected iff src has a self-loop,
// dst iff dst has one.  One wave walks row(src) (ascending) and emits global ids in ascending
// order into out[] (may be null: count only).  Returns |CN|.
template <typename Member>
__device__ __forceinline__ int common_neighbours(const int32_t* __restrict__ indptr,
                                                 const int32_t* __restrict__ indices,
                                                 Member in_s, int src, int dst, int32_t* out) {
  const int lane = lane_id();
  const int32_t* row_s = indices + indptr[src];
  const int32_t* row_d = indices + indptr[dst];
  const int cs = indptr[src + 1] - indptr[src], cd = indptr[dst + 1] - indptr[dst];
  const bool loop_d = sorted_contains(row_d, cd, dst);
  int total = 0, lt_dst = 0;
  for (int c0 = 0; c0 < cs; c0 += 64) {
    const int c = c0 + lane;
    int x = -1;
    bool sel = false;
    if (c < cs) {
      x = row_s[c];
      sel = x != dst && in_s(x) && (x == src || sorted_contains(row_d, cd, x));
    }
    const unsigned long long bal = __ballot(sel);
    const unsigned long long below = __ballot(sel && x < dst);
    if (sel && out)
      out[total + __popcll(bal & ((1ull << lane) - 1ull)) + ((loop_d && dst < x) ? 1 : 0)] = x;
    total += __popcll(bal);
    lt_dst += __popcll(below);
  }
  if (loop_d) {
    if (lane == 0 && out) out[lt_dst] = dst;
    total += 1;
  }
  return total;
}

// Walks the CSR rows of list[r0..r1) with G lanes per row.  The passes built on this are bound
// by dependent-load latency (list -> indptr -> indices -> LDS), not by lanes or bandwidth, so
// UN row groups are in flight per wave: their id, row-bounds and first-neighbour loads are
// issued back to back before anything is consumed.  visit(acc, v, u, valid) is called for every
// stored neighbour u of row v (and with valid = false for padding slots); commit(acc, t, v) once per row, by one lane, with acc summed over the
// row (fixed reduction trees: bit-reproducible).
// Hub rows (more than kHubFactor·G stored neighbours: a long row would pin its G lanes for dozens
// of trips while the other rows of the wavefront idle) are deferred to `hub` = {count, row
// positions…} in LDS and then walked by one whole wavefront each.
// Must be called by every thread of the workgroup (contains barriers).
struct RowAcc {
  float x, y;
  int n;
  int row;   // list position of the row being walked (set by walk_rows before the visits)
};

constexpr int kHubFactor = 8;
// the hub path (one extra barrier per walk) is armed only for graphs that have real hubs:
// PubMed (max degree 171) runs 5 % faster without it, a power-law graph 20 % faster with it
constexpr int kHubArmDegree = 256;
// Deferred hub rows are marked in a bitmap over the list positions of the walk — EVERY hub row is summed by
// a wavefront, whatever order the lanes reach it in (a list of the first 255 to arrive made the lane
// assignment of the 256th — and its sum's last bit — depend on the order of an atomic: 36 links of the
// collab-scale list).  kHubWords words cover kHubWords * 32 rows; longer walks go chunk by chunk.
constexpr int kHubWords = 256;
constexpr int kHubChunk = kHubWords * 32;

// all-zero before the first walk of a kernel (walk_rows leaves it that way); called by every thread
template <int T>
__device__ __forceinline__ void hub_rows_clear(int* hub) {
  if (hub)
    for (int t = threadIdx.x; t < kHubWords; t += T) hub[t] = 0;
}

template <int T, int G, int UN, typename Visit, typename Commit>
__device__ __forceinline__ void walk_rows(int r0, int r1, const int32_t* list,
                                          const int32_t* __restrict__ indptr,
                                          const int32_t* __restrict__ indices, int* hub,
                                          Visit visit, Commit commit) {
  const int tid = threadIdx.x;
  const int g = tid & (G - 1);
  constexpr int RPI = T / G;  // rows per wave-iteration slice
  const int r_begin = r0, r_end = r1;
  for (int chunk0 = r_begin; chunk0 < r_end; chunk0 += hub ? kHubChunk : max(r_end - r_begin, 1)) {
  r0 = chunk0;
  r1 = hub ? min(chunk0 + kHubChunk, r_end) : r_end;
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
      // unsigned indices: a 32-bit offset on a uniform base (no sign extension, no 64-bit add)
      be[u].x = indptr[(uint32_t)v[u]];
      be[u].y = indptr[(uint32_t)v[u] + 1u];
    }
#pragma unroll
    for (int u = 0; u < UN; ++u) {
      bool ok = t[u] < r1;
      if (hub && ok && be[u].y - be[u].x > kHubFactor * G) {   // uniform over the row's G lanes
        if (g == 0) atomicOr(reinterpret_cast<uint32_t*>(hub) + ((t[u] - r0) >> 5), 1u << ((t[u] - r0) & 31));
        ok = false;
      }
      c0[u] = ok ? be[u].x + g : 0;
      e1[u] = ok ? be[u].y : 0;
      if (!ok) v[u] = -1;
    }
    int second[UN];
#pragma unroll
    for (int u = 0; u < UN; ++u) {
      first[u] = indices[(uint32_t)(c0[u] < e1[u] ? c0[u] : 0)];
      second[u] = indices[(uint32_t)(c0[u] + G < e1[u] ? c0[u] + G : 0)];
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
      acc[u] = RowAcc{0.f, 0.f, 0, t[u]};
      const int vv = max(v[u], 0);
      visit(acc[u], vv, max(first[u], 0), first[u] >= 0);
      visit(acc[u], vv, max(second[u], 0), second[u] >= 0);
    }
#pragma unroll
    for (int u = 0; u < UN; ++u) {
      // longer rows: two neighbours per lane and step, and the pair of the NEXT step already in
      // flight while this one is visited — a step is a dependent global load, and the longest row of
      // the wavefront sets the length of the trip (same visits in the same order as a plain loop)
      int c = c0[u] + 2 * G;
      if (c < e1[u]) {
        int ua = indices[(uint32_t)c];
        int ub = indices[(uint32_t)(c + G < e1[u] ? c + G : c)];
        for (;;) {
          const int cn = c + 2 * G;
          const bool more = cn < e1[u];
          uint32_t an = (uint32_t)(more ? cn : c), bn = (uint32_t)(more && cn + G < e1[u] ? cn + G : c);
          // opaque to the optimiser: it would otherwise prove (read-only memory) that loading the next
          // pair here equals loading it at the top of the next step, and undo the pipelining
          asm volatile("" : "+v"(an), "+v"(bn));
          const int na = indices[an];
          const int nb = indices[bn];
          visit(acc[u], v[u], ua, true);
          visit(acc[u], v[u], ub, c + G < e1[u]);
          if (!more) break;
          c = cn;
          ua = na;
          ub = nb;
        }
      }
    }
    // The xor tree leaves the row total in EVERY lane of the group, so lane u of a group can
    // commit row u: the UN commits of a trip are one pass with UN of G lanes active instead of UN
    // passes with one (the link kernel is VALU-issue-bound and a commit is ~25 instructions).
#pragma unroll
    for (int u = 0; u < UN; ++u) {
#pragma unroll
      for (int o = G / 2; o > 0; o >>= 1) {
        acc[u].x += __shfl_xor(acc[u].x, o);
        acc[u].y += __shfl_xor(acc[u].y, o);
        acc[u].n += __shfl_xor(acc[u].n, o);
      }
    }
    if constexpr (UN <= G) {
      RowAcc c = acc[0];
      int ct = t[0], cv = v[0];
#pragma unroll
      for (int u = 1; u < UN; ++u) {
        if (g == u) {
          c = acc[u];
          ct = t[u];
          cv = v[u];
        }
      }
      if (g < UN && cv >= 0) commit(c, ct, cv);
    } else {
#pragma unroll
      for (int u = 0; u < UN; ++u)
        if (g == 0 && v[u] >= 0) commit(acc[u], t[u], v[u]);
    }
  }
  // deferred hub rows: one WAVEFRONT per row (64 lanes stride it), several rows in parallel per
  // workgroup, xor-tree reduction: no barrier per row.  hub == nullptr (uniform: the graph's
  // maximum degree is below the hub threshold) skips the phase and its barriers.
  if (!hub) return;
  __syncthreads();
  const int lane = tid & 63;
  const int hub_words = (r1 - r0 + 31) >> 5;
  for (int hw = tid >> 6; hw < hub_words; hw += T / 64) {
   uint32_t hub_bits = reinterpret_cast<const uint32_t*>(hub)[hw];   // (uniform over the wavefront)
   while (hub_bits) {
    const int tt = r0 + hw * 32 + __ffs(hub_bits) - 1;
    hub_bits &= hub_bits - 1;
    const int v = list[tt];
    const int e1 = indptr[v + 1];
    RowAcc acc{0.f, 0.f, 0, tt};
    for (int c = indptr[v] + lane; c < e1; c += 128) {   // two neighbours per lane in flight
      const bool vb = c + 64 < e1;
      const int ua = indices[c];
      const int ub = indices[vb ? c + 64 : c];
      visit(acc, v, ua, true);
      visit(acc, v, ub, vb);
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
      acc.x += __shfl_xor(acc.x, o);
      acc.y += __shfl_xor(acc.y, o);
      acc.n += __shfl_xor(acc.n, o);
    }
    if (lane == 0) commit(acc, tt, v);
   }
  }
  __syncthreads();
  for (int hw = tid; hw < hub_words; hw += T) hub[hw] = 0;
  __syncthreads();
  }
}

// ---------------------------------------------------------------------------------------
// ScaLed: f(u) for every node the walks of src and dst visited (WalkSets, s3grl_internal.hpp) —
// the engine's own walks, or the caller's cached node sets (reference utils.py:94-104: the cache of
// src and of dst by node, or one set per link).  Called by all T threads of the workgroup.
template <typename F>
__device__ __forceinline__ void for_each_walk_node(const WalkSets& w, int src, int dst, int link, int tid,
                                                   int T, F f) {
  if (w.raw) {
    for (int i = tid; i < 2 * w.len; i += T)
      f(w.raw[(int64_t)(i < w.len ? src : dst) * w.len + (i < w.len ? i : i - w.len)]);
  } else {
    const int64_t ka = w.per_link ? link : src;
    const int64_t a0 = w.ptr[ka], na = w.ptr[ka + 1] - a0;
    const int64_t b0 = w.per_link ? 0 : w.ptr[dst], nb = w.per_link ? 0 : w.ptr[dst + 1] - b0;
    for (int64_t i = tid; i < na + nb; i += T) {
      const int u = w.nodes[i < na ? a0 + i : b0 + (i - na)];
      if ((unsigned)u < (unsigned)w.num_nodes) f(u);   // (validated at plan creation; never index LDS out of range)
    }
  }
}

// ---------------------------------------------------------------------------------------
// Per-hop sampling (reference utils.py:66-70): of the m nodes a hop discovers, keep
// k = min(int(ratio * m), max_nodes_per_hop) drawn uniformly without replacement; the others stay
// "visited" (they are never rediscovered) but are not part of the subgraph.  The reference draws
// with Python's random.sample; here every node gets a key from a counter-based generator keyed by
// (seed, the link's endpoints as an unordered pair, node) and the k smallest keys are kept: a
// uniform k-subset as well, the same one in every kernel that re-derives the subgraph and for
// both directions of a link.  (A node is discovered in at most one hop of a link, so the hop
// needs no place in the key; ratio and cap applied one after the other keep the min(k1, k2)
// smallest keys, which is what one draw of that size keeps.)
__device__ __forceinline__ bool sampling_on(const HopSampling& s) {
  return (s.ratio > 0.0 && s.ratio < 1.0) || s.max_nodes > 0;
}

__device__ __forceinline__ int hop_keep(const HopSampling& s, int added) {
  int k = added;
  if (s.ratio > 0.0 && s.ratio < 1.0) k = (int)(s.ratio * (double)added);   // Python: int(ratio * len)
  if (s.max_nodes > 0 && s.max_nodes < k) k = s.max_nodes;
  return k;
}

// distinct for distinct nodes: the node id sits in the low half
__device__ __forceinline__ uint64_t hop_sample_key(uint32_t seed, int a, int b, int u) {
  uint64_t x = (uint64_t)seed << 32;
  x ^= (uint64_t)(uint32_t)a * 0x9E3779B97F4A7C15ull;
  x ^= (uint64_t)(uint32_t)b * 0xC2B2AE3D27D4EB4Full;
  x += (uint64_t)(uint32_t)u * 0x165667B19E3779F9ull;
  x ^= x >> 33;
  x *= 0xff51afd7ed558ccdull;
  x ^= x >> 33;
  x *= 0xc4ceb9fe1a85ec53ull;
  x ^= x >> 33;
  return (x & 0xffffffff00000000ull) | (uint32_t)u;
}

// Clears all but the `keep` smallest-keyed bits of the level bitmap `nxt` (0 < keep < popcount).
// A bisection over the 64-bit key space finds a threshold with exactly `keep` keys at or below
// it (keys are distinct, so one exists); ~log2(level size) + a few rounds of one block reduction.
template <int T>
__device__ __forceinline__ void sample_level(uint32_t* nxt, int W, int keep, uint32_t seed, int a,
                                             int b, int* sh) {
  const int tid = threadIdx.x;
  const int C = (W + T - 1) / T;
  const int w0 = min(tid * C, W), w1 = min(w0 + C, W);
  uint64_t lo = 0, hi = ~0ull, tau = 0;
  for (int it = 0; it < 70; ++it) {           // <= 64 halvings; the bound is a safety net
    const uint64_t mid = lo + ((hi - lo) >> 1);
    int c = 0;
    for (int t = w0; t < w1; ++t) {
      uint32_t w = nxt[t];
      while (w) {
        const int bit = __ffs(w) - 1;
        w &= w - 1;
        c += hop_sample_key(seed, a, b, t * 32 + bit) <= mid ? 1 : 0;
      }
    }
    c = block_sum<T>(c, sh);
    if (c == keep) {
      tau = mid;
      break;
    }
    if (c < keep) lo = mid + 1; else hi = mid - 1;
  }
  for (int t = w0; t < w1; ++t) {
    uint32_t w = nxt[t], kept = 0;
    while (w) {
      const int bit = __ffs(w) - 1;
      w &= w - 1;
      if (hop_sample_key(seed, a, b, t * 32 + bit) <= tau) kept |= 1u << bit;
    }
    nxt[t] = kept;
  }
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
                                        int32_t* list, int cap, int* lvl_end, int* sh, int* hub,
                                        int& nlev_out, const WalkSets ws = WalkSets{}, int link = 0,
                                        HopSampling smp = HopSampling{1.0, 0, 0}) {
  const bool walks = walks_on(ws);
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
  }
  hub_rows_clear<T>(hub);
  __syncthreads();
  int n = 2, nlev = 1;  // levels 0..nlev-1 are complete
  if (walks) hops = 1;  // ScaLed: "level 1" = what the cached random walks of src and dst visited
  for (int d = 1; d <= hops; ++d) {
    const int f0 = d >= 2 ? lvl_end[d - 2] : 0, f1 = n;
    if (walks) {
      for_each_walk_node(ws, src, dst, link, tid, T, [&](int u) {
        const uint32_t m = 1u << (u & 31);
        const uint32_t old = atomicOr(&vis[u >> 5], m);
        if (!(old & m)) atomicOr(&nxt[u >> 5], m);
      });
    } else {
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
    }
    __syncthreads();
    // append the new level in ascending id order: every thread owns a contiguous run of
    // bitmap words, one block scan over the per-thread popcounts (2 barriers per level)
    const int C = (W + T - 1) / T;
    const int w0 = min(tid * C, W), w1 = min(w0 + C, W);
    int mine = 0;
    for (int t = w0; t < w1; ++t) mine += __popc(nxt[t]);
    int added;
    int pos = n + block_excl_scan<T>(mine, sh, added);
    if (!walks && added > 0 && sampling_on(smp)) {   // utils.py:66-70 (uniform: block-wide values)
      const int keep = hop_keep(smp, added);
      if (keep < added) {
        if (keep > 0) {
          sample_level<T>(nxt, W, keep, smp.seed, min(src, dst), max(src, dst), sh);
          mine = 0;
          for (int t = w0; t < w1; ++t) mine += __popc(nxt[t]);
          pos = n + block_excl_scan<T>(mine, sh, added);
        } else {
          added = 0;                                   // utils.py:71-72: an empty sample ends the walk
          for (int t = w0; t < w1; ++t) nxt[t] = 0;
        }
      }
    }
    for (int t = w0; t < w1; ++t) {
      uint32_t w = nxt[t];
      nxt[t] = 0;
      while (w) {
        const int b = __ffs(w) - 1;
        w &= w - 1;
        if (pos < cap) list[pos] = t * 32 + b;   // cap = what the sizing pass counted: never exceeded
        ++pos;                                   // unless the two passes disagree (then: no fault)
      }
    }
    if (added == 0) break;  // uniform: `added` is a block-wide total
    n = min(n + added, cap);
    if (tid == 0) lvl_end[d] = n;
    nlev = d + 1;
    __syncthreads();
  }
  __syncthreads();
  if (!walks && sampling_on(smp)) {
    // vis also holds the discovered-but-dropped nodes: rebuild it as the membership bitmap of S
    for (int t = tid; t < W; t += T) vis[t] = 0;
    __syncthreads();
    for (int i = tid; i < n; i += T) atomicOr(&vis[list[i] >> 5], 1u << (list[i] & 31));
    __syncthreads();
  }
  nlev_out = nlev;
  return n;
}

// ---------------------------------------------------------------------------------------
// Hash-set flavour of the visited set, for graphs whose N-bit bitmaps would eat the LDS (a
// 235 000-node graph needs 88 KB for three bitmaps, leaving one workgroup per CU for subgraphs
// of ~100 nodes).  Open addressing over C = pow2 >= 2n slots: keys = global node ids, vals =
// position of the node in the hop-major list (its local id).
constexpr int kSparseLevelMax = 1024;   // nodes a BFS level may add in the hash flavour

__device__ __forceinline__ uint32_t hs_home(int v, uint32_t mask) {
  return ((uint32_t)v * 2654435761u >> 7) & mask;
}

__device__ __forceinline__ bool hs_insert(int32_t* keys, uint32_t mask, int v) {
  uint32_t s = hs_home(v, mask);
  for (;;) {
    const int old = atomicCAS(&keys[s], -1, v);
    if (old == -1) return true;
    if (old == v) return false;
    s = (s + 1) & mask;
  }
}

// slot of v, or -1
__device__ __forceinline__ int hs_find(const int32_t* keys, uint32_t mask, int v) {
  uint32_t s = hs_home(v, mask);
  for (;;) {
    const int k = keys[s];
    if (k == v) return (int)s;
    if (k == -1) return -1;
    s = (s + 1) & mask;
  }
}

// Same contract as bfs_list (hop-major list, ascending id inside a hop, lvl_end), visited set =
// hash.  New nodes are appended in discovery order and every finished level is rank-sorted
// (each thread counts the smaller elements of the level: no barriers inside, levels are capped
// at kSparseLevelMax by the caller's choice of links).  `cnt` is one LDS int.
template <int T, int G>
__device__ __forceinline__ int bfs_hash(const int32_t* __restrict__ indptr,
                                        const int32_t* __restrict__ indices, int src, int dst,
                                        int hops, int32_t* keys, int32_t* vals, uint32_t mask,
                                        int32_t* list, int cap, int* lvl_end, int* cnt, int* hub,
                                        int& nlev_out, const WalkSets ws = WalkSets{}, int link = 0) {
  const int tid = threadIdx.x;
  const bool walks = walks_on(ws);
  for (uint32_t t = tid; t <= mask; t += T) keys[t] = -1;
  __syncthreads();
  if (tid == 0) {
    const int a = min(src, dst), b = max(src, dst);
    hs_insert(keys, mask, a);
    hs_insert(keys, mask, b);
    vals[hs_find(keys, mask, a)] = 0;
    vals[hs_find(keys, mask, b)] = 1;
    list[0] = a;
    list[1] = b;
    lvl_end[0] = 2;
    *cnt = 2;
  }
  hub_rows_clear<T>(hub);
  __syncthreads();
  int n = 2, nlev = 1;
  if (walks) hops = 1;
  for (int d = 1; d <= hops; ++d) {
    const int f0 = d >= 2 ? lvl_end[d - 2] : 0, f1 = n;
    if (walks) {
      for_each_walk_node(ws, src, dst, link, tid, T, [&](int u) {
        if (hs_insert(keys, mask, u)) {
          const int pos = atomicAdd(cnt, 1);
          if (pos < cap) list[pos] = u;
        }
      });
    } else {
      walk_rows<T, G, 2>(
          f0, f1, list, indptr, indices, hub,
          [&](RowAcc&, int, int u, bool valid) {
            if (valid && hs_insert(keys, mask, u)) {
              const int pos = atomicAdd(cnt, 1);
              if (pos < cap) list[pos] = u;
            }
          },
          [](RowAcc&, int, int) {});
    }
    __syncthreads();
    const int n_new = min(*cnt, cap);
    const int added = n_new - n;
    if (added == 0) break;
    // rank sort of list[n .. n_new): up to kSparseLevelMax / T elements per thread
    constexpr int PER = (kSparseLevelMax + T - 1) / T;
    int x[PER], r[PER];
#pragma unroll
    for (int k = 0; k < PER; ++k) {
      const int t = tid + k * T;
      x[k] = t < added ? list[n + t] : 0x7fffffff;
      r[k] = 0;
    }
    for (int j = 0; j < added; ++j) {
      const int y = list[n + j];   // same address in every lane: LDS broadcast
#pragma unroll
      for (int k = 0; k < PER; ++k) r[k] += y < x[k] ? 1 : 0;
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < PER; ++k) {
      if (tid + k * T < added) {
        list[n + r[k]] = x[k];
        vals[hs_find(keys, mask, x[k])] = n + r[k];
      }
    }
    n = n_new;
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
