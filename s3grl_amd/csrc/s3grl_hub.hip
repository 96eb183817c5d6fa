// Hub neighbourhoods: one-hop plans on big power-law graphs (reference utils.py:57-80 with num_hops = 1,
// tuned_SIGN.py:153-175).
//
// A one-hop subgraph is S = {src, dst} ∪ N(src) ∪ N(dst).  When one endpoint is a hub h (thousands of
// neighbours) nearly all of S is N(h) — and N(h), with every edge inside it, is the same for EVERY link
// that has h as an endpoint (a split of a power-law graph holds hundreds of links per hub).
// link_full_kernel (s3grl_onehop.inl) rediscovers that neighbourhood link by link: ~30 000 probes of the
// oriented rows, a CSR build and a sort per row for a 3 000-node subgraph.  Here:
//
//   build_hub_cache   once per graph: for every node h with deg >= kHubMinDegree (no self-loop) the induced
//                     adjacency of N(h) as a CSR whose columns are POSITIONS in h's sorted row (uint16),
//                     rows ascending.  The edges h—x are implicit.
//   link_hub_kernel   per link (h, o): local ids 0 = h, 1..c = N(h) in row order, then the nodes only o
//                     brings (o itself when it is no neighbour of h, N(o) \ N(h)).  Only THOSE nodes' rows
//                     are walked (Σ deg over N(o) \ N(h): a few thousand neighbour tests against the staged
//                     row of h); the edges found are a small CSR next to the cached one, and every operator
//                     is a pull over cache + star + small CSR.  The masked edge src—dst is the star edge
//                     h—o: left out of o's row, of h's sum and of both degrees.
//
// Which links: count1_kernel decides per link — from the graph and the link alone (never from the rest
// of the list: a link gives the same bits in a sharded and in an unsharded run) — hub = the endpoint of
// higher (degree, then lower id), in the cache, Σ degree over N(other) <= kHubVolMax, LDS need within a
// CU; classify_kernel then sorts them into kHubClasses LDS classes.  Everything else stays on
// link_full_kernel.  Same rows and coefficients as there up to the summation order (cache entries in
// row order, the star edge, then the found edges ascending — a fixed order).
#include <hip/hip_runtime.h>

#include <algorithm>
#include <type_traits>
#include <cstdio>
#include <cstdlib>

#include "s3grl_device.hpp"

namespace s3grl {
namespace {

// Rows of a pull by their length (cached + found entries; the star edge comes on top).  Half of a hub's
// neighbours have no edge inside the neighbourhood at all and nine in ten at most a handful: one LANE per
// row up to kHubTinyRow entries, four lanes up to kHubLongRow, a whole wavefront beyond (at most
// kHubLongCap rows per link — the first in row order — further ones stay with their four lanes).  (Row
// counts per tier travel through one block scan packed 16 + 16 bits: a link here has a few thousand rows.)
#ifndef S3GRL_HUB_TINY_ROW
#define S3GRL_HUB_TINY_ROW 4
#endif
#ifndef S3GRL_HUB_LONG_ROW
#define S3GRL_HUB_LONG_ROW 48
#endif
constexpr int kHubTinyRow = S3GRL_HUB_TINY_ROW;   // (build-time tuning hooks, like S3GRL_HUB_G)
constexpr int kHubLongRow = S3GRL_HUB_LONG_ROW;
constexpr int kHubLongCap = 128;
constexpr int kHubShWords = 72;   // 40 of scan / counter words, 32 of row 0's partial sums
#ifndef S3GRL_HUB_G
#define S3GRL_HUB_G 4
#endif

__device__ __forceinline__ int lds_lower_bound(const int32_t* a, int n, int x) {
  int lo = 0, hi = n;
  while (lo < hi) {
    const int mid = (lo + hi) >> 1;
    if (a[mid] < x) lo = mid + 1; else hi = mid;
  }
  return lo;
}

// ---- the cache ---------------------------------------------------------------------------------------
// eight lanes per node: Σ degree over the row, self-loop test, hub flag
__global__ void hub_mark_kernel(const int32_t* __restrict__ indptr, const int32_t* __restrict__ indices, int64_t N,
                                int min_deg, int32_t* __restrict__ flag, int32_t* __restrict__ voln) {
  const int64_t gid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int64_t u = gid >> 3;
  const int g = (int)(gid & 7);
  if (u >= N) return;
  const int b = indptr[u], e = indptr[u + 1];
  long long sum = 0;
  int loop = 0;
  for (int k = b + g; k < e; k += 8) {
    const int v = indices[k];
    sum += indptr[v + 1] - indptr[v];
    loop |= v == (int)u ? 1 : 0;
  }
#pragma unroll
  for (int o = 4; o > 0; o >>= 1) {
    sum += __shfl_xor(sum, o);
    loop |= __shfl_xor(loop, o);
  }
  if (g == 0) {
    voln[u] = (int32_t)min(sum, (long long)0x7fffffff);
    const int d = e - b;
    flag[u] = (d >= min_deg && d <= 65535 && !loop) ? 1 : 0;
  }
}

__global__ void hub_list_kernel(const int32_t* __restrict__ indptr, const int32_t* __restrict__ flag,
                                const int64_t* __restrict__ off, int64_t N, int32_t* __restrict__ slot,
                                int32_t* __restrict__ hubs, int32_t* __restrict__ rows,
                                const int32_t* __restrict__ voln, unsigned long long* __restrict__ work) {
  const int64_t u = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (u >= N) return;
  if (flag[u]) {
    const int k = (int)off[u];
    slot[u] = k;
    hubs[k] = (int32_t)u;
    rows[k] = indptr[u + 1] - indptr[u] + 1;   // one offset entry per neighbour + the end
    atomicAdd(work, (unsigned long long)voln[u]);   // neighbour tests the cache of this hub costs to build
  } else {
    slot[u] = -1;
  }
}

// one wavefront per cached row (hub k, position p): N(v) ∩ N(h) for v = the p-th neighbour of h, as positions
// in h's row.  FILL = false counts, FILL = true writes (ascending: the lanes walk N(v) in order).
template <bool FILL>
__global__ __launch_bounds__(256) void hub_rows_kernel(const int32_t* __restrict__ indptr,
                                                       const int32_t* __restrict__ indices,
                                                       const int32_t* __restrict__ hubs,
                                                       const int64_t* __restrict__ row_base, int nh, int64_t rows_total,
                                                       int32_t* __restrict__ cnt, const int64_t* __restrict__ abs_off,
                                                       uint16_t* __restrict__ hcols) {
  const int lane = threadIdx.x & 63;
  const int64_t r = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (r >= rows_total) return;
  int lo = 0, hi = nh;   // the hub whose rows hold r: last k with row_base[k] <= r
  while (hi - lo > 1) {
    const int mid = (lo + hi) >> 1;
    if (row_base[mid] <= r) lo = mid; else hi = mid;
  }
  const int k = lo;
  const int p = (int)(r - row_base[k]);
  const int h = hubs[k];
  const int32_t* __restrict__ H = indices + indptr[h];
  const int c = indptr[h + 1] - indptr[h];
  if (p >= c) {   // the end entry of the hub
    if (!FILL && lane == 0) cnt[r] = 0;
    return;
  }
  const int v = H[p];
  const int32_t* __restrict__ row = indices + indptr[v];
  const int len = indptr[v + 1] - indptr[v];
  int total = 0;
  for (int e0 = 0; e0 < len; e0 += 64) {
    const int e = e0 + lane;
    int pos = -1;
    if (e < len) {
      const int y = row[e];
      int a = 0, b = c;
      while (a < b) {
        const int mid = (a + b) >> 1;
        if (H[mid] < y) a = mid + 1; else b = mid;
      }
      if (a < c && H[a] == y) pos = a;
    }
    const unsigned long long m = __ballot(pos >= 0);
    if (FILL && pos >= 0) hcols[abs_off[r] + total + __popcll(m & ((1ull << lane) - 1ull))] = (uint16_t)pos;
    total += __popcll(m);
  }
  if (!FILL && lane == 0) cnt[r] = total;
}

// offsets of a hub's rows relative to its first column; first column of every hub
__global__ void hub_offsets_kernel(const int64_t* __restrict__ row_base, int nh, int64_t rows_total,
                                   const int64_t* __restrict__ abs_off, int32_t* __restrict__ hoff,
                                   int64_t* __restrict__ col_base) {
  const int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (r <= nh) col_base[r] = abs_off[r < nh ? row_base[r] : rows_total];
  if (r >= rows_total) return;
  int lo = 0, hi = nh;
  while (hi - lo > 1) {
    const int mid = (lo + hi) >> 1;
    if (row_base[mid] <= r) lo = mid; else hi = mid;
  }
  hoff[r] = (int32_t)(abs_off[r] - abs_off[row_base[lo]]);
}

// ---- the per-link kernel -----------------------------------------------------------------------------
// LDS (dynamic): cn[cn_cap] cnpos[cn_cap] zbuf[4K] sh[72] longrows[kHubLongCap] dinv[n] offx[n+1] | region |
// the hub's cached rows (offsets, columns: uint16) | tier[n] | cols[2·xcap] (uint16), hub_lds_need() +
// hub_stage_bytes().  region = the state arrays cur[n], nxs[n] (float2) of the passes; before them it
// holds the node list nl[n-1] (local ids 1..n-1: N(h) staged, then the other endpoint's nodes) and the list
// of found edges, then the scatter cursors, the lists of rows to sort and the per-wave sort bitmaps.
// XG (the class of links whose BOUND of found edges does not fit LDS — the bound is loose, Σ degree over
// the other endpoint's neighbourhood, the edges found are a few per cent of it): a persistent grid, the
// list of found edges in this workgroup's HBM slice, the small CSR's columns on chip whenever the EXACT
// count fits what the link leaves of the LDS, else in the slice too.
template <int T, int K, bool XG>
__global__ __launch_bounds__(T) void link_hub_kernel(const HubLinkArgs a, const int32_t* __restrict__ class_list,
                                                     int count, int lds_bytes) {
  extern __shared__ uint32_t smem[];
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  constexpr int G = S3GRL_HUB_G;   // lanes per row of a pull
  int item = blockIdx.x;
  if (item >= count) return;
  do {   // (one link per workgroup; XG: a persistent grid over the class, one slice per workgroup)
  const int32_t* __restrict__ indptr = a.indptr;
  const int32_t* __restrict__ indices = a.indices;
  auto ext = [&](int v) -> int { return a.old_of_new ? a.old_of_new[v] : v; };
  unsigned long long t_prev = a.dbg ? __builtin_amdgcn_s_memtime() : 0ull;
#define S3GRL_HSTAMP(idx)                                                \
  if (a.dbg) {                                                           \
    __syncthreads();                                                     \
    const unsigned long long t_now = __builtin_amdgcn_s_memtime();       \
    if (threadIdx.x == 0) atomicAdd(&a.dbg[idx], t_now - t_prev);        \
    t_prev = t_now;                                                      \
  }
  const int l = class_list[item];
  const int64_t noff = a.node_off[l];
  const int n = (int)(a.node_off[l + 1] - noff);
  const int xcap = (int)(a.x_cap[l] & 0xffffffffll);   // (classified here: the entry is >= 0)
  const int xcap_e = XG ? 0 : ((xcap + 1) & ~1);   // found edges the LDS layout provides for
  const int mirror = a.mirror_of ? a.mirror_of[l] : -1;
  const int64_t mrp = mirror >= 0 ? a.row_ptr[mirror] : -1;
  const int cn_cap = a.cn_cap;
  const int src = (int)a.links[2 * (int64_t)l], dst = (int)a.links[2 * (int64_t)l + 1];
  const int cs = indptr[src + 1] - indptr[src], cd = indptr[dst + 1] - indptr[dst];
  const bool src_hub = cs > cd || (cs == cd && src < dst);   // the rule of count1_kernel
  const int h = src_hub ? src : dst, o = src_hub ? dst : src;
  const int c = src_hub ? cs : cd, co = src_hub ? cd : cs;
  const int32_t* __restrict__ Hrow = indices + indptr[h];
  const int32_t* __restrict__ Orow = indices + indptr[o];
  const int hk = a.hub.slot[h];
  const int32_t* __restrict__ hoff_g = a.hub.hoff + a.hub.row_base[hk];
  const uint16_t* __restrict__ hcols_g = a.hub.hcols + a.hub.col_base[hk];
  const int64_t hub_entries = a.hub.col_base[hk + 1] - a.hub.col_base[hk];
  // the hub's cached rows are staged in LDS (every pass walks all of them: from HBM / L2 each trip of a
  // pass would be two dependent loads); hubs of more than 65 535 cached entries stay where they are
  const bool staged = hub_stage_bytes(c, hub_entries) > 0;

  int32_t* cn = reinterpret_cast<int32_t*>(smem);
  int32_t* cnpos = cn + cn_cap;
  float* zbuf = reinterpret_cast<float*>(cnpos + cn_cap);          // [2][K][2]
  int* sh = reinterpret_cast<int*>(zbuf + 4 * K);                  // [kHubShWords]
  float* red = reinterpret_cast<float*>(sh + 40);                      // [2 * T / 64]
  uint16_t* longrows = reinterpret_cast<uint16_t*>(sh + kHubShWords);   // [kHubLongCap]
  float* dinv = reinterpret_cast<float*>(sh + kHubShWords + kHubLongCap / 2);
  int32_t* offx = reinterpret_cast<int32_t*>(dinv + n);            // [n + 1]
  uint32_t* region = reinterpret_cast<uint32_t*>(offx + n + 1 + ((n + 1) & 1));
  const int region_words = max(4 * n, n + xcap_e);
  uint16_t* hoff_l = reinterpret_cast<uint16_t*>(region + region_words);  // [c + 1] (staged)
  uint16_t* hcols_l = hoff_l + (staged ? ((c + 2) & ~1) : 0);             // [hub_entries]
  uint16_t* tier = hcols_l + (staged ? (((int)hub_entries + 1) & ~1) : 0);   // [n]: tiny rows from the front, four-lane rows from the back
  uint16_t* cols_l = tier + ((n + 1) & ~1);                                  // [2 * xcap] (XG: what is left)
  uint32_t* slice = XG ? a.slices + (int64_t)blockIdx.x * a.slice_words : nullptr;
  uint16_t* cols_g = XG ? reinterpret_cast<uint16_t*>(slice + a.slice_words / 2) : nullptr;
  const int cols_cap = XG ? (lds_bytes - (int)(reinterpret_cast<char*>(cols_l) - reinterpret_cast<char*>(smem))) / 2 : 0;
  bool cols_on_chip = true;
  auto cx_ld = [&](int k) -> int {
    if constexpr (XG) return cols_on_chip ? (int)cols_l[k] : (int)cols_g[k]; else return (int)cols_l[k];
  };
  auto cx_st = [&](int k, int v) {
    if constexpr (XG) {
      if (cols_on_chip) cols_l[k] = (uint16_t)v; else cols_g[k] = (uint16_t)v;
    } else {
      cols_l[k] = (uint16_t)v;
    }
  };
  auto hoff = [&](int t) -> int { return staged ? (int)hoff_l[t] : hoff_g[t]; };
  auto hcols = [&](int k) -> int { return staged ? (int)hcols_l[k] : (int)hcols_g[k]; };
  float2* cur = reinterpret_cast<float2*>(region);
  float2* nxs = cur + n;
  int32_t* nl = reinterpret_cast<int32_t*>(region);                // [n - 1]
  uint32_t* elist = XG ? slice : region + n;                       // [xcap]
  int32_t* cursor = reinterpret_cast<int32_t*>(region);            // [n] (nl is dead by then)

  // ---- the nodes: N(h) staged, then what only the other endpoint brings ---------------------------------
  for (int e = tid; e < c; e += T) nl[e] = Hrow[e];
  if (staged) {
    for (int e = tid; e <= c; e += T) hoff_l[e] = (uint16_t)hoff_g[e];
    for (int e = tid; e < (int)hub_entries; e += T) hcols_l[e] = hcols_g[e];
  }
  if (tid == 0) {
    sh[29] = 0;   // long rows registered
    sh[30] = 0;   // edges found
  }
  __syncthreads();
  const int pos_o = lds_lower_bound(nl, c, o);
  const bool o_in_h = pos_o < c && nl[pos_o] == o;
  const int lo = o_in_h ? 1 + pos_o : c + 1;      // local id of the other endpoint
  const int qbase = c + (o_in_h ? 0 : 1);         // list position of the first node of N(o) \ N(h)
  const int tn = n - 1 - qbase;                   // how many of those the sizing pass counted
  if (!o_in_h && tid == 0 && c < n - 1) nl[c] = o;
  {
    // ordered compaction of row(o): the new nodes (neither h, o nor in N(h)) and — PoS Plus — the common
    // neighbours (in N(h); o itself when it carries a self-loop: tuned_SIGN.py:233 on the masked matrix)
    const int per = (co + T - 1) / T;
    const int e0 = min(tid * per, co), e1 = min(e0 + per, co);
    int packed = 0;
    for (int e = e0; e < e1; ++e) {
      const int x = Orow[e];
      const int p = lds_lower_bound(nl, c, x);
      const bool in_h = p < c && nl[p] == x;
      packed += (!in_h && x != h && x != o) ? 1 : 0;
      packed += (a.plus && (in_h || x == o)) ? (1 << 16) : 0;
    }
    int total;
    int run = block_excl_scan<T>(packed, sh, total);
    for (int e = e0; e < e1; ++e) {
      const int x = Orow[e];
      const int p = lds_lower_bound(nl, c, x);
      const bool in_h = p < c && nl[p] == x;
      if (!in_h && x != h && x != o) {
        const int pos = qbase + (run & 0xffff);
        if (pos < n - 1) nl[pos] = x;
        run += 1;
      }
      if (a.plus && (in_h || x == o)) {
        if ((run >> 16) < cn_cap) cn[run >> 16] = x;
        run += 1 << 16;
      }
    }
  }
  __syncthreads();
  S3GRL_HSTAMP(0)
  const int32_t* Tl = nl + qbase;
  // output position of a local id: the two endpoints first (hop 0 of the exported lists), h before o
  auto opos = [&](int t) -> int { return t == 0 ? 0 : (t == lo ? 1 : (t < lo ? t + 1 : t)); };
  long long vol_local = tid == 0 ? (long long)a.hub.voln[h] + c : 0ll;
  for (int t = tid; t < n; t += T) {
    const int v = t == 0 ? h : nl[t - 1];
    a.c_ids[noff + opos(t)] = ext(v);
    if (t > c) vol_local += indptr[v + 1] - indptr[v];
  }
  const int64_t rp = a.row_ptr[l];
  const int R = (int)(a.row_ptr[l + 1] - rp);
  if (a.plus && tid < 64) {
    const int cc = R - 2;
    if (a.old_of_new && cc > 1) {   // rows in ascending order of the caller's ids (see link_kernel)
      int* key = cn + cc;
      int* srt = cn + 2 * cc;
      for (int i = tid; i < cc; i += 64) key[i] = a.old_of_new[cn[i]];
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
      __builtin_amdgcn_wave_barrier();
      for (int i = tid; i < cc; i += 64) {
        const int kk = key[i];
        int r = 0;
        for (int j = 0; j < cc; ++j) r += key[j] < kk ? 1 : 0;
        srt[r] = cn[i];
      }
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
      __builtin_amdgcn_wave_barrier();
      for (int i = tid; i < cc; i += 64) cn[i] = srt[i];
    }
  }
  if (tid == 0)
    for (int dd = 0; dd < kMaxLevels; ++dd) a.lvl[(int64_t)l * kMaxLevels + dd] = dd == 0 ? 2 : n;
  __syncthreads();
  for (int r = tid; r < R; r += T) {
    const int node = r == 0 ? src : (r == 1 ? dst : cn[r - 2]);
    a.row_nodes[rp + r] = ext(node);
    if (mirror >= 0) a.row_nodes[mrp + r] = ext(r == 0 ? dst : (r == 1 ? src : cn[r - 2]));
    if (r >= 2) cnpos[r - 2] = node == o ? lo : 1 + lds_lower_bound(nl, c, node);
  }
  const int pos_src = src_hub ? 0 : lo, pos_dst = src_hub ? lo : 0;
  S3GRL_HSTAMP(1)

  // ---- the edges the cache does not hold: rows of the other endpoint's nodes (utils.py:76-80) ----------
  // The rows are walked FLAT — entry e of their concatenation by thread e mod T — because they are rows of
  // the graph, not oriented rows: one neighbour of o that is itself a hub would pin four lanes for
  // hundreds of trips.  qoff (in dinv's space) = running offsets of the walked rows, qstart (in offx's) =
  // where each row starts in `indices`.
  const int qn = n - 1 - c;                      // walked rows: list positions c .. n-2, local ids c+1 .. n-1
  int* qoff = reinterpret_cast<int*>(dinv);      // [qn + 1]
  int* qstart = offx;                            // [qn]
  {
    const int per = (qn + T - 1) / T;
    const int r0 = min(tid * per, qn), r1 = min(r0 + per, qn);
    int mine = 0;
    for (int r = r0; r < r1; ++r) {
      const int v = nl[c + r];
      mine += indptr[v + 1] - indptr[v];
    }
    int total;
    int run = block_excl_scan<T>(mine, sh, total);
    for (int r = r0; r < r1; ++r) {
      const int v = nl[c + r];
      const int b = indptr[v];
      qoff[r] = run;
      qstart[r] = b;
      run += indptr[v + 1] - b;
    }
    if (tid == 0) qoff[qn] = total;
  }
  __syncthreads();
  const int walk_total = qoff[qn];
  for (int e0 = 0; e0 < walk_total; e0 += T) {
    const int e = e0 + tid;
    const bool valid = e < walk_total;
    int i = 0, j = -1;
    if (valid) {
      int ra = 0, rb = qn;   // the row holding entry e: last r with qoff[r] <= e
      while (rb - ra > 1) {
        const int mid = (ra + rb) >> 1;
        if (qoff[mid] <= e) ra = mid; else rb = mid;
      }
      const int u = indices[qstart[ra] + (e - qoff[ra])];
      i = c + 1 + ra;
      const int p = lds_lower_bound(nl, c, u);
      if (p < c && nl[p] == u) {
        j = 1 + p;
      } else if (!o_in_h && u == o) {
        j = c + 1;
      } else {
        const int p2 = lds_lower_bound(Tl, tn, u);
        if (p2 < tn && Tl[p2] == u) j = qbase + 1 + p2;
      }
    }
    // an edge between two of the walked rows shows up from both: kept from the lower one
    const bool found = valid && j >= 0 && (j <= c || j >= i);
    const unsigned long long fm = __ballot(found);
    int base = 0;
    if (fm) {
      const int leader = __ffsll((long long)fm) - 1;
      if (lane == leader) base = atomicAdd(&sh[30], __popcll(fm));
      base = __shfl(base, leader);
    }
    if (found) {
      const int k = base + __popcll(fm & ((1ull << lane) - 1ull));
      if (k < xcap) elist[k] = ((uint32_t)i << 16) | (uint32_t)j;
    }
  }
  if constexpr (XG) __threadfence();   // the edge list is read back by other waves (through L2)
  __syncthreads();   // nl, qoff, qstart dead from here
  S3GRL_HSTAMP(2)
  const int found_edges = min(sh[30], xcap);
  if constexpr (XG) cols_on_chip = 2 * found_edges <= cols_cap;
  for (int t = tid; t <= n; t += T) offx[t] = 0;
  __syncthreads();
  for (int k = tid; k < found_edges; k += T) {   // degrees of the small CSR
    const uint32_t w = elist[k];
    const int i = (int)(w >> 16), j = (int)(w & 0xffffu);
    atomicAdd(&offx[i], 1);
    if (i != j) atomicAdd(&offx[j], 1);
  }
  __syncthreads();

  // ---- degrees, D^-1/2 (inf -> 0), the small CSR ---------------------------------------------------------
  long long edges_local = 0;
  {
    const int per = (n + T - 1) / T;
    const int t0 = min(tid * per, n), t1 = min(t0 + per, n);
    // the rows' tiers in the pulls, in ascending row order (the lanes of a wavefront then write
    // neighbouring coefficients): tiny rows from the front of `tier`, four-lane rows from its back; row 0
    // — h, the sum over its whole row — is summed by the whole workgroup
    auto tier_of = [&](int t, int len) -> int { return t == 0 ? -1 : (len <= kHubTinyRow ? 0 : (len <= kHubLongRow ? 1 : 2)); };
    int mine = 0, packed = 0;
    for (int t = t0; t < t1; ++t) {
      const int dgx = offx[t];
      const int cached = (t >= 1 && t <= c) ? hoff(t) - hoff(t - 1) : 0;
      const int tr = tier_of(t, dgx + cached);
      mine += dgx;
      packed += tr == 0 ? 1 : (tr >= 1 ? (1 << 16) : 0);   // (long rows beyond the cap join the four-lane ones)
    }
    int nlong_mine = 0;
    for (int t = t0; t < t1; ++t) {
      const int cached = (t >= 1 && t <= c) ? hoff(t) - hoff(t - 1) : 0;
      nlong_mine += tier_of(t, offx[t] + cached) == 2 ? 1 : 0;
    }
    int total, ptotal, ltotal;
    int run = block_excl_scan<T>(mine, sh, total);
    int prun = block_excl_scan<T>(packed, sh, ptotal);
    // long rows in row order: the first kHubLongCap get a wavefront (not "the first to arrive": which
    // rows do decides the last bit of their sums)
    int lrun = block_excl_scan<T>(nlong_mine, sh, ltotal);
    if (tid == 0) {
      sh[33] = ptotal & 0xffff;
      sh[34] = ptotal >> 16;
      sh[29] = ltotal;
    }
    for (int t = t0; t < t1; ++t) {
      const int dgx = offx[t];
      const int cached = (t >= 1 && t <= c) ? hoff(t) - hoff(t - 1) : 0;
      const int star = t == 0 ? c - (o_in_h ? 1 : 0) : ((t <= c && t != lo) ? 1 : 0);
      const int dg = dgx + cached + star;
      edges_local += dg;
      const int tr = tier_of(t, dgx + cached);
      if (tr == 0) {
        tier[prun & 0xffff] = (uint16_t)t;
        prun += 1;
      } else if (tr >= 1) {
        // a slot in the four-lane list either way; long rows that get a wavefront leave a hole marked 0
        int q = kHubLongCap;
        if (tr == 2) q = lrun++;
        if (q < kHubLongCap) longrows[q] = (uint16_t)t;
        tier[n - 1 - (prun >> 16)] = q < kHubLongCap ? (uint16_t)0 : (uint16_t)t;
        prun += 1 << 16;
      }
      dinv[t] = dg > 0 ? 1.0f / sqrtf((float)dg) : 0.0f;
      offx[t] = run;
      cursor[t] = run;
      run += dgx;
    }
    if (tid == 0) offx[n] = total;
  }
  __syncthreads();
  for (int k = tid; k < found_edges; k += T) {
    const uint32_t w = elist[k];
    const int i = (int)(w >> 16), j = (int)(w & 0xffffu);
    cx_st(atomicAdd(&cursor[i], 1), j);
    if (i != j) cx_st(atomicAdd(&cursor[j], 1), i);
  }
  if constexpr (XG) {
    if (!cols_on_chip) __threadfence();
  }
  {
    // every row ascending (a fixed summation order).  Nearly all rows of the small CSR are empty or hold
    // one entry: the rows that need sorting are collected first (2..16 entries: by rank, four per
    // wavefront; longer ones one per wavefront: by rank up to 64 entries, through a per-wave bitmap of the
    // n local ids beyond).  Lists in the region behind the cursors / bitmaps (dead edge list).
    const int WB = (n + 31) >> 5;
    uint32_t* wbm = region + wv * WB;
    uint32_t* list_a = region + 2 * n;   // [n]
    uint32_t* list_b = region + 3 * n;   // [n]
    if (tid == 0) {
      sh[31] = 0;
      sh[32] = 0;
    }
    __syncthreads();   // (also: the scatter is complete, cursor dead)
    for (int r = tid; r < n; r += T) {
      const int len = offx[r + 1] - offx[r];
      if (len > 16) list_b[atomicAdd(&sh[32], 1)] = (uint32_t)r;
      else if (len > 1) list_a[atomicAdd(&sh[31], 1)] = (uint32_t)r;
    }
    __syncthreads();
    const int na = sh[31], nb = sh[32];
    const int sub = lane >> 4, sl = lane & 15;
    for (int q0 = wv * 4; q0 < na; q0 += (T / 64) * 4) {
      const bool mine = q0 + sub < na;
      const int r = (int)list_a[min(q0 + sub, na - 1)];
      const int b = offx[r], len = offx[r + 1] - b;
      const int x = mine && sl < len ? cx_ld(b + sl) : 0x7fffffff;
      int rank = 0;
#pragma unroll
      for (int k = 0; k < 16; ++k) rank += __shfl(x, (lane & 48) + k) < x ? 1 : 0;
      if (mine && sl < len) cx_st(b + rank, x);
    }
    for (int q = wv; q < nb; q += T / 64) {
      const int r = (int)list_b[q];
      const int b = offx[r], len = offx[r + 1] - b;
      if (len <= 64) {
        const int x = lane < len ? cx_ld(b + lane) : 0x7fffffff;
        int rank = 0;
        for (int k = 0; k < len; ++k) rank += __shfl(x, k) < x ? 1 : 0;
        if (lane < len) cx_st(b + rank, x);
      } else {
        for (int w = lane; w < WB; w += 64) wbm[w] = 0;
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        __builtin_amdgcn_wave_barrier();
        for (int k = lane; k < len; k += 64) {
          const int cc = cx_ld(b + k);
          atomicOr(&wbm[cc >> 5], 1u << (cc & 31));
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        __builtin_amdgcn_wave_barrier();
        int base = b;
        for (int w0 = 0; w0 < WB; w0 += 64) {
          uint32_t word = w0 + lane < WB ? wbm[w0 + lane] : 0u;
          const int cnt = __popc(word);
          int inc = cnt;
#pragma unroll
          for (int oo = 1; oo < 64; oo <<= 1) {
            const int t = __shfl_up(inc, oo);
            if (lane >= oo) inc += t;
          }
          int k = base + inc - cnt;
          while (word) {
            const int bit = __ffs(word) - 1;
            word &= word - 1;
            cx_st(k++, (w0 + lane) * 32 + bit);
          }
          base += __shfl(inc, 63);
        }
      }
    }
  }
  if constexpr (XG) {
    if (!cols_on_chip) __threadfence();
  }
  __syncthreads();   // the region becomes the state arrays
  S3GRL_HSTAMP(3)
  const int nlong = min(sh[29], kHubLongCap), ntiny = sh[33], nmid = sh[34];

  // ---- per row pair: K pulls over cache + star + small CSR -----------------------------------------------
  const int npairs = (R + 1) / 2;
  for (int pr = 0; pr < npairs; ++pr) {
    const int64_t jid = a.job_off[l] + pr;
    const int64_t coff = a.coef_off ? a.coef_off[jid] : noff;
    const int node_a = pr == 0 ? src : cn[2 * pr - 2];
    const int node_b = pr == 0 ? dst : (2 * pr + 1 < R ? cn[2 * pr - 1] : -1);
    const int la = pr == 0 ? pos_src : cnpos[2 * pr - 2];
    const int lb = pr == 0 ? pos_dst : (node_b >= 0 ? cnpos[2 * pr - 1] : -1);
    for (int w = tid; w < n; w += T)
      cur[w] = make_float2(w == la ? dinv[w] : 0.f, w == lb ? dinv[w] : 0.f);
    if (tid < 4 * K) zbuf[tid] = 0.f;
    __syncthreads();
    float2* s_in = cur;
    float2* s_out = nxs;
    float2* coef = reinterpret_cast<float2*>(a.c_coef) + coff * K;   // [K][n] float2
    const bool split = a.split_t > 0 && n > a.split_t;               // coefficients piece by piece (link_kernel)
    auto cidx_split = [&](int i, int t) -> int64_t {
      const int s0 = (t >> a.seg_shift) << a.seg_shift;
      const int len = min(1 << a.seg_shift, n - s0);
      return (int64_t)s0 * K + (int64_t)i * len + (t - s0);
    };
#pragma unroll 1
    for (int i = 0; i < K; ++i) {
      const int g = tid & (G - 1);
      float2* coef_i = coef + (int64_t)i * n;   // (unsplit lists: operator-major)
      auto commit = [&](int t, float ax, float ay) {
        const float dw = dinv[t];
        const float rx = dw * ax, ry = dw * ay;
        s_out[t] = make_float2(dw * rx, dw * ry);
        const int op = opos(t);
        if (!split) coef_i[op] = make_float2(rx, ry);
        else coef[cidx_split(i, op)] = make_float2(rx, ry);
        // label column of operator i+1: r[src] + r[dst]  (tuned_SIGN.py:177-185)
        if (t == pos_src) { zbuf[(0 * K + i) * 2] = rx; zbuf[(0 * K + i) * 2 + 1] = ry; }
        if (t == pos_dst) { zbuf[(1 * K + i) * 2] = rx; zbuf[(1 * K + i) * 2 + 1] = ry; }
      };
      if (i == 0) {
        // Operator 1 of a pair is Â e_a | Â e_b: every entry has at most ONE term per row of the pair — the
        // rows of a and b are spread instead of pulling over all of S (the same bits: a pull would add
        // that term to zeros).
        for (int t = tid; t < n; t += T) s_out[t] = make_float2(0.f, 0.f);
        __syncthreads();
        float* so = reinterpret_cast<float*>(s_out);
        auto spread = [&](int lx, int comp) {
          if (lx < 0) return;
          const float v = dinv[lx];   // = s_in[lx], the only non-zero of this component
          if (lx == 0) {
            for (int p = tid; p < c; p += T)
              if (1 + p != lo) so[2 * (1 + p) + comp] = v;
            return;
          }
          if (lx <= c) {
            const int k1 = hoff(lx);
            for (int k = hoff(lx - 1) + tid; k < k1; k += T) so[2 * (1 + hcols(k)) + comp] = v;
            if (tid == 0 && lx != lo) so[comp] = v;
          }
          const int k1 = offx[lx + 1];
          for (int k = offx[lx] + tid; k < k1; k += T) so[2 * cx_ld(k) + comp] = v;
        };
        spread(la, 0);
        spread(lb, 1);
        __syncthreads();
        for (int t = tid; t < n; t += T) {
          const float2 one = s_out[t];
          commit(t, one.x, one.y);
        }
        __syncthreads();
        float2* tmp2 = s_in;
        s_in = s_out;
        s_out = tmp2;
        continue;
      }
      {   // row 0: all of h's row but the masked edge, in slices of the workgroup (fixed reduction tree)
        float ax = 0.f, ay = 0.f;
        for (int p = tid; p < c; p += T) {
          if (1 + p != lo) {
            const float2 sv = s_in[1 + p];
            ax += sv.x;
            ay += sv.y;
          }
        }
#pragma unroll
        for (int oo = 32; oo > 0; oo >>= 1) {
          ax += __shfl_xor(ax, oo);
          ay += __shfl_xor(ay, oo);
        }
        if (lane == 0) {
          red[2 * wv] = ax;
          red[2 * wv + 1] = ay;
        }
      }
      for (int q = tid; q < ntiny; q += T) {   // one lane per tiny row: star, cached, found
        const int t = tier[q];
        float ax = 0.f, ay = 0.f;
        if (t <= c) {
          if (t != lo) {
            const float2 sv = s_in[0];
            ax = sv.x;
            ay = sv.y;
          }
          const int k1 = hoff(t);
          for (int k = hoff(t - 1); k < k1; ++k) {
            const float2 sv = s_in[1 + hcols(k)];
            ax += sv.x;
            ay += sv.y;
          }
        }
        const int k1 = offx[t + 1];
        for (int k = offx[t]; k < k1; ++k) {
          const float2 sv = s_in[cx_ld(k)];
          ax += sv.x;
          ay += sv.y;
        }
        commit(t, ax, ay);
      }
      for (int base = 0; base < nmid; base += T / G) {   // G lanes per row
        const int q = base + tid / G;
        const int t = tier[n - 1 - min(q, nmid - 1)];
        const bool mine = q < nmid && t != 0;   // (0: a long row, summed by a wavefront below)
        float ax = 0.f, ay = 0.f;
        if (mine) {
          if (t <= c) {
            const int k1 = hoff(t);
            for (int k = hoff(t - 1) + g; k < k1; k += G) {
              const float2 sv = s_in[1 + hcols(k)];
              ax += sv.x;
              ay += sv.y;
            }
            if (g == 0 && t != lo) {
              const float2 sv = s_in[0];
              ax += sv.x;
              ay += sv.y;
            }
          }
          const int k1 = offx[t + 1];
          for (int k = offx[t] + g; k < k1; k += G) {
            const float2 sv = s_in[cx_ld(k)];
            ax += sv.x;
            ay += sv.y;
          }
        }
#pragma unroll
        for (int oo = G / 2; oo > 0; oo >>= 1) {
          ax += __shfl_xor(ax, oo);
          ay += __shfl_xor(ay, oo);
        }
        if (mine && g == 0) commit(t, ax, ay);
      }
      for (int q = wv; q < nlong; q += T / 64) {   // one wavefront per long row
        const int t = longrows[q];
        float ax = 0.f, ay = 0.f;
        {
          if (t <= c) {
            const int k1 = hoff(t);
            for (int k = hoff(t - 1) + lane; k < k1; k += 64) {
              const float2 sv = s_in[1 + hcols(k)];
              ax += sv.x;
              ay += sv.y;
            }
            if (lane == 0 && t != lo) {
              const float2 sv = s_in[0];
              ax += sv.x;
              ay += sv.y;
            }
          }
          const int k1 = offx[t + 1];
          for (int k = offx[t] + lane; k < k1; k += 64) {
            const float2 sv = s_in[cx_ld(k)];
            ax += sv.x;
            ay += sv.y;
          }
        }
#pragma unroll
        for (int oo = 32; oo > 0; oo >>= 1) {
          ax += __shfl_xor(ax, oo);
          ay += __shfl_xor(ay, oo);
        }
        if (lane == 0) commit(t, ax, ay);
      }
      __syncthreads();
      if (tid == 0) {
        float ax = 0.f, ay = 0.f;
        for (int w = 0; w < T / 64; ++w) {
          ax += red[2 * w];
          ay += red[2 * w + 1];
        }
        commit(0, ax, ay);
      }
      __syncthreads();
      float2* tmp2 = s_in;
      s_in = s_out;
      s_out = tmp2;
    }
    if (tid < 2 * K) {
      const int i = tid >> 1, r = tid & 1;
      a.job_z[(jid * K + i) * 2 + r] = zbuf[(0 * K + i) * 2 + r] + zbuf[(1 * K + i) * 2 + r];
    }
    if (tid < K) a.job_lim[jid * K + tid] = n;   // one hop: every operator reaches the whole list
    if (tid == 0) {
      Job j;
      j.coef_off = coff * K;
      j.ids_off = noff;
      j.out_row = rp + 2 * pr;
      j.link = l;
      j.support = n;
      j.node_a = ext(node_a);
      j.node_b = node_b >= 0 ? ext(node_b) : -1;
      j.z_a = (node_a == src || node_a == dst) ? 1 : 0;
      j.z_b = (node_b == src || node_b == dst) ? 1 : 0;
      j.mirror_row = mirror >= 0 ? mrp + 2 * pr : -1;
      j.mirror_swap = pr == 0 ? 1 : 0;
      j.split = split ? 1 : 0;
      a.jobs[jid] = j;
      atomicAdd(stat_slot(a.tot_support), (unsigned long long)n * (mirror >= 0 ? 2ull : 1ull));
    }
    __syncthreads();
  }
  S3GRL_HSTAMP(4)
  // totals: Σ induced entries, Σ degrees in the graph (64-bit block sums through LDS)
  {
    long long* red = reinterpret_cast<long long*>(region);   // state arrays are dead
#pragma unroll
    for (int oo = 32; oo > 0; oo >>= 1) {
      edges_local += __shfl_xor(edges_local, oo);
      vol_local += __shfl_xor(vol_local, oo);
    }
    if (lane == 0) {
      red[2 * wv] = edges_local;
      red[2 * wv + 1] = vol_local;
    }
    __syncthreads();
    if (tid == 0) {
      long long e = 0, v = 0;
      for (int w = 0; w < T / 64; ++w) {
        e += red[2 * w];
        v += red[2 * w + 1];
      }
      atomicAdd(stat_slot(a.tot_edges), (unsigned long long)e * (mirror >= 0 ? 2ull : 1ull));
      atomicAdd(stat_slot(a.tot_vol), (unsigned long long)v * (mirror >= 0 ? 2ull : 1ull));
      // what this link requested (bench.py's physical-bytes figure): the two endpoint rows, the walked rows'
      // bounds and entries, the staged cache, the ids of the degree order
      atomicAdd(stat_slot(a.tot_hub_links), 1ull);
      atomicAdd(stat_slot(a.tot_hub_ends), (unsigned long long)(c + co));
      atomicAdd(stat_slot(a.tot_hub_nodes), (unsigned long long)n);
      // (count1_kernel summed the oriented-row entries of EVERY link: this one's are not probed)
      atomicAdd(stat_slot(a.tot_oriented), 0ull - (unsigned long long)(a.e_cap[l] / 2));
      atomicAdd(stat_slot(a.tot_hub_bytes),
                (unsigned long long)(4ll * c + 8ll * co + 16ll * qn + 4ll * walk_total +
                                     (staged ? 2ll * (c + 1) + 2ll * hub_entries : 8ll * c + 2ll * (K - 1) * hub_entries) +
                                     (a.old_of_new ? 4ll * n : 0ll)));
    }
  }
  S3GRL_HSTAMP(5)
  if constexpr (XG) __syncthreads();   // LDS is reused by the next item of a persistent workgroup
  item += gridDim.x;
  } while (XG && item < count);
#undef S3GRL_HSTAMP
}

template <int T, int K, bool XG>
s3grl_status launch_hub_t(const HubLinkArgs& a, int cls, const int32_t* class_list, int count, hipStream_t stream) {
  const size_t lds = (size_t)4 * hub_fixed_words(a.cn_cap, K) +
                     (size_t)hub_class_bound(std::min(cls, kHubClasses - 1), a.cn_cap, K);
  auto kern = link_hub_kernel<T, K, XG>;
  S3GRL_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize,
                                    (int)lds));
  const int grid = XG ? std::min(count, a.slice_grid) : count;
  hipLaunchKernelGGL(kern, dim3((unsigned)grid), dim3(T), lds, stream, a, class_list, count,
                     getenv("S3GRL_HUB_COLS_HBM") ? 0 : (int)lds);   // test hook: the columns in the slice too
  S3GRL_HIP_TRY(hipGetLastError());
  return S3GRL_OK;
}

template <int K>
s3grl_status launch_hub_k(const HubLinkArgs& a, int cls, const int32_t* class_list, int count, hipStream_t stream) {
  int t = cls == 0 ? 128 : (cls == 1 ? 256 : (cls == 2 ? 512 : 1024));
  if (cls < kHubClasses) {
    char name[32];   // tuning hook
    snprintf(name, sizeof(name), "S3GRL_TH_CLASS%d", cls);
    if (const char* e = getenv(name)) t = atoi(e);
  }
  if (cls == kHubClasses) return launch_hub_t<1024, K, true>(a, cls, class_list, count, stream);
  if (t <= 128) return launch_hub_t<128, K, false>(a, cls, class_list, count, stream);
  if (t <= 256) return launch_hub_t<256, K, false>(a, cls, class_list, count, stream);
  if (t <= 512) return launch_hub_t<512, K, false>(a, cls, class_list, count, stream);
  return launch_hub_t<1024, K, false>(a, cls, class_list, count, stream);
}

}  // namespace

int hub_fixed_words(int cn_cap, int K) { return 2 * cn_cap + 4 * K + kHubShWords + kHubLongCap / 2; }

int hub_class_bound(int cls, int cn_cap, int K) {
  static const int nominal[kHubClasses] = {20480, 40960, 81920, 163840};
  const int avail = 163840 - 4 * hub_fixed_words(cn_cap, K);
  return std::min(nominal[cls] - (cls < kHubClasses - 1 ? 4 * hub_fixed_words(cn_cap, K) : 0), avail);
}

s3grl_status launch_hub_class(s3grl_context* ctx, const HubLinkArgs& a, int K, int cls, const int32_t* class_list,
                              int count, hipStream_t stream) {
  (void)ctx;
  if (count <= 0) return S3GRL_OK;
  switch (K) {
    case 1: return launch_hub_k<1>(a, cls, class_list, count, stream);
    case 2: return launch_hub_k<2>(a, cls, class_list, count, stream);
    case 3: return launch_hub_k<3>(a, cls, class_list, count, stream);
    case 4: return launch_hub_k<4>(a, cls, class_list, count, stream);
    case 5: return launch_hub_k<5>(a, cls, class_list, count, stream);
    case 6: return launch_hub_k<6>(a, cls, class_list, count, stream);
    case 7: return launch_hub_k<7>(a, cls, class_list, count, stream);
    case 8: return launch_hub_k<8>(a, cls, class_list, count, stream);
    default: return S3GRL_ERR_INVALID_ARGUMENT;
  }
}

// ---- links of at most 32 / 64 nodes: half a wavefront / a wavefront each, one lane per node ---------------------
namespace {

// lanes of a group (W = 32: a half, `hb` = 0 / 32; W = 64: the wavefront, hb = 0) exchange through full-width
// shuffles with absolute lane numbers
__device__ __forceinline__ int hshfl(int v, int idx, int hb) { return __shfl(v, hb + idx); }
__device__ __forceinline__ float hshflf(float v, int idx, int hb) { return __shfl(v, hb + idx); }
template <int W>
__device__ __forceinline__ unsigned long long hballot(bool p, int hb) {
  const unsigned long long b = __ballot(p);
  return W == 32 ? (unsigned long long)(uint32_t)(b >> hb) : b;
}
// LDS written by some lanes of this wavefront, read by others (a wavefront's LDS instructions complete in
// order; the fence keeps the compiler from moving them)
__device__ __forceinline__ void wave_lds_sync() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
}
// first index in [0, len) whose value (lane idx of `vals`, ascending) is >= x; every lane runs all the steps
template <int W>
__device__ __forceinline__ int hlower_bound(int vals, int len, int x, int hb) {
  int lo = 0, hi = len;
#pragma unroll
  for (int st = 0; st < (W == 32 ? 6 : 7); ++st) {
    const int mid = (lo + hi) >> 1;
    const int y = hshfl(vals, min(mid, W - 1), hb);
    const bool go = lo < hi;
    const bool lt = y < x;
    lo = go && lt ? mid + 1 : lo;
    hi = go && !lt ? mid : hi;
  }
  return lo;
}

#ifndef S3GRL_TINY_THREADS
#define S3GRL_TINY_THREADS 256   // threads per workgroup of link_tiny_kernel (build-time tuning hook)
#endif
constexpr int kTinyThreads = S3GRL_TINY_THREADS;

template <int K, int W>
__global__ __launch_bounds__(kTinyThreads) void link_tiny_kernel(const TinyLinkArgs a,
                                                                 const int32_t* __restrict__ class_list, int count) {
  constexpr int kTinyLinksPerBlock = kTinyThreads / W;
  using Mask = typename std::conditional<W == 32, uint32_t, unsigned long long>::type;
  __shared__ int32_t s_list[kTinyLinksPerBlock][W];
  __shared__ Mask s_adj[kTinyLinksPerBlock][W];
  __shared__ float2 s_state[kTinyLinksPerBlock][W];
  const int sub = threadIdx.x / W, t = threadIdx.x & (W - 1), hb = W == 32 ? (threadIdx.x & 32) : 0;
  const int item = blockIdx.x * kTinyLinksPerBlock + sub;
  if (item >= count) return;   // (no workgroup barrier below: a half may leave)
  const int32_t* __restrict__ indptr = a.indptr;
  const int32_t* __restrict__ indices = a.indices;
  auto ext = [&](int v) -> int { return a.old_of_new ? a.old_of_new[v] : v; };
  const int l = class_list[item];
  const int64_t noff = a.node_off[l];
  const int n = (int)(a.node_off[l + 1] - noff);
  const int mirror = a.mirror_of ? a.mirror_of[l] : -1;
  const int src = (int)a.links[2 * (int64_t)l], dst = (int)a.links[2 * (int64_t)l + 1];
  const int bs = indptr[src], bd = indptr[dst];
  const int cs = indptr[src + 1] - bs, cd = indptr[dst + 1] - bd;   // <= W each: n <= W
  int32_t* list = s_list[sub];
  Mask* adjw = s_adj[sub];
  float2* state = s_state[sub];
  constexpr int kNone = 0x7fffffff;

  // ---- S in canonical order (link_full_kernel's): {min, max}, then N(src) ∪ N(dst) \ {src, dst} ascending ----
  const int xs = t < cs ? indices[bs + t] : kNone;
  const int xd = t < cd ? indices[bd + t] : kNone;
  adjw[t] = 0;
  const bool mem_s = t < cs && xs != src && xs != dst, mem_d = t < cd && xd != src && xd != dst;
  const int lb_d = hlower_bound<W>(xd, cd, xs, hb);   // members of row dst below xs (excluded entries counted)
  const int lb_s = hlower_bound<W>(xs, cs, xd, hb);
  // (every shuffle outside of divergent control flow: a lane that is switched off hands out nothing)
  const int at_d = hshfl(xd, min(lb_d, W - 1), hb), at_s = hshfl(xs, min(lb_s, W - 1), hb);
  const bool dup_s = mem_s && lb_d < cd && at_d == xs;
  const bool dup_d = mem_d && lb_s < cs && at_s == xd;
  const bool s_has_s = hballot<W>(t < cs && xs == src, hb) != 0ull, s_has_d = hballot<W>(t < cs && xs == dst, hb) != 0ull;
  const bool d_has_s = hballot<W>(t < cd && xd == src, hb) != 0ull, d_has_d = hballot<W>(t < cd && xd == dst, hb) != 0ull;
  const unsigned long long below = (1ull << t) - 1ull;
  const int cdup_s = __popcll(hballot<W>(dup_s, hb) & below), cdup_d = __popcll(hballot<W>(dup_d, hb) & below);
  if (mem_s) {   // common members are emitted from row src
    const int own = t - ((s_has_s && src < xs) ? 1 : 0) - ((s_has_d && dst < xs) ? 1 : 0);
    const int oth = lb_d - ((d_has_s && src < xs) ? 1 : 0) - ((d_has_d && dst < xs) ? 1 : 0);
    const int pos = 2 + own + oth - cdup_s;
    if (pos < n) list[pos] = xs;
  }
  if (mem_d && !dup_d) {
    const int own = t - ((d_has_s && src < xd) ? 1 : 0) - ((d_has_d && dst < xd) ? 1 : 0);
    const int oth = lb_s - ((s_has_s && src < xd) ? 1 : 0) - ((s_has_d && dst < xd) ? 1 : 0);
    const int pos = 2 + own + oth - cdup_d;
    if (pos < n) list[pos] = xd;
  }
  if (t == 0) {
    list[0] = min(src, dst);
    list[1] = max(src, dst);
  }
  wave_lds_sync();
  const bool live = t < n;
  const int v = live ? list[t] : kNone;   // this lane's node; lanes 2 .. n-1 ascending
  if (live) a.c_ids[noff + t] = ext(v);
  int vol = live ? indptr[v + 1] - indptr[v] : 0;

  // ---- masked induced adjacency through the oriented rows (reference utils.py:76-80), one mask per lane ----
  const int fb = live ? a.fwd_indptr[v] : 0;
  const int flen = live ? a.fwd_indptr[v + 1] - fb : 0;
  int incl = flen;
#pragma unroll
  for (int o = 1; o < W; o <<= 1) {
    const int up = __shfl_up(incl, o);
    if (t >= o) incl += up;
  }
  const int excl = incl - flen;
  const int walk_total = hshfl(incl, W - 1, hb);
  const int vsorted = t >= 2 ? v : -1;   // lanes 0, 1 hold the endpoints (any order): searched apart
  const int v0 = hshfl(v, 0, hb), v1 = hshfl(v, 1, hb);
  for (int e0 = 0; e0 < walk_total; e0 += W) {
    const int e = e0 + t;
    // the row holding entry e: last r with excl[r] <= e  =  (first r with excl[r] > e) - 1
    const int first_gt = hlower_bound<W>(excl, n, e + 1, hb);
    const int row = max(first_gt - 1, 0);
    const int rstart = hshfl(fb, row, hb), roff = hshfl(excl, row, hb), vi = hshfl(v, row, hb);
    const bool valid = e < walk_total;
    const int u = valid ? a.fwd_indices[rstart + (e - roff)] : -1;
    const int p = hlower_bound<W>(vsorted, n, u, hb);   // (lanes 0, 1 read as -1: below every id)
    const int at_p = hshfl(v, min(p, W - 1), hb);
    int j = -1;
    if (u == v0) j = 0;
    else if (u == v1) j = 1;
    else if (p >= 2 && p < n && at_p == u) j = p;
    const bool target = (vi == src && u == dst) || (vi == dst && u == src);
    if (valid && j >= 0 && !target) {
      atomicOr(&adjw[row], (Mask)1 << j);
      if (row != j) atomicOr(&adjw[j], (Mask)1 << row);
    }
  }
  wave_lds_sync();
  const Mask adj = live ? adjw[t] : (Mask)0;
  const int dg = __popcll((unsigned long long)adj);
  const float dinv = dg > 0 ? 1.0f / sqrtf((float)dg) : 0.0f;
  int edges = dg;
#pragma unroll
  for (int o = W / 2; o > 0; o >>= 1) {
    edges += __shfl_xor(edges, o);
    vol += __shfl_xor(vol, o);
  }

  // ---- the row pair (src, dst): K pulls, a row summed in ascending local id ------------------------------------
  const int pos_src = src < dst ? 0 : 1, pos_dst = 1 - pos_src;
  const int64_t rp = a.row_ptr[l];
  const int64_t mrp = mirror >= 0 ? a.row_ptr[mirror] : -1;
  const int64_t jid = a.job_off[l];
  const int64_t coff = a.coef_off ? a.coef_off[jid] : noff;
  float2* __restrict__ coef = reinterpret_cast<float2*>(a.c_coef) + coff * K;   // [K][n] float2
  float sx = t == pos_src ? dinv : 0.f, sy = t == pos_dst ? dinv : 0.f;
#pragma unroll 1
  for (int i = 0; i < K; ++i) {
    state[t] = make_float2(sx, sy);
    wave_lds_sync();
    float ax = 0.f, ay = 0.f;
    Mask m = adj;
    while (m) {
      const int u = __ffsll((unsigned long long)m) - 1;
      m &= m - 1;
      const float2 sv = state[u];
      ax += sv.x;
      ay += sv.y;
    }
    wave_lds_sync();
    const float rx = dinv * ax, ry = dinv * ay;
    sx = dinv * rx;
    sy = dinv * ry;
    if (live) coef[(int64_t)i * n + t] = make_float2(rx, ry);
    // label column of operator i+1: r[src] + r[dst]  (tuned_SIGN.py:177-185)
    const float zx = hshflf(rx, pos_src, hb) + hshflf(rx, pos_dst, hb);
    const float zy = hshflf(ry, pos_src, hb) + hshflf(ry, pos_dst, hb);
    if (t == 0) {
      a.job_z[(jid * K + i) * 2] = zx;
      a.job_z[(jid * K + i) * 2 + 1] = zy;
    }
  }
  if (t < K) a.job_lim[jid * K + t] = n;
  if (t < 2) {
    a.row_nodes[rp + t] = ext(t == 0 ? src : dst);
    if (mirror >= 0) a.row_nodes[mrp + t] = ext(t == 0 ? dst : src);
  }
  if (t < kMaxLevels) a.lvl[(int64_t)l * kMaxLevels + t] = t == 0 ? 2 : n;
  if (t == 0) {
    Job j;
    j.coef_off = coff * K;
    j.ids_off = noff;
    j.out_row = rp;
    j.link = l;
    j.support = n;
    j.node_a = ext(src);
    j.node_b = ext(dst);
    j.z_a = 1;
    j.z_b = 1;
    j.mirror_row = mirror >= 0 ? mrp : -1;
    j.mirror_swap = 1;
    j.split = 0;
    a.jobs[jid] = j;
    const unsigned long long mult = mirror >= 0 ? 2ull : 1ull;
    atomicAdd(stat_slot(a.tot_support), (unsigned long long)n * mult);
    atomicAdd(stat_slot(a.tot_edges), (unsigned long long)edges * mult);
    atomicAdd(stat_slot(a.tot_vol), (unsigned long long)vol * mult);
  }
}

template <int K>
s3grl_status launch_tiny_k(const TinyLinkArgs& a, int width, const int32_t* class_list, int count, hipStream_t stream) {
  const int per = kTinyThreads / width;
  const unsigned grid = (unsigned)((count + per - 1) / per);
  if (width == 32)
    hipLaunchKernelGGL((link_tiny_kernel<K, 32>), dim3(grid), dim3(kTinyThreads), 0, stream, a, class_list, count);
  else
    hipLaunchKernelGGL((link_tiny_kernel<K, 64>), dim3(grid), dim3(kTinyThreads), 0, stream, a, class_list, count);
  S3GRL_HIP_TRY(hipGetLastError());
  return S3GRL_OK;
}

}  // namespace

s3grl_status launch_tiny_class(s3grl_context* ctx, const TinyLinkArgs& a, int K, int width, const int32_t* class_list,
                               int count, hipStream_t stream) {
  (void)ctx;
  if (count <= 0) return S3GRL_OK;
  switch (K) {
    case 1: return launch_tiny_k<1>(a, width, class_list, count, stream);
    case 2: return launch_tiny_k<2>(a, width, class_list, count, stream);
    case 3: return launch_tiny_k<3>(a, width, class_list, count, stream);
    case 4: return launch_tiny_k<4>(a, width, class_list, count, stream);
    case 5: return launch_tiny_k<5>(a, width, class_list, count, stream);
    case 6: return launch_tiny_k<6>(a, width, class_list, count, stream);
    case 7: return launch_tiny_k<7>(a, width, class_list, count, stream);
    case 8: return launch_tiny_k<8>(a, width, class_list, count, stream);
    default: return S3GRL_ERR_INVALID_ARGUMENT;
  }
}

void release_hub_cache(s3grl_graph* g) {
  HubCache& hc = g->hub;
  g->ctx->arena.release(hc.slot);
  g->ctx->arena.release(hc.voln);
  g->ctx->arena.release(hc.row_base);
  g->ctx->arena.release(hc.col_base);
  g->ctx->arena.release(hc.hoff);
  g->ctx->arena.release(hc.hcols);
  hc = HubCache{};
}

// The cache of the degree-ordered graph (the one one-hop plans walk).  Nothing is built — and every link
// stays on link_full_kernel — when the graph has no node of kHubMinDegree neighbours, or when the induced
// neighbourhoods would take more than 1 GiB (dense hubs: their links are no cheaper this way).
s3grl_status build_hub_cache(s3grl_context* ctx, s3grl_graph* g) {
  if (!g->r_indptr || getenv("S3GRL_NO_HUB_CACHE")) return S3GRL_OK;
  int min_deg = kHubMinDegree;
  if (const char* e = getenv("S3GRL_HUB_MIN_DEG")) min_deg = std::max(2, atoi(e));   // test / tuning hook
  if (g->max_degree < min_deg) return S3GRL_OK;
  const int64_t N = g->num_nodes;
  const int32_t* indptr = g->r_indptr;
  const int32_t* indices = g->r_indices;
  HubCache hc;
  Transient tmp{ctx, {}};
  auto talloc = [&](size_t bytes, void** p) -> s3grl_status {
    S3GRL_TRY(ctx->arena.alloc(std::max<size_t>(bytes, 16), p));
    tmp.ptrs.push_back(*p);
    return S3GRL_OK;
  };
  void *flag_v, *off_v, *ws_v, *q;
  S3GRL_TRY(talloc((size_t)N * 4, &flag_v));
  S3GRL_TRY(talloc((size_t)(N + 1) * 8, &off_v));
  S3GRL_TRY(talloc((size_t)scan_workspace_elems(N) * 8, &ws_v));
  int32_t* flag = static_cast<int32_t*>(flag_v);
  int64_t* off = static_cast<int64_t*>(off_v);
  S3GRL_TRY(ctx->arena.alloc((size_t)N * 4, &q));
  hc.slot = static_cast<int32_t*>(q);
  S3GRL_TRY(ctx->arena.alloc((size_t)N * 4, &q));
  hc.voln = static_cast<int32_t*>(q);
  g->hub = hc;   // (released with the graph from here on, also on an error below)
  hipLaunchKernelGGL(hub_mark_kernel, dim3((unsigned)((N * 8 + 255) / 256)), dim3(256), 0, ctx->stream, indptr,
                     indices, N, min_deg, flag, hc.voln);
  S3GRL_HIP_TRY(hipGetLastError());
  S3GRL_TRY(launch_scan_i32_to_i64(ctx, flag, N, off, static_cast<int64_t*>(ws_v)));
  S3GRL_HIP_TRY(hipMemcpyAsync(ctx->h_scalars, off + N, 8, hipMemcpyDeviceToHost, ctx->stream));
  S3GRL_HIP_TRY(hipStreamSynchronize(ctx->stream));
  const int64_t nh = ctx->h_scalars[0];
  if (nh == 0 || nh >= (int64_t)1 << 24) {
    S3GRL_HIP_TRY(hipMemsetAsync(hc.slot, 0xff, (size_t)N * 4, ctx->stream));
    S3GRL_HIP_TRY(hipStreamSynchronize(ctx->stream));
    return S3GRL_OK;
  }
  void *hubs_v, *rows_v, *ws2_v;
  S3GRL_TRY(talloc((size_t)nh * 4, &hubs_v));
  S3GRL_TRY(talloc((size_t)nh * 4, &rows_v));
  S3GRL_TRY(talloc((size_t)scan_workspace_elems(nh) * 8, &ws2_v));
  S3GRL_HIP_TRY(hipMemsetAsync(ctx->d_scalars, 0, 8, ctx->stream));
  hipLaunchKernelGGL(hub_list_kernel, dim3((unsigned)((N + 255) / 256)), dim3(256), 0, ctx->stream, indptr, flag, off,
                     N, hc.slot, static_cast<int32_t*>(hubs_v), static_cast<int32_t*>(rows_v), hc.voln,
                     reinterpret_cast<unsigned long long*>(ctx->d_scalars));
  S3GRL_HIP_TRY(hipGetLastError());
  S3GRL_HIP_TRY(hipMemcpyAsync(ctx->h_scalars + 1, ctx->d_scalars, 8, hipMemcpyDeviceToHost, ctx->stream));
  S3GRL_TRY(ctx->arena.alloc((size_t)(nh + 1) * 8, &q));
  g->hub.row_base = hc.row_base = static_cast<int64_t*>(q);
  S3GRL_TRY(ctx->arena.alloc((size_t)(nh + 1) * 8, &q));
  g->hub.col_base = hc.col_base = static_cast<int64_t*>(q);
  S3GRL_TRY(launch_scan_i32_to_i64(ctx, static_cast<int32_t*>(rows_v), nh, hc.row_base, static_cast<int64_t*>(ws2_v)));
  S3GRL_HIP_TRY(hipMemcpyAsync(ctx->h_scalars, hc.row_base + nh, 8, hipMemcpyDeviceToHost, ctx->stream));
  S3GRL_HIP_TRY(hipStreamSynchronize(ctx->stream));
  const int64_t rows_total = ctx->h_scalars[0];
  // Σ over the hubs of Σ degree over their neighbours = the neighbour tests of the build (twice: count,
  // fill).  A graph whose hubs are not rare (mean degree in the hundreds) would spend seconds here, and
  // its hub links are no cheaper from a cache of dense neighbourhoods: no cache beyond 2^30 tests.
  if (ctx->h_scalars[1] > ((int64_t)1 << 30) && !getenv("S3GRL_HUB_MIN_DEG")) {
    S3GRL_HIP_TRY(hipMemsetAsync(hc.slot, 0xff, (size_t)N * 4, ctx->stream));
    S3GRL_HIP_TRY(hipStreamSynchronize(ctx->stream));
    return S3GRL_OK;
  }
  void *cnt_v, *abs_v, *ws3_v;
  S3GRL_TRY(talloc((size_t)rows_total * 4, &cnt_v));
  S3GRL_TRY(talloc((size_t)(rows_total + 1) * 8, &abs_v));
  S3GRL_TRY(talloc((size_t)scan_workspace_elems(rows_total) * 8, &ws3_v));
  const unsigned rgrid = (unsigned)((rows_total + 3) / 4);
  hipLaunchKernelGGL(hub_rows_kernel<false>, dim3(rgrid), dim3(256), 0, ctx->stream, indptr, indices,
                     static_cast<const int32_t*>(hubs_v), hc.row_base, (int)nh, rows_total,
                     static_cast<int32_t*>(cnt_v), static_cast<const int64_t*>(nullptr),
                     static_cast<uint16_t*>(nullptr));
  S3GRL_HIP_TRY(hipGetLastError());
  int64_t* abs_off = static_cast<int64_t*>(abs_v);
  S3GRL_TRY(launch_scan_i32_to_i64(ctx, static_cast<int32_t*>(cnt_v), rows_total, abs_off,
                                   static_cast<int64_t*>(ws3_v)));
  S3GRL_HIP_TRY(hipMemcpyAsync(ctx->h_scalars, abs_off + rows_total, 8, hipMemcpyDeviceToHost, ctx->stream));
  S3GRL_HIP_TRY(hipStreamSynchronize(ctx->stream));
  const int64_t entries = ctx->h_scalars[0];
  if (entries * 2 > ((int64_t)1 << 30)) {   // dense hubs: not worth a cache (see above)
    S3GRL_HIP_TRY(hipMemsetAsync(hc.slot, 0xff, (size_t)N * 4, ctx->stream));
    S3GRL_HIP_TRY(hipStreamSynchronize(ctx->stream));
    return S3GRL_OK;
  }
  S3GRL_TRY(ctx->arena.alloc((size_t)rows_total * 4, &q));
  g->hub.hoff = hc.hoff = static_cast<int32_t*>(q);
  S3GRL_TRY(ctx->arena.alloc((size_t)std::max<int64_t>(entries, 8) * 2, &q));
  g->hub.hcols = hc.hcols = static_cast<uint16_t*>(q);
  hipLaunchKernelGGL(hub_offsets_kernel, dim3((unsigned)((rows_total + 255) / 256)), dim3(256), 0, ctx->stream,
                     hc.row_base, (int)nh, rows_total, abs_off, hc.hoff, hc.col_base);
  hipLaunchKernelGGL(hub_rows_kernel<true>, dim3(rgrid), dim3(256), 0, ctx->stream, indptr, indices,
                     static_cast<const int32_t*>(hubs_v), hc.row_base, (int)nh, rows_total,
                     static_cast<int32_t*>(nullptr), abs_off, hc.hcols);
  S3GRL_HIP_TRY(hipGetLastError());
  S3GRL_HIP_TRY(hipStreamSynchronize(ctx->stream));   // tmp is released on return
  g->hub.nh = (int32_t)nh;
  if (getenv("S3GRL_DEBUG"))
    fprintf(stderr, "[s3grl] hub cache: %lld hubs (deg >= %d), %lld rows, %lld entries\n", (long long)nh, min_deg,
            (long long)rows_total, (long long)entries);
  return S3GRL_OK;
}

}  // namespace s3grl

S3GRL_DEFINE_TOUCH(hub)
