// Structure kernels of the PoS / PoS Plus path (feature independent), gfx950.
//
//   count_kernel      BFS to num_hops from {src,dst} on the unmasked graph      -> n, vol(S)
//   build_kernel      same BFS, local ids = rank in ascending global id, masked
//                     induced sub-CSR in local ids, D^-1/2, common neighbours
//   make_jobs_kernel  row pairs of every link
//   propagate_kernel  rows {a,b} of Â^1..Â^K by K pull steps r_i = r_{i-1}·Â in LDS,
//                     compacted to (node id, 2K coefficients) lists
//
// Restates (not translates) reference utils.py:47-85 (k_hop_subgraph), utils.py:33-44
// (neighbors) and tuned_SIGN.py:151-175 / :206-240: the reference materialises Â², …, Â^K of
// the whole n×n subgraph by SpGEMM and keeps R rows; here only those R rows are ever formed.
//
// One 256-thread workgroup owns one link.  Visited / frontier sets are N-bit bitmaps in LDS
// (N/8 bytes each: 2.4 KB for PubMed), so membership tests and the global->local map
// (rank = popcount prefix) never leave the CU and need no hashing.
#include "s3grl_internal.hpp"

namespace s3grl {
namespace {

__device__ __forceinline__ int lane_id() { return threadIdx.x & 63; }
__device__ __forceinline__ int wave_id() { return threadIdx.x >> 6; }

// Exclusive scan of one int per thread over the 256-thread block; `sh` holds >= 4 ints.
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
  for (int i = 0; i < kBlock / 64; ++i) {
    int s = sh[i];
    if (i < wid) woff += s;
    tot += s;
  }
  __syncthreads();
  total = tot;
  return woff + inc - v;
}

__device__ __forceinline__ int block_sum(int v, int* sh) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
  if (lane_id() == 0) sh[wave_id()] = v;
  __syncthreads();
  int tot = 0;
#pragma unroll
  for (int i = 0; i < kBlock / 64; ++i) tot += sh[i];
  __syncthreads();
  return tot;
}

__device__ __forceinline__ bool test_bit(const uint32_t* bm, int v) {
  return (bm[v >> 5] >> (v & 31)) & 1u;
}

// Level-synchronous BFS from {src,dst}, depth <= hops, on LDS bitmaps (reference
// utils.py:53-74: `fringe = neighbors(fringe, A) - visited`, early break on an empty fringe).
// On return `vis` holds S.  `on_level(d, cur)` is called by every thread after level d is
// complete, with `cur` = the nodes first reached at distance d.
template <typename LevelFn>
__device__ __forceinline__ void bfs_bitmaps(const int32_t* __restrict__ indptr,
                                            const int32_t* __restrict__ indices, int W, int src,
                                            int dst, int hops, uint32_t* vis, uint32_t* cur,
                                            uint32_t* nxt, LevelFn on_level) {
  const int tid = threadIdx.x;
  for (int t = tid; t < W; t += kBlock) {
    vis[t] = 0;
    cur[t] = 0;
    nxt[t] = 0;
  }
  __syncthreads();
  if (tid == 0) {
    atomicOr(&vis[src >> 5], 1u << (src & 31));
    atomicOr(&vis[dst >> 5], 1u << (dst & 31));
    atomicOr(&cur[src >> 5], 1u << (src & 31));
    atomicOr(&cur[dst >> 5], 1u << (dst & 31));
  }
  __syncthreads();
  on_level(0, cur);
  for (int d = 1; d <= hops; ++d) {
    for (int t = tid; t < W; t += kBlock) {
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
    __syncthreads();
    int any = 0;
    for (int t = tid; t < W; t += kBlock) {
      const uint32_t c = nxt[t];
      cur[t] = c;
      nxt[t] = 0;
      any |= (c != 0);
    }
    any = __syncthreads_or(any);
    if (!any) break;
    on_level(d, cur);
  }
}

struct NoLevel {
  __device__ void operator()(int, const uint32_t*) const {}
};

// ---------------------------------------------------------------------------------------
__global__ __launch_bounds__(kBlock) void count_kernel(
    const int32_t* __restrict__ indptr, const int32_t* __restrict__ indices, int N, int W,
    const int64_t* __restrict__ links, int hops, int plus, int32_t* __restrict__ n_nodes,
    int32_t* __restrict__ vol, int32_t* __restrict__ cn_cap, int32_t* __restrict__ err_flag) {
  extern __shared__ uint32_t smem[];
  uint32_t* vis = smem;
  uint32_t* cur = smem + W;
  uint32_t* nxt = smem + 2 * W;
  int* sh = reinterpret_cast<int*>(smem + 3 * W);
  const int l = blockIdx.x;
  const int64_t s64 = links[2 * (int64_t)l], d64 = links[2 * (int64_t)l + 1];
  if (s64 < 0 || s64 >= N || d64 < 0 || d64 >= N || s64 == d64) {
    if (threadIdx.x == 0) {
      atomicMax(err_flag, s64 == d64 ? 2 : 1);
      n_nodes[l] = 0;
      vol[l] = 0;
      cn_cap[l] = 0;
    }
    return;
  }
  const int src = (int)s64, dst = (int)d64;
  bfs_bitmaps(indptr, indices, W, src, dst, hops, vis, cur, nxt, NoLevel{});
  int n = 0, dv = 0;
  for (int t = threadIdx.x; t < W; t += kBlock) {
    uint32_t w = vis[t];
    n += __popc(w);
    while (w) {
      const int b = __ffs(w) - 1;
      w &= w - 1;
      const int v = t * 32 + b;
      dv += indptr[v + 1] - indptr[v];
    }
  }
  n = block_sum(n, sh);
  dv = block_sum(dv, sh);
  if (threadIdx.x == 0) {
    n_nodes[l] = n;
    vol[l] = dv;
    const int ds = indptr[src + 1] - indptr[src], dd = indptr[dst + 1] - indptr[dst];
    cn_cap[l] = plus ? min(ds, dd) + 2 : 0;
  }
}

// Single-workgroup exclusive scan int32[n] -> int64[n+1] (n up to a few million: each of the
// 1024 threads walks a contiguous chunk).
__global__ __launch_bounds__(1024) void scan_kernel(const int32_t* __restrict__ in, int64_t n,
                                                    int64_t* __restrict__ out) {
  __shared__ int64_t part[1024];
  const int tid = threadIdx.x;
  const int64_t chunk = (n + 1023) / 1024;
  const int64_t b = tid * chunk, e = min(n, b + chunk);
  int64_t s = 0;
  for (int64_t i = b; i < e; ++i) s += in[i];
  part[tid] = s;
  __syncthreads();
  for (int o = 1; o < 1024; o <<= 1) {
    int64_t t = tid >= o ? part[tid - o] : 0;
    __syncthreads();
    part[tid] += t;
    __syncthreads();
  }
  int64_t run = tid ? part[tid - 1] : 0;
  for (int64_t i = b; i < e; ++i) {
    out[i] = run;
    run += in[i];
  }
  if (tid == 1023) out[n] = part[1023];
}

__device__ __forceinline__ int rank_of(const uint32_t* vis, const uint32_t* wpre, int v) {
  return (int)wpre[v >> 5] + __popc(vis[v >> 5] & ((1u << (v & 31)) - 1u));
}

// lower_bound membership in an ascending int list
__device__ __forceinline__ bool sorted_contains(const int32_t* a, int n, int x) {
  int lo = 0, hi = n;
  while (lo < hi) {
    const int mid = (lo + hi) >> 1;
    if (a[mid] < x) lo = mid + 1; else hi = mid;
  }
  return lo < n && a[lo] == x;
}

// ---------------------------------------------------------------------------------------
__global__ __launch_bounds__(kBlock) void build_kernel(
    const int32_t* __restrict__ indptr, const int32_t* __restrict__ indices, int N, int W,
    const int64_t* __restrict__ links, int hops, int plus, const int64_t* __restrict__ node_off,
    const int64_t* __restrict__ edge_off, const int64_t* __restrict__ cn_off,
    int32_t* __restrict__ nodes, int32_t* __restrict__ rowstart, int32_t* __restrict__ cnt,
    float* __restrict__ dinv, int32_t* __restrict__ lcsr, int32_t* __restrict__ cn_tmp,
    int32_t* __restrict__ cn_count, int32_t* __restrict__ n_rows, int32_t* __restrict__ n_jobs,
    unsigned long long* __restrict__ tot_edges) {
  extern __shared__ uint32_t smem[];
  uint32_t* vis = smem;
  uint32_t* cur = smem + W;
  uint32_t* nxt = smem + 2 * W;
  uint32_t* wpre = smem + 3 * W;
  int* sh = reinterpret_cast<int*>(smem + 4 * W);
  const int tid = threadIdx.x;
  const int l = blockIdx.x;
  const int src = (int)links[2 * (int64_t)l], dst = (int)links[2 * (int64_t)l + 1];
  const int64_t noff = node_off[l], eoff = edge_off[l];
  const int n_expected = (int)(node_off[l + 1] - noff);
  if (n_expected == 0) {  // invalid link, flagged by count_kernel
    if (tid == 0) {
      cn_count[l] = 0;
      n_rows[l] = 0;
      n_jobs[l] = 0;
    }
    return;
  }
  bfs_bitmaps(indptr, indices, W, src, dst, hops, vis, cur, nxt, NoLevel{});

  // local id = rank of the node in ascending global id: word-level popcount prefix
  int carry = 0;
  for (int base = 0; base < W; base += kBlock) {
    const int t = base + tid;
    const int pc = t < W ? __popc(vis[t]) : 0;
    int total;
    const int ex = block_excl_scan(pc, sh, total);
    if (t < W) wpre[t] = carry + ex;
    carry += total;
  }
  const int n = carry;
  __syncthreads();

  // node list + global degrees (row capacities of the local CSR)
  for (int t = tid; t < W; t += kBlock) {
    uint32_t w = vis[t];
    int64_t o = noff + wpre[t];
    while (w) {
      const int b = __ffs(w) - 1;
      w &= w - 1;
      const int v = t * 32 + b;
      nodes[o] = v;
      rowstart[o] = indptr[v + 1] - indptr[v];
      ++o;
    }
  }
  __syncthreads();
  carry = 0;
  for (int base = 0; base < n; base += kBlock) {
    const int i = base + tid;
    const int v = i < n ? rowstart[noff + i] : 0;
    int total;
    const int ex = block_excl_scan(v, sh, total);
    if (i < n) rowstart[noff + i] = carry + ex;
    carry += total;
  }
  __syncthreads();

  // masked induced sub-CSR in local ids: 8 lanes per row, order of the global row preserved
  const int g = tid & 7;
  const int grp_in_wave = (tid & 63) >> 3;
  int edges_local = 0;
  for (int base = 0; base < n; base += kBlock / 8) {
    const int a = base + (tid >> 3);
    if (a < n) {
      const int v = nodes[noff + a];
      const int s = indptr[v], e = indptr[v + 1];
      const int64_t out = eoff + rowstart[noff + a];
      const bool is_src = v == src, is_dst = v == dst;
      int count = 0;
      for (int c0 = s; c0 < e; c0 += 8) {
        const int c = c0 + g;
        int u = -1;
        bool in = false;
        if (c < e) {
          u = indices[c];
          in = test_bit(vis, u) && !((is_src && u == dst) || (is_dst && u == src));
        }
        const unsigned long long bal = __ballot(in);
        const uint32_t gb = (uint32_t)(bal >> (8 * grp_in_wave)) & 0xffu;
        if (in) lcsr[out + count + __popc(gb & ((1u << g) - 1u))] = rank_of(vis, wpre, u);
        count += __popc(gb);
      }
      if (g == 0) {
        cnt[noff + a] = count;
        dinv[noff + a] = count > 0 ? 1.0f / sqrtf((float)count) : 0.0f;  // inf -> 0
        edges_local += count;
      }
    }
  }
  edges_local = block_sum(edges_local, sh);  // also the barrier that publishes lcsr/cnt
  if (tid == 0) atomicAdd(tot_edges, (unsigned long long)edges_local);

  if (!plus) {
    if (tid == 0) {
      cn_count[l] = 0;
      n_rows[l] = 2;
      n_jobs[l] = 1;
    }
    return;
  }
  // PoS Plus row selection, reference tuned_SIGN.py:233 on the MASKED sub-CSR:
  //   N'(0) = stored columns of row 0 = N_S(src) \ {dst} ∪ {1}   (explicit zero at [0,1])
  //   N'(1) = N_S(dst) \ {src} ∪ {0};   CN = N'(0) ∩ N'(1)  (SURVEY §8c K2/K4).
  // For x ∉ {0,1}: x ∈ N_S(src) ∩ N_S(dst).  Local 0 (src) is in CN iff src has a self-loop;
  // local 1 (dst) iff dst has one.  Emitted in ascending global id (= ascending rank).
  if (wave_id() == 0) {
    const int lane = lane_id();
    const int srcl = rank_of(vis, wpre, src), dstl = rank_of(vis, wpre, dst);
    const int32_t* row_s = lcsr + eoff + rowstart[noff + srcl];
    const int32_t* row_d = lcsr + eoff + rowstart[noff + dstl];
    const int cs = cnt[noff + srcl], cd = cnt[noff + dstl];
    const bool loop_d = sorted_contains(row_d, cd, dstl);
    const int64_t co = cn_off[l];
    int total = 0, lt_dstl = 0;
    for (int c0 = 0; c0 < cs; c0 += 64) {
      const int c = c0 + lane;
      int x = -1;
      bool sel = false;
      if (c < cs) {
        x = row_s[c];
        sel = (x == srcl) || sorted_contains(row_d, cd, x);
      }
      const unsigned long long bal = __ballot(sel);
      const unsigned long long below = __ballot(sel && x < dstl);
      const int before = __popcll(bal & ((1ull << lane) - 1ull));
      if (sel) cn_tmp[co + total + before + ((loop_d && dstl < x) ? 1 : 0)] = x;
      total += __popcll(bal);
      lt_dstl += __popcll(below);
    }
    if (loop_d) {
      if (lane == 0) cn_tmp[co + lt_dstl] = dstl;
      total += 1;
    }
    if (lane == 0) {
      cn_count[l] = total;
      n_rows[l] = 2 + total;
      n_jobs[l] = (2 + total + 1) / 2;
    }
  }
}

// Hop distances for the parity hook: BFS again, rank by binary search in the stored node list.
__global__ __launch_bounds__(kBlock) void dists_kernel(
    const int32_t* __restrict__ indptr, const int32_t* __restrict__ indices, int N, int W,
    const int64_t* __restrict__ links, int hops, const int64_t* __restrict__ node_off,
    const int32_t* __restrict__ nodes, int8_t* __restrict__ dists) {
  extern __shared__ uint32_t smem[];
  uint32_t* vis = smem;
  uint32_t* cur = smem + W;
  uint32_t* nxt = smem + 2 * W;
  const int l = blockIdx.x;
  const int64_t noff = node_off[l];
  const int n = (int)(node_off[l + 1] - noff);
  if (n == 0) return;
  const int src = (int)links[2 * (int64_t)l], dst = (int)links[2 * (int64_t)l + 1];
  const int32_t* my = nodes + noff;
  auto record = [&](int d, const uint32_t* level) {
    for (int t = threadIdx.x; t < W; t += kBlock) {
      uint32_t w = level[t];
      while (w) {
        const int b = __ffs(w) - 1;
        w &= w - 1;
        const int v = t * 32 + b;
        int lo = 0, hi = n;
        while (lo < hi) {
          const int mid = (lo + hi) >> 1;
          if (my[mid] < v) lo = mid + 1; else hi = mid;
        }
        dists[noff + lo] = (int8_t)d;
      }
    }
  };
  bfs_bitmaps(indptr, indices, W, src, dst, hops, vis, cur, nxt, record);
}

// ---------------------------------------------------------------------------------------
__global__ __launch_bounds__(kBlock) void make_jobs_kernel(
    const int64_t* __restrict__ links, int64_t L, const int64_t* __restrict__ node_off,
    const int64_t* __restrict__ row_ptr, const int64_t* __restrict__ job_off,
    const int64_t* __restrict__ cn_off, const int32_t* __restrict__ cn_tmp,
    const int32_t* __restrict__ nodes, Job* __restrict__ jobs, int64_t* __restrict__ row_nodes,
    int32_t* __restrict__ job_n) {
  const int64_t l = (int64_t)blockIdx.x * kBlock + threadIdx.x;
  if (l >= L) return;
  const int64_t noff = node_off[l];
  const int n = (int)(node_off[l + 1] - noff);
  const int R = (int)(row_ptr[l + 1] - row_ptr[l]);
  if (n == 0 || R == 0) return;
  const int src = (int)links[2 * l], dst = (int)links[2 * l + 1];
  const int32_t* my = nodes + noff;
  auto rank = [&](int v) {
    int lo = 0, hi = n;
    while (lo < hi) {
      const int mid = (lo + hi) >> 1;
      if (my[mid] < v) lo = mid + 1; else hi = mid;
    }
    return lo;
  };
  const int srcl = rank(src), dstl = rank(dst);
  const int64_t co = cn_off ? cn_off[l] : 0;
  auto row_local = [&](int r) { return r == 0 ? srcl : (r == 1 ? dstl : cn_tmp[co + r - 2]); };
  const int64_t rp = row_ptr[l];
  for (int r = 0; r < R; ++r) row_nodes[rp + r] = my[row_local(r)];
  const int64_t j0 = job_off[l];
  const int nj = (R + 1) / 2;
  for (int p = 0; p < nj; ++p) {
    Job j;
    j.coef_off = 0;
    j.out_row = rp + 2 * p;
    j.link = (int32_t)l;
    j.support = 0;
    j.local_a = row_local(2 * p);
    j.node_a = my[j.local_a];
    j.z_a = (j.node_a == src || j.node_a == dst) ? 1 : 0;
    if (2 * p + 1 < R) {
      j.local_b = row_local(2 * p + 1);
      j.node_b = my[j.local_b];
      j.z_b = (j.node_b == src || j.node_b == dst) ? 1 : 0;
    } else {
      j.local_b = -1;
      j.node_b = -1;
      j.z_b = 0;
    }
    jobs[j0 + p] = j;
    job_n[j0 + p] = n;
  }
}

// ---------------------------------------------------------------------------------------
// K pull steps in LDS.  State s_i[w] = dinv[w]·r_i[w] (float2: rows a and b), so that
//   r_i[w] = dinv[w] · Σ_{v ∈ N_S(w)} s_{i-1}[v]          (Â symmetric: pull == r_{i-1}·Â)
// Every r_i[w] is a sum over w's row in stored order, reduced over 8 lanes by a fixed xor
// tree: bit-reproducible run to run.  All terms are >= 0: no cancellation.
template <int K>
__global__ __launch_bounds__(kBlock) void propagate_kernel(
    Job* __restrict__ jobs, const int64_t* __restrict__ coef_off,
    const int64_t* __restrict__ links, const int64_t* __restrict__ node_off,
    const int64_t* __restrict__ edge_off, const int32_t* __restrict__ nodes,
    const int32_t* __restrict__ rowstart, const int32_t* __restrict__ cnt,
    const float* __restrict__ dinv, const int32_t* __restrict__ lcsr, int n_lo, int n_hi,
    int32_t* __restrict__ c_ids, float* __restrict__ c_coef,
    float* __restrict__ job_z, unsigned long long* __restrict__ tot_support) {
  extern __shared__ float2 state[];
  const int tid = threadIdx.x;
  const int64_t jid = blockIdx.x;
  const Job job = jobs[jid];
  const int l = job.link;
  const int64_t noff = node_off[l], eoff = edge_off[l];
  const int n = (int)(node_off[l + 1] - noff);
  if (n <= n_lo || n > n_hi) return;
  float2* cur = state;
  float2* nxt = state + n_hi;
  int* sh = reinterpret_cast<int*>(state + 2 * (size_t)n_hi);
  const int64_t coff = coef_off[jid];
  float* dense = c_coef + coff * (2 * K);  // [n][K][2], compacted in place below
  const int la = job.local_a, lb = job.local_b;

  for (int w = tid; w < n; w += kBlock) cur[w] = make_float2(0.f, 0.f);
  __syncthreads();
  if (tid == 0) {
    cur[la].x = dinv[noff + la];
    if (lb >= 0) cur[lb].y = dinv[noff + lb];
  }
  __syncthreads();

  const int g = tid & 7;
  for (int i = 0; i < K; ++i) {
    for (int base = 0; base < n; base += kBlock / 8) {
      const int w = base + (tid >> 3);
      if (w < n) {
        const int c = cnt[noff + w];
        const int32_t* row = lcsr + eoff + rowstart[noff + w];
        float sx = 0.f, sy = 0.f;
        for (int e = g; e < c; e += 8) {
          const float2 t = cur[row[e]];
          sx += t.x;
          sy += t.y;
        }
#pragma unroll
        for (int o = 4; o > 0; o >>= 1) {
          sx += __shfl_xor(sx, o);
          sy += __shfl_xor(sy, o);
        }
        if (g == 0) {
          const float dw = dinv[noff + w];
          const float rx = dw * sx, ry = dw * sy;
          nxt[w] = make_float2(dw * rx, dw * ry);
          float2* d2 = reinterpret_cast<float2*>(dense + ((int64_t)w * K + i) * 2);
          *d2 = make_float2(rx, ry);
        }
      }
    }
    __syncthreads();
    float2* t = cur;
    cur = nxt;
    nxt = t;
  }

  // label column of operator i: Σ_w r_i[w] z_w = r_i[src] + r_i[dst]   (tuned_SIGN.py:177-185)
  if (tid < K) {
    const int src = (int)links[2 * (int64_t)l], dst = (int)links[2 * (int64_t)l + 1];
    const int32_t* my = nodes + noff;
    auto rank = [&](int v) {
      int lo = 0, hi = n;
      while (lo < hi) {
        const int mid = (lo + hi) >> 1;
        if (my[mid] < v) lo = mid + 1; else hi = mid;
      }
      return lo;
    };
    const int sl = rank(src), dl = rank(dst);
    const float2 rs = *reinterpret_cast<const float2*>(dense + ((int64_t)sl * K + tid) * 2);
    const float2 rd = *reinterpret_cast<const float2*>(dense + ((int64_t)dl * K + tid) * 2);
    job_z[(jid * K + tid) * 2 + 0] = rs.x + rd.x;
    job_z[(jid * K + tid) * 2 + 1] = rs.y + rd.y;
  }
  __syncthreads();

  // in-place compaction to the nodes with any non-zero coefficient (tile t only writes
  // positions <= its own, and reads its tile before the barrier inside the scan)
  int carry = 0;
  for (int base = 0; base < n; base += kBlock) {
    const int w = base + tid;
    float v[2 * K];
    bool act = false;
    int id = 0;
    if (w < n) {
#pragma unroll
      for (int q = 0; q < 2 * K; ++q) {
        v[q] = dense[(int64_t)w * 2 * K + q];
        act |= (v[q] != 0.f);
      }
      id = nodes[noff + w];
    }
    int total;
    const int ex = block_excl_scan(act ? 1 : 0, sh, total);
    if (act) {
      const int64_t p = carry + ex;
#pragma unroll
      for (int q = 0; q < 2 * K; ++q) dense[p * 2 * K + q] = v[q];
      c_ids[coff + p] = id;
    }
    carry += total;
    __syncthreads();
  }
  if (tid == 0) {
    Job o = job;
    o.coef_off = coff;
    o.support = carry;
    jobs[jid] = o;
    atomicAdd(tot_support, (unsigned long long)carry);
  }
}

}  // namespace

// ---------------------------------------------------------------------------------------
static inline int words_for(int64_t N) { return (int)((N + 31) / 32); }

s3grl_status launch_count(s3grl_context* ctx, const s3grl_graph* g, const int64_t* links, int64_t L,
                          int hops, int plus, int32_t* n_nodes, int32_t* vol, int32_t* cn_cap,
                          int32_t* err_flag) {
  if (L == 0) return S3GRL_OK;
  const int W = words_for(g->num_nodes);
  const size_t lds = (size_t)(3 * W + 8) * 4;
  S3GRL_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(count_kernel),
                                    hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  hipLaunchKernelGGL(count_kernel, dim3((unsigned)L), dim3(kBlock), lds, ctx->stream, g->indptr,
                     g->indices, (int)g->num_nodes, W, links, hops, plus, n_nodes, vol, cn_cap,
                     err_flag);
  S3GRL_HIP_TRY(hipGetLastError());
  return S3GRL_OK;
}

s3grl_status launch_scan_i32_to_i64(s3grl_context* ctx, const int32_t* in, int64_t n, int64_t* out) {
  hipLaunchKernelGGL(scan_kernel, dim3(1), dim3(1024), 0, ctx->stream, in, n, out);
  S3GRL_HIP_TRY(hipGetLastError());
  return S3GRL_OK;
}

s3grl_status launch_build(s3grl_context* ctx, const s3grl_graph* g, const int64_t* links, int64_t L,
                          int hops, int plus, const int64_t* node_off, const int64_t* edge_off,
                          const int64_t* cn_off, int32_t* nodes, int8_t* dists, int32_t* rowstart,
                          int32_t* cnt, float* dinv, int32_t* lcsr, int32_t* cn_tmp,
                          int32_t* cn_count, int32_t* n_rows, int32_t* n_jobs, int64_t* tot_edges) {
  if (L == 0) return S3GRL_OK;
  const int W = words_for(g->num_nodes);
  const size_t lds = (size_t)(4 * W + 8) * 4;
  S3GRL_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(build_kernel),
                                    hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  hipLaunchKernelGGL(build_kernel, dim3((unsigned)L), dim3(kBlock), lds, ctx->stream, g->indptr,
                     g->indices, (int)g->num_nodes, W, links, hops, plus, node_off, edge_off, cn_off,
                     nodes, rowstart, cnt, dinv, lcsr, cn_tmp, cn_count, n_rows, n_jobs,
                     reinterpret_cast<unsigned long long*>(tot_edges));
  S3GRL_HIP_TRY(hipGetLastError());
  if (dists) {
    const size_t lds3 = (size_t)(3 * W + 8) * 4;
    S3GRL_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(dists_kernel),
                                      hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds3));
    hipLaunchKernelGGL(dists_kernel, dim3((unsigned)L), dim3(kBlock), lds3, ctx->stream, g->indptr,
                       g->indices, (int)g->num_nodes, W, links, hops, node_off, nodes, dists);
    S3GRL_HIP_TRY(hipGetLastError());
  }
  return S3GRL_OK;
}

s3grl_status launch_make_jobs(s3grl_context* ctx, const int64_t* links, int64_t L,
                              const int64_t* node_off, const int64_t* row_ptr,
                              const int64_t* job_off, const int64_t* cn_off, const int32_t* cn_tmp,
                              const int32_t* nodes, const int32_t* n_nodes, int K, Job* jobs,
                              int64_t* row_nodes, int32_t* job_n) {
  (void)n_nodes;
  (void)K;
  if (L == 0) return S3GRL_OK;
  const unsigned grid = (unsigned)((L + kBlock - 1) / kBlock);
  hipLaunchKernelGGL(make_jobs_kernel, dim3(grid), dim3(kBlock), 0, ctx->stream, links, L, node_off,
                     row_ptr, job_off, cn_off, cn_tmp, nodes, jobs, row_nodes, job_n);
  S3GRL_HIP_TRY(hipGetLastError());
  return S3GRL_OK;
}

template <int K>
static s3grl_status launch_propagate_k(s3grl_context* ctx, Job* jobs, int64_t njobs,
                                       const int64_t* coef_off, const int64_t* links,
                                       const int64_t* node_off, const int64_t* edge_off,
                                       const int32_t* nodes, const int32_t* rowstart,
                                       const int32_t* cnt, const float* dinv, const int32_t* lcsr,
                                       int64_t max_nodes, int32_t* c_ids, float* c_coef,
                                       float* job_z, int64_t* tot_support) {
  // Size classes: LDS = 16 B per subgraph node (two float2 state arrays).  Every class is one
  // launch over all jobs; workgroups whose subgraph is outside the class exit at once.
  static const int bounds[] = {0, 1024, 2560, 10112};
  auto kern = propagate_kernel<K>;
  S3GRL_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(kern),
                                    hipFuncAttributeMaxDynamicSharedMemorySize, 163840));
  for (int c = 0; c < 3; ++c) {
    if (max_nodes <= bounds[c]) break;
    const int hi = bounds[c + 1];
    const size_t lds = (size_t)hi * 16 + 64;
    hipLaunchKernelGGL(kern, dim3((unsigned)njobs), dim3(kBlock), lds, ctx->stream, jobs, coef_off,
                       links, node_off, edge_off, nodes, rowstart, cnt, dinv, lcsr, bounds[c], hi,
                       c_ids, c_coef, job_z, reinterpret_cast<unsigned long long*>(tot_support));
    S3GRL_HIP_TRY(hipGetLastError());
  }
  return S3GRL_OK;
}

s3grl_status launch_propagate(s3grl_context* ctx, Job* jobs, int64_t njobs, const int64_t* coef_off,
                              const int64_t* links, const int64_t* node_off,
                              const int64_t* edge_off, const int32_t* nodes,
                              const int32_t* rowstart, const int32_t* cnt, const float* dinv,
                              const int32_t* lcsr, int K, int64_t max_nodes, int32_t* c_ids,
                              float* c_coef, float* job_z, int64_t* tot_support) {
  if (njobs == 0) return S3GRL_OK;
  if (max_nodes > 10112) {
    set_last_error("subgraph with " + std::to_string(max_nodes) +
                   " nodes exceeds the LDS propagation limit (10112) of this build");
    return S3GRL_ERR_GRAPH_TOO_LARGE;
  }
#define S3GRL_PROP_CASE(KK)                                                                       \
  case KK:                                                                                        \
    return launch_propagate_k<KK>(ctx, jobs, njobs, coef_off, links, node_off, edge_off, nodes,   \
                                  rowstart, cnt, dinv, lcsr, max_nodes, c_ids, c_coef, job_z,     \
                                  tot_support)
  switch (K) {
    S3GRL_PROP_CASE(1);
    S3GRL_PROP_CASE(2);
    S3GRL_PROP_CASE(3);
    S3GRL_PROP_CASE(4);
    S3GRL_PROP_CASE(5);
    S3GRL_PROP_CASE(6);
    S3GRL_PROP_CASE(7);
    S3GRL_PROP_CASE(8);
    default:
      set_last_error("sign_k must be in 1..8");
      return S3GRL_ERR_INVALID_ARGUMENT;
  }
#undef S3GRL_PROP_CASE
}

}  // namespace s3grl
