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

// PoS Plus row selection, reference tuned_SIGN.py:233 evaluated on the MASKED sub-CSR whose
// `.indices` still hold the explicit zeros of the masking:
//   N'(0) = (N_G(src) ∩ S) \ {dst} ∪ {dst-as-explicit-zero};  N'(1) likewise;  CN = N'(0) ∩ N'(1).
// For x ∉ {src,dst}: x ∈ N(src) ∩ N(dst) ∩ S.  src itself is selected iff src has a self-loop,
// dst iff dst has one.  One wave walks row(src) (ascending) and emits global ids in ascending
// order into out[] (may be null: count only).  Returns |CN|.
__device__ __forceinline__ int common_neighbours(const int32_t* __restrict__ indptr,
                                                 const int32_t* __restrict__ indices,
                                                 const uint32_t* vis, int src, int dst,
                                                 int32_t* out) {
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
      sel = x != dst && test_bit(vis, x) && (x == src || sorted_contains(row_d, cd, x));
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

// ---------------------------------------------------------------------------------------
// count: level-synchronous BFS on LDS bitmaps, one thread per frontier word (reference
// utils.py:53-74: `fringe = neighbors(fringe, A) - visited`, early break on an empty fringe).
__global__ __launch_bounds__(kBlock) void count_kernel(
    const int32_t* __restrict__ indptr, const int32_t* __restrict__ indices, int N, int W,
    const int64_t* __restrict__ links, int hops, int plus, int32_t* __restrict__ n_nodes,
    int32_t* __restrict__ n_rows, int32_t* __restrict__ n_jobs, int32_t* __restrict__ err_flag,
    unsigned long long* __restrict__ tot_vol) {
  extern __shared__ uint32_t smem[];
  uint32_t* vis = smem;
  uint32_t* cur = smem + W;
  uint32_t* nxt = smem + 2 * W;
  int* sh = reinterpret_cast<int*>(smem + 3 * W);
  const int tid = threadIdx.x;
  const int l = blockIdx.x;
  const int64_t s64 = links[2 * (int64_t)l], d64 = links[2 * (int64_t)l + 1];
  if (s64 < 0 || s64 >= N || d64 < 0 || d64 >= N || s64 == d64) {
    if (tid == 0) {
      atomicMax(err_flag, s64 == d64 ? 2 : 1);
      n_nodes[l] = 0;
      n_rows[l] = 0;
      n_jobs[l] = 0;
    }
    return;
  }
  const int src = (int)s64, dst = (int)d64;
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
    if (!__syncthreads_or(any)) break;
  }
  int n = 0, dv = 0;
  for (int t = tid; t < W; t += kBlock) {
    uint32_t w = vis[t];
    n += __popc(w);
    while (w) {
      const int b = __ffs(w) - 1;
      w &= w - 1;
      const int v = t * 32 + b;
      dv += indptr[v + 1] - indptr[v];
    }
  }
  n = block_sum<kBlock>(n, sh);
  dv = block_sum<kBlock>(dv, sh);
  int R = 2;
  if (plus && wave_id() == 0) R = 2 + common_neighbours(indptr, indices, vis, src, dst, nullptr);
  if (tid == 0) {
    n_nodes[l] = n;
    n_rows[l] = R;
    n_jobs[l] = (R + 1) / 2;
    atomicAdd(tot_vol, (unsigned long long)dv);
  }
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

// ---------------------------------------------------------------------------------------
__global__ void classify_kernel(const int32_t* __restrict__ n_nodes, int64_t L,
                                int32_t* __restrict__ class_count, int32_t* __restrict__ class_list) {
  const int64_t l = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (l >= L) return;
  const int n = n_nodes[l];
  if (n == 0) return;
  const int bound[kNumClasses] = S3GRL_CLASS_BOUNDS;
  int c = 0;
#pragma unroll
  for (int k = 0; k < kNumClasses; ++k) c += n > bound[k] ? 1 : 0;
  if (c == kNumClasses) {
    atomicAdd(&class_count[kNumClasses], 1);  // too large for LDS: reported by the host
    return;
  }
  const int slot = atomicAdd(&class_count[c], 1);
  class_list[(int64_t)c * L + slot] = (int32_t)l;
}

// ---------------------------------------------------------------------------------------
// The fused per-link kernel.  LDS layout (dynamic, 16-byte aligned base):
//   vis[W] nxt[W] wpre[W]            bitmaps / rank prefix             (uint32)
//   list[nmax]                       S in hop-major, ascending-id order (global ids)
//   dinv[nmax]                       D^-1/2 of the masked induced subgraph, by local rank
//   cur[nmax], nxs[nmax]             float2 propagation state s_i = dinv·r_i (rows a, b)
//   cn[cn_cap]                       common neighbours (global ids)
//   misc: lvl_end[kMaxLevels], z[2][K][2], scan scratch
template <int T, int K>
__global__ __launch_bounds__(T) void link_kernel(
    const int32_t* __restrict__ indptr, const int32_t* __restrict__ indices, int W,
    const int64_t* __restrict__ links, const int32_t* __restrict__ class_list, int hops, int plus,
    int nmax, int cn_cap, const int64_t* __restrict__ node_off, const int64_t* __restrict__ row_ptr,
    const int64_t* __restrict__ job_off, const int64_t* __restrict__ coef_off,
    int32_t* __restrict__ c_ids, float* __restrict__ c_coef, Job* __restrict__ jobs,
    float* __restrict__ job_z, int64_t* __restrict__ row_nodes, int32_t* __restrict__ lvl_out,
    unsigned long long* __restrict__ tot_edges, unsigned long long* __restrict__ tot_support) {
  extern __shared__ uint32_t smem[];
  uint32_t* vis = smem;
  uint32_t* nxt = smem + W;
  uint32_t* wpre = smem + 2 * W;
  int32_t* list = reinterpret_cast<int32_t*>(smem + 3 * W);
  float* dinv = reinterpret_cast<float*>(list + nmax);
  // float2 arrays need 8-byte alignment: 3W + 2 nmax words may be odd
  float2* cur = reinterpret_cast<float2*>(smem + ((3 * W + 2 * nmax + 1) & ~1));
  float2* nxs = cur + nmax;
  int32_t* cn = reinterpret_cast<int32_t*>(nxs + nmax);
  int* lvl_end = cn + cn_cap;
  float* zbuf = reinterpret_cast<float*>(lvl_end + kMaxLevels);  // [2 (src,dst)][K][2 (rows)]
  int* sh = reinterpret_cast<int*>(zbuf + 4 * K);

  const int tid = threadIdx.x;
  const int l = class_list[blockIdx.x];
  const int src = (int)links[2 * (int64_t)l], dst = (int)links[2 * (int64_t)l + 1];
  const int64_t noff = node_off[l];
  const int g = tid & 7;

  // ---- BFS: frontier = a segment of `list`, 8 lanes per frontier node -------------------
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
  __syncthreads();
  int n = 2, nlev = 1;  // levels 0..nlev-1 are complete
  for (int d = 1; d <= hops; ++d) {
    const int f0 = d >= 2 ? lvl_end[min(d, kMaxLevels) - 2] : 0, f1 = n;
    for (int base = f0; base < f1; base += T / 8) {
      const int t = base + (tid >> 3);
      if (t < f1) {
        const int v = list[t];
        const int e1 = indptr[v + 1];
        for (int c = indptr[v] + g; c < e1; c += 8) {
          const int u = indices[c];
          const uint32_t m = 1u << (u & 31);
          const uint32_t old = atomicOr(&vis[u >> 5], m);
          if (!(old & m)) atomicOr(&nxt[u >> 5], m);
        }
      }
    }
    __syncthreads();
    // append the new level in ascending id order (deterministic), clear nxt
    int added = 0;
    for (int base = 0; base < W; base += T) {
      const int t = base + tid;
      uint32_t w = t < W ? nxt[t] : 0u;
      int total;
      int pos = n + added + block_excl_scan<T>(__popc(w), sh, total);
      while (w) {
        const int b = __ffs(w) - 1;
        w &= w - 1;
        list[pos++] = t * 32 + b;
      }
      if (t < W) nxt[t] = 0;
      added += total;
    }
    if (added == 0) break;  // uniform: `added` is a block-wide total
    n += added;
    if (d < kMaxLevels) {
      if (tid == 0) lvl_end[d] = n;
      nlev = d + 1;
    } else if (tid == 0) {
      lvl_end[kMaxLevels - 1] = n;  // deeper levels are merged into the last one
    }
    __syncthreads();
  }
  __syncthreads();

  // ---- local ids: rank in ascending global id ------------------------------------------
  {
    int carry = 0;
    for (int base = 0; base < W; base += T) {
      const int t = base + tid;
      const int pc = t < W ? __popc(vis[t]) : 0;
      int total;
      const int ex = block_excl_scan<T>(pc, sh, total);
      if (t < W) wpre[t] = carry + ex;
      carry += total;
    }
  }
  __syncthreads();
  const int srcl = rank_of(vis, wpre, src), dstl = rank_of(vis, wpre, dst);

  // ---- degrees of the masked induced subgraph -> D^-1/2 (inf -> 0) ---------------------
  // reference tuned_SIGN.py:153-161: structure only, target link removed, no self-loops added
  int edges_local = 0;
  for (int base = 0; base < n; base += T / 8) {
    const int t = base + (tid >> 3);
    if (t < n) {
      const int v = list[t];
      const int e1 = indptr[v + 1];
      const bool is_src = v == src, is_dst = v == dst;
      int cnt = 0;
      for (int c = indptr[v] + g; c < e1; c += 8) {
        const int u = indices[c];
        cnt += (test_bit(vis, u) && !((is_src && u == dst) || (is_dst && u == src))) ? 1 : 0;
      }
      cnt += __shfl_xor(cnt, 4);
      cnt += __shfl_xor(cnt, 2);
      cnt += __shfl_xor(cnt, 1);
      if (g == 0) {
        dinv[rank_of(vis, wpre, v)] = cnt > 0 ? 1.0f / sqrtf((float)cnt) : 0.0f;
        edges_local += cnt;
        c_ids[noff + t] = v;
      }
    }
  }
  edges_local = block_sum<T>(edges_local, sh);

  // ---- rows of this link ----------------------------------------------------------------
  const int64_t rp = row_ptr[l];
  const int R = (int)(row_ptr[l + 1] - rp);
  if (plus && wave_id() == 0) common_neighbours(indptr, indices, vis, src, dst, cn);
  if (tid == 0) {
    atomicAdd(tot_edges, (unsigned long long)edges_local);
    for (int d = 0; d < kMaxLevels; ++d)
      lvl_out[(int64_t)l * kMaxLevels + d] = d < nlev ? lvl_end[d] : n;
  }
  __syncthreads();
  for (int r = tid; r < R; r += T) row_nodes[rp + r] = r == 0 ? src : (r == 1 ? dst : cn[r - 2]);

  // ---- per row pair: K pull steps --------------------------------------------------------
  // State s_i[w] = dinv[w]·r_i[w] (float2: rows a and b):
  //   r_i[w] = dinv[w] · Σ_{u ∈ N_S(w)} s_{i-1}[u]            (Â symmetric: pull == r_{i-1}·Â)
  // Each r_i[w] is summed in the stored order of w's row and reduced over 8 lanes by a fixed
  // xor tree: bit-reproducible.  All terms are >= 0: no cancellation.  A walk of length i
  // from a row at hop h_r stays within hop h_r + i, so step i only visits that list prefix.
  const int npairs = (R + 1) / 2;
  for (int p = 0; p < npairs; ++p) {
    const int64_t jid = job_off[l] + p;
    const int64_t coff = coef_off[jid];
    const int node_a = p == 0 ? src : cn[2 * p - 2];
    const int node_b = p == 0 ? dst : (2 * p + 1 < R ? cn[2 * p - 1] : -1);
    const int la = rank_of(vis, wpre, node_a);
    const int lb = node_b >= 0 ? rank_of(vis, wpre, node_b) : -1;
    const int row_hop = p == 0 ? 0 : 1;
    const int support = lvl_end[min(K + row_hop, nlev - 1)];

    for (int w = tid; w < n; w += T) {
      cur[w] = make_float2(0.f, 0.f);
      nxs[w] = make_float2(0.f, 0.f);
    }
    if (tid < 4 * K) zbuf[tid] = 0.f;
    __syncthreads();
    if (tid == 0) {
      cur[la].x = dinv[la];
      if (lb >= 0) cur[lb].y = dinv[lb];
    }
    __syncthreads();

    float2* s_in = cur;
    float2* s_out = nxs;
    float* coef = c_coef + coff * (2 * K);
#pragma unroll 1
    for (int i = 0; i < K; ++i) {
      const int limit = lvl_end[min(i + 1 + row_hop, nlev - 1)];
      for (int base = 0; base < support; base += T / 8) {
        const int t = base + (tid >> 3);
        if (t < limit) {
          const int v = list[t];
          const int w = rank_of(vis, wpre, v);
          const int e1 = indptr[v + 1];
          const bool is_src = v == src, is_dst = v == dst;
          float sx = 0.f, sy = 0.f;
          for (int c = indptr[v] + g; c < e1; c += 8) {
            const int u = indices[c];
            if (test_bit(vis, u) && !((is_src && u == dst) || (is_dst && u == src))) {
              const float2 sv = s_in[rank_of(vis, wpre, u)];
              sx += sv.x;
              sy += sv.y;
            }
          }
#pragma unroll
          for (int o = 4; o > 0; o >>= 1) {
            sx += __shfl_xor(sx, o);
            sy += __shfl_xor(sy, o);
          }
          if (g == 0) {
            const float dw = dinv[w];
            const float rx = dw * sx, ry = dw * sy;
            s_out[w] = make_float2(dw * rx, dw * ry);
            *reinterpret_cast<float2*>(coef + ((int64_t)t * K + i) * 2) = make_float2(rx, ry);
            // label column of operator i+1: Σ_w r[w] z_w = r[src] + r[dst]  (tuned_SIGN.py:177-185)
            if (w == srcl) { zbuf[(0 * K + i) * 2] = rx; zbuf[(0 * K + i) * 2 + 1] = ry; }
            if (w == dstl) { zbuf[(1 * K + i) * 2] = rx; zbuf[(1 * K + i) * 2 + 1] = ry; }
          }
        } else if (t < support && g == 0) {
          *reinterpret_cast<float2*>(coef + ((int64_t)t * K + i) * 2) = make_float2(0.f, 0.f);
        }
      }
      __syncthreads();
      float2* tmp = s_in;
      s_in = s_out;
      s_out = tmp;
    }
    if (tid < 2 * K) {
      const int i = tid >> 1, r = tid & 1;
      job_z[(jid * K + i) * 2 + r] = zbuf[(0 * K + i) * 2 + r] + zbuf[(1 * K + i) * 2 + r];
    }
    if (tid == 0) {
      Job j;
      j.coef_off = coff;
      j.ids_off = noff;
      j.out_row = rp + 2 * p;
      j.link = l;
      j.support = support;
      j.node_a = node_a;
      j.node_b = node_b;
      j.z_a = (node_a == src || node_a == dst) ? 1 : 0;
      j.z_b = (node_b == src || node_b == dst) ? 1 : 0;
      jobs[jid] = j;
      atomicAdd(tot_support, (unsigned long long)support);
    }
    __syncthreads();
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

s3grl_status launch_count(s3grl_context* ctx, const s3grl_graph* g, const int64_t* links, int64_t L,
                          int hops, int plus, int32_t* n_nodes, int32_t* n_rows, int32_t* n_jobs,
                          int32_t* err_flag, int64_t* tot_vol) {
  if (L == 0) return S3GRL_OK;
  const int W = words_for(g->num_nodes);
  const size_t lds = (size_t)(3 * W + 8) * 4;
  S3GRL_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(count_kernel),
                                    hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  hipLaunchKernelGGL(count_kernel, dim3((unsigned)L), dim3(kBlock), lds, ctx->stream, g->indptr,
                     g->indices, (int)g->num_nodes, W, links, hops, plus, n_nodes, n_rows, n_jobs,
                     err_flag, reinterpret_cast<unsigned long long*>(tot_vol));
  S3GRL_HIP_TRY(hipGetLastError());
  return S3GRL_OK;
}

int64_t scan_workspace_elems(int64_t n) { return (n + kScanTile - 1) / kScanTile + 1; }

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

s3grl_status launch_classify(s3grl_context* ctx, const int32_t* n_nodes, int64_t L,
                             int32_t* class_count, int32_t* class_list) {
  if (L == 0) return S3GRL_OK;
  hipLaunchKernelGGL(classify_kernel, dim3((unsigned)((L + 255) / 256)), dim3(256), 0, ctx->stream,
                     n_nodes, L, class_count, class_list);
  S3GRL_HIP_TRY(hipGetLastError());
  return S3GRL_OK;
}

namespace {

struct LinkArgs {
  const s3grl_graph* g;
  const int64_t* links;
  const int32_t* class_list;
  int hops, plus, cn_cap;
  const int64_t *node_off, *row_ptr, *job_off, *coef_off;
  int32_t* c_ids;
  float* c_coef;
  Job* jobs;
  float* job_z;
  int64_t* row_nodes;
  int32_t* lvl;
  int64_t *tot_edges, *tot_support;
};

template <int T, int K>
s3grl_status launch_link_class(s3grl_context* ctx, const LinkArgs& a, int64_t L, int cls, int count) {
  const int nmax = kClassBoundHost[cls];
  const int W = words_for(a.g->num_nodes);
  const size_t words = (size_t)((3 * W + 2 * nmax + 1) & ~1) + 4 * (size_t)nmax + a.cn_cap +
                       kMaxLevels + 4 * K + 32;
  const size_t lds = words * 4;
  if (lds > 163840) {
    set_last_error("link_kernel needs " + std::to_string(lds) + " B of LDS (num_nodes " +
                   std::to_string(a.g->num_nodes) + ", subgraph class " + std::to_string(nmax) +
                   "): above the 160 KiB of a gfx950 CU");
    return S3GRL_ERR_GRAPH_TOO_LARGE;
  }
  auto kern = link_kernel<T, K>;
  S3GRL_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(kern),
                                    hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  hipLaunchKernelGGL(kern, dim3((unsigned)count), dim3(T), lds, ctx->stream, a.g->indptr,
                     a.g->indices, W, a.links, a.class_list + (int64_t)cls * L, a.hops, a.plus, nmax,
                     a.cn_cap, a.node_off, a.row_ptr, a.job_off, a.coef_off, a.c_ids, a.c_coef,
                     a.jobs, a.job_z, a.row_nodes, a.lvl,
                     reinterpret_cast<unsigned long long*>(a.tot_edges),
                     reinterpret_cast<unsigned long long*>(a.tot_support));
  S3GRL_HIP_TRY(hipGetLastError());
  return S3GRL_OK;
}

template <int K>
s3grl_status launch_links_k(s3grl_context* ctx, const LinkArgs& a, int64_t L,
                            const int32_t* class_count_host) {
  // largest subgraphs first: they are the long poles of the tail
  for (int c = kNumClasses - 1; c >= 0; --c) {
    const int count = class_count_host[c];
    if (count == 0) continue;
    // larger subgraphs leave room for fewer workgroups per CU: give them more waves each
    if (c <= 1) S3GRL_TRY((launch_link_class<256, K>(ctx, a, L, c, count)));
    else if (c <= 3) S3GRL_TRY((launch_link_class<512, K>(ctx, a, L, c, count)));
    else S3GRL_TRY((launch_link_class<1024, K>(ctx, a, L, c, count)));
  }
  return S3GRL_OK;
}

}  // namespace

s3grl_status launch_links(s3grl_context* ctx, const s3grl_graph* g, const int64_t* links, int64_t L,
                          const int32_t* class_list, const int32_t* class_count_host, int hops,
                          int plus, int cn_cap, int K, const int64_t* node_off,
                          const int64_t* row_ptr, const int64_t* job_off, const int64_t* coef_off,
                          int32_t* c_ids, float* c_coef, Job* jobs, float* job_z, int64_t* row_nodes,
                          int32_t* lvl, int64_t* tot_edges, int64_t* tot_support) {
  if (L == 0) return S3GRL_OK;
  if (class_count_host[kNumClasses] > 0) {
    set_last_error(std::to_string(class_count_host[kNumClasses]) +
                   " link(s) have subgraphs above " + std::to_string(kClassBoundHost[kNumClasses - 1]) +
                   " nodes: beyond the LDS-resident path of this build");
    return S3GRL_ERR_GRAPH_TOO_LARGE;
  }
  LinkArgs a{g, links, class_list, hops, plus, cn_cap, node_off, row_ptr, job_off, coef_off,
             c_ids, c_coef, jobs, job_z, row_nodes, lvl, tot_edges, tot_support};
  switch (K) {
    case 1: return launch_links_k<1>(ctx, a, L, class_count_host);
    case 2: return launch_links_k<2>(ctx, a, L, class_count_host);
    case 3: return launch_links_k<3>(ctx, a, L, class_count_host);
    case 4: return launch_links_k<4>(ctx, a, L, class_count_host);
    case 5: return launch_links_k<5>(ctx, a, L, class_count_host);
    case 6: return launch_links_k<6>(ctx, a, L, class_count_host);
    case 7: return launch_links_k<7>(ctx, a, L, class_count_host);
    case 8: return launch_links_k<8>(ctx, a, L, class_count_host);
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

}  // namespace s3grl
