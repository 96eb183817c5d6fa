// Induced-CSR flavour of the per-link kernel: plans whose every operator reaches the whole subgraph.
//
// Reference tuned_SIGN.py:153-175 builds the normalised adjacency of the masked induced subgraph and its
// powers A^2 .. A^K by SpGEMM, then keeps rows {src, dst} (+ common neighbours).  link_kernel
// (s3grl_structure.hip) obtains those rows by K pulls r_i = r_{i-1} A_hat over the GLOBAL CSR rows of the
// subgraph's nodes, filtered through N-bit bitmaps: right when an operator only reaches a prefix of the
// hop-major node list (sign_k - 1 < num_hops: the headline), but with sign_k - 1 >= num_hops every operator
// from the num_hops-th on walks ALL rows again — three full walks at PubMed sign_k = 5, 10 of 22 ms, each
// paying a global load, three bitmap words and a popcount rank per stored neighbour, members or not
// (PubMed: 43 % of the stored neighbours of a 3-hop subgraph's nodes lie outside it).
//
// Here the masked induced adjacency is materialised once per link, in LDS, as a CSR whose columns are
// 16-bit positions in the link's hop-major node list, rows in list order, columns in the stored (ascending
// id) order of the global row — a fixed summation order.  All K operators are pulls over it: per entry one
// 2-byte and one 8-byte LDS read.  D^-1/2 comes from the row lengths.
//   csr_count_kernel   sizing: member neighbours per list entry (nodes below the last BFS level hold all
//                      their neighbours in S: global degree, no walk) -> cnt [Σn] uint16, e per link.  The
//                      LDS classes are then cut by the EXACT need (the bound Σ degree is 1.75x loose).
//   link_csr_kernel    bitmap of S (from the cached balls) + rank prefix -> list position of a member;
//                      offsets = scan of cnt; one walk of the global rows scatters the columns (ballot
//                      compaction inside the 4 lanes of a row: ascending order kept); then the passes, with
//                      the bitmaps' LDS reused for the propagation state.
// Same outputs as link_kernel (node lists, row nodes, level ends, jobs, label columns, limits, coefficient
// layout incl. split jobs, statistics); coefficients equal to fp32 round-off (another fixed order).
// Which flavour a link takes depends on the graph, the plan's (num_hops, sign_k) and the link's own
// subgraph — never on the rest of the list: a link gives the same bits in a sharded and an unsharded run.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdlib>

#include "s3grl_device.hpp"

namespace s3grl {
namespace {

#ifndef S3GRL_CSR_COUNT_T
#define S3GRL_CSR_COUNT_T 128   // threads per link of the sizing walk (build-time tuning hook)
#endif
constexpr int kCsrCountT = S3GRL_CSR_COUNT_T;
#ifndef S3GRL_CSR_COUNT_UN
#define S3GRL_CSR_COUNT_UN 4   // row groups in flight per lane group of the sizing walk (build-time tuning hook)
#endif

// ---- sizing: members per row ------------------------------------------------------------------
__global__ __launch_bounds__(kCsrCountT) void csr_count_kernel(
    const int32_t* __restrict__ indptr, const int32_t* __restrict__ indices, int W, const uint32_t* __restrict__ balls,
    const int64_t* __restrict__ links, int hops, const int32_t* __restrict__ n_nodes,
    const int64_t* __restrict__ node_off, const int32_t* __restrict__ lvl, const int32_t* __restrict__ stash, int slot,
    const int32_t* __restrict__ perm, uint16_t* __restrict__ cnt, int32_t* __restrict__ csr_e) {
  extern __shared__ uint32_t smem[];
  constexpr int T = kCsrCountT;
  uint32_t* vis = smem;
  int* sh = reinterpret_cast<int*>(vis + W);
  const int tid = threadIdx.x;
  const int l = perm ? perm[blockIdx.x] : (int)blockIdx.x;
  const int n = n_nodes[l];
  if (n == 0 || n - 2 > slot) {   // folded into its reverse / list not handed over: not of this flavour
    if (tid == 0) csr_e[l] = -1;
    return;
  }
  const int src = (int)links[2 * (int64_t)l], dst = (int)links[2 * (int64_t)l + 1];
  const int64_t noff = node_off[l];
  const int32_t* lv = lvl + (int64_t)l * kMaxLevels;
  const int nlev = lv[kMaxLevels - 1];
  // nodes below level `hops` hold all their neighbours in S; a BFS that ran dry earlier holds everything
  const int walk_from = nlev - 1 == hops ? lv[hops - 1] : n;
  const uint32_t* __restrict__ bs = balls + (int64_t)src * W;
  const uint32_t* __restrict__ bd = balls + (int64_t)dst * W;
  for (int w = tid; w < W; w += T) vis[w] = bs[w] | bd[w];
  const int32_t* __restrict__ st = stash + (int64_t)l * slot;
  // (hops >= 1: src and dst sit below the walked level, so the walked rows are all in the stash)
  const int32_t* list = st + (walk_from - 2);
  int e_local = 0;
  for (int t = tid; t < walk_from; t += T) {
    const int v = t < 2 ? (t == 0 ? min(src, dst) : max(src, dst)) : st[t - 2];
    const int b = indptr[v], d = indptr[v + 1] - b;
    int c = d;
    // the target link is masked (utils.py:79-80)
    if (v == src || v == dst) c -= sorted_contains(indices + b, d, v == src ? dst : src) ? 1 : 0;
    cnt[noff + t] = (uint16_t)c;
    e_local += c;
  }
  __syncthreads();
  walk_rows<T, 4, S3GRL_CSR_COUNT_UN>(
      0, n - walk_from, list, indptr, indices, nullptr,
      [&](RowAcc& a, int v, int u, bool valid) {
        const int mp = v == src ? dst : (v == dst ? src : -1);
        a.n += (valid && test_bit(vis, u) && u != mp) ? 1 : 0;
      },
      [&](RowAcc& a, int t, int) {
        cnt[noff + walk_from + t] = (uint16_t)a.n;
        e_local += a.n;
      });
  e_local = block_sum<T>(e_local, sh);
  if (tid == 0) csr_e[l] = e_local <= 65535 ? e_local : -1;
}

// ---- the link kernel --------------------------------------------------------------------------
template <int T>
__global__ __launch_bounds__(T) void link_csr_kernel(const CsrLinkArgs a, const int32_t* __restrict__ class_list,
                                                     const int K) {
  extern __shared__ uint32_t smem[];
  constexpr int G = 4, RPI = T / G;
  const int32_t* __restrict__ indptr = a.indptr;
  const int32_t* __restrict__ indices = a.indices;
  const int tid = threadIdx.x;
  const int l = class_list[blockIdx.x];
  const int64_t noff = a.node_off[l];
  const int n = (int)(a.node_off[l + 1] - noff);
  const int e = a.csr_e[l];
  const int W = a.W;
  const int mirror = a.mirror_of ? a.mirror_of[l] : -1;   // reversed duplicate folded into l
  const int64_t mrp = mirror >= 0 ? a.row_ptr[mirror] : -1;
  auto ext = [&](int v) -> int { return a.old_of_new ? a.old_of_new[v] : v; };

  // fixed part
  int32_t* cn = reinterpret_cast<int32_t*>(smem);
  int32_t* cnpos = cn + a.cn_cap;
  int* lvl_end = cnpos + a.cn_cap;
  float* zbuf = reinterpret_cast<float*>(lvl_end + kMaxLevels);   // [2 (src,dst)][K][2 (rows)]
  int* sh = reinterpret_cast<int*>(zbuf + 4 * K);
  float* dtab = reinterpret_cast<float*>(sh + 32);
  // the link's CSR
  uint16_t* off = reinterpret_cast<uint16_t*>(dtab + kCsrDinvTable);   // [n + 1]
  uint16_t* cols = off + ((n + 2) & ~1);                               // [e]
  char* ubase = reinterpret_cast<char*>(smem) +
                (((reinterpret_cast<char*>(cols + ((e + 1) & ~1)) - reinterpret_cast<char*>(smem)) + 7) & ~(size_t)7);
  // build view of the shared region ...
  uint32_t* vis = reinterpret_cast<uint32_t*>(ubase);
  uint32_t* wpre = vis + W;
  int32_t* gb = reinterpret_cast<int32_t*>(wpre + W);      // where the global row of list entry t starts ...
  uint16_t* gdeg = reinterpret_cast<uint16_t*>(gb + n);    // ... and its length (max_degree <= 256 on this path)
  uint16_t* por = gdeg + ((n + 1) & ~1);                   // list position of the member of rank r
  // ... and the view of the passes
  float2* cur = reinterpret_cast<float2*>(ubase);
  float2* nxs = cur + n;

  // diagnostic only (S3GRL_DEBUG_STAMPS): cycles per phase, summed over workgroups
  unsigned long long t_prev = a.dbg ? __builtin_amdgcn_s_memtime() : 0ull;
#define S3GRL_CSR_STAMP(idx)                                           \
  if (a.dbg) {                                                         \
    __syncthreads();                                                   \
    const unsigned long long t_now = __builtin_amdgcn_s_memtime();     \
    if (threadIdx.x == 0) atomicAdd(&a.dbg[idx], t_now - t_prev);      \
    t_prev = t_now;                                                    \
  }
  const int src = (int)a.links[2 * (int64_t)l], dst = (int)a.links[2 * (int64_t)l + 1];
  const int pos_src = src < dst ? 0 : 1, pos_dst = 1 - pos_src;
  const int32_t* lv = a.lvl + (int64_t)l * kMaxLevels;   // left by the sizing pass; rewritten below
  const int nlev = lv[kMaxLevels - 1];
  if (tid < nlev) lvl_end[tid] = lv[tid];
  for (int d = tid; d < kCsrDinvTable; d += T) dtab[d] = d > 0 ? 1.0f / sqrtf((float)d) : 0.0f;
  {
    const uint32_t* __restrict__ bs = a.balls + (int64_t)src * W;
    const uint32_t* __restrict__ bd = a.balls + (int64_t)dst * W;
    for (int w = tid; w < W; w += T) vis[w] = bs[w] | bd[w];
  }
  __syncthreads();
  // ONE block scan for both prefix sums: the popcount prefix of the bitmap (rank of a member: <= n < 2^13) in
  // the high half, the row offsets (member counts: <= e < 2^16) in the low half — neither carries
  {
    const int CW = (W + T - 1) / T, CN = (n + T - 1) / T;
    const int w0 = min(tid * CW, W), w1 = min(w0 + CW, W);
    const int t0 = min(tid * CN, n), t1 = min(t0 + CN, n);
    int pop = 0, cnt = 0;
    for (int w = w0; w < w1; ++w) pop += __popc(vis[w]);
    for (int t = t0; t < t1; ++t) cnt += a.cnt[noff + t];
    int total;
    const int run = block_excl_scan<T>((pop << 16) | cnt, sh, total);
    int rp_ = run >> 16, ro = run & 0xffff;
    for (int w = w0; w < w1; ++w) {
      wpre[w] = rp_;
      rp_ += __popc(vis[w]);
    }
    for (int t = t0; t < t1; ++t) {
      off[t] = (uint16_t)ro;
      ro += a.cnt[noff + t];
    }
    if (tid == 0) off[n] = (uint16_t)(total & 0xffff);
  }
  __syncthreads();
  S3GRL_CSR_STAMP(0)
  int vol_local = 0;   // vol(S) = Σ global degrees, the 4·vol(S) term of the algorithmic bytes
  {
    const int32_t* __restrict__ st = a.stash + (int64_t)l * a.slot;
    // four list entries per thread and trip: their ids, then their row bounds, are in flight together (two
    // dependent global loads per entry; one entry at a time made this loop a seventh of the kernel)
    for (int t0 = tid; t0 < n; t0 += 4 * T) {
      int v[4], b[4], en[4];
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const int t = min(t0 + k * T, n - 1);
        v[k] = t < 2 ? (t == 0 ? min(src, dst) : max(src, dst)) : st[t - 2];
      }
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        b[k] = indptr[(uint32_t)v[k]];
        en[k] = indptr[(uint32_t)v[k] + 1u];
      }
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const int t = t0 + k * T;
        if (t < n) {
          a.c_ids[noff + t] = ext(v[k]);
          vol_local += en[k] - b[k];
          gb[t] = b[k];
          gdeg[t] = (uint16_t)(en[k] - b[k]);
          por[rank_of(vis, wpre, v[k])] = (uint16_t)t;
        }
      }
    }
  }
  const int64_t rp = a.row_ptr[l];
  const int R = (int)(a.row_ptr[l + 1] - rp);
  if (a.plus && wave_id() == 0) {
    auto in_s = [&](int u) -> bool { return test_bit(vis, u); };
    const int c = common_neighbours(indptr, indices, in_s, src, dst, cn);
    if (a.old_of_new && c > 1) {   // rows go out in ascending order of the CALLER's ids (as in link_kernel)
      int* key = cn + c;
      int* tmp = cn + 2 * c;
      const int lane = lane_id();
      for (int i = lane; i < c; i += 64) key[i] = a.old_of_new[cn[i]];
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
  for (int r = tid; r < R; r += T) {
    a.row_nodes[rp + r] = ext(r == 0 ? src : (r == 1 ? dst : cn[r - 2]));
    if (mirror >= 0) a.row_nodes[mrp + r] = ext(r == 0 ? dst : (r == 1 ? src : cn[r - 2]));
    if (r >= 2) cnpos[r - 2] = por[rank_of(vis, wpre, cn[r - 2])];
  }
  if (tid == 0)
    for (int d = 0; d < kMaxLevels; ++d) a.lvl[(int64_t)l * kMaxLevels + d] = d < nlev ? lvl_end[d] : n;

  S3GRL_CSR_STAMP(1)
  // ---- columns: one walk of the global rows (their bounds are on chip), four lanes per row, two rows per
  // lane group in flight ----
  // Lane g of a row's group takes the stored neighbours g, g + 4, ...; the members of a step are placed by a
  // ballot over the group's four lanes, so a row's columns keep the ascending order of the global row.
  {
    const int g = tid & (G - 1);
    const int nib_shift = (tid & 63) & ~(G - 1);
    for (int base = 0; base < n; base += 2 * RPI) {
      int c0[2], e1[2], wc[2], mp[2], nx[2];
#pragma unroll
      for (int u = 0; u < 2; ++u) {
        const int t = base + u * RPI + tid / G;
        const bool ok = t < n;
        const int b = gb[min(t, n - 1)];
        c0[u] = ok ? b + g : 0;
        e1[u] = ok ? b + (int)gdeg[min(t, n - 1)] : 0;
        wc[u] = off[min(t, n)];
        mp[u] = t == pos_src ? dst : (t == pos_dst ? src : -1);   // the target link is masked (utils.py:79-80)
      }
#pragma unroll
      for (int u = 0; u < 2; ++u) nx[u] = indices[(uint32_t)(c0[u] < e1[u] ? c0[u] : 0)];
      while (__any(c0[0] < e1[0] || c0[1] < e1[1])) {
#pragma unroll
        for (int u = 0; u < 2; ++u) {
          const bool valid = c0[u] < e1[u];
          const int uu = nx[u];
          const int cnx = c0[u] + G;
          uint32_t an = (uint32_t)(cnx < e1[u] ? cnx : 0);
          asm volatile("" : "+v"(an));   // keep the next load here: it is in flight while this one is placed
          nx[u] = indices[an];
          const uint32_t bit = 1u << (uu & 31);
          const uint32_t wv = vis[uu >> 5];
          const int r = (int)wpre[uu >> 5] + __popc(wv & (bit - 1u));
          const int col = por[min(r, n - 1)];
          const bool member = valid && (wv & bit) && uu != mp[u];
          const unsigned nib = (unsigned)(__ballot(member) >> nib_shift) & ((1u << G) - 1u);
          if (member) cols[wc[u] + __popc(nib & ((1u << g) - 1u))] = (uint16_t)col;
          wc[u] += __popc(nib);
          c0[u] = cnx;
        }
      }
    }
  }
  __syncthreads();   // the CSR is complete; bitmaps, list and rank map are dead from here on
  S3GRL_CSR_STAMP(2)

  auto dinv_of = [&](int t) -> float {
    const int d = (int)off[t + 1] - (int)off[t];
    return d < kCsrDinvTable ? dtab[d] : 1.0f / sqrtf((float)d);
  };
  // ---- per row pair: K pulls over the CSR (same recurrences as link_kernel) ------------------------
  //   s_i[u] = dinv[u]·r_i[u],  r_i[w] = dinv[w] · Σ_{u ∈ N_S(w)} s_{i-1}[u]
  // summed per lane in column order, reduced over the row's four lanes by a fixed xor tree: bit-reproducible.
  const int npairs = (R + 1) / 2;
  for (int pr = 0; pr < npairs; ++pr) {
    const int64_t jid = a.job_off[l] + pr;
    const int64_t coff = a.coef_off ? a.coef_off[jid] : noff;
    const int node_a = pr == 0 ? src : cn[2 * pr - 2];
    const int node_b = pr == 0 ? dst : (2 * pr + 1 < R ? cn[2 * pr - 1] : -1);
    const int pos_a = pr == 0 ? pos_src : cnpos[2 * pr - 2];
    const int pos_b = pr == 0 ? pos_dst : (2 * pr + 1 < R ? cnpos[2 * pr - 1] : -1);
    const int row_hop = pr == 0 ? 0 : 1;
    const int support = lvl_end[min(K + row_hop, nlev - 1)];   // == n
    for (int w = tid; w < n; w += T) {
      cur[w] = make_float2(0.f, 0.f);
      nxs[w] = make_float2(0.f, 0.f);
    }
    if (tid < 4 * K) zbuf[tid] = 0.f;
    __syncthreads();
    if (tid == 0) {
      cur[pos_a].x = dinv_of(pos_a);
      if (pos_b >= 0) cur[pos_b].y = dinv_of(pos_b);
    }
    __syncthreads();
    float2* s_in = cur;
    float2* s_out = nxs;
    float2* coef = reinterpret_cast<float2*>(a.c_coef) + coff * K;   // [K][support] float2
    const bool split = a.split_t > 0 && support > a.split_t;
    auto cidx = [&](int i, int t) -> int64_t {
      if (!split) return (int64_t)i * support + t;
      const int s0 = (t >> a.seg_shift) << a.seg_shift;
      const int len = min(1 << a.seg_shift, support - s0);
      return (int64_t)s0 * K + (int64_t)i * len + (t - s0);
    };
#pragma unroll 1
    for (int i = 0; i < K; ++i) {
      const bool last = i == K - 1;
      const int limit = last ? support : lvl_end[min(i + 1 + row_hop, nlev - 1)];
      const int g = tid & (G - 1);
      for (int base = 0; base < limit; base += RPI) {
        const int t = base + tid / G;
        const bool ok = t < limit;
        const int b = ok ? (int)off[t] : 0, en = ok ? (int)off[t + 1] : 0;
        float ax = 0.f, ay = 0.f;
        for (int c = b + g; c < en; c += 2 * G) {   // two columns per lane and step
          const bool vb = c + G < en;
          const int ca = cols[c], cb = cols[vb ? c + G : c];
          const float2 sa = s_in[ca], sb = s_in[cb];
          ax += sa.x;
          ay += sa.y;
          ax += vb ? sb.x : 0.f;
          ay += vb ? sb.y : 0.f;
        }
#pragma unroll
        for (int o = G / 2; o > 0; o >>= 1) {
          ax += __shfl_xor(ax, o);
          ay += __shfl_xor(ay, o);
        }
        if (ok && g == 0) {
          const int d = en - b;
          const float dw = d < kCsrDinvTable ? dtab[d] : 1.0f / sqrtf((float)d);
          const float rx = dw * ax, ry = dw * ay;
          if (!last) s_out[t] = make_float2(dw * rx, dw * ry);
          coef[cidx(i, t)] = make_float2(rx, ry);
          // label column of operator i+1: Σ_w r[w] z_w = r[src] + r[dst]  (tuned_SIGN.py:177-185)
          if (t == pos_src) { zbuf[(0 * K + i) * 2] = rx; zbuf[(0 * K + i) * 2 + 1] = ry; }
          if (t == pos_dst) { zbuf[(1 * K + i) * 2] = rx; zbuf[(1 * K + i) * 2 + 1] = ry; }
        }
      }
      if (!last)
        for (int t = limit + tid; t < support; t += T) coef[cidx(i, t)] = make_float2(0.f, 0.f);
      __syncthreads();
      float2* tmp = s_in;
      s_in = s_out;
      s_out = tmp;
    }
    if (tid < 2 * K) {
      const int i = tid >> 1, r = tid & 1;
      a.job_z[(jid * K + i) * 2 + r] = zbuf[(0 * K + i) * 2 + r] + zbuf[(1 * K + i) * 2 + r];
    }
    if (tid < K) a.job_lim[jid * K + tid] = tid == K - 1 ? support : lvl_end[min(tid + 1 + row_hop, nlev - 1)];
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
      a.jobs[jid] = j;
      atomicAdd(stat_slot(a.tot_support), (unsigned long long)support * (mirror >= 0 ? 2ull : 1ull));
    }
    __syncthreads();
  }
  S3GRL_CSR_STAMP(3)
  vol_local = block_sum<T>(vol_local, sh);
  if (tid == 0) {
    if (a.dbg) {
      atomicAdd(&a.dbg[4], 1ull);
      atomicAdd(&a.dbg[5], (unsigned long long)(4 * csr_fixed_words(a.cn_cap, K) + csr_lds_need(n, e, W)));
    }
    atomicAdd(stat_slot(a.tot_edges), (unsigned long long)e * (mirror >= 0 ? 2ull : 1ull));
    atomicAdd(stat_slot(a.tot_vol), (unsigned long long)vol_local * (mirror >= 0 ? 2ull : 1ull));
  }
}

}  // namespace

int csr_class_bound(int cls, int cn_cap, int K) {
  static const int nominal[kCsrClasses] = S3GRL_CSR_CLASS_BOUNDS;
  const int avail = 163840 - 4 * csr_fixed_words(cn_cap, K);
  return cls == kCsrClasses - 1 ? avail : std::min(nominal[cls], avail);
}

// Plan-level switch: plain relabelled plans with cached balls whose operators all reach the whole
// subgraph, on graphs of the bitmap flavour (two N-bit bitmaps must leave the LDS to the lists) without
// hub rows (the walks here have no whole-wavefront path for rows of hundreds of neighbours).
bool csr_mode_for(const s3grl_graph* g, int hops, int K, bool balls, bool plain) {
  if (!balls || !plain || g->directed || getenv("S3GRL_NO_CSR")) return false;
  if (K - 1 < hops || hops < 1) return false;
  if (g->max_degree > kHubArmDegree) return false;
  if (getenv("S3GRL_FORCE_CSR")) return g->num_nodes <= 65536;   // test hook: small graphs too
  return !sparse_mode_for(g) && g->num_nodes > 8192;              // (below: the direct-map flavour)
}

s3grl_status launch_csr_count(s3grl_context* ctx, const s3grl_graph* g, const int64_t* links, int64_t L, int hops,
                              const int32_t* n_nodes, const int64_t* node_off, const int32_t* lvl,
                              const int32_t* stash, int slot, const int32_t* perm, uint16_t* cnt, int32_t* csr_e) {
  if (L == 0) return S3GRL_OK;
  const int W = (int)((g->num_nodes + 31) / 32);
  const size_t lds = (size_t)4 * (W + 32);
  S3GRL_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(csr_count_kernel),
                                    hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  hipLaunchKernelGGL(csr_count_kernel, dim3((unsigned)L), dim3(kCsrCountT), lds, ctx->stream, g->indptr, g->indices, W,
                     g->balls.bits + (int64_t)(hops - 1) * g->balls.level_stride, links, hops, n_nodes, node_off, lvl,
                     stash, slot, perm, cnt, csr_e);
  S3GRL_HIP_TRY(hipGetLastError());
  return S3GRL_OK;
}

s3grl_status launch_csr_class(s3grl_context* ctx, const CsrLinkArgs& a, int K, int cls, const int32_t* class_list,
                              int count, hipStream_t stream) {
  if (count == 0) return S3GRL_OK;
  const size_t lds = (size_t)4 * csr_fixed_words(a.cn_cap, K) + (size_t)csr_class_bound(cls, a.cn_cap, K);
  // threads per link by the LDS a link of the class holds (measured on PubMed sign_k = 5, S3GRL_TC_CLASS<c>
  // sweeps: the uniform part of the kernel is most of a small link's cost — fewer threads; the classes that
  // leave a CU three or four workgroups want 512)
  int t = lds <= 18 * 1024 ? 128 : (lds <= 26 * 1024 ? 256 : (lds <= 80 * 1024 ? 512 : 1024));
  {
    char name[32];   // tuning hook
    snprintf(name, sizeof(name), "S3GRL_TC_CLASS%d", cls);
    if (const char* e = getenv(name)) t = atoi(e);
  }
  auto go = [&](auto kern, int T) -> s3grl_status {
    S3GRL_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize,
                                      (int)lds));
    hipLaunchKernelGGL(kern, dim3((unsigned)count), dim3(T), lds, stream, a, class_list, K);
    S3GRL_HIP_TRY(hipGetLastError());
    return S3GRL_OK;
  };
  if (t <= 64) return go(link_csr_kernel<64>, 64);
  if (t <= 128) return go(link_csr_kernel<128>, 128);
  if (t <= 256) return go(link_csr_kernel<256>, 256);
  if (t <= 512) return go(link_csr_kernel<512>, 512);
  return go(link_csr_kernel<1024>, 1024);
}

}  // namespace s3grl

S3GRL_DEFINE_TOUCH(csr)
