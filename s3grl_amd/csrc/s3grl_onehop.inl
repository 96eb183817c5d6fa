// One-hop plans on big graphs (included by s3grl_structure.hip inside namespace s3grl::{anon}).
//
// The reference's own answer to large graphs is num_hops = 1 (reference utils.py:57-74 with
// num_hops = 1: S = {src,dst} ∪ N(src) ∪ N(dst)).  On a power-law graph such a subgraph has a few
// hundred induced edges but its nodes store ~10^4 neighbours (hubs), and with sign_k - 1 >= 1 every
// operator reaches all of S.  Walking the global rows through a hash of S for every operator
// (link_kernel, HS flavour) costs vol(S) probes per pass.  Here instead:
//
//   count1_kernel     n = |S|, R, and a bound of the induced entries — by intersecting the two
//                     sorted rows (binary searches), one wavefront per link, no bitmaps: no limit
//                     on the number of nodes of the graph.
//   link_full_kernel  S by a rank merge of the two sorted rows (canonical order: ascending id),
//                     the masked induced adjacency ONCE as an n x n bit matrix through the
//                     degree-ORIENTED rows (`fwd`: only the neighbours of higher (degree, id); a
//                     hub's oriented row is short, Σ over S is ~10x smaller than vol(S)), from it a
//                     CSR of local ids in LDS, and every operator as a pull over that CSR.
//
// Same rows, same coefficients as link_kernel up to the summation order (ascending local id here,
// stored order of the global row there).

// ---- degree-oriented rows ------------------------------------------------------------------
// fwd(u) = { v in N(u) : (deg v, v) > (deg u, u) } ∪ ({u} if u has a self-loop), ascending id.
// Every undirected edge sits in exactly one oriented row; the longest oriented row of a graph
// with m edges has at most sqrt(2m) entries.
__device__ __forceinline__ bool fwd_keep(int du, int u, int dv, int v) {
  return v == u || dv > du || (dv == du && v > u);
}

__global__ void fwd_count_kernel(const int32_t* __restrict__ indptr, const int32_t* __restrict__ indices,
                                 int64_t N, int32_t* __restrict__ cnt) {
  const int64_t u = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (u >= N) return;
  const int b = indptr[u], e = indptr[u + 1], du = e - b;
  int c = 0;
  for (int k = b; k < e; ++k) {
    const int v = indices[k];
    c += fwd_keep(du, (int)u, indptr[v + 1] - indptr[v], v) ? 1 : 0;
  }
  cnt[u] = c;
}

__global__ void fwd_fill_kernel(const int32_t* __restrict__ indptr, const int32_t* __restrict__ indices,
                                int64_t N, const int64_t* __restrict__ off64, int32_t* __restrict__ fwd_indptr,
                                int32_t* __restrict__ fwd_indices, uint16_t* __restrict__ fwd_deg) {
  const int64_t u = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (u > N) return;
  fwd_indptr[u] = (int32_t)off64[u];
  if (u == N) return;
  // 16-bit copy of the oriented degree for the sizing pass (an oriented row has at most sqrt(2m)
  // entries; saturated beyond 65535, which only loosens a bound)
  fwd_deg[u] = (uint16_t)min((long long)(off64[u + 1] - off64[u]), 65535ll);
  const int b = indptr[u], e = indptr[u + 1], du = e - b;
  int o = (int)off64[u];
  for (int k = b; k < e; ++k) {
    const int v = indices[k];
    if (fwd_keep(du, (int)u, indptr[v + 1] - indptr[v], v)) fwd_indices[o++] = v;
  }
}

// ---- sizes of a one-hop subgraph --------------------------------------------------------------
// lower bound of x in an ascending row, through unsigned offsets on a uniform base
__device__ __forceinline__ int row_lower_bound(const int32_t* __restrict__ a, int n, int x) {
  int lo = 0, hi = n;
  while (lo < hi) {
    const int mid = (lo + hi) >> 1;
    if (a[mid] < x) lo = mid + 1; else hi = mid;
  }
  return lo;
}

// LDS bytes of link_full_kernel beyond its fixed part; `with_bm`: the bit matrix in LDS too
__host__ __device__ __forceinline__ int full_hash_slots(int n) {
  int C = 64;
  while (C < 2 * n) C <<= 1;
  return C;
}
// on_chip: the bit matrix and the CSR columns in LDS too (otherwise both sit in an HBM slice)
__host__ __device__ __forceinline__ int full_lds_need(int n, int ecap, bool on_chip) {
  const int WB = (n + 31) >> 5;
  return 8 * full_hash_slots(n) + 12 * n + 16 + (on_chip ? 2 * ((ecap + 1) & ~1) + 4 * n * WB : 64 * WB);
}

constexpr int kCount1Waves = 4;
constexpr int kLongRow = 96;    // CSR rows longer than this are summed by a whole wavefront
constexpr int kLongCap = 128;   // ... at most this many per link (the others stay with their 4 lanes)

__global__ __launch_bounds__(64 * kCount1Waves) void count1_kernel(
    const int32_t* __restrict__ indptr, const int32_t* __restrict__ indices,
    const uint16_t* __restrict__ fwd_deg, int N, const int64_t* __restrict__ links, int64_t L, int plus,
    int K, const int32_t* __restrict__ partner, const int32_t* __restrict__ mirror_of,
    int32_t* __restrict__ n_nodes, int32_t* __restrict__ p_nodes, int32_t* __restrict__ n_rows,
    int32_t* __restrict__ n_jobs, int32_t* __restrict__ lvl_max, int32_t* __restrict__ e_cap,
    int32_t* __restrict__ err_flag, unsigned long long* __restrict__ tot_nodes_alg,
    unsigned long long* __restrict__ tot_oriented, const int32_t* __restrict__ perm, const HubCache hub,
    int64_t* __restrict__ x_cap) {
  const int lane = threadIdx.x & 63;
  const int64_t li = (int64_t)blockIdx.x * kCount1Waves + (threadIdx.x >> 6);
  if (li >= L) return;
  const int64_t l = perm ? perm[li] : li;   // processing order (launch_link_order)
  const int64_t s64 = links[2 * l], d64 = links[2 * l + 1];
  const bool bad = s64 < 0 || s64 >= N || d64 < 0 || d64 >= N || s64 == d64;
  if (bad || (partner && partner[l] >= 0)) {   // invalid link, or a reversed duplicate (its primary works)
    if (lane == 0) {
      if (bad) atomicMax(err_flag, s64 == d64 ? 2 : 1);
      n_nodes[l] = 0;
      p_nodes[l] = 0;
      n_rows[l] = 0;
      n_jobs[l] = 0;
      lvl_max[l] = 0;
      e_cap[l] = 0;
      if (x_cap) x_cap[l] = -1;
    }
    return;
  }
  const int s = (int)s64, d = (int)d64;
  const int cs = indptr[s + 1] - indptr[s], cd = indptr[d + 1] - indptr[d];
  // the SHORTER row is searched in the longer one (a leaf against a hub: one chunk of searches
  // instead of forty); the longer row is only swept for its oriented degrees
  const bool s_short = cs <= cd;
  const int32_t* __restrict__ ra = indices + indptr[s_short ? s : d];   // shorter
  const int32_t* __restrict__ rb = indices + indptr[s_short ? d : s];   // longer
  const int ca = s_short ? cs : cd, cb = s_short ? cd : cs;
  const int a_own = s_short ? s : d, b_own = s_short ? d : s;
  // src / dst themselves inside a row are not members; a node in its own row is a self-loop
  int members = 0, common = 0, loops = 0;
  long long fsum = lane == 0 ? (long long)fwd_deg[s] + fwd_deg[d] : 0ll;
  for (int c0 = 0; c0 < ca; c0 += 64) {
    const int c = c0 + lane;
    if (c < ca) {
      const int x = ra[c];
      loops += x == a_own ? 1 : 0;
      if (x != s && x != d) {
        const int lb = row_lower_bound(rb, cb, x);
        const bool dup = lb < cb && rb[lb] == x;
        members += 1;
        common += dup ? 1 : 0;
        if (!dup) fsum += fwd_deg[x];            // common ones are counted from the longer row
      }
    }
  }
  for (int c0 = 0; c0 < cb; c0 += 64) {
    const int c = c0 + lane;
    if (c < cb) {
      const int y = rb[c];
      loops += y == b_own ? 1 : 0;
      if (y != s && y != d) {
        members += 1;
        fsum += fwd_deg[y];
      }
    }
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    members += __shfl_xor(members, o);
    common += __shfl_xor(common, o);
    loops += __shfl_xor(loops, o);
    fsum += __shfl_xor(fsum, o);
  }
  if (lane == 0) {
    const int n = 2 + members - common;
    // PoS Plus rows (common_neighbours above: N'(0) ∩ N'(1) on the masked sub-CSR): the common
    // neighbours, plus src / dst themselves when they carry a self-loop
    const int R = plus ? 2 + common + loops : 2;
    const int cum_a = K >= 2 ? n : 2, cum_b = n;
    n_nodes[l] = n;
    p_nodes[l] = R > 2 ? cum_b : cum_a;
    n_rows[l] = R;
    n_jobs[l] = (R + 1) / 2;
    lvl_max[l] = max(2, n - 2);
    e_cap[l] = (int)min(2ll * fsum, (long long)0x3fffffff);
    if (x_cap) {
      // Can link_hub_kernel (s3grl_hub.hip) take this link?  Decided from the graph and the link alone.
      // hub = the endpoint of higher degree (then lower id); the edges outside its cached neighbourhood
      // are at most Σ degree over N(other) ∪ {other}, and at most the oriented entries of S less the
      // hub's star and the cached edges (every induced edge sits in one oriented row of S).
      long long xc = -1;
      const bool s_hub = cs > cd || (cs == cd && s < d);
      const int hb = s_hub ? s : d, ot = s_hub ? d : s;
      const int hk = hub.slot[hb];
      if (hk >= 0 && n <= 65535) {
        const long long c_h = s_hub ? cs : cd, c_o = s_hub ? cd : cs;
        const long long vb = (long long)hub.voln[ot] + c_o;
        const long long eh = hub.col_base[hk + 1] - hub.col_base[hk];
        const long long xb = min(vb, max(fsum - c_h - eh / 2, 0ll));
        if (vb <= kHubVolMax) xc = (hub_stage_bytes(c_h, eh) << 32) | xb;   // (staged bytes of the cache, bound)
      }
      x_cap[l] = xc;
    }
    const unsigned long long mult = (mirror_of && mirror_of[l] >= 0) ? 2ull : 1ull;
    atomicAdd(stat_slot(tot_nodes_alg), mult * (unsigned long long)n);
    // oriented-row entries link_full_kernel will probe for this link (measurement: bench.py's
    // physical-bytes figure of the one-hop path)
    atomicAdd(stat_slot(tot_oriented), (unsigned long long)fsum);
  }
}

// ---- the fused per-link kernel of the full-reach one-hop case ------------------------------------
// LDS (dynamic): [hkeys C | hvals C]  (aliased by the float2 state arrays cur[n], nxs[n] once the
// probes are done: 8C >= 16n)  cn[cn_cap] cnpos[cn_cap] lvl_end[2] zbuf[4K] sh[32]
// list[n] dinv[n] off[n+1] cols[ecap] (uint16 local ids) bm[n][WB]
// (BMG, the class of the biggest subgraphs: no matrix; cols and the list of found edges in a
// per-workgroup HBM slice, four per-wave sort bitmaps of WB words behind off[])
template <int T, int K, bool BMG>
__global__ __launch_bounds__(T, (T <= 256 ? 8 : 1)) void link_full_kernel(
    const int32_t* __restrict__ indptr, const int32_t* __restrict__ indices,
    const int32_t* __restrict__ fwd_indptr, const int32_t* __restrict__ fwd_indices,
    const int64_t* __restrict__ links, const int32_t* __restrict__ class_list, int count, int plus,
    int cn_cap, const int32_t* __restrict__ e_cap, const int64_t* __restrict__ node_off,
    const int64_t* __restrict__ row_ptr, const int64_t* __restrict__ job_off,
    const int64_t* __restrict__ coef_off, const int32_t* __restrict__ mirror_of,
    int32_t* __restrict__ c_ids, float* __restrict__ c_coef, Job* __restrict__ jobs,
    float* __restrict__ job_z, int32_t* __restrict__ job_lim, int64_t* __restrict__ row_nodes,
    int32_t* __restrict__ lvl_out, unsigned long long* __restrict__ tot_edges,
    unsigned long long* __restrict__ tot_support, unsigned long long* __restrict__ tot_vol,
    uint32_t* __restrict__ bm_scratch, int64_t bm_stride_words, int lds_bytes,
    unsigned long long* __restrict__ dbg, const int32_t* __restrict__ old_of_new, int split_t, int seg_shift) {
  extern __shared__ uint32_t smem[];
  const int tid = threadIdx.x;
  constexpr int G = 4;
  // the caller's id of an internal id (the graph is walked in its degree order, s3grl_relabel.hip)
  auto ext = [&](int v) -> int { return old_of_new ? old_of_new[v] : v; };
  // diagnostic only (S3GRL_DEBUG_STAMPS): cycles per phase summed over workgroups, slots 8..15
  unsigned long long t_prev = dbg ? __builtin_amdgcn_s_memtime() : 0ull;
#define S3GRL_FSTAMP(idx)                                                             \
  if (dbg) {                                                                          \
    __syncthreads();                                                                  \
    const unsigned long long t_now = __builtin_amdgcn_s_memtime();                    \
    if (threadIdx.x == 0) atomicAdd(&dbg[8 + idx], t_now - t_prev);                   \
    t_prev = t_now;                                                                   \
  }
  // BMG: a persistent grid, every workgroup owns one bit-matrix slice and strides over the class
  for (int item = blockIdx.x; item < count; item += gridDim.x) {
    const int l = class_list[item];
    const int64_t noff = node_off[l];
    const int n = (int)(node_off[l + 1] - noff);
    const int ecap = (e_cap[l] + 1) & ~1;
    const int mirror = mirror_of ? mirror_of[l] : -1;
    const int64_t mrp = mirror >= 0 ? row_ptr[mirror] : -1;
    const int C = full_hash_slots(n);
    const uint32_t hmask = (uint32_t)(C - 1);
    const int WB = (n + 31) >> 5;
    int32_t* hkeys = reinterpret_cast<int32_t*>(smem);
    int32_t* hvals = hkeys + C;
    float2* cur = reinterpret_cast<float2*>(smem);          // aliases the hash (used after the probes)
    float2* nxs = cur + n;
    int32_t* cn = reinterpret_cast<int32_t*>(smem + 2 * C);
    int32_t* cnpos = cn + cn_cap;
    int* lvl_end = cnpos + cn_cap;                           // [2]
    float* zbuf = reinterpret_cast<float*>(lvl_end + 2);     // [2][K][2]
    int* sh = reinterpret_cast<int*>(zbuf + 4 * K);          // [32]
    uint16_t* longrows = reinterpret_cast<uint16_t*>(sh + 32);   // [kLongCap]
    int32_t* list = sh + 32 + kLongCap / 2;
    float* dinv = reinterpret_cast<float*>(list + n);
    int32_t* off = reinterpret_cast<int32_t*>(dinv + n);     // [n+1]
    // On-chip classes: CSR columns and an n x n bit matrix in LDS.  Big class (BMG): the columns
    // and a list of the found edges sit in this workgroup's HBM slice and there is no matrix — the
    // CSR is built from the edge list (degree count, scan, scatter) and every row is then sorted,
    // so that the sums run in ascending local id like in the on-chip classes.
    uint16_t* cols_l = reinterpret_cast<uint16_t*>(off + n + 1 + ((n + 1) & 1));
    uint32_t* bm_l = reinterpret_cast<uint32_t*>(cols_l + ecap);
    uint32_t* elist = bm_scratch + (int64_t)blockIdx.x * bm_stride_words;              // [ecap / 2]
    uint16_t* cols_g = reinterpret_cast<uint16_t*>(elist + ((ecap / 2 + 1) & ~1));         // [ecap]
    uint32_t* sortbm = reinterpret_cast<uint32_t*>(off + n + 1);                         // [T/64][WB] (BMG)
    // BMG: the columns stay on chip after all when the EXACT entry count (known once the degrees
    // are) fits what the class's LDS leaves — the bound the class was chosen by is loose
    uint16_t* cols_b = reinterpret_cast<uint16_t*>(sortbm + (T / 64) * WB);
    const int cols_b_cap = (lds_bytes - (int)(reinterpret_cast<char*>(cols_b) - reinterpret_cast<char*>(smem))) / 2;
    bool big_on_chip = false;
    auto cols_ld = [&](int k) -> int {
      if constexpr (BMG) return big_on_chip ? cols_b[k] : cols_g[k]; else return cols_l[k];
    };
    auto cols_st = [&](int k, int v) {
      if constexpr (BMG) {
        if (big_on_chip) cols_b[k] = (uint16_t)v; else cols_g[k] = (uint16_t)v;
      } else {
        cols_l[k] = (uint16_t)v;
      }
    };

    const int src = (int)links[2 * (int64_t)l], dst = (int)links[2 * (int64_t)l + 1];
    const int32_t* __restrict__ rs = indices + indptr[src];
    const int32_t* __restrict__ rd = indices + indptr[dst];
    const int cs = indptr[src + 1] - indptr[src], cd = indptr[dst + 1] - indptr[dst];

    // ---- S in canonical order: {min,max}, then N(src) ∪ N(dst) \ {src,dst} ascending ------------
    // rank merge of the two sorted rows: element x of one row lands at (its index among the row's
    // own members) + (members of the other row below x) - (common members below x).  Phase 1 keeps
    // (lower bound in the other row, common?, member?) per element in `tmp` (the hash's space:
    // 4(cs+cd) <= 8n <= 4C... the two tables together hold 2C >= 4n words); phase 2 turns the
    // per-row prefix counts of "common" into positions.
    // both rows are staged in LDS first (the hash's space holds 2C >= 4n >= 2(cs + cd) words):
    // the searches then cost LDS latency instead of a dozen dependent trips to L2 each
    int32_t* rows_l = reinterpret_cast<int32_t*>(smem);   // [cs + cd]: row src, then row dst
    uint32_t* tmp = smem + (cs + cd);                     // [cs + cd]
    for (int e = tid; e < cs + cd; e += T) rows_l[e] = e < cs ? rs[e] : rd[e - cs];
    if (tid == 0) {
      list[0] = min(src, dst);
      list[1] = max(src, dst);
      lvl_end[0] = 2;
      lvl_end[1] = n;
      sh[29] = 0;   // long rows registered
    }
    __syncthreads();
    const int32_t* rs_l = rows_l;
    const int32_t* rd_l = rows_l + cs;
    const bool s_has_s = sorted_contains(rs_l, cs, src), s_has_d = sorted_contains(rs_l, cs, dst);
    const bool d_has_s = sorted_contains(rd_l, cd, src), d_has_d = sorted_contains(rd_l, cd, dst);
    for (int e = tid; e < cs + cd; e += T) {
      const bool from_s = e < cs;
      const int x = rows_l[e];
      const bool member = x != src && x != dst;
      const int lb = from_s ? row_lower_bound(rd_l, cd, x) : row_lower_bound(rs_l, cs, x);
      const bool dup = member && (from_s ? (lb < cd && rd_l[lb] == x) : (lb < cs && rs_l[lb] == x));
      tmp[e] = ((uint32_t)lb << 2) | (dup ? 2u : 0u) | (member ? 1u : 0u);
    }
    __syncthreads();
    for (int side = 0; side < 2; ++side) {
      const int base = side == 0 ? 0 : cs, len = side == 0 ? cs : cd;
      const int32_t* row = rows_l + base;
      const int per = (len + T - 1) / T;
      const int e0 = min(tid * per, len), e1 = min(e0 + per, len);
      int dups = 0;
      for (int e = e0; e < e1; ++e) dups += (tmp[base + e] >> 1) & 1u;
      int total;
      int c = block_excl_scan<T>(dups, sh, total);
      // excluded entries (src, dst themselves) below x, in this row and in the other one
      const bool own_s = side == 0 ? s_has_s : d_has_s, own_d = side == 0 ? s_has_d : d_has_d;
      const bool oth_s = side == 0 ? d_has_s : s_has_s, oth_d = side == 0 ? d_has_d : s_has_d;
      for (int e = e0; e < e1; ++e) {
        const uint32_t w = tmp[base + e];
        const bool dup = (w >> 1) & 1u;
        if ((w & 1u) && !(side == 1 && dup)) {       // common members are emitted from row src only
          const int x = row[e];
          const int own = e - ((own_s && src < x) ? 1 : 0) - ((own_d && dst < x) ? 1 : 0);
          const int oth = (int)(w >> 2) - ((oth_s && src < x) ? 1 : 0) - ((oth_d && dst < x) ? 1 : 0);
          const int pos = 2 + own + oth - c;
          if (pos < n) list[pos] = x;
        }
        c += dup ? 1 : 0;
      }
    }
    __syncthreads();   // list complete, tmp dead
    S3GRL_FSTAMP(0)

    // ---- hash of S (probe structure), node list out, bit matrix zeroed --------------------------
    for (uint32_t t = tid; t <= hmask; t += T) hkeys[t] = -1;
    if constexpr (BMG) {
      for (int t = tid; t <= n; t += T) off[t] = 0;            // degree counters
      if (tid == 0) sh[30] = 0;                                // edges found
    } else {
      for (int i = tid; i < n * WB; i += T) bm_l[i] = 0;
    }
    __syncthreads();
    int vol_local = 0;
    // (on-chip classes: the bounds of the oriented rows are fetched here, with the other per-node loads —
    // the probes below then start from LDS.  They sit in dinv's / off's space, unused until the CSR is built.)
    int32_t* qstart = reinterpret_cast<int32_t*>(dinv);   // [n] first entry of node t's oriented row
    int32_t* qoff = off;                                    // [n + 1] running offsets of those rows
    for (int t = tid; t < n; t += T) {
      const int v = list[t];
      hs_insert(hkeys, hmask, v);
      hvals[hs_find(hkeys, hmask, v)] = t;
      c_ids[noff + t] = ext(v);
      vol_local += indptr[v + 1] - indptr[v];
      if constexpr (!BMG) {
        const int fb = fwd_indptr[v];
        qstart[t] = fb;
        qoff[t] = fwd_indptr[v + 1] - fb;
      }
    }
    const int64_t rp = row_ptr[l];
    const int R = (int)(row_ptr[l + 1] - rp);
    __syncthreads();
    if (plus && tid < 64) {
      const int c =
          common_neighbours(indptr, indices, [&](int x) { return hs_find(hkeys, hmask, x) >= 0; }, src, dst, cn);
      if (old_of_new && c > 1) {   // rows in ascending order of the caller's ids (see link_kernel)
        int* key = cn + c;
        int* srt = cn + 2 * c;
        for (int i = tid; i < c; i += 64) key[i] = old_of_new[cn[i]];
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        __builtin_amdgcn_wave_barrier();
        for (int i = tid; i < c; i += 64) {
          const int k = key[i];
          int r = 0;
          for (int j = 0; j < c; ++j) r += key[j] < k ? 1 : 0;
          srt[r] = cn[i];
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        __builtin_amdgcn_wave_barrier();
        for (int i = tid; i < c; i += 64) cn[i] = srt[i];
      }
    }
    if (tid == 0)
      for (int dd = 0; dd < kMaxLevels; ++dd)
        lvl_out[(int64_t)l * kMaxLevels + dd] = dd == 0 ? 2 : n;
    __syncthreads();
    for (int r = tid; r < R; r += T) {
      const int node = r == 0 ? src : (r == 1 ? dst : cn[r - 2]);
      row_nodes[rp + r] = ext(node);
      if (mirror >= 0) row_nodes[mrp + r] = ext(r == 0 ? dst : (r == 1 ? src : cn[r - 2]));
      if (r >= 2) cnpos[r - 2] = hvals[hs_find(hkeys, hmask, node)];
    }
    const int pos_src = src < dst ? 0 : 1, pos_dst = 1 - pos_src;
    S3GRL_FSTAMP(1)

    // ---- masked induced adjacency through the oriented rows (reference utils.py:76-80) ----------
    if constexpr (!BMG) {
      // The oriented rows are walked FLAT: entry e of their concatenation by thread e mod T — a subgraph of
      // 25 nodes has ~40 oriented entries, and a walk with four lanes per row paid two dependent global
      // round trips (bounds, then entries) for rows of one or two entries; here the bounds are in LDS.
      {
        const int per = (n + T - 1) / T;
        const int t0 = min(tid * per, n), t1 = min(t0 + per, n);
        int mine = 0;
        for (int t = t0; t < t1; ++t) mine += qoff[t];
        int total;
        int run = block_excl_scan<T>(mine, sh, total);
        for (int t = t0; t < t1; ++t) {
          const int len = qoff[t];
          qoff[t] = run;
          run += len;
        }
        if (tid == 0) qoff[n] = total;
      }
      __syncthreads();
      const int walk_total = qoff[n];
      for (int e = tid; e < walk_total; e += T) {
        int ra = 0, rb = n;   // the row holding entry e: last r with qoff[r] <= e
        while (rb - ra > 1) {
          const int mid = (ra + rb) >> 1;
          if (qoff[mid] <= e) ra = mid; else rb = mid;
        }
        const int u = fwd_indices[qstart[ra] + (e - qoff[ra])];
        const int v = list[ra];
        const int slot = hs_find(hkeys, hmask, u);
        const bool target = (v == src && u == dst) || (v == dst && u == src);
        if (slot >= 0 && !target) {
          const int i = ra, j = hvals[slot];
          atomicOr(&bm_l[i * WB + (j >> 5)], 1u << (j & 31));
          if (i != j) atomicOr(&bm_l[j * WB + (i >> 5)], 1u << (i & 31));
        }
      }
    } else {
    walk_rows<T, G, 2>(
        0, n, list, fwd_indptr, fwd_indices, nullptr,
        [&](RowAcc& a, int v, int u, bool valid) {
          const int slot = hs_find(hkeys, hmask, u);
          const int j = hvals[max(slot, 0)];
          const int i = a.row;
          const bool target = (v == src && u == dst) || (v == dst && u == src);
          const bool found = valid && slot >= 0 && !target;
          if constexpr (BMG) {
            // one LDS atomic per wavefront for the list position (the visit is called by all lanes)
            const unsigned long long fm = __ballot(found);
            int base = 0;
            if (fm) {
              const int leader = __ffsll((long long)fm) - 1;
              if ((tid & 63) == leader) base = atomicAdd(&sh[30], __popcll(fm));
              base = __shfl(base, leader);
            }
            if (found) {
              const int k = base + __popcll(fm & ((1ull << (tid & 63)) - 1ull));
              if (2 * k < ecap) elist[k] = ((uint32_t)i << 16) | (uint32_t)j;
              atomicAdd(&off[i], 1);
              if (i != j) atomicAdd(&off[j], 1);
            }
          } else if (found) {
            atomicOr(&bm_l[i * WB + (j >> 5)], 1u << (j & 31));
            if (i != j) atomicOr(&bm_l[j * WB + (i >> 5)], 1u << (i & 31));
          }
        },
        [](RowAcc&, int, int) {});
    }
    if constexpr (BMG) __threadfence();   // the edge list is read back by other waves (through L2)
    __syncthreads();
    S3GRL_FSTAMP(2)

    // ---- degrees, D^-1/2 (inf -> 0), CSR of local ids (ascending) --------------------------------
    if constexpr (BMG) {
      int32_t* cursor = hvals;              // the hash values are dead: positions were copied out
      const int per = (n + T - 1) / T;
      const int t0 = min(tid * per, n), t1 = min(t0 + per, n);
      int mine = 0;
      for (int t = t0; t < t1; ++t) mine += off[t];
      int total;
      int run = block_excl_scan<T>(mine, sh, total);
      for (int t = t0; t < t1; ++t) {
        const int dg = off[t];
        dinv[t] = dg > 0 ? 1.0f / sqrtf((float)dg) : 0.0f;
        off[t] = run;
        cursor[t] = run;
        run += dg;
      }
      if (tid == 0) off[n] = total;
      big_on_chip = total <= cols_b_cap;
      if (dbg && tid == 0) {   // diagnostic: links of the class / whose columns had to stay in the HBM slice
        atomicAdd(&dbg[8 + 6], 1ull);
        if (!big_on_chip) atomicAdd(&dbg[8 + 7], 1ull);
      }
      __syncthreads();
      const int found = min(sh[30], ecap / 2);
      for (int k = tid; k < found; k += T) {
        const uint32_t w = elist[k];
        const int i = (int)(w >> 16), j = (int)(w & 0xffffu);
        cols_st(atomicAdd(&cursor[i], 1), j);
        if (i != j) cols_st(atomicAdd(&cursor[j], 1), i);
      }
      if (!big_on_chip) __threadfence();
      __syncthreads();
      // every row ascending: one wavefront per row.  Short rows by rank (each lane counts the
      // smaller entries), long ones through a per-wave bitmap of the n local ids.
      const int lane = tid & 63, wv = tid >> 6;
      uint32_t* wbm = sortbm + wv * WB;
      // (a) rows of at most 16 entries — nearly all of them — four at a time per wavefront
      {
        const int sub = lane >> 4, sl = lane & 15;
        for (int r0 = wv * 4; r0 < n; r0 += (T / 64) * 4) {
          const int r = min(r0 + sub, n - 1);
          const int b = off[r], len = off[r + 1] - b;
          const bool mine = r0 + sub < n && len > 1 && len <= 16;
          const int x = mine && sl < len ? cols_ld(b + sl) : 0x7fffffff;
          int rank = 0;
#pragma unroll
          for (int k = 0; k < 16; ++k) rank += __shfl(x, (lane & 48) + k) < x ? 1 : 0;
          if (mine && sl < len) cols_st(b + rank, x);
        }
      }
      // (b) longer rows, one wavefront each: by rank up to 64 entries, through a bitmap beyond
      for (int r = wv; r < n; r += T / 64) {
        const int b = off[r], len = off[r + 1] - b;
        if (len <= 16) continue;
        if (len <= 64) {
          const int x = lane < len ? cols_ld(b + lane) : 0x7fffffff;
          int rank = 0;
          for (int k = 0; k < len; ++k) rank += __shfl(x, k) < x ? 1 : 0;
          if (lane < len) cols_st(b + rank, x);
        } else {
          for (int w = lane; w < WB; w += 64) wbm[w] = 0;
          for (int k = lane; k < len; k += 64) {
            const int c = cols_ld(b + k);
            atomicOr(&wbm[c >> 5], 1u << (c & 31));
          }
          int base = b;
          for (int w0 = 0; w0 < WB; w0 += 64) {
            uint32_t word = w0 + lane < WB ? wbm[w0 + lane] : 0u;
            const int cnt = __popc(word);
            int inc = cnt;
#pragma unroll
            for (int o = 1; o < 64; o <<= 1) {
              const int t = __shfl_up(inc, o);
              if (lane >= o) inc += t;
            }
            int k = base + inc - cnt;
            while (word) {
              const int bit = __ffs(word) - 1;
              word &= word - 1;
              cols_st(k++, (w0 + lane) * 32 + bit);
            }
            base += __shfl(inc, 63);
          }
        }
      }
      if (!big_on_chip) __threadfence();
    } else {
      const int per = (n + T - 1) / T;
      const int t0 = min(tid * per, n), t1 = min(t0 + per, n);
      int mine = 0;
      for (int t = t0; t < t1; ++t) {
        int dg = 0;
        for (int j = 0; j < WB; ++j) dg += __popc(bm_l[t * WB + j]);
        mine += dg;
      }
      int total;
      int run = block_excl_scan<T>(mine, sh, total);
      for (int t = t0; t < t1; ++t) {
        off[t] = run;
        int k = run;
        for (int j = 0; j < WB; ++j) {
          uint32_t w = bm_l[t * WB + j];
          while (w) {
            const int b = __ffs(w) - 1;
            w &= w - 1;
            if (k < ecap) cols_l[k] = (uint16_t)(j * 32 + b);
            ++k;
          }
        }
        const int dg = k - run;
        dinv[t] = dg > 0 ? 1.0f / sqrtf((float)dg) : 0.0f;
        run = k;
      }
      if (tid == 0) off[n] = total;
    }
    __syncthreads();   // hash dead from here: cur / nxs take its space
    S3GRL_FSTAMP(3)
    const int edges_total = off[n];
    // Long rows (src and dst are adjacent to about half of a one-hop subgraph each, a hub inside
    // it to more): four lanes would stride such a row for hundreds of trips while the rest of the
    // workgroup waits at the barrier of the pass.  They are registered here (sign bit of their
    // D^-1/2 as the per-row flag) and summed by a whole wavefront each, after the short rows.
    // (the first kLongCap of them in row order — a block scan, not the order of an atomic: which rows
    // get a wavefront decides the last bit of their sums)
    int nlong;
    {
      const int per = (n + T - 1) / T;
      const int t0 = min(tid * per, n), t1 = min(t0 + per, n);
      int mine = 0;
      for (int t = t0; t < t1; ++t) mine += off[t + 1] - off[t] > kLongRow ? 1 : 0;
      int total;
      int q = block_excl_scan<T>(mine, sh, total);
      for (int t = t0; t < t1; ++t) {
        if (off[t + 1] - off[t] > kLongRow) {
          if (q < kLongCap) {
            longrows[q] = (uint16_t)t;
            dinv[t] = -dinv[t];
          }
          ++q;
        }
      }
      nlong = min(total, kLongCap);
    }
    __syncthreads();

    // ---- per row pair: K pulls over the CSR ------------------------------------------------------
    const int npairs = (R + 1) / 2;
    for (int pr = 0; pr < npairs; ++pr) {
      const int64_t jid = job_off[l] + pr;
      const int64_t coff = coef_off ? coef_off[jid] : noff;
      const int node_a = pr == 0 ? src : cn[2 * pr - 2];
      const int node_b = pr == 0 ? dst : (2 * pr + 1 < R ? cn[2 * pr - 1] : -1);
      const int la = pr == 0 ? pos_src : cnpos[2 * pr - 2];
      const int lb = pr == 0 ? pos_dst : (node_b >= 0 ? cnpos[2 * pr - 1] : -1);
      for (int w = tid; w < n; w += T) {
        cur[w] = make_float2(w == la ? fabsf(dinv[w]) : 0.f, w == lb ? fabsf(dinv[w]) : 0.f);
      }
      if (tid < 4 * K) zbuf[tid] = 0.f;
      __syncthreads();
      float2* s_in = cur;
      float2* s_out = nxs;
      float2* coef = reinterpret_cast<float2*>(c_coef) + coff * K;   // [K][n] float2
      // lists longer than split_t: coefficients piece by piece (see link_kernel)
      const bool split = split_t > 0 && n > split_t;
      auto cidx = [&](int i, int t) -> int64_t {
        if (!split) return (int64_t)i * n + t;
        const int s0 = (t >> seg_shift) << seg_shift;
        const int len = min(1 << seg_shift, n - s0);
        return (int64_t)s0 * K + (int64_t)i * len + (t - s0);
      };
#pragma unroll 1
      for (int i = 0; i < K; ++i) {
        const int g = tid & (G - 1);
        auto commit = [&](int t, float ax, float ay) {
          const float dw = fabsf(dinv[t]);
          const float rx = dw * ax, ry = dw * ay;
          s_out[t] = make_float2(dw * rx, dw * ry);
          coef[cidx(i, t)] = make_float2(rx, ry);
          // label column of operator i+1: r[src] + r[dst]  (tuned_SIGN.py:177-185)
          if (t == pos_src) { zbuf[(0 * K + i) * 2] = rx; zbuf[(0 * K + i) * 2 + 1] = ry; }
          if (t == pos_dst) { zbuf[(1 * K + i) * 2] = rx; zbuf[(1 * K + i) * 2 + 1] = ry; }
        };
        for (int base = 0; base < n; base += T / G) {
          const int t = base + tid / G;
          float ax = 0.f, ay = 0.f;
          const bool mine = t < n && !(dinv[min(t, n - 1)] < 0.f);   // short row of this lane group
          if (mine) {
            const int k1 = off[t + 1];
            for (int k = off[t] + g; k < k1; k += G) {
              const float2 sv = s_in[cols_ld(k)];
              ax += sv.x;
              ay += sv.y;
            }
          }
#pragma unroll
          for (int o = G / 2; o > 0; o >>= 1) {
            ax += __shfl_xor(ax, o);
            ay += __shfl_xor(ay, o);
          }
          if (mine && g == 0) commit(t, ax, ay);
        }
        for (int q = tid >> 6; q < nlong; q += T / 64) {   // one wavefront per long row
          const int t = longrows[q];
          const int k1 = off[t + 1];
          float ax = 0.f, ay = 0.f;
          for (int k = off[t] + (tid & 63); k < k1; k += 64) {
            const float2 sv = s_in[cols_ld(k)];
            ax += sv.x;
            ay += sv.y;
          }
#pragma unroll
          for (int o = 32; o > 0; o >>= 1) {
            ax += __shfl_xor(ax, o);
            ay += __shfl_xor(ay, o);
          }
          if ((tid & 63) == 0) commit(t, ax, ay);
        }
        __syncthreads();
        float2* tmp2 = s_in;
        s_in = s_out;
        s_out = tmp2;
      }
      if (tid < 2 * K) {
        const int i = tid >> 1, r = tid & 1;
        job_z[(jid * K + i) * 2 + r] = zbuf[(0 * K + i) * 2 + r] + zbuf[(1 * K + i) * 2 + r];
      }
      // operator i+1 reaches the list prefix within i+1 hops of the row; with one hop that is the
      // whole list from the first operator on (from the second for nothing: n == support)
      if (tid < K) job_lim[jid * K + tid] = n;
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
        jobs[jid] = j;
        atomicAdd(stat_slot(tot_support), (unsigned long long)n * (mirror >= 0 ? 2ull : 1ull));
      }
      __syncthreads();
    }
    S3GRL_FSTAMP(4)
    vol_local = block_sum<T>(vol_local, sh);
    if (tid == 0) {
      atomicAdd(stat_slot(tot_edges), (unsigned long long)edges_total * (mirror >= 0 ? 2ull : 1ull));
      atomicAdd(stat_slot(tot_vol), (unsigned long long)vol_local * (mirror >= 0 ? 2ull : 1ull));
    }
    __syncthreads();   // LDS is reused by the next item of a persistent workgroup
  }
}
