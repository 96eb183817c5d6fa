// SoP path (global operators) and feature-layout helpers, gfx950.
//
// Reference: sgrl_link_pred.py:161-178 builds Â = D^-1/2 A D^-1/2 of the WHOLE train graph and
// Â², …, Â^K by SpGEMM; tuned_SIGN.py:49-134 then, per link (s, d) and operator i, takes row s
// of Â^i with column d zeroed (and row d with column s zeroed), multiplies by X, and prepends
// the diagonal entry Â^i[s,s] (Â^i[d,d]).
//
// Here no power of Â is ever materialised.  With Y_i = Â^i X (K global SpMMs, once):
//     x_i[s] = [ Â^i[s,s] | Y_i[s] − Â^i[s,d]·X[d] ]      x_i[d] = [ Â^i[d,d] | Y_i[d] − Â^i[s,d]·X[s] ]
// and the three scalars per operator are meet-in-the-middle dot products of short propagated
// rows:  Â^i[s,d] = r_a(s)·r_b(d), a + b = i, a = ⌊i/2⌋ (Â is symmetric), formed per link inside
// the ⌈K/2⌉-hop ball of {s,d} in LDS.  Everything is computed in f64 and rounded to f32 once:
// the subtraction cancels exactly where the reference's masked sum is exactly zero (a leaf s
// hanging off d), which f32 intermediates would turn into 1e-8-sized noise.
#include <algorithm>
#include <cstring>
#include <memory>

#include <cstring>


#include "s3grl_internal.hpp"
#include "s3grl_device.hpp"

namespace s3grl {
namespace {

__global__ void to_f64_pad_kernel(const float* __restrict__ X, int64_t ldx, int64_t N, int64_t F,
                                  double* __restrict__ Y, float* __restrict__ Yhi, int64_t ldy) {
  const int64_t total = N * ldy;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total;
       i += (int64_t)gridDim.x * blockDim.x) {
    const int64_t r = i / ldy, c = i - r * ldy;
    const float x = c < F ? X[r * ldx + c] : 0.0f;
    Y[i] = (double)x;
    Yhi[i] = x;      // plane 0 of the f32 table the row kernel reads: X itself, exact
  }
}

// deg = stored entries per row (reference sgrl_link_pred.py:170-172), deg^-1/2, inf -> 0.
// `mult` (may be null): multiplicity of every stored entry — the reference builds its SparseTensor
// from the UNCOALESCED edge_index (sgrl_link_pred.py:161-167), so a pair that appears m times counts
// m times in the degree and carries m times the weight in every product; scipy's A holds such a
// pair once (duplicates summed).
__global__ void global_dinv_kernel(const int32_t* __restrict__ indptr, const float* __restrict__ mult,
                                   int64_t N, double* __restrict__ dinv) {
  const int64_t v = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (v >= N) return;
  double deg = (double)(indptr[v + 1] - indptr[v]);
  if (mult) {
    deg = 0.0;
    for (int e = indptr[v]; e < indptr[v + 1]; ++e) deg += (double)mult[e];
  }
  dinv[v] = deg > 0 ? 1.0 / sqrt(deg) : 0.0;
}

// Y_out[v,:] = dinv[v] · Σ_{u ∈ N(v)} dinv[u] · Y_in[u,:]      f64, neighbours in stored order.
// One wavefront per (row, 256-column tile): a lane owns columns {2l, 2l+1} and {128+2l, 129+2l} of
// the tile (two 16-byte loads per neighbour), four neighbours in flight.  A PubMed row has 4.5
// neighbours: one wave per whole row (six tiles in sequence) left the loads of a tile waiting for
// the previous tile's sums — latency-bound at 0.4 TB/s; (row, tile) waves run at the fabric's rate.
typedef double double2_t __attribute__((ext_vector_type(2)));

__global__ __launch_bounds__(256) void spmm_norm_kernel(
    const int32_t* __restrict__ indptr, const int32_t* __restrict__ indices,
    const double* __restrict__ dinv, const double* __restrict__ Yin, double* __restrict__ Yout,
    float* __restrict__ Hout, float* __restrict__ Lout,
    int64_t N, int64_t ldy, int tiles, const float* __restrict__ mult) {
  const int lane = threadIdx.x & 63;
  const int64_t w = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (w >= N * tiles) return;
  const int64_t v = w / tiles;
  const int tile = (int)(w - v * tiles);
  const int e0 = indptr[v], e1 = indptr[v + 1];
  const int64_t ca = (int64_t)tile * 256 + 2 * lane, cb = ca + 128;
  const bool oka = ca < ldy, okb = cb < ldy;          // ldy is even: a pair is inside or outside
  const int64_t la = oka ? ca : 0, lb = okb ? cb : 0;
  double2_t acc_a = {0.0, 0.0}, acc_b = {0.0, 0.0};
  for (int e = e0; e < e1; e += 4) {
    int u[4];
    double du[4];
    double2_t ya[4], yb[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) u[k] = indices[min(e + k, e1 - 1)];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      du[k] = e + k < e1 ? dinv[u[k]] * (mult ? (double)mult[e + k] : 1.0) : 0.0;
      const double* __restrict__ yr = Yin + (int64_t)u[k] * ldy;
      ya[k] = *reinterpret_cast<const double2_t*>(yr + la);
      yb[k] = *reinterpret_cast<const double2_t*>(yr + lb);
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      if (e + k < e1) {          // uniform over the wave: the same sums in the same order as a scalar loop
        acc_a += du[k] * ya[k];
        acc_b += du[k] * yb[k];
      }
    }
  }
  const double dv = dinv[v];
  double* __restrict__ yo = Yout + v * ldy;
  // ... and the same values as an f32 pair hi + lo (hi = the value rounded to f32, lo = what the rounding
  // dropped, rounded to f32: together 48 bits of the f64 value) — what the per-link row kernel reads
  auto split_store = [&](int64_t c, const double2_t y) {
    const float hx = (float)y.x, hy = (float)y.y;
    *reinterpret_cast<float2*>(Hout + v * ldy + c) = make_float2(hx, hy);
    *reinterpret_cast<float2*>(Lout + v * ldy + c) = make_float2((float)(y.x - (double)hx), (float)(y.y - (double)hy));
  };
  if (oka) {
    const double2_t y = dv * acc_a;
    *reinterpret_cast<double2_t*>(yo + ca) = y;
    split_store(ca, y);
  }
  if (okb) {
    const double2_t y = dv * acc_b;
    *reinterpret_cast<double2_t*>(yo + cb) = y;
    split_store(cb, y);
  }
}

__device__ __forceinline__ double wave_sum_f64(double v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
  return v;
}

// Per link: Â^i[s,d], Â^i[s,s], Â^i[d,d] for i = 1..K.  LDS: vis/nxt/wpre bitmaps, list[n],
// r[HB][n] double2 (x: propagated from s, y: from d), misc.
// EXT: the three N-bit bitmaps live in an HBM slice per workgroup (`ext`, `ext_stride` words) instead of LDS —
// graphs of more than kMaxNodesLds nodes; same code, slower memory; launched over chunks of a class list, so
// that a bounded number of slices serves any number of links (like count_kernel's EXT flavour).
template <int T, int G, bool EXT = false>
__global__ __launch_bounds__(T) void sop_scalar_kernel(
    const int32_t* __restrict__ indptr, const int32_t* __restrict__ indices, int W,
    const double* __restrict__ gdinv, const int64_t* __restrict__ links,
    const int32_t* __restrict__ class_list, const int64_t* __restrict__ node_off, int K, int HB, int RB,
    int hubs, double* __restrict__ scal /* [L][K][3] = sd, ss, dd */, const float* __restrict__ mult,
    uint32_t* __restrict__ ext, int64_t ext_stride) {
  extern __shared__ uint32_t smem[];
  const int tid = threadIdx.x;
  const int l = class_list[blockIdx.x];
  const int n_alloc = (int)(node_off[l + 1] - node_off[l]);
  const int WL = EXT ? 0 : W;   // bitmap words that sit in LDS
  uint32_t* bmp = EXT ? ext + (int64_t)blockIdx.x * ext_stride : smem;
  uint32_t* vis = bmp;
  uint32_t* nxt = bmp + W;
  uint32_t* wpre = bmp + 2 * W;
  int* lvl_end = reinterpret_cast<int*>(smem + 3 * WL);
  int* sh = lvl_end + kMaxLevels;
  int* hub = hubs ? sh + 32 : nullptr;
  const int red_off = (3 * WL + kMaxLevels + 32 + kHubWords + 1) & ~1;   // doubles: 8-byte aligned
  double* red = reinterpret_cast<double*>(smem + red_off);           // [16 waves][3]
  int32_t* list = reinterpret_cast<int32_t*>(smem + red_off + 96);
  double2* r = reinterpret_cast<double2*>(smem + ((red_off + 96 + n_alloc + 3) & ~3));  // [HB][n]

  // the scalars are formed for the pair in canonical orientation (lower id first) and handed out in
  // the link's own: (d,s) then gets bit for bit what (s,d) gets — a folded reversed duplicate, a
  // pair split over two ranks and a pair computed twice all agree exactly
  const int l_s = (int)links[2 * (int64_t)l], l_d = (int)links[2 * (int64_t)l + 1];
  const bool swp = l_s > l_d;
  const int src = swp ? l_d : l_s, dst = swp ? l_s : l_d;
  const int g = tid & (G - 1);
  int nlev;
  // ball of radius RB: HB = ⌈K/2⌉ when K is even; HB - 1 when K is odd — r_HB then only ever meets
  // r_{HB-1} in a dot product (i = K: a = HB - 1, b = HB), whose support lies within HB - 1 hops,
  // so r_HB is only evaluated there (its pulls still range over whole global rows: neighbours
  // outside the ball carry r_{HB-1} = 0).  K = 3: the 1-hop ball instead of the 2-hop one.
  const int n = bfs_list<T, G>(indptr, indices, W, src, dst, RB, vis, nxt, list, n_alloc, lvl_end, sh, hub, nlev);
  rank_prefix<T>(vis, wpre, W, sh);
  for (int w = tid; w < n * HB; w += T) r[w] = make_double2(0.0, 0.0);
  __syncthreads();
  const int sl = rank_of(vis, wpre, src), dl = rank_of(vis, wpre, dst);

  // r_j[w] = dinv[w] Σ_{u ∈ N(w)} dinv[u] r_{j-1}[u];  r_0 = (e_s, e_d).  A walk of length j from
  // {s,d} stays within hop j, and every neighbour that matters is inside the ball: no masking,
  // global degrees (the target link is NOT removed for SoP, sgrl_link_pred.py:166-173).
  for (int j = 1; j <= HB; ++j) {
    const int limit = lvl_end[min(j, nlev - 1)];
    double2* out = r + (int64_t)(j - 1) * n;
    const double2* in = j >= 2 ? r + (int64_t)(j - 2) * n : nullptr;
    for (int base = 0; base < limit; base += T / G) {
      const int t = base + tid / G;
      if (t < limit) {
        const int v = list[t];
        const int e1 = indptr[v + 1];
        double sx = 0.0, sy = 0.0;
        for (int c = indptr[v] + g; c < e1; c += G) {
          const int u = indices[c];
          const double m = mult ? (double)mult[c] : 1.0;
          if (j == 1) {
            if (u == src) sx += gdinv[u] * m;
            if (u == dst) sy += gdinv[u] * m;
          } else if (test_bit(vis, u)) {
            const double2 rv = in[rank_of(vis, wpre, u)];
            const double du = gdinv[u] * m;
            sx += du * rv.x;
            sy += du * rv.y;
          }
        }
#pragma unroll
        for (int o = G / 2; o > 0; o >>= 1) {
          sx += __shfl_xor(sx, o);
          sy += __shfl_xor(sy, o);
        }
        if (g == 0) {
          const double dv = gdinv[v];
          out[rank_of(vis, wpre, v)] = make_double2(dv * sx, dv * sy);
        }
      }
    }
    __syncthreads();
  }

  // dots: a = ⌊i/2⌋, b = i − a;  r_0(s) = e_s
  for (int i = 1; i <= K; ++i) {
    const int a = i / 2, b = i - a;
    const double2* rb = r + (int64_t)(b - 1) * n;
    double sd = 0.0, ss = 0.0, dd = 0.0;
    if (a == 0) {
      if (tid == 0) {
        sd = rb[sl].y;   // Â^b[d, s] = Â^b[s, d]
        ss = rb[sl].x;
        dd = rb[dl].y;
      }
    } else {
      const double2* ra = r + (int64_t)(a - 1) * n;
      for (int w = tid; w < n; w += T) {
        const double2 x = ra[w], y = rb[w];
        sd += x.x * y.y;
        ss += x.x * y.x;
        dd += x.y * y.y;
      }
    }
    sd = wave_sum_f64(sd);
    ss = wave_sum_f64(ss);
    dd = wave_sum_f64(dd);
    if ((tid & 63) == 0) {
      red[(tid >> 6) * 3 + 0] = sd;
      red[(tid >> 6) * 3 + 1] = ss;
      red[(tid >> 6) * 3 + 2] = dd;
    }
    __syncthreads();
    if (tid < 3) {
      double acc = 0.0;
      for (int w = 0; w < T / 64; ++w) acc += red[w * 3 + tid];   // fixed order
      scal[((int64_t)l * K + (i - 1)) * 3 + ((swp && tid) ? 3 - tid : tid)] = acc;   // ss <-> dd when swapped
    }
    __syncthreads();
  }
}

// rows[2l + e, i, :] for e ∈ {src, dst}: one wave per link, f64 arithmetic, f32 out.
// Column-outer: a lane loads X[s,c], X[d,c] ONCE and then the 2K entries Y_i[s,c], Y_i[d,c] — 2K + 2
// independent loads in flight per trip, every row of the f64 table read exactly once per link.
// (Operator-outer, the first version, re-read the two X rows for each operator: 14 row reads per
// link instead of 8 at K = 3, all of them beyond L2 — 29 GB of fabric traffic against 16 GB.)
// Launch order: the links sorted by (src, dst) — `order`.  A node is an endpoint of ~16 links of a
// PubMed split, and in the caller's (permuted) order every one of them fetches its 2(K+1) table
// rows from HBM again (the f64 table is 4x the Infinity Cache).  Sorted, the four wavefronts of a
// workgroup and the workgroups next to them share their src rows out of L1 / L2 / Infinity Cache;
// only the dst rows still come from HBM.  The output keeps the caller's order (rows of link l at 2l).
__global__ void sop_order_keys_kernel(const int64_t* __restrict__ links, int64_t L, uint64_t* __restrict__ keys,
                                      int32_t* __restrict__ vals) {
  const int64_t l = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (l >= L) return;
  keys[l] = ((uint64_t)(uint32_t)links[2 * l] << 32) | (uint64_t)(uint32_t)links[2 * l + 1];
  vals[l] = (int32_t)l;
}

// The table is read as F32: plane i of `Yhi` holds Y_i rounded to f32 (plane 0 = X, exact), `Ylo` what the
// rounding dropped (planes 1..K).  x_i[s] = Y_i[s] - Â^i[s,d]·X[d] is formed in f64 from the f32 value and
// the f64 scalar: its error is the table's rounding, 2^-24·|Y_i[s,c]| — harmless unless the subtraction
// removes most of the row (a leaf s hanging off d: the reference's masked row is EXACTLY zero there).  So
// every (operator, endpoint) row keeps the largest |Y| and the largest |result| it met; where the result's
// norm is below a quarter of Y's (then the rounding could reach 2.4e-7 of it) the row is formed again from
// hi + lo — 48 bits, f64 for this purpose: an exact zero of the reference comes out below 1e-14.  The
// decision is uniform over the wavefront and depends on the link alone.  Half the bytes of the f64 table
// per read; the second pass touches a few per cent of the rows (links with a degree-1 endpoint).
#ifndef S3GRL_SOP_ROWS_UNROLL
#define S3GRL_SOP_ROWS_UNROLL 3   // column trips in flight per wavefront (build-time tuning hook)
#endif
template <int KT>
__global__ __launch_bounds__(256) void sop_rows_kernel(
    const int64_t* __restrict__ links, const int32_t* __restrict__ order,
    const int32_t* __restrict__ partner, const int32_t* __restrict__ mirror_of, int64_t L,
    const float* __restrict__ Yhi, const float* __restrict__ Ylo, int64_t N,
    int64_t ldh, int F, int K, const double* __restrict__ scal, float* __restrict__ rows) {
  const int lane = threadIdx.x & 63;
  const int64_t pos = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (pos >= L) return;
  const int64_t l = order ? (int64_t)order[pos] : pos;
  // Reversed duplicates (both directions of a train edge are in the reference's list): the link
  // (d,s) has the rows of (s,d) swapped (Â^i is symmetric) — the primary writes both outputs.
  if (partner && partner[l] >= 0) return;
  const int64_t m = mirror_of ? (int64_t)mirror_of[l] : -1;
  const int KK = KT > 0 ? KT : K;
  constexpr int KA = KT > 0 ? KT : kMaxSignK;
  const int64_t s = links[2 * l], d = links[2 * l + 1];
  const int Fp = F + 1;
  float* __restrict__ out_s = rows + (2 * l) * (int64_t)(KK + 1) * Fp;
  float* __restrict__ out_d = out_s + (int64_t)(KK + 1) * Fp;
  float* __restrict__ mir_s = m >= 0 ? rows + (2 * m) * (int64_t)(KK + 1) * Fp : nullptr;   // = our dst row
  float* __restrict__ mir_d = m >= 0 ? mir_s + (int64_t)(KK + 1) * Fp : nullptr;            // = our src row
  const float* __restrict__ h_s = Yhi + s * ldh;
  const float* __restrict__ h_d = Yhi + d * ldh;
  const int64_t plane = N * ldh;
  double sd[KA];
  float ymax_s[KA], ymax_d[KA], rmax_s[KA], rmax_d[KA];
#pragma unroll
  for (int i = 0; i < KA; ++i) {
    sd[i] = i < KK ? scal[(l * KK + i) * 3 + 0] : 0.0;
    ymax_s[i] = ymax_d[i] = rmax_s[i] = rmax_d[i] = 0.f;
  }
  // UNR column trips per iteration: all their 2(K+1) row loads are issued before the first result is stored
  // (columns past F read column 0 and are not stored)
  constexpr int UNR = S3GRL_SOP_ROWS_UNROLL;
  for (int c0 = lane; c0 < F; c0 += 64 * UNR) {
    float xs[UNR], xd[UNR], ys[UNR][KA], yd[UNR][KA];
#pragma unroll
    for (int u = 0; u < UNR; ++u) {
      const int c = c0 + 64 * u < F ? c0 + 64 * u : 0;
      xs[u] = h_s[c];
      xd[u] = h_d[c];
#pragma unroll
      for (int i = 0; i < KA; ++i) {
        if (i < KK) {
          ys[u][i] = h_s[(int64_t)(i + 1) * plane + c];
          yd[u][i] = h_d[(int64_t)(i + 1) * plane + c];
        }
      }
    }
#pragma unroll
    for (int u = 0; u < UNR; ++u) {
      const int c = c0 + 64 * u;
      if (c >= F) break;
      out_s[1 + c] = xs[u];   // operator 0: x = [[1|X[s]],[1|X[d]]]  (tuned_SIGN.py:119-125)
      out_d[1 + c] = xd[u];
      if (m >= 0) {
        mir_s[1 + c] = xd[u];
        mir_d[1 + c] = xs[u];
      }
#pragma unroll
      for (int i = 0; i < KA; ++i) {
        if (i < KK) {
          const float vs = (float)((double)ys[u][i] - sd[i] * (double)xd[u]),
                      vd = (float)((double)yd[u][i] - sd[i] * (double)xs[u]);
          ymax_s[i] = fmaxf(ymax_s[i], fabsf(ys[u][i]));
          ymax_d[i] = fmaxf(ymax_d[i], fabsf(yd[u][i]));
          rmax_s[i] = fmaxf(rmax_s[i], fabsf(vs));
          rmax_d[i] = fmaxf(rmax_d[i], fabsf(vd));
          out_s[(int64_t)(i + 1) * Fp + 1 + c] = vs;
          out_d[(int64_t)(i + 1) * Fp + 1 + c] = vd;
          if (m >= 0) {
            mir_s[(int64_t)(i + 1) * Fp + 1 + c] = vd;
            mir_d[(int64_t)(i + 1) * Fp + 1 + c] = vs;
          }
        }
      }
    }
  }
  // rows that lost most of their norm in the subtraction: once more, from hi + lo
#pragma unroll
  for (int i = 0; i < KA; ++i) {
    if (i >= KK) continue;
    float a = ymax_s[i], b = rmax_s[i], e = ymax_d[i], f = rmax_d[i];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
      a = fmaxf(a, __shfl_xor(a, o));
      b = fmaxf(b, __shfl_xor(b, o));
      e = fmaxf(e, __shfl_xor(e, o));
      f = fmaxf(f, __shfl_xor(f, o));
    }
    const bool again_s = b < 0.25f * a, again_d = f < 0.25f * e;   // uniform over the wavefront
    if (!(again_s || again_d)) continue;
    const float* __restrict__ l_s = Ylo + ((int64_t)i * N + s) * ldh;
    const float* __restrict__ l_d = Ylo + ((int64_t)i * N + d) * ldh;
    for (int c = lane; c < F; c += 64) {
      if (again_s) {
        const float v = (float)(((double)h_s[(int64_t)(i + 1) * plane + c] + (double)l_s[c]) - sd[i] * (double)h_d[c]);
        out_s[(int64_t)(i + 1) * Fp + 1 + c] = v;
        if (m >= 0) mir_d[(int64_t)(i + 1) * Fp + 1 + c] = v;
      }
      if (again_d) {
        const float v = (float)(((double)h_d[(int64_t)(i + 1) * plane + c] + (double)l_d[c]) - sd[i] * (double)h_s[c]);
        out_d[(int64_t)(i + 1) * Fp + 1 + c] = v;
        if (m >= 0) mir_s[(int64_t)(i + 1) * Fp + 1 + c] = v;
      }
    }
  }
  if (lane <= KK) {
    const float zs = lane == 0 ? 1.f : (float)scal[(l * KK + (lane - 1)) * 3 + 1];
    const float zd = lane == 0 ? 1.f : (float)scal[(l * KK + (lane - 1)) * 3 + 2];
    out_s[(int64_t)lane * Fp] = zs;
    out_d[(int64_t)lane * Fp] = zd;
    if (m >= 0) {
      mir_s[(int64_t)lane * Fp] = zd;
      mir_d[(int64_t)lane * Fp] = zs;
    }
  }
}

// n_alloc of the scalar kernel when its ball has radius <= 1: deg(s) + deg(d) + 2 bounds the ball,
// no sizing BFS needed
__global__ void sop_ball_bound_kernel(const int32_t* __restrict__ indptr, int64_t N,
                                      const int64_t* __restrict__ links, int64_t L, int radius,
                                      int32_t* __restrict__ n_nodes, int32_t* __restrict__ err_flag) {
  const int64_t l = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (l >= L) return;
  const int64_t s = links[2 * l], d = links[2 * l + 1];
  if (s < 0 || s >= N || d < 0 || d >= N || s == d) {
    atomicMax(err_flag, s == d ? 2 : 1);
    n_nodes[l] = 0;
    return;
  }
  n_nodes[l] = radius == 0 ? 2 : 2 + (indptr[s + 1] - indptr[s]) + (indptr[d + 1] - indptr[d]);
}

// Y (f64, padded ld) -> out fp32 [K, N, F]
__global__ void export_y_kernel(const double* __restrict__ Y, int64_t N, int64_t ldy, int64_t F,
                                int K, float* __restrict__ out) {
  const int64_t total = (int64_t)K * N * F;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total;
       i += (int64_t)gridDim.x * blockDim.x) {
    const int64_t c = i % F, r = (i / F) % N, k = i / (F * N);
    out[i] = (float)Y[((k + 1) * N + r) * ldy + c];
  }
}

static inline int words_for(int64_t N) { return (int)((N + 31) / 32); }

// bins links by LDS need of sop_scalar_kernel (4 + 16·HB bytes per ball node)
__global__ void sop_classify_kernel(const int32_t* __restrict__ n_nodes, int64_t L, int per_node,
                                    int b0, int b1, int b2, const int32_t* __restrict__ partner,
                                    int32_t* __restrict__ class_count,
                                    int32_t* __restrict__ class_list) {
  const int64_t l = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int need = l < L ? n_nodes[l] * per_node + 64 : 0;
  // a reversed duplicate (d,s) of a link (s,d) of the same list needs no scalars of its own
  const bool skip = l >= L || (partner && partner[l] >= 0);
  const int c = skip ? -1 : (need <= b0 ? 0 : (need <= b1 ? 1 : (need <= b2 ? 2 : 3)));
  // one atomic per (wave, class) instead of one per link
  const int lane = threadIdx.x & 63;
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const unsigned long long m = __ballot(c == k);
    if (m == 0) continue;
    const int leader = __ffsll((long long)m) - 1;
    int base = 0;
    if (lane == leader) base = atomicAdd(&class_count[k], __popcll(m));
    base = __shfl(base, leader);
    if (c == k && k < 3)
      class_list[(int64_t)k * L + base + __popcll(m & ((1ull << lane) - 1ull))] = (int32_t)l;
  }
}

}  // namespace

}  // namespace s3grl

using namespace s3grl;

extern "C" {

s3grl_status s3grl_sop_destroy(s3grl_sop* s) {
  if (!s) return S3GRL_OK;
  for (void* q : s->owned) s->ctx->arena.release(q);
  delete s;
  return S3GRL_OK;
}

s3grl_status s3grl_sop_create(s3grl_context* ctx, const s3grl_graph* g, const float* X, int64_t ldx,
                              int64_t F, int32_t K, s3grl_sop** out) {
  return s3grl_sop_create_weighted(ctx, g, X, ldx, F, K, nullptr, out);
}

s3grl_status s3grl_sop_create_weighted(s3grl_context* ctx, const s3grl_graph* g, const float* X, int64_t ldx,
                                       int64_t F, int32_t K, const float* multiplicity, s3grl_sop** out) {
  if (!ctx || !g || !out) return S3GRL_ERR_INVALID_ARGUMENT;
  if (!X) {
    set_last_error("node features are None");
    return S3GRL_ERR_NO_FEATURES;
  }
  if (K < 1 || K > kMaxSignK || F <= 0 || ldx < F) {
    set_last_error("sign_k must be in 1..8, F > 0, ldx >= F");
    return S3GRL_ERR_INVALID_ARGUMENT;
  }
  if (g->directed) {
    set_last_error("SoP on a directed graph is not implemented (the closed form relies on a symmetric operator)");
    return S3GRL_ERR_NOT_IMPLEMENTED;
  }
  S3GRL_HIP_TRY(hipSetDevice(ctx->device));
  std::unique_ptr<s3grl_sop, s3grl_status (*)(s3grl_sop*)> s(new s3grl_sop(), s3grl_sop_destroy);
  s->ctx = ctx;
  s->graph = g;
  s->K = K;
  s->F = F;
  s->ldy = (F + 1) / 2 * 2;
  const int64_t N = g->num_nodes;
  void* p = nullptr;
  if (multiplicity && g->nnz > 0) {   // copied: the caller may free its array
    S3GRL_TRY(ctx->arena.alloc((size_t)g->nnz * 4, &p));
    s->owned.push_back(p);
    S3GRL_HIP_TRY(hipMemcpyAsync(p, multiplicity, (size_t)g->nnz * 4, hipMemcpyDeviceToDevice, ctx->stream));
    s->mult = static_cast<float*>(p);
  }
  S3GRL_TRY(ctx->arena.alloc((size_t)N * 8, &p));
  s->owned.push_back(p);
  s->dinv = static_cast<double*>(p);
  S3GRL_TRY(ctx->arena.alloc((size_t)(K + 1) * N * s->ldy * 8, &p));
  s->owned.push_back(p);
  s->Y = static_cast<double*>(p);
  S3GRL_TRY(ctx->arena.alloc((size_t)(K + 1) * N * s->ldy * 4, &p));   // the same table as f32 (plane 0 = X) ...
  s->owned.push_back(p);
  s->Yhi = static_cast<float*>(p);
  S3GRL_TRY(ctx->arena.alloc((size_t)K * N * s->ldy * 4, &p));         // ... and what the rounding dropped, planes 1..K
  s->owned.push_back(p);
  s->Ylo = static_cast<float*>(p);
  if (ctx->profiling) S3GRL_HIP_TRY(hipEventRecord(ctx->ev[0], ctx->stream));
  hipLaunchKernelGGL(global_dinv_kernel, dim3((unsigned)((N + 255) / 256)), dim3(256), 0, ctx->stream,
                     g->indptr, s->mult, N, s->dinv);
  const int64_t total = N * s->ldy;
  hipLaunchKernelGGL(to_f64_pad_kernel, dim3((unsigned)std::min<int64_t>((total + 255) / 256, 8192)),
                     dim3(256), 0, ctx->stream, X, ldx, N, F, s->Y, s->Yhi, s->ldy);
  const int spmm_tiles = (int)((s->ldy + 255) / 256);
  if (ctx->profiling) S3GRL_HIP_TRY(hipEventRecord(ctx->ev[5], ctx->stream));
  for (int i = 1; i <= K; ++i)
    hipLaunchKernelGGL(spmm_norm_kernel, dim3((unsigned)((N * spmm_tiles + 3) / 4)), dim3(256), 0, ctx->stream,
                       g->indptr, g->indices, s->dinv, s->Y + (int64_t)(i - 1) * N * s->ldy,
                       s->Y + (int64_t)i * N * s->ldy, s->Yhi + (int64_t)i * N * s->ldy,
                       s->Ylo + (int64_t)(i - 1) * N * s->ldy, N, s->ldy, spmm_tiles, s->mult);
  S3GRL_HIP_TRY(hipGetLastError());
  if (ctx->profiling) {
    S3GRL_HIP_TRY(hipEventRecord(ctx->ev[1], ctx->stream));
    S3GRL_HIP_TRY(hipEventSynchronize(ctx->ev[1]));
    float ms = 0;
    S3GRL_HIP_TRY(hipEventElapsedTime(&ms, ctx->ev[0], ctx->ev[1]));
    ctx->timings[3] += ms;
    S3GRL_HIP_TRY(hipEventElapsedTime(&ms, ctx->ev[5], ctx->ev[1]));
    ctx->timings[9] += ms;
    ctx->timings[10] += 1.0;
  }
  *out = s.release();
  return S3GRL_OK;
}

s3grl_status s3grl_sop_features(s3grl_context* ctx, const s3grl_sop* s, float* out) {
  if (!ctx || !s || !out) return S3GRL_ERR_INVALID_ARGUMENT;
  S3GRL_HIP_TRY(hipSetDevice(ctx->device));
  const int64_t N = s->graph->num_nodes, total = (int64_t)s->K * N * s->F;
  hipLaunchKernelGGL(export_y_kernel, dim3((unsigned)std::min<int64_t>((total + 255) / 256, 8192)),
                     dim3(256), 0, ctx->stream, s->Y, N, s->ldy, s->F, s->K, out);
  S3GRL_HIP_TRY(hipGetLastError());
  return S3GRL_OK;
}

s3grl_status s3grl_sop_run(s3grl_context* ctx, const s3grl_sop* s, const int64_t* links, int64_t L,
                           float* rows) {
  if (!ctx || !s || L < 0 || (L > 0 && (!links || !rows))) return S3GRL_ERR_INVALID_ARGUMENT;
  if (L == 0) return S3GRL_OK;
  if (L >= (int64_t)INT32_MAX / 4) return S3GRL_ERR_INVALID_ARGUMENT;
  S3GRL_HIP_TRY(hipSetDevice(ctx->device));
  const s3grl_graph* g = s->graph;
  const int K = s->K, HB = (K + 1) / 2;
  const int RB = (K & 1) ? HB - 1 : HB;   // radius of the ball the scalars are formed in (see the kernel)
  const int W = words_for(g->num_nodes);
  std::vector<void*> tmp;
  struct Rel {
    s3grl_context* c;
    std::vector<void*>* v;
    ~Rel() {
      for (void* q : *v) c->arena.release(q);
    }
  } rel{ctx, &tmp};
  auto alloc = [&](size_t bytes, void** q) -> s3grl_status {
    S3GRL_TRY(ctx->arena.alloc(bytes, q));
    tmp.push_back(*q);
    return S3GRL_OK;
  };
  int32_t *n_nodes, *p_nodes, *n_rows, *n_jobs, *lvl_max, *class_list;
  int64_t *node_off, *scan_ws;
  double* scal;
  void* q;
  // (a cold arena: the per-link arrays below out of one block)
  S3GRL_TRY(ctx->arena.reserve((size_t)L * (size_t)(128 + 24 * K) + (size_t)mirror_table_slots(L) * 12 + (8u << 20)));
  S3GRL_TRY(alloc((size_t)L * 4, &q)); n_nodes = (int32_t*)q;
  S3GRL_TRY(alloc((size_t)L * 4, &q)); p_nodes = (int32_t*)q;
  S3GRL_TRY(alloc((size_t)L * 4, &q)); n_rows = (int32_t*)q;
  S3GRL_TRY(alloc((size_t)L * 4, &q)); n_jobs = (int32_t*)q;
  S3GRL_TRY(alloc((size_t)L * 4, &q)); lvl_max = (int32_t*)q;
  S3GRL_TRY(alloc((size_t)L * 3 * 4, &q)); class_list = (int32_t*)q;
  S3GRL_TRY(alloc((size_t)(L + 1) * 8, &q)); node_off = (int64_t*)q;
  S3GRL_TRY(alloc((size_t)scan_workspace_elems(L) * 8, &q)); scan_ws = (int64_t*)q;
  S3GRL_TRY(alloc((size_t)L * K * 3 * 8, &q)); scal = (double*)q;
  // launch order of the row kernel (see sop_order_keys_kernel)
  int32_t* order = nullptr;
  uint64_t *sort_ka = nullptr, *sort_kb = nullptr;
  int32_t* sort_va = nullptr;
  void* sort_tmp = nullptr;
  size_t sort_bytes = 0;
  if (!getenv("S3GRL_SOP_UNSORTED")) {   // comparison hook
    uint64_t *ka, *kb;
    int32_t *va, *vb;
    S3GRL_TRY(alloc((size_t)L * 8, &q)); ka = (uint64_t*)q;
    S3GRL_TRY(alloc((size_t)L * 8, &q)); kb = (uint64_t*)q;
    S3GRL_TRY(alloc((size_t)L * 4, &q)); va = (int32_t*)q;
    S3GRL_TRY(alloc((size_t)L * 4, &q)); vb = (int32_t*)q;
    S3GRL_TRY(sort_pairs_u64_i32_bytes(ctx, (size_t)L, &sort_bytes));
    S3GRL_TRY(alloc(std::max<size_t>(sort_bytes, 16), &sort_tmp));
    order = vb;
    sort_ka = ka, sort_kb = kb, sort_va = va;
  }

  int64_t* ds = ctx->d_scalars;
  int64_t* hs = ctx->h_scalars;
  int32_t* class_count = reinterpret_cast<int32_t*>(ds + 8);
  if (ctx->profiling) S3GRL_HIP_TRY(hipEventRecord(ctx->ev[0], ctx->stream));
  S3GRL_HIP_TRY(hipMemsetAsync(ds, 0, 32 * sizeof(int64_t), ctx->stream));
  int32_t *partner = nullptr, *mirror_of = nullptr;
  if (!getenv("S3GRL_NO_MIRROR")) {
    uint64_t* mk;
    int32_t* mv;
    const int64_t slots = mirror_table_slots(L);
    S3GRL_TRY(alloc((size_t)slots * 8, &q)); mk = (uint64_t*)q;
    S3GRL_TRY(alloc((size_t)slots * 4, &q)); mv = (int32_t*)q;
    S3GRL_TRY(alloc((size_t)L * 4, &q)); partner = (int32_t*)q;
    S3GRL_TRY(alloc((size_t)L * 4, &q)); mirror_of = (int32_t*)q;
    S3GRL_TRY(launch_find_mirrors(ctx, links, L, g->num_nodes, mk, mv, slots, partner, mirror_of, ds + 7));
  }
  if (order) {
    hipLaunchKernelGGL(sop_order_keys_kernel, dim3((unsigned)((L + 255) / 256)), dim3(256), 0, ctx->stream, links,
                       L, sort_ka, sort_va);
    S3GRL_HIP_TRY(hipGetLastError());
    S3GRL_TRY(sort_pairs_u64_i32(ctx, sort_tmp, sort_bytes, sort_ka, sort_kb, sort_va, order, (size_t)L));
  }
  // capacity of every link's ball: the sizing BFS of the PoS path, or a degree bound for radius <= 1
  if (RB <= 1) {
    hipLaunchKernelGGL(sop_ball_bound_kernel, dim3((unsigned)((L + 255) / 256)), dim3(256), 0, ctx->stream,
                       g->indptr, g->num_nodes, links, L, RB, n_nodes, reinterpret_cast<int32_t*>(ds));
    S3GRL_HIP_TRY(hipGetLastError());
  } else {
    S3GRL_TRY(launch_count(ctx, g, links, L, RB, 0, 1, WalkSets{}, nullptr, nullptr, n_nodes, p_nodes, n_rows,
                           n_jobs, lvl_max, reinterpret_cast<int32_t*>(ds), ctx->d_stats));
  }
  S3GRL_TRY(launch_scan_i32_to_i64(ctx, n_nodes, L, node_off, scan_ws));
  // graphs whose three N-bit bitmaps do not fit a CU's LDS (num_nodes > ~327 680): the bitmaps of the scalar
  // kernel live in HBM slices, one per workgroup of a launch, and the class lists run in chunks over them
  const bool ext = 4 * (3 * (int64_t)W + kMaxLevels + 32 + kHubWords + 6 * 16) + 64 + 4096 > 163840 ||
                   getenv("S3GRL_FORCE_EXT_BITMAPS") != nullptr;
  const int fixed = 4 * (3 * (ext ? 0 : W) + kMaxLevels + 32 + kHubWords + 6 * 16) + 64;
  const int per_node = 4 + 16 * HB;
  const int b2 = 163840 - fixed, b1 = std::min(b2, 49152), b0 = std::min(b2, 12288);
  hipLaunchKernelGGL(sop_classify_kernel, dim3((unsigned)((L + 255) / 256)), dim3(256), 0, ctx->stream,
                     n_nodes, L, per_node, b0, b1, b2, partner, class_count, class_list);
  S3GRL_HIP_TRY(hipGetLastError());
  S3GRL_HIP_TRY(hipMemcpyAsync(hs, ds, 16 * 8, hipMemcpyDeviceToHost, ctx->stream));
  S3GRL_HIP_TRY(hipStreamSynchronize(ctx->stream));
  const int err = (int)(hs[0] & 0xffffffff);
  if (err == 2) {
    set_last_error("a link has src == dst");
    return S3GRL_ERR_SELF_LINK;
  }
  if (err == 1) {
    set_last_error("a link endpoint is outside [0, num_nodes)");
    return S3GRL_ERR_INVALID_ARGUMENT;
  }
  int32_t cc[4];
  std::memcpy(cc, hs + 8, sizeof(cc));
  if (cc[3] > 0 || b2 < 1024) {
    set_last_error(std::to_string(cc[3]) + " link(s): the " + std::to_string(RB) +
                   "-hop ball does not fit the 160 KiB LDS-resident SoP scalar path");
    return S3GRL_ERR_GRAPH_TOO_LARGE;
  }
  const int bounds[3] = {b0, b1, b2};
  const bool sparse = (double)g->nnz / (double)std::max<int64_t>(g->num_nodes, 1) <= 6.0;
  uint32_t* ext_slices = nullptr;
  const int64_t ext_stride = ((int64_t)3 * W + 63) / 64 * 64;
  int ext_chunk = 0;
  if (ext) {
    const int most = std::max(cc[0], std::max(cc[1], cc[2]));
    ext_chunk = (int)std::min<int64_t>(std::max(most, 1), std::max<int64_t>(256, ((int64_t)1 << 29) / (ext_stride * 4)));
    S3GRL_TRY(alloc((size_t)ext_stride * 4 * ext_chunk, &q));
    ext_slices = static_cast<uint32_t*>(q);
  }
  for (int c = 2; c >= 0; --c) {
    if (cc[c] == 0) continue;
    const size_t lds = (size_t)fixed + bounds[c];
#define S3GRL_SOP_LAUNCH_(KERN, TT, LIST, COUNT)                                                   \
  do {                                                                                             \
    auto kern = KERN;                                                                              \
    S3GRL_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(kern),                         \
                                      hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));      \
    hipLaunchKernelGGL(kern, dim3((unsigned)(COUNT)), dim3(TT), lds, ctx->stream, g->indptr,       \
                       g->indices, W, s->dinv, links, LIST, node_off, K,                           \
                       HB, RB, g->max_degree > kHubArmDegree ? 1 : 0, scal, s->mult, ext_slices,   \
                       ext_stride);                                                                \
  } while (0)
#define S3GRL_SOP_LAUNCH(TT, GG)                                                                   \
  do {                                                                                             \
    if (!ext) {                                                                                    \
      S3GRL_SOP_LAUNCH_((sop_scalar_kernel<TT, GG, false>), TT, class_list + (int64_t)c * L, cc[c]); \
    } else {                                                                                       \
      for (int base = 0; base < cc[c]; base += ext_chunk)                                          \
        S3GRL_SOP_LAUNCH_((sop_scalar_kernel<TT, GG, true>), TT, class_list + (int64_t)c * L + base, \
                          std::min(ext_chunk, cc[c] - base));                                      \
    }                                                                                              \
  } while (0)
    if (c == 0 && RB <= 1) {   // a ball of a dozen nodes: one wavefront per link, no cross-wave barriers
      if (sparse) S3GRL_SOP_LAUNCH(64, 4); else S3GRL_SOP_LAUNCH(64, 8);
    } else if (c == 0) {
      if (sparse) S3GRL_SOP_LAUNCH(256, 4); else S3GRL_SOP_LAUNCH(256, 8);
    } else {
      if (sparse) S3GRL_SOP_LAUNCH(1024, 4); else S3GRL_SOP_LAUNCH(1024, 8);
    }
#undef S3GRL_SOP_LAUNCH
#undef S3GRL_SOP_LAUNCH_
    S3GRL_HIP_TRY(hipGetLastError());
  }
  if (ctx->profiling) S3GRL_HIP_TRY(hipEventRecord(ctx->ev[5], ctx->stream));
  {
    const dim3 grid((unsigned)((L + 3) / 4)), block(256);
#define S3GRL_ROWS(KT)                                                                            \
  hipLaunchKernelGGL(sop_rows_kernel<KT>, grid, block, 0, ctx->stream, links, order, partner, mirror_of, L, \
                     s->Yhi, s->Ylo, g->num_nodes, \
                     s->ldy, (int)s->F, K, scal, rows)
    switch (K) {   // the common sign_k get their loops unrolled (2K + 2 loads in registers)
      case 1: S3GRL_ROWS(1); break;
      case 2: S3GRL_ROWS(2); break;
      case 3: S3GRL_ROWS(3); break;
      case 4: S3GRL_ROWS(4); break;
      case 5: S3GRL_ROWS(5); break;
      default: S3GRL_ROWS(0); break;
    }
#undef S3GRL_ROWS
  }
  S3GRL_HIP_TRY(hipGetLastError());
  if (ctx->profiling) {
    S3GRL_HIP_TRY(hipEventRecord(ctx->ev[1], ctx->stream));
    S3GRL_HIP_TRY(hipEventSynchronize(ctx->ev[1]));
    float ms = 0;
    S3GRL_HIP_TRY(hipEventElapsedTime(&ms, ctx->ev[0], ctx->ev[1]));
    ctx->timings[4] += ms;
    ctx->timings[7] += 1.0;
    S3GRL_HIP_TRY(hipEventElapsedTime(&ms, ctx->ev[5], ctx->ev[1]));
    ctx->timings[8] += ms;
  }
  return S3GRL_OK;
}

}  // extern "C"

S3GRL_DEFINE_TOUCH(sop)
