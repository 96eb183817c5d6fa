// Feature half of PoS / PoS Plus on gfx950, dense-row flavour (X without exploitable zeros; the
// packed-row flavour for sparse X is s3grl_packed.hip): bound by the L2 <-> Infinity-Cache fabric.
//
//   rows[r, i, 1 + f] = Σ_w  Â^i[row r, w] · X[w, f]        i = 1..K, both rows of a pair
//   rows[r, 0, :]     = [z_r | X[node_r, :]]
//   rows[r, i, 0]     = Σ_w Â^i[row r, w] z_w                (label column)
//
// replaces the reference's `power_of_a[selected_rows] @ subg_x` (tuned_SIGN.py:175,185 and
// :240,258) and its two dense copies of X_S (utils.py:83, tuned_SIGN.py:179).
//
// One wavefront owns one row pair.  The node-id list and the [K][support] coefficient rows are
// wave-uniform, so they are read through the scalar cache into SGPRs; each lane owns 4·CH feature columns and keeps
// 2K·4·CH fp32 accumulators in VGPRs; every X row of the support is fetched ONCE per pair with
// 16-byte loads (64 lanes × 16 B = one 1 KiB wave-instruction per 256 columns), UNROLL rows
// in flight per wave.  No LDS, no MFMA: 2·2K flop per 4 B of X.
#include <cstdlib>

#include "s3grl_internal.hpp"
#include "s3grl_gather_common.hpp"

namespace s3grl {
namespace {

constexpr int kWavesPerBlock = 4;   // fabric-bound: 1 wave per workgroup measured the same (24.1 vs 23.5 ms)
constexpr int kUnroll = 8;  // rows of X in flight per wavefront (8: 23.1 ms, 4: 23.6 ms on PubMed)

template <int K, int CH>
__global__ __launch_bounds__(kWavesPerBlock * 64) void gather_kernel(
    const Job* __restrict__ jobs, int njobs, const int32_t* __restrict__ c_ids,
    const float* __restrict__ c_coef, const float* __restrict__ job_z, const float* __restrict__ X,
    int64_t ldx, int F, float* __restrict__ rows_out, float* __restrict__ prows,
    const int32_t* __restrict__ job_order) {
  const int lane = threadIdx.x & 63;
  const int wid = __builtin_amdgcn_readfirstlane(blockIdx.x * kWavesPerBlock + (threadIdx.x >> 6));
  if (wid >= njobs) return;
  // (job_order: plans that work in hub order, s3grl_relabel.hip launch_link_order; else list order)
  const int jid = job_order ? __builtin_amdgcn_readfirstlane(job_order[wid]) : wid;
  const int col0 = blockIdx.y * (CH * 256);  // first feature column of this wave's tile
  const Job job = jobs[jid];
  if (job.split == 1) return;   // gathered piece by piece (the entries with split == 2)
  float* __restrict__ rows = job.split == 2 ? prows : rows_out;   // a piece writes partial rows
  const int cnt = __builtin_amdgcn_readfirstlane(job.support);
  const int32_t* __restrict__ ids = c_ids + job.ids_off;
  // coefficients of this pair: [K][cnt] float2 (row a, row b), operator-major
  const float2* __restrict__ cf = reinterpret_cast<const float2*>(c_coef) + job.coef_off;

  // this lane's columns: col0 + (lane + 64 c) * 4 .. +3
  int coff[CH];
  bool cok[CH];
#pragma unroll
  for (int c = 0; c < CH; ++c) {
    coff[c] = col0 + (lane + 64 * c) * 4;
    cok[c] = coff[c] < F;  // rows of X are padded to a multiple of 4 floats by the caller
  }

  float4_t acc[K][2][CH];
#pragma unroll
  for (int i = 0; i < K; ++i)
#pragma unroll
    for (int r = 0; r < 2; ++r)
#pragma unroll
      for (int c = 0; c < CH; ++c) acc[i][r][c] = (float4_t)(0.f);

  int j = 0;
  for (; j + kUnroll <= cnt; j += kUnroll) {
    float4_t v[kUnroll][CH];
    int id[kUnroll];
#pragma unroll
    for (int u = 0; u < kUnroll; ++u) id[u] = ids[j + u];
#pragma unroll
    for (int u = 0; u < kUnroll; ++u) {
      const float* __restrict__ xr = X + (int64_t)id[u] * ldx;
#pragma unroll
      for (int c = 0; c < CH; ++c)
        v[u][c] = cok[c] ? *reinterpret_cast<const float4_t*>(xr + coff[c]) : (float4_t)(0.f);
    }
#pragma unroll
    for (int i = 0; i < K; ++i) {
#pragma unroll
      for (int u = 0; u < kUnroll; ++u) {
        const float2 q = cf[(int64_t)i * cnt + j + u];
#pragma unroll
        for (int c = 0; c < CH; ++c) {
          acc[i][0][c] += q.x * v[u][c];
          acc[i][1][c] += q.y * v[u][c];
        }
      }
    }
  }
  for (; j < cnt; ++j) {
    const float* __restrict__ xr = X + (int64_t)ids[j] * ldx;
#pragma unroll
    for (int c = 0; c < CH; ++c) {
      const float4_t v = cok[c] ? *reinterpret_cast<const float4_t*>(xr + coff[c]) : (float4_t)(0.f);
#pragma unroll
      for (int i = 0; i < K; ++i) {
        const float2 q = cf[(int64_t)i * cnt + j];
        acc[i][0][c] += q.x * v;
        acc[i][1][c] += q.y * v;
      }
    }
  }

  write_pair_rows<K, CH>(job, jid, acc, coff, cok, job_z, X, ldx, F, rows, blockIdx.y == 0);
}

// F <= 128 (node2vec / ogbl-style features): a row of X is at most 512 bytes, i.e. 32 lanes x 16 B —
// with one row per wave-load half of the wavefront would idle in every load and every multiply-add
// (the collab-scale config: 128 features).  Here the two halves of the wavefront take ALTERNATE list
// entries (even entries lanes 0..31, odd entries lanes 32..63): one load instruction fetches two rows,
// ids and coefficients still arrive through the scalar cache and are picked per half with one
// v_cndmask each.  The halves' partial sums are added at the end (lane l + lane l+32): a fixed order.
template <int K>
__global__ __launch_bounds__(kWavesPerBlock * 64) void gather_half_kernel(
    const Job* __restrict__ jobs, int njobs, const int32_t* __restrict__ c_ids,
    const float* __restrict__ c_coef, const float* __restrict__ job_z, const float* __restrict__ X,
    int64_t ldx, int F, float* __restrict__ rows_out, float* __restrict__ prows,
    const int32_t* __restrict__ job_order) {
  constexpr int U = 8;   // pairs of list entries per trip: 16 rows of X in flight per wavefront
  const int lane = threadIdx.x & 63;
  const bool odd = lane >= 32;
  const int wid = __builtin_amdgcn_readfirstlane(blockIdx.x * kWavesPerBlock + (threadIdx.x >> 6));
  if (wid >= njobs) return;
  const int jid = job_order ? __builtin_amdgcn_readfirstlane(job_order[wid]) : wid;
  const Job job = jobs[jid];
  if (job.split == 1) return;   // gathered piece by piece (the entries with split == 2)
  float* __restrict__ rows = job.split == 2 ? prows : rows_out;   // a piece writes partial rows
  const int cnt = __builtin_amdgcn_readfirstlane(job.support);
  const int32_t* __restrict__ ids = c_ids + job.ids_off;
  const float2* __restrict__ cf = reinterpret_cast<const float2*>(c_coef) + job.coef_off;
  int coff[1] = {(lane & 31) * 4};
  const bool col_ok = coff[0] < F;
  float4_t acc[K][2][1];
#pragma unroll
  for (int i = 0; i < K; ++i) {
    acc[i][0][0] = (float4_t)(0.f);
    acc[i][1][0] = (float4_t)(0.f);
  }
  int j = 0;
  for (; j + 2 * U <= cnt; j += 2 * U) {
    int id[U];
    float4_t v[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int a = ids[j + 2 * u], b = ids[j + 2 * u + 1];   // scalar loads
      id[u] = odd ? b : a;
    }
#pragma unroll
    for (int u = 0; u < U; ++u)
      v[u] = col_ok ? *reinterpret_cast<const float4_t*>(X + (int64_t)id[u] * ldx + coff[0]) : (float4_t)(0.f);
#pragma unroll
    for (int i = 0; i < K; ++i) {
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const float2 qa = cf[(int64_t)i * cnt + j + 2 * u], qb = cf[(int64_t)i * cnt + j + 2 * u + 1];
        const float qx = odd ? qb.x : qa.x, qy = odd ? qb.y : qa.y;
        acc[i][0][0] += qx * v[u];
        acc[i][1][0] += qy * v[u];
      }
    }
  }
  for (; j < cnt; j += 2) {   // at most 2U - 1 entries; the odd half may run past the end
    const bool has_b = j + 1 < cnt;
    const int a = ids[j], b = ids[has_b ? j + 1 : j];
    const int id = odd ? b : a;
    const bool live = col_ok && (!odd || has_b);
    const float4_t v = live ? *reinterpret_cast<const float4_t*>(X + (int64_t)id * ldx + coff[0]) : (float4_t)(0.f);
#pragma unroll
    for (int i = 0; i < K; ++i) {
      const float2 qa = cf[(int64_t)i * cnt + j], qb = cf[(int64_t)i * cnt + (has_b ? j + 1 : j)];
      const float qx = odd ? qb.x : qa.x, qy = odd ? qb.y : qa.y;
      acc[i][0][0] += qx * v;
      acc[i][1][0] += qy * v;
    }
  }
  // even + odd entries; afterwards both halves hold the sums, the lower half writes them
#pragma unroll
  for (int i = 0; i < K; ++i)
#pragma unroll
    for (int r = 0; r < 2; ++r)
#pragma unroll
      for (int e = 0; e < 4; ++e) acc[i][r][0][e] += __shfl_xor(acc[i][r][0][e], 32);
  const bool cok[1] = {col_ok && !odd};
  write_pair_rows<K, 1>(job, jid, acc, coff, cok, job_z, X, ldx, F, rows, true);
}

template <int K>
s3grl_status launch_k(s3grl_context* ctx, const Job* jobs, int64_t njobs, const int32_t* c_ids,
                      const float* c_coef, const float* job_z, const float* X, int64_t ldx,
                      int64_t F, float* rows, float* prows, const int32_t* job_order) {
  hipStream_t stream = ctx->stream;
  const unsigned gx = (unsigned)((njobs + kWavesPerBlock - 1) / kWavesPerBlock);
  static const bool no_half = getenv("S3GRL_GATHER_NO_HALF") != nullptr;   // comparison hook
  if (F <= 128 && !no_half) {
    hipLaunchKernelGGL((gather_half_kernel<K>), dim3(gx, 1), dim3(kWavesPerBlock * 64), 0, stream,
                       jobs, (int)njobs, c_ids, c_coef, job_z, X, ldx, (int)F, rows, prows, job_order);
  } else if (F <= 256) {
    hipLaunchKernelGGL((gather_kernel<K, 1>), dim3(gx, 1), dim3(kWavesPerBlock * 64), 0, stream,
                       jobs, (int)njobs, c_ids, c_coef, job_z, X, ldx, (int)F, rows, prows, job_order);
  } else {
    const unsigned gy = (unsigned)((F + 511) / 512);
    hipLaunchKernelGGL((gather_kernel<K, 2>), dim3(gx, gy), dim3(kWavesPerBlock * 64), 0,
                       stream, jobs, (int)njobs, c_ids, c_coef, job_z, X, ldx, (int)F, rows, prows, job_order);
  }
  S3GRL_HIP_TRY(hipGetLastError());
  return S3GRL_OK;
}

}  // namespace

s3grl_status launch_gather(s3grl_context* ctx, const GatherView& v, const int32_t* c_ids,
                           const float* c_coef, int K, const float* X, int64_t ldx, int64_t F, float* rows,
                           bool in_job_order) {
  const Job* jobs = v.jobs;
  const int64_t njobs = v.njobs;
  const float* job_z = v.job_z;
  const int32_t* order = in_job_order ? v.job_order : nullptr;
  if (njobs == 0) return S3GRL_OK;
  switch (K) {
    case 1: return launch_k<1>(ctx, jobs, njobs, c_ids, c_coef, job_z, X, ldx, F, rows, v.prows, order);
    case 2: return launch_k<2>(ctx, jobs, njobs, c_ids, c_coef, job_z, X, ldx, F, rows, v.prows, order);
    case 3: return launch_k<3>(ctx, jobs, njobs, c_ids, c_coef, job_z, X, ldx, F, rows, v.prows, order);
    case 4: return launch_k<4>(ctx, jobs, njobs, c_ids, c_coef, job_z, X, ldx, F, rows, v.prows, order);
    case 5: return launch_k<5>(ctx, jobs, njobs, c_ids, c_coef, job_z, X, ldx, F, rows, v.prows, order);
    case 6: return launch_k<6>(ctx, jobs, njobs, c_ids, c_coef, job_z, X, ldx, F, rows, v.prows, order);
    case 7: return launch_k<7>(ctx, jobs, njobs, c_ids, c_coef, job_z, X, ldx, F, rows, v.prows, order);
    case 8: return launch_k<8>(ctx, jobs, njobs, c_ids, c_coef, job_z, X, ldx, F, rows, v.prows, order);
    default:
      set_last_error("sign_k must be in 1..8");
      return S3GRL_ERR_INVALID_ARGUMENT;
  }
}

}  // namespace s3grl

S3GRL_DEFINE_TOUCH(gather)
