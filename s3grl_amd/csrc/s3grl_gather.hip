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

#ifndef S3GRL_GATHER_WPB
#define S3GRL_GATHER_WPB 4
#endif
constexpr int kWavesPerBlock = S3GRL_GATHER_WPB;   // fabric-bound: 1 wave per workgroup measured the same (24.1 vs 23.5 ms)
constexpr int kUnroll = 8;  // rows of X in flight per wavefront (8: 23.1 ms, 4: 23.6 ms on PubMed)
#ifndef S3GRL_GATHER_NARROW_UNROLL
#define S3GRL_GATHER_NARROW_UNROLL 8
#endif

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

// F <= 128 (node2vec / ogbl-style features: the collab-scale config has 128): a row of X is at most 512 bytes.
// A lane owns TWO columns (8 bytes), so that one wave-load fetches one whole row with every lane at work, the
// row's address is scalar arithmetic (the id comes through the scalar cache: base in SGPRs + a constant lane
// offset, no vector instruction) and its coefficients are scalar operands of the multiply-adds: K·2 packed
// multiply-adds per row and nothing else on the vector unit.
// (Until round 4 the two HALVES of the wavefront took alternate list entries with four columns per lane; picking
// each half's id and coefficients from the scalar pair cost three vector instructions per pick and the 64-bit
// row address six, two of them quarter-rate — 344 vector instructions per 16 rows, 96 of them multiply-adds, the
// vector unit 77 % busy: the kernel was bound by its own address arithmetic.  7.1 -> see DESIGN.md Part II.)
typedef float float2_t __attribute__((ext_vector_type(2)));
typedef float float2_u __attribute__((ext_vector_type(2), aligned(4)));   // (see float4_u in s3grl_gather_common.hpp)

template <int K>
__global__ __launch_bounds__(kWavesPerBlock * 64) void gather_narrow_kernel(
    const Job* __restrict__ jobs, int njobs, const int32_t* __restrict__ c_ids,
    const float* __restrict__ c_coef, const float* __restrict__ job_z, const float* __restrict__ X,
    int64_t ldx, int F, float* __restrict__ rows_out, float* __restrict__ prows,
    const int32_t* __restrict__ job_order) {
  // rows of X in flight per wavefront: as many as leave a group's 2·K·U coefficient scalars (and 2U ids) in SGPRs
  constexpr int U = K <= 3 ? S3GRL_GATHER_NARROW_UNROLL : (K == 4 ? 6 : (K == 5 ? 5 : (K == 6 ? 4 : 3)));
  const int lane = threadIdx.x & 63;
  const int wid = __builtin_amdgcn_readfirstlane(blockIdx.x * kWavesPerBlock + (threadIdx.x >> 6));
  if (wid >= njobs) return;
  const int jid = job_order ? __builtin_amdgcn_readfirstlane(job_order[wid]) : wid;
  const Job job = jobs[jid];
  if (job.split == 1) return;   // gathered piece by piece (the entries with split == 2)
  float* __restrict__ rows = job.split == 2 ? prows : rows_out;   // a piece writes partial rows
  const int cnt = __builtin_amdgcn_readfirstlane(job.support);
  const int32_t* __restrict__ ids = c_ids + job.ids_off;
  const float2* __restrict__ cf = reinterpret_cast<const float2*>(c_coef) + job.coef_off;
  const int coff = lane * 2;
  const bool col_ok = coff < F;   // (rows of X are padded to a multiple of 4 floats: both columns are readable)
  // every load unconditional: a lane beyond F reads the row's first columns and never writes its sums
  const uint32_t voff = (uint32_t)(col_ok ? coff : 0) * 4u;
  float2_t acc[K][2];
#pragma unroll
  for (int i = 0; i < K; ++i) {
    acc[i][0] = (float2_t)(0.f);
    acc[i][1] = (float2_t)(0.f);
  }
  const uint32_t row_bytes = (uint32_t)ldx * 4u;   // (launch_k: ldx * 4 < 2^32)
  auto row_of = [&](int id) -> float2_t {
    // the row's base on the SCALAR unit (unsigned 32 x 32 -> 64 bits; the signed 64-bit product cost ten scalar
    // instructions per row) and kept there: left to itself the compiler folds product and lane offset into ONE
    // quarter-rate v_mad_u64_u32 per row — a third of the loop's vector time.  With the product opaque the load
    // takes the scalar base + 32-bit lane offset form: no vector instruction per row.
    uint64_t prod = (uint64_t)(uint32_t)id * row_bytes;
    asm volatile("" : "+s"(prod));
    const char* base = reinterpret_cast<const char*>(X) + prod;
    uint32_t vo = voff;            // (opaque as well: hoisted, X + voff becomes a 64-bit vector add per row again)
    asm volatile("" : "+v"(vo));
    return *reinterpret_cast<const float2_t*>(base + vo);
  };
  // One scalar round trip per group instead of two: the ids of group g + 1 are fetched together with the
  // coefficients of group g, so that a group's row loads go out at the top of its step with nothing to wait for
  // (measured on config 5 — the kernel waits 74 % of its wave cycles: ids, then rows + coefficients, per group).
  // (Two row buffers with the loads of group g + 1 under the multiply-adds of g: 6.4 against 6.2 ms; 12 or 16
  // rows per group spill the scalar registers that hold a group's coefficients: 6.7 / 6.9 ms.)
  // (The partial last group as one masked group instead of a row at a time, and operator 0's rows fetched up
  // front: 6.3 against 6.2 ms — not kept.)
  const int nrow = job.node_b >= 0 ? 2 : 1;
  const int ng = cnt / U;   // full groups: contiguous, unclamped scalar loads (wide s_load)
  if (ng > 0) {
    int id[U];
#pragma unroll
    for (int u = 0; u < U; ++u) id[u] = ids[u];
    for (int g = 0; g < ng; ++g) {
      float2_t v[U];
#pragma unroll
      for (int u = 0; u < U; ++u) v[u] = row_of(id[u]);
      __builtin_amdgcn_sched_barrier(0);
      const int jn = min(g + 1, ng - 1) * U;   // (the last group fetches its own ids again)
#pragma unroll
      for (int u = 0; u < U; ++u) id[u] = ids[jn + u];
      float2 q[K][U];
#pragma unroll
      for (int i = 0; i < K; ++i)
#pragma unroll
        for (int u = 0; u < U; ++u) q[i][u] = cf[(int64_t)i * cnt + g * U + u];
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int i = 0; i < K; ++i) {
#pragma unroll
        for (int u = 0; u < U; ++u) {
          acc[i][0] += q[i][u].x * v[u];
          acc[i][1] += q[i][u].y * v[u];
        }
      }
      __builtin_amdgcn_sched_barrier(0);
    }
  }
  for (int j = ng * U; j < cnt; ++j) {
    const float2_t v = row_of(ids[j]);
#pragma unroll
    for (int i = 0; i < K; ++i) {
      const float2 q = cf[(int64_t)i * cnt + j];
      acc[i][0] += q.x * v;
      acc[i][1] += q.y * v;
    }
  }
  // rows are [K+1][1+F] fp32 (see write_pair_rows_part): operator 0 = X[node], operators 1..K, the label column
  const int Fp = F + 1;
  const int64_t rstride = (int64_t)(K + 1) * Fp;
  const int ncopy = job.mirror_row >= 0 ? 2 : 1;
  const int nv = min(2, F - coff);
#pragma unroll
  for (int r = 0; r < 2; ++r) {
    if (r >= nrow) break;
    const float2_t x0 = row_of(r == 0 ? job.node_a : job.node_b);
    for (int m = 0; m < ncopy; ++m) {
      const int64_t orow = m == 0 ? job.out_row + r : job.mirror_row + (job.mirror_swap ? 1 - r : r);
      float* __restrict__ out = rows + orow * rstride;
      if (col_ok) {
        float* o = out + 1 + coff;
        if (nv == 2) {
          *reinterpret_cast<float2_u*>(o) = x0;
#pragma unroll
          for (int i = 0; i < K; ++i) *reinterpret_cast<float2_u*>(o + (int64_t)(i + 1) * Fp) = acc[i][r];
        } else {
          o[0] = x0[0];
#pragma unroll
          for (int i = 0; i < K; ++i) o[(int64_t)(i + 1) * Fp] = acc[i][r][0];
        }
      }
      if (lane <= K) {
        const float z = lane == 0 ? (float)(r == 0 ? job.z_a : job.z_b)
                                  : job_z[((int64_t)jid * K + (lane - 1)) * 2 + r];
        out[(int64_t)lane * Fp] = z;
      }
    }
  }
}

template <int K>
s3grl_status launch_k(s3grl_context* ctx, const Job* jobs, int64_t njobs, const int32_t* c_ids,
                      const float* c_coef, const float* job_z, const float* X, int64_t ldx,
                      int64_t F, float* rows, float* prows, const int32_t* job_order) {
  hipStream_t stream = ctx->stream;
  const unsigned gx = (unsigned)((njobs + kWavesPerBlock - 1) / kWavesPerBlock);
  static const bool no_half = getenv("S3GRL_GATHER_NO_HALF") != nullptr;   // comparison hook
  if (F <= 128 && !no_half && ldx < ((int64_t)1 << 30)) {
    hipLaunchKernelGGL((gather_narrow_kernel<K>), dim3(gx, 1), dim3(kWavesPerBlock * 64), 0, stream,
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
