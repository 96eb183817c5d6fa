// Epilogue shared by the gather kernels: one wavefront writes the rows of its row pair.
#pragma once

#include "s3grl_internal.hpp"

namespace s3grl {
namespace {

typedef float float4_t __attribute__((ext_vector_type(4)));
// the same with the alignment of a float: output rows start at column 1 of a [K+1][1+F] block, i.e. anywhere
// modulo 16 bytes.  gfx950 runs with unaligned global access enabled (dword-aligned wide accesses are legal),
// so a store through this type is ONE global_store_dwordx4 instead of four dword stores with a 16-byte stride
// between lanes (4x the store instructions of the epilogue, each touching a quarter of every line)
typedef float float4_u __attribute__((ext_vector_type(4), aligned(4)));

// rows are [K+1][1+F] fp32; the +1 label column makes them 4-byte aligned only.  A folded
// reversed duplicate gets the same values with the two rows swapped.  acc[i][r][c]: operator i+1,
// row r of the pair, this lane's c-th float4 of columns coff[c]..coff[c]+3.
// Writes the operator rows I0+1 .. I1 (acc[I0 .. I1-1]); XROW: also operator 0 = X[node];
// ZCOL: also the label column of every operator.  The packed gather writes in two parts (the
// operators that are complete early free their accumulators for the rest of the kernel).
template <int K, int CH, int I0, int I1, bool XROW, bool ZCOL>
__device__ __forceinline__ void write_pair_rows_part(const Job& job, int jid, const float4_t (&acc)[K][2][CH],
                                                     const int (&coff)[CH], const bool (&cok)[CH],
                                                     const float* __restrict__ job_z,
                                                     const float* __restrict__ X, int64_t ldx, int F,
                                                     float* __restrict__ rows, bool first_tile) {
  const int lane = threadIdx.x & 63;
  const int Fp = F + 1;
  const int64_t rstride = (int64_t)(K + 1) * Fp;
  const int nrow = job.node_b >= 0 ? 2 : 1;
  const int ncopy = job.mirror_row >= 0 ? 2 : 1;
#pragma unroll
  for (int r = 0; r < 2; ++r) {   // static trip count: acc[..][r][..] must stay in registers
    if (r >= nrow) break;
    const int node = r == 0 ? job.node_a : job.node_b;
    const float* __restrict__ xr = X + (int64_t)node * ldx;
    for (int m = 0; m < ncopy; ++m) {
      const int64_t orow = m == 0 ? job.out_row + r
                                  : job.mirror_row + (job.mirror_swap ? 1 - r : r);
      float* __restrict__ out = rows + orow * rstride;
#pragma unroll
      for (int c = 0; c < CH; ++c) {
        if (cok[c]) {
          const int nv = min(4, F - coff[c]);  // F need not be a multiple of 4 (X is padded)
          if constexpr (XROW) {
            const float4_t x0 = *reinterpret_cast<const float4_t*>(xr + coff[c]);
            float* o = out + 1 + coff[c];
            if (nv == 4) {
              *reinterpret_cast<float4_u*>(o) = x0;
            } else {
#pragma unroll
              for (int e = 0; e < 4; ++e)
                if (e < nv) o[e] = x0[e];
            }
          }
#pragma unroll
          for (int i = I0; i < I1; ++i) {
            const float4_t a = acc[i][r][c];
            float* oi = out + (int64_t)(i + 1) * Fp + 1 + coff[c];
            if (nv == 4) {
              *reinterpret_cast<float4_u*>(oi) = a;
            } else {
#pragma unroll
              for (int e = 0; e < 4; ++e)
                if (e < nv) oi[e] = a[e];
            }
          }
        }
      }
      if constexpr (ZCOL) {
        if (first_tile && lane <= K) {
          const float z = lane == 0 ? (float)(r == 0 ? job.z_a : job.z_b)
                                    : job_z[((int64_t)jid * K + (lane - 1)) * 2 + r];
          out[(int64_t)lane * Fp] = z;
        }
      }
    }
  }
}

template <int K, int CH>
__device__ __forceinline__ void write_pair_rows(const Job& job, int jid, const float4_t (&acc)[K][2][CH],
                                                const int (&coff)[CH], const bool (&cok)[CH],
                                                const float* __restrict__ job_z,
                                                const float* __restrict__ X, int64_t ldx, int F,
                                                float* __restrict__ rows, bool first_tile) {
  write_pair_rows_part<K, CH, 0, K, true, true>(job, jid, acc, coff, cok, job_z, X, ldx, F, rows, first_tile);
}

}  // namespace
}  // namespace s3grl
