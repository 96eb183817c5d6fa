// Packed-row feature operand and its gather kernel, gfx950.
//
// The reference hands its operators a dense fp32 X, but the matrices it is run on are row-
// normalised bag-of-words / TF-IDF / one-hot rows (sgrl_link_pred.py:851,961-963): PubMed has
// ~50 non-zeros in 500 columns, Cora ~18 in 1433.  The dense gather (s3grl_gather.hip) is bound
// by the bytes it pulls through the fabric, and two thirds of its 16-byte lane loads fetch four
// zeros.  Here X is stored a second time with the all-zero 16-byte chunks squeezed out:
//
//   pk_hdr[tile][row] = { 128-bit mask of the non-zero chunks of the 512-column tile, offset }
//   pk_data           = one chunk of zeros, then the non-zero chunks in (tile, row, chunk) order
//
// The kernel is the dense one with a different fetch: a lane still OWNS chunk `lane` and chunk
// `64 + lane` of the tile and keeps their 2K accumulators in registers (no LDS, no atomics —
// the (column, value)-pair format of s3grl_features.hip lost to the dense kernel exactly there);
// the row's header arrives through the scalar cache (the row id is wave-uniform), the lane's
// chunk sits at offset + popcount(mask bits below the lane) = one v_mbcnt pair, and a lane whose
// mask bit is clear reads a shared chunk of zeros instead.  c·0 adds nothing, so the sums are the dense
// kernel's sums bit for bit.  PubMed: 33 % of the chunks are non-zero -> 0.7 KB instead of 2 KB
// per subgraph node.
#include <algorithm>
#include <cstdlib>
#include <type_traits>

#include "s3grl_internal.hpp"
#include "s3grl_gather_common.hpp"

namespace s3grl {
namespace {


constexpr int kTile = 512;          // feature columns per tile = 128 chunks = 2 mask words
// one wavefront per workgroup: the jobs of a workgroup differ in length by an order of magnitude
// (positive vs negative links), and a workgroup holds its registers until its longest job is
// done.  Measured on PubMed K=3: 8 waves per workgroup 16.9 ms, 4: 13.6 ms, 2: 11.1 ms, 1: 10.7 ms.
constexpr int kWavesPerBlock = 1;

__device__ __forceinline__ int below(uint64_t m) {   // set bits of m below this lane
  return __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u));
}

// chunk `j` of tile `tile` of row `row` of the dense X (rows 16-byte aligned, ld % 4 == 0,
// ld >= F rounded up to 4); columns >= F read as zero, whatever a borrowed X holds there
__device__ __forceinline__ float4_t dense_chunk(const float* __restrict__ X, int64_t ld, int F, int64_t row,
                                                int tile, int j) {
  const int col = tile * kTile + j * 4;
  if (col >= F) return (float4_t)(0.f);
  float4_t v = *reinterpret_cast<const float4_t*>(X + row * ld + col);
  if (col + 1 >= F) v.y = 0.f;
  if (col + 2 >= F) v.z = 0.f;
  if (col + 3 >= F) v.w = 0.f;
  return v;
}

__device__ __forceinline__ bool nonzero(float4_t v) {
  return v.x != 0.f || v.y != 0.f || v.z != 0.f || v.w != 0.f;
}

// one wave per (tile, row): number of non-zero chunks
__global__ __launch_bounds__(256) void pk_count_kernel(const float* __restrict__ X, int64_t ld, int F,
                                                       int64_t N, int tiles, int32_t* __restrict__ cnt) {
  const int lane = threadIdx.x & 63;
  const int64_t item = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (item >= N * tiles) return;
  const int tile = (int)(item / N);
  const int64_t row = item - (int64_t)tile * N;
  const uint64_t m0 = __ballot(nonzero(dense_chunk(X, ld, F, row, tile, lane)));
  const uint64_t m1 = __ballot(nonzero(dense_chunk(X, ld, F, row, tile, 64 + lane)));
  if (lane == 0) cnt[item] = __popcll(m0) + __popcll(m1);
}

__global__ __launch_bounds__(256) void pk_fill_kernel(const float* __restrict__ X, int64_t ld, int F,
                                                      int64_t N, int tiles, const int64_t* __restrict__ ptr,
                                                      PackedHdr* __restrict__ hdr,
                                                      float4_t* __restrict__ data) {
  const int lane = threadIdx.x & 63;
  const int64_t item = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (item >= N * tiles) return;
  const int tile = (int)(item / N);
  const int64_t row = item - (int64_t)tile * N;
  const float4_t v0 = dense_chunk(X, ld, F, row, tile, lane);
  const float4_t v1 = dense_chunk(X, ld, F, row, tile, 64 + lane);
  const uint64_t m0 = __ballot(nonzero(v0));
  const uint64_t m1 = __ballot(nonzero(v1));
  const int64_t off = ptr[item] + 1;   // pk_data[0] is a chunk of zeros (what a masked-off lane loads)
  if (item == 0 && lane == 0) data[0] = (float4_t)(0.f);
  if ((m0 >> lane) & 1) data[off + below(m0)] = v0;
  if ((m1 >> lane) & 1) data[off + __popcll(m0) + below(m1)] = v1;
  if (lane == 0) {
    PackedHdr h;
    h.m0 = m0;
    h.m1 = m1;
    h.off = (uint64_t)off;
    h.pad = 0;
    hdr[item] = h;
  }
}

// One wavefront owns one row pair and one 512-column tile; see the file comment.
//
// Schedule.  At 0.7 KB per row the kernel is no longer bound by bytes but by how well the chain
// id -> header -> chunk loads -> multiply-adds overlaps inside a wavefront.  It is software-
// pipelined by hand over groups of U = 4 rows with two register buffers: while the FMAs of group
// g run, the chunk loads of group g+1 are in flight and the ids/headers of group g+2 are on
// their way through the scalar cache.  For the compiler to wait for "all but the newest 2U
// loads" (vmcnt) rather than for all of them, the loads must not sit behind divergent branches:
// a lane whose mask bit is clear loads pk_data[0], a chunk of zeros (one line, L1-resident),
// chosen with one v_cndmask on the wave-uniform mask — so every load is unconditional.  The
// phases are kept apart with scheduling barriers; left alone, the compiler serialises the scalar
// loads (load, wait, use, load, wait, ...).
// Measured on PubMed PoS K=3 (164 000 links), step by step: no pipelining 17.8 ms; two buffers
// 14.4; skipping the operators that cannot reach a row 13.6; one wavefront per workgroup 10.8;
// jobs started largest first 9.7; two phases with a third buffer for the last-operator rows 8.2.
// What lost: U=2 x 4 buffers 16.6 ms (more scalar instructions per row); a third or fifth buffer
// for ALL rows 17.1 / 19.0 ms (156-250 VGPRs: fewer waves per SIMD, and the unrolled body
// outgrows the instruction cache); an XCD-contiguous job mapping 19 ms (load imbalance).
__device__ __forceinline__ uint32_t select_by_mask(uint64_t mask, uint32_t if_set) {
  uint32_t r;   // lane-wise: bit `lane` of the wave-uniform mask ? if_set : 0
  asm("v_cndmask_b32_e64 %0, 0, %1, %2" : "=v"(r) : "v"(if_set), "s"(mask));
  return r;
}

// Chunk fetch of the MASKED variant of the kernel (the default).  The unconditional form above makes
// every lane of a wave-instruction fetch 16 bytes — a masked-off lane the shared chunk of zeros —
// so a row with a third of its chunks populated still pushes 2 x 1 KiB through the CU's texture
// path: 206 M wave-loads x 1 KiB per PubMed launch against a vector-L1 return path of 64 B/clk/CU
// is 6 of the kernel's 8 ms.  Here the chunks are read through a raw BUFFER descriptor over
// pk_data and a masked-off lane is given an offset beyond the buffer: the hardware's range check
// returns zeros for it without a memory request.  Still one unconditional, compiler-visible load
// per lane (no branches, vmcnt tracked by the compiler) — only populated chunks travel.
// (An EXEC-masked global_load in inline assembly does the same on paper; the compiler cannot know
// that such a load is still in flight when it copies or reuses the destination registers, and the
// kernel faulted at PubMed scale.)
typedef unsigned int uint4_t __attribute__((ext_vector_type(4)));
constexpr uint32_t kOobOffset = 0x80000000u;   // >= any pk_data size (build_packed_rows keeps it below 2 GiB)

__device__ __forceinline__ uint32_t select_or_oob(uint64_t mask, uint32_t if_set, uint32_t oob) {
  uint32_t r;   // lane-wise: bit `lane` of the wave-uniform mask ? if_set : oob
  asm("v_cndmask_b32_e64 %0, %3, %1, %2" : "=v"(r) : "v"(if_set), "s"(mask), "v"(oob));
  return r;
}

// MINNB = 2: a plan whose last two operators reach the whole list in every job (sign_k - 1 >=
// num_hops): the variant that holds every operator's accumulators at once (NB = 1) is left out, and
// with it its registers — PubMed sign_k = 5: 128 instead of 166 VGPRs, four waves per SIMD.
// S3GRL_GATHER_WAVES (build-time experiment hook): ask the compiler for that many waves per SIMD
#ifdef S3GRL_GATHER_WAVES
#define S3GRL_GATHER_OCC __attribute__((amdgpu_waves_per_eu(S3GRL_GATHER_WAVES, S3GRL_GATHER_WAVES)))
#else
#define S3GRL_GATHER_OCC
#endif
template <int K, bool MASKED, int MINNB>
__global__ __launch_bounds__(kWavesPerBlock * 64) S3GRL_GATHER_OCC void gather_packed_kernel(
    const Job* __restrict__ jobs, int njobs, const int32_t* __restrict__ c_ids,
    const float* __restrict__ c_coef, const float* __restrict__ job_z,
    const int32_t* __restrict__ job_lim, const int32_t* __restrict__ job_order,
    const PackedHdr* __restrict__ hdr,
    const float4_t* __restrict__ data, uint32_t data_bytes, int64_t N, const float* __restrict__ X,
    int64_t ldx, int F, float* __restrict__ rows_out, float* __restrict__ prows) {
  constexpr int CH = 2;
  constexpr int U = 4;   // rows per group
  const int lane = threadIdx.x & 63;
  const int wid = __builtin_amdgcn_readfirstlane(blockIdx.x * kWavesPerBlock + (threadIdx.x >> 6));
  if (wid >= njobs) return;
  const int jid = __builtin_amdgcn_readfirstlane(job_order[wid]);   // longest jobs start first
  const int col0 = blockIdx.y * kTile;
  const Job job = jobs[jid];
  if (job.split == 1) return;   // gathered piece by piece (the entries with split == 2)
  float* __restrict__ rows = job.split == 2 ? prows : rows_out;   // a piece writes partial rows
  const int cnt = __builtin_amdgcn_readfirstlane(job.support);
  const uint32_t* __restrict__ uid = reinterpret_cast<const uint32_t*>(c_ids + job.ids_off);
  const float2* __restrict__ cf = reinterpret_cast<const float2*>(c_coef) + job.coef_off;
  const PackedHdr* __restrict__ th = hdr + (int64_t)blockIdx.y * N;
  const char* __restrict__ bytes = reinterpret_cast<const char*>(data);
  // raw buffer over pk_data (dword 3 = 0x00020000: gfx9-family untyped 32-bit data format), stride 0:
  // an offset at or beyond data_bytes is out of range and reads as zero
  const __amdgpu_buffer_rsrc_t rsrc =
      __builtin_amdgcn_make_buffer_rsrc(const_cast<float4_t*>(data), 0, (int)data_bytes, 0x00020000);
  const uint32_t oobv = kOobOffset;
  // Operator i+1 has no non-zero coefficient at list positions >= lim[i] (a walk of i+1 steps
  // stays within i+1 hops, and the list is hop-major), lim non-decreasing: its multiply-adds are
  // skipped there.  On PubMed (3 hops, K = 3) four fifths of the rows only feed the last operator.
  int lim[K];
#pragma unroll
  for (int i = 0; i < K; ++i) lim[i] = __builtin_amdgcn_readfirstlane(job_lim[(int64_t)jid * K + i]);

  int coff[CH];
  bool cok[CH];
#pragma unroll
  for (int c = 0; c < CH; ++c) {
    coff[c] = col0 + (lane + 64 * c) * 4;
    cok[c] = coff[c] < F;
  }
  float4_t acc[K][2][CH];
#pragma unroll
  for (int i = 0; i < K; ++i)
#pragma unroll
    for (int r = 0; r < 2; ++r)
#pragma unroll
      for (int c = 0; c < CH; ++c) acc[i][r][c] = (float4_t)(0.f);

  // a header through the scalar cache: the row id as a 32-bit byte offset on the uniform base (s_load with an
  // offset register: one shift per row instead of a 64-bit shift and add; build_packed_rows keeps N * 32 < 2^32)
  auto hdr_of = [&](uint32_t id) __attribute__((always_inline)) -> PackedHdr {
    static_assert(sizeof(PackedHdr) == 32, "the shift below");
    return *reinterpret_cast<const PackedHdr*>(reinterpret_cast<const char*>(th) + (id << 5));
  };
  auto load_hdrs = [&](int g, PackedHdr(&h)[U]) __attribute__((always_inline)) {   // scalar: ids (one wide load), then headers
    uint32_t id[U];
#pragma unroll
    for (int u = 0; u < U; ++u) id[u] = uid[g * U + u];
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int u = 0; u < U; ++u) h[u] = hdr_of(id[u]);
    __builtin_amdgcn_sched_barrier(0);
  };
  // the same in two halves for the steady-state loops: the ids of a group are fetched one step
  // before its headers, so that no step waits for a scalar round trip it has just started
  auto load_ids = [&](int g, uint32_t(&id)[U]) __attribute__((always_inline)) {
#pragma unroll
    for (int u = 0; u < U; ++u) id[u] = uid[g * U + u];
    __builtin_amdgcn_sched_barrier(0);
  };
  auto hdrs_from = [&](const uint32_t(&id)[U], PackedHdr(&h)[U]) __attribute__((always_inline)) {
#pragma unroll
    for (int u = 0; u < U; ++u) h[u] = hdr_of(id[u]);
    __builtin_amdgcn_sched_barrier(0);
  };
  // phase B's form: the ids pass through an opaque statement — without it the header addresses
  // (plain arithmetic on the ids) are computed right behind the id load of the step before, and
  // the wait moves there with them
  auto hdrs_from_late = [&](const uint32_t(&id)[U], PackedHdr(&h)[U]) __attribute__((always_inline)) {
    uint32_t idv[U];
#pragma unroll
    for (int u = 0; u < U; ++u) idv[u] = id[u];
    asm volatile("" : "+s"(idv[0]), "+s"(idv[1]), "+s"(idv[2]), "+s"(idv[3]));
    static_assert(U == 4, "the opaque statement above lists four ids");
#pragma unroll
    for (int u = 0; u < U; ++u) h[u] = hdr_of(idv[u]);
    __builtin_amdgcn_sched_barrier(0);
  };
  // (valid = false: a group beyond the end of the list — every lane reads as zero)
  auto issue = [&](const PackedHdr(&h)[U], float4_t(&v)[U][CH], bool valid = true) __attribute__((always_inline)) {   // 2U unconditional loads
#pragma unroll
    for (int u = 0; u < U; ++u) {
      // MASKED: a group beyond the end gets a base whose every chunk address is out of range (one scalar select
      // per row instead of one per mask word)
      const uint32_t base = MASKED ? (valid ? (uint32_t)h[u].off : (kOobOffset >> 4)) : (uint32_t)h[u].off;
      const uint32_t a0 = (base + (uint32_t)below(h[u].m0)) << 4;
      const uint32_t a1 = (base + (uint32_t)__popcll(h[u].m0) + (uint32_t)below(h[u].m1)) << 4;
      if constexpr (MASKED) {
        v[u][0] = __builtin_bit_cast(float4_t, __builtin_amdgcn_raw_buffer_load_b128(
                                                   rsrc, (int)select_or_oob(h[u].m0, a0, oobv), 0, 0));
        v[u][1] = __builtin_bit_cast(float4_t, __builtin_amdgcn_raw_buffer_load_b128(
                                                   rsrc, (int)select_or_oob(h[u].m1, a1, oobv), 0, 0));
      } else {
        v[u][0] = *reinterpret_cast<const float4_t*>(bytes + select_by_mask(valid ? h[u].m0 : 0ull, a0));
        v[u][1] = *reinterpret_cast<const float4_t*>(bytes + select_by_mask(valid ? h[u].m1 : 0ull, a1));
      }
    }
    __builtin_amdgcn_sched_barrier(0);
  };
  auto fma_from = [&](auto first, auto last, int g, const float4_t(&v)[U][CH]) __attribute__((always_inline)) {   // operators first+1 .. last
    constexpr int I0 = decltype(first)::value, I1 = decltype(last)::value;
#pragma unroll
    for (int i = I0; i < I1; ++i) {
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const float2 q = cf[(int64_t)i * cnt + g * U + u];
#pragma unroll
        for (int c = 0; c < CH; ++c) {
          acc[i][0][c] += q.x * v[u][c];
          acc[i][1][c] += q.y * v[u][c];
        }
      }
    }
  };
  // Two phases.  The last NB operators all reach the whole list (lim[K-NB .. K-1] == cnt; NB = 1 when
  // sign_k - 1 < num_hops, more when the operators outrun the BFS depth; capped at 3).  A: the list
  // prefix the operators before them can reach (rows < lim[K-NB-1]), all accumulators live, two
  // chunk buffers.  Then the rows of operators 1..K-NB are complete and are written out, which
  // frees their accumulator registers.  B: the rest of the list feeds the last NB operators only;
  // with NB = 1 the freed registers hold a third chunk buffer (8 rows in flight under the
  // multiply-adds of 4), with NB = 2, 3 the rows do 2 or 3 operators' multiply-adds instead of K
  // (PubMed sign_k = 5: four fifths of the rows, 3 operators instead of 5).
  auto tail_rows = [&](auto first, auto last, int j0) __attribute__((always_inline)) {   // at most U-1 rows, operators first+1 .. last
    constexpr int I0 = decltype(first)::value, I1 = decltype(last)::value;
    for (int j = j0; j < cnt; ++j) {
      const PackedHdr h = hdr_of(uid[j]);
      const uint32_t base = (uint32_t)h.off;
      const uint32_t a0 = (base + (uint32_t)below(h.m0)) << 4;
      const uint32_t a1 = (base + (uint32_t)__popcll(h.m0) + (uint32_t)below(h.m1)) << 4;
      float4_t v[CH];
      if constexpr (MASKED) {
        v[0] = __builtin_bit_cast(float4_t, __builtin_amdgcn_raw_buffer_load_b128(
                                                rsrc, (int)select_or_oob(h.m0, a0, oobv), 0, 0));
        v[1] = __builtin_bit_cast(float4_t, __builtin_amdgcn_raw_buffer_load_b128(
                                                rsrc, (int)select_or_oob(h.m1, a1, oobv), 0, 0));
      } else {
        v[0] = *reinterpret_cast<const float4_t*>(bytes + select_by_mask(h.m0, a0));
        v[1] = *reinterpret_cast<const float4_t*>(bytes + select_by_mask(h.m1, a1));
      }
#pragma unroll
      for (int i = I0; i < I1; ++i) {
        const float2 q = cf[(int64_t)i * cnt + j];
#pragma unroll
        for (int c = 0; c < CH; ++c) {
          acc[i][0][c] += q.x * v[c];
          acc[i][1][c] += q.y * v[c];
        }
      }
    }
  };

  // Groups [g0, g1) through two chunk buffers, operators first+1 .. K.  Every scalar load of a
  // half-step (next headers, next ids, this group's coefficients) goes out before the chunk loads
  // are issued, so that the one wait at the multiply-adds has the address arithmetic and the other
  // waves between it and them (see pass3 below) — as long as the 2·(K-first)·U coefficient scalars
  // of a group fit next to the headers (up to 3 operators; 5 measured 6 % slower this way).
  auto pass2 = [&](auto first, auto last, int g0, int g1) __attribute__((always_inline)) {
    constexpr int I0 = decltype(first)::value;
    constexpr int NO_RAW = decltype(last)::value - I0;
    constexpr int NO = NO_RAW > 0 ? NO_RAW : 1;   // (an empty operator range, NB == K: nothing to do)
    if (NO_RAW <= 0) return;
    if (g1 <= g0) return;
    int g = g0;
    PackedHdr hA[U], hB[U];
    float4_t vA[U][CH], vB[U][CH];
    uint32_t idn[U];
    load_hdrs(g0, hA);
    issue(hA, vA);
    if (g1 - g0 > 1) load_hdrs(g0 + 1, hB);
    if (g1 - g0 > 2) load_ids(g0 + 2, idn);
    // steady state: vA = group g in flight, hB = headers of group g+1, idn = ids of group g+2
    float2 qa[NO][U];
    auto load_qa = [&](int gq) {
#pragma unroll
      for (int i = 0; i < NO; ++i)
#pragma unroll
        for (int u = 0; u < U; ++u) qa[i][u] = cf[(int64_t)(I0 + i) * cnt + gq * U + u];
      __builtin_amdgcn_sched_barrier(0);
    };
    auto fma_qa = [&](const float4_t(&v)[U][CH]) {
#pragma unroll
      for (int i = 0; i < NO; ++i)
#pragma unroll
        for (int u = 0; u < U; ++u)
#pragma unroll
          for (int c = 0; c < CH; ++c) {
            acc[I0 + i][0][c] += qa[i][u].x * v[u][c];
            acc[I0 + i][1][c] += qa[i][u].y * v[u][c];
          }
      __builtin_amdgcn_sched_barrier(0);
    };
    auto fma_g = [&](int gq, const float4_t(&v)[U][CH]) {
      fma_from(first, last, gq, v);
      __builtin_amdgcn_sched_barrier(0);
    };
    for (; g + 4 < g1; g += 2) {
      if constexpr (NO <= 3) {
        hdrs_from(idn, hA);
        load_ids(g + 3, idn);
        load_qa(g);
        issue(hB, vB);
        fma_qa(vA);
        hdrs_from(idn, hB);
        load_ids(g + 4, idn);
        load_qa(g + 1);
        issue(hA, vA);
        fma_qa(vB);
      } else {
        issue(hB, vB);
        hdrs_from(idn, hA);
        load_ids(g + 3, idn);
        fma_g(g, vA);
        issue(hA, vA);
        hdrs_from(idn, hB);
        load_ids(g + 4, idn);
        fma_g(g + 1, vB);
      }
    }
    fma_g(g, vA);
    ++g;
    for (; g < g1; ++g) {   // at most 3 groups
      load_hdrs(g, hA);
      issue(hA, vA);
      fma_g(g, vA);
    }
  };

  // Groups [g0, g1) through three chunk buffers, the last operator only.
  // Every scalar load is issued a whole step before its first use.  A wavefront can only wait for
  // ALL of its outstanding scalar loads (they return out of order: lgkmcnt(0)), so a step that loads
  // ids, then headers through them, then coefficients at the multiply-adds exposes two scalar round
  // trips (~800 cycles each through the scalar cache to L2) with four waves per SIMD to cover them
  // — that, not bytes or cache misses, was two thirds of a step (with the whole operand resident in
  // L2 the kernel ran 4 % faster).  Step k multiplies group k; at its top the ids of group k+3, the
  // headers of k+2 and the coefficients of k are complete (loaded during step k-1): it loads the
  // headers of k+3, the ids of k+4 and the coefficients of k+1, issues the chunk loads of k+2 and
  // multiplies.  One wait per step, for loads that had a step to arrive.  Scalar buffers alternate
  // (X/Y), chunk buffers rotate over three: six steps per trip.  Groups beyond the end are clamped
  // to the last one and their chunk loads read as zero (out-of-range offsets): the multiply-adds
  // stay unconditional — behind a branch the compiler sinks the scalar loads to their use.
  auto pass3 = [&](int g0, int g1) __attribute__((always_inline)) {
    const int nB = g1 - g0;
    if (nB <= 0) return;
    auto grp = [&](int k) { return g0 + min(k, nB - 1); };
    uint32_t idX[U], idY[U];
    PackedHdr hX[U], hY[U];
    float2 qX[U], qY[U];
    float4_t v0[U][CH], v1[U][CH], v2[U][CH];
    auto load_q = [&](int g, float2(&q)[U]) {
#pragma unroll
      for (int u = 0; u < U; ++u) q[u] = cf[(int64_t)(K - 1) * cnt + g * U + u];
      __builtin_amdgcn_sched_barrier(0);
    };
    auto fma_q = [&](const float2(&q)[U], const float4_t(&v)[U][CH]) {
#pragma unroll
      for (int u = 0; u < U; ++u) {
#pragma unroll
        for (int c = 0; c < CH; ++c) {
          acc[K - 1][0][c] += q[u].x * v[u][c];
          acc[K - 1][1][c] += q[u].y * v[u][c];
        }
      }
      __builtin_amdgcn_sched_barrier(0);
    };
    load_ids(grp(0), idX);
    hdrs_from(idX, hX);
    issue(hX, v0);
    load_ids(grp(1), idX);
    hdrs_from(idX, hY);
    issue(hY, v1, 1 < nB);
    load_ids(grp(2), idX);
    hdrs_from(idX, hX);    // headers of group 2
    load_ids(grp(3), idX);   // ids of group 3
    load_q(grp(0), qX);
#define S3GRL_GATHER_STEP(k, IDr, IDl, Hr, Hl, Qr, Ql, Vissue, Vfma) \
  hdrs_from_late(IDr, Hl);                                            \
  load_ids(grp((k) + 4), IDl);                                        \
  load_q(grp((k) + 1), Ql);                                           \
  issue(Hr, Vissue, (k) + 2 < nB);                                    \
  fma_q(Qr, Vfma);
    for (int k = 0; k < nB; k += 6) {
      S3GRL_GATHER_STEP(k, idX, idY, hX, hY, qX, qY, v2, v0)
      S3GRL_GATHER_STEP(k + 1, idY, idX, hY, hX, qY, qX, v0, v1)
      S3GRL_GATHER_STEP(k + 2, idX, idY, hX, hY, qX, qY, v1, v2)
      S3GRL_GATHER_STEP(k + 3, idY, idX, hY, hX, qY, qX, v2, v0)
      S3GRL_GATHER_STEP(k + 4, idX, idY, hX, hY, qX, qY, v0, v1)
      S3GRL_GATHER_STEP(k + 5, idY, idX, hY, hX, qY, qX, v1, v2)
    }
#undef S3GRL_GATHER_STEP
  };

  const int ngf = cnt / U;   // full groups
  auto run = [&](auto nb_c) __attribute__((always_inline)) {
    constexpr int NB = decltype(nb_c)::value;   // trailing operators that reach the whole list
    using IC0 = std::integral_constant<int, 0>;
    using ICS = std::integral_constant<int, K - NB>;   // first trailing operator
    using ICK = std::integral_constant<int, K>;
    const int limA = NB < K ? lim[NB < K ? K - NB - 1 : 0] : 0;
    const int gA = min(ngf, (limA + U - 1) / U);       // groups [0, gA): the prefix the leading operators reach
    const bool tail_in_A = limA > ngf * U;
    if constexpr (NB == 1) {
      // A: every operator on the prefix; B: the last operator on the rest through three chunk buffers
      pass2(IC0{}, ICK{}, 0, gA);
      if (tail_in_A) tail_rows(IC0{}, ICK{}, ngf * U);
      write_pair_rows_part<K, CH, 0, K - NB, true, false>(job, jid, acc, coff, cok, job_z, X, ldx, F, rows,
                                                          blockIdx.y == 0);
      __builtin_amdgcn_sched_barrier(0);
      pass3(gA, ngf);
      if (!tail_in_A) tail_rows(ICS{}, ICK{}, ngf * U);
    } else {
      // A: the LEADING operators only, on the prefix; B: the trailing ones on the WHOLE list (the
      // prefix rows are fetched twice — a fifth more row loads on PubMed sign_k = 5 — but no phase
      // holds more than max(K-NB, NB) operators' accumulators: 128 instead of 166 VGPRs there,
      // four waves per SIMD instead of three)
      pass2(IC0{}, ICS{}, 0, gA);
      if (tail_in_A) tail_rows(IC0{}, ICS{}, ngf * U);
      write_pair_rows_part<K, CH, 0, K - NB, true, false>(job, jid, acc, coff, cok, job_z, X, ldx, F, rows,
                                                          blockIdx.y == 0);
      __builtin_amdgcn_sched_barrier(0);
      pass2(ICS{}, ICK{}, 0, ngf);
      tail_rows(ICS{}, ICK{}, ngf * U);
    }
    write_pair_rows_part<K, CH, K - NB, K, false, true>(job, jid, acc, coff, cok, job_z, X, ldx, F, rows,
                                                        blockIdx.y == 0);
  };
  int nb = 1;
#pragma unroll
  for (int i = K - 2; i >= 0; --i)
    if (lim[i] == lim[K - 1] && nb == K - 1 - i) nb = K - i;
  if constexpr (K >= 3) {
    if (nb >= 3) return run(std::integral_constant<int, 3>{});
  }
  if constexpr (K >= 2) {
    if (nb >= 2 || MINNB >= 2) return run(std::integral_constant<int, 2>{});
  }
  // (any NB gives the right sums — the coefficient lists hold zeros beyond an operator's reach —
  // the choice only decides how many multiply-adds are skipped)
  if constexpr (MINNB <= 1 || K < 2) run(std::integral_constant<int, 1>{});
}

// Measurement only (s3grl_plan_gather_traffic): the bytes the gather launch of a plan requests,
// summed exactly over its jobs with the same phase arithmetic the kernels use.  One wavefront per
// job; out[0..7] as documented in include/s3grl.h.
__global__ __launch_bounds__(256) void gather_traffic_kernel(
    const Job* __restrict__ jobs, int njobs, const int32_t* __restrict__ c_ids,
    const int32_t* __restrict__ job_lim, int K, int packed, const PackedHdr* __restrict__ hdr,
    int64_t N, int F, int pieces, unsigned long long* __restrict__ out) {
  const int lane = threadIdx.x & 63;
  const int jid = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (jid >= njobs) return;
  const Job job = jobs[jid];
  if (job.split == 1) {   // gathered piece by piece; the combine step writes its rows (and reads X for operator 0)
    if (lane == 0) {
      const unsigned long long nr = (job.node_b >= 0 ? 2 : 1) * (job.mirror_row >= 0 ? 2 : 1);
      atomicAdd(&out[4], 4ull * nr * (K + 1) * (unsigned long long)(F + 1));
      atomicAdd(&out[5], 16ull * ((F + 3) / 4) * (job.node_b >= 0 ? 2 : 1));
    }
    return;
  }
  const int cnt = job.support;
  const int32_t* __restrict__ ids = c_ids + job.ids_off;
  const int tile_cols = packed ? kTile : (F <= 256 ? 256 : 512);
  const int tiles = (F + tile_cols - 1) / tile_cols;
  const unsigned long long chunks_row = (unsigned long long)((F + 3) / 4);   // 16-byte loads inside a row
  // phase arithmetic of gather_packed_kernel: nb trailing operators reach the whole list; with nb >= 2
  // the nA rows of the prefix are fetched twice (leading operators, then trailing ones)
  int nb = 1, nA = 0;
  if (packed) {
    constexpr int U = 4;
    const int ngf = cnt / U;
    for (int i = K - 2; i >= 0; --i)
      if (job_lim[(int64_t)jid * K + i] == job_lim[(int64_t)jid * K + K - 1] && nb == K - 1 - i) nb = K - i;
    nb = min(nb, 3);
    const int limA = nb < K ? job_lim[(int64_t)jid * K + (K - nb - 1)] : 0;
    const int gA = min(ngf, (limA + U - 1) / U);
    const bool tail_in_A = limA > ngf * U;
    nA = tail_in_A ? cnt : gA * U;
  }
  const int twice = (packed && nb >= 2) ? nA : 0;
  unsigned long long feat = 0;
  if (packed) {
    for (int j = lane; j < cnt; j += 64) {
      const int id = ids[j];
      for (int t = 0; t < tiles; ++t) {
        const PackedHdr h = hdr[(int64_t)t * N + id];
        feat += (j < twice ? 32ull : 16ull) * (unsigned long long)(__popcll(h.m0) + __popcll(h.m1));
      }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) feat += __shfl_xor(feat, o);
  } else {
    feat = 16ull * chunks_row * (unsigned long long)cnt;
  }
  if (lane != 0) return;
  // coefficient entries read per tile: nb == 1: every operator on the prefix, the last one beyond it;
  // nb >= 2: the leading operators on the prefix, the trailing ones everywhere; dense kernel: all K
  unsigned long long coef_entries = (unsigned long long)K * cnt;
  if (packed)
    coef_entries = nb == 1 ? (unsigned long long)K * nA + (unsigned long long)(cnt - nA)
                           : (unsigned long long)(K - nb) * nA + (unsigned long long)nb * cnt;
  const int nrow = job.node_b >= 0 ? 2 : 1;
  const int ncopy = job.mirror_row >= 0 ? 2 : 1;
  atomicAdd(&out[0], 4ull * (cnt + twice) * tiles);
  atomicAdd(&out[1], packed ? 32ull * (cnt + twice) * tiles : 0ull);
  atomicAdd(&out[2], feat);
  atomicAdd(&out[3], 8ull * coef_entries * tiles);
  atomicAdd(&out[4], 4ull * nrow * ncopy * (K + 1) * (unsigned long long)(F + 1));
  // a piece's partial rows are read back once by the combine step
  atomicAdd(&out[5], 16ull * chunks_row * nrow + (pieces ? 4ull * nrow * (K + 1) * (unsigned long long)(F + 1) : 0ull));
  atomicAdd(&out[6], (unsigned long long)tiles * (sizeof(Job) + 4ull * K + 4ull) + 8ull * K);
  atomicAdd(&out[7], (unsigned long long)tiles);
}

// TIMING-ONLY prototype (S3GRL_GATHER_PROTO=<variant>, results are WRONG): what would the
// gather cost at eight wavefronts per SIMD?  The real kernel holds 117 VGPRs (four waves per SIMD): the
// accumulators of up to three operators and two or three chunk buffers of 32 registers each.  Eight waves
// leave 64 registers: one operator's accumulators (16) and ONE buffer of U = 4 rows (32), or two of U = 2 —
// i.e. phase B of the real kernel (the last operator over the rows beyond the prefix, four fifths of the
// headline's rows) with nothing pipelined by hand, the wavefronts covering each other's latency.  This
// kernel runs that phase over the WHOLE list of every job and writes the last operator's rows only.
// DB: two buffers of U rows (the loads of group g + 1 in flight under the multiply-adds of group g).
// HALVES = 2: a wavefront takes HALF of the tile (64 of its 128 chunks: one mask word, one load per row, half
// the accumulators) — twice the wavefronts, each walking the whole list.  PIN: the two halves run on disjoint
// sets of XCDs (workgroup id mod 8 < 4: half 0), so that an XCD's L2 only ever sees half of the packed operand.
// HDRWIN: the headers are read from a window of 256 rows (every scalar load a cache hit; the chunk loads then
// stay inside those rows' data as well) — what the kernel would cost without misses on its per-row chain.
// VHDR: the headers of the next 64 rows are fetched by ONE vector gather (lane u: row u's mask words and offset)
// and handed to the wavefront with v_readlane, instead of one scalar load per row through the scalar cache.
// ONELOAD (timing only, wrong sums): ONE wave-load per row — lane l fetches the row's l-th populated chunk (a
// row has ~42 of 128) — and the multiply-adds use it for both halves: what the kernel would cost if the texture
// path saw one 1-KiB wave-load per row instead of two (the data would then have to reach its owner lanes
// through LDS).
// EXECM (timing only): the chunk loads under an EXEC mask (inline assembly; the lanes of empty slots are switched
// off instead of being sent out of range) — does the texture addresser charge per ACTIVE lane?
template <int K, int U, bool DB, int WAVES, int HALVES = 1, bool PIN = false, bool HDRWIN = false, bool VHDR = false,
          bool ONELOAD = false, bool EXECM = false>
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(WAVES, WAVES))) void gather_last_proto_kernel(
    const Job* __restrict__ jobs, int njobs, const int32_t* __restrict__ c_ids, const float* __restrict__ c_coef,
    const int32_t* __restrict__ job_order, const PackedHdr* __restrict__ hdr, const float4_t* __restrict__ data,
    uint32_t data_bytes, int64_t N, int F, float* __restrict__ rows_out) {
  constexpr int CH = 2 / HALVES;
  const int lane = threadIdx.x & 63;
  int wid = __builtin_amdgcn_readfirstlane(blockIdx.x), half = 0;
  if constexpr (HALVES == 2) {
    if constexpr (PIN) {
      const int sub = wid & 7;
      half = sub >> 2;
      wid = (wid >> 3) * 4 + (sub & 3);
    } else {
      half = wid & 1;
      wid >>= 1;
    }
  }
  if (wid >= njobs) return;
  const int jid = __builtin_amdgcn_readfirstlane(job_order[wid]);
  const int col0 = blockIdx.y * kTile;
  const Job job = jobs[jid];
  if (job.split != 0) return;
  const int cnt = __builtin_amdgcn_readfirstlane(job.support);
  const uint32_t* __restrict__ uid = reinterpret_cast<const uint32_t*>(c_ids + job.ids_off);
  const float2* __restrict__ cf = reinterpret_cast<const float2*>(c_coef) + job.coef_off + (int64_t)(K - 1) * cnt;
  const PackedHdr* __restrict__ th = hdr + (int64_t)blockIdx.y * N;
  const __amdgpu_buffer_rsrc_t rsrc =
      __builtin_amdgcn_make_buffer_rsrc(const_cast<float4_t*>(data), 0, (int)data_bytes, 0x00020000);
  const uint32_t oobv = kOobOffset;
  float4_t acc[2][CH];
#pragma unroll
  for (int r = 0; r < 2; ++r)
#pragma unroll
    for (int c = 0; c < CH; ++c) acc[r][c] = (float4_t)(0.f);
  // VHDR: headers of rows [hb, hb + 64) in registers, one row per lane (m0, m1 as two dwords each, offset)
  uint32_t hv[5] = {0, 0, 0, 0, 0};
  int hb = -64;
  auto fetch_headers = [&](int base_row) __attribute__((always_inline)) {
    const int r = min(base_row + lane, cnt - 1);
    const uint32_t id = uid[r];
    const uint32_t* __restrict__ hp = reinterpret_cast<const uint32_t*>(th + id);
    const uint4_t w = *reinterpret_cast<const uint4_t*>(hp);
    hv[0] = w.x; hv[1] = w.y; hv[2] = w.z; hv[3] = w.w;
    hv[4] = hp[4];
    hb = base_row;
  };
  auto issue = [&](int g, float4_t(&v)[U][CH]) __attribute__((always_inline)) {
    uint32_t id[U];
    PackedHdr h[U];
    if constexpr (VHDR) {
      if (g * U >= hb + 64) fetch_headers(g * U);      // (64 % U == 0: a group never straddles two batches)
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const int ln = g * U + u - hb;
        h[u].m0 = (uint64_t)(uint32_t)__builtin_amdgcn_readlane((int)hv[0], ln) |
                  ((uint64_t)(uint32_t)__builtin_amdgcn_readlane((int)hv[1], ln) << 32);
        h[u].m1 = (uint64_t)(uint32_t)__builtin_amdgcn_readlane((int)hv[2], ln) |
                  ((uint64_t)(uint32_t)__builtin_amdgcn_readlane((int)hv[3], ln) << 32);
        h[u].off = (uint64_t)(uint32_t)__builtin_amdgcn_readlane((int)hv[4], ln);
      }
    } else {
#pragma unroll
      for (int u = 0; u < U; ++u) id[u] = uid[min(g * U + u, cnt - 1)];
#pragma unroll
      for (int u = 0; u < U; ++u) h[u] = th[HDRWIN ? (id[u] & 255u) : id[u]];
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const bool ok = g * U + u < cnt;
      const uint32_t base = (uint32_t)h[u].off;
      const uint32_t a0 = (base + (uint32_t)below(h[u].m0)) << 4;
      const uint32_t a1 = (base + (uint32_t)__popcll(h[u].m0) + (uint32_t)below(h[u].m1)) << 4;
      if constexpr (HALVES == 2) {
        const uint64_t m = half ? h[u].m1 : h[u].m0;
        v[u][0] = __builtin_bit_cast(float4_t, __builtin_amdgcn_raw_buffer_load_b128(
                                                   rsrc, (int)select_or_oob(ok ? m : 0ull, half ? a1 : a0, oobv), 0, 0));
      } else if constexpr (ONELOAD) {
        const uint32_t nch = (uint32_t)(__popcll(h[u].m0) + __popcll(h[u].m1));
        const uint32_t ac = (uint32_t)lane < nch && ok ? (base + (uint32_t)lane) << 4 : oobv;
        v[u][0] = __builtin_bit_cast(float4_t, __builtin_amdgcn_raw_buffer_load_b128(rsrc, (int)ac, 0, 0));
        v[u][CH - 1] = v[u][0];
        (void)a0;
        (void)a1;
      } else if constexpr (EXECM) {
        const uint64_t mm0 = ok ? h[u].m0 : 0ull, mm1 = ok ? h[u].m1 : 0ull;
        v[u][0] = (float4_t)(0.f);
        v[u][CH - 1] = (float4_t)(0.f);
        asm volatile("s_mov_b64 exec, %2\n\tbuffer_load_dwordx4 %0, %4, %6, 0 offen\n\t"
                     "s_mov_b64 exec, %3\n\tbuffer_load_dwordx4 %1, %5, %6, 0 offen\n\ts_mov_b64 exec, -1"
                     : "+v"(v[u][0]), "+v"(v[u][CH - 1])
                     : "s"(mm0), "s"(mm1), "v"(a0), "v"(a1), "s"(rsrc)
                     : "memory");
      } else {
        v[u][0] = __builtin_bit_cast(float4_t, __builtin_amdgcn_raw_buffer_load_b128(
                                                   rsrc, (int)select_or_oob(ok ? h[u].m0 : 0ull, a0, oobv), 0, 0));
        v[u][CH - 1] = __builtin_bit_cast(float4_t, __builtin_amdgcn_raw_buffer_load_b128(
                                                        rsrc, (int)select_or_oob(ok ? h[u].m1 : 0ull, a1, oobv), 0, 0));
      }
    }
  };
  auto fma = [&](int g, float4_t(&v)[U][CH]) __attribute__((always_inline)) {
    if constexpr (EXECM) {   // the compiler does not see those loads: wait for all of them, the values tied to the wait
#pragma unroll
      for (int u = 0; u < U; ++u) asm volatile("s_waitcnt vmcnt(0)" : "+v"(v[u][0]), "+v"(v[u][CH - 1]));
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const float2 q = cf[min(g * U + u, cnt - 1)];
#pragma unroll
      for (int c = 0; c < CH; ++c) {
        acc[0][c] += q.x * v[u][c];
        acc[1][c] += q.y * v[u][c];
      }
    }
  };
  const int ng = (cnt + U - 1) / U;
  if constexpr (DB) {
    float4_t va[U][CH], vb[U][CH];
    issue(0, va);
    for (int g = 0; g < ng; g += 2) {
      issue(min(g + 1, ng - 1), vb);
      fma(g, va);
      issue(min(g + 2, ng - 1), va);
      if (g + 1 < ng) fma(g + 1, vb);
    }
  } else {
    float4_t va[U][CH];
    for (int g = 0; g < ng; ++g) {
      issue(g, va);
      fma(g, va);
    }
  }
  const int Fp = F + 1;
  float* __restrict__ out = rows_out + job.out_row * (int64_t)(K + 1) * Fp + (int64_t)K * Fp + 1;
#pragma unroll
  for (int r = 0; r < 2; ++r) {
    if (r == 1 && job.node_b < 0) break;
#pragma unroll
    for (int c = 0; c < CH; ++c) {
      const int co = col0 + (lane + 64 * (HALVES == 2 ? half : c)) * 4;
#pragma unroll
      for (int e = 0; e < 4; ++e)
        if (co + e < F) out[(int64_t)r * (K + 1) * Fp + co + e] = acc[r][c][e];
    }
  }
}

template <int K>
s3grl_status launch_packed_k(s3grl_context* ctx, const s3grl_plan* p, const GatherView& v,
                             const s3grl_features* f, float* rows) {
  hipStream_t stream = ctx->stream;
  const unsigned gx = (unsigned)((v.njobs + kWavesPerBlock - 1) / kWavesPerBlock);
  // timing-only experiment (see gather_last_proto_kernel); instantiated for the headline's sign_k = 3 only
  if (const char* proto = K == 3 ? getenv("S3GRL_GATHER_PROTO") : nullptr) {
   if constexpr (K == 3) {
    const uint32_t db = (uint32_t)((f->pk_chunks + 1) * 16);
#define S3GRL_PROTO(UU, DBB, WW)                                                                               \
  hipLaunchKernelGGL((gather_last_proto_kernel<K, UU, DBB, WW>), dim3((unsigned)v.njobs, (unsigned)f->tiles),   \
                     dim3(64), 0, stream, v.jobs, (int)v.njobs, p->c_ids, p->c_coef, v.job_order,              \
                     static_cast<const PackedHdr*>(f->pk_hdr), static_cast<const float4_t*>(f->pk_data), db,   \
                     f->N, (int)f->F, rows)
    switch (atoi(proto)) {
      case 1: S3GRL_PROTO(4, false, 8); break;   // one buffer of four rows, eight waves per SIMD
      case 2: S3GRL_PROTO(2, true, 8); break;    // two buffers of two rows, eight waves
      case 3: S3GRL_PROTO(4, true, 5); break;    // two buffers of four rows, five waves
      case 4: S3GRL_PROTO(4, false, 4); break;   // control: one buffer at the real kernel's four waves
      case 5: S3GRL_PROTO(2, false, 8); break;
#define S3GRL_PROTO_H(UU, DBB, WW, PINN)                                                                        \
  hipLaunchKernelGGL((gather_last_proto_kernel<K, UU, DBB, WW, 2, PINN>),                                       \
                     dim3((unsigned)((v.njobs + 3) / 4 * 8), (unsigned)f->tiles), dim3(64), 0, stream, v.jobs,  \
                     (int)v.njobs, p->c_ids, p->c_coef, v.job_order, static_cast<const PackedHdr*>(f->pk_hdr),  \
                     static_cast<const float4_t*>(f->pk_data), db, f->N, (int)f->F, rows)
      case 6: S3GRL_PROTO_H(4, false, 8, true); break;    // half tiles pinned to XCD halves, one buffer, eight waves
      case 7: S3GRL_PROTO_H(4, false, 8, false); break;   // half tiles, not pinned (control)
      case 8: S3GRL_PROTO_H(4, true, 8, true); break;     // half tiles pinned, two buffers of four rows
      case 9: S3GRL_PROTO_H(4, true, 8, false); break;
#undef S3GRL_PROTO_H
#define S3GRL_PROTO_X(UU, DBB, WW, WIN, VH)                                                                      \
  hipLaunchKernelGGL((gather_last_proto_kernel<K, UU, DBB, WW, 1, false, WIN, VH>),                              \
                     dim3((unsigned)v.njobs, (unsigned)f->tiles), dim3(64), 0, stream, v.jobs, (int)v.njobs,     \
                     p->c_ids, p->c_coef, v.job_order, static_cast<const PackedHdr*>(f->pk_hdr),                 \
                     static_cast<const float4_t*>(f->pk_data), db, f->N, (int)f->F, rows)
      case 10: S3GRL_PROTO_X(4, false, 8, true, false); break;   // headers (and chunks) from a 256-row window
      case 11: S3GRL_PROTO_X(4, false, 8, false, true); break;   // headers by vector gather + readlane, one buffer
      case 12: S3GRL_PROTO_X(4, true, 5, false, true); break;    // ... two buffers of four rows at five waves
      case 13: S3GRL_PROTO_X(2, true, 8, false, true); break;    // ... two buffers of two rows at eight waves
      case 14: S3GRL_PROTO_X(4, true, 5, true, false); break;    // window control with two buffers
#undef S3GRL_PROTO_X
#define S3GRL_PROTO_O(UU, DBB, WW, VH)                                                                           \
  hipLaunchKernelGGL((gather_last_proto_kernel<K, UU, DBB, WW, 1, false, false, VH, true>),                      \
                     dim3((unsigned)v.njobs, (unsigned)f->tiles), dim3(64), 0, stream, v.jobs, (int)v.njobs,     \
                     p->c_ids, p->c_coef, v.job_order, static_cast<const PackedHdr*>(f->pk_hdr),                 \
                     static_cast<const float4_t*>(f->pk_data), db, f->N, (int)f->F, rows)
      case 15: S3GRL_PROTO_O(4, false, 8, true); break;    // one compact load per row, vector headers, eight waves
      case 16: S3GRL_PROTO_O(4, true, 8, true); break;     // ... two buffers of four rows (16 VGPRs each)
      case 17: S3GRL_PROTO_O(4, false, 8, false); break;   // ... scalar headers
#undef S3GRL_PROTO_O
      case 21:   // chunk loads under an EXEC mask, scalar headers, one buffer of four rows, eight waves
        hipLaunchKernelGGL((gather_last_proto_kernel<K, 4, false, 8, 1, false, false, false, false, true>),
                           dim3((unsigned)v.njobs, (unsigned)f->tiles), dim3(64), 0, stream, v.jobs, (int)v.njobs,
                           p->c_ids, p->c_coef, v.job_order, static_cast<const PackedHdr*>(f->pk_hdr),
                           static_cast<const float4_t*>(f->pk_data), db, f->N, (int)f->F, rows);
        break;
      case 22:   // ... vector headers
        hipLaunchKernelGGL((gather_last_proto_kernel<K, 4, false, 8, 1, false, false, true, false, true>),
                           dim3((unsigned)v.njobs, (unsigned)f->tiles), dim3(64), 0, stream, v.jobs, (int)v.njobs,
                           p->c_ids, p->c_coef, v.job_order, static_cast<const PackedHdr*>(f->pk_hdr),
                           static_cast<const float4_t*>(f->pk_data), db, f->N, (int)f->F, rows);
        break;
      default: S3GRL_PROTO(2, false, 8); break;
    }
#undef S3GRL_PROTO
    S3GRL_HIP_TRY(hipGetLastError());
    return S3GRL_OK;
   }
  }
  static const bool masked = !getenv("S3GRL_GATHER_UNMASKED");   // comparison hook: the unconditional loads
  const uint32_t data_bytes = (uint32_t)((f->pk_chunks + 1) * 16);
  // experiment hook: dynamic LDS per workgroup caps the resident waves (timing only)
  static const size_t lds_cap = getenv("S3GRL_GATHER_LDS") ? (size_t)atoi(getenv("S3GRL_GATHER_LDS")) : 0;
  // every job's last two operators reach its whole list when sign_k - 1 >= the BFS depth (one hop for
  // random-walk subgraphs)
  const int depth = p->walk_plan ? 1 : p->cfg.num_hops;
  if (K >= 2 && K - 1 >= depth && masked) {
    hipLaunchKernelGGL((gather_packed_kernel<K, true, (K >= 2 ? 2 : 1)>), dim3(gx, (unsigned)f->tiles),
                       dim3(kWavesPerBlock * 64), lds_cap, stream, v.jobs, (int)v.njobs, p->c_ids, p->c_coef,
                       v.job_z, v.job_lim, v.job_order, static_cast<const PackedHdr*>(f->pk_hdr),
                       static_cast<const float4_t*>(f->pk_data), data_bytes, f->N, f->dense, f->ld, (int)f->F, rows, v.prows);
    S3GRL_HIP_TRY(hipGetLastError());
    return S3GRL_OK;
  }
  if (masked)
    hipLaunchKernelGGL((gather_packed_kernel<K, true, 1>), dim3(gx, (unsigned)f->tiles), dim3(kWavesPerBlock * 64),
                       lds_cap, stream, v.jobs, (int)v.njobs, p->c_ids, p->c_coef, v.job_z, v.job_lim,
                       v.job_order, static_cast<const PackedHdr*>(f->pk_hdr),
                       static_cast<const float4_t*>(f->pk_data), data_bytes, f->N, f->dense, f->ld, (int)f->F, rows, v.prows);
  else
    hipLaunchKernelGGL((gather_packed_kernel<K, false, 1>), dim3(gx, (unsigned)f->tiles), dim3(kWavesPerBlock * 64),
                       0, stream, v.jobs, (int)v.njobs, p->c_ids, p->c_coef, v.job_z, v.job_lim,
                       v.job_order, static_cast<const PackedHdr*>(f->pk_hdr),
                       static_cast<const float4_t*>(f->pk_data), data_bytes, f->N, f->dense, f->ld, (int)f->F, rows, v.prows);
  S3GRL_HIP_TRY(hipGetLastError());
  return S3GRL_OK;
}

}  // namespace

// Builds the packed copy of f->dense when at most `max_density` of its chunks are non-zero
// (otherwise leaves f->packed false: the dense kernel moves no more bytes and issues fewer
// instructions).  One host round trip for the chunk total.
s3grl_status build_packed_rows(s3grl_context* ctx, s3grl_features* f, double max_density) {
  const int64_t N = f->N;
  const int tiles = (int)((f->F + kTile - 1) / kTile);
  const int64_t items = N * tiles;
  Transient tmp{ctx, {}};
  void* q = nullptr;
  S3GRL_TRY(ctx->arena.alloc((size_t)items * 4, &q));
  tmp.ptrs.push_back(q);
  int32_t* cnt = static_cast<int32_t*>(q);
  S3GRL_TRY(ctx->arena.alloc((size_t)(items + 1) * 8, &q));
  tmp.ptrs.push_back(q);
  int64_t* ptr = static_cast<int64_t*>(q);
  S3GRL_TRY(ctx->arena.alloc((size_t)scan_workspace_elems(items) * 8, &q));
  tmp.ptrs.push_back(q);
  int64_t* ws = static_cast<int64_t*>(q);
  const unsigned grid = (unsigned)((items + 3) / 4);
  hipLaunchKernelGGL(pk_count_kernel, dim3(grid), dim3(256), 0, ctx->stream, f->dense, f->ld, (int)f->F, N, tiles, cnt);
  S3GRL_HIP_TRY(hipGetLastError());
  S3GRL_TRY(launch_scan_i32_to_i64(ctx, cnt, items, ptr, ws));
  S3GRL_HIP_TRY(hipMemcpyAsync(ctx->h_scalars, ptr + items, 8, hipMemcpyDeviceToHost, ctx->stream));
  S3GRL_HIP_TRY(hipStreamSynchronize(ctx->stream));
  const int64_t chunks = ctx->h_scalars[0];
  const double slots = (double)N * (double)((f->F + 3) / 4);
  f->pk_chunks = chunks;
  if ((double)chunks > max_density * slots) return S3GRL_OK;
  if ((chunks + 1) * 16 >= ((int64_t)1 << 31)) return S3GRL_OK;   // 32-bit byte offsets, and room for the out-of-range one
  if (N >= ((int64_t)1 << 27)) return S3GRL_OK;                   // 32-bit byte offsets into a tile's headers
  void* hdr = nullptr;
  void* data = nullptr;
  S3GRL_TRY(ctx->arena.alloc((size_t)items * sizeof(PackedHdr), &hdr));
  f->owned.push_back(hdr);
  S3GRL_TRY(ctx->arena.alloc((size_t)(chunks + 1) * 16, &data));
  f->owned.push_back(data);
  hipLaunchKernelGGL(pk_fill_kernel, dim3(grid), dim3(256), 0, ctx->stream, f->dense, f->ld, (int)f->F, N, tiles, ptr,
                     static_cast<PackedHdr*>(hdr), static_cast<float4_t*>(data));
  S3GRL_HIP_TRY(hipGetLastError());
  f->pk_hdr = hdr;
  f->pk_data = data;
  f->packed = true;
  return S3GRL_OK;
}

s3grl_status launch_gather_traffic(s3grl_context* ctx, const s3grl_plan* p, const s3grl_features* f,
                                   unsigned long long* d_out) {
  if (p->njobs == 0) return S3GRL_OK;
  if (f->sparse) {
    set_last_error("gather traffic accounting covers the dense and the packed operand");
    return S3GRL_ERR_NOT_IMPLEMENTED;
  }
  hipLaunchKernelGGL(gather_traffic_kernel, dim3((unsigned)((p->njobs + 3) / 4)), dim3(256), 0, ctx->stream,
                     p->jobs, (int)p->njobs, p->c_ids, p->job_lim, p->cfg.sign_k, f->packed ? 1 : 0,
                     static_cast<const PackedHdr*>(f->pk_hdr), f->N, (int)f->F, 0, d_out);
  if (p->npieces)
    hipLaunchKernelGGL(gather_traffic_kernel, dim3((unsigned)((p->npieces + 3) / 4)), dim3(256), 0, ctx->stream,
                       p->gjobs + p->njobs, (int)p->npieces, p->c_ids, p->g_lim + p->njobs * p->cfg.sign_k,
                       p->cfg.sign_k, f->packed ? 1 : 0, static_cast<const PackedHdr*>(f->pk_hdr), f->N, (int)f->F, 1,
                       d_out);
  S3GRL_HIP_TRY(hipGetLastError());
  return S3GRL_OK;
}

s3grl_status launch_gather_packed(s3grl_context* ctx, const s3grl_plan* p, const GatherView& v,
                                  const s3grl_features* f, float* rows) {
  if (v.njobs == 0) return S3GRL_OK;
  switch (p->cfg.sign_k) {
    case 1: return launch_packed_k<1>(ctx, p, v, f, rows);
    case 2: return launch_packed_k<2>(ctx, p, v, f, rows);
    case 3: return launch_packed_k<3>(ctx, p, v, f, rows);
    case 4: return launch_packed_k<4>(ctx, p, v, f, rows);
    case 5: return launch_packed_k<5>(ctx, p, v, f, rows);
    case 6: return launch_packed_k<6>(ctx, p, v, f, rows);
    case 7: return launch_packed_k<7>(ctx, p, v, f, rows);
    case 8: return launch_packed_k<8>(ctx, p, v, f, rows);
    default:
      set_last_error("sign_k must be in 1..8");
      return S3GRL_ERR_INVALID_ARGUMENT;
  }
}

}  // namespace s3grl

S3GRL_DEFINE_TOUCH(packed)
