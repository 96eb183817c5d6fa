// Internal declarations shared by the translation units of libs3grl_hip.so (gfx950 only).
#pragma once

#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>
#include <map>
#include <string>
#include <vector>

#include "../../include/s3grl.h"

namespace s3grl {

void set_last_error(const std::string& msg);

#define S3GRL_HIP_TRY(expr)                                                                   \
  do {                                                                                        \
    hipError_t _e = (expr);                                                                   \
    if (_e != hipSuccess) {                                                                   \
      ::s3grl::set_last_error(std::string(#expr) + ": " + hipGetErrorString(_e) + " (" +      \
                              __FILE__ + ":" + std::to_string(__LINE__) + ")");               \
      return _e == hipErrorOutOfMemory ? S3GRL_ERR_OUT_OF_MEMORY : S3GRL_ERR_HIP;             \
    }                                                                                         \
  } while (0)

#define S3GRL_TRY(expr)                    \
  do {                                     \
    s3grl_status _s = (expr);              \
    if (_s != S3GRL_OK) return _s;         \
  } while (0)

constexpr int kBlock = 256;            // threads per workgroup in the structure kernels
constexpr int kMaxNodesLds = 327680;   // 4 bitmaps of N bits must fit 160 KiB of LDS
constexpr int kMaxSignK = 8;
constexpr int kStatShards = 64;        // see stat_slot() in s3grl_device.hpp
constexpr int kStatStride = 16;        // int64 per shard: one 128-byte line
constexpr int kStatRows = 9;           // Σ edges, Σ support, Σ vol, Σ n (algorithmic), Σ oriented-row entries,
                                       // links of link_hub_kernel, bytes they read, Σ of their endpoint degrees, Σ of their n
constexpr int kMaxLevels = 32;         // BFS levels tracked per link (num_hops <= 30)
// link_kernel keeps a whole subgraph on-chip; links are binned by LDS need into classes
constexpr int kNumClasses = 6;
// variable LDS bytes per link (list + state on the propagation prefix), upper bound per class
#define S3GRL_CLASS_BOUNDS {6144, 12288, 24576, 49152, 98304, 163840}

// per-hop sampling settings of a plan (reference utils.py:66-70); ratio outside (0,1) and
// max_nodes == 0 mean "keep every node"
struct HopSampling {
  double ratio;
  int32_t max_nodes;
  uint32_t seed;
};
static inline bool hop_sampling_on(const HopSampling& s) {
  return (s.ratio > 0.0 && s.ratio < 1.0) || s.max_nodes > 0;
}

// ScaLed subgraphs (reference utils.py:86-150): where the "walk nodes" of a link come from.
// raw: the engine's own walks, [N, len] per NODE (s3grl_cfg.rw_m / rw_M);  ptr / nodes: node sets
// handed in by the caller as a CSR (s3grl_plan_create_sets), one set per node or one per link.
struct WalkSets {
  const int32_t* raw = nullptr;
  int len = 0;
  const int64_t* ptr = nullptr;
  const int32_t* nodes = nullptr;
  int per_link = 0;
  int num_nodes = 0;
};
__host__ __device__ static inline bool walks_on(const WalkSets& w) { return w.raw != nullptr || w.ptr != nullptr; }

// One gather job = one pair of output rows of one link (rows 2p, 2p+1 of that link).
struct Job {
  int64_t coef_off;   // first float2 of this job's coefficients, laid out [K][support]
  int64_t ids_off;    // first entry of the link's node-id list (shared by its row pairs)
  int64_t out_row;    // index of the first output row
  int32_t link;       // link index
  int32_t support;    // list entries this job reads (a hop-major prefix of the link's nodes)
  int32_t node_a;     // global id of row a
  int32_t node_b;     // global id of row b, or -1 when the pair has a single row
  int32_t z_a;        // label column of operator 0 (1 for src/dst)
  int32_t z_b;
  int64_t mirror_row; // first output row of the same pair of the REVERSED link (dst,src) when
                      // that link is in the list too and was folded into this one, else -1
  int32_t mirror_swap;// rows a,b go to mirror_row+1, mirror_row (the src/dst pair), else same order
  int32_t split;      // 1: the list is cut into pieces that are gathered on their own (s3grl_plan::gjobs)
                      // and summed by combine_kernel — the gather skips this entry; 2: this entry IS such a
                      // piece: its two rows go to the partial-row scratch instead of the output
};

// Jobs whose list is longer than kSplitThreshold entries are cut into pieces of 2^kSplitSegShift
// entries: the longest unit of a gather launch is bounded (a 5 900-node PubMed subgraph is a 1.9 ms
// wavefront — the tail of a launch once the list is sharded over 8 GPUs), and no fp32 running sum is
// longer than the threshold (the pieces are added in f64): the accumulation error stops growing
// with the subgraph.  A piece costs a wavefront's fixed work and 2 partial rows.  Measured on the
// headline (164 000 links, gather 6.96 ms unsplit): threshold 4096 7.00 ms, 3072 7.16, 2048 7.48; the
// pieces in a launch of their own in front of the main one: +0.3 ms even for a hundred split jobs
// (they are gather units of the same launch).  The threshold is a constant, NOT a function of the plan: whether
// a job is split decides its summation order, and a link must come out bit for bit the same in a
// sharded and in an unsharded run.  S3GRL_SPLIT_T / S3GRL_SPLIT_SEG_SHIFT override (0 = never split).
constexpr int kSplitThreshold = 4096;
constexpr int kSplitSegShift = 10;
// plans on graphs / lists at least this big work on their links in hub order (launch_link_order)
constexpr int64_t kHubOrderMinNodes = 65536;
constexpr int64_t kHubOrderMinLinks = 65536;

// what a gather launch works on: the jobs of a plan — followed, when some are split, by their pieces
struct GatherView {
  const Job* jobs;
  int64_t njobs;
  const float* job_z;
  const int32_t* job_lim;
  const int32_t* job_order;
  float* prows;   // partial rows of the pieces (Job::split == 2), or null
};

// Grow-only caching device allocator: plans are created and destroyed every benchmark step,
// hipMalloc/hipFree of multi-GB blocks must not sit inside the timed region.
class Arena {
 public:
  ~Arena();
  s3grl_status alloc(size_t bytes, void** out);
  void release(void* p);
  size_t trim();   // hipFree every cached (not live) block; returns the bytes given back
  size_t bytes_held() const { return held_; }
  // A cold arena pays one hipMalloc per block, and a hipMalloc of a megabyte or more costs 0.3-3 ms (a
  // process's first s3grl_graph_create on PubMed: a dozen of them, 4 of its 4.4 ms; its first plan: thirty).
  // reserve(bytes): when the cached blocks could not serve about that much, ONE block of that size is
  // allocated and the allocations that follow are carved out of it.  Carved blocks are cached and reused
  // like any other; their slab goes back to HIP when none of them is live or cached any more (trim).
  s3grl_status reserve(size_t bytes);
  size_t cached_bytes() const { return cached_; }

 private:
  struct Slab {
    char* base;
    size_t size, used;
    int blocks;   // carved blocks still live or cached
  };
  std::multimap<size_t, void*> free_;
  std::map<void*, size_t> live_;
  std::map<void*, int> carved_;   // block -> slab
  std::vector<Slab> slabs_;
  int cur_slab_ = -1;
  size_t held_ = 0, cached_ = 0;
};

}  // namespace s3grl

struct s3grl_context {
  int device = 0;
  hipStream_t stream = nullptr;
  s3grl::Arena arena;
  bool profiling = false;
  bool gather_pending = false;  // ev[3], ev[4] recorded but not yet read
  double timings[16] = {0};
  hipEvent_t ev[8] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
  // side streams for launches that do not depend on each other (the link kernels of the LDS
  // classes): forked from and joined back into `stream` with events, created on first use
  static constexpr int kSide = 3;
  hipStream_t side[kSide] = {nullptr, nullptr, nullptr};
  hipEvent_t side_ev[kSide + 1] = {nullptr, nullptr, nullptr, nullptr};
  int64_t* d_scalars = nullptr;  // small device scratch for totals (64 x int64)
  int64_t* h_scalars = nullptr;  // pinned host mirror
  int64_t* d_stats = nullptr;    // [kStatRows][kStatShards * kStatStride] sharded totals of a plan
  int64_t* h_stats = nullptr;    // pinned host mirror
  // the plan whose link-kernel totals (Σ edges / support / vol) are still on their way into h_stats:
  // plan creation does not wait for its link kernels — the gather is queued right behind them — and
  // the totals are read when somebody asks (s3grl_plan_get_stats), or before the buffers are reused
  s3grl_plan* stats_owner = nullptr;
};

// Hub neighbourhoods (s3grl_hub.hip), built once per graph for one-hop plans on big graphs: for every
// node h of at least kHubMinDegree neighbours (no self-loop, at most 65 535 neighbours) the induced
// adjacency of N(h) as a CSR of positions in h's sorted row.  A link with such an endpoint shares all of
// it with every other link of that hub: link_hub_kernel only finds the edges of the OTHER endpoint's
// neighbours and pulls over the cached rows.
struct HubCache {
  int32_t nh = 0;
  int32_t* slot = nullptr;       // [N] hub number of a node, -1 for the others
  int32_t* voln = nullptr;       // [N] Σ degree over a node's stored neighbours
  int64_t* row_base = nullptr;   // [nh + 1] first entry of hub k in hoff (deg(h) + 1 entries per hub)
  int64_t* col_base = nullptr;   // [nh + 1] first entry of hub k in hcols
  int32_t* hoff = nullptr;       // row offsets, relative to col_base[k]
  uint16_t* hcols = nullptr;     // positions in the hub's row, ascending inside a row
};
constexpr int kHubMinDegree = 128;
constexpr int kHubClasses = 4;           // LDS classes of link_hub_kernel: 128 / 256 / 512 / 1024 threads; one more
                                         // (index kHubClasses) keeps its list of found edges in HBM slices
// Σ degree over the other endpoint's neighbourhood: beyond, the old path (it sizes the HBM slices of the class whose
// found edges do not fit LDS).  65 536 until round 4: the 561 hub - hub links of config 5 beyond it cost 2.2 ms on
// link_full_kernel's big class and the general kernel, one after the other; here 0.4 ms of the phase less.
constexpr int64_t kHubVolMax = 262144;

// BFS balls of every node of the degree-ordered graph as N-bit bitmaps (s3grl_balls.hip): level d - 1 of
// `bits` holds ball_d(x) = the nodes within d hops of x, x included, for all x — built on first use, level by
// level, for the plans whose sizing pass is bitmap arithmetic on them
struct BallCache {
  int hops = 0;                 // levels built
  int64_t level_stride = 0;     // words per level = N * ceil(N / 32)
  uint32_t* bits = nullptr;     // [hops][N][ceil(N / 32)]
};

struct s3grl_graph {
  s3grl_context* ctx = nullptr;
  HubCache hub;                // of the degree-ordered graph (r_indptr / r_indices)
  BallCache balls;             // of the degree-ordered graph
  int64_t num_nodes = 0;
  int64_t nnz = 0;
  int32_t max_degree = 0;      // decides whether the hub-row path of the row walker is armed
  int32_t* indptr = nullptr;   // [N+1] device, int32 (nnz < 2^31)
  int32_t* indices = nullptr;  // [nnz] device
  // degree-oriented rows (s3grl_onehop.inl), built for big graphs only: every undirected edge once,
  // in the row of its endpoint of lower (degree, id); self-loops in their own row
  int32_t* fwd_indptr = nullptr;   // [N+1]
  int32_t* fwd_indices = nullptr;  // [nnz / 2 (+ self-loops)]
  uint16_t* fwd_deg = nullptr;     // [N] length of the oriented rows, for the sizing pass
  // the same graph with ids in descending degree order (s3grl_relabel.hip): what the multi-hop
  // kernels walk; null when the graph is beyond their reach
  int32_t* r_indptr = nullptr;     // [N+1]
  int32_t* r_indices = nullptr;    // [nnz], rows ascending in the new ids
  int32_t* new_of_old = nullptr;   // [N]
  int32_t* old_of_new = nullptr;   // [N]
  int32_t deg_le2_from = 0;        // new ids >= this have at most two stored neighbours
  int32_t* r_fwd_indptr = nullptr;   // the degree-oriented rows of the relabelled graph (big graphs)
  int32_t* r_fwd_indices = nullptr;
  uint16_t* r_fwd_deg = nullptr;
  // directed graphs (s3grl_graph_create_directed): indptr / indices above hold the UNION of out- and
  // in-neighbours (what the reference's directed BFS follows, utils.py:60-63); the arcs themselves:
  bool directed = false;
  int64_t arcs = 0;
  int32_t* out_indptr = nullptr;   // [N+1] CSR of A: successors (degrees, common neighbours)
  int32_t* out_indices = nullptr;  // [arcs]
  int32_t* in_indptr = nullptr;    // [N+1] CSC of A = CSR of A^T: predecessors (the pulls r_i = r_{i-1} A_hat)
  int32_t* in_indices = nullptr;   // [arcs]
};

// the arcs of a directed graph as the kernels take them (all null for an undirected graph)
struct DirGraph {
  const int32_t* out_indptr = nullptr;
  const int32_t* out_indices = nullptr;
  const int32_t* in_indptr = nullptr;
  const int32_t* in_indices = nullptr;
};

struct s3grl_plan {
  s3grl_context* ctx = nullptr;
  const s3grl_graph* graph = nullptr;
  s3grl_cfg cfg{};
  int64_t L = 0;
  s3grl_plan_stats stats{};
  // per link
  int64_t* links = nullptr;      // [L,2]
  int32_t* n_nodes = nullptr;    // [L]
  int64_t* node_off = nullptr;   // [L+1]
  int64_t* row_ptr = nullptr;    // [L+1]
  int64_t* job_off = nullptr;    // [L+1]
  int32_t* lvl = nullptr;        // [L, kMaxLevels] cumulative node count per BFS level
  int32_t* e_cap = nullptr;      // [L] bound of the induced entries (one-hop plans on big graphs), else null
  int64_t* x_cap = nullptr;      // [L] ... and, on graphs with cached hub neighbourhoods: >= 0 for the links
                                 // link_hub_kernel can take (staged cache bytes << 32 | bound of the found edges)
  bool relabelled = false;       // the kernels walked the graph's degree order (s3grl_relabel.hip)
  bool stats_pending = false;    // total_sub_edges / total_support / total_volume not read back yet
  bool hub_order = false;        // links worked on in hub order (launch_link_order); job_order follows it
  bool walk_plan = false;        // ScaLed: subgraph = walk nodes of src and dst (one "hop", whatever num_hops)
  int32_t* c_ids = nullptr;      // [Σn] subgraph nodes, hop-major (ascending id inside a hop unless relabelled)
  // per job (row pair)
  s3grl::Job* jobs = nullptr;    // [njobs]
  int64_t njobs = 0;
  float* job_z = nullptr;        // [njobs, K, 2] label column of operators 1..K
  int32_t* job_order = nullptr;  // [njobs] the order the gather starts its jobs in (largest first)
  int32_t* job_lim = nullptr;    // [njobs, K] operator i+1 has no non-zero coefficient at list
                                 // positions >= job_lim[j, i] (non-decreasing in i)
  float* c_coef = nullptr;       // [Σ_jobs n, K, 2]
  int64_t* row_nodes = nullptr;  // [ΣR]
  // split jobs (Job::split): their pieces as gather units of their own
  int split_t = 0, seg_shift = 0;  // threshold and piece size the link kernels laid the coefficients out for
  int64_t npieces = 0;
  int64_t* piece_off = nullptr;  // [njobs + 1] first piece of every job (no pieces for an unsplit job)
  // the gather units of a plan with split jobs: its njobs jobs followed by the npieces pieces
  s3grl::Job* gjobs = nullptr;   // [njobs + npieces]; piece q: out_row = 2q into the partial-row scratch
  float* g_z = nullptr;          // [njobs + npieces, K, 2] (zeros for pieces: combine_kernel writes the label column)
  int32_t* g_lim = nullptr;      // [njobs + npieces, K]
  int32_t* g_order = nullptr;    // [njobs + npieces] the pieces first, then job_order
  int32_t* piece_job = nullptr;  // [npieces] the job a piece belongs to
  std::vector<void*> owned;      // everything above, for release
};

struct s3grl_features {
  s3grl_context* ctx = nullptr;
  int64_t N = 0, F = 0;
  const float* dense = nullptr;   // 16-byte aligned rows (the caller's X or an owned padded copy)
  int64_t ld = 0;
  bool sparse = false;            // per-tile CSR below is populated
  int tiles = 0;                  // column tiles of 512
  int64_t nnz = 0;
  const int64_t* sp_ptr = nullptr;  // [tiles*N + 1]
  const void* sp_ent = nullptr;     // [nnz] (column-in-tile int32, value fp32)
  // packed rows (s3grl_packed.hip): per (tile, row) a 128-bit mask of the non-zero 16-byte chunks
  // of the 512-column tile + where its chunks start; the non-zero chunks back to back
  bool packed = false;
  const void* pk_hdr = nullptr;     // [tiles*N] PackedHdr
  const void* pk_data = nullptr;    // [chunks] float4
  int64_t pk_chunks = 0;            // non-zero chunks (of tiles*N*128 slots... F/4 real ones per row)
  std::vector<void*> owned;
};

// header of one (tile, row) of the packed operand: 32 bytes, read with one scalar load
struct PackedHdr {
  uint64_t m0, m1;   // bit j of m0: chunk j (columns 4j..4j+3 of the tile) is non-zero; m1: chunks 64..127
  uint64_t off;      // index of the row's first chunk in pk_data
  uint64_t pad;
};

struct s3grl_sop {
  s3grl_context* ctx = nullptr;
  const s3grl_graph* graph = nullptr;
  int32_t K = 0;
  int64_t F = 0;
  int64_t ldy = 0;        // leading dimension of Y_i (even)
  const float* mult = nullptr;  // [nnz] multiplicity of the stored entries (null: 1), see s3grl_sop_create_weighted
  double* dinv = nullptr; // [N] global D^-1/2
  double* Y = nullptr;    // [K+1, N, ldy] f64; Y[0] = X, Y[i] = Â Y[i-1]
  float* Yhi = nullptr;   // [K+1, N, ldy] the same rounded to f32 — what the per-link row kernel reads
  float* Ylo = nullptr;   // [K, N, ldy] Y[i] - Yhi[i] rounded to f32, i = 1..K (read where a row cancels)
  std::vector<void*> owned;
};

// Code-object preload (s3grl_context_preload): HIP loads a translation unit's code object at the first launch
// of one of its kernels — 16 of the 17 ms of a process's first s3grl_graph_create.  Every unit defines an
// empty kernel and a function that asks for its attributes, which loads the unit's code object on the spot.
#define S3GRL_DEFINE_TOUCH_(unit)                                                              \
  namespace s3grl {                                                                            \
  namespace {                                                                                  \
  __global__ void touch_kernel_##unit() {}                                                     \
  }                                                                                            \
  void touch_##unit() {                                                                        \
    hipFuncAttributes at;                                                                      \
    (void)hipFuncGetAttributes(&at, reinterpret_cast<const void*>(touch_kernel_##unit));       \
  }                                                                                            \
  }
#define S3GRL_DEFINE_TOUCH(unit) S3GRL_DEFINE_TOUCH_(unit)

namespace s3grl {

void touch_api();
void touch_structure();
void touch_links_a();
void touch_links_b();
void touch_links_c();
void touch_gather();
void touch_packed();
void touch_features();
void touch_sop();
void touch_pool();
void touch_relabel();
void touch_hub();
void touch_balls();
void touch_csr();

struct Transient {  // released on scope exit (stream-ordered reuse is safe: one stream per context)
  s3grl_context* ctx;
  std::vector<void*> ptrs;
  ~Transient() {
    for (void* p : ptrs) ctx->arena.release(p);
  }
};

// api.hip: the context's side streams and their fork / join events, created on first use
s3grl_status ensure_side_streams(s3grl_context* ctx);

// relabel.hip
s3grl_status build_degree_order(s3grl_context* ctx, s3grl_graph* g);
s3grl_status launch_translate_links(s3grl_context* ctx, const s3grl_graph* g, const int64_t* links, int64_t L,
                                    int64_t* out);
// perm [L]: the link indices ordered by (higher-degree endpoint, other endpoint) — the order in which a
// plan on a big graph works on its links (locality; outputs stay in the caller's order)
s3grl_status launch_link_order(s3grl_context* ctx, const int64_t* links, int64_t L, int64_t N,
                               const int32_t* indptr, int32_t* perm);
// (u64 key, i32 value) radix sort, instantiated in relabel.hip only (see there)
s3grl_status sort_pairs_u64_i32_bytes(s3grl_context* ctx, size_t n, size_t* bytes);
s3grl_status sort_pairs_u64_i32(s3grl_context* ctx, void* tmp, size_t bytes, uint64_t* keys_in, uint64_t* keys_out,
                                int32_t* vals_in, int32_t* vals_out, size_t n);
// segsort.hip
s3grl_status segmented_sort_i32(s3grl_context* ctx, void* tmp, size_t* bytes, const int32_t* keys_in, int32_t* keys_out,
                                size_t n, unsigned segments, const int64_t* seg);
// balls.hip
s3grl_status ensure_ball_cache(s3grl_context* ctx, s3grl_graph* g, int hops, bool* usable);
void release_ball_cache(s3grl_graph* g);
s3grl_status launch_count_balls(s3grl_context* ctx, const s3grl_graph* g, const int64_t* links, int64_t L, int hops,
                                int plus, int K, const int32_t* partner, const int32_t* mirror_of, int32_t* n_nodes,
                                int32_t* p_nodes, int32_t* n_rows, int32_t* n_jobs, int32_t* lvl_max,
                                int32_t* err_flag, int64_t* tot_nodes_alg, int32_t* stash, int slot,
                                int32_t* lvl_stash, const int32_t* perm);
// hub.hip
s3grl_status build_hub_cache(s3grl_context* ctx, s3grl_graph* g);
void release_hub_cache(s3grl_graph* g);
// LDS bytes of link_hub_kernel beyond its fixed part (n nodes, at most xcap edges outside the hub's cache)
__host__ __device__ inline int64_t hub_lds_need(int64_t n, int64_t xcap) {
  const int64_t state = 16 * n, build = 4 * n + 4 * xcap;
  return 8 * n + 16 + (state > build ? state : build) + 4 * ((xcap + 1) & ~(int64_t)1) + 2 * ((n + 1) & ~(int64_t)1);
}
// ... and of the hub's cached rows staged next to them (offsets and columns as uint16; a hub of more than
// 65 535 cached entries is read from HBM instead)
__host__ __device__ inline int64_t hub_stage_bytes(int64_t c, int64_t entries) {
  return entries <= 65535 ? 2 * ((c + 2) & ~(int64_t)1) + 2 * ((entries + 1) & ~(int64_t)1) : 0;
}
int hub_fixed_words(int cn_cap, int K);
int hub_class_bound(int cls, int cn_cap, int K);   // LDS bytes (beyond the fixed part) of hub class 0..kHubClasses-1
struct HubLinkArgs {
  const int32_t *indptr, *indices;
  HubCache hub;
  const int64_t* links;
  int plus, cn_cap;
  const int64_t* x_cap;
  const int64_t *node_off, *row_ptr, *job_off, *coef_off;
  const int32_t* mirror_of;
  int32_t* c_ids;
  float* c_coef;
  Job* jobs;
  float* job_z;
  int32_t* job_lim;
  int64_t* row_nodes;
  int32_t* lvl;
  unsigned long long *tot_edges, *tot_support, *tot_vol, *tot_oriented, *tot_hub_links, *tot_hub_bytes, *tot_hub_ends, *tot_hub_nodes;
  const int32_t* e_cap;      // 2 x the oriented-row entries of a link's subgraph (count1_kernel)
  const int32_t* old_of_new;
  int split_t, seg_shift;
  unsigned long long* dbg;   // diagnostic (S3GRL_DEBUG_STAMPS): cycles per phase, summed over workgroups
  uint32_t* slices;          // class kHubClasses (found edges beyond LDS): one HBM slice per resident workgroup
  int64_t slice_words;
  int slice_grid;
};
s3grl_status launch_hub_class(s3grl_context* ctx, const HubLinkArgs& a, int K, int cls, const int32_t* class_list,
                              int count, hipStream_t stream);
s3grl_status launch_sort_hops(s3grl_context* ctx, const s3grl_plan* p, int32_t* nodes);

// hub.hip, link_tiny_kernel — one-hop PoS links of at most kTinyNodes nodes (two rows per link: no common-
// neighbour rows), half a wavefront or a wavefront per link: one lane per node, the masked induced adjacency as
// one 32- / 64-bit mask per lane, no hash, no CSR, no block scans, no barriers.  link_full_kernel spends ~2 000 wave
// instructions on such a link with a third of its lanes at work (config 5: 763 000 of them, the vector unit
// 93 % busy).  Class list kTinyList of classify_kernel.
constexpr int kTinyNodes = 64;   // (up to 32: half a wavefront per link, class list kTinyList; up to 64: a wavefront, kTinyList + 1)
struct TinyLinkArgs {
  const int32_t *indptr, *indices, *fwd_indptr, *fwd_indices;
  const int64_t* links;
  const int64_t *node_off, *row_ptr, *job_off, *coef_off;
  const int32_t* mirror_of;
  int32_t* c_ids;
  float* c_coef;
  Job* jobs;
  float* job_z;
  int32_t* job_lim;
  int64_t* row_nodes;
  int32_t* lvl;
  unsigned long long *tot_edges, *tot_support, *tot_vol;
  const int32_t* old_of_new;
};
s3grl_status launch_tiny_class(s3grl_context* ctx, const TinyLinkArgs& a, int K, int width, const int32_t* class_list,
                               int count, hipStream_t stream);

// csr.hip — links whose every operator reaches the whole subgraph (sign_k - 1 >= num_hops: p == n), on
// graphs of the bitmap flavour with cached balls: the masked induced adjacency is built ONCE per link as a
// CSR of 16-bit list positions in LDS (member counts per row by a sizing kernel of their own, so that the
// classes are cut by the EXACT LDS need), and all K operators are pulls over it — instead of walking the
// global rows through the bitmaps once per operator (link_kernel: three full walks at PubMed sign_k = 5).
// LDS of a link beyond the fixed part: row offsets (uint16 [n + 1]), columns (uint16 [e]) and a region
// that holds the build's bitmaps / list / rank map first and the two float2 state arrays afterwards.
constexpr int kCsrBase = 32;             // class lists kCsrBase .. kCsrBase + kCsrClasses - 1 (class_count idx alike)
// A link holds the LDS of its CLASS, and the kernel's speed follows the links in flight (measured: +35 % LDS
// per link = +0.8 ms on PubMed sign_k = 5): finer classes than the six of link_kernel
constexpr int kCsrClasses = 14;
#define S3GRL_CSR_CLASS_BOUNDS {4096, 6144, 8192, 12288, 16384, 20480, 24576, 32768, 40960, 49152, 65536, 98304, 131072, 163840}
struct CsrBounds {
  int b[kCsrClasses];
};
constexpr int kCsrDinvTable = 256;       // D^-1/2 of degrees below this from an LDS table
__host__ __device__ inline int csr_fixed_words(int cn_cap, int K) {
  return 2 * cn_cap + kMaxLevels + 4 * K + 32 + kCsrDinvTable;
}
__host__ __device__ inline int csr_lds_need(int n, int e, int W) {
  const int off_b = 2 * ((n + 2) & ~1), col_b = 2 * ((e + 1) & ~1);
  const int build = 8 * W + 4 * n + 4 * ((n + 2) & ~1), pass = 16 * n;
  return ((off_b + col_b + 7) & ~7) + (build > pass ? build : pass) + 8;
}
struct CsrLinkArgs {
  const int32_t *indptr, *indices;   // the degree-ordered graph
  int W, hops;
  const uint32_t* balls;             // level `hops` of the ball cache (ball_hops(x) for every x), W words per node
  const int64_t* links;              // translated into the degree order
  int plus, cn_cap;
  const uint16_t* cnt;               // [Σn] member neighbours of every list entry (csr_count_kernel)
  const int32_t* csr_e;              // [L] their sum per link, -1: not a link of this flavour
  const int64_t *node_off, *row_ptr, *job_off, *coef_off;
  const int32_t* mirror_of;
  int32_t* c_ids;
  float* c_coef;
  Job* jobs;
  float* job_z;
  int32_t* job_lim;
  int64_t* row_nodes;
  int32_t* lvl;
  unsigned long long *tot_edges, *tot_support, *tot_vol;
  const int32_t* stash;
  int slot;
  const int32_t* old_of_new;
  int split_t, seg_shift;
  unsigned long long* dbg;
};
int csr_class_bound(int cls, int cn_cap, int K);   // LDS bytes beyond the fixed part of class 0..kCsrClasses-1
bool csr_mode_for(const s3grl_graph* g, int hops, int K, bool balls, bool plain);
s3grl_status launch_csr_count(s3grl_context* ctx, const s3grl_graph* g_walk, const int64_t* links, int64_t L, int hops,
                              const int32_t* n_nodes, const int64_t* node_off, const int32_t* lvl,
                              const int32_t* stash, int slot, const int32_t* perm, uint16_t* cnt, int32_t* csr_e);
s3grl_status launch_csr_class(s3grl_context* ctx, const CsrLinkArgs& a, int K, int cls, const int32_t* class_list,
                              int count, hipStream_t stream);

// structure.hip
s3grl_status launch_count(s3grl_context* ctx, const s3grl_graph* g, const int64_t* links, int64_t L,
                          int hops, int plus, int K, WalkSets ws,
                          const int32_t* partner, const int32_t* mirror_of, int32_t* n_nodes, int32_t* p_nodes,
                          int32_t* n_rows, int32_t* n_jobs, int32_t* lvl_max, int32_t* err_flag,
                          int64_t* tot_nodes_alg, HopSampling smp = HopSampling{1.0, 0, 0},
                          int32_t* stash = nullptr, int slot = 0, int32_t* lvl_stash = nullptr,
                          const int32_t* perm = nullptr);
// union of two sorted adjacency structures of the same nodes (a directed graph's successors and
// predecessors): out_indptr [N+1] / *out_indices (arena-owned) / *nnz
s3grl_status build_union_graph(s3grl_context* ctx, int64_t N, const int32_t* a_indptr, const int32_t* a_indices,
                               const int32_t* b_indptr, const int32_t* b_indices, int32_t* u_indptr,
                               int32_t** u_indices, int64_t* u_nnz);
int num_class_lists();
s3grl_status launch_random_walks(s3grl_context* ctx, const s3grl_graph* g, int m, int M,
                                 uint32_t seed, int32_t* raw);
// structure check of a caller's node sets: bit 0 set_ptr not monotone from 0, bit 1 an id outside [0,N)
s3grl_status launch_validate_sets(s3grl_context* ctx, const int64_t* set_ptr, const int32_t* set_nodes,
                                  int64_t num_sets, int64_t total, int64_t num_nodes,
                                  int64_t* flags /* device, zeroed */);
// per start node the sorted unique nodes of its M walks of length m, the start included (the cache
// reference utils.create_rw_cache builds, utils.py:425-443)
s3grl_status launch_walk_sets(s3grl_context* ctx, const s3grl_graph* g, const int64_t* starts, int64_t num_starts,
                              int m, int M, uint32_t seed, int64_t* set_ptr, int32_t* set_nodes);
// folds a reversed duplicate (dst,src) of a link (src,dst) into it: partner[l] = primary of a
// folded link (else -1), mirror_of[l] = the link folded into l (else -1)
int64_t mirror_table_slots(int64_t L);
s3grl_status launch_find_mirrors(s3grl_context* ctx, const int64_t* links, int64_t L, int64_t N,
                                 uint64_t* keys, int32_t* vals, int64_t slots, int32_t* partner,
                                 int32_t* mirror_of, int64_t* n_mirrored);
s3grl_status launch_mirror_rows(s3grl_context* ctx, const int32_t* partner, int64_t L,
                                int32_t* n_rows);
s3grl_status launch_job_order(s3grl_context* ctx, const int32_t* n_nodes, const int32_t* n_jobs,
                              const int64_t* job_off, int64_t L, int32_t* hist, int32_t* job_order,
                              const int32_t* perm = nullptr, int32_t* scratch_cnt = nullptr,
                              int64_t* scratch_off = nullptr, int64_t* scan_ws = nullptr);
int64_t scan_workspace_elems(int64_t n);
s3grl_status launch_scan_i32_to_i64(s3grl_context* ctx, const int32_t* in, int64_t n, int64_t* out,
                                    int64_t* workspace);
// exclusive scans of three int32 [n] arrays -> int64 [n+1] in one set of launches (n > 0);
// workspace3: 3 x scan_workspace_elems(n); max0 / max1 (device int64, zeroed; may be null): maxima of
// in0 / in1; totals (device int64 [3]; may be null): the three totals
s3grl_status launch_scan3(s3grl_context* ctx, const int32_t* in0, const int32_t* in1, const int32_t* in2,
                          int64_t n, int64_t* out0, int64_t* out1, int64_t* out2, int64_t* workspace3,
                          int64_t* max0, int64_t* max1, int64_t* totals);
s3grl_status launch_classify(s3grl_context* ctx, const s3grl_graph* g, int cn_cap, int K,
                             const int32_t* n_nodes, const int32_t* p_nodes,
                             const int32_t* lvl_max, int64_t L, int32_t* class_count,
                             int32_t* class_list, bool allow_hash = true, const int32_t* e_cap = nullptr,
                             int stash_slot = 0, const int32_t* perm = nullptr, const int64_t* x_cap = nullptr,
                             const int32_t* csr_e = nullptr, bool tiny_ok = false);
// one-hop plans on big graphs (s3grl_onehop.inl): degree-oriented rows of the graph, and the
// sizing pass that needs no bitmaps
bool sparse_mode_for(const s3grl_graph* g);
bool onehop_mode_for(const s3grl_graph* g);
s3grl_status build_forward_rows(s3grl_context* ctx, s3grl_graph* g);
s3grl_status launch_count1(s3grl_context* ctx, const s3grl_graph* g, const int64_t* links, int64_t L,
                           int plus, int K, const int32_t* partner, const int32_t* mirror_of,
                           int32_t* n_nodes, int32_t* p_nodes, int32_t* n_rows, int32_t* n_jobs,
                           int32_t* lvl_max, int32_t* e_cap, int32_t* err_flag, int64_t* tot_nodes_alg,
                           int64_t* tot_oriented, const int32_t* perm = nullptr, int64_t* x_cap = nullptr);
s3grl_status launch_links(s3grl_context* ctx, const s3grl_graph* g, const int64_t* links, int64_t L,
                          const int32_t* class_list, const int32_t* class_count_host, int hops,
                          int plus, int cn_cap, int full_stats, int K, WalkSets ws, const int32_t* p_nodes,
                          const int64_t* node_off, const int64_t* row_ptr, const int64_t* job_off,
                          const int64_t* coef_off, const int32_t* mirror_of, int32_t* c_ids,
                          float* c_coef, Job* jobs, float* job_z, int32_t* job_lim, int64_t* row_nodes,
                          int32_t* lvl,
                          int64_t* tot_edges, int64_t* tot_support, int64_t* tot_vol,
                          HopSampling smp = HopSampling{1.0, 0, 0}, const int32_t* stash = nullptr,
                          int slot = 0, const int32_t* e_cap = nullptr, int64_t max_nodes = 0,
                          const int32_t* old_of_new = nullptr, const int32_t* new_of_old = nullptr,
                          int split_t = 0, int seg_shift = 0, const int64_t* x_cap = nullptr,
                          const uint16_t* csr_cnt = nullptr, const int32_t* csr_e = nullptr, int sop2 = 0);
// pieces of the split jobs: piece_off [njobs + 1] (device) and *total (device scalar) first, the
// piece arrays once the host knows the total
s3grl_status launch_split_count(s3grl_context* ctx, const Job* jobs, int64_t njobs, int seg_shift,
                                int32_t* cnt, int64_t* piece_off, int64_t* scan_ws);
s3grl_status launch_split_fill(s3grl_context* ctx, const Job* jobs, const int32_t* job_lim, int64_t njobs,
                               const int32_t* job_order, int K, int seg_shift, const int64_t* piece_off,
                               int64_t npieces, Job* gjobs, int32_t* g_lim, int32_t* g_order, int32_t* piece_job);
// rows of the split jobs = f64 sum of their pieces' partial rows (+ operator 0, label column, mirror)
s3grl_status launch_combine(s3grl_context* ctx, const s3grl_plan* p, const float* prows, const float* X,
                            int64_t ldx, int64_t F, float* rows);
s3grl_status launch_dists(s3grl_context* ctx, const int64_t* node_off, const int32_t* lvl, int64_t L,
                          int8_t* dists);
// gather.hip
s3grl_status launch_gather(s3grl_context* ctx, const GatherView& v, const int32_t* c_ids,
                           const float* c_coef, int K, const float* X, int64_t ldx, int64_t F, float* rows,
                           bool in_job_order = false);
// features.hip
s3grl_status build_packed_rows(s3grl_context* ctx, s3grl_features* f, double max_density);
s3grl_status launch_gather_packed(s3grl_context* ctx, const s3grl_plan* p, const GatherView& v,
                                  const s3grl_features* f, float* rows);
s3grl_status launch_gather_traffic(s3grl_context* ctx, const s3grl_plan* p, const s3grl_features* f,
                                   unsigned long long* d_out /* [8] device, zeroed */);
s3grl_status launch_gather_sparse(s3grl_context* ctx, const s3grl_plan* p, const GatherView& v,
                                  const s3grl_features* f, float* rows);
s3grl_status launch_copy_pad(s3grl_context* ctx, const float* X, int64_t ldx, int64_t N, int64_t F,
                             float* Y, int64_t ldy);

}  // namespace s3grl
