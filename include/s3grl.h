/*
 * s3grl.h — C ABI of libs3grl_hip.so, the MI355X (gfx950) S3GRL operator-precompute engine.
 *
 * This is the drop-in boundary for ONE path of venomouscyanide/S3GRL: the per-link PoS /
 * PoS Plus / SoP operator precompute that `utils.extract_enclosing_subgraphs`
 * (reference utils.py:446-554) dispatches to
 *     OptimizedSignOperations.get_PoS_prepped_ds       (reference tuned_SIGN.py:137-189)
 *     OptimizedSignOperations.get_PoS_Plus_prepped_ds  (reference tuned_SIGN.py:192-262)
 *     OptimizedSignOperations.get_SoP_prepped_ds       (reference tuned_SIGN.py:49-134)
 * including the k-hop extraction they call (reference utils.py:33-85) and, for SoP, the global
 * operator setup of `SEALDataset.process` (reference sgrl_link_pred.py:161-178).
 *
 * The reference is pure Python and has no FFI of its own; a maintainer binds these entry
 * points with ctypes from `tuned_SIGN.py` (stub in INTEGRATION.md).  All pointers are plain
 * DEVICE pointers unless a parameter says "host"; no torch / PyG types cross this boundary.
 * Every function returns an s3grl_status; nothing is printed, nothing throws.
 *
 * Output contract (what reference models.py:372 `torch.cat(xs, dim=-1)` feeds the MLP):
 *     rows     fp32 [total_rows, sign_k+1, 1+F]   operator 0 is x itself, column 0 the label
 *                                                  column z (1 for src/dst rows, else 0)
 *     row_ptr  int64 [L+1]                         rows of link l are row_ptr[l]..row_ptr[l+1];
 *                                                  first two are src, dst; PoS Plus appends the
 *                                                  common-neighbour rows in ascending node id
 * Thread-compatibility: one context per host thread / stream; contexts share nothing.
 */
#ifndef S3GRL_H_
#define S3GRL_H_

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define S3GRL_ABI_VERSION 6

typedef enum s3grl_status {
  S3GRL_OK = 0,
  S3GRL_ERR_INVALID_ARGUMENT = 1, /* null pointer, negative size, sign_k < 1, link id out of range,
                                     malformed CSR (indptr not monotone, column id outside
                                     [0,N), a row not strictly ascending) */
  S3GRL_ERR_NOT_IMPLEMENTED = 2,  /* maps to the reference's NotImplementedError:
                                     k_node_set_strategy other than "intersection"
                                     (tuned_SIGN.py:235; "union" is unusable as shipped),
                                     SoP on a directed graph */
  S3GRL_ERR_NO_FEATURES = 3,      /* X == NULL: the reference's `assert subgraph_features is not
                                     None` (tuned_SIGN.py:166,221) */
  S3GRL_ERR_OUT_OF_MEMORY = 4,
  S3GRL_ERR_HIP = 5,              /* a HIP runtime call failed; see s3grl_last_error() */
  S3GRL_ERR_NO_DEVICE = 6,        /* no gfx950 device visible */
  S3GRL_ERR_GRAPH_TOO_LARGE = 7,  /* nnz or num_nodes >= 2^31; a SoP ball (radius ceil(K/2), or one less for
                                     odd K) that does not fit LDS.  No plan has a node limit: beyond ~327 680
                                     nodes the N-bit bitmaps of the sizing pass, of the overflow class and of
                                     the SoP scalar kernel live in HBM slices instead of LDS */
  S3GRL_ERR_SELF_LINK = 8         /* src == dst: the reference duplicates the node; unsupported */
} s3grl_status;

typedef enum s3grl_mode {
  S3GRL_MODE_POS = 0,      /* get_PoS_prepped_ds       : rows {src,dst} */
  S3GRL_MODE_POS_PLUS = 1, /* get_PoS_Plus_prepped_ds  : rows {src,dst} + N(src) ∩ N(dst) */
  S3GRL_MODE_SOP = 2,      /* get_SoP_prepped_ds       : global operators, rows {src,dst} */
  S3GRL_MODE_SOP_RESTRICTED = 3 /* NOT a reference flow (its SoP ignores num_hops, tuned_SIGN.py:49-134): the SoP rows
                              with every operator row restricted to the num_hops-ball of {src,dst} — the optional
                              twin BASELINE config 3 ("2-hop subgraphs") and SURVEY §8(d) name, reported separately.
                              x_i[s] = [Â^i[s,s] | Σ_{w in ball, w != d} Â^i[s,w] X[w]] with the GLOBAL Â; through
                              s3grl_plan_create / s3grl_run like PoS; needs sign_k - 1 <= num_hops */
} s3grl_mode;

typedef enum s3grl_strategy {
  S3GRL_STRATEGY_INTERSECTION = 0, /* sign_kwargs['k_node_set_strategy'] == 'intersection' */
  S3GRL_STRATEGY_UNION = 1         /* accepted by the reference's parser, broken in its code */
} s3grl_strategy;

/* s3grl_cfg.flags */
#define S3GRL_FLAG_FULL_STATS 1u /* per-link diagnostics: exact total_sub_edges even when
                                    sign_k < num_hops, subgraph export for every link (turns
                                    the folding of reversed duplicates off) */
#define S3GRL_FLAG_NO_FOLD 2u    /* do not serve (d,s) from the extraction of (s,d) */
#define S3GRL_FLAG_COUNT_ONLY 4u /* sizing pass only: the plan holds the subgraph sizes (node_ptr of
                                    s3grl_plan_export_subgraphs, total_nodes / max_nodes / total_rows
                                    of the stats) and nothing else; it cannot be run.  Used to
                                    balance the shards of a multi-GPU job by exact subgraph size.
                                    Reversed duplicates are folded like in a full plan (size 0 for
                                    the folded link) unless S3GRL_FLAG_NO_FOLD is set too */

/* sign_kwargs / call arguments of the reference operators (tuned_SIGN.py:137-138,145,200,229) */
typedef struct s3grl_cfg {
  int32_t mode;      /* s3grl_mode */
  int32_t num_hops;  /* k of the k-hop enclosing subgraph (ignored for SoP, like the reference) */
  int32_t sign_k;    /* number of operators K >= 1 */
  int32_t strategy;  /* s3grl_strategy, PoS Plus only */
  int32_t directed;  /* 1 iff the graph was made by s3grl_graph_create_directed (A and A_csc) */
  uint32_t flags;    /* S3GRL_FLAG_* */
  int32_t rw_m;      /* ScaLed subgraphs (reference utils.py:86-150, rw_kwargs): rw_M random walks */
  int32_t rw_M;      /*   of length rw_m per node, drawn by the engine, replace the BFS (num_hops is
                          then ignored, like in the reference); 0 = k-hop BFS.  Walk node sets cached
                          by the caller: s3grl_plan_create_sets */
  uint32_t seed;     /* seed of the engine's own counter-based generator (walks, hop sampling) */
  int32_t max_nodes_per_hop; /* utils.py:68-70: keep at most this many nodes of every hop;
                                0 = no cap (None) */
  double ratio_per_hop;      /* utils.py:66-67: keep int(ratio * |fringe|) nodes of every hop
                                (before the cap), uniformly; >= 1.0 = keep all (every paper
                                config).  A double, like the Python float whose product is
                                truncated.  Offset 40. */
  int32_t reserved[4];       /* must be 0 */
} s3grl_cfg;

/* sizes a plan measured while extracting; the benchmark's algorithmic-bytes figure
 * (B_link = 8n + 4 vol(S) + 4 n F + 4 R (K+1)(1+F)) is computed from these. */
typedef struct s3grl_plan_stats {
  int64_t num_links;
  int64_t total_rows;       /* ΣR */
  int64_t total_nodes;      /* Σ n        subgraph nodes over all links */
  int64_t total_volume;     /* Σ vol(S)   global degrees of those nodes */
  int64_t total_sub_edges;  /* Σ e        directed entries of the masked induced sub-CSRs */
  int64_t total_support;    /* Σ over row pairs of nodes with a non-zero operator coefficient */
  int64_t num_row_pairs;    /* gather jobs */
  int64_t max_nodes;        /* max n */
  int64_t workspace_bytes;  /* device bytes held by the plan + context arena */
  int64_t folded_links;     /* links (d,s) served by the extraction of their reversed duplicate
                               (s,d) earlier in the list: same subgraph, rows swapped.  The
                               totals above count them like any other link (algorithmic). */
  int64_t extracted_nodes;  /* Σ n over the links actually extracted */
  int64_t oriented_entries; /* one-hop plans on big graphs: Σ over the extracted links of the degree-oriented
                               row entries of their subgraph's nodes, hub_links excepted (what link_full_kernel probes: the
                               physical counterpart of total_volume); 0 for other plans */
  int64_t hub_links;        /* ... of which: links served from a cached hub neighbourhood (link_hub_kernel:
                               the induced adjacency of a hub's neighbours is built once per graph and
                               shared by all of the hub's links); their oriented rows are NOT probed */
  int64_t hub_read_bytes;   /* bytes those links requested: the two endpoint rows, the rows of the nodes
                               only the non-hub endpoint brings (bounds + entries), the cached rows staged */
  int64_t hub_endpoint_entries; /* Σ deg(src) + deg(dst) over those links (their share of the endpoint rows) */
  int64_t hub_nodes;        /* Σ n over those links (their share of extracted_nodes) */
} s3grl_plan_stats;

typedef struct s3grl_context s3grl_context; /* device, stream, workspace arena */
typedef struct s3grl_graph s3grl_graph;     /* structure of the train graph A (values ignored,
                                               exactly as ssp.find -> SparseTensor(row, col)
                                               ignores them, tuned_SIGN.py:153-156) */
typedef struct s3grl_plan s3grl_plan;       /* extraction + operator coefficients of one call */
typedef struct s3grl_sop s3grl_sop;         /* SoP global state: Â, Y_i = Â^i X */
typedef struct s3grl_features s3grl_features; /* X prepared for the gather (dense or sparse rows) */

int32_t s3grl_abi_version(void);
const char* s3grl_status_string(s3grl_status s);
/* message of the last failing call on this host thread (HIP error text), never NULL */
const char* s3grl_last_error(void);

/* `stream` is a hipStream_t passed as void* (NULL = the legacy default stream). */
s3grl_status s3grl_context_create(int32_t device, void* stream, s3grl_context** out);
s3grl_status s3grl_context_destroy(s3grl_context* ctx);
/* Loads the library's GPU code now instead of at the first launch of each kernel family (HIP loads a code
 * object lazily: ~16 ms of a process's first s3grl_graph_create and ~7-20 ms of its first plan + run are that,
 * whatever the size of the graph).  units: bit 0 = what every PoS / PoS Plus call at sign_k 3 or 4 needs, bit 1 =
 * the link kernels of the other sign_k, bit 2 = SoP and the pooling; ms (host double [16], may be NULL) =
 * milliseconds per unit.  Optional — nothing depends on it; a host thread can call it while the caller still
 * loads its dataset (s3grl_amd.tuned_SIGN does at import). */
s3grl_status s3grl_context_preload(s3grl_context* ctx, uint32_t units, double* ms);

/* CSR of A as scipy holds it (reference sgrl_link_pred.py:111-114): indptr [N+1], indices
 * [nnz] strictly ascending within a row (canonical format: sorted, duplicates summed); the
 * matrix must be structurally symmetric (undirected train graph).  The arrays are copied; the
 * caller may free them afterwards.  The structure is validated on the device (indptr[0] == 0,
 * indptr monotone and ending at nnz, ids in [0,N), rows strictly ascending):
 * S3GRL_ERR_INVALID_ARGUMENT otherwise.  Symmetry is the caller's promise (the Python binding
 * checks it). */
s3grl_status s3grl_graph_create(s3grl_context* ctx, int64_t num_nodes, const int64_t* indptr,
                                const int32_t* indices, int64_t nnz, s3grl_graph** out);
/* A DIRECTED graph, as the reference holds it when `directed` is set (sgrl_link_pred.py:107-119: A from the
 * directed edge_index, A_csc = A.tocsc(); only ogbl-citation2 of the reference's datasets): both forms
 * of the same nnz arcs — csr_* the successors of every node, csc_* the predecessors (scipy's CSC arrays
 * as they are: indptr over columns, row ids ascending).  Plans on such a graph need s3grl_cfg.directed
 * = 1: the BFS follows successors and predecessors (utils.py:58-63), the induced matrix keeps the
 * directions, D = out-degrees (tuned_SIGN.py:158-161).  That the two forms describe the same arcs is
 * the caller's promise.  SoP is not available on directed graphs (S3GRL_ERR_NOT_IMPLEMENTED). */
s3grl_status s3grl_graph_create_directed(s3grl_context* ctx, int64_t num_nodes, const int64_t* csr_indptr,
                                         const int32_t* csr_indices, const int64_t* csc_indptr,
                                         const int32_t* csc_indices, int64_t nnz, s3grl_graph** out);
s3grl_status s3grl_graph_destroy(s3grl_graph* g);

/* PoS / PoS Plus, feature-independent half: BFS to num_hops from {src,dst} on the unmasked
 * graph, induced sub-CSR with the target link removed, D^-1/2 A D^-1/2, and rows {src,dst}
 * (+ common neighbours) of Â^1..Â^K by row propagation.  `links` is int64 [L,2] (the
 * reference iterates link_index.t().tolist(), tuned_SIGN.py:147). */
s3grl_status s3grl_plan_create(s3grl_context* ctx, const s3grl_graph* g, const int64_t* links,
                               int64_t num_links, const s3grl_cfg* cfg, s3grl_plan** out);

/* ScaLed subgraphs from node sets the CALLER cached (reference utils.py:94-108: with rw_kwargs the
 * subgraph of (src,dst) is torch.unique(cat(cache[src], cache[dst])) of rw_kwargs['cached_pos_rws'] /
 * ['cached_neg_rws'] — dict node -> unique nodes of its walks, built by utils.create_rw_cache,
 * utils.py:425-443, sgrl_link_pred.py:123-128 — or rw_kwargs['unique_nodes'][(src,dst)]; src and dst
 * first, utils.py:134-135).  A CSR of sets on the device:
 *   per_link == 0: num_sets == the graph's num_nodes, set i belongs to NODE i (empty for nodes the
 *                  cache does not hold); subgraph of (s,d) = {s,d} ∪ set[s] ∪ set[d];
 *   per_link == 1: num_sets == num_links, set l belongs to LINK l; subgraph = {s,d} ∪ set[l].
 * Nodes beyond {src,dst} form "hop 1" (dists 0,0,1,1,..., utils.py:145-146); num_hops, rw_m, rw_M and
 * the per-hop sampling fields of the cfg play no part (rw_m / rw_M must be 0).  The arrays are read
 * while the plan is created, not afterwards.  S3GRL_ERR_INVALID_ARGUMENT for a malformed set_ptr
 * or an id outside [0, num_nodes). */
typedef struct s3grl_node_sets {
  const int64_t* set_ptr;   /* device int64 [num_sets + 1], monotone from 0 to num_set_nodes */
  const int32_t* set_nodes; /* device int32 [num_set_nodes] (duplicates and src / dst themselves allowed) */
  int64_t num_sets;
  int64_t num_set_nodes;
  int32_t per_link;         /* 0 or 1 */
  int32_t reserved;         /* must be 0 */
} s3grl_node_sets;
s3grl_status s3grl_plan_create_sets(s3grl_context* ctx, const s3grl_graph* g, const int64_t* links,
                                    int64_t num_links, const s3grl_cfg* cfg, const s3grl_node_sets* sets,
                                    s3grl_plan** out);
/* The producer of such a cache, in place of reference utils.create_rw_cache (utils.py:425-443:
 * torch_cluster random_walk from every start node repeated rw_M times, walk length rw_m, unique per
 * start): for start i the ascending unique nodes of its rw_M uniform walks of length rw_m, the start
 * itself included, in set_nodes[set_ptr[i] .. set_ptr[i+1]).  starts: device int64 [num_starts];
 * set_ptr: device int64 [num_starts + 1] out; set_nodes: device int32 out with room for
 * num_starts * (rw_m * rw_M + 1) entries.  The walks are those a plan with the same cfg.rw_m / rw_M /
 * seed draws by itself (counter-based generator keyed by seed, node, walk, step): same distribution
 * as torch_cluster's, other random numbers.  rw_m * rw_M + 1 <= 8192. */
s3grl_status s3grl_walk_sets(s3grl_context* ctx, const s3grl_graph* g, const int64_t* starts,
                             int64_t num_starts, int32_t rw_m, int32_t rw_M, uint32_t seed,
                             int64_t* set_ptr, int32_t* set_nodes);

s3grl_status s3grl_plan_destroy(s3grl_plan* p);
/* host struct out.  Plan creation does not wait for its last kernels (the gather of s3grl_run* is
 * queued right behind them): total_sub_edges / total_support / total_volume are read back here, i.e.
 * this call waits for the context's stream when it is the first to ask. */
s3grl_status s3grl_plan_get_stats(const s3grl_plan* p, s3grl_plan_stats* out);
/* ΣR alone (host int64 out): what the caller sizes `rows` by; never waits */
s3grl_status s3grl_plan_total_rows(const s3grl_plan* p, int64_t* total_rows);
/* what the host knows of a plan the moment it exists (host int64 [4] out; never waits): [0] links,
 * [1] total rows ΣR, [2] links folded into their reversed duplicate, [3] row pairs (gather jobs) */
s3grl_status s3grl_plan_counts(const s3grl_plan* p, int64_t* what);
/* device int64 [L+1] out */
s3grl_status s3grl_plan_row_ptr(const s3grl_plan* p, int64_t* row_ptr);
/* device int64 [total_rows] out: global node id of every output row */
s3grl_status s3grl_plan_row_nodes(const s3grl_plan* p, int64_t* row_nodes);
/* parity hook: node_ptr int64 [L+1] (always), then, when non-NULL, nodes int32 [Σn] per link in
 * hop-major order, ascending id inside a hop, and dists int8 [Σn] (hop distance from
 * {src,dst}) — the quantities reference utils.k_hop_subgraph returns as `nodes`, `dists`
 * (utils.py:53-54,73-74; its order inside a hop is CPython set order). */
s3grl_status s3grl_plan_export_subgraphs(const s3grl_plan* p, int64_t* node_ptr, int32_t* nodes,
                                         int8_t* dists);

/* Relative cost of every link (device fp32 [L] out), in arbitrary units, from the sizes the plan
 * measured — what a multi-GPU caller balances its shards by (a count-only plan is enough).  The
 * model follows the kernels and was fitted on MI355X: multi-hop plans  pairs * n + 400  (gather and
 * row walks grow with the subgraph and with the row pairs of PoS Plus, ~400 nodes' worth of fixed
 * work per link); one-hop plans on big graphs  e_bound + 220 + (pairs - 1) * n  with e_bound the
 * bound of the induced entries the sizing pass derives from the degree-oriented rows (a hub-rich
 * positive costs several times a random negative of the collab-scale workload; node counts alone
 * say 1.5x) — except for the links served from a cached hub neighbourhood (plan statistics: hub_links),
 * whose oriented rows are not probed:  12.1 * n + 1380 + (pairs - 1) * n  (both fitted with
 * tools/cost_fit_onehop.py: the two kinds of links timed apart, two size buckets each; round 4, after
 * link_tiny_kernel and gather_narrow_kernel: profiles/r04_cost_fit_onehop.txt).  A reversed duplicate (d,s) that the plan folds into its primary (s,d) costs 250: its two
 * output rows and bookkeeping (least squares over the shards of 2-, 4- and 8-way splits of PubMed:
 * 410 / 290 at sign_k = 3, 310 / 260 at sign_k = 5) — so the cost of a pair is what ONE rank pays for it when both directions are kept
 * together (s3grl_amd.parallel.shard_assignment); with S3GRL_FLAG_NO_FOLD every link is priced in full. */
s3grl_status s3grl_plan_link_cost(const s3grl_plan* p, float* cost);

/* PoS / PoS Plus, feature half: rows[r, i, :] = [z | Σ_w Â^i[row r, w] X[w, :]].
 * X fp32 [N, F] row-major with leading dimension ldx (elements); rows fp32
 * [total_rows, K+1, 1+F] dense.  Asynchronous on the context's stream. */
s3grl_status s3grl_run(s3grl_context* ctx, const s3grl_plan* p, const float* X, int64_t ldx,
                       int64_t num_features, float* rows);

/* The feature operand x of the reference operators (dense fp32 [N,F], utils.py:83), prepared
 * once: 16-byte aligned rows (borrowed when X already is: keep X alive and unchanged while the
 * handle is in use).  flags: 0 = let the engine choose: when at most half of X's 16-byte
 * chunks hold a non-zero (bag-of-words / TF-IDF / one-hot rows) it also keeps a packed copy —
 * per row a bit mask of the non-zero chunks + those chunks — and the gather fetches only them;
 * same sums bit for bit.  1 = dense rows only; 4 = packed rows whatever the density;
 * 2 = (column, value)-pair rows accumulated in LDS (kept for comparison: slower than both). */
s3grl_status s3grl_features_create(s3grl_context* ctx, const float* X, int64_t ldx,
                                   int64_t num_nodes, int64_t num_features, int32_t flags,
                                   s3grl_features** out);
s3grl_status s3grl_features_destroy(s3grl_features* f);
/* host outs (either may be NULL): stored non-zeros (0 when never counted), sparse rows in use */
s3grl_status s3grl_features_info(const s3grl_features* f, int64_t* nnz, int32_t* is_sparse);
/* same contract as s3grl_run, with a prepared operand */
s3grl_status s3grl_run_features(s3grl_context* ctx, const s3grl_plan* p, const s3grl_features* f,
                                float* rows);

/* SoP: one-off global setup (reference sgrl_link_pred.py:161-178 recomputes it per split):
 * Â = D^-1/2 A D^-1/2 of the whole graph and Y_i = Â^i X, i = 1..K. */
s3grl_status s3grl_sop_create(s3grl_context* ctx, const s3grl_graph* g, const float* X,
                              int64_t ldx, int64_t num_features, int32_t sign_k, s3grl_sop** out);
/* The same on a MULTIGRAPH: the reference builds its global operator from the uncoalesced edge_index
 * (sgrl_link_pred.py:161-172: `SparseTensor(row, col)`, degree = entries per row), so a pair that
 * occurs m times counts m times in the degree and weighs m in every product, while the CSR of A
 * holds it once (scipy sums duplicates).  multiplicity: device fp32 [nnz], aligned with the graph's
 * `indices` as handed to s3grl_graph_create (NULL = all 1 = s3grl_sop_create); copied.
 * A_hat = D^-1/2 M D^-1/2 with D = row sums of M.  Every paper dataset is coalesced (m = 1). */
s3grl_status s3grl_sop_create_weighted(s3grl_context* ctx, const s3grl_graph* g, const float* X,
                                       int64_t ldx, int64_t num_features, int32_t sign_k,
                                       const float* multiplicity, s3grl_sop** out);
s3grl_status s3grl_sop_destroy(s3grl_sop* s);
/* the global SIGN features themselves, out fp32 [K, N, F]: out[i-1] = Â^i X.  This is what the
 * reference's non-optimised twin `TunedSIGN.__call__` (tuned_SIGN.py:18-23, PyG SIGN(K)) computes
 * for the graph it is handed. */
s3grl_status s3grl_sop_features(s3grl_context* ctx, const s3grl_sop* s, float* out);
/* rows fp32 [2L, K+1, 1+F]: x_i[src] = [Â^i[s,s] | Σ_{w != d} Â^i[s,w] X[w]] and the mirror
 * image for dst (reference tuned_SIGN.py:71-78,92-113,119-132). */
s3grl_status s3grl_sop_run(s3grl_context* ctx, const s3grl_sop* s, const int64_t* links,
                           int64_t num_links, float* rows);

/* Consumer-side centre / common-neighbour pooling, reference SIGNNet._centre_pool_helper
 * (models.py:339-369) on the engine's layout: h fp32 [total_rows, H] (the output of
 * operator_diff), rows of link b = row_ptr[b]..row_ptr[b+1], the first two being src, dst.
 * mode 0 (k_heuristic == 0): out [B, H] = h[src] * h[dst];  mode 1 / 2 (k_pool_strategy 'mean' /
 * 'sum'): out [B, 2H] = [h[src] * h[dst] | mean or sum of the remaining rows], zeros when a link
 * has none (the reference's `size=B`).  'concat' (models.py:363-367) is mode 0 plus a strided
 * view of the k rows, done by the Python binding (s3grl_amd/pool.py).  No host sync (the
 * reference does np.unique on the CPU per batch, models.py:341).  backward: grad_h [total_rows,
 * H] is fully overwritten. */
s3grl_status s3grl_centre_pool_forward(s3grl_context* ctx, const float* h, const int64_t* row_ptr,
                                       int64_t num_links, int64_t hidden, int32_t mode, float* out);
s3grl_status s3grl_centre_pool_backward(s3grl_context* ctx, const float* h, const int64_t* row_ptr,
                                        int64_t num_links, int64_t hidden, int32_t mode,
                                        const float* grad_out, float* grad_h);

/* Per-phase device time accumulated since profiling was last switched on, in milliseconds,
 * from HIP events recorded on the context's stream around the kernels themselves:
 * what[0]=structure (count+scan+build+jobs), [1]=propagate, [2]=gather (the dominant kernel),
 * [3]=sop setup, [4]=sop run, [5]=number of gather launches in [2], [6]=number of plans in
 * [0],[1], [7]=number of sop runs in [4], [8]=sop_rows_kernel alone (part of [4]),
 * [9]=the spmm_norm_kernel launches alone (part of [3]), [10]=number of sop setups in [3],
 * [11..15] reserved (0).  Host array of 16 doubles. */
s3grl_status s3grl_context_timings(s3grl_context* ctx, double* what);
/* switch the HIP-event timing on/off and zero the accumulators (off by default) */
s3grl_status s3grl_context_set_profiling(s3grl_context* ctx, int32_t enabled);

/* Gives the workspace blocks the context caches between calls (plans, stashes, scratch: several
 * GB after a PubMed-scale plan) back to the HIP allocator; blocks of live handles stay.  Call it
 * when the precompute phase is over and the training step needs the memory. *released (host,
 * may be NULL) = bytes freed. */
s3grl_status s3grl_context_trim(s3grl_context* ctx, int64_t* released);

/* Measurement: the bytes the gather launch of `p` on operand `f` REQUESTS, computed exactly from
 * the plan (no timing, no sampling): what[0] = node-id bytes (4 per list entry and column tile),
 * [1] = packed-row header bytes (32 per entry and tile; 0 for a dense operand), [2] = feature
 * bytes (packed: 16 per non-zero chunk of every gathered row; dense: the 16-byte lane loads that
 * fall inside the row), [3] = coefficient bytes (8 per entry, operator and tile actually read:
 * the packed kernel reads only the last operator beyond the prefix the others can reach),
 * [4] = output bytes written (folded reversed duplicates included), [5] = operator-0 rows of X
 * read for the output, [6] = per-job metadata bytes, [7] = wavefronts launched.
 * Host array of 8 int64.  Synchronises the context's stream. */
s3grl_status s3grl_plan_gather_traffic(s3grl_context* ctx, const s3grl_plan* p,
                                       const s3grl_features* f, int64_t* what);

/* Measurement aid, not on the product path: reads a KNOWN number of bytes of `buf` (device, `bytes`
 * long) in one of the engine's access shapes, so that rocprofv3's FETCH_SIZE can be calibrated on
 * gfx950 for that shape (tools/pmc_calibrate.py).  pattern 0 / 1 / 2: one coalesced pass with 16 /
 * 8 / 4 bytes per lane; 3: `rows` rows of `row_bytes` (multiple of 16) at pseudo-random places, one
 * wavefront per row, 16 bytes per lane.  *requested_bytes (host) = the bytes the loads asked for. */
s3grl_status s3grl_calibration_read(s3grl_context* ctx, const void* buf, int64_t bytes, int32_t pattern,
                                    int64_t rows, int32_t row_bytes, int64_t* requested_bytes);

#ifdef __cplusplus
}
#endif
#endif /* S3GRL_H_ */
