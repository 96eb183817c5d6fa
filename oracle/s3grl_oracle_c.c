/*
 * TEST INFRASTRUCTURE ONLY — second, independent CPU restatement of the S3GRL PoS / PoS Plus
 * operator precompute in plain C (OpenMP over links).  Nothing under s3grl_amd/ may link, load or
 * call this file; it exists to (1) cross-check the numpy/scipy oracle (oracle/s3grl_oracle.py)
 * with a different formulation, (2) check far more links of the full-size workloads than the
 * Python oracle can finish, and (3) give bench.py an all-cores CPU baseline.
 *
 * Pinning: the extraction half agrees bit-for-bit with tests/golden/extract_*.npz, which were
 * produced by the reference's own utils.k_hop_subgraph / utils.neighbors; the diffusion half is
 * checked against tests/golden/diffusion_*.npz (fp64 output of the Python restatement): like
 * that restatement it is "parity unpinned" with respect to torch_sparse 0.6.13 itself.
 *
 * What it follows (reference file:line):
 *   utils.py:33-44    neighbors(): union of the CSR rows of the frontier
 *   utils.py:47-85    k_hop_subgraph(): BFS from {src,dst} to depth h on the UNMASKED graph,
 *                     nodes = [src,dst] + hop 1 + hop 2 ..., early stop on an empty frontier,
 *                     induced A[nodes][:,nodes], target link (0,1),(1,0) zeroed
 *   tuned_SIGN.py:153-161  binary structure (values dropped, explicit zeros of the masked link
 *                     dropped by ssp.find), deg = row count, D^-1/2 A D^-1/2, inf -> 0
 *   tuned_SIGN.py:168-185  rows {0,1} of A^1..A^K times [z | X_S], z = (1,1,0,...)
 *   tuned_SIGN.py:229-258  PoS Plus: rows [0,1] + common neighbours of 0 and 1 in the masked
 *                     subgraph ('intersection')
 * The reference materialises A^i by SpGEMM and selects rows; here row t of A^i is obtained as
 * e_t A^i by i sparse vector-matrix products (same numbers, fp64).  Node order inside a hop is
 * ascending global id (the reference's order is Python set iteration order; the order
 * does not change any output row).
 *
 * Not handled here (the Python oracle covers them): self-loops, hops = 0 is handled, directed
 * graphs, ScaLed walks.  The graph must be structurally symmetric with sorted, duplicate-free rows.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

/* Arithmetic type of the diffusion half.  double (default) = the adjudicator of the 1e-5 bar;
 * `make f32` builds the same source with float: the reference's own precision (torch_sparse runs
 * the normalisation, the powers and the product in fp32), whose distance from the fp64 result is
 * the noise floor reported next to the engine's error.  Outputs are double either way. */
#ifndef S3GRL_ORACLE_REAL
#define S3GRL_ORACLE_REAL double
#define S3GRL_ORACLE_SQRT sqrt
#endif
typedef S3GRL_ORACLE_REAL real_t;

typedef struct {
    int32_t *local;      /* [N] local id or -1 */
    int32_t *nodes;      /* [N] hop-major list */
    int32_t *sub_ptr;    /* [N+1] */
    int32_t *sub_idx;    /* grows */
    int64_t sub_cap;
    real_t *dinv;        /* [N] */
    real_t *r0, *r1;     /* [N] */
    real_t *acc;         /* [1 + F] one output row being summed */
    int32_t *cn;         /* [N] local ids of the CN rows */
} scratch_t;

static int cmp_i32(const void *a, const void *b) {
    int32_t x = *(const int32_t *)a, y = *(const int32_t *)b;
    return (x > y) - (x < y);
}

static int scratch_init(scratch_t *s, int64_t N) {
    s->local = (int32_t *)malloc(sizeof(int32_t) * (size_t)N);
    s->nodes = (int32_t *)malloc(sizeof(int32_t) * (size_t)N);
    s->sub_ptr = (int32_t *)malloc(sizeof(int32_t) * (size_t)(N + 1));
    s->sub_cap = 1 << 16;
    s->sub_idx = (int32_t *)malloc(sizeof(int32_t) * (size_t)s->sub_cap);
    s->dinv = (real_t *)malloc(sizeof(real_t) * (size_t)N);
    s->r0 = (real_t *)malloc(sizeof(real_t) * (size_t)N);
    s->r1 = (real_t *)malloc(sizeof(real_t) * (size_t)N);
    s->acc = NULL;
    s->cn = (int32_t *)malloc(sizeof(int32_t) * (size_t)N);
    if (!s->local || !s->nodes || !s->sub_ptr || !s->sub_idx || !s->dinv || !s->r0 || !s->r1 || !s->cn)
        return -1;
    for (int64_t i = 0; i < N; ++i) s->local[i] = -1;
    return 0;
}

static void scratch_free(scratch_t *s) {
    free(s->local); free(s->nodes); free(s->sub_ptr); free(s->sub_idx);
    free(s->dinv); free(s->r0); free(s->r1); free(s->cn); free(s->acc);
}

/* utils.py:47-85: hop-major node list of the h-hop subgraph around (src, dst); returns n.
 * `hop_end[j]` = number of nodes with distance <= j (hop_end[0] = 2). */
static int32_t extract(scratch_t *s, const int64_t *indptr, const int32_t *indices, int32_t src,
                       int32_t dst, int num_hops, int32_t *hop_end) {
    int32_t n = 0;
    s->nodes[n] = src; s->local[src] = n++;
    s->nodes[n] = dst; s->local[dst] = n++;
    int32_t lo = 0;
    if (hop_end) hop_end[0] = 2;
    for (int hop = 1; hop <= num_hops; ++hop) {
        int32_t hi = n;
        for (int32_t a = lo; a < hi; ++a) {
            int32_t v = s->nodes[a];
            for (int64_t e = indptr[v]; e < indptr[v + 1]; ++e) {
                int32_t u = indices[e];
                if (s->local[u] == -1) { s->local[u] = -2; s->nodes[n++] = u; }
            }
        }
        if (n == hi) {                       /* utils.py:71-72: empty fringe ends the walk */
            if (hop_end) for (int j = hop; j <= num_hops; ++j) hop_end[j] = n;
            break;
        }
        qsort(s->nodes + hi, (size_t)(n - hi), sizeof(int32_t), cmp_i32);
        for (int32_t a = hi; a < n; ++a) s->local[s->nodes[a]] = a;
        if (hop_end) hop_end[hop] = n;
        lo = hi;
    }
    return n;
}

static void release(scratch_t *s, int32_t n) {
    for (int32_t a = 0; a < n; ++a) s->local[s->nodes[a]] = -1;
}

/* induced, masked, binary sub-CSR in local ids + D^-1/2 (tuned_SIGN.py:153-161) */
static int induce(scratch_t *s, const int64_t *indptr, const int32_t *indices, int32_t n) {
    int64_t m = 0;
    s->sub_ptr[0] = 0;
    for (int32_t a = 0; a < n; ++a) {
        int32_t v = s->nodes[a];
        int64_t need = m + (indptr[v + 1] - indptr[v]);
        if (need > s->sub_cap) {
            while (s->sub_cap < need) s->sub_cap *= 2;
            int32_t *p = (int32_t *)realloc(s->sub_idx, sizeof(int32_t) * (size_t)s->sub_cap);
            if (!p) return -1;
            s->sub_idx = p;
        }
        for (int64_t e = indptr[v]; e < indptr[v + 1]; ++e) {
            int32_t b = s->local[indices[e]];
            if (b < 0) continue;
            if ((a == 0 && b == 1) || (a == 1 && b == 0)) continue;   /* utils.py:79-80 + find() */
            s->sub_idx[m++] = b;
        }
        int64_t d = m - s->sub_ptr[a];
        s->dinv[a] = d > 0 ? (real_t)1 / S3GRL_ORACLE_SQRT((real_t)d) : (real_t)0;   /* inf -> 0 */
        s->sub_ptr[a + 1] = (int32_t)m;
    }
    return 0;
}

/* rows-per-link pass: R[l] = 2 (PoS) or 2 + |N(src) ∩ N(dst)| (PoS Plus, hops >= 1) */
int s3grl_oracle_c_rows_per_link(int64_t N, const int64_t *indptr, const int32_t *indices,
                                 const int64_t *links, int64_t L, int num_hops, int plus, int64_t *R) {
    (void)N;
    for (int64_t l = 0; l < L; ++l) {
        int64_t s = links[2 * l], d = links[2 * l + 1];
        if (s == d) return -2;
        int64_t c = 0;
        if (plus && num_hops >= 1) {
            int64_t i = indptr[s], j = indptr[d];
            while (i < indptr[s + 1] && j < indptr[d + 1]) {
                if (indices[i] < indices[j]) ++i;
                else if (indices[i] > indices[j]) ++j;
                else { ++c; ++i; ++j; }
            }
        }
        R[l] = 2 + c;
    }
    return 0;
}

/*
 * rows: [sum R, K+1, 1+F] fp64, row_ptr: [L+1] (exclusive scan of R), row_nodes: [sum R] global
 * ids of the emitted rows, node_count: [L] subgraph sizes (may be NULL).
 * X is fp32 [N, F] with leading dimension ldx.  Returns 0, or <0 on error.
 */
int s3grl_oracle_c_pos(int64_t N, const int64_t *indptr, const int32_t *indices, const float *X,
                       int64_t ldx, int64_t F, const int64_t *links, int64_t L, int num_hops, int sign_k,
                       int plus, int threads, const int64_t *row_ptr, double *rows,
                       int64_t *row_nodes, int32_t *node_count) {
    int err = 0;
    const int64_t W = 1 + F;
    const int K = sign_k;
#ifdef _OPENMP
    if (threads > 0) omp_set_num_threads(threads);
#else
    (void)threads;
#endif
#pragma omp parallel
    {
        scratch_t s;
        int bad = scratch_init(&s, N);
        if (!bad) {
            s.acc = (real_t *)malloc(sizeof(real_t) * (size_t)W);
            bad = s.acc == NULL;
        }
        if (bad) {
#pragma omp atomic write
            err = -1;
        }
#pragma omp for schedule(dynamic, 8)
        for (int64_t l = 0; l < L; ++l) {
            if (bad || err) continue;
            int32_t src = (int32_t)links[2 * l], dst = (int32_t)links[2 * l + 1];
            int32_t n = extract(&s, indptr, indices, src, dst, num_hops, NULL);
            if (induce(&s, indptr, indices, n)) {
#pragma omp atomic write
                err = -1;
                release(&s, n);
                continue;
            }
            if (node_count) node_count[l] = n;
            /* selected rows: [0, 1] + common neighbours of 0 and 1 in the masked subgraph */
            int32_t nr = 0;
            s.cn[nr++] = 0;
            s.cn[nr++] = 1;
            if (plus) {
                int32_t i = s.sub_ptr[0], j = s.sub_ptr[1];
                /* local ids inside one sub-CSR row are not sorted (hop-major relabelling):
                 * mark row 0's neighbours in r1 as a flag array instead of merging */
                for (int32_t a = 0; a < n; ++a) s.r1[a] = 0;
                for (; i < s.sub_ptr[1]; ++i) s.r1[s.sub_idx[i]] = 1;
                int32_t first = nr;
                for (; j < s.sub_ptr[2]; ++j)
                    if (s.r1[s.sub_idx[j]] != 0) s.cn[nr++] = s.sub_idx[j];
                /* emit CN rows in ascending global id */
                for (int32_t a = first + 1; a < nr; ++a) {
                    int32_t key = s.cn[a], b = a - 1;
                    while (b >= first && s.nodes[s.cn[b]] > s.nodes[key]) { s.cn[b + 1] = s.cn[b]; --b; }
                    s.cn[b + 1] = key;
                }
            }
            if ((int64_t)nr != row_ptr[l + 1] - row_ptr[l]) {
#pragma omp atomic write
                err = -3;
                release(&s, n);
                continue;
            }
            for (int32_t t = 0; t < nr; ++t) {
                int64_t orow = row_ptr[l] + t;
                double *out = rows + orow * (K + 1) * W;
                int32_t a0 = s.cn[t];
                if (row_nodes) row_nodes[orow] = s.nodes[a0];
                /* operator 0: [z | X[node]] (tuned_SIGN.py:177-181) */
                out[0] = a0 < 2 ? 1.0 : 0.0;
                const float *xr = X + (int64_t)s.nodes[a0] * ldx;
                for (int64_t f = 0; f < F; ++f) out[1 + f] = (double)xr[f];
                real_t *r = s.r0, *rn = s.r1;
                for (int32_t a = 0; a < n; ++a) r[a] = 0;
                r[a0] = 1;
                for (int i = 1; i <= K; ++i) {
                    /* rn = r A_hat ; symmetric structure: pull over b's own neighbour list */
                    for (int32_t b = 0; b < n; ++b) {
                        real_t acc = 0;
                        for (int32_t e = s.sub_ptr[b]; e < s.sub_ptr[b + 1]; ++e) {
                            int32_t a = s.sub_idx[e];
                            acc += r[a] * s.dinv[a];
                        }
                        rn[b] = acc * s.dinv[b];
                    }
                    real_t *o = s.acc;
                    for (int64_t f = 0; f < W; ++f) o[f] = 0;
                    for (int32_t b = 0; b < n; ++b) {
                        real_t c = rn[b];
                        if (c == 0) continue;
                        if (b < 2) o[0] += c;
                        const float *xb = X + (int64_t)s.nodes[b] * ldx;
                        for (int64_t f = 0; f < F; ++f) o[1 + f] += c * (real_t)xb[f];
                    }
                    double *od = out + (int64_t)i * W;
                    for (int64_t f = 0; f < W; ++f) od[f] = (double)o[f];
                    real_t *tmp = r; r = rn; rn = tmp;
                }
            }
            release(&s, n);
        }
        scratch_free(&s);
    }
    return err;
}

/* extraction only: nodes/dists of every link, for the bit-exact checks.
 * node_ptr: [L+1] out (exclusive scan), nodes/dists: capacity `cap` entries; returns total or <0. */
int64_t s3grl_oracle_c_extract(int64_t N, const int64_t *indptr, const int32_t *indices,
                               const int64_t *links, int64_t L, int num_hops, int64_t cap,
                               int64_t *node_ptr, int32_t *nodes, int8_t *dists) {
    scratch_t s;
    if (scratch_init(&s, N)) return -1;
    int32_t hop_end[64];
    if (num_hops > 62) { scratch_free(&s); return -4; }
    int64_t tot = 0;
    node_ptr[0] = 0;
    for (int64_t l = 0; l < L; ++l) {
        int32_t n = extract(&s, indptr, indices, (int32_t)links[2 * l], (int32_t)links[2 * l + 1],
                            num_hops, hop_end);
        if (tot + n > cap) { release(&s, n); scratch_free(&s); return -5; }
        int hop = 0;
        for (int32_t a = 0; a < n; ++a) {
            while (a >= hop_end[hop]) ++hop;
            nodes[tot + a] = s.nodes[a];
            dists[tot + a] = (int8_t)hop;
        }
        release(&s, n);
        tot += n;
        node_ptr[l + 1] = tot;
    }
    scratch_free(&s);
    return tot;
}
