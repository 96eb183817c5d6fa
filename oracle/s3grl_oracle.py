"""CPU oracle: a restatement of the reference's PoS / PoS Plus / SoP operator precompute.

TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).  numpy + scipy only; every function names
the reference lines whose behaviour it restates.  The structure deliberately follows the
reference's algorithm (per-link Python loop, scipy fancy-index extraction, full sparse powers
of the whole subgraph, row select, sparse x dense product) because `bench.py` also times this
module as the `cpu_baseline` ("port") — it is *not* the engine's row-propagation algorithm.

Declared deviations from the reference (none change a value the consumer reads):
  * node order inside one BFS hop is ascending global id instead of CPython set order
    (reference utils.py:73 appends `list(fringe)`).  Rows 0/1 of every output
    are invariant under that permutation; common-neighbour rows are emitted in ascending
    global id and are compared as multisets.  `order="set"` reproduces the set order.
  * outputs are numpy arrays in a dict instead of torch tensors in a PyG `Data`.
  * `dtype=np.float64` runs the same arithmetic in double precision (the adjudicator for the
    1e-5 tolerance); `dtype=np.float32` mirrors the reference's precision.
"""
from __future__ import annotations

import warnings

import numpy as np
import scipy.sparse as ssp

__all__ = [
    "neighbors",
    "k_hop_subgraph",
    "normalized_subgraph_operator",
    "pos_link",
    "get_PoS_prepped_ds",
    "get_PoS_Plus_prepped_ds",
    "global_normalized_powers",
    "get_SoP_prepped_ds",
    "hybrid_combine",
    "centre_pool",
    "collate_rows",
]


# --------------------------------------------------------------------------------------
# extraction half  (reference utils.py:33-85)
# --------------------------------------------------------------------------------------
def neighbors(fringe, A, outgoing=True):
    """Union of the CSR rows of `fringe` — reference utils.py:33-44; `outgoing=False`: union of the
    CSC columns (`A` is then the CSC form, utils.py:41-42: the nodes with an edge INTO the fringe).

    Reads `.indices`, so *structural* entries count even when their stored value is zero
    (SURVEY §8c K4): after the target link is masked, local node 1 is still a "neighbour"
    of local node 0.
    """
    fringe = list(fringe)
    if not fringe:
        return set()
    if outgoing:
        return set(int(v) for v in A[fringe].indices)
    return set(int(v) for v in A[:, fringe].indices)


_M64 = (1 << 64) - 1


def hop_sample_key(seed, a, b, u):
    """The engine's per-node sampling key (s3grl_device.hpp `hop_sample_key`), in Python ints:
    a 64-bit mix of (seed, unordered endpoints a <= b, node) with the node id in the low half,
    so keys of distinct nodes are distinct."""
    x = (seed & 0xffffffff) << 32
    x ^= ((a & 0xffffffff) * 0x9E3779B97F4A7C15) & _M64
    x ^= ((b & 0xffffffff) * 0xC2B2AE3D27D4EB4F) & _M64
    x = (x + (u & 0xffffffff) * 0x165667B19E3779F9) & _M64
    x ^= x >> 33
    x = (x * 0xff51afd7ed558ccd) & _M64
    x ^= x >> 33
    x = (x * 0xc4ceb9fe1a85ec53) & _M64
    x ^= x >> 33
    return (x & 0xffffffff00000000) | (u & 0xffffffff)


def hash_sampler(seed, src, dst):
    """`sampler` for k_hop_subgraph that makes the engine's draw: the k smallest keys of the hop.
    (The reference draws with `random.sample`; any uniform k-subset is the same distribution.)"""
    a, b = min(int(src), int(dst)), max(int(src), int(dst))

    def pick(fringe, k, dist=None):
        return sorted(fringe, key=lambda u: hop_sample_key(seed, a, b, int(u)))[:k]
    return pick


def k_hop_subgraph(src, dst, num_hops, A, node_features=None, y=1, order="canonical",
                   rw_nodes=None, sample_ratio=1.0, max_nodes_per_hop=None, sampler=None,
                   directed=False, A_csc=None):
    """k-hop enclosing subgraph of link (src, dst) — reference utils.py:47-85, non-rw branch.
    `directed` (utils.py:58-63): a hop follows the out-edges (rows of A) AND the in-edges (columns,
    through `A_csc`) of the fringe; the induced matrix A[nodes][:, nodes] keeps the directions.
    Every paper config runs it with sample_ratio=1.0 / max_nodes_per_hop=None; the
    per-hop sampling of utils.py:66-70 is restated too, with the draw behind `sampler(fringe, k,
    dist)` (default: `random.sample`, what the reference calls).

    BFS runs on the *unmasked* graph from both endpoints at once; the target link is removed
    afterwards from the induced matrix by assignment, which on scipy CSR inserts an explicit
    zero when the entry is absent (utils.py:79-80).

    Returns (nodes, sub_csr, dists, X_S, y) like the reference.
    """
    src, dst = int(src), int(dst)
    if rw_nodes is not None:
        # ScaLed branch with sign=True — reference utils.py:101-150: the node set is the union of
        # the (cached) random-walk nodes of src and dst, made unique and sorted (torch.unique),
        # then dst and src are moved to the front (:134-135); dists = [0, 0, 1, 1, ...] (:145-146);
        # induced matrix and masking as in the BFS branch (:137-143).  num_hops plays no part.
        rest = sorted(set(int(v) for v in rw_nodes) - {src, dst})
        nodes = [src, dst] + rest
        dists = [0, 0] + [1] * len(rest)
        sub = A[nodes, :][:, nodes]
        with warnings.catch_warnings():
            warnings.simplefilter("ignore", ssp.SparseEfficiencyWarning)
            sub[0, 1] = 0
            sub[1, 0] = 0
        if node_features is not None:
            node_features = np.asarray(node_features)[nodes]
        return nodes, sub, dists, node_features, y
    nodes = [src, dst]
    dists = [0, 0]
    visited = {src, dst}
    fringe = {src, dst}
    if sampler is None:
        import random

        def sampler(fr, k, dist):
            return random.sample(sorted(fr), k)
    for dist in range(1, num_hops + 1):
        if not directed:
            fringe = neighbors(fringe, A)
        else:                                              # utils.py:60-63
            fringe = neighbors(fringe, A) | neighbors(fringe, A_csc, False)
        fringe = fringe - visited
        visited |= fringe                                  # dropped nodes stay visited (utils.py:65)
        if sample_ratio < 1.0:                             # utils.py:66-67
            fringe = set(sampler(fringe, int(sample_ratio * len(fringe)), dist))
        if max_nodes_per_hop is not None:                  # utils.py:68-70
            if max_nodes_per_hop < len(fringe):
                fringe = set(sampler(fringe, max_nodes_per_hop, dist))
        if not fringe:
            break
        hop = sorted(fringe) if order == "canonical" else list(fringe)
        nodes += hop
        dists += [dist] * len(hop)
    sub = A[nodes, :][:, nodes]
    with warnings.catch_warnings():
        warnings.simplefilter("ignore", ssp.SparseEfficiencyWarning)
        sub[0, 1] = 0
        sub[1, 0] = 0
    if node_features is not None:
        node_features = np.asarray(node_features)[nodes]
    return nodes, sub, dists, node_features, y


# --------------------------------------------------------------------------------------
# diffusion half  (reference tuned_SIGN.py:151-185; torch_sparse semantics restated)
# --------------------------------------------------------------------------------------
def normalized_subgraph_operator(sub_csr, dtype=np.float32):
    """D^-1/2 A D^-1/2 of the masked subgraph — reference tuned_SIGN.py:153-161.

    `ssp.find` drops explicit zeros (so the masked target link disappears) and its values are
    discarded: the operator is built from structure only.  deg = stored entries per row
    (torch_sparse `SparseTensor.sum(dim=1)` of a value-less tensor is the row count);
    deg^-1/2 with inf -> 0 for isolated nodes; no self-loops are added.
    """
    n = sub_csr.shape[0]
    u, v, _ = ssp.find(sub_csr)
    deg = np.bincount(u, minlength=n).astype(dtype)
    with np.errstate(divide="ignore"):
        dinv = np.power(deg, dtype(-0.5))
    dinv[np.isinf(dinv)] = 0
    vals = (dinv[u] * dinv[v]).astype(dtype)
    return ssp.csr_matrix((vals, (u, v)), shape=(n, n), dtype=dtype)


def _powers(op, K):
    """[Â, Â·Â, …] of the WHOLE subgraph — reference tuned_SIGN.py:168-170."""
    out = [op]
    for _ in range(K - 1):
        out.append(op @ out[-1])
    return out


def pos_link(src, dst, num_hops, A, x, K, *, plus=False, strategy="intersection",
             dtype=np.float32, order="canonical", rw_nodes=None, sample_ratio=1.0,
             max_nodes_per_hop=None, sampler=None, directed=False, A_csc=None):
    """One iteration of the reference's PoS / PoS Plus hot loop — tuned_SIGN.py:147-187 and
    :202-260.  Returns a dict with x, x1..xK ([R, 1+F]), the selected local rows, the global
    ids of those rows, the node list and hop distances."""
    nodes, sub, dists, X_S, _ = k_hop_subgraph(src, dst, num_hops, A, node_features=x,
                                                order=order, rw_nodes=rw_nodes,
                                                sample_ratio=sample_ratio,
                                                max_nodes_per_hop=max_nodes_per_hop, sampler=sampler,
                                                directed=directed, A_csc=A_csc)
    n = sub.shape[0]
    op = normalized_subgraph_operator(sub, dtype)
    powers = _powers(op, K)

    rows = [0, 1]
    if plus:
        if strategy == "intersection":
            cn = neighbors({0}, sub) & neighbors({1}, sub)      # tuned_SIGN.py:233
        elif strategy == "union":
            # reference tuned_SIGN.py:243 builds a ragged label column for `union`, which
            # torch.tensor rejects unless n == 3: the branch is unusable as shipped.
            raise NotImplementedError("k_node_set_strategy='union' is broken in the reference")
        else:
            raise NotImplementedError(f"check strat {strategy}")
        # reference order is set-iteration order; canonical = ascending global id
        rows = rows + sorted(cn, key=lambda a: nodes[a])

    z = np.zeros((n, 1), dtype=dtype)
    z[0, 0] = 1
    z[1, 0] = 1
    subg_x = np.hstack([z, np.asarray(X_S, dtype=dtype)])       # tuned_SIGN.py:177-179

    out = {"x": subg_x[rows].copy()}
    for i, P in enumerate(powers, start=1):
        out[f"x{i}"] = np.asarray(P[rows] @ subg_x)             # tuned_SIGN.py:175,185
    out["rows_local"] = np.asarray(rows, dtype=np.int64)
    out["rows_global"] = np.asarray([nodes[a] for a in rows], dtype=np.int64)
    out["nodes"] = np.asarray(nodes, dtype=np.int64)
    out["dists"] = np.asarray(dists, dtype=np.int64)
    return out


def _links(link_index):
    li = np.asarray(link_index)
    assert li.ndim == 2 and li.shape[0] == 2, "link_index must be [2, L]"
    return li.T.tolist()


def _sampling_of(src, dst, ratio_per_hop, max_nodes_per_hop, sample_seed):
    """kwargs for pos_link: the reference's (ratio_per_hop, max_nodes_per_hop) with the draw made
    by the engine's keyed generator when `sample_seed` is given, by `random.sample` otherwise."""
    return {"sample_ratio": 1.0 if ratio_per_hop is None else ratio_per_hop,
            "max_nodes_per_hop": max_nodes_per_hop,
            "sampler": None if sample_seed is None else hash_sampler(sample_seed, src, dst)}


def get_PoS_prepped_ds(link_index, num_hops, A, x, y, sign_kwargs, *, dtype=np.float32,
                       order="canonical", rw_node_sets=None, ratio_per_hop=1.0,
                       max_nodes_per_hop=None, sample_seed=None, directed=False, A_csc=None):
    """Reference tuned_SIGN.py:137-189 (optimised PoS flow), one dict per link.
    `rw_node_sets[l]` = the random-walk node set of link l (ScaLed branch), else k-hop BFS.
    `directed` / `A_csc` as the reference passes them through to k_hop_subgraph (:149-150); the
    operator is D^-1/2 A D^-1/2 of the DIRECTED induced matrix with D = out-degrees (row counts,
    :158-161), and rows [0, 1] of its powers are taken as they are."""
    assert x is not None
    K = sign_kwargs["sign_k"]
    out = []
    for l, (src, dst) in enumerate(_links(link_index)):
        d = pos_link(src, dst, num_hops, A, x, K, plus=False, dtype=dtype, order=order,
                     rw_nodes=None if rw_node_sets is None else rw_node_sets[l],
                     directed=directed, A_csc=A_csc,
                     **_sampling_of(src, dst, ratio_per_hop, max_nodes_per_hop, sample_seed))
        d["y"] = y
        out.append(d)
    return out


def get_PoS_Plus_prepped_ds(link_index, num_hops, A, x, y, sign_kwargs, *, dtype=np.float32,
                            order="canonical", rw_node_sets=None, ratio_per_hop=1.0,
                            max_nodes_per_hop=None, sample_seed=None, directed=False, A_csc=None):
    """Reference tuned_SIGN.py:192-262 (optimised PoS Plus flow), one dict per link."""
    assert x is not None
    K = sign_kwargs["sign_k"]
    strat = sign_kwargs["k_node_set_strategy"]
    out = []
    for l, (src, dst) in enumerate(_links(link_index)):
        d = pos_link(src, dst, num_hops, A, x, K, plus=True, strategy=strat, dtype=dtype,
                     order=order, rw_nodes=None if rw_node_sets is None else rw_node_sets[l],
                     directed=directed, A_csc=A_csc,
                     **_sampling_of(src, dst, ratio_per_hop, max_nodes_per_hop, sample_seed))
        d["y"] = y
        out.append(d)
    return out


# --------------------------------------------------------------------------------------
# SoP  (reference sgrl_link_pred.py:161-178 and tuned_SIGN.py:49-134)
# --------------------------------------------------------------------------------------
def global_normalized_powers(A, K, dtype=np.float32, edge_index=None):
    """[Â, Â², …, Â^K] of the WHOLE train graph — reference sgrl_link_pred.py:161-178.
    Binary structure, deg = row count, inf -> 0, no self-loops added, target links NOT
    removed.  `edge_index` ([2, E], optional): the caller's UNCOALESCED edge list, which is what
    the reference builds the operator from (`SparseTensor(row, col)`, :164-167) — an entry listed m
    times counts m times in the degree and weighs m in every product; without it the structure of
    A is taken (every entry once), which is the same thing for a coalesced graph."""
    A = ssp.csr_matrix(A)
    N = A.shape[0]
    if edge_index is None:
        coo = A.tocoo()
        u, v = coo.row, coo.col
    else:
        u, v = np.asarray(edge_index[0], dtype=np.int64), np.asarray(edge_index[1], dtype=np.int64)
    deg = np.bincount(u, minlength=N).astype(dtype)
    with np.errstate(divide="ignore"):
        dinv = np.power(deg, dtype(-0.5))
    dinv[np.isinf(dinv)] = 0
    # (duplicate (u, v) pairs are summed by csr_matrix: the additive reading of an uncoalesced tensor)
    op = ssp.csr_matrix(((dinv[u] * dinv[v]).astype(dtype), (u, v)), shape=(N, N), dtype=dtype)
    out = [op]
    for _ in range(2, K + 1):
        out.append(op @ out[-1])
    return out


def get_SoP_prepped_ds(powers_of_A, link_index, A, x, y, *, dtype=np.float32):
    """Reference tuned_SIGN.py:49-134 (optimised SoP flow).

    Per operator i and link (s, d): row s of Â^i with column d zeroed and row d with column s
    zeroed (:71-78), both times X (:92-100); the diagonal entries Â^i[s,s], Â^i[d,d] are
    prepended as the label column (:102-113); x = [[1|X[s]],[1|X[d]]] (:119-125).
    `num_hops` plays no part.  Vectorised per operator (the reference loops in Python over
    LIL/DOK rows); the arithmetic per output element is the same masked row-times-X sum.
    """
    links = np.asarray(_links(link_index), dtype=np.int64).reshape(-1, 2)
    L = links.shape[0]
    X = np.asarray(x, dtype=dtype)
    F = X.shape[1]
    src, dst = links[:, 0], links[:, 1]
    xi = []
    for P in powers_of_A:
        P = ssp.csr_matrix(P).astype(dtype)
        diag = np.asarray(P.diagonal(), dtype=dtype)
        stacked_idx = np.empty(2 * L, dtype=np.int64)
        stacked_idx[0::2] = src
        stacked_idx[1::2] = dst
        masked_col = np.empty(2 * L, dtype=np.int64)
        masked_col[0::2] = dst
        masked_col[1::2] = src
        rows = P[stacked_idx].tocsr()
        rows.sort_indices()
        # zero the masked column of every stacked row
        rid = np.repeat(np.arange(2 * L), np.diff(rows.indptr))
        kill = rows.indices == masked_col[rid]
        rows.data[kill] = 0
        G = np.asarray(rows @ X, dtype=dtype)
        h = np.empty((2 * L, 1), dtype=dtype)
        h[0::2, 0] = diag[src]
        h[1::2, 0] = diag[dst]
        xi.append(np.hstack([h, G]).reshape(L, 2, 1 + F))
    out = []
    ones = np.ones((2, 1), dtype=dtype)
    for l in range(L):
        d = {"x": np.hstack([ones, X[[src[l], dst[l]]]]), "y": y}
        for i, arr in enumerate(xi, start=1):
            d[f"x{i}"] = arr[l]
        d["rows_global"] = np.array([src[l], dst[l]], dtype=np.int64)
        out.append(d)
    return out


def get_SoP_restricted_ds(powers_of_A, link_index, num_hops, A, x, y, *, dtype=np.float32):
    """NOT a reference flow — the optional twin SURVEY §8(d) names for BASELINE config 3 ("2-hop subgraphs"): the
    SoP rows of `get_SoP_prepped_ds` (tuned_SIGN.py:49-134) with every operator row additionally restricted to the
    `num_hops`-ball of {src, dst} on the unmasked graph (the node set `k_hop_subgraph` extracts, utils.py:53-74):
        x_i[s] = [ Â^i[s,s] | Σ_{w in ball, w != d} Â^i[s,w] X[w] ].
    Plain per-link loop over the global powers."""
    links = np.asarray(_links(link_index), dtype=np.int64).reshape(-1, 2)
    A = ssp.csr_matrix(A)
    X = np.asarray(x, dtype=dtype)
    P = [ssp.csr_matrix(p).astype(dtype) for p in powers_of_A]
    out = []
    ones = np.ones((2, 1), dtype=dtype)
    for s_, d_ in links:
        ball = {int(s_), int(d_)}
        fringe = set(ball)
        for _ in range(int(num_hops)):
            fringe = neighbors(fringe, A) - ball
            if not fringe:
                break
            ball |= fringe
        keep = np.zeros(A.shape[0], dtype=bool)
        keep[list(ball)] = True
        d = {"x": np.hstack([ones, X[[s_, d_]]]), "y": y, "rows_global": np.array([s_, d_], dtype=np.int64)}
        for i, Pi in enumerate(P, start=1):
            rows = []
            for a, b in ((s_, d_), (d_, s_)):
                r = Pi[a]
                w = np.asarray(r.todense(), dtype=dtype).ravel() * keep
                w[b] = 0
                rows.append(np.concatenate([[Pi[a, a]], w @ X]))
            d[f"x{i}"] = np.asarray(rows, dtype=dtype)
        out.append(d)
    return out


def hybrid_combine(pos_list, sop_list, sign_k):
    """Reference utils.py:472-480: PoS keys kept, SoP x2..xK appended as x{K+1}..x{2K-1}."""
    out = []
    for p, s in zip(pos_list, sop_list):
        d = dict(p)
        for k in range(sign_k + 1, sign_k * 2):
            d[f"x{k}"] = s[f"x{k - sign_k + 1}"]
        out.append(d)
    return out


# --------------------------------------------------------------------------------------
# consumer-side contract  (reference sgrl_link_pred.py:204,449-459; models.py:339-372)
# --------------------------------------------------------------------------------------
def collate_rows(data_list, K):
    """What PyG collate + `torch.cat(xs, dim=-1)` hand to the MLP: rows [ΣR, (K+1), 1+F],
    row_ptr [L+1], y [L] — reference sgrl_link_pred.py:204, :449-459, models.py:372."""
    keys = ["x"] + [f"x{i}" for i in range(1, K + 1)]
    counts = [d["x"].shape[0] for d in data_list]
    row_ptr = np.zeros(len(data_list) + 1, dtype=np.int64)
    np.cumsum(counts, out=row_ptr[1:])
    if not data_list:
        return np.zeros((0, K + 1, 0)), row_ptr, np.zeros(0, dtype=np.int64)
    rows = np.concatenate(
        [np.stack([d[k] for k in keys], axis=1) for d in data_list], axis=0)
    y = np.asarray([d["y"] for d in data_list], dtype=np.int64)
    return rows, row_ptr, y


def centre_pool(h, row_ptr, k_heuristic=0, k_pool_strategy="mean"):
    """Reference models.py:339-369 on the collated layout: h [ΣR, H]; the two centre rows of
    every link come first.  Links without extra rows pool to zeros (`size=B`, :357-362)."""
    h = np.asarray(h)
    row_ptr = np.asarray(row_ptr)
    c = row_ptr[:-1]
    h_a = h[c] * h[c + 1]
    if not k_heuristic:
        return h_a
    B, H = len(c), h.shape[1]
    if k_pool_strategy in ("mean", "sum"):
        pooled = np.zeros((B, H), dtype=h.dtype)
        for b in range(B):
            extra = h[c[b] + 2: row_ptr[b + 1]]
            if len(extra):
                pooled[b] = extra.sum(0) if k_pool_strategy == "sum" else extra.mean(0)
        return np.concatenate([h_a, pooled], axis=-1)
    if k_pool_strategy == "concat":
        extra = np.concatenate([h[c[b] + 2: row_ptr[b + 1]] for b in range(B)], axis=0)
        return np.concatenate([h_a, extra.reshape(B, H * k_heuristic)], axis=-1)
    raise NotImplementedError(f"Check pool strat: {k_pool_strategy}")
