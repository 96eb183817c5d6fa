"""Callers of the operators, mirrored: the flow dispatch `extract_enclosing_subgraphs`
(reference utils.py:446-554) and the per-split orchestration of `SEALDataset.process`
(reference sgrl_link_pred.py:96-220) on top of the MI355X engine and the bundle cache.

The reference's own `utils.py` / `sgrl_link_pred.py` can keep calling the drop-in operators
(`s3grl_amd.tuned_SIGN`) unchanged; these twins exist for callers that do not carry PyG around —
the harness, the tests, the benchmark — and hold the same decision table (`_FLOW_TABLE`):

    sign_kwargs, powers_of_A, optimize_sign, sign_type == 'hybrid'      -> PoS + SoP, SoP x2..xK
                                                                          appended as x{K+1}..x{2K-1}
    sign_kwargs, powers_of_A, optimize_sign                             -> SoP
    sign_kwargs, no powers_of_A, optimize_sign, not k_heuristic         -> PoS
    sign_kwargs, no powers_of_A, optimize_sign, k_heuristic             -> PoS Plus
    sign_kwargs, not optimize_sign                                      -> per-link SIGN + SEAL flow
    otherwise                                                           -> NotImplementedError

The per-link SIGN + SEAL flow (utils.py:497-550: k_hop_subgraph + construct_pyg_graph + TunedSIGN
per link) and the plain SEAL flow without sign_kwargs (utils.py:556-573) produce labelled PyG
graphs for the MPNN baselines; they are outside the engine and raise.
"""
from __future__ import annotations

import numpy as np
import scipy.sparse as ssp
import torch

from . import cache as _cache
from .tuned_SIGN import LinkDataList, OptimizedSignOperations


class OperatorStandIn:
    """One element of `GlobalOperators`: no values, only the number of entries the reference's
    `SparseTensor(row, col)` would hold — one per column of the edge_index, duplicates included."""

    def __init__(self, num_entries):
        self._n = int(num_entries)

    def nnz(self):
        return self._n


class GlobalOperators(list):
    """Stand-in for the reference's `powers_of_A` (sgrl_link_pred.py:161-178: K torch_sparse
    SparseTensors Â, Â², …): the engine rebuilds Â from A's structure and never materialises a
    power, so the list carries its length (= sign_k), truthiness and — with `num_entries`, the
    number of columns of the edge_index Â is built from — what the SoP operator needs to tell a
    pair listed m times (m entries) from a pair of weight m (one entry), which A's summed data
    cannot (`tuned_SIGN._multiplicity_of`).  Without it: every stored pair counts once."""

    def __init__(self, sign_k, num_entries=None):
        super().__init__([None if num_entries is None else OperatorStandIn(num_entries)] * int(sign_k))


def _hybrid_combine(pos_part, sop_part, sign_k):
    """Hybrid elements = the PoS element plus the SoP operators 2..K under the keys x{K+1}..x{2K-1}
    (reference utils.py:472-480).  Two lazy lists: one `cat` of their collated tensors."""
    if isinstance(pos_part, LinkDataList) and isinstance(sop_part, LinkDataList) \
            and len(pos_part._chunks) == 1 and len(sop_part._chunks) == 1:
        rows_p, ptr, y = pos_part._chunks[0]
        rows_s = sop_part._chunks[0][0]
        return LinkDataList([(torch.cat([rows_p, rows_s[:, 2:, :]], dim=1), ptr, y)], 2 * sign_k - 1)
    merged = []
    for elem, extra in zip(pos_part, sop_part):
        for j in range(2, sign_k + 1):
            elem[f'x{sign_k + j - 1}'] = extra[f'x{j}']
        merged.append(elem)
    return merged


# The decision table of reference utils.py:454-496 as data.  A row matches when the caller did /
# did not hand in global operators, and — where the column is not None — on sign_type and on the
# truthiness of k_heuristic; first match wins.
#            global operators   sign_type   k_heuristic   flow
_FLOW_TABLE = ((True,            "hybrid",   None,         "hybrid"),
               (True,            None,       None,         "sop"),
               (False,           None,       False,        "pos"),
               (False,           None,       True,         "pos_plus"))


def select_flow(sign_kwargs, powers_of_A):
    """Name of the optimised flow the reference's dispatch would take, or NotImplementedError for the
    flows that are not part of the engine (per-link SIGN + SEAL graphs, plain SEAL)."""
    if not sign_kwargs:
        raise NotImplementedError("the SEAL flow without sign_kwargs (labelled subgraphs for the MPNN "
                                  "baselines, reference utils.py:556-573) is not part of the MI355X engine")
    if not sign_kwargs['optimize_sign']:
        raise NotImplementedError("optimize_sign=False (per-link SIGN + SEAL graphs, reference "
                                  "utils.py:497-550) is not part of the MI355X engine")
    have_ops = bool(powers_of_A)
    for need_ops, sign_type, heuristic, flow in _FLOW_TABLE:
        if need_ops == have_ops and sign_type in (None, sign_kwargs['sign_type']) \
                and heuristic in (None, bool(sign_kwargs['k_heuristic'])):
            return flow
    raise NotImplementedError("No matching configuration for model data prep found. Please check code.")


def extract_enclosing_subgraphs(link_index, A, x, y, num_hops, node_label='drnl',
                                ratio_per_hop=1.0, max_nodes_per_hop=None,
                                directed=False, A_csc=None, rw_kwargs=None, sign_kwargs=None, powers_of_A=None,
                                data=None):
    """Reference utils.py:446-554, same positional signature.  Returns the per-link list (a lazy
    `LinkDataList`, see s3grl_amd.tuned_SIGN)."""
    flow = select_flow(sign_kwargs, powers_of_A)
    ops = OptimizedSignOperations

    def subgraph_flow(operator):          # the PoS-shaped operators share one argument list
        return operator(link_index, num_hops, A, ratio_per_hop, max_nodes_per_hop, directed, A_csc, x, y,
                        sign_kwargs, rw_kwargs)

    if flow == "pos":
        return subgraph_flow(ops.get_PoS_prepped_ds)
    if flow == "pos_plus":
        return subgraph_flow(ops.get_PoS_Plus_prepped_ds)
    if flow == "sop":
        return ops.get_SoP_prepped_ds(powers_of_A, link_index, A, x, y)
    # hybrid: plain PoS (never PoS Plus) + SoP; a single operator has nothing to append
    pos_part = subgraph_flow(ops.get_PoS_prepped_ds)
    if sign_kwargs['sign_k'] == 1:
        return pos_part
    return _hybrid_combine(pos_part, ops.get_SoP_prepped_ds(powers_of_A, link_index, A, x, y),
                           sign_kwargs['sign_k'])


def create_rw_cache(A, edges, device, rw_m, rw_M, seed=0):
    """`utils.create_rw_cache` for `process_split` (reference sgrl_link_pred.py:123-128) on the engine:
    walks on the device copy of A the operators will use anyway (uploaded once, tuned_SIGN's cache)."""
    from . import scaled
    from .tuned_SIGN import _device_graph

    eng, g_dev = _device_graph(A)
    return scaled.create_rw_cache(g_dev, edges, device, rw_m, rw_M, seed=seed, engine=eng)


def coalesce(edge_index, edge_weight, num_nodes):
    """torch_sparse `coalesce(index, value, m, n)` as sgrl_link_pred.py:102-105 uses it (`use_coalesce`,
    ogbl-collab): pairs sorted row-major, duplicates merged, their weights ADDED (None stays None)."""
    ei = np.asarray(edge_index)
    key = ei[0].astype(np.int64) * int(num_nodes) + ei[1]
    uniq, inv = np.unique(key, return_inverse=True)
    out = np.stack([uniq // int(num_nodes), uniq % int(num_nodes)])
    if edge_weight is None:
        return out, None
    w = np.asarray(edge_weight).reshape(-1)
    return out, np.bincount(inv, weights=w, minlength=len(uniq)).astype(w.dtype)


def train_graph(edge_index, num_nodes, edge_weight=None):
    """sgrl_link_pred.py:107-114: `ssp.csr_matrix((edge_weight, (row, col)), shape=(N, N))`, int
    ones when the data has no weights; duplicate entries are summed by scipy."""
    ei = np.asarray(edge_index)
    w = np.ones(ei.shape[1], dtype=int) if edge_weight is None else np.asarray(edge_weight).reshape(-1)
    return ssp.csr_matrix((w, (ei[0], ei[1])), shape=(int(num_nodes), int(num_nodes)))


def _shuffled_share(edges, percent):
    """A random `percent` of the columns of `edges` [2, n], in random order, drawn from numpy's
    GLOBAL generator — one `np.random.permutation(n)` per list, like the reference, so a seeded run
    sees the same links in the same order."""
    n = edges.size(1)
    keep = np.random.permutation(n)[:int(percent / 100 * n)]
    return edges[:, keep]


def pos_neg_edges(split, split_edge, percent=100):
    """utils.py:637-659 for splits that carry pre-sampled negatives (`do_edge_split` always writes
    'edge_neg'): [2, P] positives, [2, Q] negatives; positives are drawn first, then negatives, and
    the shuffle happens at percent = 100 too."""
    if 'edge_neg' not in split_edge['train']:
        raise NotImplementedError("on-the-fly negative sampling (PyG negative_sampling) is the "
                                  "producer's job: pass split_edge with 'edge_neg'")
    lists = [torch.as_tensor(split_edge[split][key]).t() for key in ('edge', 'edge_neg')]
    return tuple(_shuffled_share(e, percent) for e in lists)


def make_sign_kwargs(*, sign_k, sign_type, optimize_sign=True, k_heuristic=0,
                     k_node_set_strategy="intersection", use_feature=True):
    """sgrl_link_pred.py:142-154."""
    return {"sign_k": sign_k, "use_feature": use_feature, "sign_type": sign_type,
            "optimize_sign": optimize_sign, "k_heuristic": k_heuristic,
            "k_node_set_strategy": k_node_set_strategy}


def process_split(split, split_edge, edge_index, num_nodes, x, num_hops, *, sign_k, sign_type="PoS",
                  optimize_sign=True, k_heuristic=0, k_node_set_strategy="intersection",
                  use_feature=True, node_label="zo", ratio_per_hop=1.0, max_nodes_per_hop=None,
                  directed=False, edge_weight=None, use_coalesce=False, m=0, M=0, rw_seed=0, percent=100,
                  dataset_root=None, seed=0, device=None):
    """`SEALDataset.process` for `model == 'SIGN'`, non-pairwise (sgrl_link_pred.py:96-220):
    link lists of the split -> train graph A -> sign_kwargs (+ the global-operator stand-in for
    SoP / hybrid) -> positives with y = 1, negatives with y = 0 -> collate -> save.

    Returns (rows fp32 [ΣR, K'+1, 1+F], row_ptr int64 [L+1], y int64 [L], meta); positives first,
    then negatives, as `self.collate(pos_list + neg_list)` orders them.  With `dataset_root` the
    result is kept as a bundle under the reference's `data_appendix` directory extended by the
    operator settings (s3grl_amd.cache) and reloaded when it is already there — the counterpart of
    `InMemoryDataset` skipping `process()` when `processed_paths[0]` exists (sgrl_link_pred.py:87-94)."""
    mode = {"PoS": "pos_plus" if k_heuristic else "pos", "SoP": "sop", "hybrid": "hybrid"}.get(sign_type)
    if mode is None:
        raise NotImplementedError(f"sign_type {sign_type!r}")

    def compute():
        pos_edge, neg_edge = pos_neg_edges(split, split_edge, percent)
        ei, ew = (coalesce(edge_index, edge_weight, num_nodes) if use_coalesce      # :102-105
                  else (np.asarray(edge_index), edge_weight))
        A = train_graph(ei, num_nodes, ew)
        A_csc = A.tocsc() if directed else None
        sign_kwargs = make_sign_kwargs(sign_k=sign_k, sign_type=sign_type, optimize_sign=optimize_sign,
                                       k_heuristic=k_heuristic, k_node_set_strategy=k_node_set_strategy,
                                       use_feature=use_feature)
        # sgrl_link_pred.py:123-140,156-159: rw_kwargs is None unless ScaLed sampling is on; for the
        # optimised PoS flows the walks are cached per node and per (pos / neg) list first, and the
        # operators extract exactly the cached sets
        rw_kwargs = None
        if m:
            cached_pos = cached_neg = None
            if optimize_sign and sign_type == "PoS":
                cached_pos = create_rw_cache(A, pos_edge, device, m, M, seed=rw_seed)
                cached_neg = create_rw_cache(A, neg_edge, device, m, M, seed=rw_seed)
            rw_kwargs = {"rw_m": m, "rw_M": M, "sign": True, "seed": rw_seed,
                         "cached_pos_rws": cached_pos, "cached_neg_rws": cached_neg}
        powers_of_A = GlobalOperators(sign_k, ei.shape[1]) if sign_type in ("SoP", "hybrid") else []
        print("Setting up Positive Subgraphs")
        pos_list = extract_enclosing_subgraphs(pos_edge, A, x, 1, num_hops, node_label, ratio_per_hop,
                                               max_nodes_per_hop, directed, A_csc, rw_kwargs, sign_kwargs,
                                               powers_of_A=powers_of_A)
        print("Setting up Negative Subgraphs")
        neg_list = extract_enclosing_subgraphs(neg_edge, A, x, 0, num_hops, node_label, ratio_per_hop,
                                               max_nodes_per_hop, directed, A_csc, rw_kwargs, sign_kwargs,
                                               powers_of_A=powers_of_A)
        both = pos_list + neg_list                      # sgrl_link_pred.py:204
        if not isinstance(both, LinkDataList):
            both = _collate_plain(both)
        rows, row_ptr, y = both.collate()
        meta = {"split": split, "mode": mode, "sign_k": int(sign_k), "num_hops": int(num_hops),
                "num_pos": int(pos_edge.size(1)), "num_neg": int(neg_edge.size(1))}
        return rows, row_ptr, y, meta

    if dataset_root is None:
        return compute()
    appendix = _cache.data_appendix(num_hops=num_hops, node_label=node_label, ratio_per_hop=ratio_per_hop,
                                    seed=seed, max_nodes_per_hop=max_nodes_per_hop, m=m, M=M)
    path = _cache.cache_dir(dataset_root, appendix, mode=mode, sign_k=sign_k,
                            strategy=k_node_set_strategy) / _cache.bundle_name(split, percent)
    n_links = 0
    for key in ("edge", "edge_neg"):
        n_links += int(percent / 100 * len(split_edge[split][key]))
    return _cache.get_or_compute(path, compute, expect={"num_links": n_links, "mode": mode,
                                                        "sign_k": int(sign_k), "num_hops": int(num_hops)},
                                 device=device)


def _collate_plain(items):
    """A materialised list of per-link objects (a caller mixed in plain lists) -> LinkDataList."""
    if not items:
        return LinkDataList([], 0)
    keys = sorted((k for k in items[0].keys() if k.startswith("x") and k != "x"), key=lambda s: int(s[1:]))
    K = len(keys)
    chunks = []
    for d in items:
        rows = torch.stack([d["x"]] + [d[k] for k in keys], dim=1)
        chunks.append((rows, np.array([0, rows.shape[0]], dtype=np.int64), int(d.y)))
    return LinkDataList(chunks, K)
