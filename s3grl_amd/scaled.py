"""ScaLed walk caches at the drop-in boundary (reference utils.py:86-150, 425-443;
sgrl_link_pred.py:123-140).

The reference's caller builds, per (split, pos/neg) list, a dict `node -> unique nodes of the
node's rw_M random walks of length rw_m` (`utils.create_rw_cache`) and hands the two dicts to the
operators inside `rw_kwargs` (`cached_pos_rws`, `cached_neg_rws`); `k_hop_subgraph` then takes
`torch.unique(cat(cache[src], cache[dst]))` with src, dst moved to the front as the link's node
set.  Here:

* `create_rw_cache` is the same call on the engine (HIP random walks + sort/unique per node,
  `s3grl_walk_sets`): it returns a `WalkCache`, a read-only dict of CPU tensors that also keeps
  the CSR form, so handing it back costs no conversion;
* `resolve(rw_kwargs, y, ...)` turns whatever the caller put into `rw_kwargs` into what the engine
  needs: the caller's sets as a CSR (per node, or per link for `unique_nodes`), or — when no cache
  was handed in — the (m, M, seed) of the engine's own walks.  An ordinary dict of tensors (a
  cache built by the reference's own `create_rw_cache` with torch_cluster) works the same way:
  the node sets the engine extracts are then exactly the caller's.
"""
from __future__ import annotations

from collections.abc import Mapping

import numpy as np
import torch


class WalkCache(Mapping):
    """`node -> 1-D int64 tensor of the unique nodes its walks visited` (sorted, the node itself
    included), like the dict `utils.create_rw_cache` returns; backed by CSR arrays on the host."""

    def __init__(self, keys, ptr, nodes):
        self.keys_sorted = np.ascontiguousarray(keys, dtype=np.int64)     # ascending start nodes
        self.ptr = np.ascontiguousarray(ptr, dtype=np.int64)              # [len(keys) + 1]
        self.nodes = np.ascontiguousarray(nodes, dtype=np.int32)

    def _slot(self, node):
        node = int(node)
        i = int(np.searchsorted(self.keys_sorted, node))
        if i >= len(self.keys_sorted) or self.keys_sorted[i] != node:
            raise KeyError(node)
        return i

    def __getitem__(self, node):
        i = self._slot(node)
        return torch.from_numpy(self.nodes[self.ptr[i]:self.ptr[i + 1]].astype(np.int64))

    def __iter__(self):
        return iter(self.keys_sorted.tolist())

    def __len__(self):
        return len(self.keys_sorted)

    def __contains__(self, node):
        try:
            self._slot(node)
            return True
        except (KeyError, TypeError, ValueError):
            return False

    def node_csr(self, num_nodes):
        """One set per NODE of the graph (empty for nodes that are not keys)."""
        cnt = np.zeros(int(num_nodes), dtype=np.int64)
        cnt[self.keys_sorted] = np.diff(self.ptr)
        ptr = np.zeros(int(num_nodes) + 1, dtype=np.int64)
        np.cumsum(cnt, out=ptr[1:])
        return ptr, self.nodes          # keys ascend, so the concatenation is already in node order


def _cache_from_dict(d):
    """A plain dict node -> tensor / sequence (what the reference's own create_rw_cache returns)."""
    keys = np.fromiter((int(k) for k in d.keys()), dtype=np.int64, count=len(d))
    order = np.argsort(keys, kind="stable")
    vals = list(d.values())
    parts = [np.asarray(vals[i].cpu() if torch.is_tensor(vals[i]) else vals[i], dtype=np.int64).reshape(-1)
             for i in order]
    ptr = np.zeros(len(parts) + 1, dtype=np.int64)
    np.cumsum([len(p) for p in parts], out=ptr[1:])
    nodes = np.concatenate(parts) if parts else np.zeros(0, dtype=np.int64)
    return WalkCache(keys[order], ptr, nodes)


def _graph_of(sparse_adj, engine):
    """The walk graph: a scipy matrix, an engine Graph, or anything with torch_sparse's `.csr()`
    (rowptr, col, value) — what reference sgrl_link_pred.py:72-75 builds from data.edge_index."""
    from . import engine as _engine

    if isinstance(sparse_adj, _engine.Graph):
        return sparse_adj, False
    if hasattr(sparse_adj, "csr") and not hasattr(sparse_adj, "tocsr"):
        rowptr, col, _ = sparse_adj.csr()
        rowptr, col = rowptr.cpu().numpy(), col.cpu().numpy()
        import scipy.sparse as ssp

        n = len(rowptr) - 1
        sparse_adj = ssp.csr_matrix((np.ones(len(col), dtype=np.int64), col, rowptr), shape=(n, n))
    return engine.graph(sparse_adj), True


def create_rw_cache(sparse_adj, edges, device, rw_m, rw_M, seed=0, engine=None):
    """Reference utils.create_rw_cache (utils.py:425-443), same positional arguments: the unique
    endpoints of `edges` ([2, E] or any shape) are the start nodes, every start node walks rw_M
    times rw_m steps, and the cache maps it to the sorted unique nodes visited (itself included).
    `device` is accepted for signature parity (the engine's device is used).  `seed` selects the
    walks of the engine's counter-based generator (torch_cluster's are unseeded draws)."""
    from . import engine as _engine

    print("Setting up rw cache")
    eng = engine or _engine.default_engine()
    g, own = _graph_of(sparse_adj, eng)
    try:
        e = torch.as_tensor(edges).reshape(-1).to(torch.int64)
        starts = torch.unique(e).to(eng.device)
        ptr, nodes = eng.walk_sets(g, starts, int(rw_m), int(rw_M), int(seed))
        return WalkCache(starts.cpu().numpy(), ptr.cpu().numpy(), nodes.cpu().numpy())
    finally:
        if own:
            g.close()


def resolve(rw_kwargs, y, link_index, num_nodes):
    """What the operators do with `rw_kwargs` (reference utils.py:86-108).  Returns None (k-hop
    BFS), ("walks", m, M, seed) (no cache handed in: the engine draws the walks, where the
    reference calls torch_cluster per link), or ("sets", set_ptr, set_nodes, per_link) as numpy
    arrays.  Raises like the reference: ValueError for y not in {0, 1} (utils.py:94-99), KeyError
    for a link endpoint (or link) the cache does not hold."""
    if not rw_kwargs:
        return None
    if y == 1:
        cached = rw_kwargs.get('cached_pos_rws')
    elif y == 0:
        cached = rw_kwargs.get('cached_neg_rws')
    else:
        raise ValueError(f"Value of y is set to {y}, not 0/1")
    li = np.asarray(torch.as_tensor(link_index).cpu(), dtype=np.int64)
    if cached:
        cache = cached if isinstance(cached, WalkCache) else _cache_from_dict(cached)
        ends = np.unique(li)
        missing = ends[~np.isin(ends, cache.keys_sorted)]
        if len(missing):
            raise KeyError(int(missing[0]))
        ptr, nodes = cache.node_csr(num_nodes)
        return ("sets", ptr, nodes, 0)
    unique_nodes = rw_kwargs.get('unique_nodes')
    if unique_nodes:
        parts = [np.asarray(unique_nodes[(int(s), int(d))], dtype=np.int64).reshape(-1) for s, d in li.T]
        ptr = np.zeros(len(parts) + 1, dtype=np.int64)
        np.cumsum([len(p) for p in parts], out=ptr[1:])
        nodes = np.concatenate(parts) if parts else np.zeros(0, dtype=np.int64)
        return ("sets", ptr, nodes.astype(np.int32), 1)
    if not rw_kwargs.get('rw_m'):
        return None
    return ("walks", int(rw_kwargs['rw_m']), int(rw_kwargs['rw_M']), int(rw_kwargs.get('seed', 0)))
