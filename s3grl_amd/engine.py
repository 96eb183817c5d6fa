"""Host side of the engine above the C ABI: device buffers come from torch (plumbing only), every
computation goes through libs3grl_hip.so.  Nothing here computes on the CPU.

    eng = Engine()                                  # context on the current HIP device / stream
    g = eng.graph(A)                                # scipy CSR (structure only) -> device CSR
    res = eng.precompute(g, x, links, mode="pos", num_hops=3, sign_k=3)
    res.rows      # fp32 [ΣR, K+1, 1+F] on the device  == torch.cat([x, x1..xK], -1) of the
                  # reference's collated Data (sgrl_link_pred.py:204,449-459; models.py:372)
    res.row_ptr   # int64 [L+1]
"""
from __future__ import annotations

import ctypes as C
import os
import weakref
from dataclasses import dataclass, field

import numpy as np
import torch

from . import _native as N


def _ptr(t):
    return C.c_void_p(t.data_ptr()) if t is not None else C.c_void_p(0)


@dataclass
class Precomputed:
    rows: torch.Tensor        # [ΣR, K+1, 1+F] fp32, device
    row_ptr: torch.Tensor     # [L+1] int64, device
    row_nodes: torch.Tensor   # [ΣR] int64, device: global node id of every row
    stats: dict = field(default_factory=dict)

    @property
    def num_links(self):
        return self.row_ptr.numel() - 1


class Graph:
    """Device CSR of the train graph A.  Values are ignored (reference tuned_SIGN.py:153-156
    drops them at `ssp.find` -> `SparseTensor(row, col)`)."""

    def __init__(self, engine, indptr, indices, num_nodes, csc=None):
        """csc = (indptr, indices) of the same arcs in CSC form: a DIRECTED graph (the reference's
        `directed=True` with A and A_csc, sgrl_link_pred.py:116-119)."""
        self.engine = engine
        self.num_nodes = int(num_nodes)
        self.indptr = indptr      # int64 [N+1] device (kept alive for the caller's benefit)
        self.indices = indices    # int32 [nnz] device
        self.nnz = int(indices.numel())
        self.directed = csc is not None
        h = C.c_void_p()
        if csc is None:
            N.check(N.lib().s3grl_graph_create(engine._ctx, self.num_nodes, _ptr(indptr), _ptr(indices),
                                               self.nnz, C.byref(h)), "s3grl_graph_create")
        else:
            assert int(csc[1].numel()) == self.nnz
            N.check(N.lib().s3grl_graph_create_directed(engine._ctx, self.num_nodes, _ptr(indptr), _ptr(indices),
                                                        _ptr(csc[0]), _ptr(csc[1]), self.nnz, C.byref(h)),
                    "s3grl_graph_create_directed")
        self._h = h
        engine._children.add(self)

    def close(self):
        if getattr(self, "_h", None):
            if self.engine._ctx:          # the context owns the arena the handle points into
                N.lib().s3grl_graph_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class Features:
    """The feature operand x, prepared once for the gather: aligned dense rows, plus — `mode`
    "auto" when at most half of x's 16-byte chunks are non-zero, "packed" always — a packed copy
    (per row a bit mask of its non-zero chunks + those chunks) that the gather fetches instead.
    "dense": dense rows only.  "sparse": (column, value) rows accumulated in LDS (comparison only)."""

    _MODES = {"auto": 0, "dense": 1, "sparse": 2, "packed": 4}

    def __init__(self, engine, x, mode="auto"):
        assert x.is_cuda and x.dtype == torch.float32 and x.dim() == 2 and x.stride(1) == 1
        self.engine, self.tensor = engine, x          # keep x alive: the handle may borrow it
        h = C.c_void_p()
        N.check(N.lib().s3grl_features_create(engine._ctx, _ptr(x), x.stride(0), x.shape[0],
                                              x.shape[1], self._MODES[mode], C.byref(h)),
                "s3grl_features_create")
        self._h = h
        engine._children.add(self)
        nnz, sp = C.c_int64(), C.c_int32()
        N.check(N.lib().s3grl_features_info(h, C.byref(nnz), C.byref(sp)), "s3grl_features_info")
        self.nnz, self.is_sparse, self.is_packed = int(nnz.value), sp.value == 1, sp.value == 2

    @property
    def shape(self):
        return self.tensor.shape

    def close(self):
        if getattr(self, "_h", None):
            if self.engine._ctx:
                N.lib().s3grl_features_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class Plan:
    def __init__(self, engine, graph, links, cfg, node_sets=None):
        self.engine, self.graph, self.cfg = engine, graph, cfg
        self.num_links = int(links.shape[0])
        h = C.c_void_p()
        if node_sets is None:
            N.check(N.lib().s3grl_plan_create(engine._ctx, graph._h, _ptr(links), self.num_links,
                                              C.byref(cfg), C.byref(h)), "s3grl_plan_create")
        else:
            set_ptr, set_nodes, per_link = node_sets
            assert set_ptr.is_cuda and set_ptr.dtype == torch.int64 and set_ptr.is_contiguous()
            assert set_nodes.is_cuda and set_nodes.dtype == torch.int32 and set_nodes.is_contiguous()
            ns = N.NodeSets(set_ptr.data_ptr(), set_nodes.data_ptr() if set_nodes.numel() else None,
                            set_ptr.numel() - 1, set_nodes.numel(), int(bool(per_link)), 0)
            N.check(N.lib().s3grl_plan_create_sets(engine._ctx, graph._h, _ptr(links), self.num_links,
                                                   C.byref(cfg), C.byref(ns), C.byref(h)),
                    "s3grl_plan_create_sets")
        self._h = h
        engine._children.add(self)
        self._stats = None
        cnt = (C.c_int64 * 4)()
        N.check(N.lib().s3grl_plan_counts(h, cnt), "s3grl_plan_counts")
        self.total_rows, self.folded_links = int(cnt[1]), int(cnt[2])

    @property
    def stats(self):
        """Sizes and totals the plan measured (s3grl_plan_stats).  Read lazily: the first access
        waits for the plan's kernels, so ask after `run()` has queued the gather, not before."""
        if self._stats is None:
            st = N.PlanStats()
            N.check(N.lib().s3grl_plan_get_stats(self._h, C.byref(st)), "s3grl_plan_get_stats")
            self._stats = st.as_dict()
        return self._stats

    def row_ptr(self):
        out = torch.empty(self.num_links + 1, dtype=torch.int64, device=self.engine.device)
        N.check(N.lib().s3grl_plan_row_ptr(self._h, _ptr(out)), "s3grl_plan_row_ptr")
        return out

    def row_nodes(self):
        out = torch.empty(self.total_rows, dtype=torch.int64, device=self.engine.device)
        N.check(N.lib().s3grl_plan_row_nodes(self._h, _ptr(out)), "s3grl_plan_row_nodes")
        return out

    def export_subgraphs(self):
        """(node_ptr [L+1], nodes [Σn] hop-major / ascending id per hop, dists [Σn])."""
        dev = self.engine.device
        node_ptr = torch.empty(self.num_links + 1, dtype=torch.int64, device=dev)
        nodes = torch.empty(self.stats["extracted_nodes"], dtype=torch.int32, device=dev)
        dists = torch.empty(self.stats["extracted_nodes"], dtype=torch.int8, device=dev)
        N.check(N.lib().s3grl_plan_export_subgraphs(self._h, _ptr(node_ptr), _ptr(nodes),
                                                    _ptr(dists)), "s3grl_plan_export_subgraphs")
        return node_ptr, nodes, dists

    def run(self, x, out=None):
        if x is None:
            N.check(N.ERR_NO_FEATURES, "s3grl_run")
        eng = self.engine
        feat = x if isinstance(x, Features) else None
        if feat is None:
            assert x.is_cuda and x.dtype == torch.float32 and x.dim() == 2 and x.stride(1) == 1
        F = x.shape[1]
        K = self.cfg.sign_k
        R = self.total_rows
        if out is None:
            out = torch.empty((R, K + 1, F + 1), dtype=torch.float32, device=eng.device)
        else:
            assert out.is_contiguous() and out.numel() == R * (K + 1) * (F + 1)
        if feat is not None:
            N.check(N.lib().s3grl_run_features(eng._ctx, self._h, feat._h, _ptr(out)),
                    "s3grl_run_features")
        else:
            N.check(N.lib().s3grl_run(eng._ctx, self._h, _ptr(x), x.stride(0), F, _ptr(out)),
                    "s3grl_run")
        return out

    def gather_traffic(self, feat):
        """Bytes the gather launch of this plan on the prepared operand `feat` requests, exact
        (measurement only; see s3grl_plan_gather_traffic in include/s3grl.h)."""
        buf = (C.c_int64 * 8)()
        N.check(N.lib().s3grl_plan_gather_traffic(self.engine._ctx, self._h, feat._h, buf),
                "s3grl_plan_gather_traffic")
        keys = ["ids", "headers", "features", "coefficients", "output", "x_rows", "job_meta", "waves"]
        return {k: int(buf[i]) for i, k in enumerate(keys)}

    def close(self):
        if getattr(self, "_h", None):
            if self.engine._ctx:          # the context owns the arena the handle points into
                N.lib().s3grl_plan_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class Sop:
    """SoP global state: Â = D^-1/2 A D^-1/2 of the whole graph and Y_i = Â^i X
    (reference sgrl_link_pred.py:161-178 + tuned_SIGN.py:92-100 in closed form)."""

    def __init__(self, engine, graph, x, sign_k, multiplicity=None):
        """multiplicity (optional, fp32 [nnz] aligned with the graph's CSR entries): how often every
        stored pair occurs in the caller's uncoalesced edge_index — the reference's SoP operator counts
        and weighs duplicates (sgrl_link_pred.py:161-172); None = a coalesced graph."""
        if isinstance(x, Features):
            x = x.tensor
        assert x.is_cuda and x.dtype == torch.float32 and x.dim() == 2 and x.stride(1) == 1
        self.engine, self.graph, self.sign_k = engine, graph, int(sign_k)
        self.F = int(x.shape[1])
        h = C.c_void_p()
        if multiplicity is None:
            N.check(N.lib().s3grl_sop_create(engine._ctx, graph._h, _ptr(x), x.stride(0), self.F,
                                             self.sign_k, C.byref(h)), "s3grl_sop_create")
        else:
            m = torch.as_tensor(multiplicity).to(device=engine.device, dtype=torch.float32).contiguous()
            if m.numel() != graph.nnz:
                raise ValueError("multiplicity must have one entry per stored entry of A")
            N.check(N.lib().s3grl_sop_create_weighted(engine._ctx, graph._h, _ptr(x), x.stride(0), self.F,
                                                      self.sign_k, _ptr(m), C.byref(h)), "s3grl_sop_create_weighted")
        self._h = h
        engine._children.add(self)

    def sign_features(self):
        """[K, N, F] fp32: Â^1 X .. Â^K X of the whole graph (PyG SIGN(K) on it)."""
        out = torch.empty((self.sign_k, self.graph.num_nodes, self.F), dtype=torch.float32,
                          device=self.engine.device)
        N.check(N.lib().s3grl_sop_features(self.engine._ctx, self._h, _ptr(out)), "s3grl_sop_features")
        return out

    def run(self, links, out=None):
        eng = self.engine
        L = int(links.shape[0])
        if out is None:
            out = torch.empty((2 * L, self.sign_k + 1, self.F + 1), dtype=torch.float32,
                              device=eng.device)
        N.check(N.lib().s3grl_sop_run(eng._ctx, self._h, _ptr(links), L, _ptr(out)), "s3grl_sop_run")
        return out

    def close(self):
        if getattr(self, "_h", None):
            if self.engine._ctx:          # the context owns the arena the handle points into
                N.lib().s3grl_sop_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class Engine:
    """One context = one device + one stream + one workspace arena."""

    def __init__(self, device=None, preload=None):
        """preload: which code objects to load at once (Engine.PRELOAD_* bits; default: the PoS / PoS Plus set,
        S3GRL_NO_PRELOAD=1 or 0 = none: everything loads at first use)."""
        if not torch.cuda.is_available():
            raise RuntimeError("s3grl_amd needs a HIP device (MI355X); there is no CPU fallback")
        self.device = torch.device("cuda", torch.cuda.current_device()) if device is None \
            else torch.device(device)
        with torch.cuda.device(self.device):
            stream = torch.cuda.current_stream().cuda_stream
        h = C.c_void_p()
        N.check(N.lib().s3grl_context_create(self.device.index or 0, C.c_void_p(stream), C.byref(h)),
                "s3grl_context_create")
        self._ctx = h
        self._children = weakref.WeakSet()
        self.preload_ms = None
        if preload is None:
            preload = 0 if os.environ.get("S3GRL_NO_PRELOAD") else self.PRELOAD_POS
        if preload:
            self.preload(preload)

    PRELOAD_POS, PRELOAD_OTHER_K, PRELOAD_SOP = 1, 2, 4
    _UNITS = ["api", "relabel", "structure", "balls", "features", "packed", "gather", "csr", "hub", "links_a",
              "links_b", "links_c", "sop", "pool"]

    def preload(self, units=1):
        """Load the library's GPU code now (s3grl_context_preload): HIP loads a code object at the first launch
        of one of its kernels, which is most of a process's first Graph() and first plan (16 + 7 ms on USAir,
        whatever the graph).  Returns the milliseconds per unit."""
        ms = (C.c_double * 16)()
        N.check(N.lib().s3grl_context_preload(self._ctx, int(units), ms), "s3grl_context_preload")
        self.preload_ms = {u: float(ms[i]) for i, u in enumerate(self._UNITS) if ms[i] > 0.005}
        return self.preload_ms

    # ---- inputs ---------------------------------------------------------------------------
    def graph(self, A=None, *, indptr=None, indices=None, num_nodes=None, directed=False, A_csc=None):
        """From a scipy CSR matrix (what the reference hands its operators) or raw CSR arrays.
        `directed` (with the reference's `A_csc`, or derived from A when it is None): the arcs of A
        keep their direction — plans on the graph must then say `directed=True` too."""
        csc = None
        if A is not None:
            import scipy.sparse as ssp

            A = ssp.csr_matrix(A)
            if not A.has_canonical_format:
                A = A.copy()
                A.sum_duplicates()
            if A.nnz and (A.data == 0).any():
                raise ValueError("stored zeros in A are not supported (the reference's BFS would "
                                 "follow them while its operator drops them)")
            if directed:
                C_ = ssp.csc_matrix(A if A_csc is None else A_csc)
                if not C_.has_canonical_format:
                    C_ = C_.copy()
                    C_.sum_duplicates()
                if C_.shape != A.shape or C_.nnz != A.nnz:
                    raise ValueError("A_csc does not hold the arcs of A")
                csc = (torch.as_tensor(np.asarray(C_.indptr, dtype=np.int64)).to(self.device),
                       torch.as_tensor(np.asarray(C_.indices, dtype=np.int32)).to(self.device))
            elif (A != A.T).nnz:
                raise ValueError("A is not structurally symmetric: a directed graph needs directed=True "
                                 "(and its A_csc), like the reference's callers pass them")
            indptr, indices, num_nodes = A.indptr, A.indices, A.shape[0]
        elif directed:
            raise ValueError("a directed graph is given as a scipy matrix (A, A_csc)")
        ip = torch.as_tensor(np.asarray(indptr, dtype=np.int64)).to(self.device)
        ix = torch.as_tensor(np.asarray(indices, dtype=np.int32)).to(self.device)
        return Graph(self, ip, ix, num_nodes, csc)

    def features(self, x, mode="auto"):
        """Upload x (fp32 [N,F]) and prepare it for the gather; see `Features`."""
        x = torch.as_tensor(x)
        if not x.is_cuda:       # converted on the host: the upload is a plain copy
            x = x.to(dtype=torch.float32).contiguous()
        x = x.to(device=self.device, dtype=torch.float32).contiguous()
        return Features(self, x, mode)

    def links(self, link_index):
        """`link_index` in the reference's layout, [2, L] (tuned_SIGN.py:147 iterates
        `link_index.t().tolist()`), -> int64 [L, 2] on the device."""
        li = torch.as_tensor(link_index)
        if li.dim() != 2 or li.shape[0] != 2:
            raise ValueError("link_index must be [2, L]")
        if not li.is_cuda:      # transposed on the host: the upload is then a plain copy (no strided-copy kernel,
            return li.t().to(dtype=torch.int64).contiguous().to(self.device)   # whose first use costs ~60 ms)
        return li.t().to(device=self.device, dtype=torch.int64).contiguous()

    def link_pairs(self, pairs):
        """[L, 2] (src, dst) rows -> int64 [L, 2] on the device."""
        li = torch.as_tensor(pairs)
        if li.dim() != 2 or li.shape[1] != 2:
            raise ValueError("pairs must be [L, 2]")
        if not li.is_cuda:
            li = li.to(dtype=torch.int64).contiguous()
        return li.to(device=self.device, dtype=torch.int64).contiguous()

    # ---- the batched native entry ----------------------------------------------------------
    def subgraph_sizes(self, graph, links, *, num_hops=1, rw=None, ratio_per_hop=1.0,
                       max_nodes_per_hop=None, seed=0):
        """int64 [L] on the device: |S| of every link's enclosing subgraph — the sizing pass alone
        (count kernel + scan, ~1/8 of a plan), e.g. to balance multi-GPU shards by exact size."""
        p = self.plan(graph, links, mode="pos", num_hops=num_hops, sign_k=1, rw=rw, ratio_per_hop=ratio_per_hop,
                      max_nodes_per_hop=max_nodes_per_hop, seed=seed, count_only=True, fold_reversed=False)
        try:
            node_ptr = torch.empty(p.num_links + 1, dtype=torch.int64, device=self.device)
            N.check(N.lib().s3grl_plan_export_subgraphs(p._h, _ptr(node_ptr), C.c_void_p(0), C.c_void_p(0)),
                    "s3grl_plan_export_subgraphs")
            return node_ptr.diff()
        finally:
            p.close()

    def folded_mask(self, graph, links):
        """bool [L] on the device: the links a plan serves from the extraction of their reversed duplicate
        earlier in the list (a count-only plan reports them with an empty subgraph)."""
        p = self.plan(graph, links, mode="pos", num_hops=1, sign_k=1, count_only=True, fold_reversed=True)
        try:
            node_ptr = torch.empty(p.num_links + 1, dtype=torch.int64, device=self.device)
            N.check(N.lib().s3grl_plan_export_subgraphs(p._h, _ptr(node_ptr), C.c_void_p(0), C.c_void_p(0)),
                    "s3grl_plan_export_subgraphs")
            return node_ptr.diff() == 0
        finally:
            p.close()

    def link_costs(self, graph, links, *, num_hops=1, rw=None, ratio_per_hop=1.0, max_nodes_per_hop=None,
                   seed=0, mode="pos", fold_reversed=True):
        """fp32 [L] on the device: the relative cost of every link (s3grl_plan_link_cost), from the
        sizing pass alone — the weights a multi-GPU job balances its shards by.  `mode` as in the
        real run (PoS Plus rows enter the cost); a reversed duplicate that the real run folds into
        its primary is priced at its two output rows (`fold_reversed=False`: every link in full)."""
        p = self.plan(graph, links, mode=mode if mode in ("pos", "pos_plus") else "pos", num_hops=num_hops,
                      sign_k=2, rw=rw, ratio_per_hop=ratio_per_hop, max_nodes_per_hop=max_nodes_per_hop,
                      seed=seed, count_only=True, fold_reversed=fold_reversed)
        try:
            cost = torch.empty(p.num_links, dtype=torch.float32, device=self.device)
            N.check(N.lib().s3grl_plan_link_cost(p._h, _ptr(cost)), "s3grl_plan_link_cost")
            return cost
        finally:
            p.close()

    def walk_sets(self, graph, starts, m, M, seed=0):
        """(set_ptr int64 [S+1], set_nodes int32) on the device: for every start node the ascending
        unique nodes of its M uniform random walks of length m, itself included — the cache of
        reference utils.create_rw_cache (utils.py:425-443); see `s3grl_amd.scaled`."""
        starts = torch.as_tensor(starts).to(device=self.device, dtype=torch.int64).contiguous()
        S = int(starts.numel())
        set_ptr = torch.empty(S + 1, dtype=torch.int64, device=self.device)
        room = torch.empty(max(S * (int(m) * int(M) + 1), 1), dtype=torch.int32, device=self.device)
        N.check(N.lib().s3grl_walk_sets(self._ctx, graph._h, _ptr(starts), S, int(m), int(M),
                                        int(seed) & 0xffffffff, _ptr(set_ptr), _ptr(room)), "s3grl_walk_sets")
        return set_ptr, room[:int(set_ptr[-1])].clone()

    def node_sets(self, set_ptr, set_nodes, per_link=False):
        """Upload caller-side node sets (CSR: numpy / torch) for `plan(..., node_sets=...)`."""
        p = torch.as_tensor(np.asarray(set_ptr, dtype=np.int64) if not torch.is_tensor(set_ptr) else set_ptr)
        n = torch.as_tensor(np.asarray(set_nodes) if not torch.is_tensor(set_nodes) else set_nodes)
        if n.numel() and (int(n.min()) < -2**31 or int(n.max()) >= 2**31):
            raise ValueError("node ids in the sets do not fit int32")
        return (p.to(device=self.device, dtype=torch.int64).contiguous(),
                n.to(device=self.device, dtype=torch.int32).contiguous(), bool(per_link))

    def plan(self, graph, links, *, mode="pos", num_hops=1, sign_k=3, strategy="intersection",
             directed=False, full_stats=False, fold_reversed=True, rw=None, ratio_per_hop=1.0,
             max_nodes_per_hop=None, seed=0, count_only=False, node_sets=None):
        cfg = N.Cfg()
        cfg.mode = {"pos": N.MODE_POS, "pos_plus": N.MODE_POS_PLUS, "sop_restricted": N.MODE_SOP_RESTRICTED}[mode]
        cfg.num_hops = int(num_hops)
        cfg.sign_k = int(sign_k)
        if strategy not in N.STRATEGY:
            raise NotImplementedError(f"check strat {strategy}")      # tuned_SIGN.py:235
        cfg.strategy = N.STRATEGY[strategy]
        cfg.directed = int(bool(directed) or graph.directed)
        cfg.flags = (N.FLAG_FULL_STATS if full_stats else 0) | (0 if fold_reversed else N.FLAG_NO_FOLD) | \
            (N.FLAG_COUNT_ONLY if count_only else 0)
        cfg.seed = int(seed) & 0xffffffff
        cfg.ratio_per_hop = 1.0
        if node_sets is not None:
            # ScaLed subgraphs from node sets the caller cached (reference utils.py:94-108):
            # (set_ptr, set_nodes, per_link) device tensors, see `node_sets()`; num_hops, rw and the
            # per-hop sampling play no part
            if rw is not None:
                raise ValueError("node_sets and rw are alternatives")
            return Plan(self, graph, links, cfg, node_sets)
        if rw is not None:
            # ScaLed subgraphs (reference rw_kwargs): rw = (m, M[, seed]) — M walks of length m per
            # node instead of the k-hop BFS
            m, M = int(rw[0]), int(rw[1])
            if not (0 < m < 65536 and 0 < M < 65536):
                raise ValueError("rw = (m, M[, seed]) with 0 < m, M < 65536")
            cfg.rw_m, cfg.rw_M = m, M
            if len(rw) > 2:
                cfg.seed = int(rw[2]) & 0xffffffff
        else:
            # per-hop sampling (reference utils.py:66-70); the rw branch of the reference ignores it
            if ratio_per_hop is not None:
                if not ratio_per_hop > 0.0:
                    raise ValueError("ratio_per_hop must be > 0")
                cfg.ratio_per_hop = min(float(ratio_per_hop), 1.0)
            if max_nodes_per_hop is not None:
                if int(max_nodes_per_hop) < 1:
                    raise ValueError("max_nodes_per_hop must be >= 1 (or None)")
                cfg.max_nodes_per_hop = int(max_nodes_per_hop)
        return Plan(self, graph, links, cfg)

    def precompute(self, graph, x, links, *, mode="pos", num_hops=1, sign_k=3,
                   strategy="intersection", directed=False, out=None, rw=None, ratio_per_hop=1.0,
                   max_nodes_per_hop=None, seed=0, node_sets=None, multiplicity=None):
        """links: int64 [L,2] device tensor (see `links()`); x: fp32 [N,F] device tensor.
        mode: "pos", "pos_plus", "sop", "hybrid" (the reference's flows) or "sop_restricted" — NOT a reference
        flow: the SoP rows with every operator row restricted to the `num_hops`-ball of {src, dst} (BASELINE
        config 3's "2-hop subgraphs", SURVEY §8d's optional twin; needs sign_k - 1 <= num_hops)."""
        if x is None:
            N.check(N.ERR_NO_FEATURES, "precompute")
        if mode == "hybrid":
            # reference utils.py:454-480: PoS keys kept, SoP x2..xK appended as x{K+1}..x{2K-1}
            pos = self.precompute(graph, x, links, mode="pos", num_hops=num_hops, sign_k=sign_k, rw=rw,
                                  ratio_per_hop=ratio_per_hop, max_nodes_per_hop=max_nodes_per_hop,
                                  seed=seed, node_sets=node_sets)
            if sign_k == 1:
                return pos
            sop = self.precompute(graph, x, links, mode="sop", sign_k=sign_k, multiplicity=multiplicity)
            rows = torch.cat([pos.rows, sop.rows[:, 2:, :]], dim=1)
            return Precomputed(rows, pos.row_ptr, pos.row_nodes, dict(pos.stats))
        if mode == "sop_restricted" and multiplicity is not None:
            raise NotImplementedError("the num_hops-restricted SoP takes a coalesced graph (no multiplicities)")
        if mode == "sop":
            sop = Sop(self, graph, x, sign_k, multiplicity)
            try:
                rows = sop.run(links, out)
            finally:
                sop.close()
            L = links.shape[0]
            row_ptr = torch.arange(0, 2 * L + 1, 2, dtype=torch.int64, device=self.device)
            return Precomputed(rows, row_ptr, links.reshape(-1).clone(), {"num_links": L,
                                                                          "total_rows": 2 * L})
        plan = self.plan(graph, links, mode=mode, num_hops=num_hops, sign_k=sign_k,
                         strategy=strategy, directed=directed, rw=rw, ratio_per_hop=ratio_per_hop,
                         max_nodes_per_hop=max_nodes_per_hop, seed=seed, node_sets=node_sets)
        try:
            rows = plan.run(x, out)
            row_ptr, row_nodes = plan.row_ptr(), plan.row_nodes()      # queued behind the gather ...
            res = Precomputed(rows, row_ptr, row_nodes, dict(plan.stats))   # ... before the totals are waited for
        finally:
            plan.close()
        return res

    # ---- measurement -----------------------------------------------------------------------
    def set_profiling(self, on):
        N.check(N.lib().s3grl_context_set_profiling(self._ctx, int(bool(on))), "set_profiling")

    def timings(self):
        buf = (C.c_double * 16)()
        N.check(N.lib().s3grl_context_timings(self._ctx, buf), "s3grl_context_timings")
        keys = ["structure_ms", "propagate_ms", "gather_ms", "sop_setup_ms", "sop_run_ms",
                "gather_launches", "plans", "sop_runs", "sop_rows_ms", "sop_spmm_ms", "sop_setups"]
        return {k: float(buf[i]) for i, k in enumerate(keys)}

    def trim(self):
        """Give the workspace blocks cached between calls back to the HIP allocator (call it when
        the precompute phase is over and training needs the memory); returns the bytes freed."""
        n = C.c_int64()
        N.check(N.lib().s3grl_context_trim(self._ctx, C.byref(n)), "s3grl_context_trim")
        return int(n.value)

    def close(self):
        if getattr(self, "_ctx", None):
            for child in list(self._children):
                child.close()
            N.lib().s3grl_context_destroy(self._ctx)
            self._ctx = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


_default = {}
_default_lock = __import__("threading").Lock()


def default_engine(device=None, preload=None):
    """The process-wide engine of a device (created on first use; a warm-up thread and the caller may ask for
    it at the same time)."""
    with _default_lock:
        key = str(device) if device is not None else "cur%d" % torch.cuda.current_device()
        if key not in _default:
            _default[key] = Engine(device, preload=preload)
        return _default[key]


def precompute(indptr, indices, X, links, *, mode="pos", num_hops=1, sign_k=3,
               strategy="intersection", rw=None, device=None):
    """The native batched entry the `tuned_SIGN` wrappers delegate to, on the
    current HIP device:

        rows, row_ptr, node_count = precompute(indptr, indices, X, links, mode=..., ...)

    `indptr`/`indices`: CSR structure of the (symmetric) train graph; `X`: fp32 [N, F];
    `links`: int64 [2, L] as the reference passes them.  Returns `rows` fp32 [sum R, K+1, 1+F]
    (the collated x, x1..xK of every link, label column first), `row_ptr` int64 [L+1] and
    `node_count` int32 [L] (subgraph sizes; zeros for SoP, which extracts no subgraph)."""
    eng = default_engine(device)
    n = int(len(indptr) - 1)
    g = eng.graph(indptr=indptr, indices=indices, num_nodes=n)
    try:
        lk = eng.links(links)
        xd = eng.features(X)
        if mode in ("pos", "pos_plus"):
            plan = eng.plan(g, lk, mode=mode, num_hops=num_hops, sign_k=sign_k, strategy=strategy,
                            full_stats=True, rw=rw)
            try:
                rows = plan.run(xd)
                row_ptr = plan.row_ptr()
                node_count = plan.export_subgraphs()[0].diff().to(torch.int32)
            finally:
                plan.close()
        else:
            res = eng.precompute(g, xd, lk, mode=mode, num_hops=num_hops, sign_k=sign_k, rw=rw)
            rows, row_ptr = res.rows, res.row_ptr
            node_count = torch.zeros(lk.shape[0], dtype=torch.int32, device=eng.device)
    finally:
        g.close()
    return rows, row_ptr, node_count
