"""Synthetic twins of BASELINE.json's configs: real public topologies
(s3grl_amd/data/topo_*.npz, see tools/make_topologies.py), synthetic features, and the link
lists the reference's driver would hand to the operators.

`edge_split` restates the *shape* of reference utils.py:588-634 (`do_edge_split` ->
PyG `train_test_split_edges` + `negative_sampling`): 85/5/10 split of the undirected edges,
train positives in BOTH directions sorted by (row, col), val/test positives in one direction,
as many negatives as positives per split, the train graph = train edges only
(sgrl_link_pred.py:852-855).  It is host-side numpy input preparation, not the hot path.
"""
from __future__ import annotations

from dataclasses import dataclass
from pathlib import Path

import numpy as np
import scipy.sparse as ssp

DATA = Path(__file__).resolve().parent / "data"


def load_topology(name):
    t = np.load(DATA / f"topo_{name}.npz")
    n, e = int(t["num_nodes"]), t["edges"].astype(np.int64)
    return n, e


def load_features(name):
    """The public node features of a dataset as a dense fp32 matrix, NOT yet normalised
    (s3grl_amd/data/feat_<name>.npz: the CSR structure of the non-zeros, see tools/make_features.py;
    an empty `data` array means every stored value is 1 — Cora's binary bag of words)."""
    t = np.load(DATA / f"feat_{name}.npz")
    n, f = (int(v) for v in t["shape"])
    indptr, indices = t["indptr"].astype(np.int64), t["indices"].astype(np.int64)
    data = t["data"] if t["data"].size else np.ones(len(indices), dtype=np.float32)
    X = np.zeros((n, f), dtype=np.float32)
    X[np.repeat(np.arange(n), np.diff(indptr)), indices] = data
    return X


def chung_lu(n, m, gamma=2.5, d_max=700, seed=3):
    """Power-law expected-degree graph (BASELINE config 5), self-loops / multi-edges removed."""
    rng = np.random.default_rng(seed)
    w = (1.0 - rng.random(n)) ** (-1.0 / (gamma - 1.0))
    w = np.minimum(w, d_max)
    p = w / w.sum()
    u = rng.choice(n, size=int(m * 1.15), p=p)
    v = rng.choice(n, size=int(m * 1.15), p=p)
    e = np.stack([np.minimum(u, v), np.maximum(u, v)], 1)
    e = e[e[:, 0] != e[:, 1]]
    e = np.unique(e, axis=0)
    if len(e) > m:
        e = e[rng.choice(len(e), m, replace=False)]
    e = e.astype(np.int64)
    return n, e


def csr_from_undirected(n, edges):
    """int64 ones, both directions — reference sgrl_link_pred.py:107-114."""
    e = np.asarray(edges, dtype=np.int64).reshape(-1, 2)
    r = np.concatenate([e[:, 0], e[:, 1]])
    c = np.concatenate([e[:, 1], e[:, 0]])
    return ssp.csr_matrix((np.ones(len(r), dtype=np.int64), (r, c)), shape=(n, n))


def _sample_non_edges(n, forbidden_keys, count, rng, directed_pairs=False):
    out = np.empty((0, 2), dtype=np.int64)
    seen = set()
    while len(out) < count:
        k = int((count - len(out)) * 1.3) + 16
        a = rng.integers(0, n, size=k)
        b = rng.integers(0, n, size=k)
        if not directed_pairs:
            a, b = np.minimum(a, b), np.maximum(a, b)
        key = a * n + b
        ok = (a != b) & ~np.isin(key, forbidden_keys)
        cand = np.stack([a[ok], b[ok]], 1)
        keep = []
        for i, kk in enumerate(key[ok].tolist()):
            if kk not in seen:
                seen.add(kk)
                keep.append(i)
        out = np.concatenate([out, cand[keep]])
    return out[:count]


@dataclass
class Split:
    num_nodes: int
    train_edges: np.ndarray          # undirected train edges [E_tr, 2], u < v
    A: ssp.csr_matrix                # train graph, both directions
    links: dict                      # split -> (pos [2,P], neg [2,Q]) in the reference's layout

    def split_edge(self):
        """The dict `do_edge_split` returns (reference utils.py:626-633): per split 'edge' [P, 2] and
        'edge_neg' [Q, 2]."""
        return {s: {"edge": self.links[s][0].T.copy(), "edge_neg": self.links[s][1].T.copy()}
                for s in ("train", "valid", "test")}

    def edge_index(self):
        """`data.edge_index` after the split = the train positives, both directions
        (sgrl_link_pred.py:852-855)."""
        return self.links["train"][0]

    def all_links(self, shuffle=True, seed=12345):
        """Concatenation in the order the reference's driver issues its 6 operator calls
        (train/valid/test x pos,neg; sgrl_link_pred.py:1116-1243, :195-203) -> [2, L], y [L].
        Every list is permuted first, as `get_pos_neg_edges` does with `np.random.permutation`
        even at percent = 100 (utils.py:650-657): the operators never see the coalesced
        (row, col) order of the split.  `shuffle=False` keeps that order (locality experiments)."""
        import os

        if os.environ.get("S3GRL_SORTED_LINKS"):
            shuffle = False
        rng = np.random.default_rng(seed)
        parts, ys = [], []
        for s in ("train", "valid", "test"):
            pos, neg = self.links[s]
            if shuffle:
                pos = pos[:, rng.permutation(pos.shape[1])]
                neg = neg[:, rng.permutation(neg.shape[1])]
            parts += [pos, neg]
            ys += [np.ones(pos.shape[1], dtype=np.int64), np.zeros(neg.shape[1], dtype=np.int64)]
        return np.concatenate(parts, axis=1), np.concatenate(ys)


def edge_split(n, edges, seed=0, val_ratio=0.05, test_ratio=0.1):
    rng = np.random.default_rng(seed)
    e = np.asarray(edges, dtype=np.int64)
    e = e[rng.permutation(len(e))]
    n_v = int(np.floor(val_ratio * len(e)))
    n_t = int(np.floor(test_ratio * len(e)))
    val, test, train = e[:n_v], e[n_v:n_v + n_t], e[n_v + n_t:]
    both = np.concatenate([train, train[:, ::-1]])
    both = both[np.lexsort((both[:, 1], both[:, 0]))]          # to_undirected -> coalesced order
    all_keys = np.concatenate([e[:, 0] * n + e[:, 1], e[:, 1] * n + e[:, 0]])
    neg_vt = _sample_non_edges(n, all_keys, n_v + n_t, rng)
    train_keys = np.concatenate([train[:, 0] * n + train[:, 1], train[:, 1] * n + train[:, 0]])
    neg_tr = _sample_non_edges(n, train_keys, len(both), rng, directed_pairs=True)
    links = {
        "train": (both.T.copy(), neg_tr.T.copy()),
        "valid": (val.T.copy(), neg_vt[:n_v].T.copy()),
        "test": (test.T.copy(), neg_vt[n_v:].T.copy()),
    }
    return Split(n, train, csr_from_undirected(n, train), links)


def normalize_features(X):
    """PyG `NormalizeFeatures` as the reference applies it (Planetoid transform
    sgrl_link_pred.py:851 and again :1000-1003 after `init_features`):
    `value = value - value.min(); value.div_(value.sum(dim=-1, keepdim=True).clamp_(min=1.))` *(3p)*
    — the GLOBAL minimum is subtracted (a no-op for the non-negative bag-of-words / one-hot
    matrices), then every row is divided by its sum clamped from below at 1, so all-zero rows
    stay zero and rows whose sum is below 1 are left unscaled.  fp32 like the reference."""
    X = np.asarray(X, dtype=np.float32)
    X = X - X.min() if X.size else X
    s = np.maximum(X.sum(axis=1, keepdims=True, dtype=np.float32), np.float32(1.0))
    return (X / s).astype(np.float32)


def row_normalize(X):
    """Kept name of `normalize_features` (the synthetic twins' feature step)."""
    return normalize_features(X)


def read_seal_edges(path):
    """Reader of the SEAL txt datasets (USAir, NS, PB, Yeast, ...; reference data_utils.py:76-93,
    used at sgrl_link_pred.py:868-875): `edges.txt` under `path` (or `path` itself when it is a
    file), the first two whitespace-separated columns of every line are node names; ids are the
    rank of the name in the SORTED LIST OF STRINGS ('10' < '2').  Returns (num_nodes,
    edges int64 [E, 2]) in file order, duplicates and self-loops kept exactly as the reference's
    `read_edges` returns them; `undirected_unique` gives the simple-graph topology."""
    from pathlib import Path as _P

    p = _P(path)
    if p.is_dir():
        p = p / "edges.txt"
    rows = []
    with open(p) as f:
        for line in f.readlines():
            a, b = line.strip().split()[:2]
            rows.append((a, b))
    names = sorted(set(x for ab in rows for x in ab))
    idx = {name: i for i, name in enumerate(names)}
    edges = np.array([[idx[a], idx[b]] for a, b in rows], dtype=np.int64).reshape(-1, 2)
    return len(names), edges


def undirected_unique(edges):
    """u < v, sorted, duplicates and self-loops dropped (PyG `to_undirected` + coalesce, upper half)."""
    e = np.asarray(edges, dtype=np.int64).reshape(-1, 2)
    e = e[e[:, 0] != e[:, 1]]
    e = np.unique(np.sort(e, axis=1), axis=0)
    return e


def sparse_uniform_features(n, dim, nnz_per_row, seed):
    rng = np.random.default_rng(seed)
    X = np.zeros((n, dim), dtype=np.float32)
    cols = np.argsort(rng.random((n, dim)), axis=1)[:, :nnz_per_row]
    X[np.arange(n)[:, None], cols] = rng.random((n, nnz_per_row), dtype=np.float32)
    return row_normalize(X)


def one_hot_degree(A, max_degree=1024):
    """PyG `OneHotDegree(max_degree=1024)` on the TRAIN graph (reference sgrl_link_pred.py:961-963;
    `data.edge_index` holds the train edges only at that point, :852-855): out-degree = stored
    entries of the row, one-hot over max_degree + 1 = 1 025 classes.  A degree above max_degree
    makes PyG's `F.one_hot(deg, num_classes=max_degree + 1)` raise; so does this.  The caller
    concatenates it to x (`cat=True`): `init_degree_features`."""
    deg = np.diff(A.indptr)
    if len(deg) and deg.max() > max_degree:
        raise RuntimeError("Class values must be smaller than num_classes.")   # torch's message
    oh = np.zeros((A.shape[0], max_degree + 1), dtype=np.float32)
    oh[np.arange(A.shape[0]), deg] = 1
    return oh


def init_degree_features(X, A, max_degree=1024):
    """`init_features == "degree"` (sgrl_link_pred.py:961-963 + :1000-1003): the one-hot degree is
    appended to the existing x (PyG `cat=True`; x alone when the dataset has no features), then
    NormalizeFeatures."""
    oh = one_hot_degree(A, max_degree)
    X = oh if X is None else np.hstack([np.asarray(X, dtype=np.float32).reshape(A.shape[0], -1), oh])
    return normalize_features(X)


@dataclass
class Workload:
    name: str
    split: Split
    X: np.ndarray
    mode: str
    num_hops: int
    sign_k: int

    @property
    def A(self):
        return self.split.A


def make(name):
    """BASELINE.md §3 configs by name."""
    if name == "pubmed_pos_k3":      # headline: PubMed PoS sign_k=3, h=3, F=500
        n, e = load_topology("pubmed")
        return Workload(name, edge_split(n, e, seed=2), sparse_uniform_features(n, 500, 50, 2),
                        "pos", 3, 3)
    if name == "pubmed_pos_k3_dense":   # control: node2vec-like DENSE features (sgrl_link_pred.py:966-971
        # produces such an x for init_features=n2v), same graph / links / F as the headline, so that
        # SURVEY §8(d)'s B_link IS the traffic (no zero chunks to skip)
        n, e = load_topology("pubmed")
        X = np.random.default_rng(7).standard_normal((n, 500)).astype(np.float32)
        return Workload(name, edge_split(n, e, seed=2), normalize_features(X), "pos", 3, 3)
    if name == "pubmed_pos_k5":      # config 4
        n, e = load_topology("pubmed")
        return Workload(name, edge_split(n, e, seed=2), sparse_uniform_features(n, 500, 50, 2),
                        "pos", 3, 5)
    if name == "pubmed_sop_k3":      # config 3: degree features appended (F = 1525)
        n, e = load_topology("pubmed")
        sp = edge_split(n, e, seed=2)
        return Workload(name, sp, init_degree_features(sparse_uniform_features(n, 500, 50, 2), sp.A),
                        "sop", 2, 3)
    if name == "pubmed_sop_k3_2hop":   # config 3's optional twin: SoP rows restricted to the 2-hop ball —
        n, e = load_topology("pubmed")   # NOT the reference's semantics (its SoP ignores num_hops), reported separately
        sp = edge_split(n, e, seed=2)
        return Workload(name, sp, init_degree_features(sparse_uniform_features(n, 500, 50, 2), sp.A),
                        "sop_restricted", 2, 3)
    if name == "cora_posplus_k3":    # config 2
        n, e = load_topology("cora")
        rng = np.random.default_rng(1)
        X = (rng.random((n, 1433)) < 0.0127).astype(np.float32)
        return Workload(name, edge_split(n, e, seed=1), row_normalize(X), "pos_plus", 3, 3)
    if name == "cora_posplus_k3_real":   # config 2 on Cora's own bag-of-words rows (paper entry
        # configs/paper/auc_s3grl.json:259; Planetoid NormalizeFeatures, sgrl_link_pred.py:851)
        n, e = load_topology("cora")
        return Workload(name, edge_split(n, e, seed=1), normalize_features(load_features("cora")), "pos_plus", 3, 3)
    if name == "usair_pos_k2":       # config 1
        n, e = load_topology("usair")
        X = np.random.default_rng(0).standard_normal((n, 16)).astype(np.float32)
        return Workload(name, edge_split(n, e, seed=0), row_normalize(X), "pos", 1, 2)
    if name == "collab_pos_k3":      # config 5 (synthetic collab-scale)
        n, e = chung_lu(235000, 1300000, seed=3)
        X = np.random.default_rng(4).standard_normal((n, 128)).astype(np.float32)
        sp = Split(n, e, csr_from_undirected(n, e), {})
        rng = np.random.default_rng(5)
        pos = e[rng.choice(len(e), 500000, replace=False)]
        keys = np.concatenate([e[:, 0] * n + e[:, 1], e[:, 1] * n + e[:, 0]])
        neg = _sample_non_edges(n, keys, 500000, rng)
        sp.links = {"train": (pos.T.copy(), neg.T.copy()),
                    "valid": (np.zeros((2, 0), np.int64),) * 2, "test": (np.zeros((2, 0), np.int64),) * 2}
        return Workload(name, sp, X, "pos", 1, 3)
    raise KeyError(name)
