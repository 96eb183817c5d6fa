#!/usr/bin/env python3
"""Generate the golden fixtures under tests/golden/.  Run in the BUILD CONTAINER only
(`python tests/golden/make_golden.py`); /root/reference does not exist on the GPU box.

Two kinds of fixture:

extract_*.npz, extract_directed_*.npz  — REFERENCE-PINNED.  Produced by executing the reference's own
    `utils.k_hop_subgraph` / `utils.neighbors` (reference utils.py:33-85) from /root/reference.
    Those two functions are pure scipy + python sets, but `utils.py` imports torch_geometric /
    torch_sparse / graphistry at module scope (absent from this image), so the import is made
    possible with INERT placeholder modules: they define names only, compute nothing, and no
    code path exercised here touches them.  Stored per case (ragged arrays concatenated with
    offsets): the node set per hop, hop distances, the masked induced matrix as global (u, v, value) triples *including* explicit
    zeros, and the common-neighbour set N(0) ∩ N(1) the PoS Plus flow selects
    (tuned_SIGN.py:233).

diffusion_*.npz — ORACLE-GENERATED (fp64 restatement, oracle/s3grl_oracle.py), regression pins
    and GPU-box inputs.  The reference cannot produce these here (its arithmetic lives in
    torch_sparse, not installed), see oracle/__init__.py.

Only inputs and expected outputs are stored; no reference source text.
"""
import os
import sys
import types
from pathlib import Path

os.environ["PYTHONDONTWRITEBYTECODE"] = "1"
sys.dont_write_bytecode = True

import numpy as np
import scipy.sparse as ssp

HERE = Path(__file__).resolve().parent
REPO = HERE.parent.parent
sys.path.insert(0, str(REPO))

REFERENCE = Path("/root/reference")


def _inert(name, **attrs):
    m = types.ModuleType(name)
    m.__dict__.update(attrs)
    sys.modules[name] = m
    return m


def import_reference_utils():
    """Import /root/reference/utils.py with inert placeholders for the absent packages."""
    class _Nothing:  # placeholder base so `class TunedSIGN(SIGN)` parses
        def __init__(self, *a, **k):
            pass

    def _absent(*a, **k):
        raise RuntimeError("placeholder for a package that is not installed")

    tg = _inert("torch_geometric")
    tg.utils = _inert("torch_geometric.utils", negative_sampling=_absent, add_self_loops=_absent,
                      train_test_split_edges=_absent, to_networkx=_absent, subgraph=_absent,
                      to_scipy_sparse_matrix=_absent, k_hop_subgraph=_absent)
    tg.loader = _inert("torch_geometric.loader", DataLoader=_Nothing)
    tg.data = _inert("torch_geometric.data", Data=_Nothing)
    tg.transforms = _inert("torch_geometric.transforms", SIGN=_Nothing)
    _inert("torch_sparse", SparseTensor=_Nothing, from_scipy=_absent, spspmm=_absent)
    _inert("graphistry")
    sys.path.insert(0, str(REFERENCE))
    import utils as ref_utils  # noqa: E402  (the reference module)
    return ref_utils


# ------------------------------------------------------------------------------------------
# fixture graphs
# ------------------------------------------------------------------------------------------
def csr_from_undirected(n, edges):
    """Same construction as reference sgrl_link_pred.py:107-114: int64 ones, both directions."""
    e = np.asarray(edges, dtype=np.int64).reshape(-1, 2)
    r = np.concatenate([e[:, 0], e[:, 1]])
    c = np.concatenate([e[:, 1], e[:, 0]])
    return ssp.csr_matrix((np.ones(len(r), dtype=np.int64), (r, c)), shape=(n, n))


def fixture_graphs():
    rng = np.random.default_rng(20240607)
    g = {}
    # (v) of SURVEY §8c: 5-node probe
    g["probe5"] = (5, np.array([[0, 2], [1, 2], [0, 3], [1, 3], [3, 4]]))
    g["triangle"] = (3, np.array([[0, 1], [0, 2], [1, 2]]))
    g["pair"] = (2, np.array([[0, 1]]))
    g["star_iso"] = (7, np.array([[0, 1], [0, 2], [0, 3], [0, 4], [4, 5]]))  # node 6 isolated
    # random sparse graph with hubs
    n = 300
    m = 900
    e = rng.integers(0, n, size=(m, 2))
    hub = rng.integers(0, n, size=(150, 1))
    e = np.vstack([e, np.hstack([np.full_like(hub, 7), hub]), np.hstack([np.full_like(hub, 11), hub])])
    e = e[e[:, 0] != e[:, 1]]
    e = np.unique(np.sort(e, axis=1), axis=0)
    g["rand300"] = (n, e)
    topo = np.load(REPO / "s3grl_amd" / "data" / "topo_usair.npz")
    g["usair"] = (int(topo["num_nodes"]), topo["edges"].astype(np.int64))
    topo = np.load(REPO / "s3grl_amd" / "data" / "topo_cora.npz")
    g["cora"] = (int(topo["num_nodes"]), topo["edges"].astype(np.int64))
    return g


def fixture_links(name, n, edges, rng):
    """Present edges, absent pairs, hub endpoints, degree-0 endpoints."""
    if name == "probe5":
        return [(0, 1), (1, 0), (3, 4), (2, 4), (0, 3)]
    if name == "triangle":
        return [(0, 1), (2, 0)]
    if name == "pair":
        return [(0, 1), (1, 0)]
    if name == "star_iso":
        return [(0, 1), (1, 2), (0, 6), (5, 6), (4, 5), (3, 5)]
    k = 16 if name in ("usair", "rand300") else 24
    pos = edges[rng.choice(len(edges), size=k, replace=False)]
    pos = [(int(a), int(b)) if i % 2 == 0 else (int(b), int(a)) for i, (a, b) in enumerate(pos)]
    deg = np.bincount(edges.ravel(), minlength=n)
    hubs = np.argsort(-deg)[:3]
    neg = []
    have = {(int(a), int(b)) for a, b in edges} | {(int(b), int(a)) for a, b in edges}
    while len(neg) < k:
        a, b = (int(v) for v in rng.integers(0, n, size=2))
        if a != b and (a, b) not in have:
            neg.append((a, b))
    extra = [(int(hubs[0]), int(hubs[1])), (int(hubs[1]), int(hubs[2]))]
    iso = np.flatnonzero(deg == 0)
    if len(iso):
        extra.append((int(iso[0]), int(hubs[0])))
    return pos + neg + extra


def canonical_case(ref_utils, A, src, dst, h, A_csc=None):
    X = np.arange(A.shape[0], dtype=np.float32).reshape(-1, 1)
    nodes, sub, dists, xs, y = ref_utils.k_hop_subgraph(
        src, dst, h, A, 1.0, None, node_features=X, y=1, directed=A_csc is not None, A_csc=A_csc,
        rw_kwargs=None)
    nodes = [int(v) for v in nodes]
    assert nodes[0] == src and nodes[1] == dst
    assert np.array_equal(np.asarray(xs).ravel(), np.asarray(nodes, dtype=np.float32))
    sub = ssp.csr_matrix(sub)
    coo_r = np.repeat(np.arange(sub.shape[0]), np.diff(sub.indptr))
    trip = np.stack([np.asarray(nodes)[coo_r], np.asarray(nodes)[sub.indices],
                     sub.data.astype(np.int64)], axis=1)
    trip = trip[np.lexsort((trip[:, 1], trip[:, 0]))]
    cn_local = ref_utils.neighbors({0}, sub) & ref_utils.neighbors({1}, sub)
    cn = sorted(nodes[int(a)] for a in cn_local)
    order = np.lexsort((np.asarray(nodes), np.asarray(dists)))
    return (np.asarray(nodes)[order], np.asarray(dists)[order], trip, np.asarray(cn, dtype=np.int64))


def make_extract(ref_utils):
    rng = np.random.default_rng(7)
    for name, (n, edges) in fixture_graphs().items():
        A = csr_from_undirected(n, edges)
        links = fixture_links(name, n, edges, rng)
        hops = {"cora": [3], "usair": [1, 2]}.get(name, [1, 2, 3])
        blob = {"num_nodes": np.int64(n), "edges": np.asarray(edges, dtype=np.int32),
                "links": np.asarray(links, dtype=np.int64), "hops": np.asarray(hops)}
        for h in hops:
            cat = {"nodes": [], "dists": [], "sub": [], "cn": []}
            for s, d in links:
                nodes, dists, trip, cn = canonical_case(ref_utils, A, s, d, h)
                cat["nodes"].append(nodes.astype(np.int32))
                cat["dists"].append(dists.astype(np.int8))
                cat["sub"].append(trip.astype(np.int32))
                cat["cn"].append(cn.astype(np.int32))
            # ragged arrays stored concatenated + offsets (one zip member per kind)
            for k, parts in cat.items():
                off = np.zeros(len(parts) + 1, dtype=np.int64)
                np.cumsum([len(p) for p in parts], out=off[1:])
                blob[f"h{h}_{k}"] = np.concatenate(parts, axis=0)
                blob[f"h{h}_{k}_off"] = off
        np.savez_compressed(HERE / f"extract_{name}.npz", **blob)
        print(f"extract_{name}.npz: {len(links)} links x hops {hops}")


def directed_fixture_graphs():
    """Directed versions of three fixture graphs: every undirected edge keeps one direction (chosen
    at random), a third of them both; plus a hand-made 6-node digraph with a source, a sink, a
    2-cycle and a self-loop-free chain."""
    rng = np.random.default_rng(31)
    g = {"tiny": (6, np.array([[0, 1], [1, 0], [0, 2], [3, 0], [2, 1], [1, 4], [4, 5], [3, 5]]))}
    und = fixture_graphs()
    for name in ("rand300", "usair", "cora"):
        n, e = und[name]
        e = np.asarray(e, dtype=np.int64)
        flip = rng.random(len(e)) < 0.5
        one = np.where(flip[:, None], e[:, ::-1], e)
        both = e[rng.random(len(e)) < 0.33]
        arcs = np.unique(np.vstack([one, both, both[:, ::-1]]), axis=0)
        g[name] = (n, arcs)
    return g


def make_extract_directed(ref_utils):
    """extract_directed_*.npz — REFERENCE-PINNED like extract_*.npz, for the directed branch of the
    reference's BFS (utils.py:58-63: out-neighbours through the CSR, in-neighbours through `A_csc`)
    and its directed induced matrix: `utils.k_hop_subgraph(directed=True, A_csc=A.tocsc())` itself."""
    rng = np.random.default_rng(17)
    und = fixture_graphs()
    for name, (n, arcs) in directed_fixture_graphs().items():
        A = ssp.csr_matrix((np.ones(len(arcs), dtype=np.int64), (arcs[:, 0], arcs[:, 1])), shape=(n, n))
        A_csc = A.tocsc()
        if name == "tiny":
            links = [(0, 1), (1, 0), (2, 5), (3, 4), (0, 5), (4, 3)]
        else:
            links = fixture_links(name, n, und[name][1], rng)
        hops = {"cora": [3], "usair": [1, 2]}.get(name, [1, 2, 3])
        blob = {"num_nodes": np.int64(n), "arcs": np.asarray(arcs, dtype=np.int32),
                "links": np.asarray(links, dtype=np.int64), "hops": np.asarray(hops)}
        for h in hops:
            cat = {"nodes": [], "dists": [], "sub": [], "cn": []}
            for s, d in links:
                nodes, dists, trip, cn = canonical_case(ref_utils, A, s, d, h, A_csc)
                cat["nodes"].append(nodes.astype(np.int32))
                cat["dists"].append(dists.astype(np.int8))
                cat["sub"].append(trip.astype(np.int32))
                cat["cn"].append(cn.astype(np.int32))
            for k, parts in cat.items():
                off = np.zeros(len(parts) + 1, dtype=np.int64)
                np.cumsum([len(p) for p in parts], out=off[1:])
                blob[f"h{h}_{k}"] = np.concatenate(parts, axis=0)
                blob[f"h{h}_{k}_off"] = off
        np.savez_compressed(HERE / f"extract_directed_{name}.npz", **blob)
        print(f"extract_directed_{name}.npz: {len(links)} links x hops {hops}, {len(arcs)} arcs")


def make_sampled(ref_utils):
    """sampled_*.npz — REFERENCE-PINNED control flow of the per-hop sampling (utils.py:62-74:
    what stays visited, ratio before cap, int() truncation, the empty-sample break), executed by
    the reference's own k_hop_subgraph.  Its draw, `random.sample(fringe, k)`, is swapped for the
    engine's keyed pick (oracle.hash_sampler) while the call runs, so the outcome is a fixed
    vector instead of a function of Python's RNG state and set iteration order."""
    import random

    import oracle

    seed = 1234
    settings = [(0.5, None), (1.0, 10), (0.7, 25), (0.05, None), (0.9, 3)]
    rng = np.random.default_rng(7)
    all_links = {name: fixture_links(name, n, edges, rng) for name, (n, edges) in fixture_graphs().items()}
    for name in ["rand300", "usair", "cora"]:
        n, edges = fixture_graphs()[name]
        A = csr_from_undirected(n, edges)
        links = all_links[name]
        h = 3 if name == "cora" else 2
        blob = {"num_nodes": np.int64(n), "edges": np.asarray(edges, dtype=np.int32),
                "links": np.asarray(links, dtype=np.int64), "num_hops": np.int64(h),
                "seed": np.int64(seed),
                "ratio": np.asarray([r for r, _ in settings]),
                "max_nodes": np.asarray([-1 if m is None else m for _, m in settings])}
        X = np.arange(n, dtype=np.float32).reshape(-1, 1)
        for si, (ratio, cap) in enumerate(settings):
            cat = {"nodes": [], "dists": [], "cn": []}
            for s_, d_ in links:
                pick = oracle.hash_sampler(seed, s_, d_)
                real = random.sample
                random.sample = lambda pop, k: pick(pop, k)
                try:
                    nodes, sub, dists, _, _ = ref_utils.k_hop_subgraph(
                        s_, d_, h, A, ratio, cap, node_features=X, y=1, directed=False, A_csc=None,
                        rw_kwargs=None)
                finally:
                    random.sample = real
                nodes = [int(v) for v in nodes]
                sub = ssp.csr_matrix(sub)
                cn_local = ref_utils.neighbors({0}, sub) & ref_utils.neighbors({1}, sub)
                order = np.lexsort((np.asarray(nodes), np.asarray(dists)))
                cat["nodes"].append(np.asarray(nodes, dtype=np.int32)[order])
                cat["dists"].append(np.asarray(dists, dtype=np.int8)[order])
                cat["cn"].append(np.asarray(sorted(nodes[int(a)] for a in cn_local), dtype=np.int32))
            for k, parts in cat.items():
                off = np.zeros(len(parts) + 1, dtype=np.int64)
                np.cumsum([len(p) for p in parts], out=off[1:])
                blob[f"s{si}_{k}"] = np.concatenate(parts, axis=0)
                blob[f"s{si}_{k}_off"] = off
        np.savez_compressed(HERE / f"sampled_{name}.npz", **blob)
        print(f"sampled_{name}.npz: {len(links)} links x {len(settings)} settings, h={h}")


def make_diffusion():
    import oracle

    rng = np.random.default_rng(11)
    for name, K, h in [("probe5", 3, 2), ("star_iso", 3, 2), ("rand300", 3, 2), ("usair", 2, 1)]:
        n, edges = fixture_graphs()[name]
        A = csr_from_undirected(n, edges)
        links = fixture_links(name, n, edges, np.random.default_rng(7))[:16]
        X = rng.standard_normal((n, 5))
        li = np.asarray(links).T
        kw = {"sign_k": K, "k_node_set_strategy": "intersection"}
        pos = oracle.get_PoS_prepped_ds(li, h, A, X, 1, kw, dtype=np.float64)
        plus = oracle.get_PoS_Plus_prepped_ds(li, h, A, X, 1, kw, dtype=np.float64)
        sop = oracle.get_SoP_prepped_ds(oracle.global_normalized_powers(A, K, np.float64), li, A, X,
                                        1, dtype=np.float64)
        blob = {"num_nodes": np.int64(n), "edges": np.asarray(edges, dtype=np.int64),
                "links": np.asarray(links, dtype=np.int64), "X": X, "K": np.int64(K),
                "num_hops": np.int64(h)}
        for tag, lst in [("pos", pos), ("plus", plus), ("sop", sop)]:
            rows, row_ptr, _ = oracle.collate_rows(lst, K)
            blob[f"{tag}_rows"] = rows
            blob[f"{tag}_row_ptr"] = row_ptr
            blob[f"{tag}_rows_global"] = np.concatenate([d["rows_global"] for d in lst])
        np.savez_compressed(HERE / f"diffusion_{name}.npz", **blob)
        print(f"diffusion_{name}.npz: {len(links)} links K={K} h={h}")


if __name__ == "__main__":
    if not REFERENCE.exists():
        sys.exit("needs /root/reference (build container only)")
    ref = import_reference_utils()
    only = sys.argv[1:]          # e.g. `make_golden.py directed`: regenerate one family, leave the rest
    if not only or "extract" in only:
        make_extract(ref)
    if not only or "directed" in only:
        make_extract_directed(ref)
    if not only or "sampled" in only:
        make_sampled(ref)
    if not only or "diffusion" in only:
        make_diffusion()
