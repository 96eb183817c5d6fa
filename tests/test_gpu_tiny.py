"""link_tiny_kernel (csrc/s3grl_hub.hip): one-hop PoS links of at most 32 / 64 nodes run with one lane per node
— half a wavefront or a wavefront per link, the masked induced adjacency as one bit mask per lane — instead of
link_full_kernel.  What the reference computes there: utils.py:47-85 (one-hop extraction, masked target edge),
tuned_SIGN.py:153-185 (D^-1/2 A D^-1/2, its powers, rows src / dst).

S3GRL_NO_TINY keeps those links on link_full_kernel; S3GRL_FORCE_ONEHOP sends the small fixture graphs down the
one-hop road (by default it is for big graphs)."""
import numpy as np
import pytest

import oracle
from conftest import csr_from_undirected, load_extract

pytestmark = pytest.mark.gpu
TOL = 1e-5


@pytest.fixture(scope="module")
def eng():
    import torch
    from s3grl_amd.engine import Engine

    assert torch.cuda.is_available()
    e = Engine("cuda:0")
    yield e
    e.close()


def rel_err(got, ref):
    ref = np.asarray(ref, dtype=np.float64)
    got = np.asarray(got, dtype=np.float64)
    if not ref.size:
        return 0.0
    scale = np.maximum(np.abs(ref), np.abs(ref).max(axis=-1, keepdims=True))
    return float(np.max(np.clip(np.abs(got - ref) - 1e-10, 0, None) / np.maximum(scale, 1e-30)))


def _class_counts(eng, monkeypatch, G, L, K):
    """class counts of a plan as the library prints them with S3GRL_DEBUG (lists 26 / 27: the tiny classes)"""
    import os
    import re
    import tempfile

    monkeypatch.setenv("S3GRL_DEBUG", "1")
    with tempfile.TemporaryFile(mode="w+b") as tmp:
        saved = os.dup(2)
        os.dup2(tmp.fileno(), 2)
        try:
            eng.plan(G, L, mode="pos", num_hops=1, sign_k=K).close()
        finally:
            os.dup2(saved, 2)
            os.close(saved)
        tmp.seek(0)
        text = tmp.read().decode()
    monkeypatch.delenv("S3GRL_DEBUG", raising=False)
    m = re.search(r"classes:((?: -?\d+)+)", text)
    assert m, text
    return [int(x) for x in m.group(1).split()]


@pytest.mark.parametrize("name", ["cora", "rand300", "usair", "triangle", "pair", "star_iso", "probe5"])
@pytest.mark.parametrize("K", [2, 3, 5])   # (sign_k = 1: the operator stays within one hop of the row, the general kernel)
def test_tiny_links_equal_the_general_one_hop_kernel(eng, monkeypatch, name, K):
    """Node lists, row nodes, statistics (n, vol(S), induced entries, support): exactly; rows: two summation
    orders of the same fp32 sums, and the oracle.  Reversed duplicates (folded) and self-loops included."""
    import torch

    g = load_extract(name)
    n = int(g["num_nodes"])
    A = csr_from_undirected(n, g["edges"])
    X = np.random.default_rng(3).standard_normal((n, 23)).astype(np.float32)
    links = np.concatenate([g["links"], g["links"][:3, ::-1]])
    f = eng.features(X)
    L = eng.links(links.T)
    monkeypatch.setenv("S3GRL_FORCE_ONEHOP", "1")
    out = []
    for tiny in (False, True):
        if not tiny:
            monkeypatch.setenv("S3GRL_NO_TINY", "1")
        else:
            monkeypatch.delenv("S3GRL_NO_TINY", raising=False)
        G = eng.graph(A)
        counts = _class_counts(eng, monkeypatch, G, L, K)
        p = eng.plan(G, L, mode="pos", num_hops=1, sign_k=K, full_stats=True)
        rows = p.run(f).clone()
        st = dict(p.stats)
        st.pop("workspace_bytes")
        out.append((rows, st, [t.clone() for t in p.export_subgraphs()], p.row_ptr().clone(), p.row_nodes().clone(),
                    counts))
        p.close(), G.close()
    monkeypatch.delenv("S3GRL_FORCE_ONEHOP", raising=False)
    (ra, sa, ea, pa, na, ca), (rb, sb, eb, pb, nb, cb) = out
    assert sum(ca[26:28]) == 0
    sizes = [len(set(A.indices[A.indptr[s]:A.indptr[s + 1]]) | set(A.indices[A.indptr[d]:A.indptr[d + 1]]) | {s, d})
             for s, d in g["links"]]
    small = sum(1 for x in sizes if x <= 64)
    assert (sum(cb[26:28]) > 0) == (small > 0)      # the hook really switches kernels
    if small:
        assert sum(cb[14:21]) < sum(ca[14:21])
    assert sa == sb and torch.equal(pa, pb) and torch.equal(na, nb)
    assert all(torch.equal(x, y) for x, y in zip(ea, eb))
    assert rel_err(rb.cpu().numpy(), ra.cpu().numpy()) < 3e-6
    kw = {"sign_k": K, "k_node_set_strategy": "intersection"}
    ref, ptr, _ = oracle.collate_rows(oracle.get_PoS_prepped_ds(links.T, 1, A, X.astype(np.float64), 1, kw,
                                                               dtype=np.float64), K)
    assert np.array_equal(pb.cpu().numpy(), ptr)
    assert rel_err(rb.cpu().numpy(), ref) < TOL
    f.close()


def test_tiny_links_with_self_loops_and_both_widths(eng, monkeypatch):
    """A random graph with self-loops whose one-hop subgraphs straddle 32 and 64 nodes: both widths of the
    kernel and link_full_kernel in one plan, against the C restatement."""
    from oracle import c_oracle
    from s3grl_amd import workloads

    rng = np.random.default_rng(21)
    n = 4000
    e = rng.integers(0, n, size=(26000, 2))
    e = np.unique(np.sort(e, axis=1), axis=0)      # (self-loops stay)
    A = workloads.csr_from_undirected(n, e[e[:, 0] != e[:, 1]])
    A = (A + __import__("scipy.sparse").sparse.diags((np.bincount(e[e[:, 0] == e[:, 1]][:, 0], minlength=n) > 0).astype(A.dtype))).tocsr()
    A.sort_indices()
    links = np.concatenate([e[e[:, 0] != e[:, 1]][rng.choice(20000, 1500, replace=False)],
                            rng.integers(0, n, size=(500, 2))])
    links = links[links[:, 0] != links[:, 1]]
    X = rng.random((n, 17)).astype(np.float32)
    monkeypatch.setenv("S3GRL_FORCE_ONEHOP", "1")
    G = eng.graph(A)
    f = eng.features(X)
    L = eng.links(links.T)
    counts = _class_counts(eng, monkeypatch, G, L, 3)
    assert counts[26] > 0 and counts[27] > 0
    res = eng.precompute(G, f, L, mode="pos", num_hops=1, sign_k=3)
    ref, ptr, nodes, _ = c_oracle.pos_rows(links.T, 1, A, X, 3, plus=False)
    assert np.array_equal(res.row_nodes.cpu().numpy(), nodes)
    assert np.array_equal(res.row_ptr.cpu().numpy(), ptr)
    assert rel_err(res.rows.cpu().numpy(), ref) < TOL
    monkeypatch.delenv("S3GRL_FORCE_ONEHOP", raising=False)
    f.close(), G.close()


@pytest.mark.parametrize("seed", [1, 2, 3, 4])
@pytest.mark.parametrize("K", [2, 4, 8])
def test_tiny_links_fuzz_vs_c(eng, monkeypatch, seed, K):
    """Random sparse graphs of mixed shape (isolated nodes, leaves, a few denser nodes, self-loops), links between
    adjacent and non-adjacent nodes, endpoints without neighbours, reversed duplicates: node lists and row nodes
    exactly, rows within the tolerance, every sign_k instantiation of the kernel that the other tests leave out."""
    from oracle import c_oracle
    from s3grl_amd import workloads
    import scipy.sparse as sp

    rng = np.random.default_rng(100 + seed)
    n = int(rng.choice([200, 1500, 6000]))
    m = int(n * rng.choice([0.6, 1.5, 4.0]))
    e = rng.integers(0, n, size=(m, 2))
    if seed % 2:   # a few denser nodes
        hubs = rng.choice(n, 5, replace=False)
        e = np.concatenate([e, np.stack([rng.choice(hubs, 40 * 5), rng.integers(0, n, 40 * 5)], 1)])
    e = np.unique(np.sort(e, axis=1), axis=0)
    loops = e[e[:, 0] == e[:, 1]][:, 0]
    e = e[e[:, 0] != e[:, 1]]
    A = workloads.csr_from_undirected(n, e)
    if len(loops):
        A = (A + sp.diags((np.bincount(loops, minlength=n) > 0).astype(A.dtype))).tocsr()
        A.sort_indices()
    pos = e[rng.choice(len(e), min(len(e), 400), replace=False)]
    neg = rng.integers(0, n, size=(400, 2))
    links = np.concatenate([pos, neg, pos[:20, ::-1]])
    links = links[links[:, 0] != links[:, 1]]
    X = rng.standard_normal((n, int(rng.choice([5, 64, 128])))).astype(np.float32)
    monkeypatch.setenv("S3GRL_FORCE_ONEHOP", "1")
    G = eng.graph(A)
    f = eng.features(X)
    L = eng.links(links.T)
    counts = _class_counts(eng, monkeypatch, G, L, K)
    assert counts[26] + counts[27] > 0
    res = eng.precompute(G, f, L, mode="pos", num_hops=1, sign_k=K)
    ref, ptr, nodes, _ = c_oracle.pos_rows(links.T, 1, A, X, K, plus=False)
    assert np.array_equal(res.row_nodes.cpu().numpy(), nodes)
    assert np.array_equal(res.row_ptr.cpu().numpy(), ptr)
    assert rel_err(res.rows.cpu().numpy(), ref) < TOL
    monkeypatch.delenv("S3GRL_FORCE_ONEHOP", raising=False)
    f.close(), G.close()
