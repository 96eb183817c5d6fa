"""The induced-CSR flavour of the per-link kernel (csrc/s3grl_csr.hip): plans whose every operator reaches
the whole subgraph (sign_k - 1 >= num_hops) build the masked induced adjacency once per link in LDS and
run all K operators as pulls over it.  What the reference computes there: tuned_SIGN.py:153-175 (the
normalised induced adjacency, its powers, rows {src, dst} + common neighbours).

S3GRL_FORCE_CSR sends the small fixture graphs down that road (by default it is for graphs of more than
8 192 nodes: PubMed); S3GRL_NO_CSR keeps the bitmap / direct-map flavours."""
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


def _flavours(monkeypatch, run):
    out = []
    for force in (False, True):
        monkeypatch.delenv("S3GRL_NO_CSR", raising=False)
        monkeypatch.delenv("S3GRL_FORCE_CSR", raising=False)
        monkeypatch.setenv("S3GRL_FORCE_CSR" if force else "S3GRL_NO_CSR", "1")
        out.append(run())
    monkeypatch.delenv("S3GRL_NO_CSR", raising=False)
    monkeypatch.delenv("S3GRL_FORCE_CSR", raising=False)
    return out


@pytest.mark.parametrize("name,hops,K", [("rand300", 2, 3), ("rand300", 2, 5), ("rand300", 1, 2), ("cora", 3, 4),
                                          ("cora", 2, 5), ("usair", 1, 3), ("usair", 2, 3), ("star_iso", 2, 4),
                                          ("probe5", 3, 5), ("triangle", 1, 2), ("pair", 2, 3)])
@pytest.mark.parametrize("mode", ["pos", "pos_plus"])
def test_csr_flavour_equals_the_walking_flavours(eng, monkeypatch, name, hops, K, mode):
    """Everything integer a plan hands out — row pointers, row nodes, exported node lists, distances,
    statistics incl. the exact edge count — equal exactly with the flavour on and off; rows to fp32
    round-off (another fixed summation order) and against the oracle."""
    import torch

    g = load_extract(name)
    n = int(g["num_nodes"])
    A = csr_from_undirected(n, g["edges"])
    X = np.random.default_rng(5).random((n, 21))
    links = np.concatenate([g["links"], g["links"][:4, ::-1]])   # reversed duplicates are folded
    G = eng.graph(A)
    f = eng.features(X)
    L = eng.links(links.T)

    def run():
        plan = eng.plan(G, L, mode=mode, num_hops=hops, sign_k=K, full_stats=True)
        exp = [t.clone() for t in plan.export_subgraphs()]
        st = dict(plan.stats)
        st.pop("workspace_bytes", None)
        plan.close()
        res = eng.precompute(G, f, L, mode=mode, num_hops=hops, sign_k=K)
        return exp, st, res.rows.clone(), res.row_ptr.clone(), res.row_nodes.clone(), dict(res.stats)

    (ea, sa, ra, pa, na, fa), (eb, sb, rb, pb, nb, fb) = _flavours(monkeypatch, run)
    assert all(torch.equal(x, y) for x, y in zip(ea, eb))
    assert sa == sb
    assert fa["total_sub_edges"] == fb["total_sub_edges"] and fa["total_volume"] == fb["total_volume"]
    assert fa["total_support"] == fb["total_support"]
    assert torch.equal(pa, pb) and torch.equal(na, nb)
    assert rel_err(rb.cpu().numpy(), ra.cpu().numpy()) < 3e-6
    kw = {"sign_k": K, "k_node_set_strategy": "intersection"}
    fn = oracle.get_PoS_prepped_ds if mode == "pos" else oracle.get_PoS_Plus_prepped_ds
    ref, ref_ptr, _ = oracle.collate_rows(fn(links.T, hops, A, X, 1, kw, dtype=np.float64), K)
    assert np.array_equal(pb.cpu().numpy(), ref_ptr)
    # common-neighbour rows: the oracle's order inside a link is ascending id too
    assert rel_err(rb.cpu().numpy(), ref) < TOL
    f.close(), G.close()


def test_csr_flavour_is_taken(eng, monkeypatch):
    """The hook really switches flavours (a silently ignored switch would make the test above vacuous):
    with S3GRL_DEBUG the plan prints its class counts; the induced-CSR classes are lists 32..45."""
    import ctypes
    import io
    import os
    import re
    import tempfile

    g = load_extract("cora")
    n = int(g["num_nodes"])
    A = csr_from_undirected(n, g["edges"])
    G = eng.graph(A)
    L = eng.links(g["links"].T)
    counts = []
    for force in (False, True):
        monkeypatch.delenv("S3GRL_NO_CSR", raising=False)
        monkeypatch.delenv("S3GRL_FORCE_CSR", raising=False)
        monkeypatch.setenv("S3GRL_FORCE_CSR" if force else "S3GRL_NO_CSR", "1")
        monkeypatch.setenv("S3GRL_DEBUG", "1")
        with tempfile.TemporaryFile(mode="w+b") as tmp:
            saved = os.dup(2)
            os.dup2(tmp.fileno(), 2)
            try:
                eng.plan(G, L, mode="pos", num_hops=2, sign_k=4).close()
            finally:
                os.dup2(saved, 2)
                os.close(saved)
            tmp.seek(0)
            text = tmp.read().decode()
        m = re.search(r"classes:((?: -?\d+)+)", text)
        assert m, text
        counts.append([int(x) for x in m.group(1).split()])
    monkeypatch.delenv("S3GRL_DEBUG", raising=False)
    monkeypatch.delenv("S3GRL_FORCE_CSR", raising=False)
    off, on = counts
    assert sum(off[32:46]) == 0
    assert sum(on[32:46]) == len(g["links"]) and sum(on[:26]) == 0
    G.close()


@pytest.mark.parametrize("feat", ["dense", "packed"])
def test_csr_flavour_with_split_jobs(eng, monkeypatch, feat):
    """Lists longer than the split threshold are laid out piece by piece (kSplitThreshold): the induced-CSR
    flavour writes the same layout; S3GRL_SPLIT_T = 64 splits Cora's two-hop subgraphs."""
    g = load_extract("cora")
    n = int(g["num_nodes"])
    A = csr_from_undirected(n, g["edges"])
    rng = np.random.default_rng(3)
    X = rng.random((n, 40)) * (rng.random((n, 40)) < 0.3)
    links = g["links"]
    G = eng.graph(A)
    f = eng.features(X, mode=feat)
    L = eng.links(links.T)
    monkeypatch.setenv("S3GRL_SPLIT_T", "64")
    monkeypatch.setenv("S3GRL_SPLIT_SEG_SHIFT", "5")

    def run():
        return eng.precompute(G, f, L, mode="pos_plus", num_hops=2, sign_k=4).rows.clone()

    a, b = _flavours(monkeypatch, run)
    monkeypatch.delenv("S3GRL_SPLIT_T", raising=False)
    monkeypatch.delenv("S3GRL_SPLIT_SEG_SHIFT", raising=False)
    assert rel_err(b.cpu().numpy(), a.cpu().numpy()) < 3e-6
    kw = {"sign_k": 4, "k_node_set_strategy": "intersection"}
    ref, _, _ = oracle.collate_rows(oracle.get_PoS_Plus_prepped_ds(links.T, 2, A, X, 1, kw, dtype=np.float64), 4)
    assert rel_err(b.cpu().numpy(), ref) < TOL
    f.close(), G.close()


def test_csr_rows_do_not_depend_on_the_plan(eng, monkeypatch):
    """A link's rows are the same bits whatever list it is computed in (a sharded run must reproduce the
    unsharded one): the whole list twice, its halves, and a permutation."""
    import torch

    monkeypatch.setenv("S3GRL_FORCE_CSR", "1")
    g = load_extract("cora")
    n = int(g["num_nodes"])
    A = csr_from_undirected(n, g["edges"])
    X = np.random.default_rng(7).random((n, 19))
    links = np.concatenate([g["links"], g["links"][::3, ::-1]])
    G = eng.graph(A)
    f = eng.features(X)
    whole = eng.precompute(G, f, eng.links(links.T), mode="pos", num_hops=3, sign_k=5).rows.clone()
    again = eng.precompute(G, f, eng.links(links.T), mode="pos", num_hops=3, sign_k=5).rows.clone()
    assert torch.equal(whole, again)
    h = len(links) // 2
    for sel in (slice(0, h), slice(h, None)):
        part = eng.precompute(G, f, eng.links(links[sel].T), mode="pos", num_hops=3, sign_k=5).rows
        idx = np.arange(len(links))[sel]
        assert torch.equal(part.view(-1, 2, *part.shape[1:]), whole.view(-1, 2, *whole.shape[1:])[idx])
    perm = np.random.default_rng(0).permutation(len(links))
    shuf = eng.precompute(G, f, eng.links(links[perm].T), mode="pos", num_hops=3, sign_k=5).rows
    assert torch.equal(shuf.view(-1, 2, *shuf.shape[1:]), whole.view(-1, 2, *whole.shape[1:])[perm])
    monkeypatch.delenv("S3GRL_FORCE_CSR", raising=False)
    f.close(), G.close()


def test_csr_flavour_on_a_mid_sized_graph_vs_c(eng):
    """The default switch (no hook): a 20 000-node sparse random graph, two hops, sign_k = 4, PoS and
    PoS Plus, against the C restatement; isolated endpoints and links inside one component included."""
    from oracle import c_oracle
    from s3grl_amd import workloads

    rng = np.random.default_rng(11)
    n = 20000
    e = rng.integers(0, n, size=(45000, 2))
    e = e[e[:, 0] != e[:, 1]]
    e = np.unique(np.sort(e, axis=1), axis=0)
    A = workloads.csr_from_undirected(n, e)
    links = np.concatenate([e[rng.choice(len(e), 600, replace=False)], rng.integers(0, n, size=(600, 2))])
    links = links[links[:, 0] != links[:, 1]]
    X = rng.random((n, 24)).astype(np.float32)
    G = eng.graph(A)
    f = eng.features(X)
    for mode in ("pos", "pos_plus"):
        res = eng.precompute(G, f, eng.links(links.T), mode=mode, num_hops=2, sign_k=4)
        ref, ptr, nodes, _ = c_oracle.pos_rows(links.T, 2, A, X, 4, plus=mode == "pos_plus")
        assert np.array_equal(res.row_nodes.cpu().numpy(), nodes)
        assert np.array_equal(res.row_ptr.cpu().numpy(), ptr)
        assert rel_err(res.rows.cpu().numpy(), ref) < TOL
    f.close(), G.close()
