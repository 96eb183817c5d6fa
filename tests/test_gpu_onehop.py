"""-m gpu: one-hop plans on the row-intersection path (count1_kernel + link_full_kernel,
csrc/s3grl_onehop.inl) — forced here on the small fixture graphs (S3GRL_FORCE_ONEHOP; big graphs
take it by themselves) and compared with the bitmap flavour of the same plan, with the
reference-pinned extraction fixtures and with the fp64 oracle."""
import numpy as np
import pytest

import oracle
from conftest import EXTRACT_NAMES, csr_from_undirected, load_extract
from test_gpu_parity import TOL, _ragged, rel_err

pytestmark = pytest.mark.gpu


@pytest.fixture()
def eng():
    import torch
    from s3grl_amd.engine import Engine

    assert torch.cuda.is_available()
    e = Engine("cuda:0")
    yield e
    e.close()


def _plans(eng, monkeypatch, A, links, mode, K, bm_hbm=False):
    """(bitmap-flavour plan, one-hop-path plan) of the same one-hop request."""
    for k in ("S3GRL_FORCE_ONEHOP", "S3GRL_FORCE_HASH", "S3GRL_FORCE_BM_HBM", "S3GRL_BIG_COLS_HBM", "S3GRL_HUB_MIN_DEG",
              "S3GRL_FORCE_HUB_SLICES", "S3GRL_HUB_COLS_HBM"):
        monkeypatch.delenv(k, raising=False)
    G0 = eng.graph(A)
    p0 = eng.plan(G0, links, mode=mode, num_hops=1, sign_k=K, full_stats=True)
    monkeypatch.setenv("S3GRL_FORCE_ONEHOP", "1")
    monkeypatch.setenv("S3GRL_FORCE_HASH", "1")
    if bm_hbm in ("hub", "hubx", "hubxx"):     # cached hub neighbourhoods (s3grl_hub.hip): every node of 3+ neighbours
        monkeypatch.setenv("S3GRL_HUB_MIN_DEG", "3")
        if bm_hbm != "hub":                    # ... the class with its found edges in HBM slices
            monkeypatch.setenv("S3GRL_FORCE_HUB_SLICES", "1")
        if bm_hbm == "hubxx":                  # ... and the small CSR's columns there too
            monkeypatch.setenv("S3GRL_HUB_COLS_HBM", "1")
    elif bm_hbm:                               # the class of the biggest subgraphs (edge list + sort)
        monkeypatch.setenv("S3GRL_FORCE_BM_HBM", "1")
    if bm_hbm == "hbm":                        # ... with its CSR columns in the HBM slice
        monkeypatch.setenv("S3GRL_BIG_COLS_HBM", "1")
    G1 = eng.graph(A)                      # the oriented rows are built with the graph
    p1 = eng.plan(G1, links, mode=mode, num_hops=1, sign_k=K, full_stats=True)
    for k in ("S3GRL_FORCE_ONEHOP", "S3GRL_FORCE_HASH", "S3GRL_FORCE_BM_HBM", "S3GRL_BIG_COLS_HBM", "S3GRL_HUB_MIN_DEG",
              "S3GRL_FORCE_HUB_SLICES", "S3GRL_HUB_COLS_HBM"):
        monkeypatch.delenv(k, raising=False)
    return (G0, p0), (G1, p1)


@pytest.mark.parametrize("name", EXTRACT_NAMES)
@pytest.mark.parametrize("K,bm_hbm", [(1, False), (2, "hbm"), (3, "lds"), (5, False), (2, "hub"), (3, "hub"), (5, "hub"),
                                      (3, "hubx"), (2, "hubxx")])
def test_onehop_path_equals_bitmap_flavour(eng, monkeypatch, name, K, bm_hbm):
    import torch

    g = load_extract(name)
    n = int(g["num_nodes"])
    A = csr_from_undirected(n, g["edges"])
    X = np.random.default_rng(17).standard_normal((n, 19)).astype(np.float32)
    f = eng.features(X)
    links = eng.links(g["links"].T)
    for mode in ("pos", "pos_plus"):
        (G0, p0), (G1, p1) = _plans(eng, monkeypatch, A, links, mode, K, bm_hbm)
        r0, r1 = p0.run(f), p1.run(f)
        assert torch.equal(p0.row_ptr(), p1.row_ptr()) and torch.equal(p0.row_nodes(), p1.row_nodes())
        for a, b in zip(p0.export_subgraphs(), p1.export_subgraphs()):
            assert torch.equal(a, b)                      # node lists, canonical order, hop distances
        s0, s1 = dict(p0.stats), dict(p1.stats)
        s0.pop("workspace_bytes"), s1.pop("workspace_bytes")
        assert s0.pop("hub_links") == 0 and s0.pop("hub_read_bytes") == 0 and s0.pop("hub_endpoint_entries") == 0 and s0.pop("hub_nodes") == 0
        hub_links = s1.pop("hub_links")
        assert (hub_links > 0) == (s1.pop("hub_read_bytes") > 0) == (s1.pop("hub_endpoint_entries") > 0) == (s1.pop("hub_nodes") > 0)
        # one-hop-path figures: rows probed by link_full_kernel, links served from a cached hub neighbourhood
        assert s0.pop("oriented_entries") == 0 and s1.pop("oriented_entries") + hub_links > 0
        if str(bm_hbm).startswith("hub") and K >= 2 and name in ("usair", "cora", "rand300"):
            assert hub_links > 0                          # the cached neighbourhoods were used
        assert s0 == s1                                   # n, vol(S), induced edges, support: exact
        # two summation orders of the same fp32 sums (the multi-hop path walks the degree order)
        assert rel_err(r1.cpu().numpy(), r0.cpu().numpy()) < 3e-6
        # bit-reproducible against itself
        r1b = p1.run(f)
        assert torch.equal(r1, r1b)
        p0.close(), p1.close(), G0.close(), G1.close()


@pytest.mark.parametrize("name", ["usair", "rand300", "star_iso"])
@pytest.mark.parametrize("bm_hbm", [False, "lds", "hub"])
def test_onehop_path_degree_order_is_invisible(eng, monkeypatch, name, bm_hbm):
    """The one-hop path walks the graph's degree order too (csrc/s3grl_relabel.hip); with
    S3GRL_NO_RELABEL it walks the caller's order.  Same node lists, rows nodes and statistics."""
    import torch

    g = load_extract(name)
    n = int(g["num_nodes"])
    A = csr_from_undirected(n, g["edges"])
    f = eng.features(np.random.default_rng(3).standard_normal((n, 11)).astype(np.float32))
    links = eng.links(np.concatenate([g["links"], g["links"][:3, ::-1]]).T)
    outs = []
    for off in (False, True):
        if off:
            monkeypatch.setenv("S3GRL_NO_RELABEL", "1")
        (G0, p0), (G1, p1) = _plans(eng, monkeypatch, A, links, "pos_plus", 3, bm_hbm)
        st = dict(p1.stats)
        st.pop("workspace_bytes")
        st.pop("oriented_entries")        # ties of the (degree, id) orientation follow the id order walked
        st.pop("hub_links"), st.pop("hub_read_bytes"), st.pop("hub_endpoint_entries"), st.pop("hub_nodes")   # (the cache belongs to the degree order)
        outs.append((p1.run(f).clone(), p1.row_ptr().clone(), p1.row_nodes().clone(),
                     [t.clone() for t in p1.export_subgraphs()], st))
        p0.close(), p1.close(), G0.close(), G1.close()
    monkeypatch.delenv("S3GRL_NO_RELABEL", raising=False)
    a, b = outs
    assert torch.equal(a[1], b[1]) and torch.equal(a[2], b[2]) and a[4] == b[4]
    assert all(torch.equal(x, y) for x, y in zip(a[3], b[3]))
    assert rel_err(a[0].cpu().numpy(), b[0].cpu().numpy()) < 3e-6


@pytest.mark.parametrize("flavour", [False, "hub"])
@pytest.mark.parametrize("name", ["usair", "rand300", "probe5"])
def test_onehop_path_node_sets_vs_reference_fixture(eng, monkeypatch, name, flavour):
    """Hop-1 node sets and CN rows against what the reference's own k_hop_subgraph produced."""
    g = load_extract(name)
    if 1 not in [int(h) for h in g["hops"]]:
        pytest.skip("fixture has no 1-hop case")
    n = int(g["num_nodes"])
    A = csr_from_undirected(n, g["edges"])
    links = eng.links(g["links"].T)
    (G0, p0), (G1, p1) = _plans(eng, monkeypatch, A, links, "pos_plus", 3, flavour)
    p0.close(), G0.close()
    node_ptr, nodes, dists = (t.cpu().numpy() for t in p1.export_subgraphs())
    row_ptr, row_nodes = p1.row_ptr().cpu().numpy(), p1.row_nodes().cpu().numpy()
    for li, (s, d) in enumerate(g["links"]):
        np.testing.assert_array_equal(nodes[node_ptr[li]:node_ptr[li + 1]], _ragged(g, "h1_nodes", li))
        np.testing.assert_array_equal(dists[node_ptr[li]:node_ptr[li + 1]], _ragged(g, "h1_dists", li))
        rn = row_nodes[row_ptr[li]:row_ptr[li + 1]]
        assert rn[0] == s and rn[1] == d
        np.testing.assert_array_equal(rn[2:], _ragged(g, "h1_cn", li))
    exp_e = sum(int((_ragged(g, "h1_sub", li)[:, 2] != 0).sum()) for li in range(len(g["links"])))
    assert p1.stats["total_sub_edges"] == exp_e
    p1.close(), G1.close()


@pytest.mark.parametrize("mode", ["pos", "pos_plus"])
@pytest.mark.parametrize("bm_hbm", [False, "lds", "hbm", "hub", "hubx", "hubxx"])
def test_onehop_path_vs_oracle_with_self_loops_and_hubs(eng, monkeypatch, mode, bm_hbm):
    """A graph with self-loops at link endpoints and at common neighbours, a hub adjacent to
    everything, isolated endpoints, and links in both directions (folded) — against the oracle."""
    import scipy.sparse as ssp

    rng = np.random.default_rng(23)
    n = 160
    e = rng.integers(0, n, size=(420, 2))
    e = e[e[:, 0] != e[:, 1]]
    hub = np.stack([np.full(n - 2, 5), np.array([v for v in range(n - 1) if v != 5])], 1)   # n-1 isolated
    loops = np.array([[3, 3], [4, 4], [5, 5], [9, 9], [20, 20]])
    e = np.vstack([e[(e != n - 1).all(1)], hub, loops, [[3, 4], [3, 9], [4, 9]]])
    r = np.concatenate([e[:, 0], e[:, 1]])
    c = np.concatenate([e[:, 1], e[:, 0]])
    A = ssp.csr_matrix((np.ones(len(r), dtype=np.int64), (r, c)), shape=(n, n))
    A.sum_duplicates()
    A.data[:] = 1
    X = rng.random((n, 11)).astype(np.float32)
    links = np.array([[3, 4], [4, 3], [3, 5], [5, 20], [20, 5], [7, n - 1], [n - 1, 30], [40, 41], [9, 3],
                      [5, 4], [60, 61], [61, 60], [8, 5]]).T
    K = 3
    kw = {"sign_k": K, "k_node_set_strategy": "intersection"}
    fn = oracle.get_PoS_Plus_prepped_ds if mode == "pos_plus" else oracle.get_PoS_prepped_ds
    ref, ref_ptr, _ = oracle.collate_rows(fn(links, 1, A, X.astype(np.float64), 1, kw, dtype=np.float64), K)
    (G0, p0), (G1, p1) = _plans(eng, monkeypatch, A, eng.links(links), mode, K, bm_hbm)
    p0.close(), G0.close()
    monkeypatch.setenv("S3GRL_FORCE_ONEHOP", "1")
    monkeypatch.setenv("S3GRL_FORCE_HASH", "1")
    if bm_hbm:
        monkeypatch.setenv("S3GRL_FORCE_BM_HBM", "1")
    if bm_hbm == "hbm":
        monkeypatch.setenv("S3GRL_BIG_COLS_HBM", "1")
    if bm_hbm in ("hubx", "hubxx"):
        monkeypatch.setenv("S3GRL_FORCE_HUB_SLICES", "1")
    if bm_hbm == "hubxx":
        monkeypatch.setenv("S3GRL_HUB_COLS_HBM", "1")
    pf = eng.plan(G1, eng.links(links), mode=mode, num_hops=1, sign_k=K)       # with folding
    for p in (p1, pf):
        rows = p.run(eng.features(X)).cpu().numpy()
        assert np.array_equal(p.row_ptr().cpu().numpy(), ref_ptr)
        assert rel_err(rows, ref) < TOL
    assert pf.stats["folded_links"] == 3
    p1.close(), pf.close(), G1.close()


def test_onehop_plans_have_no_node_limit(eng):
    """A graph beyond the LDS-bitmap limit (327 680 nodes): one-hop PoS / PoS Plus run on the
    row-intersection path (the setting the reference uses on its large graphs); multi-hop plans run
    too, with their bitmaps in HBM (tests/test_gpu_parity.py::test_half_million_nodes_two_hops_vs_c
    checks their rows); a hub-hub link beyond the on-chip one-hop classes takes that road as well."""
    from s3grl_amd import workloads

    n, e = workloads.chung_lu(400000, 1200000, seed=21)
    A = workloads.csr_from_undirected(n, e)
    rng = np.random.default_rng(22)
    X = rng.random((n, 8)).astype(np.float32)
    deg = np.bincount(e.ravel(), minlength=n)
    hubs = np.argsort(-deg)[:4]
    pos = e[rng.choice(len(e), 300, replace=False)]
    neg = rng.integers(0, n, size=(300, 2))
    neg = neg[neg[:, 0] != neg[:, 1]]
    # (a link between the two biggest hubs, ~5 000 nodes, is beyond the on-chip one-hop classes and
    # would need the LDS bitmaps of the general path: see DESIGN §3 "limits")
    links = np.concatenate([pos, neg, [[hubs[3], int(pos[1, 0])], [hubs[2], int(pos[0, 0])]]]).T
    G = eng.graph(A)
    kw = {"sign_k": 3, "k_node_set_strategy": "intersection"}
    for mode, fn in (("pos", oracle.get_PoS_prepped_ds), ("pos_plus", oracle.get_PoS_Plus_prepped_ds)):
        res = eng.precompute(G, eng.features(X), eng.links(links), mode=mode, num_hops=1, sign_k=3)
        sel = np.concatenate([np.arange(0, 600, 12), [links.shape[1] - 2, links.shape[1] - 1]])
        ref, ref_ptr, _ = oracle.collate_rows(fn(links[:, sel], 1, A, X.astype(np.float64), 1, kw,
                                                 dtype=np.float64), 3)
        ptr = res.row_ptr.cpu().numpy()
        take = np.concatenate([np.arange(ptr[l], ptr[l + 1]) for l in sel])
        assert np.array_equal(np.diff(ptr)[sel], np.diff(ref_ptr))
        assert rel_err(res.rows.cpu().numpy()[take], ref) < TOL
    # a link between the two biggest hubs (~5 000 nodes) does not fit the on-chip one-hop classes: it
    # falls back to the general path, whose bitmaps live in HBM on a graph of this size
    from oracle import c_oracle

    big = np.array([[hubs[0], hubs[1]], [int(pos[2, 0]), int(pos[2, 1])]]).T
    res = eng.precompute(G, eng.features(X), eng.links(big), mode="pos_plus", num_hops=1, sign_k=3)
    ref, ptr, nodes, _ = c_oracle.pos_rows(big, 1, A, X, 3, plus=True)
    assert np.array_equal(res.row_ptr.cpu().numpy(), ptr) and np.array_equal(res.row_nodes.cpu().numpy(), nodes)
    assert rel_err(res.rows.cpu().numpy(), ref) < TOL
    p2 = eng.plan(G, eng.links(links[:, :40]), mode="pos", num_hops=2, sign_k=3)     # multi-hop: no refusal
    assert p2.stats["max_nodes"] > 2
    p2.close()
    G.close()


@pytest.mark.parametrize("mode", ["pos", "pos_plus"])
def test_cached_hub_neighbourhoods_equal_the_per_link_road(eng, monkeypatch, mode):
    """A power-law graph with real hubs (csrc/s3grl_hub.hip): links at hubs take link_hub_kernel — the
    hub's induced neighbourhood from the per-graph cache, only the other endpoint's rows walked — and
    must give what link_full_kernel gives with the cache switched off: node lists, row nodes and
    statistics exactly, rows to fp32 round-off (another fixed summation order); both directions of a
    link, folded and unfolded, bit for bit the same; hub-hub links and leaves included."""
    import torch
    from s3grl_amd import workloads

    n, e = workloads.chung_lu(30000, 150000, seed=5)
    A = workloads.csr_from_undirected(n, e)
    deg = np.diff(A.indptr)
    hubs = np.argsort(-deg)[:12]
    assert deg[hubs[-1]] >= 64
    rng = np.random.default_rng(2)
    links = []
    for h in hubs:
        nb = A.indices[A.indptr[h]:A.indptr[h + 1]]
        links += [(h, int(v)) for v in rng.choice(nb, 6, replace=False)]          # positive links at the hub
        links += [(int(v), h) for v in rng.integers(0, n, 6) if v != h]           # negative ones, hub as dst
    links += [(int(hubs[0]), int(hubs[1])), (int(hubs[2]), int(hubs[5]))]        # hub - hub
    links = np.array(links)
    links = np.concatenate([links, links[:10, ::-1]])                             # + reversed duplicates
    X = rng.standard_normal((n, 24)).astype(np.float32)
    out = []
    monkeypatch.setenv("S3GRL_FORCE_ONEHOP", "1")
    monkeypatch.setenv("S3GRL_FORCE_HASH", "1")
    for cache in (False, True):
        monkeypatch.setenv("S3GRL_HUB_MIN_DEG", "64")
        if not cache:
            monkeypatch.setenv("S3GRL_NO_HUB_CACHE", "1")
        else:
            monkeypatch.delenv("S3GRL_NO_HUB_CACHE", raising=False)
        G = eng.graph(A)
        f = eng.features(X)
        L = eng.links(links.T)
        p = eng.plan(G, L, mode=mode, num_hops=1, sign_k=3, full_stats=True)
        rows = p.run(f).clone()
        st = dict(p.stats)
        st.pop("workspace_bytes")
        assert (st.pop("hub_links") > len(links) // 2) == cache and (st.pop("hub_read_bytes") > 0) == cache
        st.pop("hub_endpoint_entries"), st.pop("oriented_entries"), st.pop("hub_nodes")
        exp = [t.clone() for t in p.export_subgraphs()]
        pf = eng.plan(G, L, mode=mode, num_hops=1, sign_k=3)                      # folded
        assert pf.stats["folded_links"] == 10
        assert torch.equal(pf.run(f), rows)
        out.append((rows, st, exp, p.row_ptr().clone(), p.row_nodes().clone()))
        p.close(), pf.close(), f.close(), G.close()
    (ra, sa, ea, pa, na), (rb, sb, eb, pb, nb_) = out
    assert sa == sb and torch.equal(pa, pb) and torch.equal(na, nb_)
    assert all(torch.equal(x, y) for x, y in zip(ea, eb))
    assert not torch.equal(ra, rb)                        # (another summation order: the cache WAS used)
    # (fp32 round-off of two summation orders; the narrow gather sums a list front to back, and the hub - hub
    # links of this list — thousands of nodes — take the cached road too since the volume cap was raised)
    assert rel_err(rb.cpu().numpy(), ra.cpu().numpy()) < 8e-6
    from oracle import c_oracle

    ref, ptr, _, _ = c_oracle.pos_rows(links.T, 1, A, X, 3, plus=mode == "pos_plus")
    assert rel_err(rb.cpu().numpy(), ref) < TOL


def test_no_hub_cache_on_a_graph_whose_hubs_are_not_rare(eng, monkeypatch):
    """Mean degree in the hundreds: every node would be a "hub", the cache of all those dense
    neighbourhoods would cost more than 2^30 neighbour tests to build — it is skipped and every link
    stays on link_full_kernel (hub_links == 0), results against the C restatement."""
    from oracle import c_oracle
    from s3grl_amd import workloads

    rng = np.random.default_rng(4)
    n, m = 8000, 1_700_000
    e = rng.integers(0, n, size=(m, 2))
    e = e[e[:, 0] != e[:, 1]]
    A = workloads.csr_from_undirected(n, np.unique(np.sort(e, axis=1), axis=0))
    assert np.diff(A.indptr).min() >= 256
    links = rng.integers(0, n, size=(12, 2))
    links = links[links[:, 0] != links[:, 1]]
    X = rng.standard_normal((n, 8)).astype(np.float32)
    monkeypatch.setenv("S3GRL_FORCE_ONEHOP", "1")
    monkeypatch.setenv("S3GRL_FORCE_HASH", "1")
    G = eng.graph(A)
    p = eng.plan(G, eng.links(links.T), mode="pos", num_hops=1, sign_k=2)
    rows = p.run(eng.features(X)).cpu().numpy()
    assert p.stats["hub_links"] == 0 and p.stats["oriented_entries"] > 0
    ref, ptr, _, _ = c_oracle.pos_rows(links.T, 1, A, X, 2)
    assert rel_err(rows, ref) < TOL
    p.close(), G.close()
