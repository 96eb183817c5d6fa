"""Oracle vs the REFERENCE-PINNED extraction fixtures (tests/golden/extract_*.npz, produced by
running the reference's own utils.k_hop_subgraph / utils.neighbors — see make_golden.py) and
vs its own committed fp64 diffusion vectors (regression pin)."""
import numpy as np
import pytest
import scipy.sparse as ssp

import oracle
from conftest import (DIFFUSION_NAMES, DIRECTED_NAMES, EXTRACT_NAMES, SAMPLED_NAMES, csr_from_arcs,
                      csr_from_undirected, load_diffusion, load_extract, load_extract_directed, load_sampled)


def _ragged(blob, key, i):
    off = blob[key + "_off"]
    return blob[key][off[i]:off[i + 1]]


@pytest.mark.parametrize("name", EXTRACT_NAMES)
def test_extraction_matches_reference(name):
    g = load_extract(name)
    n = int(g["num_nodes"])
    A = csr_from_undirected(n, g["edges"])
    for h in g["hops"]:
        for li, (s, d) in enumerate(g["links"]):
            nodes, sub, dists, _, _ = oracle.k_hop_subgraph(int(s), int(d), int(h), A)
            assert nodes[0] == s and nodes[1] == d
            order = np.lexsort((np.asarray(nodes), np.asarray(dists)))
            # node-index sets per hop: bit-exact
            np.testing.assert_array_equal(np.asarray(nodes)[order], _ragged(g, f"h{h}_nodes", li))
            np.testing.assert_array_equal(np.asarray(dists)[order], _ragged(g, f"h{h}_dists", li))
            # masked induced matrix incl. explicit zeros (K1, K2)
            sub = ssp.csr_matrix(sub)
            r = np.repeat(np.arange(sub.shape[0]), np.diff(sub.indptr))
            trip = np.stack([np.asarray(nodes)[r], np.asarray(nodes)[sub.indices],
                             sub.data.astype(np.int64)], axis=1)
            trip = trip[np.lexsort((trip[:, 1], trip[:, 0]))]
            np.testing.assert_array_equal(trip, _ragged(g, f"h{h}_sub", li))
            # PoS Plus row selection (K4)
            cn = oracle.neighbors({0}, sub) & oracle.neighbors({1}, sub)
            np.testing.assert_array_equal(sorted(nodes[a] for a in cn), _ragged(g, f"h{h}_cn", li))


@pytest.mark.parametrize("name", DIRECTED_NAMES)
def test_directed_extraction_matches_reference(name):
    """The directed branch (utils.py:58-63: out-neighbours through A, in-neighbours through A_csc),
    against what the reference's own k_hop_subgraph(directed=True) produced."""
    g = load_extract_directed(name)
    n = int(g["num_nodes"])
    A = csr_from_arcs(n, g["arcs"])
    A_csc = A.tocsc()
    assert (A != A.T).nnz > 0
    for h in g["hops"]:
        for li, (s, d) in enumerate(g["links"]):
            nodes, sub, dists, _, _ = oracle.k_hop_subgraph(int(s), int(d), int(h), A, directed=True, A_csc=A_csc)
            order = np.lexsort((np.asarray(nodes), np.asarray(dists)))
            np.testing.assert_array_equal(np.asarray(nodes)[order], _ragged(g, f"h{h}_nodes", li))
            np.testing.assert_array_equal(np.asarray(dists)[order], _ragged(g, f"h{h}_dists", li))
            sub = ssp.csr_matrix(sub)
            r = np.repeat(np.arange(sub.shape[0]), np.diff(sub.indptr))
            trip = np.stack([np.asarray(nodes)[r], np.asarray(nodes)[sub.indices],
                             sub.data.astype(np.int64)], axis=1)
            trip = trip[np.lexsort((trip[:, 1], trip[:, 0]))]
            np.testing.assert_array_equal(trip, _ragged(g, f"h{h}_sub", li))
            cn = oracle.neighbors({0}, sub) & oracle.neighbors({1}, sub)
            np.testing.assert_array_equal(sorted(nodes[a] for a in cn), _ragged(g, f"h{h}_cn", li))


@pytest.mark.parametrize("name", SAMPLED_NAMES)
def test_sampled_extraction_matches_reference(name):
    """Per-hop sampling (utils.py:62-74) as the reference's own k_hop_subgraph executes it with the
    keyed draw in place of random.sample (tests/golden/make_golden.py:make_sampled)."""
    g = load_sampled(name)
    A = csr_from_undirected(int(g["num_nodes"]), g["edges"])
    h, seed = int(g["num_hops"]), int(g["seed"])
    shrunk = 0
    for si, (ratio, cap) in enumerate(zip(g["ratio"], g["max_nodes"])):
        cap = None if cap < 0 else int(cap)
        for li, (s, d) in enumerate(g["links"]):
            nodes, sub, dists, _, _ = oracle.k_hop_subgraph(
                int(s), int(d), h, A, sample_ratio=float(ratio), max_nodes_per_hop=cap,
                sampler=oracle.hash_sampler(seed, int(s), int(d)))
            order = np.lexsort((np.asarray(nodes), np.asarray(dists)))
            np.testing.assert_array_equal(np.asarray(nodes)[order], _ragged(g, f"s{si}_nodes", li))
            np.testing.assert_array_equal(np.asarray(dists)[order], _ragged(g, f"s{si}_dists", li))
            cn = oracle.neighbors({0}, ssp.csr_matrix(sub)) & oracle.neighbors({1}, ssp.csr_matrix(sub))
            np.testing.assert_array_equal(sorted(nodes[a] for a in cn), _ragged(g, f"s{si}_cn", li))
            full = oracle.k_hop_subgraph(int(s), int(d), h, A)[0]
            assert set(nodes) <= set(full)
            shrunk += len(nodes) < len(full)
    assert shrunk > 0


def test_sampling_key_is_symmetric_and_default_sampler_is_random_sample():
    assert oracle.hop_sample_key(5, 3, 9, 77) & 0xffffffff == 77
    a = oracle.hash_sampler(5, 3, 9)(range(100), 10)
    b = oracle.hash_sampler(5, 9, 3)(range(100), 10)
    assert a == b and len(set(a)) == 10 and a != oracle.hash_sampler(6, 3, 9)(range(100), 10)
    g = load_extract("usair")
    A = csr_from_undirected(int(g["num_nodes"]), g["edges"])
    s, d = (int(v) for v in g["links"][0])
    import random

    random.seed(0)
    n1 = oracle.k_hop_subgraph(s, d, 2, A, sample_ratio=0.5)[0]
    full = oracle.k_hop_subgraph(s, d, 2, A)[0]
    assert set(n1) < set(full) and n1[:2] == [s, d]


def test_set_order_variant_same_sets():
    g = load_extract("rand300")
    A = csr_from_undirected(int(g["num_nodes"]), g["edges"])
    s, d = (int(v) for v in g["links"][0])
    a = oracle.k_hop_subgraph(s, d, 2, A, order="canonical")
    b = oracle.k_hop_subgraph(s, d, 2, A, order="set")
    assert sorted(zip(a[2], a[0])) == sorted(zip(b[2], b[0]))


@pytest.mark.parametrize("name", DIFFUSION_NAMES)
def test_diffusion_regression(name):
    g = load_diffusion(name)
    n, K, h = int(g["num_nodes"]), int(g["K"]), int(g["num_hops"])
    A = csr_from_undirected(n, g["edges"])
    li = g["links"].T
    kw = {"sign_k": K, "k_node_set_strategy": "intersection"}
    X = g["X"]
    for tag, lst in [
        ("pos", oracle.get_PoS_prepped_ds(li, h, A, X, 1, kw, dtype=np.float64)),
        ("plus", oracle.get_PoS_Plus_prepped_ds(li, h, A, X, 1, kw, dtype=np.float64)),
        ("sop", oracle.get_SoP_prepped_ds(oracle.global_normalized_powers(A, K, np.float64), li, A,
                                          X, 1, dtype=np.float64)),
    ]:
        rows, row_ptr, _ = oracle.collate_rows(lst, K)
        np.testing.assert_array_equal(row_ptr, g[f"{tag}_row_ptr"])
        np.testing.assert_allclose(rows, g[f"{tag}_rows"], rtol=1e-12, atol=1e-14)
        # fp32 mode of the oracle (the reference's precision) stays inside the 1e-5 band
        if tag == "pos":
            lst32 = oracle.get_PoS_prepped_ds(li, h, A, X.astype(np.float32), 1, kw,
                                              dtype=np.float32)
            r32, _, _ = oracle.collate_rows(lst32, K)
            scale = np.abs(rows).max(axis=-1, keepdims=True) + 1e-30
            assert np.max(np.abs(r32 - rows) / scale) < 1e-5
