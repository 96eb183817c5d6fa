"""The plain-C restatement (oracle/s3grl_oracle_c.c) against the reference-pinned extraction
fixtures, the Python oracle and the committed fp64 diffusion vectors.  Two independently written
restatements (SpGEMM + row select in scipy; sparse vector x matrix in C) agreeing to fp64
round-off is the cross-check of the "parity unpinned" diffusion half."""
import numpy as np
import pytest

import oracle
from conftest import DIFFUSION_NAMES, EXTRACT_NAMES, csr_from_undirected, load_diffusion, load_extract
from oracle import c_oracle


def _ragged(blob, key, i):
    off = blob[key + "_off"]
    return blob[key][off[i]:off[i + 1]]


@pytest.fixture(scope="module", autouse=True)
def _built():
    c_oracle.build()


@pytest.mark.parametrize("name", EXTRACT_NAMES)
def test_c_extraction_matches_reference(name):
    g = load_extract(name)
    A = csr_from_undirected(int(g["num_nodes"]), g["edges"])
    for h in g["hops"]:
        node_ptr, nodes, dists = c_oracle.extract(g["links"].T, int(h), A)
        for li in range(len(g["links"])):
            mine = nodes[node_ptr[li]:node_ptr[li + 1]]
            md = dists[node_ptr[li]:node_ptr[li + 1]]
            assert tuple(mine[:2]) == tuple(g["links"][li])
            order = np.lexsort((mine, md))
            np.testing.assert_array_equal(mine[order], _ragged(g, f"h{h}_nodes", li))
            np.testing.assert_array_equal(md[order], _ragged(g, f"h{h}_dists", li))


@pytest.mark.parametrize("name", DIFFUSION_NAMES)
def test_c_diffusion_vs_python_oracle_and_golden(name):
    g = load_diffusion(name)
    n, K, h = int(g["num_nodes"]), int(g["K"]), int(g["num_hops"])
    A = csr_from_undirected(n, g["edges"])
    li = g["links"].T
    X32 = g["X"].astype(np.float32)                     # the C entry takes fp32 X, like the engine
    kw = {"sign_k": K, "k_node_set_strategy": "intersection"}
    for tag, plus, fn in [("pos", False, oracle.get_PoS_prepped_ds),
                          ("plus", True, oracle.get_PoS_Plus_prepped_ds)]:
        rows, row_ptr, row_nodes, node_count = c_oracle.pos_rows(li, h, A, X32, K, plus=plus)
        lst = fn(li, h, A, X32.astype(np.float64), 1, kw, dtype=np.float64)
        ref, ref_ptr, _ = oracle.collate_rows(lst, K)
        np.testing.assert_array_equal(row_ptr, ref_ptr)
        np.testing.assert_array_equal(row_ptr, g[f"{tag}_row_ptr"])
        np.testing.assert_array_equal(row_nodes, g[f"{tag}_rows_global"])
        assert list(node_count) == [len(d["nodes"]) for d in lst]
        np.testing.assert_allclose(rows, ref, rtol=1e-12, atol=1e-14)
        # committed fp64 vectors were made from the fp64 X: only X's fp32 rounding separates them
        np.testing.assert_allclose(rows, g[f"{tag}_rows"], rtol=1e-6, atol=1e-7)


def test_c_threads_do_not_change_results():
    g = load_extract("cora")
    n = int(g["num_nodes"])
    A = csr_from_undirected(n, g["edges"])
    rng = np.random.default_rng(0)
    X = rng.standard_normal((n, 19)).astype(np.float32)
    links = np.stack([rng.integers(0, n, 300), rng.integers(0, n, 300)])
    links = links[:, links[0] != links[1]]
    a = c_oracle.pos_rows(links, 2, A, X, 3, plus=True, threads=1)
    b = c_oracle.pos_rows(links, 2, A, X, 3, plus=True, threads=4)
    for u, v in zip(a, b):
        np.testing.assert_array_equal(u, v)
    lst = oracle.get_PoS_Plus_prepped_ds(links[:, :40], 2, A, X.astype(np.float64), 1,
                                         {"sign_k": 3, "k_node_set_strategy": "intersection"}, dtype=np.float64)
    ref, ref_ptr, _ = oracle.collate_rows(lst, 3)
    np.testing.assert_allclose(a[0][:ref_ptr[-1]], ref, rtol=1e-12, atol=1e-14)


def test_c_hops_zero_and_self_link():
    g = load_extract("probe5")
    A = csr_from_undirected(int(g["num_nodes"]), g["edges"])
    X = np.arange(15, dtype=np.float32).reshape(5, 3)
    rows, ptr, rn, nc = c_oracle.pos_rows(np.array([[0], [1]]), 0, A, X, 2, plus=True)
    assert list(ptr) == [0, 2] and list(nc) == [2]
    assert np.all(rows[:, 1:, :] == 0)                       # the only edge candidate is the masked link
    with pytest.raises(ValueError):
        c_oracle.pos_rows(np.array([[2], [2]]), 1, A, X, 2)
