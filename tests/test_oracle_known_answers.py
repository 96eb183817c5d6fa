"""Hand-derived known answers — these anchor the *diffusion* half of the oracle,
which the reference itself cannot pin (its arithmetic is in torch_sparse, not installed)."""
import numpy as np
import scipy.sparse as ssp

import oracle
from conftest import csr_from_undirected

KW = {"sign_k": 3, "k_node_set_strategy": "intersection"}


def test_pair_graph_everything_masked():
    # (i) single edge 0-1, link (0,1): masked subgraph empty -> deg 0 -> inf->0 -> x_i = 0
    A = csr_from_undirected(2, [[0, 1]])
    X = np.array([[2.0, 3.0], [5.0, 7.0]])
    (d,) = oracle.get_PoS_prepped_ds(np.array([[0], [1]]), 2, A, X, 1, KW, dtype=np.float64)
    np.testing.assert_array_equal(d["x"], [[1, 2, 3], [1, 5, 7]])
    for i in (1, 2, 3):
        np.testing.assert_array_equal(d[f"x{i}"], np.zeros((2, 3)))


def test_triangle_pos_and_plus():
    # (ii) triangle, link (0,1): after masking edges 0-2, 1-2; d=(1,1,2); Â[0,2]=Â[1,2]=1/√2
    A = csr_from_undirected(3, [[0, 1], [0, 2], [1, 2]])
    X = np.array([[1.0, 10.0], [2.0, 20.0], [3.0, 30.0]])
    li = np.array([[0], [1]])
    (d,) = oracle.get_PoS_prepped_ds(li, 1, A, X, 1, KW, dtype=np.float64)
    s = 1 / np.sqrt(2)
    np.testing.assert_allclose(d["x1"], [[0, 3 * s, 30 * s], [0, 3 * s, 30 * s]])
    # Â²[0,0] = Â²[0,1] = 1/2 -> x2[0] = ½[1|X0] + ½[1|X1]
    np.testing.assert_allclose(d["x2"][0], [1.0, 1.5, 15.0])
    np.testing.assert_allclose(d["x2"][1], [1.0, 1.5, 15.0])
    # Â³ = Â (eigen-structure of the path 0-2-1): x3 == x1
    np.testing.assert_allclose(d["x3"], d["x1"])
    (p,) = oracle.get_PoS_Plus_prepped_ds(li, 1, A, X, 1, KW, dtype=np.float64)
    np.testing.assert_array_equal(p["rows_global"], [0, 1, 2])          # CN = {2}, R = 3
    np.testing.assert_allclose(p["x"], [[1, 1, 10], [1, 2, 20], [0, 3, 30]])
    # row of node 2: Â[2,0]=Â[2,1]=1/√2 -> x1[2] = s[1|X0] + s[1|X1]
    np.testing.assert_allclose(p["x1"][2], [2 * s, 3 * s, 30 * s])


def test_isolated_node_inf_to_zero():
    # (iii) star 0-{1,2,3,4}, 4-5, node 6 isolated; link (0,6): row of node 6 has deg 0
    A = csr_from_undirected(7, [[0, 1], [0, 2], [0, 3], [0, 4], [4, 5]])
    X = np.arange(14, dtype=np.float64).reshape(7, 2) + 1
    (d,) = oracle.get_PoS_prepped_ds(np.array([[0], [6]]), 2, A, X, 0, KW, dtype=np.float64)
    assert np.all(np.isfinite(d["x1"]))
    np.testing.assert_array_equal(d["x1"][1], 0)
    np.testing.assert_array_equal(d["x3"][1], 0)
    # x1[0] = Σ_{j=1..4} Â[0,j] X_j ; deg(0)=4, deg(1..3)=1, deg(4)=2
    w = np.array([0.5, 0.5, 0.5, 1 / np.sqrt(8)])
    np.testing.assert_allclose(d["x1"][0, 1:], w @ X[1:5])
    assert d["x1"][0, 0] == 0                                          # label column: Â[0,0]+Â[0,6]


def test_sop_triangle():
    # (iv) global (unmasked) triangle: all deg 2, Â = ½(J−I), Â² = ¼(J+I)
    A = csr_from_undirected(3, [[0, 1], [0, 2], [1, 2]])
    X = np.array([[1.0, 10.0], [2.0, 20.0], [3.0, 30.0]])
    P = oracle.global_normalized_powers(A, 2, np.float64)
    np.testing.assert_allclose(P[0].toarray(), 0.5 * (np.ones((3, 3)) - np.eye(3)))
    np.testing.assert_allclose(P[1].toarray(), 0.25 * (np.ones((3, 3)) + np.eye(3)))
    (d,) = oracle.get_SoP_prepped_ds(P, np.array([[0], [1]]), A, X, 1, dtype=np.float64)
    np.testing.assert_allclose(d["x"], [[1, 1, 10], [1, 2, 20]])
    # x1[0] = [Â[0,0]=0 | ½·X2]  (the dst term ½·X1 is masked)
    np.testing.assert_allclose(d["x1"][0], [0, 1.5, 15])
    np.testing.assert_allclose(d["x1"][1], [0, 1.5, 15])
    # x2[0] = [Â²[0,0]=½ | ½·X0 + ¼·X2]
    np.testing.assert_allclose(d["x2"][0], [0.5, 0.5 * 1 + 0.75, 0.5 * 10 + 7.5])


def test_probe5_cn_and_explicit_zero_semantics():
    # (v) edges 0-2,1-2,0-3,1-3,3-4; link (0,1) absent -> K2: two explicit zeros inserted
    A = csr_from_undirected(5, [[0, 2], [1, 2], [0, 3], [1, 3], [3, 4]])
    nodes, sub, dists, _, _ = oracle.k_hop_subgraph(0, 1, 1, A)
    assert nodes == [0, 1, 2, 3] and dists == [0, 0, 1, 1]
    assert sub.nnz == 10 and len(ssp.find(sub)[0]) == 8                # K2 / K3
    assert oracle.neighbors({0}, sub) == {1, 2, 3}                      # K4
    (p,) = oracle.get_PoS_Plus_prepped_ds(np.array([[0], [1]]), 1, A, np.eye(5), 1, KW,
                                          dtype=np.float64)
    np.testing.assert_array_equal(p["rows_global"], [0, 1, 2, 3])


def test_hybrid_and_pool_contract():
    A = csr_from_undirected(5, [[0, 2], [1, 2], [0, 3], [1, 3], [3, 4]])
    X = np.random.default_rng(0).standard_normal((5, 3))
    li = np.array([[0, 3], [1, 4]])
    pos = oracle.get_PoS_prepped_ds(li, 2, A, X, 1, KW, dtype=np.float64)
    sop = oracle.get_SoP_prepped_ds(oracle.global_normalized_powers(A, 3, np.float64), li, A, X, 1,
                                    dtype=np.float64)
    hyb = oracle.hybrid_combine(pos, sop, 3)
    assert sorted(k for k in hyb[0] if k.startswith("x")) == ["x", "x1", "x2", "x3", "x4", "x5"]
    np.testing.assert_array_equal(hyb[1]["x4"], sop[1]["x2"])
    np.testing.assert_array_equal(hyb[1]["x5"], sop[1]["x3"])
    plus = oracle.get_PoS_Plus_prepped_ds(li, 2, A, X, 1, KW, dtype=np.float64)
    rows, row_ptr, y = oracle.collate_rows(plus, 3)
    assert rows.shape == (row_ptr[-1], 4, 4) and list(y) == [1, 1]
    h = rows.reshape(rows.shape[0], -1)
    out = oracle.centre_pool(h, row_ptr, k_heuristic=1, k_pool_strategy="mean")
    assert out.shape == (2, 2 * h.shape[1])
    np.testing.assert_allclose(out[0, :h.shape[1]], h[0] * h[1])
    np.testing.assert_allclose(out[0, h.shape[1]:], h[2:row_ptr[1]].mean(0))
    np.testing.assert_array_equal(out[1, h.shape[1]:], 0)               # link (3,4): no CN -> zeros


def test_directed_operator_hand_derived():
    """Directed graph 0->2, 2->1, 1->3, 3->0, link (0,1), one hop: S = {0,1} + out/in neighbours =
    {0,1,2,3}; nothing to mask (no arc between 0 and 1); every node has out-degree 1, so D^-1/2 = 1
    and A_hat = A: row 0 of A_hat is e_2, of A_hat^2 is e_1 (0->2->1), of A_hat^3 is e_3; row 1: e_3,
    e_0, e_2 (tuned_SIGN.py:153-175: deg = row counts = out-degrees, no symmetrisation)."""
    import scipy.sparse as ssp

    arcs = np.array([[0, 2], [2, 1], [1, 3], [3, 0]])
    A = ssp.csr_matrix((np.ones(4, dtype=np.int64), (arcs[:, 0], arcs[:, 1])), shape=(4, 4))
    X = np.array([[1.0, 0.0], [0.0, 2.0], [3.0, 0.0], [0.0, 5.0]])
    kw = {"sign_k": 3, "k_node_set_strategy": "intersection"}
    d = oracle.get_PoS_prepped_ds(np.array([[0], [1]]), 1, A, X, 1, kw, dtype=np.float64, directed=True,
                                  A_csc=A.tocsc())[0]
    assert sorted(d["nodes"].tolist()) == [0, 1, 2, 3]
    np.testing.assert_allclose(d["x"], [[1, 1, 0], [1, 0, 2]])
    np.testing.assert_allclose(d["x1"], [[0, 3, 0], [0, 0, 5]])          # e_2 X, e_3 X; label column z_2 = z_3 = 0
    np.testing.assert_allclose(d["x2"], [[1, 0, 2], [1, 1, 0]])          # e_1 [z|X], e_0 [z|X]
    np.testing.assert_allclose(d["x3"], [[0, 0, 5], [0, 3, 0]])
    # an undirected BFS would not find node 3 from {0,1} through out-edges alone: the in-edge 3->0 does
    only_out = oracle.k_hop_subgraph(0, 1, 1, A)[0]
    assert sorted(only_out) == [0, 1, 2, 3] or 3 in only_out      # (1->3 is an out-edge of dst)
    # out-degree normalisation: add 0->3; node 0 has out-degree 2, row 0 of A_hat = (e_2 + e_3) / sqrt(2)
    A2 = ssp.csr_matrix((np.ones(5, dtype=np.int64), ([0, 2, 1, 3, 0], [2, 1, 3, 0, 3])), shape=(4, 4))
    d2 = oracle.get_PoS_prepped_ds(np.array([[0], [1]]), 1, A2, X, 1, kw, dtype=np.float64, directed=True,
                                   A_csc=A2.tocsc())[0]
    np.testing.assert_allclose(d2["x1"][0], (np.array([0, 3.0, 0]) + np.array([0, 0, 5.0])) / np.sqrt(2))
    # row 1 of A_hat: 1 -> 3, D^-1/2: d_1 = 1, d_3 = 1
    np.testing.assert_allclose(d2["x1"][1], [0, 0, 5.0])


def test_sop_operator_counts_duplicate_edges():
    """Multigraph, hand-derived: the pair 0-1 listed twice, 1-2 once (both directions each).  The
    reference's SparseTensor(row, col) keeps the duplicates (sgrl_link_pred.py:161-172): degrees
    (2, 3, 1), A_hat[0,1] = 2 / sqrt(2*3), A_hat[1,2] = 1 / sqrt(3*1); on the coalesced structure
    the degrees would be (1, 2, 1)."""
    import scipy.sparse as ssp

    ei = np.array([[0, 1, 0, 1, 1, 2], [1, 0, 1, 0, 2, 1]])
    A = ssp.csr_matrix((np.ones(6, dtype=np.int64), (ei[0], ei[1])), shape=(3, 3))
    assert A[0, 1] == 2
    P = oracle.global_normalized_powers(A, 2, np.float64, edge_index=ei)
    np.testing.assert_allclose(P[0].toarray(), [[0, 2 / np.sqrt(6), 0], [2 / np.sqrt(6), 0, 1 / np.sqrt(3)],
                                                [0, 1 / np.sqrt(3), 0]])
    np.testing.assert_allclose(P[1].toarray()[0], [4 / 6, 0, 2 / np.sqrt(18)])
    Q = oracle.global_normalized_powers(A, 1, np.float64)
    np.testing.assert_allclose(Q[0].toarray()[0, 1], 1 / np.sqrt(2))


def test_restricted_sop_twin_equals_sop_inside_the_ball():
    """The optional twin of SURVEY §8(d) (SoP rows restricted to the num_hops-ball): operator i has its support
    within i hops, so operators 1..num_hops are the unrestricted ones; with the ball = the whole component all are."""
    import oracle
    from conftest import csr_from_undirected, load_extract

    g = load_extract("usair")
    n = int(g["num_nodes"])
    A = csr_from_undirected(n, g["edges"])
    X = np.random.default_rng(4).standard_normal((n, 5))
    links = g["links"][:8].T
    P = oracle.global_normalized_powers(A, 3, np.float64)
    full = oracle.get_SoP_prepped_ds(P, links, A, X, 1, dtype=np.float64)
    two = oracle.get_SoP_restricted_ds(P, links, 2, A, X, 1, dtype=np.float64)
    big = oracle.get_SoP_restricted_ds(P, links, 30, A, X, 1, dtype=np.float64)
    for a, b, c in zip(full, two, big):
        for i in (1, 2):
            np.testing.assert_allclose(b[f"x{i}"], a[f"x{i}"], rtol=0, atol=1e-14)
        assert np.abs(b["x3"] - a["x3"]).max() > 1e-6          # USAir: three hops leave the two-hop ball
        for i in (1, 2, 3):
            np.testing.assert_allclose(c[f"x{i}"], a[f"x{i}"], rtol=0, atol=1e-14)
