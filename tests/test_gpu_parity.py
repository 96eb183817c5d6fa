"""-m gpu: the HIP engine (through the C ABI) against the reference-pinned extraction fixtures
and the fp64 oracle.  Bars (BASELINE.json north_star): node-index sets bit-exact; float
diffusion products within 1e-5 relative, measured as |got - ref64| <= 1e-5 * max(|ref64|,
‖ref64 row‖∞) + 1e-10 per output row [1+F] (the row-norm floor keeps elements that cancel to ~0
from dividing by nothing; the absolute 1e-10 only matters where a whole reference row is exactly
zero)."""
import numpy as np
import pytest

import oracle
from conftest import (DIFFUSION_NAMES, EXTRACT_NAMES, SAMPLED_NAMES, csr_from_undirected, load_diffusion,
                      load_extract, load_sampled)

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


ATOL = 1e-10   # only matters where the reference is EXACTLY zero (e.g. SoP: a leaf's masked row)


def rel_err(got, ref):
    """max over elements of (|got - ref| - ATOL)+ / max(|ref|, ‖ref row‖∞)"""
    ref = np.asarray(ref, dtype=np.float64)
    got = np.asarray(got, dtype=np.float64)
    if not ref.size:
        return 0.0
    scale = np.maximum(np.abs(ref), np.abs(ref).max(axis=-1, keepdims=True))
    return float(np.max(np.clip(np.abs(got - ref) - ATOL, 0, None) / np.maximum(scale, 1e-30)))


def elem_rel_err(got, ref):
    """north_star's bar taken literally: max over elements with a non-zero reference of
    |got - ref| / |ref| (no row-norm floor); elements whose reference is exactly zero must be
    within ATOL and are not part of the ratio."""
    ref = np.asarray(ref, dtype=np.float64)
    got = np.asarray(got, dtype=np.float64)
    nz = ref != 0
    if not nz.any():
        return 0.0
    assert np.all(np.abs(got[~nz]) <= ATOL)
    return float(np.max(np.abs(got[nz] - ref[nz]) / np.abs(ref[nz])))


def _elem_floor(c32, ref):
    """elem_rel_err of the fp32 restatement, without its assertion on exact zeros (a float sum of
    signed terms need not cancel to the exact zero the fp64 sum gives)."""
    ref = np.asarray(ref, dtype=np.float64)
    nz = ref != 0
    return float(np.max(np.abs(np.asarray(c32)[nz] - ref[nz]) / np.abs(ref[nz]))) if nz.any() else 0.0


def report_errors(name, row_norm, elementwise, floor=None):
    """Both figures side by side (printed, and appended to gpurun_out/ when it exists).  `floor`:
    (row-norm, element-wise) error of the fp32 restatement — the same algorithm in the reference's own
    precision on the CPU (oracle/s3grl_oracle_c.c built with float arithmetic) against the fp64
    result: the noise floor SURVEY §8(c) asks to see next to the engine's error."""
    import json
    from pathlib import Path

    rec = {"test": name, "row_norm_rel_err": row_norm, "elementwise_rel_err": elementwise, "bar": TOL}
    if floor is not None:
        rec["fp32_restatement_row_norm_rel_err"], rec["fp32_restatement_elementwise_rel_err"] = floor
    print("[parity]", json.dumps(rec))
    out = Path(__file__).resolve().parent.parent / "gpurun_out"
    if out.is_dir():
        with open(out / "parity_errors.jsonl", "a") as f:
            f.write(json.dumps(rec) + "\n")


def _ragged(blob, key, i):
    off = blob[key + "_off"]
    return blob[key][off[i]:off[i + 1]]


@pytest.mark.parametrize("name", EXTRACT_NAMES)
def test_subgraph_node_sets_bit_exact(eng, name):
    g = load_extract(name)
    n = int(g["num_nodes"])
    A = csr_from_undirected(n, g["edges"])
    G = eng.graph(A)
    links = eng.links(g["links"].T)
    for h in g["hops"]:
        plan = eng.plan(G, links, mode="pos_plus", num_hops=int(h), sign_k=2, full_stats=True)
        node_ptr, nodes, dists = (t.cpu().numpy() for t in plan.export_subgraphs())
        row_ptr = plan.row_ptr().cpu().numpy()
        row_nodes = plan.row_nodes().cpu().numpy()
        for li, (s, d) in enumerate(g["links"]):
            mine = nodes[node_ptr[li]:node_ptr[li + 1]]
            md = dists[node_ptr[li]:node_ptr[li + 1]]
            # fixture order = hop-major, ascending id inside a hop = the engine's canonical order
            np.testing.assert_array_equal(mine, _ragged(g, f"h{h}_nodes", li))   # set per hop
            np.testing.assert_array_equal(md, _ragged(g, f"h{h}_dists", li))
            rn = row_nodes[row_ptr[li]:row_ptr[li + 1]]
            assert rn[0] == s and rn[1] == d
            np.testing.assert_array_equal(rn[2:], _ragged(g, f"h{h}_cn", li))   # CCN rows
        # induced, masked edge count == non-zero triples of the reference's sub-CSR
        exp_e = sum(int((_ragged(g, f"h{h}_sub", li)[:, 2] != 0).sum()) for li in range(len(g["links"])))
        assert plan.stats["total_sub_edges"] == exp_e
        assert plan.stats["total_nodes"] == int(node_ptr[-1])
        plan.close()
    G.close()


@pytest.mark.parametrize("name", DIFFUSION_NAMES)
@pytest.mark.parametrize("mode", ["pos", "pos_plus"])
def test_diffusion_vs_golden(eng, name, mode):
    g = load_diffusion(name)
    n, K, h = int(g["num_nodes"]), int(g["K"]), int(g["num_hops"])
    G = eng.graph(csr_from_undirected(n, g["edges"]))
    x = eng.features(g["X"])
    res = eng.precompute(G, x, eng.links(g["links"].T), mode=mode, num_hops=h, sign_k=K)
    tag = "pos" if mode == "pos" else "plus"
    np.testing.assert_array_equal(res.row_ptr.cpu().numpy(), g[f"{tag}_row_ptr"])
    np.testing.assert_array_equal(res.row_nodes.cpu().numpy(), g[f"{tag}_rows_global"])
    assert rel_err(res.rows.cpu().numpy(), g[f"{tag}_rows"]) < TOL
    G.close()


@pytest.mark.parametrize("F", [1, 5, 16, 130, 500, 513, 1433])
def test_feature_widths(eng, F):
    """column tiling / padding paths of the gather kernel (F not a multiple of 4, > 512, ...)"""
    g = load_extract("rand300")
    n = int(g["num_nodes"])
    A = csr_from_undirected(n, g["edges"])
    rng = np.random.default_rng(F)
    X = rng.standard_normal((n, F)).astype(np.float32).astype(np.float64)   # what the engine holds, exactly
    if F == 1:
        # one column: the row norm the tolerance scales with IS the element, and a sum of signed terms
        # that cancels has no 1e-5 bar in fp32 under any summation order — keep the terms one-signed
        X = np.abs(X)
    links = g["links"][:12].T
    kw = {"sign_k": 3, "k_node_set_strategy": "intersection"}
    G = eng.graph(A)
    res = eng.precompute(G, eng.features(X), eng.links(links), mode="pos_plus", num_hops=2, sign_k=3)
    ref, ptr, _ = oracle.collate_rows(
        oracle.get_PoS_Plus_prepped_ds(links, 2, A, X, 1, kw, dtype=np.float64), 3)
    np.testing.assert_array_equal(res.row_ptr.cpu().numpy(), ptr)
    assert rel_err(res.rows.cpu().numpy(), ref) < TOL
    G.close()


@pytest.mark.parametrize("K", [1, 2, 4, 5, 8])
def test_sign_k_range(eng, K):
    g = load_extract("usair")
    n = int(g["num_nodes"])
    A = csr_from_undirected(n, g["edges"])
    X = np.random.default_rng(K).random((n, 24))
    links = g["links"][:10].T
    kw = {"sign_k": K, "k_node_set_strategy": "intersection"}
    G = eng.graph(A)
    res = eng.precompute(G, eng.features(X), eng.links(links), mode="pos", num_hops=2, sign_k=K)
    ref, _, _ = oracle.collate_rows(oracle.get_PoS_prepped_ds(links, 2, A, X, 1, kw, dtype=np.float64), K)
    assert rel_err(res.rows.cpu().numpy(), ref) < TOL
    G.close()


def test_known_answers_on_device(eng):
    # triangle, link (0,1): x1[0] = (1/√2)[0|X2]; x2[0] = ½[1|X0] + ½[1|X1]; CN = {2}
    A = csr_from_undirected(3, [[0, 1], [0, 2], [1, 2]])
    X = np.array([[1.0, 10.0], [2.0, 20.0], [3.0, 30.0]])
    G = eng.graph(A)
    res = eng.precompute(G, eng.features(X), eng.links(np.array([[0], [1]])), mode="pos_plus",
                         num_hops=1, sign_k=2)
    rows = res.rows.cpu().numpy()
    s = 1 / np.sqrt(2)
    assert res.row_nodes.cpu().tolist() == [0, 1, 2]
    np.testing.assert_allclose(rows[0, 0], [1, 1, 10])
    np.testing.assert_allclose(rows[2, 0], [0, 3, 30])
    np.testing.assert_allclose(rows[0, 1], [0, 3 * s, 30 * s], rtol=1e-6)
    np.testing.assert_allclose(rows[0, 2], [1.0, 1.5, 15.0], rtol=1e-6)
    np.testing.assert_allclose(rows[2, 1], [2 * s, 3 * s, 30 * s], rtol=1e-6)
    G.close()
    # single edge: everything masked -> inf -> 0 -> operators are all zero
    A = csr_from_undirected(2, [[0, 1]])
    G = eng.graph(A)
    res = eng.precompute(G, eng.features(np.array([[2.0, 3.0], [5.0, 7.0]])),
                         eng.links(np.array([[0], [1]])), mode="pos", num_hops=2, sign_k=3)
    rows = res.rows.cpu().numpy()
    np.testing.assert_array_equal(rows[:, 0], [[1, 2, 3], [1, 5, 7]])
    np.testing.assert_array_equal(rows[:, 1:], 0)
    G.close()


def test_empty_and_errors(eng):
    import torch

    A = csr_from_undirected(5, [[0, 2], [1, 2], [0, 3], [1, 3], [3, 4]])
    G = eng.graph(A)
    x = eng.features(np.eye(5))
    empty = torch.zeros((0, 2), dtype=torch.int64, device=eng.device)
    res = eng.precompute(G, x, empty, mode="pos", num_hops=2, sign_k=2)
    assert res.rows.shape == (0, 3, 6) and res.row_ptr.cpu().tolist() == [0]
    with pytest.raises(ValueError):
        eng.precompute(G, x, eng.links(np.array([[0], [7]])), mode="pos", num_hops=1, sign_k=2)
    with pytest.raises(ValueError):
        eng.precompute(G, x, eng.links(np.array([[2], [2]])), mode="pos", num_hops=1, sign_k=2)
    with pytest.raises(NotImplementedError):
        eng.precompute(G, x, eng.links(np.array([[0], [1]])), mode="pos_plus", num_hops=1, sign_k=2,
                       strategy="union")
    with pytest.raises(AssertionError):
        eng.precompute(G, None, eng.links(np.array([[0], [1]])), mode="pos", num_hops=1, sign_k=2)
    # the engine is still usable after errors, and deterministic run to run
    a = eng.precompute(G, x, eng.links(np.array([[0, 3], [1, 4]])), mode="pos", num_hops=2, sign_k=3)
    b = eng.precompute(G, x, eng.links(np.array([[0, 3], [1, 4]])), mode="pos", num_hops=2, sign_k=3)
    assert torch.equal(a.rows, b.rows)
    G.close()


def test_dropin_operator_api(eng):
    """Same positional signature and return shape as reference tuned_SIGN.py:137-138,192-193."""
    import torch
    from s3grl_amd.tuned_SIGN import OptimizedSignOperations, clear_cache

    g = load_extract("usair")
    n = int(g["num_nodes"])
    A = csr_from_undirected(n, g["edges"])
    X = torch.from_numpy(np.random.default_rng(3).random((n, 16)).astype(np.float32))
    link_index = torch.from_numpy(g["links"][:8].T.copy())
    kw = {"sign_k": 2, "use_feature": True, "sign_type": "PoS", "optimize_sign": True,
          "k_heuristic": 1, "k_node_set_strategy": "intersection"}
    lst = OptimizedSignOperations.get_PoS_Plus_prepped_ds(link_index, 1, A, 1.0, None, False, None, X,
                                                          1, kw, None)
    ref = oracle.get_PoS_Plus_prepped_ds(link_index.numpy(), 1, A, X.numpy().astype(np.float64), 1, kw,
                                         dtype=np.float64)
    assert len(lst) == 8
    for d, r in zip(lst, ref):
        assert d.y == 1 and d.x.shape == r["x"].shape and d["x2"].shape == r["x2"].shape
        for k in ("x", "x1", "x2"):
            assert rel_err(d[k].numpy(), r[k]) < TOL
    # directed=True with A_csc on a symmetric A (arcs both ways): the same operator as the undirected call
    und = OptimizedSignOperations.get_PoS_prepped_ds(link_index, 1, A, 1.0, None, False, None, X, 1, kw, None)
    dirl = OptimizedSignOperations.get_PoS_prepped_ds(link_index, 1, A, 1.0, None, True, A.tocsc(), X, 1, kw,
                                                      None)
    for a, b in zip(und, dirl):
        for k in ("x", "x1", "x2"):
            assert rel_err(b[k].numpy(), a[k].numpy()) < 3e-6
    with pytest.raises(AssertionError):
        OptimizedSignOperations.get_PoS_prepped_ds(link_index, 1, A, 1.0, None, False, None, None, 1,
                                                   kw, None)
    clear_cache()


def test_full_size_properties(eng):
    """Cora-size run: properties that need no oracle — row 0/1 of operator 0 are [1|X[src/dst]],
    outputs are linear in X, and swapping (src,dst) swaps the two rows."""
    import torch

    topo = np.load(__import__("pathlib").Path(__file__).resolve().parent.parent / "s3grl_amd" /
                   "data" / "topo_cora.npz")
    n, e = int(topo["num_nodes"]), topo["edges"].astype(np.int64)
    A = csr_from_undirected(n, e)
    rng = np.random.default_rng(5)
    X1 = torch.from_numpy(rng.random((n, 64)).astype(np.float32))
    X2 = torch.from_numpy(rng.random((n, 64)).astype(np.float32))
    links = np.concatenate([e[rng.choice(len(e), 600, replace=False)],
                            rng.integers(0, n, size=(600, 2))])
    links = links[links[:, 0] != links[:, 1]]
    G = eng.graph(A)
    L = eng.links(links.T)
    a = eng.precompute(G, eng.features(X1), L, mode="pos", num_hops=3, sign_k=3).rows
    b = eng.precompute(G, eng.features(X2), L, mode="pos", num_hops=3, sign_k=3).rows
    c = eng.precompute(G, eng.features(X1 + 2 * X2), L, mode="pos", num_hops=3, sign_k=3).rows
    lin = a[:, :, 1:] + 2 * b[:, :, 1:]
    assert torch.allclose(c[:, :, 1:], lin, rtol=1e-4, atol=1e-5)
    assert torch.equal(c[:, :, 0], a[:, :, 0])                               # label column: X-free
    src = torch.from_numpy(links[:, 0])
    assert torch.equal(a[0::2, 0, 1:].cpu(), X1[src])
    assert torch.all(a[:, 0, 0] == 1)
    sw = eng.precompute(G, eng.features(X1), eng.links(links[:, ::-1].T.copy()), mode="pos",
                        num_hops=3, sign_k=3).rows
    assert torch.equal(sw[0::2], a[1::2]) and torch.equal(sw[1::2], a[0::2])
    G.close()


# ------------------------------------------------------------------------------------------
# SoP (reference tuned_SIGN.py:49-134 + sgrl_link_pred.py:161-178)
# ------------------------------------------------------------------------------------------
@pytest.mark.parametrize("name", DIFFUSION_NAMES)
def test_sop_vs_golden(eng, name):
    g = load_diffusion(name)
    n, K = int(g["num_nodes"]), int(g["K"])
    G = eng.graph(csr_from_undirected(n, g["edges"]))
    res = eng.precompute(G, eng.features(g["X"]), eng.links(g["links"].T), mode="sop", sign_k=K)
    np.testing.assert_array_equal(res.row_ptr.cpu().numpy(), g["sop_row_ptr"])
    assert rel_err(res.rows.cpu().numpy(), g["sop_rows"]) < TOL
    G.close()


@pytest.mark.parametrize("K", [1, 2, 3, 4, 5, 6])
def test_sop_sign_k_range(eng, K):
    g = load_extract("rand300")
    n = int(g["num_nodes"])
    A = csr_from_undirected(n, g["edges"])
    X = np.random.default_rng(K).standard_normal((n, 37))
    links = g["links"].T
    G = eng.graph(A)
    res = eng.precompute(G, eng.features(X), eng.links(links), mode="sop", sign_k=K)
    ref, _, _ = oracle.collate_rows(
        oracle.get_SoP_prepped_ds(oracle.global_normalized_powers(A, K, np.float64), links, A,
                                  X.astype(np.float32).astype(np.float64), 1, dtype=np.float64), K)
    assert rel_err(res.rows.cpu().numpy(), ref) < TOL
    G.close()


@pytest.mark.parametrize("K", [2, 3])
def test_sop_reversed_duplicates_and_launch_order(eng, monkeypatch, K):
    """SoP folds a reversed duplicate (d,s) into its (s,d) (rows swapped) and runs its row kernel in
    (src, dst) order; the output keeps the caller's order.  Against the oracle, and bit for bit
    against the unfolded / unsorted run of the same list."""
    g = load_extract("rand300")
    n = int(g["num_nodes"])
    A = csr_from_undirected(n, g["edges"])
    X = np.random.default_rng(50 + K).random((n, 29)).astype(np.float32)
    base = g["links"]
    rng = np.random.default_rng(K)
    links = np.concatenate([base, base[::2, ::-1], base[:5]])       # reversed duplicates, exact duplicates
    links = links[rng.permutation(len(links))].T
    G = eng.graph(A)
    f = eng.features(X)
    res = eng.precompute(G, f, eng.links(links), mode="sop", sign_k=K).rows.clone()
    ref, _, _ = oracle.collate_rows(
        oracle.get_SoP_prepped_ds(oracle.global_normalized_powers(A, K, np.float64), links, A,
                                  X.astype(np.float64), 1, dtype=np.float64), K)
    assert rel_err(res.cpu().numpy(), ref) < TOL
    monkeypatch.setenv("S3GRL_NO_MIRROR", "1")
    monkeypatch.setenv("S3GRL_SOP_UNSORTED", "1")
    plain = eng.precompute(G, f, eng.links(links), mode="sop", sign_k=K).rows
    monkeypatch.delenv("S3GRL_NO_MIRROR")
    monkeypatch.delenv("S3GRL_SOP_UNSORTED")
    import torch

    # the scalars are formed in canonical orientation of the pair: folding changes no bit
    assert torch.equal(res, plain)
    G.close()


def test_sop_known_answers_and_exact_cancellation(eng):
    # (iv) triangle: Â = ½(J−I), Â² = ¼(J+I); link (0,1)
    A = csr_from_undirected(3, [[0, 1], [0, 2], [1, 2]])
    X = np.array([[1.0, 10.0], [2.0, 20.0], [3.0, 30.0]])
    G = eng.graph(A)
    rows = eng.precompute(G, eng.features(X), eng.links(np.array([[0], [1]])), mode="sop",
                          sign_k=2).rows.cpu().numpy()
    np.testing.assert_allclose(rows[:, 0], [[1, 1, 10], [1, 2, 20]])
    np.testing.assert_allclose(rows[0, 1], [0, 1.5, 15], rtol=1e-6)      # dst term masked
    np.testing.assert_allclose(rows[0, 2], [0.5, 1.25, 12.5], rtol=1e-6)
    G.close()
    # star: leaf s hanging off hub d.  Every odd power of Â sends s only to d, so the masked
    # row is EXACTLY zero in the reference; the f64 closed form must not leave f32-sized noise.
    A = csr_from_undirected(6, [[0, 1], [0, 2], [0, 3], [0, 4], [0, 5]])
    X = np.random.default_rng(0).standard_normal((6, 9))
    G = eng.graph(A)
    rows = eng.precompute(G, eng.features(X), eng.links(np.array([[3], [0]])), mode="sop",
                          sign_k=5).rows.cpu().numpy()
    ref, _, _ = oracle.collate_rows(
        oracle.get_SoP_prepped_ds(oracle.global_normalized_powers(A, 5, np.float64),
                                  np.array([[3], [0]]), A, X.astype(np.float32).astype(np.float64),
                                  1, dtype=np.float64), 5)
    for i in (1, 3, 5):
        assert np.all(np.abs(ref[0, i, 1:]) < 1e-15)
        assert np.all(np.abs(rows[0, i, 1:]) < 1e-12)
    assert rel_err(rows, ref) < TOL
    G.close()


def test_sop_dropin_and_errors(eng):
    import torch
    from s3grl_amd.tuned_SIGN import OptimizedSignOperations, clear_cache

    g = load_extract("usair")
    n = int(g["num_nodes"])
    A = csr_from_undirected(n, g["edges"])
    X = torch.from_numpy(np.random.default_rng(4).random((n, 12)).astype(np.float32))
    li = torch.from_numpy(g["links"][:9].T.copy())
    lst = OptimizedSignOperations.get_SoP_prepped_ds([None, None, None], li, A, X, 0)
    ref = oracle.get_SoP_prepped_ds(oracle.global_normalized_powers(A, 3, np.float64), li.numpy(), A,
                                    X.numpy().astype(np.float64), 0, dtype=np.float64)
    assert len(lst) == 9 and lst[0].y == 0
    for d, r in zip(lst, ref):
        for k in ("x", "x1", "x2", "x3"):
            assert d[k].shape == (2, 13) and rel_err(d[k].numpy(), r[k]) < TOL
    with pytest.raises(ValueError):
        eng.precompute(eng.graph(A), eng.features(X), eng.links(np.array([[5], [5]])), mode="sop",
                       sign_k=2)
    clear_cache()


# ------------------------------------------------------------------------------------------
# feature operand: dense rows vs sparse rows give the same sums
# ------------------------------------------------------------------------------------------
@pytest.mark.parametrize("F,density", [(7, 0.2), (500, 0.1), (513, 0.02), (1433, 0.0127), (600, 0.6)])
def test_sparse_and_dense_feature_paths(eng, F, density):
    g = load_extract("rand300")
    n = int(g["num_nodes"])
    A = csr_from_undirected(n, g["edges"])
    rng = np.random.default_rng(F)
    X = rng.standard_normal((n, F)) * (rng.random((n, F)) < density)
    X[7] = rng.standard_normal(F)          # one fully dense row: > 64 non-zeros per tile
    X[11] = 0                              # one empty row
    links = g["links"][:14].T
    kw = {"sign_k": 3, "k_node_set_strategy": "intersection"}
    ref, ptr, _ = oracle.collate_rows(
        oracle.get_PoS_Plus_prepped_ds(links, 2, A, X.astype(np.float32).astype(np.float64), 1, kw,
                                       dtype=np.float64), 3)
    G = eng.graph(A)
    L = eng.links(links)
    outs = {}
    Xp = np.zeros((n, (F + 3) // 4 * 4), np.float32)
    Xp[:, :F] = X
    chunk_density = float((Xp.reshape(n, -1, 4) != 0).any(-1).mean())
    for mode in ("auto", "dense", "sparse", "packed"):
        f = eng.features(X, mode)
        assert f.is_sparse == (mode == "sparse")
        assert f.is_packed == (mode == "packed" or (mode == "auto" and chunk_density <= 0.5))
        res = eng.precompute(G, f, L, mode="pos_plus", num_hops=2, sign_k=3)
        np.testing.assert_array_equal(res.row_ptr.cpu().numpy(), ptr)
        assert rel_err(res.rows.cpu().numpy(), ref) < TOL
        outs[mode] = res.rows
        f.close()
    # raw tensor (s3grl_run, plain dense) agrees with the prepared dense operand bit for bit
    raw = eng.precompute(G, eng.features(X, "dense").tensor, L, mode="pos_plus", num_hops=2, sign_k=3)
    import torch
    assert torch.equal(raw.rows, outs["dense"])
    # packed rows skip only all-zero chunks and zero coefficients: the same sums as the dense
    # kernel's, up to where the compiler fuses a multiply-add (1 ulp)
    assert rel_err(outs["packed"].cpu().numpy(), outs["dense"].cpu().numpy()) < 1e-6
    G.close()


@pytest.mark.parametrize("K", [1, 2, 5, 8])
@pytest.mark.parametrize("F", [3, 130, 515])
def test_packed_rows_on_a_borrowed_strided_operand(eng, F, K):
    """Packed copy built from a view whose row stride exceeds F (the columns between F and the
    stride hold OTHER data, which must not leak in), every sign_k instantiation."""
    import torch

    g = load_extract("usair")
    n = int(g["num_nodes"])
    A = csr_from_undirected(n, g["edges"])
    rng = np.random.default_rng(F + K)
    ld = (F + 3) // 4 * 4 + 8
    big = rng.standard_normal((n, ld)).astype(np.float32)
    big[:, :F] *= rng.random((n, F)) < 0.2
    Xd = torch.from_numpy(big).to(eng.device)
    view = Xd[:, :F]
    assert view.stride(0) == ld and view.data_ptr() % 16 == 0
    links = g["links"][:10].T
    kw = {"sign_k": K, "k_node_set_strategy": "intersection"}
    ref, _, _ = oracle.collate_rows(
        oracle.get_PoS_prepped_ds(links, 1, A, big[:, :F].astype(np.float64), 1, kw, dtype=np.float64), K)
    G = eng.graph(A)
    from s3grl_amd.engine import Features

    outs = {}
    for mode in ("dense", "packed"):
        f = Features(eng, view, mode)
        res = eng.precompute(G, f, eng.links(links), mode="pos", num_hops=1, sign_k=K)
        assert rel_err(res.rows.cpu().numpy(), ref) < TOL
        outs[mode] = res.rows
        f.close()
    assert rel_err(outs["packed"].cpu().numpy(), outs["dense"].cpu().numpy()) < 1e-6
    G.close()


# ------------------------------------------------------------------------------------------
# consumer-side centre / common-neighbour pooling (reference models.py:339-369)
# ------------------------------------------------------------------------------------------
@pytest.mark.parametrize("H", [256, 37])
@pytest.mark.parametrize("strategy", ["", "mean", "sum"])
def test_centre_pool_forward_backward(eng, H, strategy):
    import torch
    from s3grl_amd.pool import centre_pool, row_ptr_from_batch

    rng = np.random.default_rng(H)
    counts = np.concatenate([[2, 2, 5, 3, 2, 9], rng.integers(2, 8, size=40)])
    if not strategy:
        counts[:] = 2
    row_ptr = np.zeros(len(counts) + 1, dtype=np.int64)
    np.cumsum(counts, out=row_ptr[1:])
    h = rng.standard_normal((row_ptr[-1], H)).astype(np.float32)
    k = 1 if strategy else 0
    ref = oracle.centre_pool(h.astype(np.float64), row_ptr, k_heuristic=k, k_pool_strategy=strategy)
    ht = torch.tensor(h, device=eng.device, requires_grad=True)
    rp = torch.tensor(row_ptr, device=eng.device)
    out = centre_pool(ht, rp, k_heuristic=k, k_pool_strategy=strategy)
    assert rel_err(out.detach().cpu().numpy(), ref) < 1e-6
    # backward against autograd of a plain torch restatement of the same formula
    g = torch.tensor(rng.standard_normal(out.shape).astype(np.float32), device=eng.device)
    out.backward(g)
    h2 = torch.tensor(h, device=eng.device, requires_grad=True)
    c = rp[:-1]
    parts = [h2[c] * h2[c + 1]]
    if strategy:
        pooled = []
        for b in range(len(counts)):
            extra = h2[row_ptr[b] + 2: row_ptr[b + 1]]
            pooled.append(extra.sum(0) if strategy == "sum" or len(extra) == 0 else extra.mean(0))
        parts.append(torch.stack(pooled))
    torch.cat(parts, -1).backward(g)
    assert torch.allclose(ht.grad, h2.grad, rtol=1e-5, atol=1e-6)
    # batch-vector entry point
    batch = torch.repeat_interleave(torch.arange(len(counts), device=eng.device),
                                    torch.tensor(counts, device=eng.device))
    assert torch.equal(row_ptr_from_batch(batch), rp)
    with pytest.raises(NotImplementedError):
        centre_pool(ht, rp, k_heuristic=1, k_pool_strategy="max")
    with pytest.raises(RuntimeError):                      # ragged rows: the reference's reshape fails too
        centre_pool(ht, rp, k_heuristic=50, k_pool_strategy="concat")
    # 'concat' (models.py:363-367): exactly k_heuristic rows after the two centre rows of every link
    k, B = 3, 17
    hc = rng.standard_normal((B * (2 + k), H)).astype(np.float32)
    rpc = torch.arange(0, B * (2 + k) + 1, 2 + k, dtype=torch.int64, device=eng.device)
    hct = torch.from_numpy(hc).to(eng.device).requires_grad_(True)
    got = centre_pool(hct, rpc, k_heuristic=k, k_pool_strategy="concat")
    ref = oracle.centre_pool(hc, rpc.cpu().numpy(), k_heuristic=k, k_pool_strategy="concat")
    assert got.shape == (B, H * (1 + k)) and np.allclose(got.detach().cpu().numpy(), ref, rtol=1e-6, atol=1e-6)
    got.sum().backward()
    assert hct.grad is not None and torch.isfinite(hct.grad).all()


def test_reversed_duplicates_are_folded_bit_exactly(eng):
    """(d,s) after (s,d) in the list is served by one extraction; the result must be bit-identical
    to computing every link on its own (fold_reversed=False), for PoS and PoS Plus."""
    import torch

    g = load_extract("usair")
    n = int(g["num_nodes"])
    A = csr_from_undirected(n, g["edges"])
    X = np.random.default_rng(9).standard_normal((n, 33))
    base = g["links"][:20]
    links = np.concatenate([base, base[::-1, ::-1], base[:5], base[:3, ::-1]])   # reversed + repeats
    G = eng.graph(A)
    f = eng.features(X)
    L = eng.links(links.T.copy())
    for mode in ("pos", "pos_plus"):
        outs = []
        for fold in (True, False):
            plan = eng.plan(G, L, mode=mode, num_hops=2, sign_k=3, fold_reversed=fold)
            outs.append((plan.run(f), plan.row_ptr(), plan.row_nodes(), dict(plan.stats)))
            plan.close()
        (r1, p1, n1, s1), (r0, p0, n0, s0) = outs
        assert s1["folded_links"] >= 20 and s0["folded_links"] == 0
        assert torch.equal(p1, p0) and torch.equal(n1, n0) and torch.equal(r1, r0)
        for k in ("total_nodes", "total_volume", "total_support", "total_rows"):
            assert s1[k] == s0[k], k                       # algorithmic totals do not change
        assert s1["extracted_nodes"] < s0["extracted_nodes"]
    G.close()


def test_hbm_scratch_path_for_oversized_subgraphs(eng, monkeypatch):
    """Links whose lists do not fit LDS run with them in HBM scratch (GS = true): same results.
    The LDS budget is shrunk through the test hook so that most links take that path."""
    import torch

    g = load_extract("rand300")
    n = int(g["num_nodes"])
    A = csr_from_undirected(n, g["edges"])
    X = np.random.default_rng(21).standard_normal((n, 19))
    G = eng.graph(A)
    f = eng.features(X)
    L = eng.links(g["links"].T)
    ref = eng.precompute(G, f, L, mode="pos_plus", num_hops=2, sign_k=3)
    monkeypatch.setenv("S3GRL_LDS_BUDGET", "600")
    alt = eng.precompute(G, f, L, mode="pos_plus", num_hops=2, sign_k=3)
    monkeypatch.delenv("S3GRL_LDS_BUDGET")
    assert torch.equal(ref.row_ptr, alt.row_ptr) and torch.equal(ref.row_nodes, alt.row_nodes)
    # same sums, not always the same bits: in LDS a small subgraph that every operator reaches
    # entirely is propagated through its adjacency bit matrix, in HBM scratch through the row walker
    assert rel_err(alt.rows.cpu().numpy(), ref.rows.cpu().numpy()) < 1e-6
    assert ref.stats == {**alt.stats, "workspace_bytes": ref.stats["workspace_bytes"]}
    G.close()


# ------------------------------------------------------------------------------------------
# corner cases of the reference's semantics (SURVEY §8c K2/K4 and friends)
# ------------------------------------------------------------------------------------------
def _csr_with_self_loops(n, edges, loops):
    import scipy.sparse as ssp

    A = csr_from_undirected(n, edges).tolil()
    for v in loops:
        A[v, v] = 1
    return ssp.csr_matrix(A)


@pytest.mark.parametrize("mode", ["pos", "pos_plus"])
@pytest.mark.parametrize("hops,K", [(0, 2), (1, 1), (1, 4), (2, 5), (3, 2)])
def test_corner_cases_vs_oracle(eng, mode, hops, K):
    """self-loops (they put src/dst themselves among the 'common neighbours', K4), isolated and
    cross-component endpoints, repeated and reversed links, num_hops = 0, sign_k > num_hops."""
    n = 14
    edges = [[0, 1], [0, 2], [1, 2], [2, 3], [3, 4], [4, 5], [5, 0], [6, 7], [7, 8], [2, 9], [9, 10]]
    A = _csr_with_self_loops(n, edges, loops=[0, 3, 7, 9])         # 11, 12, 13 isolated
    X = np.random.default_rng(hops * 10 + K).standard_normal((n, 6))
    links = np.array([[0, 1], [1, 0], [0, 3], [3, 0], [0, 1], [6, 8], [7, 6], [0, 7], [11, 12],
                      [13, 2], [9, 2], [2, 9], [9, 3], [4, 10]]).T
    kw = {"sign_k": K, "k_node_set_strategy": "intersection"}
    fn = oracle.get_PoS_prepped_ds if mode == "pos" else oracle.get_PoS_Plus_prepped_ds
    lst = fn(links, hops, A, X.astype(np.float32).astype(np.float64), 1, kw, dtype=np.float64)
    ref, ptr, _ = oracle.collate_rows(lst, K)
    G = eng.graph(A)
    for fold in (True, False):
        plan = eng.plan(G, eng.links(links), mode=mode, num_hops=hops, sign_k=K, fold_reversed=fold)
        rows = plan.run(eng.features(X)).cpu().numpy()
        np.testing.assert_array_equal(plan.row_ptr().cpu().numpy(), ptr)
        got_nodes = plan.row_nodes().cpu().numpy()
        for l, d in enumerate(lst):                    # CCN rows as a multiset, centre rows in order
            mine = got_nodes[ptr[l]:ptr[l + 1]]
            assert list(mine[:2]) == list(d["rows_global"][:2])
            assert sorted(mine[2:]) == sorted(d["rows_global"][2:])
        assert rel_err(rows, ref) < TOL
        plan.close()
    G.close()


def test_wrapper_rejects_what_the_engine_cannot_mirror(eng):
    import scipy.sparse as ssp

    A = ssp.csr_matrix(np.array([[0, 1, 0], [0, 0, 1], [0, 0, 0]]))     # not symmetric: needs directed=True
    with pytest.raises(ValueError, match="directed"):
        eng.graph(A)
    B = csr_from_undirected(3, [[0, 1], [1, 2]]).astype(np.float64)
    B.data[0] = 0.0                                                      # stored zero
    with pytest.raises(ValueError):
        eng.graph(B)
    with pytest.raises(ValueError):
        eng.links(np.zeros((3, 4), dtype=np.int64))


def test_hybrid_and_tuned_sign_twin(eng):
    import torch
    from s3grl_amd.tuned_SIGN import LinkData, TunedSIGN, clear_cache

    g = load_extract("usair")
    n = int(g["num_nodes"])
    A = csr_from_undirected(n, g["edges"])
    X = np.random.default_rng(8).random((n, 10))
    links = g["links"][:12].T
    K = 3
    kw = {"sign_k": K, "k_node_set_strategy": "intersection"}
    X64 = X.astype(np.float32).astype(np.float64)
    pos = oracle.get_PoS_prepped_ds(links, 2, A, X64, 1, kw, dtype=np.float64)
    sop = oracle.get_SoP_prepped_ds(oracle.global_normalized_powers(A, K, np.float64), links, A, X64, 1,
                                    dtype=np.float64)
    hyb = oracle.hybrid_combine(pos, sop, K)
    ref = np.concatenate([np.stack([d[k] for k in ["x"] + [f"x{i}" for i in range(1, 2 * K)]], axis=1)
                          for d in hyb])
    G = eng.graph(A)
    res = eng.precompute(G, eng.features(X), eng.links(links), mode="hybrid", num_hops=2, sign_k=K)
    assert res.rows.shape == (24, 2 * K, 11) and rel_err(res.rows.cpu().numpy(), ref) < TOL
    # TunedSIGN twin == PyG SIGN(K): x_i = Â^i x for all rows of the graph it is given
    coo = A.tocoo()
    data = LinkData(x=torch.from_numpy(X.astype(np.float32)),
                    edge_index=torch.from_numpy(np.stack([coo.row, coo.col]).astype(np.int64)),
                    num_nodes=n)
    out = TunedSIGN(K)(data, K)
    P = oracle.global_normalized_powers(A, K, np.float64)
    for i in range(1, K + 1):
        assert rel_err(out[f"x{i}"].numpy(), np.asarray(P[i - 1] @ X64)) < TOL
    data2 = TunedSIGN(K)(LinkData(x=data.x, edge_index=data.edge_index, num_nodes=n), -1)
    assert "x1" not in data2 and "x2" not in data2 and f"x{K}" in data2
    G.close()
    clear_cache()


@pytest.mark.parametrize("name,hops,K", [("rand300", 2, 3), ("usair", 2, 2), ("cora", 3, 3), ("probe5", 3, 5)])
def test_hash_flavour_matches_bitmap_flavour(eng, monkeypatch, name, hops, K):
    """Graphs whose bitmaps would hog the LDS use a hash table as visited set (HS = true); forced
    here on small graphs: node lists, row selection and statistics must be identical bit for bit,
    the rows equal to fp32 round-off (when every operator reaches the whole of a small subgraph
    the hash flavour propagates through its adjacency bit matrix: another summation order)."""
    import torch

    g = load_extract(name)
    n = int(g["num_nodes"])
    A = csr_from_undirected(n, g["edges"])
    X = np.random.default_rng(31).standard_normal((n, 21))
    G = eng.graph(A)
    f = eng.features(X)
    L = eng.links(g["links"].T)
    for mode in ("pos", "pos_plus"):
        monkeypatch.delenv("S3GRL_FORCE_HASH", raising=False)
        p0 = eng.plan(G, L, mode=mode, num_hops=hops, sign_k=K, full_stats=True)
        r0, e0 = p0.run(f), p0.export_subgraphs()
        monkeypatch.setenv("S3GRL_FORCE_HASH", "1")
        p1 = eng.plan(G, L, mode=mode, num_hops=hops, sign_k=K, full_stats=True)
        r1, e1 = p1.run(f), p1.export_subgraphs()
        monkeypatch.delenv("S3GRL_FORCE_HASH")
        assert rel_err(r1.cpu().numpy(), r0.cpu().numpy()) < 1e-6 and torch.equal(p0.row_ptr(), p1.row_ptr())
        assert torch.equal(p0.row_nodes(), p1.row_nodes())
        for a, b in zip(e0, e1):
            assert torch.equal(a, b)
        s0, s1 = dict(p0.stats), dict(p1.stats)
        s0.pop("workspace_bytes"), s1.pop("workspace_bytes")
        assert s0 == s1
        p0.close(), p1.close()
    G.close()


def test_headline_workload_full_size(eng):
    """BASELINE.json's headline config at full size (PubMed PoS sign_k=3, 3-hop, F=500, all
    164 000 links of the three splits): size-independent properties on everything, and the fp64
    oracle on a sample of links drawn from the full result."""
    import torch
    from s3grl_amd import workloads

    w = workloads.make("pubmed_pos_k3")
    link_index, y = w.split.all_links()
    L = link_index.shape[1]
    assert L == 164000
    G = eng.graph(w.A)
    f = eng.features(w.X)
    res = eng.precompute(G, f, eng.links(link_index), mode="pos", num_hops=3, sign_k=3)
    rows = res.rows
    assert rows.shape == (2 * L, 4, 501) and res.stats["folded_links"] > 30000
    X = torch.from_numpy(w.X).to(rows.device)
    li = torch.from_numpy(link_index).to(rows.device)
    # operator 0 is [1 | X[node]] for both centre rows of every link
    assert torch.equal(rows[0::2, 0, 1:], X[li[0]]) and torch.equal(rows[1::2, 0, 1:], X[li[1]])
    assert torch.all(rows[:, 0, 0] == 1)
    # X >= 0 and every operator entry >= 0: nothing negative, nothing non-finite
    assert torch.isfinite(rows).all() and (rows >= 0).all()
    # both directions of a train edge: same rows, swapped
    P = w.split.links["train"][0].shape[1]
    key = {(int(a), int(b)): i for i, (a, b) in enumerate(link_index[:, :P].T)}
    probe = np.random.default_rng(0).choice(P, 2000, replace=False)
    fwd = torch.tensor(probe, device=rows.device)
    rev = torch.tensor([key[(int(link_index[1, i]), int(link_index[0, i]))] for i in probe],
                       device=rows.device)
    assert torch.equal(rows[2 * fwd], rows[2 * rev + 1]) and torch.equal(rows[2 * fwd + 1], rows[2 * rev])
    # sampled links against the fp64 oracle
    rng = np.random.default_rng(1)
    sample = np.concatenate([rng.choice(np.flatnonzero(y == 1), 40, replace=False),
                             rng.choice(np.flatnonzero(y == 0), 40, replace=False)])
    kw = {"sign_k": 3, "k_node_set_strategy": "intersection"}
    ref, _, _ = oracle.collate_rows(
        oracle.get_PoS_prepped_ds(link_index[:, sample], 3, w.A, w.X.astype(np.float64), 1, kw,
                                  dtype=np.float64), 3)
    idx = torch.tensor(np.stack([2 * sample, 2 * sample + 1], 1).reshape(-1), device=rows.device)
    assert rel_err(rows[idx].cpu().numpy(), ref) < TOL
    # EVERY link against the plain-C fp64 restatement (all host cores), streamed in chunks
    from oracle import c_oracle

    worst, worst_el, chunk = 0.0, 0.0, 8000
    floor, floor_el = 0.0, 0.0
    node_total = 0
    for lo in range(0, L, chunk):
        hi = min(lo + chunk, L)
        cref, cptr, cnodes, ccount = c_oracle.pos_rows(link_index[:, lo:hi], 3, w.A, w.X, 3)
        node_total += int(ccount.sum())
        got = rows[2 * lo:2 * hi].cpu().numpy()
        worst = max(worst, rel_err(got, cref))
        worst_el = max(worst_el, elem_rel_err(got, cref))
        c32 = c_oracle.pos_rows(link_index[:, lo:hi], 3, w.A, w.X, 3, f32=True)[0]
        floor, floor_el = max(floor, rel_err(c32, cref)), max(floor_el, _elem_floor(c32, cref))
    report_errors("headline pubmed_pos_k3, all 164000 links vs C fp64", worst, worst_el, (floor, floor_el))
    assert worst < TOL, worst
    # PoS sums have no negative term (X >= 0, operator entries >= 0): the element-wise bar holds too
    assert worst_el < TOL, worst_el
    assert node_total == res.stats["total_nodes"]          # subgraph sizes, summed over all links
    G.close()


# ------------------------------------------------------------------------------------------
# ScaLed random-walk subgraphs (reference utils.py:86-150 with sign=True, rw_kwargs)
# ------------------------------------------------------------------------------------------
@pytest.mark.parametrize("mode", ["pos", "pos_plus"])
def test_scaled_random_walk_subgraphs(eng, monkeypatch, mode):
    import torch
    from scipy.sparse.csgraph import shortest_path

    g = load_extract("usair")
    n = int(g["num_nodes"])
    A = csr_from_undirected(n, g["edges"])
    X = np.random.default_rng(13).standard_normal((n, 9))
    links = g["links"]
    m, M, K = 3, 5, 3
    G = eng.graph(A)
    f = eng.features(X)
    L = eng.links(links.T)
    plan = eng.plan(G, L, mode=mode, num_hops=7, sign_k=K, rw=(m, M, 123), full_stats=True)
    rows = plan.run(f)
    node_ptr, nodes, dists = (t.cpu().numpy() for t in plan.export_subgraphs())
    # walks are walks: every node is within m steps of src or dst, sizes are bounded, src/dst first
    D = shortest_path(A, unweighted=True)
    sets = []
    for l, (s_, d_) in enumerate(links):
        mine = nodes[node_ptr[l]:node_ptr[l + 1]]
        dd = dists[node_ptr[l]:node_ptr[l + 1]]
        assert set(mine[:2]) == {s_, d_} and list(dd[:2]) == [0, 0] and np.all(dd[2:] == 1)
        assert len(mine) <= 2 + 2 * m * M and len(set(mine)) == len(mine)
        assert np.all(np.minimum(D[s_, mine], D[d_, mine]) <= m)
        assert list(mine[2:]) == sorted(mine[2:])
        sets.append(mine)
    # operators on those node sets == the oracle's restatement of the rw branch
    kw = {"sign_k": K, "k_node_set_strategy": "intersection"}
    fn = oracle.get_PoS_prepped_ds if mode == "pos" else oracle.get_PoS_Plus_prepped_ds
    ref, ptr, _ = oracle.collate_rows(
        fn(links.T, 7, A, X.astype(np.float32).astype(np.float64), 1, kw, dtype=np.float64,
           rw_node_sets=sets), K)
    np.testing.assert_array_equal(plan.row_ptr().cpu().numpy(), ptr)
    assert rel_err(rows.cpu().numpy(), ref) < TOL
    # same seed -> same walks (per NODE: a node's walks are shared by all its links); other seed differs
    again = eng.plan(G, L, mode=mode, num_hops=7, sign_k=K, rw=(m, M, 123), full_stats=True)
    assert torch.equal(again.run(f), rows)
    other = eng.plan(G, L, mode=mode, num_hops=7, sign_k=K, rw=(m, M, 124), full_stats=True)
    assert not torch.equal(other.export_subgraphs()[1], plan.export_subgraphs()[1])
    # the hash flavour agrees to round-off (bit matrix there), folded duplicates bit for bit
    monkeypatch.setenv("S3GRL_FORCE_HASH", "1")
    hs = eng.plan(G, L, mode=mode, num_hops=7, sign_k=K, rw=(m, M, 123), full_stats=True)
    monkeypatch.delenv("S3GRL_FORCE_HASH")
    assert rel_err(hs.run(f).cpu().numpy(), rows.cpu().numpy()) < 1e-6
    both = np.concatenate([links[:10], links[:10, ::-1]])
    Lb = eng.links(both.T.copy())
    a = eng.plan(G, Lb, mode=mode, num_hops=7, sign_k=K, rw=(m, M, 5))
    b = eng.plan(G, Lb, mode=mode, num_hops=7, sign_k=K, rw=(m, M, 5), fold_reversed=False)
    assert a.stats["folded_links"] == 10 and torch.equal(a.run(f), b.run(f))
    for p_ in (plan, again, other, hs, a, b):
        p_.close()
    G.close()


def test_scaled_through_the_dropin_api(eng):
    import torch
    from s3grl_amd.tuned_SIGN import OptimizedSignOperations, clear_cache

    g = load_extract("usair")
    n = int(g["num_nodes"])
    A = csr_from_undirected(n, g["edges"])
    X = torch.from_numpy(np.random.default_rng(2).random((n, 8)).astype(np.float32))
    li = torch.from_numpy(g["links"][:6].T.copy())
    kw = {"sign_k": 2, "k_node_set_strategy": "intersection"}
    rw_kwargs = {"rw_m": 2, "rw_M": 4, "sign": True, "seed": 7}
    lst = OptimizedSignOperations.get_PoS_prepped_ds(li, 3, A, 1.0, None, False, None, X, 1, kw, rw_kwargs)
    assert len(lst) == 6 and lst[0].x.shape == (2, 9) and lst[0]["x2"].shape == (2, 9)
    clear_cache()


# ------------------------------------------------------------------------------------------
# the other BASELINE.json configs at full size: sampled links against the fp64 oracle
# ------------------------------------------------------------------------------------------
def _sampled_check(eng, w, mode, n_sample, seed, **kw):
    import torch

    link_index, y = w.split.all_links()
    G = eng.graph(w.A)
    res = eng.precompute(G, eng.features(w.X), eng.links(link_index), mode=mode, sign_k=w.sign_k, **kw)
    rng = np.random.default_rng(seed)
    sample = np.sort(rng.choice(link_index.shape[1], n_sample, replace=False))
    X64 = w.X.astype(np.float64)
    skw = {"sign_k": w.sign_k, "k_node_set_strategy": "intersection"}
    if mode == "sop":
        lst = oracle.get_SoP_prepped_ds(oracle.global_normalized_powers(w.A, w.sign_k, np.float64),
                                        link_index[:, sample], w.A, X64, 1, dtype=np.float64)
    else:
        fn = oracle.get_PoS_prepped_ds if mode == "pos" else oracle.get_PoS_Plus_prepped_ds
        lst = fn(link_index[:, sample], w.num_hops, w.A, X64, 1, skw, dtype=np.float64)
    row_ptr = res.row_ptr.cpu().numpy()
    row_nodes = res.row_nodes.cpu().numpy()
    worst = 0.0
    for s, d in zip(sample, lst):
        blk = res.rows[row_ptr[s]:row_ptr[s + 1]].cpu().numpy()
        ref = np.stack([d[k] for k in ["x"] + [f"x{i}" for i in range(1, w.sign_k + 1)]], axis=1)
        assert blk.shape == ref.shape
        mine = row_nodes[row_ptr[s]:row_ptr[s + 1]]
        assert list(mine[:2]) == list(d["rows_global"][:2]) and list(mine[2:]) == list(d["rows_global"][2:])
        worst = max(worst, rel_err(blk, ref))
    assert worst < TOL, worst
    assert torch.isfinite(res.rows).all()
    G.close()
    return res


def test_cora_pos_plus_full_size(eng):
    """BASELINE config 2: Cora PoS Plus sign_k=3, 3-hop, F=1433, 19 532 links."""
    from s3grl_amd import workloads

    w = workloads.make("cora_posplus_k3")
    res = _sampled_check(eng, w, "pos_plus", 60, 2, num_hops=3)
    assert res.num_links == 19532 and res.rows.shape[0] == int(res.row_ptr[-1])


def test_pubmed_sop_full_size(eng):
    """BASELINE config 3: PubMed SoP sign_k=3 with one-hot degree features (F = 1525), 164 000 links."""
    from s3grl_amd import workloads

    w = workloads.make("pubmed_sop_k3")
    res = _sampled_check(eng, w, "sop", 40, 3)
    assert res.rows.shape == (328000, 4, 1526)


def test_power_law_graph_hash_flavour_and_hubs(eng):
    """A 70 000-node power-law graph: its bitmaps (26 KB) switch the engine to the hash flavour of
    the visited set, its hubs (degree >> 256) arm the wave-per-row path; 1-hop subgraphs, sign_k=3."""
    from s3grl_amd import workloads

    n, e = workloads.chung_lu(70000, 300000, seed=9)
    rng = np.random.default_rng(10)
    X = rng.standard_normal((n, 24)).astype(np.float32)
    pos = e[rng.choice(len(e), 3000, replace=False)]
    neg = rng.integers(0, n, size=(3000, 2))
    neg = neg[neg[:, 0] != neg[:, 1]]
    deg = np.bincount(e.ravel(), minlength=n)
    hubs = np.argsort(-deg)[:8]
    extra = np.array([[hubs[0], hubs[1]], [hubs[2], int(pos[0, 0])], [hubs[3], hubs[4]]])
    sp = workloads.Split(n, e, workloads.csr_from_undirected(n, e),
                         {"train": (np.concatenate([pos, extra]).T.copy(), neg.T.copy()),
                          "valid": (np.zeros((2, 0), np.int64),) * 2, "test": (np.zeros((2, 0), np.int64),) * 2})
    w = workloads.Workload("powerlaw", sp, X, "pos_plus", 1, 3)
    assert deg.max() > 256
    # force the hub rows into the sample: the last 3 positives are hub-hub links
    link_index, _ = sp.all_links()
    res = _sampled_check(eng, w, "pos_plus", 50, 4, num_hops=1)
    G = eng.graph(w.A)
    hub_links = eng.links(np.concatenate([pos[:5], extra]).T.copy())
    got = eng.precompute(G, eng.features(X), hub_links, mode="pos", num_hops=1, sign_k=3)
    ref, _, _ = oracle.collate_rows(
        oracle.get_PoS_prepped_ds(np.concatenate([pos[:5], extra]).T, 1, w.A, X.astype(np.float64), 1,
                                  {"sign_k": 3}, dtype=np.float64), 3)
    assert rel_err(got.rows.cpu().numpy(), ref) < TOL
    G.close()


def test_module_level_precompute_entry(eng):
    """precompute(indptr, indices, X, links, ...) -> (rows, row_ptr, node_count)."""
    import s3grl_amd

    fx = load_diffusion("usair")
    A = csr_from_undirected(int(fx["num_nodes"]), fx["edges"])
    X = fx["X"].astype(np.float32)
    links = fx["links"]
    rows, row_ptr, node_count = s3grl_amd.precompute(A.indptr, A.indices, X, links.T.copy(), mode="pos_plus",
                                                     num_hops=1, sign_k=2)
    lst = oracle.get_PoS_Plus_prepped_ds(links.T, 1, A, X.astype(np.float64), 1,
                                         {"sign_k": 2, "k_node_set_strategy": "intersection"}, dtype=np.float64)
    ref, ref_ptr, _ = oracle.collate_rows(lst, 2)
    assert np.array_equal(row_ptr.cpu().numpy(), ref_ptr)
    assert rel_err(rows.cpu().numpy(), ref) < TOL
    assert list(node_count.cpu().numpy()) == [len(d["nodes"]) for d in lst]


# ------------------------------------------------------------------------------------------
# every link of the remaining PoS configs against the plain-C fp64 restatement
# ------------------------------------------------------------------------------------------
def _all_links_vs_c(eng, w, mode, links_sel=None, chunk=8000):
    from oracle import c_oracle

    link_index, _ = w.split.all_links()
    G = eng.graph(w.A)
    res = eng.precompute(G, eng.features(w.X), eng.links(link_index), mode=mode, num_hops=w.num_hops,
                         sign_k=w.sign_k)
    row_ptr = res.row_ptr.cpu().numpy()
    row_nodes = res.row_nodes.cpu().numpy()
    L = link_index.shape[1]
    sel = np.arange(L) if links_sel is None else np.sort(links_sel)
    worst, worst_el, floor, floor_el = 0.0, 0.0, 0.0, 0.0
    for lo in range(0, len(sel), chunk):
        part = sel[lo:lo + chunk]
        cref, cptr, cnodes, _ = c_oracle.pos_rows(link_index[:, part], w.num_hops, w.A, w.X, w.sign_k,
                                                  plus=mode == "pos_plus")
        c32 = c_oracle.pos_rows(link_index[:, part], w.num_hops, w.A, w.X, w.sign_k, plus=mode == "pos_plus",
                                f32=True)[0]
        floor, floor_el = max(floor, rel_err(c32, cref)), max(floor_el, _elem_floor(c32, cref))
        assert np.array_equal(np.diff(cptr), np.diff(row_ptr)[part])         # rows per link: exact
        take = np.concatenate([np.arange(row_ptr[l], row_ptr[l + 1]) for l in part]) if len(part) else part
        assert np.array_equal(row_nodes[take], cnodes)                       # which rows: exact
        import torch

        got = res.rows[torch.from_numpy(take).to(res.rows.device)].cpu().numpy()
        worst = max(worst, rel_err(got, cref))
        worst_el = max(worst_el, elem_rel_err(got, cref))
    G.close()
    report_errors(f"{w.name} {mode}, {len(sel)} links vs C fp64", worst, worst_el, (floor, floor_el))
    assert worst < TOL, worst
    return res


def test_cora_pos_plus_every_link(eng):
    from s3grl_amd import workloads

    _all_links_vs_c(eng, workloads.make("cora_posplus_k3"), "pos_plus", chunk=2000)


def test_pubmed_pos_k5_every_link(eng):
    """BASELINE config 4 (per-GPU share = the whole list when run on one GPU)."""
    from s3grl_amd import workloads

    res = _all_links_vs_c(eng, workloads.make("pubmed_pos_k5"), "pos")
    assert res.rows.shape == (328000, 6, 501)


def test_collab_scale_sampled_links(eng):
    """BASELINE config 5: 235 000-node power-law graph, 1 M links, 1-hop, sign_k=3; 40 000 of the
    links (plus the 200 with the largest endpoint degrees) against the C restatement."""
    from s3grl_amd import workloads

    w = workloads.make("collab_pos_k3")
    link_index, _ = w.split.all_links()
    deg = np.diff(w.A.indptr)
    heavy = np.argsort(-(deg[link_index[0]] + deg[link_index[1]]))[:200]
    sel = np.unique(np.concatenate([np.random.default_rng(5).choice(link_index.shape[1], 40000, replace=False),
                                    heavy]))
    _all_links_vs_c(eng, w, "pos", links_sel=sel, chunk=10000)


@pytest.mark.parametrize("mode", ["pos", "pos_plus"])
def test_big_graph_two_hops_vs_c(eng, mode):
    """A 150 000-node power-law graph, two hops: the big-graph flavours of the general path (bitmaps
    of 56 KB per link in the sizing pass, four wavefronts per link there; hash-set and HBM-scratch
    classes in the link kernels; the degree order of a graph with hubs) against the C restatement."""
    from oracle import c_oracle
    from s3grl_amd import workloads

    n, e = workloads.chung_lu(150000, 420000, d_max=400, seed=11)
    A = csr_from_undirected(n, e)
    rng = np.random.default_rng(12)
    X = (rng.random((n, 24)) * (rng.random((n, 24)) < 0.4)).astype(np.float32)
    deg = np.diff(A.indptr)
    pos = e[rng.choice(len(e), 150, replace=False)]
    hub = e[np.argsort(-(deg[e[:, 0]] + deg[e[:, 1]]))[:20]]
    neg = rng.integers(0, n, size=(150, 2))
    neg = neg[neg[:, 0] != neg[:, 1]]
    links = np.concatenate([pos, hub, neg, pos[:10, ::-1]]).T
    G = eng.graph(A)
    res = eng.precompute(G, eng.features(X), eng.links(links), mode=mode, num_hops=2, sign_k=3)
    ref, ptr, nodes, _ = c_oracle.pos_rows(links, 2, A, X.astype(np.float64), 3, plus=(mode == "pos_plus"))
    np.testing.assert_array_equal(res.row_ptr.cpu().numpy(), ptr)
    np.testing.assert_array_equal(res.row_nodes.cpu().numpy(), nodes)
    got = res.rows.cpu().numpy()
    c32 = c_oracle.pos_rows(links, 2, A, X.astype(np.float64), 3, plus=(mode == "pos_plus"), f32=True)[0]
    report_errors(f"big_graph_two_hops[{mode}]", rel_err(got, ref), elem_rel_err(got, ref),
                  (rel_err(c32, ref), _elem_floor(c32, ref)))
    # subgraphs of up to 39 000 nodes: their lists are gathered in pieces (kSplitThreshold) and the
    # pieces added in f64, so the accumulation error no longer grows with the subgraph (7.7e-6 before)
    assert rel_err(got, ref) < 3e-6
    G.close()


@pytest.mark.parametrize("feat", ["dense", "packed", "sparse"])
@pytest.mark.parametrize("name,hops,mode", [("usair", 2, "pos_plus"), ("cora", 3, "pos"), ("rand300", 2, "pos_plus")])
def test_split_jobs_equal_whole_jobs(eng, monkeypatch, name, hops, mode, feat):
    """Lists longer than the split threshold are gathered in pieces and summed in f64 (csrc:
    kSplitThreshold, split_fill_kernel, combine_kernel).  With the threshold forced down to 48 entries
    (pieces of 16) nearly every job of a small fixture is split: same row pointers and row nodes, rows
    equal to the whole-job run to fp32 round-off and within the bar of the oracle — every gather
    flavour, common-neighbour pairs, folded reversed links, sign_k beyond the BFS depth."""
    import torch

    g = load_extract(name)
    n = int(g["num_nodes"])
    A = csr_from_undirected(n, g["edges"])
    rng = np.random.default_rng(21)
    X = (rng.random((n, 37)) * (rng.random((n, 37)) < 0.3)).astype(np.float32)
    links = np.concatenate([g["links"], g["links"][:6, ::-1]])
    G = eng.graph(A)
    f = eng.features(X, mode=feat)
    L = eng.links(links.T)
    for K in (2, 4):
        monkeypatch.setenv("S3GRL_SPLIT_T", "0")
        whole = eng.precompute(G, f, L, mode=mode, num_hops=hops, sign_k=K)
        monkeypatch.setenv("S3GRL_SPLIT_T", "48")
        monkeypatch.setenv("S3GRL_SPLIT_SEG_SHIFT", "4")
        split = eng.precompute(G, f, L, mode=mode, num_hops=hops, sign_k=K)
        monkeypatch.delenv("S3GRL_SPLIT_T")
        monkeypatch.delenv("S3GRL_SPLIT_SEG_SHIFT")
        assert split.stats["max_nodes"] > 48                 # the split path really ran
        assert torch.equal(whole.row_ptr, split.row_ptr) and torch.equal(whole.row_nodes, split.row_nodes)
        a, b = whole.rows.cpu().numpy(), split.rows.cpu().numpy()
        assert rel_err(b, a) < 2e-6
        kw = {"sign_k": K, "k_node_set_strategy": "intersection"}
        fn = oracle.get_PoS_prepped_ds if mode == "pos" else oracle.get_PoS_Plus_prepped_ds
        ref, ptr, _ = oracle.collate_rows(fn(links.T, hops, A, X.astype(np.float64), 1, kw, dtype=np.float64), K)
        np.testing.assert_array_equal(split.row_ptr.cpu().numpy(), ptr)
        assert rel_err(b, ref) < TOL
    G.close()


def test_split_jobs_on_the_one_hop_path(eng, monkeypatch):
    """The same through link_full_kernel (one-hop plans on big graphs lay their coefficients out in
    pieces too)."""
    import torch

    g = load_extract("usair")
    n = int(g["num_nodes"])
    A = csr_from_undirected(n, g["edges"])
    X = np.random.default_rng(4).standard_normal((n, 21)).astype(np.float32)
    G_plain = eng.graph(A)
    L = eng.links(np.concatenate([g["links"], g["links"][:5, ::-1]]).T)
    monkeypatch.setenv("S3GRL_FORCE_ONEHOP", "1")
    G = eng.graph(A)
    f = eng.features(X)
    monkeypatch.setenv("S3GRL_SPLIT_T", "0")
    whole = eng.precompute(G, f, L, mode="pos_plus", num_hops=1, sign_k=3)
    monkeypatch.setenv("S3GRL_SPLIT_T", "32")
    monkeypatch.setenv("S3GRL_SPLIT_SEG_SHIFT", "4")
    split = eng.precompute(G, f, L, mode="pos_plus", num_hops=1, sign_k=3)
    for k in ("S3GRL_SPLIT_T", "S3GRL_SPLIT_SEG_SHIFT", "S3GRL_FORCE_ONEHOP"):
        monkeypatch.delenv(k)
    assert split.stats["max_nodes"] > 32
    assert torch.equal(whole.row_ptr, split.row_ptr) and torch.equal(whole.row_nodes, split.row_nodes)
    assert rel_err(split.rows.cpu().numpy(), whole.rows.cpu().numpy()) < 2e-6
    G.close()
    G_plain.close()


@pytest.mark.parametrize("K", [1, 2, 3, 4])
def test_sop_on_a_multigraph(eng, K):
    """SoP's global operator counts duplicate entries of the caller's edge_index (the reference builds
    it from the uncoalesced SparseTensor, sgrl_link_pred.py:161-172).  The drop-in learns how often a
    pair is listed from the entries of `powers_of_A[0]` (a stand-in exposing `nnz()` here), never from
    A's data alone; against the oracle's operator built from the edge list itself.  The same A.data
    as summed WEIGHTS of a coalesced edge_index (`use_coalesce`, :102-105) gives the unweighted
    operator."""
    import scipy.sparse as ssp
    import torch
    from s3grl_amd.tuned_SIGN import OptimizedSignOperations as ops, clear_cache

    g = load_extract("usair")
    n = int(g["num_nodes"])
    e = g["edges"].astype(np.int64)
    rng = np.random.default_rng(K)
    extra = e[rng.choice(len(e), 300, replace=True)]            # some pairs twice, some three times
    both = np.concatenate([e, extra])
    ei = np.concatenate([both, both[:, ::-1]]).T
    A = ssp.csr_matrix((np.ones(ei.shape[1], dtype=np.int64), (ei[0], ei[1])), shape=(n, n))
    assert A.data.max() >= 2
    X = rng.standard_normal((n, 7)).astype(np.float32)
    links = g["links"].T
    from s3grl_amd.dataset import GlobalOperators

    lst = ops.get_SoP_prepped_ds(GlobalOperators(K, ei.shape[1]), torch.from_numpy(links), A,
                                 torch.from_numpy(X), 1)
    P = oracle.global_normalized_powers(A, K, np.float64, edge_index=ei)
    ref = oracle.get_SoP_prepped_ds(P, links, A, X.astype(np.float64), 1, dtype=np.float64)
    for i in range(links.shape[1]):
        for k in ["x"] + [f"x{j}" for j in range(1, K + 1)]:
            assert rel_err(lst[i][k].numpy(), ref[i][k]) < TOL
    # the multiplicity matters: the coalesced operator gives other numbers
    plain = oracle.get_SoP_prepped_ds(oracle.global_normalized_powers(A, K, np.float64), links, A,
                                      X.astype(np.float64), 1, dtype=np.float64)
    assert max(rel_err(lst[i]["x1"].numpy(), plain[i]["x1"]) for i in range(links.shape[1])) > 1e-3
    # coalesced edge_index, integer weights: one entry per pair -> the unweighted operator
    for standin in (GlobalOperators(K, A.nnz), [None] * K):
        lw = ops.get_SoP_prepped_ds(standin, torch.from_numpy(links), A, torch.from_numpy(X), 1)
        for i in range(links.shape[1]):
            for k in ["x"] + [f"x{j}" for j in range(1, K + 1)]:
                assert rel_err(lw[i][k].numpy(), plain[i][k]) < TOL
    clear_cache()


@pytest.mark.parametrize("name,hops,mode", [("usair", 2, "pos_plus"), ("cora", 3, "pos"), ("rand300", 3, "pos_plus")])
def test_bitmaps_in_hbm_change_nothing(eng, monkeypatch, name, hops, mode):
    """Graphs of more than ~327 680 nodes do not fit their N-bit bitmaps into a CU's LDS: the sizing
    pass and the link kernel of the HBM-scratch class then keep them in HBM slices, processed in
    chunks.  S3GRL_FORCE_EXT_BITMAPS sends a small fixture down that road: node lists, distances,
    row nodes and statistics equal exactly, rows to fp32 round-off (the walk follows the caller's id
    order there), sampled plans included."""
    import torch

    g = load_extract(name)
    n = int(g["num_nodes"])
    A = csr_from_undirected(n, g["edges"])
    X = np.random.default_rng(8).standard_normal((n, 13))
    links = np.concatenate([g["links"], g["links"][:4, ::-1]])
    G = eng.graph(A)
    f = eng.features(X)
    L = eng.links(links.T)
    for kw in ({}, {"ratio_per_hop": 0.7, "max_nodes_per_hop": 30, "seed": 4}):
        out = []
        for ext in (False, True):
            if ext:
                monkeypatch.setenv("S3GRL_FORCE_EXT_BITMAPS", "1")
            else:
                monkeypatch.delenv("S3GRL_FORCE_EXT_BITMAPS", raising=False)
            plan = eng.plan(G, L, mode=mode, num_hops=hops, sign_k=3, full_stats=True, **kw)
            exp = [t.clone() for t in plan.export_subgraphs()]
            st = dict(plan.stats)
            st.pop("workspace_bytes", None)
            rows = plan.run(f).clone()
            out.append((exp, st, rows, plan.row_ptr().clone(), plan.row_nodes().clone()))
            plan.close()
        monkeypatch.delenv("S3GRL_FORCE_EXT_BITMAPS", raising=False)
        (ea, sa, ra, pa, na), (eb, sb, rb, pb, nb) = out
        assert all(torch.equal(x, y) for x, y in zip(ea, eb))
        assert sa == sb and torch.equal(pa, pb) and torch.equal(na, nb)
        assert rel_err(rb.cpu().numpy(), ra.cpu().numpy()) < 3e-6
    G.close()


@pytest.mark.parametrize("feat", ["dense", "packed"])
@pytest.mark.parametrize("name,hops,mode", [("usair", 1, "pos"), ("usair", 2, "pos_plus"), ("cora", 3, "pos"),
                                            ("rand300", 2, "pos_plus")])
def test_hub_processing_order_changes_no_bit(eng, monkeypatch, name, hops, mode, feat):
    """On big graphs a plan works on its links in the order of their higher-degree endpoint
    (s3grl_relabel.hip launch_link_order: the sizing pass, the class lists of the link kernels, the
    gather's job order) so that neighbouring workgroups read the same hub rows.  Only the order of the
    WORK changes: every output — rows, row nodes, node lists, statistics — is bit for bit the
    list-order plan's.  S3GRL_HUB_ORDER forces either road on a small fixture; split jobs included."""
    import torch

    g = load_extract(name)
    n = int(g["num_nodes"])
    A = csr_from_undirected(n, g["edges"])
    X = np.random.default_rng(8).standard_normal((n, 70))
    links = np.concatenate([g["links"], g["links"][:6, ::-1]])
    G = eng.graph(A)
    f = eng.features(X, feat)
    L = eng.links(links.T)
    for split in (False, True):
        if split:
            monkeypatch.setenv("S3GRL_SPLIT_T", "32")
            monkeypatch.setenv("S3GRL_SPLIT_SEG_SHIFT", "4")
        out = []
        for hub in ("0", "1"):
            monkeypatch.setenv("S3GRL_HUB_ORDER", hub)
            plan = eng.plan(G, L, mode=mode, num_hops=hops, sign_k=3, full_stats=not split)
            exp = [t.clone() for t in plan.export_subgraphs()]
            st = dict(plan.stats)
            st.pop("workspace_bytes", None)
            out.append((exp, st, plan.run(f).clone(), plan.row_ptr().clone(), plan.row_nodes().clone()))
            plan.close()
        (ea, sa, ra, pa, na), (eb, sb, rb, pb, nb) = out
        assert all(torch.equal(x, y) for x, y in zip(ea, eb))
        assert sa == sb and torch.equal(pa, pb) and torch.equal(na, nb)
        assert torch.equal(ra, rb)
    f.close()
    G.close()


def test_many_hub_rows_are_summed_the_same_way_in_every_plan(eng):
    """Rows of more than 8·G stored neighbours are summed by a wavefront each (walk_rows).  They used to
    be collected in a list of 255 — which rows got a slot, and with it the last bit of the others' sums,
    depended on the order of an atomic once a subgraph held more (36 links of the collab-scale list
    differed between two plans of the same list).  Now every such row is marked in a bitmap.  Two-hop
    subgraphs around the hubs of a power-law graph (hundreds of hub rows each): the same bits from two
    plans, from a plan of a part of the list, and — one-hop, the long rows of link_full_kernel — too."""
    import torch
    from oracle import c_oracle
    from s3grl_amd import workloads

    n, e = workloads.chung_lu(30000, 150000, seed=5)
    A = workloads.csr_from_undirected(n, e)
    deg = np.diff(A.indptr)
    hubs = np.argsort(-deg)[:10]
    assert deg[hubs[0]] > 256 and (deg > 64).sum() > 300
    rng = np.random.default_rng(1)
    links = np.array([(int(a), int(b)) for i, a in enumerate(hubs) for b in hubs[i + 1:i + 4]] +
                     [(int(h), int(v)) for h in hubs for v in rng.integers(0, n, 3) if v != h])
    X = rng.standard_normal((n, 16)).astype(np.float32)
    G = eng.graph(A)
    f = eng.features(X)
    for hops in (2, 1):
        outs = []
        for sel in (slice(None), slice(None), slice(0, len(links) // 2)):
            p = eng.plan(G, eng.links(links[sel].T), mode="pos", num_hops=hops, sign_k=3)
            outs.append(p.run(f).clone())
            big = p.stats["max_nodes"]
            p.close()
        assert torch.equal(outs[0], outs[1])
        assert torch.equal(outs[0][:outs[2].shape[0]], outs[2])
        ref, _, _, _ = c_oracle.pos_rows(links.T, hops, A, X, 3)
        assert rel_err(outs[0].cpu().numpy(), ref) < TOL, (hops, big)
    f.close(), G.close()


@pytest.mark.parametrize("name", ["usair", "cora", "rand300", "star_iso"])
def test_sizing_pass_from_cached_balls_equals_the_bfs(eng, monkeypatch, name):
    """Plain multi-hop plans size their links by bitmap arithmetic on the cached BFS balls of the two
    endpoints (csrc/s3grl_balls.hip) instead of a BFS per link (count_kernel): every output of the plan — node
    lists, distances, row pointers and nodes, statistics, rows — is bit for bit the BFS plan's; the cache
    grows level by level (2 hops, then 3 on the same graph), and a graph whose balls pass the memory cap
    keeps the BFS."""
    import torch

    g = load_extract(name)
    n = int(g["num_nodes"])
    A = csr_from_undirected(n, g["edges"])
    X = np.random.default_rng(11).standard_normal((n, 9))
    links = np.concatenate([g["links"], g["links"][:5, ::-1]])
    f = eng.features(X)
    L = eng.links(links.T)
    outs = {}
    for flavour in ("balls", "bfs", "capped"):
        for k in ("S3GRL_NO_BALL_CACHE", "S3GRL_BALL_CACHE_BYTES"):
            monkeypatch.delenv(k, raising=False)
        if flavour == "bfs":
            monkeypatch.setenv("S3GRL_NO_BALL_CACHE", "1")
        if flavour == "capped":
            monkeypatch.setenv("S3GRL_BALL_CACHE_BYTES", "64")
        G = eng.graph(A)
        res = []
        for hops, mode, K in ((2, "pos", 3), (3, "pos_plus", 2), (1, "pos", 2), (3, "pos", 5)):
            p = eng.plan(G, L, mode=mode, num_hops=hops, sign_k=K, full_stats=(hops == 2))
            st = dict(p.stats)
            st.pop("workspace_bytes")
            res.append(([t.clone() for t in p.export_subgraphs()], p.row_ptr().clone(), p.row_nodes().clone(), st,
                        p.run(f).clone()))
            p.close()
        outs[flavour] = res
        G.close()
    for other in ("bfs", "capped"):
        for (ea, pa, na, sa, ra), (eb, pb, nb, sb, rb) in zip(outs["balls"], outs[other]):
            assert all(torch.equal(x, y) for x, y in zip(ea, eb))
            assert torch.equal(pa, pb) and torch.equal(na, nb) and sa == sb
            assert torch.equal(ra, rb)
    f.close()


@pytest.mark.parametrize("mode", ["pos", "pos_plus"])
def test_half_million_nodes_two_hops_vs_c(eng, mode):
    """A 500 000-node power-law graph, two hops: beyond the LDS bitmap limit.  The sizing pass keeps
    its bitmaps in HBM; the links run on the hash flavour where their subgraph fits and on the
    HBM-scratch class with external bitmaps where it does not (hub links of tens of thousands of
    nodes) — against the C restatement."""
    from oracle import c_oracle
    from s3grl_amd import workloads

    n, e = workloads.chung_lu(500000, 1400000, d_max=400, seed=21)
    A = csr_from_undirected(n, e)
    rng = np.random.default_rng(22)
    X = (rng.random((n, 16)) * (rng.random((n, 16)) < 0.4)).astype(np.float32)
    deg = np.diff(A.indptr)
    pos = e[rng.choice(len(e), 120, replace=False)]
    hub = e[np.argsort(-(deg[e[:, 0]] + deg[e[:, 1]]))[:12]]
    neg = rng.integers(0, n, size=(120, 2))
    neg = neg[neg[:, 0] != neg[:, 1]]
    links = np.concatenate([pos, hub, neg, pos[:8, ::-1]]).T
    G = eng.graph(A)
    res = eng.precompute(G, eng.features(X), eng.links(links), mode=mode, num_hops=2, sign_k=3)
    ref, ptr, nodes, _ = c_oracle.pos_rows(links, 2, A, X.astype(np.float64), 3, plus=(mode == "pos_plus"))
    np.testing.assert_array_equal(res.row_ptr.cpu().numpy(), ptr)
    np.testing.assert_array_equal(res.row_nodes.cpu().numpy(), nodes)
    got = res.rows.cpu().numpy()
    c32 = c_oracle.pos_rows(links, 2, A, X.astype(np.float64), 3, plus=(mode == "pos_plus"), f32=True)[0]
    report_errors(f"half_million_nodes_two_hops[{mode}]", rel_err(got, ref), elem_rel_err(got, ref),
                  (rel_err(c32, ref), _elem_floor(c32, ref)))
    assert res.stats["max_nodes"] > 5000 and rel_err(got, ref) < 3e-6
    G.close()


# ------------------------------------------------------------------------------------------
# per-hop sampling (reference utils.py:66-70: ratio_per_hop, max_nodes_per_hop)
# ------------------------------------------------------------------------------------------
@pytest.mark.parametrize("name", SAMPLED_NAMES)
def test_sampled_extraction_vs_reference_fixture(eng, name):
    """Node sets per hop and CN rows, bit-exact against what the reference's own k_hop_subgraph
    produced with the keyed draw (tests/golden/sampled_*.npz)."""
    g = load_sampled(name)
    n, h, seed = int(g["num_nodes"]), int(g["num_hops"]), int(g["seed"])
    G = eng.graph(csr_from_undirected(n, g["edges"]))
    L = eng.links(g["links"].T)
    for si, (ratio, cap) in enumerate(zip(g["ratio"], g["max_nodes"])):
        cap = None if cap < 0 else int(cap)
        plan = eng.plan(G, L, mode="pos_plus", num_hops=h, sign_k=2, full_stats=True,
                        ratio_per_hop=float(ratio), max_nodes_per_hop=cap, seed=seed)
        node_ptr, nodes, dists = (t.cpu().numpy() for t in plan.export_subgraphs())
        row_ptr = plan.row_ptr().cpu().numpy()
        row_nodes = plan.row_nodes().cpu().numpy()
        for li in range(len(g["links"])):
            mine = nodes[node_ptr[li]:node_ptr[li + 1]]
            md = dists[node_ptr[li]:node_ptr[li + 1]]
            order = np.lexsort((mine, md))
            np.testing.assert_array_equal(mine[order], _ragged(g, f"s{si}_nodes", li))
            np.testing.assert_array_equal(md[order], _ragged(g, f"s{si}_dists", li))
            np.testing.assert_array_equal(row_nodes[row_ptr[li] + 2:row_ptr[li + 1]], _ragged(g, f"s{si}_cn", li))
        plan.close()
    G.close()


@pytest.mark.parametrize("mode", ["pos", "pos_plus"])
@pytest.mark.parametrize("ratio,cap", [(0.5, None), (1.0, 7), (0.8, 12)])
def test_sampled_diffusion_vs_oracle(eng, mode, ratio, cap):
    g = load_extract("rand300")
    n = int(g["num_nodes"])
    A = csr_from_undirected(n, g["edges"])
    X = np.random.default_rng(3).standard_normal((n, 11))
    links = np.concatenate([g["links"], g["links"][:10, ::-1]])          # + reversed duplicates (folded)
    G = eng.graph(A)
    res = eng.precompute(G, eng.features(X), eng.links(links.T), mode=mode, num_hops=2, sign_k=3,
                         ratio_per_hop=ratio, max_nodes_per_hop=cap, seed=99)
    fn = oracle.get_PoS_prepped_ds if mode == "pos" else oracle.get_PoS_Plus_prepped_ds
    lst = fn(links.T, 2, A, X.astype(np.float32).astype(np.float64), 1,
             {"sign_k": 3, "k_node_set_strategy": "intersection"}, dtype=np.float64,
             ratio_per_hop=ratio, max_nodes_per_hop=cap, sample_seed=99)
    ref, ref_ptr, _ = oracle.collate_rows(lst, 3)
    np.testing.assert_array_equal(res.row_ptr.cpu().numpy(), ref_ptr)
    assert rel_err(res.rows.cpu().numpy(), ref) < TOL
    assert res.stats["folded_links"] == 10
    full = eng.precompute(G, eng.features(X), eng.links(links.T), mode=mode, num_hops=2, sign_k=3)
    assert res.stats["total_nodes"] < full.stats["total_nodes"]
    # another seed draws other nodes
    other = eng.precompute(G, eng.features(X), eng.links(links.T), mode=mode, num_hops=2, sign_k=3,
                           ratio_per_hop=ratio, max_nodes_per_hop=cap, seed=100)
    assert not np.array_equal(other.rows.cpu().numpy(), res.rows.cpu().numpy())
    G.close()


def test_sampling_through_the_dropin_api_and_hbm_scratch(eng, monkeypatch):
    import torch
    from s3grl_amd import tuned_SIGN

    g = load_extract("usair")
    n = int(g["num_nodes"])
    A = csr_from_undirected(n, g["edges"])
    X = torch.from_numpy(np.random.default_rng(4).standard_normal((n, 6)).astype(np.float32))
    link_index = torch.from_numpy(g["links"].T.copy())
    kw = {"sign_k": 2, "k_node_set_strategy": "intersection"}
    monkeypatch.setattr(tuned_SIGN, "SAMPLING_SEED", 21)
    for budget in (None, "2048"):                    # LDS classes, then the HBM-scratch class
        if budget:
            monkeypatch.setenv("S3GRL_LDS_BUDGET", budget)
        lst = tuned_SIGN.OptimizedSignOperations.get_PoS_Plus_prepped_ds(link_index, 2, A, 0.3, 40, False, None,
                                                                         X, 1, kw, None)
        ref = oracle.get_PoS_Plus_prepped_ds(link_index.numpy(), 2, A, X.numpy().astype(np.float64), 1, kw,
                                             dtype=np.float64, ratio_per_hop=0.3, max_nodes_per_hop=40,
                                             sample_seed=21)
        for d, r in zip(lst, ref):
            assert d.x.shape == r["x"].shape
            for k in ("x", "x1", "x2"):
                assert rel_err(d[k].numpy(), r[k]) < TOL
    tuned_SIGN.clear_cache()


def test_sampling_argument_errors(eng):
    G = eng.graph(csr_from_undirected(5, [[0, 2], [1, 2], [0, 3], [1, 3], [3, 4]]))
    L = eng.links(np.array([[0], [1]]))
    with pytest.raises(ValueError):
        eng.plan(G, L, num_hops=1, sign_k=1, ratio_per_hop=0.0)
    with pytest.raises(ValueError):
        eng.plan(G, L, num_hops=1, sign_k=1, max_nodes_per_hop=0)
    # a ratio small enough to empty the first hop ends the walk there: the subgraph is {src,dst}
    p = eng.plan(G, L, num_hops=2, sign_k=1, ratio_per_hop=0.1, full_stats=True)
    assert p.stats["total_nodes"] == 2
    p.close()
    G.close()


@pytest.mark.parametrize("K", [1, 2, 3, 4])
@pytest.mark.parametrize("name", ["probe5", "triangle", "pair", "star_iso"])
def test_packed_gather_on_tiny_subgraphs(eng, name, K):
    """The packed-row gather's phase logic on supports of 2..7 rows (no full group of 4 rows, the
    prefix limits inside the tail, empty phases), hops 1..3, PoS and PoS Plus."""
    g = load_extract(name)
    n = int(g["num_nodes"])
    A = csr_from_undirected(n, g["edges"])
    rng = np.random.default_rng(K)
    X = (rng.standard_normal((n, 9)) * (rng.random((n, 9)) < 0.4)).astype(np.float32)
    G = eng.graph(A)
    f = eng.features(X, "packed")
    assert f.is_packed
    kw = {"sign_k": K, "k_node_set_strategy": "intersection"}
    for h in (1, 2, 3):
        for mode, fn in (("pos", oracle.get_PoS_prepped_ds), ("pos_plus", oracle.get_PoS_Plus_prepped_ds)):
            res = eng.precompute(G, f, eng.links(g["links"].T), mode=mode, num_hops=h, sign_k=K)
            ref, ptr, _ = oracle.collate_rows(fn(g["links"].T, h, A, X.astype(np.float64), 1, kw, dtype=np.float64), K)
            np.testing.assert_array_equal(res.row_ptr.cpu().numpy(), ptr)
            assert rel_err(res.rows.cpu().numpy(), ref) < TOL, (name, K, h, mode)
    G.close()



@pytest.mark.parametrize("onehop", [False, True])
def test_many_common_neighbours(eng, monkeypatch, onehop):
    """A dense graph: pairs with 100-200 common neighbours (more than a wavefront: the rank sort that
    puts the common-neighbour rows into the caller's id order runs in several rounds), general path
    and one-hop path, against the C restatement."""
    from oracle import c_oracle

    rng = np.random.default_rng(41)
    n = 420
    M = np.triu(rng.random((n, n)) < 0.55, 1)
    e = np.argwhere(M)
    A = csr_from_undirected(n, e)
    X = rng.random((n, 7)).astype(np.float32)
    links = np.concatenate([e[rng.choice(len(e), 24, replace=False)],
                            np.argwhere(~M & np.triu(np.ones((n, n), bool), 1))[:12]]).T
    ref, ptr, nodes, _ = c_oracle.pos_rows(links, 1, A, X.astype(np.float64), 2, plus=True)
    assert np.diff(ptr).max() > 66          # more common neighbours than lanes
    if onehop:
        monkeypatch.setenv("S3GRL_FORCE_ONEHOP", "1")
        monkeypatch.setenv("S3GRL_FORCE_HASH", "1")
    G = eng.graph(A)
    res = eng.precompute(G, eng.features(X), eng.links(links), mode="pos_plus", num_hops=1, sign_k=2)
    np.testing.assert_array_equal(res.row_ptr.cpu().numpy(), ptr)
    np.testing.assert_array_equal(res.row_nodes.cpu().numpy(), nodes)
    assert rel_err(res.rows.cpu().numpy(), ref) < TOL
    G.close()
    monkeypatch.delenv("S3GRL_FORCE_ONEHOP", raising=False)
    monkeypatch.delenv("S3GRL_FORCE_HASH", raising=False)


@pytest.mark.parametrize("K", [1, 2, 4, 6, 7, 8])
@pytest.mark.parametrize("hops", [1, 2])
def test_every_sign_k_up_to_the_limit(eng, monkeypatch, K, hops):
    """sign_k = 1..8 are separate kernel instantiations (spread over four translation units): each
    against the C restatement, on the general path and on the one-hop path of big graphs."""
    from oracle import c_oracle

    g = load_extract("rand300")
    n = int(g["num_nodes"])
    A = csr_from_undirected(n, g["edges"])
    X = np.abs(np.random.default_rng(K).standard_normal((n, 9))).astype(np.float32)
    links = g["links"][:40].T
    ref, ptr, nodes, _ = c_oracle.pos_rows(links, hops, A, X.astype(np.float64), K, plus=True)
    for onehop in ([False, True] if hops == 1 else [False]):
        if onehop:
            monkeypatch.setenv("S3GRL_FORCE_ONEHOP", "1")
            monkeypatch.setenv("S3GRL_FORCE_HASH", "1")
        G = eng.graph(A)
        res = eng.precompute(G, eng.features(X), eng.links(links), mode="pos_plus", num_hops=hops, sign_k=K)
        np.testing.assert_array_equal(res.row_ptr.cpu().numpy(), ptr)
        np.testing.assert_array_equal(res.row_nodes.cpu().numpy(), nodes)
        assert rel_err(res.rows.cpu().numpy(), ref) < TOL
        G.close()
    monkeypatch.delenv("S3GRL_FORCE_ONEHOP", raising=False)
    monkeypatch.delenv("S3GRL_FORCE_HASH", raising=False)


@pytest.mark.parametrize("name,hops", [("rand300", 2), ("cora", 3), ("usair", 2), ("star_iso", 2)])
@pytest.mark.parametrize("mode", ["pos", "pos_plus"])
def test_degree_order_is_invisible(eng, monkeypatch, name, hops, mode):
    """Plain multi-hop plans walk the graph with its ids in descending degree order
    (csrc/s3grl_relabel.hip); S3GRL_NO_RELABEL walks the caller's order.  Everything a plan hands out
    is in the caller's ids either way: row pointers, row nodes (common-neighbour rows ascending),
    exported node lists (hop-major, ascending inside a hop), distances and statistics are equal
    exactly, the rows to fp32 round-off (another summation order)."""
    import torch

    g = load_extract(name)
    n = int(g["num_nodes"])
    A = csr_from_undirected(n, g["edges"])
    X = np.random.default_rng(5).standard_normal((n, 33))
    links = np.concatenate([g["links"], g["links"][:4, ::-1]])
    G = eng.graph(A)
    f = eng.features(X)
    L = eng.links(links.T)
    out = []
    for off in (False, True):
        if off:
            monkeypatch.setenv("S3GRL_NO_RELABEL", "1")
        else:
            monkeypatch.delenv("S3GRL_NO_RELABEL", raising=False)
        plan = eng.plan(G, L, mode=mode, num_hops=hops, sign_k=3, full_stats=True)
        exp = [t.clone() for t in plan.export_subgraphs()]
        st = dict(plan.stats)
        st.pop("workspace_bytes", None)
        plan.close()
        res = eng.precompute(G, f, L, mode=mode, num_hops=hops, sign_k=3)
        out.append((exp, st, res.rows.clone(), res.row_ptr.clone(), res.row_nodes.clone()))
    monkeypatch.delenv("S3GRL_NO_RELABEL", raising=False)
    (ea, sa, ra, pa, na), (eb, sb, rb, pb, nb) = out
    assert all(torch.equal(x, y) for x, y in zip(ea, eb))
    assert sa == sb
    assert torch.equal(pa, pb) and torch.equal(na, nb)
    assert rel_err(ra.cpu().numpy(), rb.cpu().numpy()) < 3e-6
    G.close()


@pytest.mark.parametrize("name,hops", [("rand300", 2), ("cora", 3), ("usair", 1)])
def test_export_segmented_sort_flavour(eng, monkeypatch, name, hops):
    """s3grl_plan_export_subgraphs restores "ascending id inside a hop" through an N-bit LDS bitmap
    on small graphs and through a segmented radix sort (independent of num_nodes) on graphs whose
    bitmap passes 64 KiB; S3GRL_FORCE_SEGSORT runs the second flavour here: same lists."""
    import torch

    g = load_extract(name)
    n = int(g["num_nodes"])
    A = csr_from_undirected(n, g["edges"])
    G = eng.graph(A)
    L = eng.links(g["links"].T)
    plan = eng.plan(G, L, mode="pos", num_hops=hops, sign_k=2, full_stats=True)
    monkeypatch.delenv("S3GRL_FORCE_SEGSORT", raising=False)
    a = [t.clone() for t in plan.export_subgraphs()]
    monkeypatch.setenv("S3GRL_FORCE_SEGSORT", "1")
    b = [t.clone() for t in plan.export_subgraphs()]
    monkeypatch.delenv("S3GRL_FORCE_SEGSORT", raising=False)
    plan.close()
    assert all(torch.equal(x, y) for x, y in zip(a, b))
    ptr, nodes = a[0].cpu().numpy(), a[1].cpu().numpy()
    for li in range(len(g["links"])):
        mine = nodes[ptr[li]:ptr[li + 1]]
        np.testing.assert_array_equal(mine, _ragged(g, f"h{hops}_nodes", li))
    G.close()


@pytest.mark.parametrize("name,hops", [("rand300", 2), ("cora", 3), ("star_iso", 2)])
def test_leaf_tail_walk_changes_no_bit(eng, monkeypatch, name, hops):
    """The last hop's rows of at most two neighbours are walked with one lane per row instead of
    four (contiguous in the degree order); S3GRL_NO_LEAF_WALK walks them like the rest.  Two terms
    add up to the same bits either way."""
    import torch

    g = load_extract(name)
    n = int(g["num_nodes"])
    A = csr_from_undirected(n, g["edges"])
    X = np.random.default_rng(9).standard_normal((n, 17))
    G = eng.graph(A)
    f = eng.features(X)
    L = eng.links(g["links"].T)
    for K in (2, 3, 5):
        monkeypatch.delenv("S3GRL_NO_LEAF_WALK", raising=False)
        a = eng.precompute(G, f, L, mode="pos_plus", num_hops=hops, sign_k=K).rows.clone()
        monkeypatch.setenv("S3GRL_NO_LEAF_WALK", "1")
        b = eng.precompute(G, f, L, mode="pos_plus", num_hops=hops, sign_k=K).rows
        assert torch.equal(a, b)
    monkeypatch.delenv("S3GRL_NO_LEAF_WALK", raising=False)
    G.close()


@pytest.mark.parametrize("name,hops", [("rand300", 2), ("cora", 3), ("usair", 1)])
@pytest.mark.parametrize("mode", ["pos", "pos_plus"])
def test_direct_map_flavour_equals_bitmap_flavour(eng, monkeypatch, name, hops, mode):
    """Small graphs keep the visited set as a direct map (one uint16 per graph node = list position);
    S3GRL_NO_DM sends the same links through the three-bitmap flavour.  Same rows walked by the same
    lanes in the same order: every output is equal bit for bit, statistics included."""
    import torch

    g = load_extract(name)
    n = int(g["num_nodes"])
    A = csr_from_undirected(n, g["edges"])
    X = np.random.default_rng(77).standard_normal((n, 21))
    links = np.concatenate([g["links"], g["links"][:4, ::-1]])
    G = eng.graph(A)
    f = eng.features(X)
    L = eng.links(links.T)
    for K in (1, 3, 5):
        monkeypatch.delenv("S3GRL_NO_DM", raising=False)
        a = eng.precompute(G, f, L, mode=mode, num_hops=hops, sign_k=K)
        a = (a.rows.clone(), a.row_ptr.clone(), a.row_nodes.clone(), dict(a.stats))
        monkeypatch.setenv("S3GRL_NO_DM", "1")
        b = eng.precompute(G, f, L, mode=mode, num_hops=hops, sign_k=K)
        assert torch.equal(a[0], b.rows) and torch.equal(a[1], b.row_ptr) and torch.equal(a[2], b.row_nodes)
        sa, sb = a[3], dict(b.stats)
        sa.pop("workspace_bytes", None), sb.pop("workspace_bytes", None)
        assert sa == sb
    monkeypatch.delenv("S3GRL_NO_DM", raising=False)
    G.close()


@pytest.mark.parametrize("flavour", ["bitmap", "hash"])
def test_node_list_handover_and_its_fallback(eng, monkeypatch, flavour):
    """count_kernel hands every link's node list to link_kernel through an HBM slot; a list longer
    than the slot (or no slot at all) makes link_kernel walk the graph again.  All three ways give
    the same bits, in both flavours of the visited set."""
    import torch

    g = load_extract("rand300")
    n = int(g["num_nodes"])
    A = csr_from_undirected(n, g["edges"])
    X = np.random.default_rng(8).standard_normal((n, 13))
    links = np.concatenate([g["links"], g["links"][:6, ::-1]])
    if flavour == "hash":
        monkeypatch.setenv("S3GRL_FORCE_HASH", "1")
    G = eng.graph(A)
    f = eng.features(X)
    L = eng.links(links.T)
    outs = []
    for slot in (None, "0", "8", "64"):
        if slot is None:
            monkeypatch.delenv("S3GRL_STASH_SLOT", raising=False)
        else:
            monkeypatch.setenv("S3GRL_STASH_SLOT", slot)
        res = eng.precompute(G, f, L, mode="pos_plus", num_hops=2, sign_k=3)
        outs.append((res.rows.clone(), res.row_ptr.clone(), res.row_nodes.clone()))
    for o in outs[1:]:
        assert all(torch.equal(a, b) for a, b in zip(o, outs[0]))
    ref, ptr, _ = oracle.collate_rows(
        oracle.get_PoS_Plus_prepped_ds(links.T, 2, A, X.astype(np.float32).astype(np.float64), 1,
                                       {"sign_k": 3, "k_node_set_strategy": "intersection"}, dtype=np.float64), 3)
    np.testing.assert_array_equal(outs[0][1].cpu().numpy(), ptr)
    assert rel_err(outs[0][0].cpu().numpy(), ref) < TOL
    G.close()


@pytest.mark.parametrize("K", [2, 3, 4])
def test_sop_bitmaps_in_hbm_change_nothing(eng, monkeypatch, K):
    """Beyond ~327 680 nodes the three N-bit bitmaps of the SoP scalar kernel do not fit a CU's LDS: they then
    live in HBM slices, the class lists run in chunks over them (like the sizing pass of PoS).
    S3GRL_FORCE_EXT_BITMAPS sends the small fixtures down that road: the same rows bit for bit."""
    import torch

    g = load_extract("usair")
    n = int(g["num_nodes"])
    A = csr_from_undirected(n, g["edges"])
    X = np.random.default_rng(3).standard_normal((n, 11))
    links = np.concatenate([g["links"], g["links"][:5, ::-1]])
    G = eng.graph(A)
    f = eng.features(X)
    L = eng.links(links.T)
    out = []
    for force in (False, True):
        if force:
            monkeypatch.setenv("S3GRL_FORCE_EXT_BITMAPS", "1")
        else:
            monkeypatch.delenv("S3GRL_FORCE_EXT_BITMAPS", raising=False)
        out.append(eng.precompute(G, f, L, mode="sop", sign_k=K).rows.clone())
    monkeypatch.delenv("S3GRL_FORCE_EXT_BITMAPS", raising=False)
    assert torch.equal(out[0], out[1])
    G.close()


@pytest.mark.parametrize("K", [2, 3])
def test_sop_on_half_a_million_nodes(eng, K):
    """SoP has no node limit any more (reference tuned_SIGN.py:49-134 has none either): a 500 000-node sparse
    graph, links at its best-connected nodes and at random, against the oracle's global powers."""
    from s3grl_amd import workloads

    rng = np.random.default_rng(21)
    n = 500000
    e = rng.integers(0, n, size=(750000, 2))
    e = e[e[:, 0] != e[:, 1]]
    e = np.unique(np.sort(e, axis=1), axis=0)
    A = workloads.csr_from_undirected(n, e)
    deg = np.diff(A.indptr)
    top = np.argsort(-deg)[:40]
    links = np.concatenate([e[rng.choice(len(e), 150, replace=False)], rng.integers(0, n, size=(100, 2)),
                            np.stack([top[:-1], top[1:]], 1)])
    links = links[links[:, 0] != links[:, 1]]
    X = rng.random((n, 6)).astype(np.float32)
    G = eng.graph(A)
    f = eng.features(X)
    res = eng.precompute(G, f, eng.links(links.T), mode="sop", sign_k=K)
    P = oracle.global_normalized_powers(A, K, np.float64)
    ref, _, _ = oracle.collate_rows(oracle.get_SoP_prepped_ds(P, links.T, A, X.astype(np.float64), 1, dtype=np.float64), K)
    err = rel_err(res.rows.cpu().numpy(), ref)
    report_errors(f"half_million_nodes_sop[K={K}]", err, err)
    assert err < TOL
    f.close(), G.close()


@pytest.mark.parametrize("name,hops,K", [("usair", 2, 3), ("usair", 1, 2), ("cora", 2, 3), ("cora", 3, 3), ("rand300", 2, 2),
                                          ("star_iso", 2, 3), ("triangle", 1, 2)])
def test_sop_restricted_to_the_k_hop_ball(eng, name, hops, K):
    """`mode="sop_restricted"` — NOT a reference flow: the optional twin SURVEY §8(d) names for BASELINE config 3
    ("2-hop subgraphs"): the SoP rows (tuned_SIGN.py:49-134) with every operator row restricted to the num_hops-ball
    of {src, dst}.  Against the oracle's twin; operators 1..num_hops equal the unrestricted SoP rows (their
    support lies inside the ball anyway); reversed duplicates come out as their primaries' rows swapped."""
    g = load_extract(name)
    n = int(g["num_nodes"])
    A = csr_from_undirected(n, g["edges"])
    X = np.random.default_rng(4).standard_normal((n, 9))
    links = np.concatenate([g["links"], g["links"][:5, ::-1]])
    G = eng.graph(A)
    f = eng.features(X)
    res = eng.precompute(G, f, eng.links(links.T), mode="sop_restricted", num_hops=hops, sign_k=K)
    P = oracle.global_normalized_powers(A, K, np.float64)
    ref, ptr, _ = oracle.collate_rows(oracle.get_SoP_restricted_ds(P, links.T, hops, A, X, 1, dtype=np.float64), K)
    assert np.array_equal(res.row_ptr.cpu().numpy(), ptr)
    got = res.rows.cpu().numpy()
    assert rel_err(got, ref) < TOL
    full = eng.precompute(G, f, eng.links(links.T), mode="sop", sign_k=K).rows.cpu().numpy()
    for i in range(0, min(hops, K) + 1):
        assert rel_err(got[:, i], full[:, i]) < TOL
    L, m = len(links), min(5, len(g["links"]))
    v = got.reshape(L, 2, K + 1, -1)
    assert np.array_equal(v[L - m:, 0], v[:m, 1]) and np.array_equal(v[L - m:, 1], v[:m, 0])
    f.close(), G.close()


def test_sop_restricted_errors(eng):
    g = load_extract("usair")
    A = csr_from_undirected(int(g["num_nodes"]), g["edges"])
    G = eng.graph(A)
    f = eng.features(np.ones((int(g["num_nodes"]), 3)))
    L = eng.links(g["links"].T)
    with pytest.raises(NotImplementedError):       # sign_k - 1 > num_hops: the rows would leave the ball earlier
        eng.precompute(G, f, L, mode="sop_restricted", num_hops=1, sign_k=3)
    with pytest.raises(NotImplementedError):
        eng.precompute(G, f, L, mode="sop_restricted", num_hops=2, sign_k=3, ratio_per_hop=0.5)
    f.close(), G.close()
