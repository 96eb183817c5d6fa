"""Seeded random sweep of the PoS / PoS Plus path against the plain-C fp64 restatement: graph
shape (uniform / power-law with hubs / with isolated nodes), size, hops, sign_k, feature width and
sparsity (dense and packed operand), visited-set flavour and LDS budget (LDS classes, HBM-scratch
class) all vary together.  Every case is deterministic."""
import numpy as np
import pytest

from conftest import csr_from_undirected
from oracle import c_oracle

pytestmark = pytest.mark.gpu

TOL = 1e-5
ATOL = 1e-10
TERMS_ULPS = 0.2    # with TOL: 2e-6 of the sum of the absolute terms (~32 ulps of fp32)


def rel_err(got, ref, terms=None):
    ref = np.asarray(ref, dtype=np.float64)
    got = np.asarray(got, dtype=np.float64)
    if not ref.size:
        return 0.0
    scale = np.maximum(np.abs(ref), np.abs(ref).max(axis=-1, keepdims=True))
    if terms is not None:
        # signed features: an element (with F = 3 a whole row) can cancel to ~0 while its terms are O(1); fp32
        # leaves round-off of the TERMS there, in the engine as in the reference's own fp32 arithmetic.
        # `terms` = the same rows computed on |X| = the sum of the absolute terms; a few ulps of it are allowed
        # (1 case in 400 of the directed sweep needs it: |Δ| = 3.8e-9 on an element of 6.1e-5).
        scale = np.maximum(scale, TERMS_ULPS * np.abs(terms))
    return float(np.max(np.clip(np.abs(got - ref) - ATOL, 0, None) / np.maximum(scale, 1e-30)))


@pytest.fixture(scope="module")
def eng():
    import torch
    from s3grl_amd.engine import Engine

    assert torch.cuda.is_available()
    e = Engine("cuda:0")
    yield e
    e.close()


def _graph(rng, kind, n):
    from s3grl_amd import workloads

    if kind == "powerlaw":
        n, e = workloads.chung_lu(n, int(n * rng.uniform(2.0, 6.0)), seed=int(rng.integers(1 << 30)))
        return n, e
    m = int(n * rng.uniform(0.8, 5.0))
    e = rng.integers(0, n if kind == "uniform" else max(2, int(n * 0.8)), size=(m, 2))
    e = e[e[:, 0] != e[:, 1]]
    e = np.unique(np.sort(e, axis=1), axis=0)
    return n, e


import os

CASES = list(range(int(os.environ.get("S3GRL_FUZZ_CASES", "36"))))   # more for a one-off soak


@pytest.mark.parametrize("case", CASES)
def test_random_configuration(eng, monkeypatch, case):
    rng = np.random.default_rng(1000 + case)
    kind = ["uniform", "powerlaw", "isolated"][case % 3]
    n = int(rng.choice([40, 300, 2500, 9000]))
    n, edges = _graph(rng, kind, n)
    A = csr_from_undirected(n, edges)
    hops = int(rng.integers(1, 4))
    K = int(rng.integers(1, 6))
    plus = bool(rng.integers(0, 2))
    F = int(rng.choice([3, 17, 64, 300, 515]))
    density = float(rng.choice([1.0, 0.3, 0.05]))
    X = (rng.standard_normal((n, F)) * (rng.random((n, F)) < density)).astype(np.float32)
    L = int(rng.integers(20, 120))
    pos = edges[rng.choice(len(edges), min(L // 2, len(edges)), replace=False)]
    neg = rng.integers(0, n, size=(L, 2))
    neg = neg[neg[:, 0] != neg[:, 1]]
    links = np.concatenate([pos, neg, pos[:5, ::-1]])                  # + a few reversed duplicates
    if case % 4 == 1:
        monkeypatch.setenv("S3GRL_FORCE_HASH", "1")
    if case % 5 == 2:
        monkeypatch.setenv("S3GRL_LDS_BUDGET", "3072")                 # pushes links into the HBM-scratch class
    if case % 7 == 3:
        monkeypatch.setenv("S3GRL_STASH_SLOT", "16")                   # most lists overflow their slot
    if case % 4 == 3:
        monkeypatch.setenv("S3GRL_NO_RELABEL", "1")                    # the caller's node order instead of the degree order
    if case % 6 == 4:
        monkeypatch.setenv("S3GRL_NO_DM", "1")                         # bitmap flavour instead of the direct map
    if case % 5 == 0:
        monkeypatch.setenv("S3GRL_SPLIT_T", "64")                      # most jobs gathered in pieces of 32
        monkeypatch.setenv("S3GRL_SPLIT_SEG_SHIFT", "5")
    if case % 9 == 5:
        monkeypatch.setenv("S3GRL_FORCE_EXT_BITMAPS", "1")             # bitmaps in HBM slices (graphs beyond the LDS limit)
    if hops == 1 and case % 2 == 0:
        monkeypatch.setenv("S3GRL_FORCE_ONEHOP", "1")                  # row-intersection path + cached hub
        monkeypatch.setenv("S3GRL_FORCE_HASH", "1")                    # neighbourhoods (s3grl_hub.hip) from degree 4
        monkeypatch.setenv("S3GRL_HUB_MIN_DEG", "4")
    if case % 2 == 1:
        monkeypatch.setenv("S3GRL_HUB_ORDER", "1")                     # big-graph processing order (by hub endpoint)
    G = eng.graph(A)
    f = eng.features(X, ["auto", "dense", "packed"][case % 3])
    res = eng.precompute(G, f, eng.links(links.T), mode="pos_plus" if plus else "pos", num_hops=hops, sign_k=K)
    ref, ptr, nodes, _ = c_oracle.pos_rows(links.T, hops, A, X, K, plus=plus)
    np.testing.assert_array_equal(res.row_ptr.cpu().numpy(), ptr)
    np.testing.assert_array_equal(res.row_nodes.cpu().numpy(), nodes)
    err = rel_err(res.rows.cpu().numpy(), ref)
    if err >= TOL:      # cancellation of signed features?  judge against the sum of the absolute terms
        err = rel_err(res.rows.cpu().numpy(), ref, c_oracle.pos_rows(links.T, hops, A, np.abs(X), K, plus=plus)[0])
    assert err < TOL, (case, kind, n, hops, K, plus, F, density, err)
    f.close()
    G.close()


@pytest.mark.parametrize("case", list(range(int(os.environ.get("S3GRL_FUZZ_CASES", "18")))))
def test_random_directed_configuration(eng, monkeypatch, case):
    """The same sweep on DIRECTED graphs (every undirected edge keeps one direction, some both),
    against the reference-structured Python restatement with directed=True / A_csc: hops, sign_k,
    PoS / PoS Plus, per-hop sampling, reversed duplicates, split jobs, HBM-scratch class."""
    import scipy.sparse as ssp

    import oracle

    rng = np.random.default_rng(9000 + case)
    kind = ["uniform", "powerlaw", "isolated"][case % 3]
    n, edges = _graph(rng, kind, int(rng.choice([30, 150, 600])))
    flip = rng.random(len(edges)) < 0.5
    arcs = np.where(flip[:, None], edges[:, ::-1], edges)
    both = edges[rng.random(len(edges)) < 0.3]
    arcs = np.unique(np.vstack([arcs, both, both[:, ::-1]]), axis=0)
    A = ssp.csr_matrix((np.ones(len(arcs), dtype=np.int64), (arcs[:, 0], arcs[:, 1])), shape=(n, n))
    A_csc = A.tocsc()
    hops = int(rng.integers(1, 4))
    K = int(rng.integers(1, 5))
    plus = bool(rng.integers(0, 2))
    F = int(rng.choice([3, 20, 70]))
    X = (rng.standard_normal((n, F)) * (rng.random((n, F)) < float(rng.choice([1.0, 0.3])))).astype(np.float32)
    pos = arcs[rng.choice(len(arcs), min(15, len(arcs)), replace=False)]
    neg = rng.integers(0, n, size=(25, 2))
    neg = neg[neg[:, 0] != neg[:, 1]]
    links = np.concatenate([pos, neg, pos[:4, ::-1]])
    smp = {}
    if case % 3 == 1:
        smp = {"ratio_per_hop": 0.6, "max_nodes_per_hop": 12, "seed": case}
    if case % 4 == 2:
        monkeypatch.setenv("S3GRL_LDS_BUDGET", "2048")
    if case % 5 == 3:
        monkeypatch.setenv("S3GRL_SPLIT_T", "32")
        monkeypatch.setenv("S3GRL_SPLIT_SEG_SHIFT", "4")
    G = eng.graph(A, directed=True, A_csc=A_csc)
    res = eng.precompute(G, eng.features(X), eng.links(links.T), mode="pos_plus" if plus else "pos", num_hops=hops,
                         sign_k=K, **smp)
    kw = {"sign_k": K, "k_node_set_strategy": "intersection"}
    fn = oracle.get_PoS_Plus_prepped_ds if plus else oracle.get_PoS_prepped_ds
    okw = {} if not smp else {"ratio_per_hop": 0.6, "max_nodes_per_hop": 12, "sample_seed": case}
    ref, ptr, _ = oracle.collate_rows(fn(links.T, hops, A, X.astype(np.float64), 1, kw, dtype=np.float64,
                                         directed=True, A_csc=A_csc, **okw), K)
    np.testing.assert_array_equal(res.row_ptr.cpu().numpy(), ptr)
    err = rel_err(res.rows.cpu().numpy(), ref)
    if err >= TOL:      # cancellation of signed features?  judge against the sum of the absolute terms
        terms, _, _ = oracle.collate_rows(fn(links.T, hops, A, np.abs(X).astype(np.float64), 1, kw, dtype=np.float64,
                                             directed=True, A_csc=A_csc, **okw), K)
        err = rel_err(res.rows.cpu().numpy(), ref, terms)
    assert err < TOL, (case, kind, n, hops, K, plus, F, err)
    G.close()


@pytest.mark.parametrize("case", list(range(16)))
def test_random_sop_configuration(eng, monkeypatch, case):
    """The same sweep for SoP, against the reference-structured Python restatement (global powers
    materialised, row with the partner's column zeroed times X): small graphs, sign_k 1..6, lists
    with reversed and exact duplicates, isolated endpoints."""
    import oracle

    rng = np.random.default_rng(5000 + case)
    kind = ["uniform", "powerlaw", "isolated"][case % 3]
    n, edges = _graph(rng, kind, int(rng.choice([30, 120, 400])))
    A = csr_from_undirected(n, edges)
    K = int(rng.integers(1, 7))
    F = int(rng.choice([1, 9, 64, 130]))
    X = (rng.random((n, F)) * (rng.random((n, F)) < float(rng.choice([1.0, 0.3])))).astype(np.float32)
    pos = edges[rng.choice(len(edges), min(25, len(edges)), replace=False)]
    neg = rng.integers(0, n, size=(40, 2))
    neg = neg[neg[:, 0] != neg[:, 1]]
    links = np.concatenate([pos, neg, pos[:8, ::-1], neg[:3]])
    links = links[rng.permutation(len(links))].T
    if case % 4 == 1:
        monkeypatch.setenv("S3GRL_NO_MIRROR", "1")
    if case % 4 == 2:
        monkeypatch.setenv("S3GRL_SOP_UNSORTED", "1")
    G = eng.graph(A)
    res = eng.precompute(G, eng.features(X), eng.links(links), mode="sop", sign_k=K)
    ref, _, _ = oracle.collate_rows(
        oracle.get_SoP_prepped_ds(oracle.global_normalized_powers(A, K, np.float64), links, A,
                                  X.astype(np.float64), 1, dtype=np.float64), K)
    assert rel_err(res.rows.cpu().numpy(), ref) < TOL, (case, kind, n, K, F)
    G.close()
