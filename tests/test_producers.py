"""Producers either side of the hot path: the 85/5/10 edge split, the feature
initialisers and the SEAL txt reader.  Host-side numpy, pinned here to the reference's semantics
(utils.py:588-659, sgrl_link_pred.py:851,961-963,1000-1003, data_utils.py:76-93)."""
import numpy as np
import pytest

from conftest import GOLDEN
from s3grl_amd import workloads as W


# ---- SEAL txt reader: data_utils.py:76-93 -------------------------------------------------------
def test_seal_reader_matches_reference_run():
    """tests/golden/seal_usair.npz = the reference's own read_label / read_edges executed on the
    same file (tests/golden/make_seal_reader_golden.py)."""
    gold = np.load(GOLDEN / "seal_usair.npz")
    n, edges = W.read_seal_edges(GOLDEN / "usair_edges.txt")
    assert n == len(gold["names"]) == 332
    assert np.array_equal(edges, gold["edges"])
    # ids are ranks in the sorted list of STRINGS, not of integers
    names = [str(s) for s in gold["names"]]
    assert names == sorted(names) and names != sorted(names, key=int)
    assert np.array_equal(gold["ids"], np.arange(n))
    # the simple-graph topology of it is the committed USAir topology the benches use
    n0, e0 = W.load_topology("usair")
    assert n0 == n and np.array_equal(W.undirected_unique(edges), e0)


def test_seal_reader_takes_directory_and_ignores_extra_columns(tmp_path):
    (tmp_path / "edges.txt").write_text("b a 0.5\n10 2 1.0 extra\n2 b\n")
    n, e = W.read_seal_edges(tmp_path)
    # sorted strings: '10' < '2' < 'a' < 'b'
    assert n == 4 and e.tolist() == [[3, 2], [0, 1], [1, 3]]


# ---- edge split: utils.py:588-634 (train_test_split_edges + negative_sampling), :637-659 ----------
@pytest.mark.parametrize("name,seed", [("usair", 0), ("cora", 1)])
def test_edge_split_semantics(name, seed):
    n, e = W.load_topology(name)
    sp = W.edge_split(n, e, seed=seed)
    E = len(e)
    n_v, n_t = int(np.floor(0.05 * E)), int(np.floor(0.1 * E))
    pos_tr, neg_tr = sp.links["train"]
    pos_v, neg_v = sp.links["valid"]
    pos_t, neg_t = sp.links["test"]
    # sizes: floor(5 %), floor(10 %), the rest; train positives hold BOTH directions
    assert pos_v.shape == (2, n_v) and pos_t.shape == (2, n_t)
    assert pos_tr.shape == (2, 2 * (E - n_v - n_t))
    key = lambda a: a[0] * n + a[1]                                    # noqa: E731
    ktr = key(pos_tr)
    assert np.array_equal(np.sort(ktr), ktr), "train positives are in coalesced (row, col) order"
    assert len(np.unique(ktr)) == len(ktr)
    assert np.array_equal(np.sort(key(pos_tr[::-1])), ktr), "every train edge appears in both directions"
    # the three positive sets partition the undirected edge set
    und = lambda a: np.minimum(a[0], a[1]) * n + np.maximum(a[0], a[1])  # noqa: E731
    allk = np.concatenate([np.unique(und(pos_tr)), und(pos_v), und(pos_t)])
    assert len(allk) == E and np.array_equal(np.sort(allk), np.sort(e[:, 0] * n + e[:, 1]))
    # val / test positives are stored once, row < col (train_test_split_edges works on the upper half)
    assert (pos_v[0] < pos_v[1]).all() and (pos_t[0] < pos_t[1]).all()
    # as many negatives as positives per split (neg_ratio = 1)
    assert neg_tr.shape == pos_tr.shape and neg_v.shape == pos_v.shape and neg_t.shape == pos_t.shape
    # val / test negatives: non-edges of the FULL graph, upper half, distinct
    full = set((e[:, 0] * n + e[:, 1]).tolist())
    kvt = np.concatenate([key(neg_v), key(neg_t)])
    assert (np.concatenate([neg_v, neg_t], 1)[0] < np.concatenate([neg_v, neg_t], 1)[1]).all()
    assert len(np.unique(kvt)) == len(kvt) and not (set(kvt.tolist()) & full)
    # train negatives: negative_sampling on the TRAIN edges + self-loops -> directed non-edges of the
    # train graph, no self-loops (they may coincide with a held-out positive, like the reference's)
    assert (neg_tr[0] != neg_tr[1]).all()
    assert not (set(key(neg_tr).tolist()) & set(ktr.tolist()))
    # train graph = train edges only, both directions, int64 ones (sgrl_link_pred.py:107-114,852-855)
    A = sp.A
    assert A.shape == (n, n) and A.dtype == np.int64 and A.nnz == pos_tr.shape[1]
    assert (A != A.T).nnz == 0 and (A.data == 1).all()
    coo = A.tocoo()
    assert np.array_equal(np.sort(coo.row * n + coo.col), ktr)
    for held in (pos_v, pos_t):
        assert A[held[0], held[1]].sum() == 0, "held-out positives are not in the train graph"
    # the 6 operator calls: positives then negatives, per split (sgrl_link_pred.py:195-203)
    li, y = sp.all_links(shuffle=False)
    assert li.shape[1] == 2 * (pos_tr.shape[1] + n_v + n_t)
    assert np.array_equal(li[:, :pos_tr.shape[1]], pos_tr) and y[:pos_tr.shape[1]].all()
    assert not y[pos_tr.shape[1]:2 * pos_tr.shape[1]].any()
    # each list is permuted like get_pos_neg_edges does (utils.py:650-657), the blocks stay in place
    ls, ys = sp.all_links()
    assert np.array_equal(ys, y) and not np.array_equal(ls, li)
    Ptr = pos_tr.shape[1]
    assert np.array_equal(np.sort(key(ls[:, :Ptr])), ktr)
    assert np.array_equal(np.sort(key(ls[:, Ptr:2 * Ptr])), np.sort(key(neg_tr)))


def test_edge_split_counts_of_the_baseline_configs():
    """L = 7 868 (USAir), 19 532 (Cora), 164 000 (PubMed)."""
    for name, L in (("usair", 7868), ("cora", 19532), ("pubmed", 164000)):
        n, e = W.load_topology(name)
        li, _ = W.edge_split(n, e, seed=0).all_links()
        assert li.shape[1] == L


# ---- NormalizeFeatures: sgrl_link_pred.py:851, :1000-1003 ------------------------------------------
def test_normalize_features_known_answers():
    X = np.array([[1, 3], [0, 0], [0.2, 0.3]], dtype=np.float32)
    got = W.normalize_features(X)
    assert got.dtype == np.float32
    # min is 0: rows / max(sum, 1) -> zero rows stay zero, a row summing to 0.5 is left unscaled
    assert np.allclose(got, [[0.25, 0.75], [0, 0], [0.2, 0.3]])
    # the GLOBAL minimum is subtracted first (signed node2vec features)
    assert np.allclose(W.normalize_features(np.array([[-1.0, 1.0], [0.0, -1.0]])), [[0, 1], [1, 0]])
    # applied twice (Planetoid transform, then again after init_features): idempotent on its own
    # output when that is non-negative with a zero somewhere (bag-of-words rows)
    B = (np.random.default_rng(0).random((50, 40)) < 0.1).astype(np.float32)
    once = W.normalize_features(B)
    assert np.array_equal(W.normalize_features(once), once)
    assert np.allclose(once.sum(1)[B.sum(1) > 0], 1.0) and (once[B.sum(1) == 0] == 0).all()


# ---- init_features = degree: sgrl_link_pred.py:961-963 ---------------------------------------------
def test_one_hot_degree_on_the_train_graph():
    n, e = W.load_topology("usair")
    sp = W.edge_split(n, e, seed=0)
    oh = W.one_hot_degree(sp.A)
    assert oh.shape == (n, 1025) and oh.dtype == np.float32          # max_degree 1024 -> 1 025 classes
    deg_train = np.bincount(sp.train_edges.ravel(), minlength=n)      # TRAIN graph, not the full one
    assert np.array_equal(oh.argmax(1), deg_train) and (oh.sum(1) == 1).all()
    full_deg = np.bincount(e.ravel(), minlength=n)
    assert (deg_train <= full_deg).all() and (deg_train < full_deg).any()
    # cat=True: appended to the existing features, then NormalizeFeatures over the joint row
    X0 = W.normalize_features((np.random.default_rng(1).random((n, 7)) < 0.5).astype(np.float32))
    X = W.init_degree_features(X0, sp.A)
    assert X.shape == (n, 7 + 1025)
    has = X0.sum(1) > 0
    assert np.allclose(X[has, :7], X0[has] / 2) and np.allclose(X[has, 7:].sum(1), 0.5)
    assert np.allclose(X[~has, 7:].sum(1), 1.0)
    # no features at all: the one-hot alone
    assert np.array_equal(W.init_degree_features(None, sp.A), oh)


def test_one_hot_degree_rejects_degrees_above_the_class_count():
    import scipy.sparse as ssp

    hub = np.zeros((1030, 2), dtype=np.int64)
    hub[:, 1] = np.arange(1, 1031)
    A = W.csr_from_undirected(1031, hub)
    assert isinstance(A, ssp.csr_matrix)
    with pytest.raises(RuntimeError):
        W.one_hot_degree(A)                      # F.one_hot(1030, num_classes=1025) raises in PyG
    assert W.one_hot_degree(A, max_degree=2048).shape == (1031, 2049)


def test_sop_workload_uses_degree_features():
    """BASELINE config 3: 500 raw + 1 025 one-hot columns (F = 1 525)."""
    w = W.make("pubmed_sop_k3")
    assert w.X.shape == (19717, 1525) and w.mode == "sop" and w.sign_k == 3
    assert np.array_equal(w.X[:, 500:].argmax(1), np.diff(w.A.indptr))
