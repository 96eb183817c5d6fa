"""Host logic of the callers' twins (s3grl_amd/dataset.py) on CPU: the flow decision table of
reference utils.py:446-554 and the per-split orchestration of sgrl_link_pred.py:96-220, with the
operators replaced by stand-ins that answer from the oracle (no GPU here).  The same functions
run against the engine in tests/test_gpu_flows.py."""
import numpy as np
import pytest
import torch

import oracle
from conftest import csr_from_undirected, load_extract
from s3grl_amd import dataset as ds
from s3grl_amd import workloads
from s3grl_amd.tuned_SIGN import LinkDataList


class FakeOps:
    """OptimizedSignOperations stand-in: records the calls, answers with the oracle's rows."""
    calls = []

    @staticmethod
    def _wrap(lst, K, y):
        rows, ptr, _ = oracle.collate_rows(lst, K)
        return LinkDataList([(torch.from_numpy(rows.astype(np.float32)), np.asarray(ptr, dtype=np.int64), y)], K)

    @staticmethod
    def get_PoS_prepped_ds(link_index, num_hops, A, ratio_per_hop, max_nodes_per_hop, directed, A_csc, x, y,
                           sign_kwargs, rw_kwargs):
        FakeOps.calls.append(("pos", int(link_index.shape[1]), num_hops, y, rw_kwargs))
        K = sign_kwargs["sign_k"]
        return FakeOps._wrap(oracle.get_PoS_prepped_ds(np.asarray(link_index), num_hops, A, np.asarray(x, dtype=np.float64),
                                                       y, sign_kwargs, dtype=np.float64), K, y)

    @staticmethod
    def get_PoS_Plus_prepped_ds(link_index, num_hops, A, ratio_per_hop, max_nodes_per_hop, directed, A_csc, x, y,
                                sign_kwargs, rw_kwargs):
        FakeOps.calls.append(("pos_plus", int(link_index.shape[1]), num_hops, y, rw_kwargs))
        K = sign_kwargs["sign_k"]
        return FakeOps._wrap(oracle.get_PoS_Plus_prepped_ds(np.asarray(link_index), num_hops, A,
                                                            np.asarray(x, dtype=np.float64), y, sign_kwargs,
                                                            dtype=np.float64), K, y)

    @staticmethod
    def get_SoP_prepped_ds(powers_of_A, link_index, A, x, y):
        FakeOps.calls.append(("sop", int(link_index.shape[1]), None, y, None))
        K = len(powers_of_A)
        P = oracle.global_normalized_powers(A, K, np.float64)
        return FakeOps._wrap(oracle.get_SoP_prepped_ds(P, np.asarray(link_index), A, np.asarray(x, dtype=np.float64), y,
                                                       dtype=np.float64), K, y)


@pytest.fixture()
def fake_ops(monkeypatch):
    FakeOps.calls = []
    monkeypatch.setattr(ds, "OptimizedSignOperations", FakeOps)
    return FakeOps


def _inputs(nlinks=10):
    g = load_extract("usair")
    n = int(g["num_nodes"])
    A = csr_from_undirected(n, g["edges"])
    X = np.random.default_rng(4).random((n, 5)).astype(np.float32)
    return A, torch.from_numpy(X), torch.from_numpy(g["links"][:nlinks].T.copy())


def _kw(sign_type, K, k_heuristic=0, optimize=True):
    return {"sign_k": K, "use_feature": True, "sign_type": sign_type, "optimize_sign": optimize,
            "k_heuristic": k_heuristic, "k_node_set_strategy": "intersection"}


@pytest.mark.parametrize("sign_type,k_heur,powers,expect", [
    ("PoS", 0, False, ["pos"]), ("PoS", 1, False, ["pos_plus"]), ("SoP", 0, True, ["sop"]),
    ("hybrid", 0, True, ["pos", "sop"]), ("PoS", 0, True, ["sop"]),      # powers_of_A decide before sign_type
])
def test_decision_table(fake_ops, sign_type, k_heur, powers, expect):
    A, x, li = _inputs()
    K = 3
    out = ds.extract_enclosing_subgraphs(li, A, x, 1, 2, "zo", 1.0, None, False, None, None,
                                         _kw(sign_type, K, k_heur),
                                         powers_of_A=ds.GlobalOperators(K) if powers else [])
    assert [c[0] for c in fake_ops.calls] == expect
    assert len(out) == li.shape[1]
    if sign_type == "hybrid":
        assert isinstance(out, LinkDataList) and out._K == 2 * K - 1
        assert f"x{2 * K - 1}" in out[0] and f"x{2 * K}" not in out[0]


def test_hybrid_combine_matches_the_references_loop(fake_ops):
    """The one-`cat` fast path gives what utils.py:472-480 builds element by element."""
    A, x, li = _inputs(8)
    K = 3
    okw = {"sign_k": K, "k_node_set_strategy": "intersection"}
    X64 = x.numpy().astype(np.float64)
    pos = oracle.get_PoS_prepped_ds(li.numpy(), 2, A, X64, 1, okw, dtype=np.float64)
    sop = oracle.get_SoP_prepped_ds(oracle.global_normalized_powers(A, K, np.float64), li.numpy(), A, X64, 1,
                                    dtype=np.float64)
    ref = oracle.hybrid_combine(pos, sop, K)
    got = ds.extract_enclosing_subgraphs(li, A, x, 1, 2, "zo", 1.0, None, False, None, None, _kw("hybrid", K),
                                         powers_of_A=ds.GlobalOperators(K))
    for d, r in zip(got, ref):
        for k in ["x"] + [f"x{i}" for i in range(1, 2 * K)]:
            assert np.allclose(d[k].numpy(), r[k], rtol=1e-6, atol=1e-7), k
    # the generic path (plain lists) does the same
    slow = ds._hybrid_combine(list(FakeOps._wrap(pos, K, 1)), list(FakeOps._wrap(sop, K, 1)), K)
    assert all(torch.equal(a[f"x{K + 1}"], b[f"x{K + 1}"]) for a, b in zip(slow, got))
    # sign_k == 1: the PoS list alone (utils.py:467-468)
    fake_ops.calls.clear()
    one = ds.extract_enclosing_subgraphs(li, A, x, 1, 2, "zo", 1.0, None, False, None, None, _kw("hybrid", 1),
                                         powers_of_A=ds.GlobalOperators(1))
    assert [c[0] for c in fake_ops.calls] == ["pos"] and one._K == 1


def test_unsupported_flows_raise(fake_ops):
    A, x, li = _inputs()
    with pytest.raises(NotImplementedError):
        ds.extract_enclosing_subgraphs(li, A, x, 1, 2, "zo", 1.0, None, False, None, None, _kw("PoS", 3, optimize=False),
                                       powers_of_A=[])
    with pytest.raises(NotImplementedError):
        ds.extract_enclosing_subgraphs(li, A, x, 1, 2, "drnl", 1.0, None, False, None, {"rw_m": 0}, None)
    assert fake_ops.calls == []


def test_process_split_order_cache_and_kwargs(fake_ops, tmp_path, monkeypatch):
    n, e = workloads.load_topology("usair")
    sp = workloads.edge_split(n, e, seed=0)
    x = torch.from_numpy(np.random.default_rng(2).random((n, 4)).astype(np.float32))
    se = sp.split_edge()
    P, Q = len(se["test"]["edge"]), len(se["test"]["edge_neg"])
    np.random.seed(3)
    made = []

    def fake_cache(A, edges, device, rw_m, rw_M, seed=0):      # stands in for the engine's walks
        made.append((int(torch.as_tensor(edges).shape[1]), rw_m, rw_M, seed))
        return {"cache": len(made)}

    monkeypatch.setattr(ds, "create_rw_cache", fake_cache)
    rows, ptr, y, meta = ds.process_split("test", se, sp.edge_index(), n, x, 1, sign_k=2, sign_type="PoS",
                                          m=3, M=20, rw_seed=9, dataset_root=tmp_path / "USAir", seed=5)
    # positives (y = 1) first, then negatives (y = 0); the ScaLed settings travel as rw_kwargs, with
    # one walk cache per list like sgrl_link_pred.py:123-140 builds them (positives first)
    assert [(c[0], c[1], c[3]) for c in fake_ops.calls] == [("pos", P, 1), ("pos", Q, 0)]
    assert made == [(P, 3, 20, 9), (Q, 3, 20, 9)]
    assert fake_ops.calls[0][4] == {"rw_m": 3, "rw_M": 20, "sign": True, "seed": 9,
                                    "cached_pos_rws": {"cache": 1}, "cached_neg_rws": {"cache": 2}}
    assert np.asarray(y).tolist() == [1] * P + [0] * Q and meta["num_pos"] == P
    # the train graph handed to the operators is built from edge_index only: int ones, duplicates summed
    A = ds.train_graph(np.array([[0, 0, 1], [1, 1, 0]]), 3)
    assert A[0, 1] == 2 and A[1, 0] == 1 and A.dtype.kind == "i"
    # directory = the reference's data_appendix (here the ScaLed form) + the operator settings
    hit = list(tmp_path.rglob("SEAL_test_data.s3grl"))
    assert len(hit) == 1 and "_seal_m3_M20_dropedge0.0_seed5_pos_k2" in str(hit[0])
    # second call with the same settings: loaded, the operators are not called again
    fake_ops.calls.clear()
    rows2, _, _, _ = ds.process_split("test", se, sp.edge_index(), n, x, 1, sign_k=2, sign_type="PoS",
                                      m=3, M=20, rw_seed=9, dataset_root=tmp_path / "USAir", seed=5)
    assert fake_ops.calls == [] and np.array_equal(np.asarray(rows2), np.asarray(rows))
    # SoP / hybrid get the global-operator stand-in, PoS an empty list (sgrl_link_pred.py:161-178)
    ds.process_split("valid", se, sp.edge_index(), n, x, -1, sign_k=2, sign_type="SoP")
    assert [c[0] for c in fake_ops.calls] == ["sop", "sop"]
    with pytest.raises(NotImplementedError):
        ds.process_split("valid", se, sp.edge_index(), n, x, 1, sign_k=2, sign_type="nope")
    # percent: int(percent / 100 * count) links of each list (utils.py:650-657)
    fake_ops.calls.clear()
    ds.process_split("train", se, sp.edge_index(), n, x, 1, sign_k=2, sign_type="PoS", percent=10)
    Ptr = len(se["train"]["edge"])
    assert [c[1] for c in fake_ops.calls] == [int(0.1 * Ptr), int(0.1 * len(se["train"]["edge_neg"]))]
