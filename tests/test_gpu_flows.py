"""-m gpu: the callers of the operators, mirrored: the flow dispatch
`extract_enclosing_subgraphs` (reference utils.py:446-554) per flow against the oracle, the
per-split orchestration `process_split` (reference sgrl_link_pred.py:96-220), the lazy drop-in
list on real hardware, and the sharded path with the ENGINE as the per-rank compute."""
import os
import socket

import numpy as np
import pytest

import oracle
from conftest import csr_from_undirected, load_extract
from test_gpu_parity import TOL, rel_err

pytestmark = pytest.mark.gpu


def _usair(seed=3, F=12, nlinks=14):
    import torch

    g = load_extract("usair")
    n = int(g["num_nodes"])
    A = csr_from_undirected(n, g["edges"])
    X = np.random.default_rng(seed).random((n, F)).astype(np.float32)
    li = torch.from_numpy(g["links"][:nlinks].T.copy())
    return A, X, li


def _kw(sign_type, K, k_heuristic=0, optimize=True):
    return {"sign_k": K, "use_feature": True, "sign_type": sign_type, "optimize_sign": optimize,
            "k_heuristic": k_heuristic, "k_node_set_strategy": "intersection"}


def _check_list(got, ref, names):
    assert len(got) == len(ref)
    for d, r in zip(got, ref):
        assert d.y == r["y"]
        for k in names:
            assert tuple(d[k].shape) == r[k].shape, k
            assert rel_err(d[k].numpy(), r[k]) < TOL, k


@pytest.mark.parametrize("flow", ["pos", "pos_plus", "sop", "hybrid", "hybrid_k1"])
def test_flow_dispatch_matches_oracle(flow):
    """One test per branch of reference utils.py:454-496."""
    import torch
    from s3grl_amd import extract_enclosing_subgraphs
    from s3grl_amd.dataset import GlobalOperators
    from s3grl_amd.tuned_SIGN import LinkDataList, clear_cache

    A, X, li = _usair()
    x = torch.from_numpy(X)
    X64 = X.astype(np.float64)
    K, hops = (1 if flow == "hybrid_k1" else 3), 2
    okw = {"sign_k": K, "k_node_set_strategy": "intersection"}
    names = ["x"] + [f"x{i}" for i in range(1, K + 1)]
    if flow == "pos":
        got = extract_enclosing_subgraphs(li, A, x, 1, hops, "zo", 1.0, None, False, None, None,
                                          _kw("PoS", K), powers_of_A=[])
        ref = oracle.get_PoS_prepped_ds(li.numpy(), hops, A, X64, 1, okw, dtype=np.float64)
    elif flow == "pos_plus":
        got = extract_enclosing_subgraphs(li, A, x, 0, hops, "zo", 1.0, None, False, None, None,
                                          _kw("PoS", K, k_heuristic=1), powers_of_A=[])
        ref = oracle.get_PoS_Plus_prepped_ds(li.numpy(), hops, A, X64, 0, okw, dtype=np.float64)
        assert any(d.x.shape[0] > 2 for d in got)          # the fixture has common neighbours
    elif flow == "sop":
        got = extract_enclosing_subgraphs(li, A, x, 1, -1, "zo", 1.0, None, False, None, None,
                                          _kw("SoP", K), powers_of_A=GlobalOperators(K))
        ref = oracle.get_SoP_prepped_ds(oracle.global_normalized_powers(A, K, np.float64), li.numpy(), A,
                                        X64, 1, dtype=np.float64)
    else:
        got = extract_enclosing_subgraphs(li, A, x, 1, hops, "zo", 1.0, None, False, None, None,
                                          _kw("hybrid", K), powers_of_A=GlobalOperators(K))
        pos = oracle.get_PoS_prepped_ds(li.numpy(), hops, A, X64, 1, okw, dtype=np.float64)
        sop = oracle.get_SoP_prepped_ds(oracle.global_normalized_powers(A, K, np.float64), li.numpy(), A,
                                        X64, 1, dtype=np.float64)
        ref = oracle.hybrid_combine(pos, sop, K) if K > 1 else pos
        names = ["x"] + [f"x{i}" for i in range(1, 2 * K)]       # x{K+1}..x{2K-1} = SoP x2..xK
        if K > 1:
            assert f"x{2 * K - 1}" in got[0] and f"x{2 * K}" not in got[0]
            assert rel_err(got[0][f"x{K + 1}"].numpy(), sop[0]["x2"]) < TOL
    assert isinstance(got, LinkDataList)
    _check_list(got, ref, names)
    clear_cache()


def test_flow_dispatch_unsupported_branches():
    import torch
    from s3grl_amd import extract_enclosing_subgraphs

    A, X, li = _usair()
    x = torch.from_numpy(X)
    with pytest.raises(NotImplementedError):      # utils.py:497: per-link SIGN + SEAL graphs
        extract_enclosing_subgraphs(li, A, x, 1, 2, "zo", 1.0, None, False, None, None,
                                    _kw("PoS", 3, optimize=False), powers_of_A=[])
    with pytest.raises(NotImplementedError):      # utils.py:556: SEAL flow
        extract_enclosing_subgraphs(li, A, x, 1, 2, "drnl", 1.0, None, False, None, {"rw_m": 0}, None)
    with pytest.raises(NotImplementedError):      # tuned_SIGN.py:235 through the dispatch
        kw = _kw("PoS", 3, k_heuristic=1)
        kw["k_node_set_strategy"] = "neither"
        extract_enclosing_subgraphs(li, A, x, 1, 2, "zo", 1.0, None, False, None, None, kw, powers_of_A=[])


def test_lazy_list_from_the_engine_is_fresh_cpu_memory():
    """a list of L per-link objects owning fresh CPU tensors."""
    import torch
    from s3grl_amd.tuned_SIGN import LinkDataList, OptimizedSignOperations, clear_cache

    A, X, li = _usair(nlinks=20)
    x = torch.from_numpy(X)
    kw = _kw("PoS", 2)
    a = OptimizedSignOperations.get_PoS_prepped_ds(li, 1, A, 1.0, None, False, None, x, 1, kw, None)
    b = OptimizedSignOperations.get_PoS_prepped_ds(li, 1, A, 1.0, None, False, None, x, 0, kw, None)
    assert isinstance(a, LinkDataList) and len(a) == len(b) == 20
    assert a[0].x.device.type == "cpu" and a[0].x.dtype == torch.float32
    assert a[0].x.data_ptr() != b[0].x.data_ptr()            # the second call did not reuse a's memory
    assert torch.equal(a[3].x1, b[3].x1) and a[3].y == 1 and b[3].y == 0
    keep = a[5].x2.clone()
    view = a[5].x2
    del a                                                    # a view keeps its block alive
    c = OptimizedSignOperations.get_PoS_prepped_ds(li, 1, A, 1.0, None, False, None, x, 1, kw, None)
    assert torch.equal(view, keep)
    both = b + c                                             # sgrl_link_pred.py:204
    rows, ptr, y = both.collate()
    assert rows.shape[0] == 80 and ptr[-1] == 80 and y.tolist() == [0] * 20 + [1] * 20
    # operator 0 = [z | X[node]] with z = 1 on the src / dst rows (tuned_SIGN.py:177-182)
    assert torch.equal(both[0].x[:, 1:], x[li[:, 0]]) and both[0].x[:, 0].tolist() == [1.0, 1.0]
    clear_cache()


def test_process_split_orders_caches_and_subsamples(tmp_path):
    import torch
    from s3grl_amd import process_split, workloads
    from s3grl_amd.engine import default_engine
    from s3grl_amd.tuned_SIGN import clear_cache

    n, e = workloads.load_topology("usair")
    sp = workloads.edge_split(n, e, seed=0)
    X = np.random.default_rng(1).random((n, 9)).astype(np.float32)
    x = torch.from_numpy(X)
    se = sp.split_edge()
    root = tmp_path / "dataset" / "USAir"
    kwargs = dict(sign_k=2, sign_type="PoS", k_heuristic=1, dataset_root=root, seed=0)
    np.random.seed(11)
    rows, ptr, y, meta = process_split("valid", se, sp.edge_index(), n, x, 1, **kwargs)
    P, Q = se["valid"]["edge"].shape[0], se["valid"]["edge_neg"].shape[0]
    assert y.tolist() == [1] * P + [0] * Q and len(ptr) == P + Q + 1
    assert meta["num_pos"] == P and meta["num_neg"] == Q and meta["mode"] == "pos_plus"
    # == the engine on the concatenated list with the train graph of the split.  The reference
    # shuffles each list with numpy's global generator even at percent = 100 (utils.py:650-657)
    eng = default_engine()
    np.random.seed(11)
    li = np.concatenate([se["valid"]["edge"][np.random.permutation(P)],
                         se["valid"]["edge_neg"][np.random.permutation(Q)]]).T
    res = eng.precompute(eng.graph(sp.A), eng.features(X), eng.links(li), mode="pos_plus", num_hops=1, sign_k=2)
    assert np.array_equal(np.asarray(rows), res.rows.cpu().numpy())
    assert np.array_equal(np.asarray(ptr), res.row_ptr.cpu().numpy())
    # the bundle sits under the reference's data_appendix directory + the operator settings
    hits = list(tmp_path.rglob("SEAL_valid_data.s3grl"))
    assert len(hits) == 1 and "_h1_zo_rph10_seed0_pos_plus_k2_intersection" in str(hits[0])
    # second call: loaded, not recomputed
    import s3grl_amd.dataset as ds

    calls = []
    orig = ds.extract_enclosing_subgraphs
    ds.extract_enclosing_subgraphs = lambda *a, **k: calls.append(1) or orig(*a, **k)
    try:
        rows2, ptr2, y2, _ = process_split("valid", se, sp.edge_index(), n, x, 1, **kwargs)
        assert calls == [] and np.array_equal(np.asarray(rows2), np.asarray(rows))
        # other operator settings -> other directory -> recomputed (the reference's key omits them)
        process_split("valid", se, sp.edge_index(), n, x, 1, sign_k=3, sign_type="PoS", dataset_root=root, seed=0)
        assert calls == [1, 1]
    finally:
        ds.extract_enclosing_subgraphs = orig
    # percent < 100: numpy's global generator picks the links, positives first (utils.py:650-657)
    np.random.seed(7)
    rows50, ptr50, y50, m50 = process_split("test", se, sp.edge_index(), n, x, 1, sign_k=2, sign_type="SoP", percent=50)
    np.random.seed(7)
    Pt = se["test"]["edge"].shape[0]
    perm_p = np.random.permutation(Pt)[:int(0.5 * Pt)]
    perm_n = np.random.permutation(se["test"]["edge_neg"].shape[0])[:int(0.5 * se["test"]["edge_neg"].shape[0])]
    li50 = np.concatenate([se["test"]["edge"][perm_p], se["test"]["edge_neg"][perm_n]]).T
    ref = eng.precompute(eng.graph(sp.A), eng.features(X), eng.links(li50), mode="sop", sign_k=2)
    assert np.array_equal(np.asarray(rows50), ref.rows.cpu().numpy()) and m50["num_pos"] == len(perm_p)
    clear_cache()


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


@pytest.mark.parametrize("mode,chunks", [("pos", 1), ("pos", 3), ("sop", 2)])
def test_sharded_precompute_with_the_engine_world1_nccl(mode, chunks):
    """SURVEY §8(e) with the ENGINE as the per-rank compute, on RCCL (world size 1: one GPU here).
    The pipelined path is forced by calling the fixed-rows flavour with gather=True semantics on a
    1-rank group is a no-op, so the shard logic is exercised by cutting the list into `world`
    virtual ranks that all run on this device, and the collective by an in-place all-gather on the
    1-rank RCCL group."""
    import torch
    import torch.distributed as dist
    from s3grl_amd import parallel
    from s3grl_amd.engine import default_engine

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(_free_port())
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    try:
        g = load_extract("usair")
        n = int(g["num_nodes"])
        A = csr_from_undirected(n, g["edges"])
        X = np.random.default_rng(0).random((n, 6)).astype(np.float32)
        links = g["links"].T
        eng = default_engine()
        G, xd = eng.graph(A), eng.features(X)
        K = 2
        compute = parallel.engine_compute(eng, G, xd, mode=mode, num_hops=1, sign_k=K)
        whole = torch.empty((2 * links.shape[1], K + 1, 7), dtype=torch.float32, device=eng.device)
        compute(torch.from_numpy(links), whole)                       # unsharded reference
        cost = parallel.link_cost(A, links)
        # the collective itself on RCCL: 1-rank in-place all-gather round trip
        rows, ptr, (lo, hi) = parallel.sharded_precompute(
            compute, torch.from_numpy(links).to(eng.device), rank=0, world_size=1, cost=cost,
            rows_per_link=2, chunks=chunks, row_shape=(K + 1, 7), device=eng.device)
        assert (lo, hi) == (0, links.shape[1]) and torch.equal(rows, whole)
        # the pipelined pieces + IN-PLACE all_gather_into_tensor + compaction, on real RCCL
        for _ in range(2):          # the second call reuses the cached slots
            rows_c, ptr_c, _ = parallel.sharded_precompute(
                compute, torch.from_numpy(links).to(eng.device), rank=0, world_size=1, cost=cost,
                rows_per_link=2, chunks=chunks, row_shape=(K + 1, 7), device=eng.device,
                collective_at_world1=True)
            torch.cuda.synchronize()
            assert torch.equal(rows_c, whole) and ptr_c.tolist() == list(range(0, 2 * links.shape[1] + 1, 2))
        # reversed duplicates rebuilt from their primaries instead of exchanged (mirror_rows), with the real
        # engine: its rows of (d, s) are the rows of (s, d) swapped bit for bit — PoS and SoP alike
        if mode in ("pos", "sop"):
            both = np.concatenate([links, links[::-1, :15], links[:, :4]], axis=1)
            both = np.ascontiguousarray(both[:, np.random.default_rng(3).permutation(both.shape[1])])
            whole_b = torch.empty((2 * both.shape[1], K + 1, 7), dtype=torch.float32, device=eng.device)
            compute(torch.from_numpy(both), whole_b)
            sp = parallel.ShardPlan(torch.from_numpy(both), 1, None, pair_aware=True, device=eng.device)
            assert sp.reverse_of_previous.sum() >= 15
            x_dev, li_dev = xd.tensor, torch.from_numpy(both).to(eng.device)

            def fill0(fl):
                fl[:, :, 0, 0] = 1.0
                fl[:, 0, 0, 1:] = x_dev[li_dev[0]]
                fl[:, 1, 0, 1:] = x_dev[li_dev[1]]

            for f0 in (None, fill0) if mode == "pos" else (None,):
                rows_m, _, _ = parallel.sharded_precompute(
                    compute, li_dev, rank=0, world_size=1, rows_per_link=2, chunks=chunks, row_shape=(K + 1, 7),
                    device=eng.device, collective_at_world1=True, shards=sp, mirror_rows=True, local_operator0=f0)
                torch.cuda.synchronize()
                assert torch.equal(rows_m, whole_b)
            # ... and the cheapest links computed by every rank itself instead of exchanged (replicate)
            cost_b = parallel.link_cost(A, both).astype(np.float64)
            rep = parallel.replicate_cheapest(both, cost_b, 0.4)
            sp_r = parallel.ShardPlan(torch.from_numpy(both), 1, cost_b, pair_aware=True, device=eng.device, replicate=rep)
            assert 0 < sp_r.rep_start < both.shape[1]
            rows_r, _, where_r = parallel.sharded_precompute(
                compute, li_dev, rank=0, world_size=1, rows_per_link=2, chunks=chunks, row_shape=(K + 1, 7),
                device=eng.device, collective_at_world1=True, shards=sp_r, mirror_rows=True)
            torch.cuda.synchronize()
            assert torch.equal(rows_r, whole_b) and where_r.numel() == both.shape[1]
        probe = whole[:8].clone()
        out = torch.empty_like(probe)
        dist.all_gather_into_tensor(out, probe)
        assert torch.equal(out, probe)
        # the shard logic with the engine: every virtual rank's range, bit-equal to its slice
        for world in (2, 3):
            b = parallel.shard_bounds(links.shape[1], world, cost)
            for r in range(world):
                rows_r, ptr_r, (lo, hi) = parallel.sharded_precompute(
                    compute, links, rank=r, world_size=world, cost=cost, rows_per_link=2,
                    row_shape=(K + 1, 7), device=eng.device, gather=False)
                assert (lo, hi) == (b[r], b[r + 1])
                assert torch.equal(rows_r, whole[2 * lo:2 * hi]), (world, r)
                assert ptr_r.tolist() == list(range(0, 2 * (hi - lo) + 1, 2))
        G.close()
    finally:
        dist.destroy_process_group()
        parallel._Buffers.clear()


def test_graph_create_rejects_malformed_csr_and_trim_returns_memory():
    """the raw-CSR entry validates its input on the device (ids in range, rows strictly
    ascending, monotone indptr) instead of corrupting LDS; `trim` gives the cached workspace back."""
    import torch
    from s3grl_amd.engine import Engine

    eng = Engine("cuda:0")
    good_ptr, good_idx = np.array([0, 2, 4, 6]), np.array([1, 2, 0, 2, 0, 1])
    g = eng.graph(indptr=good_ptr, indices=good_idx, num_nodes=3)
    g.close()
    for ptr, idx, what in [
        (np.array([0, 2, 4, 6]), np.array([1, 7, 0, 2, 0, 1]), "outside"),          # id out of range
        (np.array([0, 2, 4, 6]), np.array([2, 1, 0, 2, 0, 1]), "ascending"),        # unsorted row
        (np.array([0, 2, 4, 6]), np.array([1, 1, 0, 2, 0, 1]), "ascending"),        # duplicate entry
        (np.array([0, 4, 2, 6]), np.array([1, 2, 0, 2, 0, 1]), "monotone"),         # indptr not monotone
        (np.array([0, 2, 4, 5]), np.array([1, 2, 0, 2, 0, 1]), "monotone"),         # does not end at nnz
    ]:
        with pytest.raises(ValueError, match=what):
            eng.graph(indptr=ptr, indices=idx, num_nodes=3)
    # a PubMed-scale plan leaves GBs cached in the context's arena; trim hands them back
    from s3grl_amd import workloads

    w = workloads.make("cora_posplus_k3")
    li, _ = w.split.all_links()
    G = eng.graph(w.A)
    res = eng.precompute(G, eng.features(w.X), eng.links(li), mode="pos_plus", num_hops=3, sign_k=3)
    before = res.stats["workspace_bytes"]
    del res
    G.close()
    torch.cuda.synchronize()
    freed = eng.trim()
    assert freed > 0 and freed <= before
    assert eng.trim() == 0
    # and the engine keeps working afterwards
    G = eng.graph(w.A)
    res = eng.precompute(G, eng.features(w.X), eng.links(li[:, :100]), mode="pos", num_hops=2, sign_k=2)
    assert res.rows.shape[0] == 200
    eng.close()


def test_link_costs_and_sizes_from_the_sizing_pass():
    from s3grl_amd.engine import Engine

    eng = Engine("cuda:0")
    g = load_extract("usair")
    n = int(g["num_nodes"])
    A = csr_from_undirected(n, g["edges"])
    G = eng.graph(A)
    links = eng.links(g["links"].T)
    sizes = eng.subgraph_sizes(G, links, num_hops=2).cpu().numpy()
    plan = eng.plan(G, links, mode="pos", num_hops=2, sign_k=2, full_stats=True)
    node_ptr = plan.export_subgraphs()[0].cpu().numpy()
    assert np.array_equal(sizes, np.diff(node_ptr))            # folding off: every link sized
    cost = eng.link_costs(G, links, num_hops=2).cpu().numpy()
    assert np.allclose(cost, sizes + 400.0)                    # (the fixture holds no reversed duplicates)
    plan.close()
    # a reversed duplicate is folded into its primary in the real run: priced at its output rows;
    # PoS Plus: every row pair is another pass over the subgraph
    both = np.concatenate([g["links"][:8], g["links"][:8, ::-1]])
    lb = eng.links(both.T.copy())
    cb = eng.link_costs(G, lb, num_hops=2).cpu().numpy()
    # (the direction with src < dst is the primary, wherever it stands in the list)
    fwd = g["links"][:8, 0] < g["links"][:8, 1]
    assert np.allclose(np.where(fwd, cb[:8], cb[8:]), sizes[:8] + 400.0)
    assert np.allclose(np.where(fwd, cb[8:], cb[:8]), 250.0)
    full = eng.link_costs(G, lb, num_hops=2, fold_reversed=False).cpu().numpy()
    assert np.allclose(full[8:], sizes[:8] + 400.0)
    pp = eng.plan(G, links, mode="pos_plus", num_hops=2, sign_k=2)
    pairs = (np.diff(pp.row_ptr().cpu().numpy()) + 1) // 2
    pp.close()
    cp = eng.link_costs(G, links, num_hops=2, mode="pos_plus").cpu().numpy()
    assert np.allclose(cp, pairs * sizes + 400.0) and pairs.max() > 1
    # a count-only plan cannot be run
    p = eng.plan(G, links, mode="pos", num_hops=2, sign_k=2, count_only=True)
    with pytest.raises(ValueError, match="count-only"):
        p.run(eng.features(np.ones((n, 4), dtype=np.float32)))
    p.close()
    eng.close()


def test_pos_call_computed_and_copied_piece_by_piece(monkeypatch):
    """Long PoS lists through the drop-in operator are computed in pieces whose rows travel to the host
    while the next piece is computed (tuned_SIGN._pos_pipelined): the same tensor, bit for bit, as the
    whole-list call — reversed duplicates cut apart by a piece boundary included."""
    import torch
    from s3grl_amd import tuned_SIGN, workloads
    from s3grl_amd.tuned_SIGN import OptimizedSignOperations as ops

    w = workloads.make("cora_posplus_k3")
    A = w.A
    n = A.shape[0]
    rng = np.random.default_rng(0)
    pos = np.stack(A.nonzero())                                    # both directions of every edge
    neg = rng.integers(0, n, size=(2, 40000 - pos.shape[1] % 40000))
    neg = neg[:, neg[0] != neg[1]]
    li = torch.from_numpy(np.concatenate([pos, neg], 1)[:, :40000].astype(np.int64))
    assert li.shape[1] >= tuned_SIGN._PIPE_MIN_LINKS
    X = torch.from_numpy(np.ascontiguousarray(w.X[:, :24], dtype=np.float32))
    kw = {"sign_k": 3, "k_node_set_strategy": "intersection"}
    outs = []
    for pieces in ("1", "4", "3"):
        monkeypatch.setenv("S3GRL_D2H_PIECES", pieces)
        lst = ops.get_PoS_prepped_ds(li, 2, A, 1.0, None, False, None, X, 1, kw, None)
        rows, ptr, _ = lst.collate()
        assert rows.device.type == "cpu" and rows.shape[0] == 2 * li.shape[1]
        outs.append(rows.clone())
    tuned_SIGN.clear_cache()
    assert torch.equal(outs[0], outs[1]) and torch.equal(outs[0], outs[2])
