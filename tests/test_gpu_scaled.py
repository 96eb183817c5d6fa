"""-m gpu: ScaLed subgraphs from walk caches the CALLER hands in (reference utils.py:94-108,134-146;
caches built at sgrl_link_pred.py:123-140 / utils.py:425-443).  The clause of north_star that can
be exact here — bit-exact node-index sets — is checked for every link of USAir and Cora: the
node list a plan exports equals [src, dst] + sorted(unique(cat(cache[src], cache[dst])) - {src, dst}),
the list the reference's rw branch builds from the same cache; the rows match the oracle's
restatement of that branch on the same sets."""
import numpy as np
import pytest
import torch

import oracle
from s3grl_amd import scaled, workloads

pytestmark = pytest.mark.gpu

TOL = 1e-5


@pytest.fixture(scope="module")
def eng():
    from s3grl_amd.engine import Engine

    assert torch.cuda.is_available()
    e = Engine("cuda:0")
    yield e
    e.close()


def rel_err(got, ref):
    ref = np.asarray(ref, dtype=np.float64)
    got = np.asarray(got, dtype=np.float64)
    scale = np.maximum(np.abs(ref), np.abs(ref).max(axis=-1, keepdims=True))
    return float(np.max(np.clip(np.abs(got - ref) - 1e-10, 0, None) / np.maximum(scale, 1e-30)))


def foreign_cache(A, edges, m, M, seed):
    """A cache as the reference's own create_rw_cache would hand it in: a plain dict node -> sorted
    unique int64 tensor, drawn HERE with numpy (stands for torch_cluster's walks: any walks do)."""
    rng = np.random.default_rng(seed)
    indptr, indices = A.indptr, A.indices
    out = {}
    for v in np.unique(np.asarray(edges).reshape(-1)):
        seen = [int(v)]
        for _ in range(M):
            cur = int(v)
            for _ in range(m):
                deg = indptr[cur + 1] - indptr[cur]
                if deg:
                    cur = int(indices[indptr[cur] + rng.integers(deg)])
                seen.append(cur)
        out[int(v)] = torch.unique(torch.tensor(seen, dtype=torch.long))
    return out


def expected_sets(cache, links):
    sets = []
    for s, d in links:
        u = torch.unique(torch.cat([cache[int(s)], cache[int(d)]])).tolist()     # utils.py:101-104
        rest = [v for v in u if v != s and v != d]
        sets.append([int(s), int(d)] + rest)                                     # utils.py:134-135
    return sets


@pytest.mark.parametrize("name,mode", [("usair_pos_k2", "pos"), ("cora_posplus_k3", "pos_plus")])
@pytest.mark.parametrize("source", ["foreign", "engine"])
def test_cached_walk_sets_are_extracted_bit_for_bit(eng, name, mode, source):
    w = workloads.make(name)
    link_index, y = w.split.all_links()
    links = link_index.T
    K, m, M = w.sign_k, 3, 8
    G = eng.graph(w.A)
    f = eng.features(w.X)
    for yy in (1, 0):
        sel = links[y == yy]
        if source == "foreign":
            cache = foreign_cache(w.A, sel, m, M, seed=11 + yy)
        else:
            cache = scaled.create_rw_cache(w.A, sel.T, None, m, M, seed=5 + yy, engine=eng)
            # what the engine cached is a walk cache: the node itself first among sorted unique ids
            k0 = int(sel[0, 0])
            assert k0 in cache and k0 in cache[k0].tolist() and cache[k0].tolist() == sorted(set(cache[k0].tolist()))
            assert len(cache) == len(np.unique(sel)) and len(cache[k0]) <= m * M + 1
        rw_kwargs = {"rw_m": m, "rw_M": M, "sign": True,
                     "cached_pos_rws": cache if yy == 1 else None, "cached_neg_rws": cache if yy == 0 else None}
        what = scaled.resolve(rw_kwargs, yy, sel.T, w.A.shape[0])
        assert what[0] == "sets" and what[3] == 0
        plan = eng.plan(G, eng.link_pairs(sel), mode=mode, num_hops=w.num_hops, sign_k=K, full_stats=True,
                        node_sets=eng.node_sets(*what[1:]))
        node_ptr, nodes, dists = (t.cpu().numpy() for t in plan.export_subgraphs())
        want = expected_sets(cache, sel)
        assert np.array_equal(np.diff(node_ptr), [len(s) for s in want])
        # every link, bit for bit (the export is ascending inside a hop: hop 0 reads min, max of the
        # pair where the reference's list reads src, dst — the rows keep src first)
        canon = [sorted(s[:2]) + s[2:] for s in want]
        assert np.array_equal(nodes, np.concatenate(canon))
        first = node_ptr[:-1]
        assert np.all(dists[first] == 0) and np.all(dists[first + 1] == 0) and int(dists.sum()) == len(nodes) - 2 * len(sel)
        rows = plan.run(f).cpu().numpy()
        row_ptr = plan.row_ptr().cpu().numpy()
        plan.close()
        # rows: a sample of links against the oracle's rw branch on the same sets
        pick = np.random.default_rng(3).choice(len(sel), 150, replace=False)
        kw = {"sign_k": K, "k_node_set_strategy": "intersection"}
        fn = oracle.get_PoS_prepped_ds if mode == "pos" else oracle.get_PoS_Plus_prepped_ds
        ref = fn(sel[pick].T, w.num_hops, w.A, w.X.astype(np.float64), yy, kw, dtype=np.float64,
                 rw_node_sets=[want[i] for i in pick])
        for j, i in enumerate(pick):
            got = rows[row_ptr[i]:row_ptr[i + 1]]
            exp = np.stack([ref[j]["x"]] + [ref[j][f"x{k}"] for k in range(1, K + 1)], axis=1)
            assert got.shape == exp.shape
            assert rel_err(got, exp) < TOL
    G.close()


def test_engine_walks_equal_their_own_cache(eng):
    """A plan that draws the walks itself (cfg.rw_m / rw_M / seed) and a plan on the cache
    s3grl_walk_sets built with the same seed extract the same sets and emit the same bits."""
    w = workloads.make("usair_pos_k2")
    link_index, _ = w.split.all_links()
    G = eng.graph(w.A)
    f = eng.features(w.X)
    L = eng.links(link_index)
    m, M, seed = 3, 20, 77
    own = eng.plan(G, L, mode="pos_plus", num_hops=1, sign_k=3, rw=(m, M, seed), full_stats=True)
    cache = scaled.create_rw_cache(G, link_index, None, m, M, seed=seed, engine=eng)
    what = scaled.resolve({"rw_m": m, "rw_M": M, "cached_pos_rws": cache}, 1, link_index, w.A.shape[0])
    cached = eng.plan(G, L, mode="pos_plus", num_hops=1, sign_k=3, full_stats=True,
                      node_sets=eng.node_sets(*what[1:]))
    for a, b in zip(own.export_subgraphs(), cached.export_subgraphs()):
        assert torch.equal(a, b)
    assert torch.equal(own.row_ptr(), cached.row_ptr()) and torch.equal(own.run(f), cached.run(f))
    own.close()
    cached.close()
    G.close()


def test_dropin_operators_use_the_callers_cache(eng):
    """Through the reference's operator API: y selects the cache (utils.py:94-97), other y raise
    ValueError (:98-99), an endpoint the cache does not hold raises KeyError like `dict[src]`,
    `unique_nodes` (one set per link, :106-107) is honoured, and with no cache at all the engine
    draws its own walks."""
    from s3grl_amd.tuned_SIGN import OptimizedSignOperations as ops, clear_cache

    w = workloads.make("usair_pos_k2")
    link_index, _ = w.split.all_links()
    li = torch.from_numpy(link_index[:, :40].copy())
    X = torch.from_numpy(w.X)
    kw = {"sign_k": 2, "k_node_set_strategy": "intersection"}
    pos = foreign_cache(w.A, link_index[:, :40].T, 2, 4, seed=1)
    neg = foreign_cache(w.A, link_index[:, :40].T, 2, 4, seed=2)
    rw = {"rw_m": 2, "rw_M": 4, "sign": True, "cached_pos_rws": pos, "cached_neg_rws": neg}
    args = (li, 3, w.A, 1.0, None, False, None, X)
    got = {yy: ops.get_PoS_prepped_ds(*args, yy, kw, rw) for yy in (1, 0)}
    for yy, cache in ((1, pos), (0, neg)):
        sets = expected_sets(cache, link_index[:, :40].T)
        ref = oracle.get_PoS_prepped_ds(link_index[:, :40], 3, w.A, w.X.astype(np.float64), yy, kw,
                                        dtype=np.float64, rw_node_sets=sets)
        for i in (0, 7, 39):
            for k in ("x", "x1", "x2"):
                assert rel_err(got[yy][i][k].numpy(), ref[i][k]) < TOL
        assert got[yy][0].y == yy
    with pytest.raises(ValueError, match="not 0/1"):
        ops.get_PoS_prepped_ds(*args, 2, kw, rw)
    short = {k: v for k, v in pos.items() if k != int(li[0, 0])}
    with pytest.raises(KeyError):
        ops.get_PoS_prepped_ds(*args, 1, kw, dict(rw, cached_pos_rws=short))
    # unique_nodes: one set per link, whatever the link's endpoints cached
    per_link = {(int(s), int(d)): [int(s), int(d), 0, 1, 2] for s, d in link_index[:, :40].T}
    lst = ops.get_PoS_Plus_prepped_ds(*args, 1, kw, {"rw_m": 2, "rw_M": 4, "sign": True, "cached_pos_rws": None,
                                                     "cached_neg_rws": None, "unique_nodes": per_link})
    sets = [[int(s), int(d)] + [v for v in (0, 1, 2) if v not in (s, d)] for s, d in link_index[:, :40].T]
    ref = oracle.get_PoS_Plus_prepped_ds(link_index[:, :40], 3, w.A, w.X.astype(np.float64), 1, kw,
                                         dtype=np.float64, rw_node_sets=sets)
    for i in range(40):
        assert lst[i].x.shape == ref[i]["x"].shape
        assert rel_err(lst[i]["x2"].numpy(), ref[i]["x2"]) < TOL
    # no cache: the engine's own walks
    own = ops.get_PoS_prepped_ds(*args, 1, kw, {"rw_m": 2, "rw_M": 4, "sign": True, "seed": 3})
    assert len(own) == 40 and own[0].x.shape == (2, w.X.shape[1] + 1)
    clear_cache()


def test_malformed_node_sets_are_rejected(eng):
    w = workloads.make("usair_pos_k2")
    n = w.A.shape[0]
    G = eng.graph(w.A)
    L = eng.link_pairs(np.array([[0, 1], [2, 3]]))
    ptr = np.zeros(n + 1, dtype=np.int64)
    ptr[1:] = 2
    with pytest.raises(ValueError, match="outside"):
        eng.plan(G, L, mode="pos", sign_k=2, node_sets=eng.node_sets(ptr, np.array([5, n]), False))
    bad = ptr.copy()
    bad[3] = 1
    bad[2] = 2
    with pytest.raises(ValueError, match="monotone"):
        eng.plan(G, L, mode="pos", sign_k=2, node_sets=eng.node_sets(bad, np.array([5, 6]), False))
    with pytest.raises(ValueError, match="num_sets"):
        eng.plan(G, L, mode="pos", sign_k=2, node_sets=eng.node_sets(ptr[:5], np.array([5, 6]), False))
    # per-link sets are never folded: the reversed link may carry another set
    Lr = eng.link_pairs(np.array([[0, 1], [1, 0]]))
    p = eng.plan(G, Lr, mode="pos", sign_k=2, node_sets=eng.node_sets(np.array([0, 2, 3]), np.array([5, 6, 7]), True))
    assert p.stats["folded_links"] == 0
    node_ptr, nodes, _ = (t.cpu().numpy() for t in p.export_subgraphs())
    assert nodes.tolist() == [0, 1, 5, 6, 0, 1, 7]
    p.close()
    G.close()
