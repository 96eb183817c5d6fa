"""World-size-2 gloo test of the sharding + reassembly logic (the N > 1 path).  The per-rank
compute is the oracle here (CPU), standing in for the engine call — this tests the distributed
plumbing, not the kernels."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.multiprocessing as mp

import oracle
from conftest import csr_from_undirected, load_extract
from s3grl_amd import parallel


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, plus, q):
    import torch.distributed as dist

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    g = load_extract("usair")
    n = int(g["num_nodes"])
    A = csr_from_undirected(n, g["edges"])
    X = np.random.default_rng(0).random((n, 6))
    links = g["links"].T
    kw = {"sign_k": 2, "k_node_set_strategy": "intersection"}
    fn = oracle.get_PoS_Plus_prepped_ds if plus else oracle.get_PoS_prepped_ds

    def compute(shard):
        rows, ptr, _ = oracle.collate_rows(fn(shard.numpy(), 1, A, X, 1, kw, dtype=np.float64), 2)
        if rows.shape[0] == 0:
            rows = np.zeros((0, 3, 7))
        return torch.from_numpy(rows), torch.from_numpy(ptr)

    rows, ptr, (lo, hi) = parallel.sharded_precompute(
        compute, links, rank=rank, world_size=world, cost=parallel.link_cost(A, links))
    full_rows, full_ptr, _ = oracle.collate_rows(fn(links, 1, A, X, 1, kw, dtype=np.float64), 2)
    ok = np.array_equal(ptr.numpy(), full_ptr) and np.array_equal(rows.numpy(), full_rows)
    # pair-aware shards (both directions of a pair on one rank): same result in the caller's order
    both = np.concatenate([links, links[::-1, :7]], axis=1)
    both = both[:, np.random.default_rng(1).permutation(both.shape[1])]
    rows2, ptr2, where = parallel.sharded_precompute(
        compute, both, rank=rank, world_size=world, cost=parallel.link_cost(A, both), pair_aware=True)
    full2, fptr2, _ = oracle.collate_rows(fn(both, 1, A, X, 1, kw, dtype=np.float64), 2)
    ok = ok and np.array_equal(ptr2.numpy(), fptr2) and np.array_equal(rows2.numpy(), full2)
    ok = ok and torch.is_tensor(where) and where.dtype == torch.int64
    q.put((rank, ok, lo, hi))
    dist.barrier()
    dist.destroy_process_group()


def _worker_fixed(rank, world, port, chunks, q):
    """Fixed-rows flavour (PoS / SoP: 2 rows per link), pipelined pieces, in-place padded
    all-gather.  The oracle stands in for the engine; the check is bit-equality with the unsharded
    result on every rank — the same assertion tests/test_gpu_parity.py makes with the engine."""
    import torch.distributed as dist

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    g = load_extract("usair")
    n = int(g["num_nodes"])
    A = csr_from_undirected(n, g["edges"])
    X = np.random.default_rng(0).random((n, 6))
    links = g["links"].T
    kw = {"sign_k": 2, "k_node_set_strategy": "intersection"}

    def rows_of(li):
        rows, _, _ = oracle.collate_rows(oracle.get_PoS_prepped_ds(li, 1, A, X, 1, kw, dtype=np.float64), 2)
        return rows.reshape(-1, 3, 7)

    calls = []

    def compute(piece, out):
        calls.append(int(piece.shape[1]))
        out.copy_(torch.from_numpy(rows_of(piece.numpy())))

    ok = True
    for _ in range(2):          # second call reuses the cached buffers
        rows, ptr, (lo, hi) = parallel.sharded_precompute(
            compute, links, rank=rank, world_size=world, cost=parallel.link_cost(A, links),
            rows_per_link=2, chunks=chunks, row_shape=(3, 7), dtype=torch.float64, device="cpu")
        full = rows_of(links)
        ok = ok and np.array_equal(rows.numpy(), full) and \
            np.array_equal(ptr.numpy(), np.arange(0, 2 * links.shape[1] + 1, 2))
    ok = ok and sum(calls) == 2 * (hi - lo)
    # the caller owns what it gets back: a second call of the same shape (pos, then neg of a split,
    # reference sgrl_link_pred.py:195-204) must not overwrite the first result ...
    flipped = np.ascontiguousarray(links[::-1])
    kwargs = dict(rank=rank, world_size=world, rows_per_link=2, chunks=chunks, row_shape=(3, 7),
                  dtype=torch.float64, device="cpu")
    first, _, _ = parallel.sharded_precompute(compute, links, **kwargs)
    second, _, _ = parallel.sharded_precompute(compute, flipped, **kwargs)
    ok = ok and first.data_ptr() != second.data_ptr() and np.array_equal(first.numpy(), rows_of(links)) \
        and np.array_equal(second.numpy(), rows_of(flipped))
    # ... unless the caller asked for the shared buffer (a benchmark loop)
    r1, _, _ = parallel.sharded_precompute(compute, links, reuse_buffers=True, **kwargs)
    r2, _, _ = parallel.sharded_precompute(compute, flipped, reuse_buffers=True, **kwargs)
    ok = ok and r1.data_ptr() == r2.data_ptr()
    calls.clear()
    # pair-aware shards: a reversed duplicate lands on its partner's rank, the reassembled tensor is
    # in the caller's order all the same, and a shard plan made once serves every step
    both = np.concatenate([links, links[::-1, :9]], axis=1)
    both = np.ascontiguousarray(both[:, np.random.default_rng(2).permutation(both.shape[1])])
    plan = parallel.ShardPlan(both, world, parallel.link_cost(A, both), pair_aware=True)
    for _ in range(2):
        rows_p, ptr_p, where = parallel.sharded_precompute(compute, both, shards=plan, **kwargs)
        ok = ok and np.array_equal(rows_p.numpy(), rows_of(both)) and int(ptr_p[-1]) == 2 * both.shape[1]
    mine = set(map(tuple, both[:, where.numpy()].T.tolist()))
    ok = ok and all(((d, s_) in mine) == ((s_, d) in mine) or (d, s_) not in set(map(tuple, both.T.tolist()))
                    for s_, d in mine)
    rows_l, _, where_l = parallel.sharded_precompute(compute, both, shards=plan, gather=False,
                                                     **{k: v for k, v in kwargs.items() if k != "chunks"})
    ok = ok and np.array_equal(rows_l.numpy(), rows_of(both[:, where_l.numpy()]))
    calls.clear()
    # operator 0 filled by every rank itself, operators 1.. exchanged: the same tensor, fewer bytes on the wire
    full_both = rows_of(both)

    def fill0(fl):
        fl[:, :, 0, :] = torch.from_numpy(full_both).view(both.shape[1], 2, 3, 7)[:, :, 0, :]

    for sp_ in (None, plan):
        rows_x, _, _ = parallel.sharded_precompute(compute, both, shards=sp_, local_operator0=fill0,
                                                   cost=parallel.link_cost(A, both), **kwargs)
        ok = ok and np.array_equal(rows_x.numpy(), full_both)
    calls.clear()
    # reversed duplicates rebuilt from their primaries instead of exchanged (mirror_rows): a compute whose
    # rows of (d, s) ARE the rows of (s, d) in swapped order, like the engine's; lists with (a,b),(b,a),(a,b)
    # runs and duplicates of one direction; with and without the local operator 0
    def rows_sym(p):
        p = np.asarray(p)
        sd = np.stack([p[0] * 1000.0 + p[1], p[1] * 1000.0 + p[0]], 1)
        return (sd[:, :, None, None] + 0.5 * np.arange(3)[None, None, :, None] +
                0.25 * np.arange(7)[None, None, None, :]).reshape(-1, 3, 7)

    def compute_sym(piece, out):
        out.copy_(torch.from_numpy(rows_sym(piece.numpy() if torch.is_tensor(piece) else piece)))

    tri = np.concatenate([both, both[::-1, :20], both[:, :6], both[::-1, 3:9]], axis=1)
    tri = np.ascontiguousarray(tri[:, np.random.default_rng(5).permutation(tri.shape[1])])
    plan_t = parallel.ShardPlan(tri, world, None, pair_aware=True)
    full_tri = rows_sym(tri)
    sent = sum(sum(nr) for nr in plan_t.transport(chunks, "cpu")[3])
    ok = ok and 0 < tri.shape[1] - sent <= int(plan_t.reverse_of_previous.sum())   # (a piece boundary may part a pair)

    def fill0_t(fl):
        fl[:, :, 0, :] = torch.from_numpy(full_tri).view(tri.shape[1], 2, 3, 7)[:, :, 0, :]

    for f0 in (None, fill0_t):
        rows_m, _, _ = parallel.sharded_precompute(compute_sym, tri, shards=plan_t, mirror_rows=True,
                                                   local_operator0=f0, **kwargs)
        ok = ok and np.array_equal(rows_m.numpy(), full_tri)
    # the cheapest links computed by every rank itself instead of exchanged (ShardPlan(replicate=…)): the same
    # tensor, fewer links on the wire; `where` = what this rank computed (its own share + the replicated links)
    cost_t = parallel.link_cost(A, tri).astype(np.float64)
    rep = parallel.replicate_cheapest(tri, cost_t, 0.3)
    plan_r = parallel.ShardPlan(tri, world, cost_t, pair_aware=True, replicate=rep)
    ok = ok and 0 < rep.sum() <= 0.3 * tri.shape[1] and plan_r.rep_start == tri.shape[1] - rep.sum()
    both_dirs = set(map(tuple, tri.T.tolist()))
    ok = ok and all(rep[i] == rep[j] for i, (a_, b_) in enumerate(tri.T.tolist()) if (b_, a_) in both_dirs
                    for j in [next(k for k, ab in enumerate(tri.T.tolist()) if ab == [b_, a_])])
    for f0, mr in ((None, False), (fill0_t, True)):
        rows_r, _, where_r = parallel.sharded_precompute(compute_sym, tri, shards=plan_r, mirror_rows=mr,
                                                         local_operator0=f0, **kwargs)
        ok = ok and np.array_equal(rows_r.numpy(), full_tri)
        ok = ok and set(np.flatnonzero(rep).tolist()) <= set(where_r.tolist())
    # gather=False hands back the local shard only
    rows_l, ptr_l, _ = parallel.sharded_precompute(
        compute, links, rank=rank, world_size=world, cost=parallel.link_cost(A, links),
        rows_per_link=2, row_shape=(3, 7), dtype=torch.float64, device="cpu", gather=False)
    ok = ok and np.array_equal(rows_l.numpy(), rows_of(links[:, lo:hi])) and int(ptr_l[0]) == 0 \
        and int(ptr_l[-1]) == 2 * (hi - lo)
    q.put((rank, ok, lo, hi))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world,chunks", [(2, 1), (2, 3), (3, 2)])
def test_sharded_fixed_rows_pipelined(world, chunks):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker_fixed, args=(r, world, port, chunks, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in procs]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert all(ok for _, ok, _, _ in res)
    spans = sorted((lo, hi) for _, _, lo, hi in res)
    assert spans[0][0] == 0 and spans[-1][1] == 34
    assert all(spans[i][1] == spans[i + 1][0] for i in range(world - 1))


@pytest.mark.parametrize("plus", [False, True])
def test_sharded_precompute_world2(plus):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, plus, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in procs]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert all(ok for _, ok, _, _ in res)
    spans = sorted((lo, hi) for _, _, lo, hi in res)
    assert spans[0][0] == 0 and spans[0][1] == spans[1][0] and spans[1][1] == 34


def test_shard_bounds_balanced_and_contiguous():
    cost = np.array([1, 1, 1, 1, 100, 1, 1, 1])
    b = parallel.shard_bounds(8, 2, cost)
    assert b[0] == 0 and b[-1] == 8 and b == sorted(b)
    assert parallel.shard_bounds(10, 4) == [0, 2, 5, 7, 10]
    assert parallel.shard_bounds(0, 3) == [0, 0, 0, 0]
    # pieces of a shard: contiguous, cover it, empty pieces allowed
    assert parallel.chunk_bounds(5, 9, 2) == [5, 7, 9]
    assert parallel.chunk_bounds(5, 6, 3, cost=np.ones(10))[0] == 5
    assert parallel.chunk_bounds(5, 6, 3, cost=np.ones(10))[-1] == 6


def test_khop_cost_tracks_ball_size():
    g = load_extract("usair")
    n = int(g["num_nodes"])
    A = csr_from_undirected(n, g["edges"])
    links = g["links"].T
    c1 = parallel.khop_cost(A, links, 1)
    assert np.array_equal(c1, parallel.link_cost(A, links) + 1)   # 1 + deg per endpoint
    c2 = parallel.khop_cost(A, links, 2)
    assert (c2 >= c1).all()


def test_shard_plan_with_replicated_links_is_a_permutation_that_keeps_pairs_together():
    """ShardPlan(replicate=…): every link appears once; the sharded part comes first, rank by rank, the
    links every rank computes itself behind it; both directions of a pair are on the same side and next to
    each other; the "reverse of the previous" marks never reach across a rank boundary or into the
    replicated part; replicate_cheapest takes whole pairs, cheapest first."""
    rng = np.random.default_rng(7)
    n = 200
    li = rng.integers(0, n, size=(2, 600))
    li = li[:, li[0] != li[1]]
    li = np.concatenate([li, li[::-1, :250], li[:, :30]], axis=1)
    li = np.ascontiguousarray(li[:, rng.permutation(li.shape[1])])
    L = li.shape[1]
    cost = rng.random(L) * 100 + 1
    rep = parallel.replicate_cheapest(li, cost, 0.25)
    assert 0 < rep.sum() <= 0.25 * L
    key = np.minimum(li[0], li[1]) * n + np.maximum(li[0], li[1])
    for k in np.unique(key):                                   # whole pairs
        assert len(set(rep[key == k].tolist())) == 1
    mean_cost = lambda m: np.mean([cost[key == k].mean() for k in np.unique(key[m])])
    assert mean_cost(rep) < mean_cost(~rep)
    for world in (2, 3, 8):
        sp = parallel.ShardPlan(li, world, cost, pair_aware=True, replicate=rep)
        order = sp.order.numpy()
        assert sorted(order.tolist()) == list(range(L))
        assert sp.rep_start == L - rep.sum() and sp.bounds[0] == 0 and sp.bounds[-1] == sp.rep_start
        assert not rep[order[:sp.rep_start]].any() and rep[order[sp.rep_start:]].all()
        rank_of = np.full(L, world)
        for r in range(world):
            rank_of[order[sp.bounds[r]:sp.bounds[r + 1]]] = r
        for k in np.unique(key):                               # a pair lives on one rank (or is replicated)
            assert len(set(rank_of[key == k].tolist())) == 1
        rev = sp.reverse_of_previous
        links = sp.links.numpy()
        for i in np.flatnonzero(rev):
            assert links[0, i] == links[1, i - 1] and links[1, i] == links[0, i - 1] and not rev[i - 1]
            assert i not in sp.bounds and i != sp.rep_start


def _rows_sym(p):
    """rows of (d, s) = rows of (s, d) in swapped order, like the engine's (3 operators, 7 columns)."""
    p = np.asarray(p)
    sd = np.stack([p[0] * 1000.0 + p[1], p[1] * 1000.0 + p[0]], 1)
    return (sd[:, :, None, None] + 0.5 * np.arange(3)[None, None, :, None] +
            0.25 * np.arange(7)[None, None, None, :]).reshape(-1, 3, 7)


def _pair_rich_list(seed, n=120, m=260):
    rng = np.random.default_rng(seed)
    li = rng.integers(0, n, size=(2, m))
    li = li[:, li[0] != li[1]]
    li = np.concatenate([li, li[::-1, :m // 2], li[:, :10], li[::-1, 5:15]], axis=1)
    return np.ascontiguousarray(li[:, rng.permutation(li.shape[1])])


@pytest.mark.parametrize("replicate", [False, True])
@pytest.mark.parametrize("gather", [True, False])
def test_one_rank_with_a_pair_aware_plan_keeps_the_callers_order(replicate, gather):
    """world_size == 1 without the collective path (what a single-GPU caller of the multi-GPU entry gets):
    with a pair-aware ShardPlan the list is computed in the plan's grouped order — and must come back in
    the caller's, the replicated links included (they used to be dropped, the rest came back grouped)."""
    li = _pair_rich_list(3)
    L = li.shape[1]
    cost = np.random.default_rng(0).random(L) + 1.0
    rep = parallel.replicate_cheapest(li, cost, 0.3) if replicate else None
    plan = parallel.ShardPlan(li, 1, cost, pair_aware=True, replicate=rep)
    if replicate:
        assert plan.rep_start < L

    def compute(piece, out):
        out.copy_(torch.from_numpy(_rows_sym(piece.numpy() if torch.is_tensor(piece) else piece)))

    if replicate and not gather:
        with pytest.raises(AssertionError):      # replicated links belong to the gathered flavour
            parallel.sharded_precompute(compute, li, rank=0, world_size=1, shards=plan, gather=False,
                                        rows_per_link=2, row_shape=(3, 7), dtype=torch.float64, device="cpu")
        return
    rows, ptr, where = parallel.sharded_precompute(compute, li, rank=0, world_size=1, shards=plan, gather=gather,
                                                   rows_per_link=2, row_shape=(3, 7), dtype=torch.float64,
                                                   device="cpu")
    if gather:
        assert np.array_equal(rows.numpy(), _rows_sym(li))
        assert np.array_equal(ptr.numpy(), np.arange(0, 2 * L + 1, 2))
    else:                                        # the local shard: rows in the order of `where`
        assert np.array_equal(rows.numpy(), _rows_sym(li[:, where.numpy()]))
    assert sorted(where.tolist()) == list(range(L))


def _worker_everything(rank, world, port, chunks, q):
    """All layers of the exchange together, as bench.py --gpus N runs them: a pair-aware ShardPlan with
    replicated links, pipelined pieces, reversed duplicates rebuilt from their primaries, operator 0 filled
    locally.  Bit-equal to the unsharded tensor on every rank."""
    import torch.distributed as dist

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    li = _pair_rich_list(11, n=300, m=900)
    L = li.shape[1]
    cost = np.random.default_rng(1).random(L) * 50 + 1
    full = _rows_sym(li)
    computed = []

    def compute(piece, out):
        p = piece.numpy() if torch.is_tensor(piece) else piece
        computed.append(p.shape[1])
        out.copy_(torch.from_numpy(_rows_sym(p)))

    def fill0(fl):
        fl[:, :, 0, :] = torch.from_numpy(full).view(L, 2, 3, 7)[:, :, 0, :]

    ok = True
    rep = parallel.replicate_cheapest(li, cost, 0.2)
    for replicate in (None, rep):
        plan = parallel.ShardPlan(li, world, cost, pair_aware=True, replicate=replicate)
        for f0, mr in ((None, False), (fill0, True), (None, True), (fill0, False)):
            for _ in range(2):      # the second step reuses plan, transport tables and buffers
                rows, ptr, where = parallel.sharded_precompute(
                    compute, li, rank=rank, world_size=world, shards=plan, rows_per_link=2, chunks=chunks,
                    row_shape=(3, 7), dtype=torch.float64, device="cpu", mirror_rows=mr, local_operator0=f0,
                    reuse_buffers=True)
                ok = ok and np.array_equal(rows.numpy(), full) and int(ptr[-1]) == 2 * L
        own = plan.bounds[rank + 1] - plan.bounds[rank]
        ok = ok and len(where) == own + (L - plan.rep_start)
    q.put((rank, ok, 0, 0))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world,chunks", [(8, 3)])
def test_world8_every_exchange_layer_together(world, chunks):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker_everything, args=(r, world, port, chunks, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=240) for _ in procs]
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    assert len(res) == world and all(ok for _, ok, _, _ in res)
