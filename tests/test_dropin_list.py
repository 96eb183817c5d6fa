"""The lazy per-link list the drop-in operators return (s3grl_amd.tuned_SIGN.LinkDataList) and the
host-side pieces around it — CPU only: list behaviour the reference's caller relies on
(sgrl_link_pred.py:195-205, utils.py:472-480), the PyG collate hook, the staging pool, the
upload cache."""
import sys
import types

import numpy as np
import pytest
import torch

from s3grl_amd import tuned_SIGN as ts


def _chunk(L, K, F, y, seed, ragged=False):
    rng = np.random.default_rng(seed)
    cnt = rng.integers(2, 5, size=L) if ragged else np.full(L, 2)
    ptr = np.concatenate([[0], np.cumsum(cnt)]).astype(np.int64)
    rows = torch.from_numpy(rng.random((int(ptr[-1]), K + 1, F + 1)).astype(np.float32))
    return (rows, ptr, y)


def _eager(chunks, K):
    out = []
    for rows, ptr, y in chunks:
        for a, b in zip(ptr[:-1], ptr[1:]):
            out.append({"y": y, **{("x" if k == 0 else f"x{k}"): rows[a:b, k, :] for k in range(K + 1)}})
    return out


def _same(d, e, K):
    return d.y == e["y"] and all(torch.equal(d[n], e[n]) for n in ["x"] + [f"x{k}" for k in range(1, K + 1)])


@pytest.mark.parametrize("ragged", [False, True])
def test_list_protocol(ragged):
    K, F = 3, 5
    c1, c2 = _chunk(7, K, F, 1, 0, ragged), _chunk(4, K, F, 0, 1, ragged)
    pos, neg = ts.LinkDataList([c1], K), ts.LinkDataList([c2], K)
    want = _eager([c1, c2], K)
    assert len(pos) == 7 and len(neg) == 4
    both = pos + neg                                   # sgrl_link_pred.py:204
    assert isinstance(both, ts.LinkDataList) and len(both) == 11
    assert all(_same(d, e, K) for d, e in zip(both, want))            # iteration
    assert all(_same(both[i], want[i], K) for i in range(11))         # indexing
    assert _same(both[-1], want[-1], K) and _same(both[-11], want[0], K)
    with pytest.raises(IndexError):
        both[11]
    assert [d.y for d in both[5:9]] == [1, 1, 0, 0]                    # slicing -> plain list
    assert both[0].x.shape == (int(c1[1][1]), F + 1) and both[0].num_features == F + 1
    # mixing with real lists degrades to a real list
    extra = [both[0]]
    assert isinstance(both + extra, list) and len(both + extra) == 12
    assert isinstance(extra + both, list) and len(extra + both) == 12
    # the hybrid flow writes new keys into the elements it iterates over (utils.py:472-480)
    combined = []
    for sup, sop in zip(pos, ts.LinkDataList([c1], K)):
        for k in range(K + 1, 2 * K):
            sup[f"x{k}"] = sop[f"x{k - K + 1}"]
        combined.append(sup)
    assert len(combined) == 7 and torch.equal(combined[3][f"x{K + 1}"], want[3]["x2"])
    assert "11 links" in repr(both)


def test_collate_matches_concatenation():
    K, F = 2, 4
    c1, c2 = _chunk(5, K, F, 1, 2, True), _chunk(3, K, F, 0, 3, True)
    both = ts.LinkDataList([c1], K) + ts.LinkDataList([c2], K)
    rows, ptr, y = both.collate()
    assert torch.equal(rows, torch.cat([c1[0], c2[0]]))
    assert ptr.tolist() == np.concatenate([c1[1], c2[1][1:] + c1[1][-1]]).tolist()
    assert y.tolist() == [1] * 5 + [0] * 3
    data, slices = both.collate_pyg()
    for k, name in enumerate(["x", "x1", "x2"]):
        assert torch.equal(data[name], rows[:, k, :]) and data[name].is_contiguous()
        assert torch.equal(slices[name], ptr)
    assert torch.equal(data.y, y) and slices["y"].tolist() == list(range(9))
    # a single chunk collates without copying
    one = ts.LinkDataList([c1], K)
    assert one.collate()[0].data_ptr() == c1[0].data_ptr()


def test_pyg_collate_hook_with_a_stand_in_module(monkeypatch):
    """PyG is absent from this image: a minimal stand-in with PyG's layout (InMemoryDataset with a
    static `collate`) checks that the hook routes a LinkDataList to the fast path and leaves every
    other argument to the original."""
    calls = []

    class InMemoryDataset:
        @staticmethod
        def collate(data_list):
            calls.append(len(data_list))
            return "orig", None

    tg = types.ModuleType("torch_geometric")
    tgd = types.ModuleType("torch_geometric.data")
    tgd.InMemoryDataset = InMemoryDataset
    tg.data = tgd
    monkeypatch.setitem(sys.modules, "torch_geometric", tg)
    monkeypatch.setitem(sys.modules, "torch_geometric.data", tgd)
    assert ts.install_pyg_fast_collate() and ts.install_pyg_fast_collate()   # idempotent
    lst = ts.LinkDataList([_chunk(6, 2, 3, 1, 5)], 2)

    class Sub(InMemoryDataset):                       # the reference calls self.collate(...)
        pass

    data, slices = Sub().collate(lst + lst)
    assert calls == [] and data.x.shape == (24, 4) and len(slices["y"]) == 13
    assert Sub().collate([1, 2, 3]) == ("orig", None) and calls == [3]


def test_staging_pool_recycles_only_dead_blocks(monkeypatch):
    allocs = []

    def fake_pinned(n):
        allocs.append(n)
        return torch.empty(n, dtype=torch.float32)

    monkeypatch.setattr(ts, "_alloc_pinned", fake_pinned)
    ts._pool.clear()
    a = ts._staging(1000)
    a.fill_(1.0)
    view = a[10:20].view(2, 5)                        # a per-link view cut from the list's tensor
    b = ts._staging(1000)                             # `a` is alive: a second block
    assert len(allocs) == 2 and a.data_ptr() != b.data_ptr()
    del a
    c = ts._staging(500)                              # the view keeps the first block busy
    assert len(allocs) == 3
    del view, c
    d = ts._staging(900)                              # now a dead block is handed out again
    assert len(allocs) == 3
    e = ts._staging(100)
    assert len(allocs) == 3 and e.data_ptr() != d.data_ptr()
    ts._pool.clear()


def test_staging_pool_is_capped(monkeypatch):
    """Results beyond S3GRL_PINNED_CAP_BYTES are not page-locked (the caller gets a pageable copy);
    dead blocks make room first."""
    monkeypatch.setattr(ts, "_alloc_pinned", lambda n: torch.empty(n, dtype=torch.float32))
    monkeypatch.setenv("S3GRL_PINNED_CAP_BYTES", str(40 << 20))      # two 16 MiB blocks fit, three do not
    ts._pool.clear()
    a = ts._staging(10)
    b = ts._staging(10)
    assert a is not None and b is not None and len(ts._pool) == 2
    assert ts._staging(10) is None                     # both alive, a third block would pass the cap
    del a, b
    c = ts._staging(10 + (4 << 20))                    # a 32 MiB block: both dead blocks are dropped for it
    assert c is not None and len(ts._pool) == 1 and ts._pool[0][0].numel() * 4 == 32 << 20
    del c
    ts._pool.clear()


def test_upload_cache_detects_new_and_mutated_inputs(monkeypatch):
    import scipy.sparse as ssp

    made = []

    class Handle:
        def __init__(self, kind):
            self.kind, self.closed = kind, False
            made.append(self)

        def close(self):
            self.closed = True

    class FakeEngine:
        def graph(self, A, directed=False, A_csc=None):
            return Handle("g")

        def features(self, x):
            return Handle("x")

    monkeypatch.setattr(ts._engine, "default_engine", lambda *a: FakeEngine())
    ts._cache.clear()
    A = ssp.csr_matrix(np.array([[0, 1, 0], [1, 0, 1], [0, 1, 0]]))
    x = torch.ones(3, 2)
    _, g1, x1 = ts._device_inputs(A, x)
    _, g2, x2 = ts._device_inputs(A, x)
    assert g1 is g2 and x1 is x2 and len(made) == 2
    B = ssp.csr_matrix(np.array([[0, 0, 1], [0, 0, 1], [1, 1, 0]]))    # same shape, same nnz
    _, g3, _ = ts._device_inputs(B, x)
    assert g3 is not g1 and g1.closed
    B.indices[:] = B.indices[::-1].copy()                              # mutated in place
    _, g4, _ = ts._device_inputs(B, x)
    assert g4 is not g3
    x.add_(1.0)                                                        # in-place change of x
    _, _, x3 = ts._device_inputs(B, x)
    assert x3 is not x1 and x1.closed
    xn = np.ones((3, 2), dtype=np.float32)
    _, _, x4 = ts._device_inputs(B, xn)
    xn[0, 0] = 5
    _, _, x5 = ts._device_inputs(B, xn)
    assert x5 is not x4
    ts._cache.clear()


# ---- SoP: how often a pair counts is decided from powers_of_A[0], not from A.data ------------------------
class _SparseTensorLike:
    """What the drop-in may see of torch_sparse.SparseTensor: `coo()` -> (row, col, value), `nnz()`."""

    def __init__(self, row, col):
        self._r, self._c = torch.as_tensor(row), torch.as_tensor(col)

    def coo(self):
        return self._r, self._c, None

    def nnz(self):
        return int(self._r.numel())


class _CountOnly:
    def __init__(self, n):
        self._n = n

    def nnz(self):
        return self._n


def _multi_edge_index(seed=0, n=12, m=30, dup=9):
    rng = np.random.default_rng(seed)
    e = np.unique(np.sort(rng.integers(0, n, size=(m, 2)), axis=1), axis=0)
    e = e[e[:, 0] != e[:, 1]]
    both = np.concatenate([e, e[rng.choice(len(e), dup)]])
    return n, np.concatenate([both, both[:, ::-1]]).T


def test_sop_multiplicity_follows_the_callers_operator():
    """sgrl_link_pred.py:102-114,161-172: Â is built from SparseTensor(row, col) of the edge_index as
    it stands — duplicates are entries, weights are not.  A's data cannot tell (scipy sums both)."""
    import scipy.sparse as ssp
    from s3grl_amd.dataset import GlobalOperators, coalesce, train_graph

    n, ei = _multi_edge_index()
    A = train_graph(ei, n)                                   # int ones, duplicates summed
    assert A.data.max() >= 2
    want = A.data.astype(np.float32)
    # uncoalesced edge_index: entries exposed, count only, the twin's stand-in
    for ops in ([_SparseTensorLike(ei[0], ei[1])] * 2, [_CountOnly(ei.shape[1])] * 2,
                GlobalOperators(2, ei.shape[1])):
        assert np.array_equal(ts._multiplicity_of(A, ops), want)
    # use_coalesce: one entry per pair, the counts have become integer WEIGHTS -> every pair once
    cei, cw = coalesce(ei, np.ones(ei.shape[1], dtype=np.int64), n)
    Aw = train_graph(cei, n, cw)
    assert np.array_equal(Aw.data, A.data) and cei.shape[1] == A.nnz
    for ops in ([_SparseTensorLike(cei[0], cei[1])], [_CountOnly(cei.shape[1])], GlobalOperators(1, cei.shape[1])):
        assert ts._multiplicity_of(Aw, ops) is None
    # a placeholder says nothing: structural
    assert ts._multiplicity_of(A, [None, None]) is None and ts._multiplicity_of(A, GlobalOperators(3)) is None
    # duplicates AND weights: only the entries can tell; a bare count cannot
    w = np.arange(1, ei.shape[1] + 1, dtype=np.int64)
    Adw = train_graph(ei, n, w)
    assert np.array_equal(ts._multiplicity_of(Adw, [_SparseTensorLike(ei[0], ei[1])]), want)
    with pytest.raises(ValueError):
        ts._multiplicity_of(Adw, [_CountOnly(ei.shape[1])])
    # float weights, no duplicates
    Af = ssp.csr_matrix((np.random.default_rng(1).random(cei.shape[1]), (cei[0], cei[1])), shape=(n, n))
    assert ts._multiplicity_of(Af, [_CountOnly(cei.shape[1])]) is None
    # an operator of another graph is refused
    with pytest.raises(ValueError):
        ts._multiplicity_of(A, [_SparseTensorLike(ei[0][:-2], (ei[1][:-2] + 1) % n)])
    with pytest.raises(ValueError):
        ts._multiplicity_of(A, [_CountOnly(A.nnz - 1)])


def test_coalesce_twin():
    """torch_sparse.coalesce as `use_coalesce` calls it: sorted pairs, weights of duplicates added."""
    from s3grl_amd.dataset import coalesce

    ei = np.array([[2, 0, 2, 1, 0], [1, 3, 1, 0, 3]])
    out, w = coalesce(ei, np.array([1, 2, 3, 4, 5]), 4)
    assert out.tolist() == [[0, 1, 2], [3, 0, 1]] and w.tolist() == [7, 4, 4]
    out, w = coalesce(ei, None, 4)
    assert out.tolist() == [[0, 1, 2], [3, 0, 1]] and w is None


def test_host_result_memory_and_copy_threads():
    """The host side of the default result path (tuned_SIGN._to_host): fresh pageable memory on a 2 MiB boundary
    that asked for huge pages, and the module's own copy threads (exact copies, any length, reusable pool)."""
    for n in (1, 1000, (1 << 19) + 3, 3 * (1 << 20) + 17):
        t = ts._huge_empty(n)
        assert t.shape == (n,) and t.dtype == torch.float32 and t.is_contiguous() and not t.is_pinned()
        assert t.data_ptr() % (1 << 21) == 0 or n * 4 < (1 << 21)
        src = torch.arange(n, dtype=torch.float32)
        ts._parallel_copy(t, src)
        assert torch.equal(t, src)
    assert ts._usable_cpus() >= 1
    # a second copy reuses the pool; views of a result stay valid after the base name is dropped
    big = ts._huge_empty(1 << 21)
    ts._parallel_copy(big, torch.ones(1 << 21))
    view = big[5:9]
    del big
    assert view.tolist() == [1.0, 1.0, 1.0, 1.0]
