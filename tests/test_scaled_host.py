"""Host logic of the ScaLed boundary (s3grl_amd/scaled.py) on CPU: which node sets the operators
take from the caller's rw_kwargs (reference utils.py:86-108) and the dict view of a walk cache."""
import numpy as np
import pytest
import torch

from s3grl_amd import scaled


def test_walk_cache_is_a_read_only_dict_of_tensors():
    c = scaled.WalkCache([2, 5, 9], [0, 2, 2, 5], [2, 7, 1, 9, 11])
    assert len(c) == 3 and list(c) == [2, 5, 9] and 5 in c and 3 not in c and bool(c)
    assert c[2].tolist() == [2, 7] and c[5].tolist() == [] and c[9].dtype == torch.int64
    with pytest.raises(KeyError):
        c[4]
    ptr, nodes = c.node_csr(12)
    assert ptr.tolist() == [0, 0, 0, 2, 2, 2, 2, 2, 2, 2, 5, 5, 5] and nodes.tolist() == [2, 7, 1, 9, 11]
    assert not scaled.WalkCache([], [0], [])


def test_resolve_follows_the_reference_branches():
    li = np.array([[0, 3], [1, 4]])
    pos = {0: torch.tensor([0, 2]), 1: torch.tensor([1]), 3: torch.tensor([3, 0]), 4: torch.tensor([4, 2, 2])}
    neg = {0: torch.tensor([0]), 1: torch.tensor([1]), 3: torch.tensor([3]), 4: torch.tensor([4])}
    rw = {"rw_m": 2, "rw_M": 3, "sign": True, "cached_pos_rws": pos, "cached_neg_rws": neg}
    assert scaled.resolve(None, 1, li, 6) is None and scaled.resolve({}, 1, li, 6) is None
    kind, ptr, nodes, per_link = scaled.resolve(rw, 1, li, 6)          # y = 1: the positives' cache
    assert kind == "sets" and per_link == 0 and ptr.tolist() == [0, 2, 3, 3, 5, 8, 8]
    assert nodes.tolist() == [0, 2, 1, 3, 0, 4, 2, 2]
    assert scaled.resolve(rw, 0, li, 6)[2].tolist() == [0, 1, 3, 4]    # y = 0: the negatives'
    with pytest.raises(ValueError, match="not 0/1"):                   # utils.py:98-99
        scaled.resolve(rw, 2, li, 6)
    with pytest.raises(KeyError):                                      # cache[src] of a missing node
        scaled.resolve(dict(rw, cached_pos_rws={0: torch.tensor([0])}), 1, li, 6)
    # empty / absent caches: unique_nodes per link (utils.py:106-107), else the engine's own walks
    un = {(0, 1): [0, 1, 5], (3, 4): [4, 3]}
    kind, ptr, nodes, per_link = scaled.resolve({"rw_m": 2, "rw_M": 3, "cached_pos_rws": {}, "unique_nodes": un}, 1, li, 6)
    assert (kind, per_link, ptr.tolist(), nodes.tolist()) == ("sets", 1, [0, 3, 5], [0, 1, 5, 4, 3])
    assert scaled.resolve({"rw_m": 2, "rw_M": 3, "seed": 9, "cached_pos_rws": None}, 1, li, 6) == ("walks", 2, 3, 9)
    # a WalkCache goes through without conversion
    c = scaled.WalkCache([0, 1, 3, 4], [0, 1, 2, 3, 4], [0, 1, 3, 4])
    assert scaled.resolve({"cached_neg_rws": c}, 0, li, 6)[1].tolist() == [0, 1, 2, 2, 3, 4, 4]
