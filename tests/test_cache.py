"""On-disk bundle (s3grl_amd/cache.py): round trip, key strings, staleness, atomicity."""
import numpy as np
import pytest

from s3grl_amd import cache


def _bundle(L=7, K=2, F=3, seed=0):
    rng = np.random.default_rng(seed)
    R = rng.integers(2, 5, L)
    row_ptr = np.concatenate([[0], np.cumsum(R)]).astype(np.int64)
    rows = rng.standard_normal((int(row_ptr[-1]), K + 1, 1 + F)).astype(np.float32)
    y = rng.integers(0, 2, L)
    return rows, row_ptr, y


def test_data_appendix_matches_reference_strings():
    # sgrl_link_pred.py:797-806
    assert cache.data_appendix(num_hops=3, node_label="zo", ratio_per_hop=1.0, seed=1) == "_h3_zo_rph10_seed1"
    assert cache.data_appendix(num_hops=2, node_label="drnl", ratio_per_hop=0.5, seed=7,
                               max_nodes_per_hop=100, use_valedges_as_input=True) \
        == "_h2_drnl_rph05_seed7_mnph100_uvai"
    assert cache.data_appendix(num_hops=1, node_label="zo", ratio_per_hop=1.0, seed=3, m=3, M=20,
                               dropedge=0.0) == "_m3_M20_dropedge0.0_seed3"
    assert cache.bundle_name("train") == "SEAL_train_data.s3grl"
    assert cache.bundle_name("valid", 50) == "SEAL_valid_data_50.s3grl"
    d = cache.cache_dir("dataset/Cora", "_h3_zo_rph10_seed1", mode="pos_plus", sign_k=3)
    assert str(d) == "dataset/Cora_seal_h3_zo_rph10_seed1_pos_plus_k3_intersection/processed"
    assert cache.operator_tag(mode="sop", sign_k=5) == "_sop_k5"
    with pytest.raises(ValueError):
        cache.operator_tag(mode="gcn", sign_k=1)


def test_round_trip_and_alignment(tmp_path):
    rows, row_ptr, y = _bundle()
    p = cache.save(tmp_path / "a" / "SEAL_train_data.s3grl", rows, row_ptr, y, {"sign_k": 2})
    hdr = cache.read_header(p)
    assert all(d["offset"] % cache.ALIGN == 0 for d in hdr["arrays"].values())
    r2, p2, y2, meta = cache.load(p)
    np.testing.assert_array_equal(r2, rows)
    np.testing.assert_array_equal(p2, row_ptr)
    np.testing.assert_array_equal(y2, y)
    assert meta == {"sign_k": 2, "num_links": 7}
    import torch

    r3, p3, y3, _ = cache.load(p, device="cpu")
    assert torch.equal(r3, torch.from_numpy(rows)) and r3.dtype == torch.float32 and p3.dtype == torch.int64
    assert not list((tmp_path / "a").glob("*.tmp*"))


def test_empty_split(tmp_path):
    rows = np.zeros((0, 3, 4), np.float32)
    p = cache.save(tmp_path / "e.s3grl", rows, np.zeros(1, np.int64), np.zeros(0, np.int64))
    r, ptr, y, meta = cache.load(p)
    assert r.shape == (0, 3, 4) and list(ptr) == [0] and meta["num_links"] == 0


def test_rejects_inconsistent_input(tmp_path):
    rows, row_ptr, y = _bundle()
    with pytest.raises(ValueError):
        cache.save(tmp_path / "x.s3grl", rows[:-1], row_ptr, y)
    with pytest.raises(ValueError):
        cache.save(tmp_path / "x.s3grl", rows.astype(np.float64), row_ptr, y)
    (tmp_path / "junk.s3grl").write_bytes(b"not a bundle at all")
    with pytest.raises(ValueError):
        cache.load(tmp_path / "junk.s3grl")


def test_get_or_compute_reuses_only_matching_settings(tmp_path):
    calls = []

    def make(seed):
        def fn():
            calls.append(seed)
            return _bundle(seed=seed) + ({"made_by": seed},)
        return fn

    p = tmp_path / "SEAL_test_data.s3grl"
    a = cache.get_or_compute(p, make(1), expect={"sign_k": 3, "mode": "pos"})
    b = cache.get_or_compute(p, make(2), expect={"sign_k": 3, "mode": "pos"})          # hit
    np.testing.assert_array_equal(a[0], b[0])
    assert calls == [1] and b[3]["made_by"] == 1
    c = cache.get_or_compute(p, make(3), expect={"sign_k": 5, "mode": "pos"})          # stale: recompute
    assert calls == [1, 3] and c[3]["sign_k"] == 5
    p.write_bytes(b"garbage")                                                          # corrupt: recompute
    cache.get_or_compute(p, make(4), expect={"sign_k": 5, "mode": "pos"})
    assert calls == [1, 3, 4]


def test_header_length_around_the_alignment_boundary(tmp_path):
    """The first blob's offset depends on the header's length, which depends on the offsets it
    records: whatever the meta's size, the written header must not run into the first blob."""
    rows = np.arange(2 * 2 * 3, dtype=np.float32).reshape(2, 2, 3)
    ptr, y = np.array([0, 2], dtype=np.int64), np.array([1], dtype=np.int64)
    for pad in list(range(3600, 4000, 7)) + list(range(7700, 8200, 11)):
        p = cache.save(tmp_path / "b.s3grl", rows, ptr, y, {"pad": "x" * pad})
        r, rp, yy, meta = cache.load(p)
        assert np.array_equal(r, rows) and np.array_equal(rp, ptr) and len(meta["pad"]) == pad
        hdr = cache.read_header(p)
        import json as _json

        assert 16 + len(_json.dumps(hdr, sort_keys=True).encode()) <= hdr["arrays"]["rows"]["offset"]
