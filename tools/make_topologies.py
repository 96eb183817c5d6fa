#!/usr/bin/env python3
"""Extract graph TOPOLOGY (undirected unique edge lists) of the public datasets BASELINE.json's
configs are quoted on into compact .npz data files, so benches and tests can run where
/root/reference does not exist (the GPU box).

Data only: no reference source text is read or copied.  Inputs are the public Planetoid pickles
(ind.cora.graph, ind.pubmed.graph: dict node -> neighbour list) and the SEAL USAir edge list
(first two whitespace columns; ids remapped by sorted string order like data_utils.py:76-93
of the reference does).  Run in the build container only:

    python tools/make_topologies.py /root/reference/data
"""
import pickle
import sys
from pathlib import Path

import numpy as np

OUT = Path(__file__).resolve().parent.parent / "s3grl_amd" / "data"


def planetoid(path):
    with open(path, "rb") as f:
        g = pickle.load(f, encoding="latin1")
    n = max(max(g.keys()), max(max(v) for v in g.values() if len(v))) + 1
    e = set()
    for u, nbrs in g.items():
        for v in nbrs:
            if u != v:
                e.add((min(u, v), max(u, v)))
    e = np.array(sorted(e), dtype=np.int32)
    return n, e


def seal_txt(path):
    rows = [ln.split()[:2] for ln in open(path) if ln.strip()]
    names = sorted({a for a, _ in rows} | {b for _, b in rows})
    idx = {s: i for i, s in enumerate(names)}
    e = set()
    for a, b in rows:
        u, v = idx[a], idx[b]
        if u != v:
            e.add((min(u, v), max(u, v)))
    return len(names), np.array(sorted(e), dtype=np.int32)


def main(root):
    root = Path(root)
    OUT.mkdir(parents=True, exist_ok=True)
    for name, fn, p in [
        ("usair", seal_txt, root / "link_prediction/usair/edges.txt"),
        ("cora", planetoid, root / "cora/raw/ind.cora.graph"),
        ("pubmed", planetoid, root / "pubmed/raw/ind.pubmed.graph"),
    ]:
        n, e = fn(p)
        np.savez_compressed(OUT / f"topo_{name}.npz", num_nodes=np.int64(n), edges=e)
        deg = np.bincount(e.ravel(), minlength=n)
        print(f"{name}: N={n} E={len(e)} mean_deg={deg.mean():.2f} max_deg={deg.max()}")


if __name__ == "__main__":
    main(sys.argv[1] if len(sys.argv) > 1 else "/root/reference/data")
