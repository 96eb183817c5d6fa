#!/usr/bin/env python3
"""Golden vector for the SEAL txt reader (reference data_utils.py:76-93).  BUILD CONTAINER only.

Runs the reference's own `read_label` / `read_edges` on its USAir edge list and stores what they
return.  `data_utils.py` imports torch_geometric at module scope (absent here): the import is made
possible with inert placeholder modules that compute nothing; the two functions are plain Python.
The edge list itself (a public dataset file, data not source) is committed next to the vector as
tests/golden/usair_edges.txt so that the reader can be exercised where /root/reference is absent.
"""
import os
import shutil
import sys
import types
from pathlib import Path

os.environ["PYTHONDONTWRITEBYTECODE"] = "1"
sys.dont_write_bytecode = True

import numpy as np

HERE = Path(__file__).resolve().parent
REFERENCE = Path("/root/reference")


def _inert(name, **attrs):
    m = types.ModuleType(name)
    m.__dict__.update(attrs)
    sys.modules[name] = m
    return m


def main():
    def _absent(*a, **k):
        raise RuntimeError("placeholder for a package that is not installed")

    class _Nothing:
        pass

    tg = _inert("torch_geometric")
    tg.utils = _inert("torch_geometric.utils", to_undirected=_absent, from_scipy_sparse_matrix=_absent,
                      is_undirected=_absent)
    tg.data = _inert("torch_geometric.data", Data=_Nothing)
    sys.path.insert(0, str(REFERENCE))
    import data_utils as ref  # the reference module

    src = REFERENCE / "data" / "link_prediction" / "usair"
    mapping = ref.read_label(str(src))
    edges = ref.read_edges(str(src), mapping)
    names = sorted(mapping, key=mapping.get)
    np.savez_compressed(HERE / "seal_usair.npz",
                        names=np.array(names), ids=np.array([mapping[n] for n in names], dtype=np.int64),
                        edges=np.array(edges, dtype=np.int64))
    shutil.copyfile(src / "edges.txt", HERE / "usair_edges.txt")
    print(f"seal_usair.npz: {len(names)} nodes, {len(edges)} edges")


if __name__ == "__main__":
    main()
