#!/usr/bin/env python3
"""Extract the public Planetoid node FEATURES of Cora into a compact .npz data file, so that the downstream
check (AUC on the paper's own configuration) runs on the real bag-of-words operand where
/root/reference does not exist (the GPU box).

Data only: no reference source text is read or copied.  Inputs are the public Planetoid pickles
ind.cora.{allx,tx,test.index} (scipy sparse matrices; rows of `tx` belong to the node ids listed in
test.index, PyG's reader puts them back in id order).  The matrix is stored as the CSR structure of its
non-zeros (all values are 1.0 for Cora) — row normalisation is the producer's job at run time
(`workloads.normalize_features`, reference sgrl_link_pred.py:851,1000-1003).  Run in the build container only:

    python tools/make_features.py /root/reference/data
"""
import pickle
import sys
from pathlib import Path

import numpy as np
import scipy.sparse as ssp

OUT = Path(__file__).resolve().parent.parent / "s3grl_amd" / "data"


def _load(path):
    with open(path, "rb") as f:
        return pickle.load(f, encoding="latin1")


def planetoid_x(raw, name):
    allx, tx = ssp.csr_matrix(_load(raw / f"ind.{name}.allx")), ssp.csr_matrix(_load(raw / f"ind.{name}.tx"))
    test_index = np.array([int(ln) for ln in open(raw / f"ind.{name}.test.index")], dtype=np.int64)
    x = ssp.vstack([allx, tx]).tolil()
    # the rows of tx are the nodes test_index (in that order): back into id order
    x[test_index, :] = x[np.sort(test_index), :]
    return ssp.csr_matrix(x)


def main(root):
    raw = Path(root) / "cora" / "raw"
    x = planetoid_x(raw, "cora")
    x.sum_duplicates()
    x.eliminate_zeros()
    x.sort_indices()
    assert x.shape == (2708, 1433), x.shape
    binary = bool(np.all(x.data == 1.0))
    OUT.mkdir(parents=True, exist_ok=True)
    np.savez_compressed(OUT / "feat_cora.npz", shape=np.array(x.shape, dtype=np.int64),
                        indptr=x.indptr.astype(np.int32), indices=x.indices.astype(np.int16),
                        data=np.zeros(0, np.float32) if binary else x.data.astype(np.float32))
    print(f"cora: x {x.shape}, nnz {x.nnz} ({x.nnz / x.shape[0]:.1f} per row), binary {binary}, "
          f"empty rows {(np.diff(x.indptr) == 0).sum()}")


if __name__ == "__main__":
    main(sys.argv[1] if len(sys.argv) > 1 else "/root/reference/data")
