import os
import sys
from pathlib import Path

import numpy as np
import pytest

REPO = Path(__file__).resolve().parent.parent
if str(REPO) not in sys.path:
    sys.path.insert(0, str(REPO))
GOLDEN = REPO / "tests" / "golden"


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def csr_from_undirected(n, edges):
    """Both directions, int64 ones — what reference sgrl_link_pred.py:107-114 builds."""
    import scipy.sparse as ssp

    e = np.asarray(edges, dtype=np.int64).reshape(-1, 2)
    r = np.concatenate([e[:, 0], e[:, 1]])
    c = np.concatenate([e[:, 1], e[:, 0]])
    return ssp.csr_matrix((np.ones(len(r), dtype=np.int64), (r, c)), shape=(n, n))


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN


def csr_from_arcs(n, arcs):
    """A directed graph as the reference holds it: A[u, v] = 1 for every arc u -> v
    (sgrl_link_pred.py:107-114 on a directed edge_index)."""
    import scipy.sparse as ssp

    a = np.asarray(arcs, dtype=np.int64).reshape(-1, 2)
    return ssp.csr_matrix((np.ones(len(a), dtype=np.int64), (a[:, 0], a[:, 1])), shape=(n, n))


def load_extract_directed(name):
    return np.load(GOLDEN / f"extract_directed_{name}.npz")


def load_extract(name):
    return np.load(GOLDEN / f"extract_{name}.npz")


def load_diffusion(name):
    return np.load(GOLDEN / f"diffusion_{name}.npz")


def load_sampled(name):
    return np.load(GOLDEN / f"sampled_{name}.npz")


SAMPLED_NAMES = ["rand300", "usair", "cora"]
DIRECTED_NAMES = ["tiny", "rand300", "usair", "cora"]
EXTRACT_NAMES = ["probe5", "triangle", "pair", "star_iso", "rand300", "usair", "cora"]
DIFFUSION_NAMES = ["probe5", "star_iso", "rand300", "usair"]
