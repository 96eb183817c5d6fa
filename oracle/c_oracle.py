"""TEST INFRASTRUCTURE ONLY — ctypes binding of oracle/s3grl_oracle_c.c (the plain-C, OpenMP
restatement).  Same rules as the rest of `oracle/`: importable from tests/, smoke() and bench.py's
cpu_baseline leg only.  `build()` runs oracle/Makefile (gcc) into oracle/_build/ (git-ignored)."""
from __future__ import annotations

import ctypes as C
import os
import subprocess
from pathlib import Path

import numpy as np

HERE = Path(__file__).resolve().parent
SRC = HERE / "s3grl_oracle_c.c"
LIB = HERE / "_build" / "libs3grl_oracle_c.so"
LIB_F32 = HERE / "_build" / "libs3grl_oracle_c_f32.so"

_lib = None
_lib_f32 = None


def build(force=False):
    """`make -C oracle` (the committed recipe, oracle/Makefile)."""
    if force and LIB.exists():
        LIB.unlink()
    subprocess.run(["make", "-s", "-C", str(HERE), "all"], check=True)
    return LIB


def lib():
    global _lib
    if _lib is None:
        # S3GRL_ORACLE_C_LIB: another build of the same source (the sanitized one of `make asan`,
        # tests/test_oracle_sanitized.py)
        path = os.environ.get("S3GRL_ORACLE_C_LIB")
        if not path:
            build()
            path = str(LIB)
        _lib = C.CDLL(path)
        _lib.s3grl_oracle_c_extract.restype = C.c_int64
    return _lib


def lib_f32():
    """The same source built with float arithmetic (`make f32`): the reference's own precision."""
    global _lib_f32
    if _lib_f32 is None:
        build()
        _lib_f32 = C.CDLL(str(LIB_F32))
    return _lib_f32


def _p(a):
    return a.ctypes.data_as(C.c_void_p)


def _csr(A):
    indptr = np.ascontiguousarray(A.indptr, dtype=np.int64)
    indices = np.ascontiguousarray(A.indices, dtype=np.int32)
    return indptr, indices


def pos_rows(link_index, num_hops, A, X, sign_k, *, plus=False, threads=0, f32=False):
    """(rows fp64 [sum R, K+1, 1+F], row_ptr int64 [L+1], row_nodes int64 [sum R],
    node_count int32 [L]) for `link_index` [2, L] — the collated output of the reference's
    get_PoS_prepped_ds / get_PoS_Plus_prepped_ds ('intersection').  `f32`: the diffusion half in
    float arithmetic (the reference's precision) instead of double; the result is still returned as
    fp64 values."""
    compute = lib_f32() if f32 else lib()
    links = np.ascontiguousarray(np.asarray(link_index, dtype=np.int64).T)
    L = links.shape[0]
    indptr, indices = _csr(A)
    X = np.ascontiguousarray(X, dtype=np.float32)
    N, F = X.shape
    R = np.zeros(L, dtype=np.int64)
    rc = lib().s3grl_oracle_c_rows_per_link(C.c_int64(N), _p(indptr), _p(indices), _p(links),
                                            C.c_int64(L), int(num_hops), int(plus), _p(R))
    if rc:
        raise ValueError("self link" if rc == -2 else f"rows_per_link failed ({rc})")
    row_ptr = np.zeros(L + 1, dtype=np.int64)
    np.cumsum(R, out=row_ptr[1:])
    rows = np.empty((int(row_ptr[-1]), sign_k + 1, 1 + F), dtype=np.float64)
    row_nodes = np.empty(int(row_ptr[-1]), dtype=np.int64)
    node_count = np.empty(L, dtype=np.int32)
    rc = compute.s3grl_oracle_c_pos(C.c_int64(N), _p(indptr), _p(indices), _p(X), C.c_int64(F),
                                  C.c_int64(F), _p(links), C.c_int64(L), int(num_hops), int(sign_k),
                                  int(plus), int(threads), _p(row_ptr), _p(rows), _p(row_nodes),
                                  _p(node_count))
    if rc:
        raise RuntimeError(f"s3grl_oracle_c_pos failed ({rc})")
    return rows, row_ptr, row_nodes, node_count


def extract(link_index, num_hops, A):
    """(node_ptr [L+1], nodes int32 hop-major / ascending id per hop, dists int8)."""
    links = np.ascontiguousarray(np.asarray(link_index, dtype=np.int64).T)
    L = links.shape[0]
    indptr, indices = _csr(A)
    N = len(indptr) - 1
    cap = max(int(min(L * N, 1 << 28)), 2)
    node_ptr = np.zeros(L + 1, dtype=np.int64)
    nodes = np.empty(cap, dtype=np.int32)
    dists = np.empty(cap, dtype=np.int8)
    tot = lib().s3grl_oracle_c_extract(C.c_int64(N), _p(indptr), _p(indices), _p(links), C.c_int64(L),
                                       int(num_hops), C.c_int64(cap), _p(node_ptr), _p(nodes), _p(dists))
    if tot < 0:
        raise RuntimeError(f"s3grl_oracle_c_extract failed ({tot})")
    return node_ptr, nodes[:tot].copy(), dists[:tot].copy()


def cpu_threads():
    """Usable host threads: the affinity mask, capped by the cgroup CPU quota when there is one
    (a GPU box exposes all of the host's cpus but grants a share of them)."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    for path in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us"):
        try:
            txt = open(path).read().split()
            if path.endswith("cpu.max"):
                if txt[0] != "max":
                    n = min(n, max(1, int(int(txt[0]) / int(txt[1]) + 0.5)))
            else:
                quota = int(txt[0])
                period = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
                if quota > 0:
                    n = min(n, max(1, int(quota / period + 0.5)))
            break
        except Exception:
            continue
    return n
