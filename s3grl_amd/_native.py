"""ctypes binding of libs3grl_hip.so (the C ABI declared in include/s3grl.h).

There is NO fallback: if the shared library is missing or a call fails, this module raises.
Build it with `python -c "import __graft_entry__ as g; g.build()"` (hipcc, gfx950).
"""
from __future__ import annotations

import ctypes as C
import os
from pathlib import Path

LIB_PATH = Path(__file__).resolve().parent / "lib" / "libs3grl_hip.so"

# status codes (include/s3grl.h)
OK = 0
ERR_INVALID_ARGUMENT = 1
ERR_NOT_IMPLEMENTED = 2
ERR_NO_FEATURES = 3
ERR_OUT_OF_MEMORY = 4
ERR_HIP = 5
ERR_NO_DEVICE = 6
ERR_GRAPH_TOO_LARGE = 7
ERR_SELF_LINK = 8

MODE_POS, MODE_POS_PLUS, MODE_SOP, MODE_SOP_RESTRICTED = 0, 1, 2, 3
STRATEGY = {"intersection": 0, "union": 1}

# every symbol include/s3grl.h declares; tests check the library exports all of them
SYMBOLS = [
    "s3grl_abi_version", "s3grl_status_string", "s3grl_last_error",
    "s3grl_context_create", "s3grl_context_preload", "s3grl_context_destroy", "s3grl_context_timings",
    "s3grl_context_set_profiling", "s3grl_context_trim", "s3grl_plan_gather_traffic",
    "s3grl_graph_create", "s3grl_graph_create_directed", "s3grl_graph_destroy",
    "s3grl_plan_create", "s3grl_plan_create_sets", "s3grl_walk_sets", "s3grl_plan_destroy", "s3grl_plan_get_stats", "s3grl_plan_total_rows", "s3grl_plan_counts", "s3grl_plan_row_ptr",
    "s3grl_plan_row_nodes", "s3grl_plan_export_subgraphs", "s3grl_plan_link_cost", "s3grl_run",
    "s3grl_sop_create", "s3grl_sop_create_weighted", "s3grl_sop_destroy", "s3grl_sop_run", "s3grl_sop_features",
    "s3grl_features_create", "s3grl_features_destroy", "s3grl_features_info", "s3grl_run_features",
    "s3grl_centre_pool_forward", "s3grl_centre_pool_backward", "s3grl_calibration_read",
]


class Cfg(C.Structure):
    _fields_ = [("mode", C.c_int32), ("num_hops", C.c_int32), ("sign_k", C.c_int32),
                ("strategy", C.c_int32), ("directed", C.c_int32), ("flags", C.c_uint32),
                ("rw_m", C.c_int32), ("rw_M", C.c_int32), ("seed", C.c_uint32),
                ("max_nodes_per_hop", C.c_int32), ("ratio_per_hop", C.c_double),
                ("reserved", C.c_int32 * 4)]


class NodeSets(C.Structure):
    _fields_ = [("set_ptr", C.c_void_p), ("set_nodes", C.c_void_p), ("num_sets", C.c_int64),
                ("num_set_nodes", C.c_int64), ("per_link", C.c_int32), ("reserved", C.c_int32)]


ABI_VERSION = 6
FLAG_FULL_STATS, FLAG_NO_FOLD, FLAG_COUNT_ONLY = 1, 2, 4


class PlanStats(C.Structure):
    _fields_ = [(n, C.c_int64) for n in (
        "num_links", "total_rows", "total_nodes", "total_volume", "total_sub_edges",
        "total_support", "num_row_pairs", "max_nodes", "workspace_bytes", "folded_links",
        "extracted_nodes", "oriented_entries", "hub_links", "hub_read_bytes", "hub_endpoint_entries", "hub_nodes")]

    def as_dict(self):
        return {n: int(getattr(self, n)) for n, _ in self._fields_}


class S3GRLError(RuntimeError):
    def __init__(self, status, what, detail):
        self.status = status
        super().__init__(f"{what}: {detail}" if detail else what)


_lib = None


def lib():
    """The loaded library; raises (loudly) when it has not been built."""
    global _lib
    if _lib is not None:
        return _lib
    # torch ships its own libamdhip64.so.7; import it FIRST so that the dynamic loader binds
    # this library to the same HIP runtime (one runtime per process: device pointers and
    # streams are shared with torch).
    import torch  # noqa: F401

    path = Path(os.environ.get("S3GRL_LIB", LIB_PATH))
    if not path.exists():
        raise ImportError(
            f"{path} not found: the HIP engine is not built. Run __graft_entry__.build() "
            "(hipcc --offload-arch=gfx950). There is no CPU fallback.")
    L = C.CDLL(str(path))
    vp, i64, i32 = C.c_void_p, C.c_int64, C.c_int32
    L.s3grl_abi_version.restype = i32
    L.s3grl_status_string.restype = C.c_char_p
    L.s3grl_status_string.argtypes = [i32]
    L.s3grl_last_error.restype = C.c_char_p
    proto = {
        "s3grl_context_create": [i32, vp, C.POINTER(vp)],
        "s3grl_context_preload": [vp, C.c_uint32, C.POINTER(C.c_double)],
        "s3grl_context_destroy": [vp],
        "s3grl_context_timings": [vp, C.POINTER(C.c_double)],
        "s3grl_context_set_profiling": [vp, i32],
        "s3grl_context_trim": [vp, C.POINTER(i64)],
        "s3grl_plan_gather_traffic": [vp, vp, vp, C.POINTER(i64)],
        "s3grl_graph_create": [vp, i64, vp, vp, i64, C.POINTER(vp)],
        "s3grl_graph_create_directed": [vp, i64, vp, vp, vp, vp, i64, C.POINTER(vp)],
        "s3grl_graph_destroy": [vp],
        "s3grl_plan_create": [vp, vp, vp, i64, C.POINTER(Cfg), C.POINTER(vp)],
        "s3grl_plan_create_sets": [vp, vp, vp, i64, C.POINTER(Cfg), C.POINTER(NodeSets), C.POINTER(vp)],
        "s3grl_walk_sets": [vp, vp, vp, i64, i32, i32, C.c_uint32, vp, vp],
        "s3grl_plan_destroy": [vp],
        "s3grl_plan_get_stats": [vp, C.POINTER(PlanStats)],
        "s3grl_plan_total_rows": [vp, C.POINTER(i64)],
        "s3grl_plan_counts": [vp, C.POINTER(i64)],
        "s3grl_plan_row_ptr": [vp, vp],
        "s3grl_plan_row_nodes": [vp, vp],
        "s3grl_plan_export_subgraphs": [vp, vp, vp, vp],
        "s3grl_plan_link_cost": [vp, vp],
        "s3grl_run": [vp, vp, vp, i64, i64, vp],
        "s3grl_sop_create": [vp, vp, vp, i64, i64, i32, C.POINTER(vp)],
        "s3grl_sop_create_weighted": [vp, vp, vp, i64, i64, i32, vp, C.POINTER(vp)],
        "s3grl_sop_destroy": [vp],
        "s3grl_sop_run": [vp, vp, vp, i64, vp],
        "s3grl_sop_features": [vp, vp, vp],
        "s3grl_features_create": [vp, vp, i64, i64, i64, i32, C.POINTER(vp)],
        "s3grl_features_destroy": [vp],
        "s3grl_features_info": [vp, C.POINTER(i64), C.POINTER(i32)],
        "s3grl_run_features": [vp, vp, vp, vp],
        "s3grl_centre_pool_forward": [vp, vp, vp, i64, i64, i32, vp],
        "s3grl_centre_pool_backward": [vp, vp, vp, i64, i64, i32, vp, vp],
        "s3grl_calibration_read": [vp, vp, i64, i32, i64, i32, C.POINTER(i64)],
    }
    for name, args in proto.items():
        fn = getattr(L, name)
        fn.argtypes = args
        fn.restype = i32
    if L.s3grl_abi_version() != ABI_VERSION:
        raise ImportError("libs3grl_hip.so ABI version mismatch")
    _lib = L
    return L


_EXC = {
    ERR_NOT_IMPLEMENTED: NotImplementedError,   # reference tuned_SIGN.py:235,251; utils.py:553
    ERR_NO_FEATURES: AssertionError,            # reference tuned_SIGN.py:166,221
    ERR_INVALID_ARGUMENT: ValueError,
    ERR_SELF_LINK: ValueError,
    ERR_OUT_OF_MEMORY: MemoryError,
}


def check(status, what):
    """Map a status code onto the Python exception the reference would raise."""
    if status == OK:
        return
    L = lib()
    detail = L.s3grl_last_error().decode()
    name = L.s3grl_status_string(status).decode()
    exc = _EXC.get(status)
    msg = f"{what}: {name}" + (f" ({detail})" if detail else "")
    if exc is None:
        raise S3GRLError(status, f"{what}: {name}", detail)
    raise exc(msg)
