"""On-disk bundle of one split's precomputed operators.

The reference caches each split as `torch.save(self.collate(pos_list + neg_list), processed_paths[0])`
under `<root>_seal{data_appendix}/processed/SEAL_{split}_data[_{percent}].pt`
(sgrl_link_pred.py:85-92, :204, :797-806): a pickled PyG `(data, slices)` pair of L `Data` objects.
Its `data_appendix` names only the extraction settings (hops, labelling, ratio, seed, ScaLed m/M),
so a cache written with one `sign_k` / flow is silently reused by a run with another.

The engine's output is already collated — `rows` fp32 [sum R, K+1, 1+F], `row_ptr` int64 [L+1],
`y` [L] — so the bundle is those three arrays as raw, 4 KiB-aligned blobs behind a JSON header:
`load()` memory-maps them (no unpickling, no per-link objects) and one `.to(device)` puts a whole
split in HBM.  The key is the reference's `data_appendix` string plus the operator settings.

    name = bundle_name("train", 100)                        # SEAL_train_data.s3grl
    root = cache_dir("dataset/Cora", data_appendix(num_hops=3, node_label="zo", ratio_per_hop=1.0,
                                                   seed=1), mode="pos_plus", sign_k=3)
    rows, row_ptr, y, meta = get_or_compute(root / name, compute_fn, expect={"num_links": L})
"""
from __future__ import annotations

import json
import os
import struct
from pathlib import Path

import numpy as np

MAGIC = b"S3GRLB1\0"
ALIGN = 4096
_DTYPES = {"float32": np.float32, "int64": np.int64, "int32": np.int32, "int8": np.int8,
           "float64": np.float64}


def data_appendix(*, num_hops, node_label, ratio_per_hop, seed, max_nodes_per_hop=None, m=0, M=0,
                  dropedge=0.0, use_valedges_as_input=False):
    """The reference's cache-directory suffix, character for character (sgrl_link_pred.py:797-806)."""
    if m and M:
        s = f"_m{m}_M{M}_dropedge{dropedge}_seed{seed}"
    else:
        s = "_h{}_{}_rph{}_seed{}".format(num_hops, node_label, "".join(str(ratio_per_hop).split(".")), seed)
        if max_nodes_per_hop is not None:
            s += "_mnph{}".format(max_nodes_per_hop)
    if use_valedges_as_input:
        s += "_uvai"
    return s


def operator_tag(*, mode, sign_k, strategy="intersection"):
    """What the reference's key leaves out: which operators the rows hold."""
    if mode not in ("pos", "pos_plus", "sop", "hybrid"):
        raise ValueError(f"unknown mode {mode}")
    tag = f"_{mode}_k{int(sign_k)}"
    if mode == "pos_plus":
        tag += f"_{strategy}"
    return tag


def cache_dir(dataset_root, appendix, *, mode, sign_k, strategy="intersection"):
    """`<dataset_root>_seal<data_appendix><operator_tag>/processed`, next to where the reference
    keeps its own `processed/` directory (sgrl_link_pred.py:1098: `dataset.root + "_seal{}".format(args.data_appendix)`)."""
    return Path(str(dataset_root) + "_seal" + appendix + operator_tag(mode=mode, sign_k=sign_k,
                                                                      strategy=strategy)) / "processed"


def bundle_name(split, percent=100):
    """sgrl_link_pred.py:85-92 with the bundle's own extension."""
    name = f"SEAL_{split}_data" if int(percent) == 100 else f"SEAL_{split}_data_{percent}"
    return name + ".s3grl"


def _np(a):
    if hasattr(a, "detach"):
        a = a.detach().cpu().numpy()
    return np.ascontiguousarray(a)


def save(path, rows, row_ptr, y, meta=None):
    """Write atomically (temp file + rename): a killed run never leaves a half bundle behind."""
    path = Path(path)
    path.parent.mkdir(parents=True, exist_ok=True)
    arrays = {"rows": _np(rows), "row_ptr": _np(row_ptr).astype(np.int64, copy=False),
              "y": _np(y).astype(np.int64, copy=False)}
    if arrays["rows"].dtype != np.float32 or arrays["rows"].ndim != 3:
        raise ValueError("rows must be fp32 [sum R, K+1, 1+F]")
    L = arrays["row_ptr"].shape[0] - 1
    if arrays["y"].shape != (L,) or int(arrays["row_ptr"][-1]) != arrays["rows"].shape[0]:
        raise ValueError("row_ptr / y do not describe rows")
    header = {"version": 1, "meta": dict(meta or {}), "arrays": {}}
    header["meta"].setdefault("num_links", int(L))
    # the header's own length moves the first blob, and the offsets it then records can lengthen it
    # again: iterate until the header that is written is the one the offsets were computed from
    off, first = 0, -1
    for _ in range(8):
        blob = json.dumps(header, sort_keys=True).encode()
        start = -(-(len(MAGIC) + 8 + len(blob)) // ALIGN) * ALIGN
        if start == first:
            break
        first = off = start
        for k, a in arrays.items():
            header["arrays"][k] = {"dtype": str(a.dtype), "shape": list(a.shape), "offset": off,
                                   "nbytes": int(a.nbytes)}
            off = -(-(off + a.nbytes) // ALIGN) * ALIGN
    blob = json.dumps(header, sort_keys=True).encode()
    if len(MAGIC) + 8 + len(blob) > header["arrays"]["rows"]["offset"]:
        raise RuntimeError("bundle header overlaps the first blob")
    tmp = path.with_name(path.name + f".tmp{os.getpid()}")
    with open(tmp, "wb") as f:
        f.write(MAGIC)
        f.write(struct.pack("<Q", len(blob)))
        f.write(blob)
        for k, a in arrays.items():
            f.seek(header["arrays"][k]["offset"])
            a.tofile(f)
        f.truncate(off)
    os.replace(tmp, path)
    return path


def read_header(path):
    with open(path, "rb") as f:
        if f.read(len(MAGIC)) != MAGIC:
            raise ValueError(f"{path}: not an s3grl bundle")
        (n,) = struct.unpack("<Q", f.read(8))
        return json.loads(f.read(n))


def load(path, device=None):
    """-> (rows, row_ptr, y, meta).  numpy memmaps when `device` is None, torch tensors on
    `device` otherwise (one copy per array, straight from the page cache)."""
    header = read_header(path)
    out = {}
    for k, d in header["arrays"].items():
        shape = tuple(d["shape"])
        out[k] = np.memmap(path, dtype=_DTYPES[d["dtype"]], mode="r", offset=d["offset"], shape=shape) \
            if int(np.prod(shape)) else np.zeros(shape, dtype=_DTYPES[d["dtype"]])
    if device is not None:
        import torch

        dev = torch.device(device)
        # mmap pages are read-only: copy on the host only when the result stays on the host
        import warnings

        with warnings.catch_warnings():
            warnings.simplefilter("ignore", UserWarning)   # "array is not writable": it is only read
            out = {k: (torch.from_numpy(np.array(v)) if dev.type == "cpu"
                       else torch.from_numpy(np.asarray(v)).to(dev)) for k, v in out.items()}
    return out["rows"], out["row_ptr"], out["y"], header["meta"]


def get_or_compute(path, compute, *, expect=None, device=None):
    """Load the bundle at `path` if it exists and its meta matches `expect`; otherwise call
    `compute() -> (rows, row_ptr, y[, meta])`, save and return that."""
    path = Path(path)
    if path.exists():
        try:
            meta = read_header(path)["meta"]
            if all(meta.get(k) == v for k, v in (expect or {}).items()):
                return load(path, device)
        except (ValueError, KeyError, json.JSONDecodeError):
            pass                                   # unreadable or foreign file: recompute
    res = compute()
    rows, row_ptr, y = res[:3]
    meta = dict(res[3]) if len(res) > 3 else {}
    meta.update(expect or {})
    save(path, rows, row_ptr, y, meta)
    if device is None:
        return load(path, None)
    import torch

    return (torch.as_tensor(rows).to(device), torch.as_tensor(row_ptr).to(device),
            torch.as_tensor(y).to(device), meta)
