#!/usr/bin/env python3
"""Does the headline's packed gather gain from pinning graph-local jobs to one XCD (its own 4 MB L2)?
Workgroup b of a launch runs on XCD b mod 8.  The plan's job order (longest first) is replaced through the
measurement hook s3grl_debug_set_job_order by orders that give XCD k the links whose endpoints lie in part
k of a locality order of the graph (reverse Cuthill-McKee, 8 equal chunks; longest first inside a part),
and — control — by the same interleaving with links assigned to parts at random.

    python3 tools/xcd_locality_probe.py [workload]"""
import ctypes as C
import sys
import time
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import numpy as np
import torch
from scipy.sparse.csgraph import reverse_cuthill_mckee

from s3grl_amd import _native as N
from s3grl_amd import workloads
from s3grl_amd.engine import Engine

wl = sys.argv[1] if len(sys.argv) > 1 else "pubmed_pos_k3"
w = workloads.make(wl)
li, y = w.split.all_links()
li = np.asarray(li)
eng = Engine("cuda:0")
g, x = eng.graph(w.A), eng.features(w.X)
K, F = w.sign_k, w.X.shape[1]
L = li.shape[1]
lk = eng.links(li)
sizes = eng.subgraph_sizes(g, lk, num_hops=w.num_hops).cpu().numpy()
out = torch.empty((2 * L, K + 1, F + 1), device=eng.device)
lib = N.lib()
lib.s3grl_debug_set_job_order.restype = C.c_int32
lib.s3grl_debug_set_job_order.argtypes = [C.c_void_p, C.c_void_p, C.c_int64]


def interleave(part):
    """order[b] for block b -> a job of part b % 8 (longest first inside a part); leftovers at the end."""
    lists = [np.flatnonzero(part == k) for k in range(8)]
    lists = [l[np.argsort(-sizes[l], kind="stable")] for l in lists]
    m = min(len(l) for l in lists)
    head = np.stack([l[:m] for l in lists], 1).reshape(-1)        # b = 8 * i + k
    rest = np.concatenate([l[m:] for l in lists])
    rest = rest[np.argsort(-sizes[rest], kind="stable")]
    return np.concatenate([head, rest]).astype(np.int32)


def gather_ms(order):
    p = eng.plan(g, lk, mode=w.mode, num_hops=w.num_hops, sign_k=K, fold_reversed=False)
    assert p.stats["num_row_pairs"] == L
    if order is not None:
        o = torch.from_numpy(order).to(eng.device)
        N.check(lib.s3grl_debug_set_job_order(p._h, C.c_void_p(o.data_ptr()), L), "set_job_order")
    best = 1e9
    for _ in range(5):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        p.run(x, out)
        torch.cuda.synchronize()
        best = min(best, time.perf_counter() - t0)
    ref = out[:2000].clone()
    p.close()
    return best * 1e3, ref


base, ref = gather_ms(None)
print(f"{wl}: {L} jobs (no folding); longest first (the plan's order): {base:.3f} ms")
rng = np.random.default_rng(0)
t, r2 = gather_ms(interleave(rng.integers(0, 8, L)))
print(f"random parts, interleaved over the XCDs:           {t:.3f} ms   (same rows: {torch.equal(ref, r2)})")
perm = reverse_cuthill_mckee(w.A.tocsr(), symmetric_mode=True)
pos = np.empty(len(perm), dtype=np.int64)
pos[perm] = np.arange(len(perm))
n = w.A.shape[0]
for name, key in (("RCM position of src", pos[li[0]]), ("mean RCM position of src, dst", (pos[li[0]] + pos[li[1]]) // 2),
                  ("RCM position of the higher-degree endpoint",
                   np.where(np.diff(w.A.indptr)[li[0]] >= np.diff(w.A.indptr)[li[1]], pos[li[0]], pos[li[1]]))):
    part = np.minimum(key * 8 // n, 7)
    t, r2 = gather_ms(interleave(part))
    print(f"parts by {name:45s}: {t:.3f} ms   part sizes {np.bincount(part, minlength=8).tolist()}  (same rows: {torch.equal(ref, r2)})")
