#!/usr/bin/env python3
"""Where the drop-in call spends its time (GPU box): engine, D2H variants, list construction."""
import sys, time
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import numpy as np
import torch
from s3grl_amd import workloads, tuned_SIGN as ts
from s3grl_amd.engine import default_engine

w = workloads.make("pubmed_pos_k3")
li, y = w.split.all_links()
eng = default_engine()
g, x = eng.graph(w.A), eng.features(w.X)
links = eng.links(li)
def t(f, n=3):
    best = 1e9
    for _ in range(n):
        torch.cuda.synchronize(); t0 = time.perf_counter(); r = f(); torch.cuda.synchronize()
        best = min(best, time.perf_counter() - t0)
    return best, r
dt, res = t(lambda: eng.precompute(g, x, links, mode="pos", num_hops=3, sign_k=3))
print("engine precompute %.1f ms" % (dt * 1e3))
rows = res.rows
nb = rows.numel() * 4
pin = torch.empty(rows.shape, dtype=torch.float32, pin_memory=True)
dt, _ = t(lambda: pin.copy_(rows, non_blocking=True))
print("copy into torch pinned tensor      %.1f ms  %.1f GB/s" % (dt * 1e3, nb / dt / 1e9))
alias = torch.from_numpy(pin.numpy())
dt, _ = t(lambda: alias.copy_(rows, non_blocking=True))
print("copy into numpy alias of it        %.1f ms  %.1f GB/s  is_pinned=%s" % (dt * 1e3, nb / dt / 1e9, alias.is_pinned()))
dt, _ = t(lambda: alias.copy_(rows, non_blocking=False))
print("blocking copy into the alias       %.1f ms  %.1f GB/s" % (dt * 1e3, nb / dt / 1e9))
dt, _ = t(lambda: rows.cpu(), 2)
print("pageable .cpu()                    %.1f ms  %.1f GB/s" % (dt * 1e3, nb / dt / 1e9))
dt, _ = t(lambda: ts._to_host(rows))
print("_to_host (pooled)                  %.1f ms" % (dt * 1e3))
kw = {"sign_k": 3, "k_node_set_strategy": "intersection"}
xt = torch.from_numpy(w.X)
lt = torch.from_numpy(li)
dt, lst = t(lambda: ts.OptimizedSignOperations.get_PoS_prepped_ds(lt, 3, w.A, 1.0, None, False, None, xt, 1, kw, None))
print("get_PoS_prepped_ds whole list      %.1f ms  -> %.2f M pairs/s" % (dt * 1e3, li.shape[1] / dt / 1e6))
import cProfile, pstats
pr = cProfile.Profile(); pr.enable()
ts.OptimizedSignOperations.get_PoS_prepped_ds(lt, 3, w.A, 1.0, None, False, None, xt, 1, kw, None)
pr.disable(); pstats.Stats(pr).sort_stats("cumulative").print_stats(14)
