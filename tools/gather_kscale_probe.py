#!/usr/bin/env python3
"""Config 5's gather launch against sign_k (coefficient scalars and multiply-adds per row scale with it; ids, rows
of X and the per-job chain do not).  GPU box: python3 tools/gather_kscale_probe.py"""
import sys, time
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import torch
from s3grl_amd import workloads
from s3grl_amd.engine import Engine

w = workloads.make("collab_pos_k3")
li, y = w.split.all_links()
e = Engine("cuda:0")
G = e.graph(w.A); L = e.links(li); f = e.features(w.X, mode="dense")
for K in (2, 3, 4, 6, 8):
    p = e.plan(G, L, mode="pos", num_hops=1, sign_k=K)
    out = p.run(f); torch.cuda.synchronize()
    ts = []
    for _ in range(6):
        t0 = time.perf_counter(); p.run(f); torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
    print("sign_k", K, "gather ms", round(1e3 * min(ts), 3))
    p.close(); del out
