#!/usr/bin/env python3
"""How well do the shard weights balance? One GPU: every shard of an N-way split is timed on its
own (what each rank of an N-GPU job would spend computing), for several fixed per-link shares."""
import sys, time
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import numpy as np, torch
from s3grl_amd import workloads, parallel
from s3grl_amd.engine import Engine

wl = sys.argv[1] if len(sys.argv) > 1 else "pubmed_pos_k3"
w = workloads.make(wl)
li, y = w.split.all_links()
eng = Engine("cuda:0")
g, x = eng.graph(w.A), eng.features(w.X)
K, F = w.sign_k, w.X.shape[1]
n = eng.subgraph_sizes(g, eng.links(li), num_hops=w.num_hops).cpu().numpy().astype(np.float64)
model = parallel.measured_cost(eng, g, li, w.num_hops)
for world in (2, 4, 8):
    for per_link in (-1.0, 400.0):
        b = parallel.shard_bounds(li.shape[1], world, model if per_link < 0 else n + per_link)
        ts = []
        for r in range(world):
            lk = eng.links(li[:, b[r]:b[r + 1]])
            out = torch.empty((2 * (b[r + 1] - b[r]), K + 1, F + 1), device=eng.device)
            best = 1e9
            for _ in range(3):
                torch.cuda.synchronize(); t0 = time.perf_counter()
                p = eng.plan(g, lk, mode=w.mode, num_hops=w.num_hops, sign_k=K); p.run(x, out); p.close()
                torch.cuda.synchronize(); best = min(best, time.perf_counter() - t0)
            ts.append(best * 1e3)
        print(f"{wl} world {world} per_link {per_link:5.0f}: shard ms {[round(t, 2) for t in ts]}  max/mean {max(ts) / (sum(ts) / len(ts)):.3f}  max {max(ts):.2f}")
