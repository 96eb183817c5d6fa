#!/usr/bin/env python3
"""How well do the shards of an N-GPU job balance, and what does sharding cost in total work?
One GPU: every shard of an N-way split is planned + run on its own (what each rank of an N-GPU job
would spend computing; 3 repeats, best), pair-aware shards (parallel.ShardPlan) next to contiguous
ranges, against the unsharded step.

    python3 tools/shard_balance_probe.py [workload]"""
import sys
import time
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import numpy as np
import torch

from s3grl_amd import parallel, workloads
from s3grl_amd.engine import Engine

wl = sys.argv[1] if len(sys.argv) > 1 else "pubmed_pos_k3"
w = workloads.make(wl)
li, y = w.split.all_links()
eng = Engine("cuda:0")
g, x = eng.graph(w.A), eng.features(w.X)
K, F = w.sign_k, w.X.shape[1]
L = li.shape[1]


def run(links):
    lk = eng.links(links)
    out = torch.empty((2 * links.shape[1], K + 1, F + 1), device=eng.device)
    best, folded = 1e9, 0
    for _ in range(3):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        p = eng.plan(g, lk, mode=w.mode, num_hops=w.num_hops, sign_k=K)
        p.run(x, out)
        folded = p.stats["folded_links"]
        p.close()
        torch.cuda.synchronize()
        best = min(best, time.perf_counter() - t0)
    return best * 1e3, folded


sizes = eng.subgraph_sizes(g, eng.links(li), num_hops=w.num_hops).cpu().numpy().astype(np.float64)
whole, folded_whole = run(li)
print(f"{wl}: unsharded step {whole:.2f} ms, {folded_whole} of {L} links folded")
cost_fold = parallel.measured_cost(eng, g, li, w.num_hops, mode=w.mode)
cost_full = eng.link_costs(g, eng.links(li), num_hops=w.num_hops, mode=w.mode, fold_reversed=False).cpu().numpy()
for world in (2, 4, 8):
    for name, pair_aware, cost in (("pair-aware", True, cost_fold), ("contiguous", False, cost_full)):
        sp = parallel.ShardPlan(li, world, cost, pair_aware=pair_aware)
        links = sp.links.cpu().numpy()
        ts, fs = [], []
        order = sp.order.numpy()
        for r in range(world):
            t, f = run(links[:, sp.bounds[r]:sp.bounds[r + 1]])
            ts.append(t)
            fs.append(f)
            mine = order[sp.bounds[r]:sp.bounds[r + 1]]
            is_f = cost_fold[mine] == 250.0 if pair_aware else np.zeros(len(mine), bool)
            # one line per shard for refitting the cost model: ms, Σn / Σn² of the extracted links, counts
            print(f"  fit {wl} {name} {world} {r}: ms {t:.3f} sum_n {sizes[mine][~is_f].sum():.0f} "
                  f"sum_n2 {(sizes[mine][~is_f] ** 2).sum():.0f} extracted {int((~is_f).sum())} folded_model {int(is_f.sum())} folded_real {f}")
        print(f"{wl} world {world} {name:10s}: shard ms {[round(t, 2) for t in ts]}  max {max(ts):.2f}  "
              f"max/mean {max(ts) / (sum(ts) / len(ts)):.3f}  sum/unsharded {sum(ts) / whole:.2f}  "
              f"speed-up of compute {whole / max(ts):.2f}x  folded {sum(fs)}")
