import sys, time
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import numpy as np, torch
from s3grl_amd import workloads, parallel
from s3grl_amd.engine import Engine
wl = sys.argv[1] if len(sys.argv) > 1 else "pubmed_pos_k3"
w = workloads.make(wl); li, y = w.split.all_links()
eng = Engine("cuda:0"); g, x = eng.graph(w.A), eng.features(w.X)
K, F = w.sign_k, w.X.shape[1]
cost = parallel.measured_cost(eng, g, li, w.num_hops)
world = 8
b = parallel.shard_bounds(li.shape[1], world, cost)
for r in range(world):
    lk = eng.links(li[:, b[r]:b[r + 1]])
    out = torch.empty((2 * (b[r + 1] - b[r]), K + 1, F + 1), device=eng.device)
    for _ in range(2):
        p = eng.plan(g, lk, mode=w.mode, num_hops=w.num_hops, sign_k=K); p.run(x, out); p.close()
    torch.cuda.synchronize(); eng.set_profiling(True)
    t0 = time.perf_counter()
    for _ in range(5):
        p = eng.plan(g, lk, mode=w.mode, num_hops=w.num_hops, sign_k=K); p.run(x, out); st = p.stats; p.close()
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 5 * 1e3
    tm = eng.timings(); eng.set_profiling(False)
    print(f"shard {r}: links {b[r+1]-b[r]} wall {dt:.2f} ms  structure {tm['structure_ms']/5:.2f} propagate {tm['propagate_ms']/5:.2f} gather {tm['gather_ms']/5:.2f}  max_n {st['max_nodes']}")
