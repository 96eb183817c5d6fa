import sys, time
sys.path.insert(0, "/root/repo")
import numpy as np, torch
from s3grl_amd import workloads
from s3grl_amd.engine import Engine
w = workloads.make("collab_pos_k3")
li, y = w.split.all_links()
e = Engine("cuda:0")
G = e.graph(w.A); L = e.links(li)
p = e.plan(G, L, mode=w.mode, num_hops=w.num_hops, sign_k=w.sign_k)
for F in (8, 32, 64, 96, 128):
    f = e.features(np.ascontiguousarray(w.X[:, :F]), mode="dense")
    out = p.run(f); torch.cuda.synchronize()
    ts = []
    for _ in range(8):
        t0 = time.perf_counter(); p.run(f); torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
    print("F", F, "gather ms", round(1e3 * min(ts), 3))
    f.close(); del out
