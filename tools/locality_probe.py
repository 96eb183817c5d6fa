import sys, time
sys.path.insert(0, '.')
import numpy as np, torch
from s3grl_amd import workloads
from s3grl_amd.engine import Engine
w = workloads.make(sys.argv[1] if len(sys.argv) > 1 else "collab_pos_k3")
li, y = w.split.all_links()
eng = Engine("cuda:0")
g, x = eng.graph(w.A), eng.features(w.X)
K, F = w.sign_k, w.X.shape[1]
def run(links, tag):
    lk = eng.links(links)
    out = None
    eng.set_profiling(True)
    for _ in range(4):
        p = eng.plan(g, lk, mode=w.mode, num_hops=w.num_hops, sign_k=K); out = p.run(x, out); _ = p.stats; p.close()
    torch.cuda.synchronize()
    tm = eng.timings(); eng.set_profiling(False)
    print(tag, {k: round(v / 4, 3) for k, v in tm.items() if k.endswith("_ms") and v})
run(li, "list order      ")
o = np.lexsort((li[1], li[0]))
run(li[:, o], "sorted by src   ")
mn, mx = np.minimum(li[0], li[1]), np.maximum(li[0], li[1])
deg = np.diff(w.A.indptr)
hub = np.where(deg[li[0]] >= deg[li[1]], li[0], li[1])
o = np.lexsort((mn, hub))
run(li[:, o], "sorted by hub   ")
# XCD-contiguous: sorted list cut into 8 runs, block b takes run b % 8
L = li.shape[1]; seg = (L + 7) // 8
idx = np.arange(L); xo = (idx % 8) * seg + idx // 8; xo = xo[xo < L]
rest = np.setdiff1d(np.arange(L), xo, assume_unique=False)
perm = np.concatenate([xo, rest])
run(li[:, o][:, perm], "hub + xcd runs  ")
