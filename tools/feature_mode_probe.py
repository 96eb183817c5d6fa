import sys, time
sys.path.insert(0, "/root/repo")
import torch
from s3grl_amd import workloads
from s3grl_amd.engine import Engine
name = sys.argv[1] if len(sys.argv) > 1 else "cora_posplus_k3_real"
w = workloads.make(name)
li, y = w.split.all_links()
e = Engine("cuda:0")
G = e.graph(w.A); L = e.links(li)
for mode in ("dense", "packed", "sparse"):
    try:
        f = e.features(w.X, mode=mode)
    except Exception as ex:
        print(mode, "ERR", ex); continue
    p = e.plan(G, L, mode=w.mode, num_hops=w.num_hops, sign_k=w.sign_k)
    out = p.run(f); torch.cuda.synchronize()
    ts = []
    for _ in range(10):
        t0 = time.perf_counter(); p.run(f, out=out) if "out" in p.run.__code__.co_varnames else p.run(f); torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
    print(name, mode, "gather-only run ms:", round(1e3 * min(ts), 3))
    p.close(); f.close()
