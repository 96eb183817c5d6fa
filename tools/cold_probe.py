#!/usr/bin/env python3
"""Where does the fixed set-up time go?  (`graph_prepare_ms` is 15-21 ms whatever the size
of the graph.)  Times, in ONE fresh process and with wall clocks around synchronised calls: the import, the
context, the first and the second Graph() / Features() of the same inputs, the first and the second plan +
run.  The second call of each is the control: what the first one pays on top is one-off per process (code
objects loaded on first launch, first allocations, rocPRIM's first call), not work.

    python3 tools/cold_probe.py [--workload usair_pos_k2] [--json]
"""
import argparse
import json
import sys
import time
from pathlib import Path

REPO = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(REPO))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--workload", default="usair_pos_k2")
    ap.add_argument("--json", action="store_true")
    ap.add_argument("--preload", action="store_true", help="load all code objects right after the context (timed per unit)")
    a = ap.parse_args()
    t = {}
    c0 = time.perf_counter()
    import numpy as np
    import torch
    t["import_torch_ms"] = (time.perf_counter() - c0) * 1e3
    c0 = time.perf_counter()
    from s3grl_amd import workloads
    from s3grl_amd.engine import Engine, Features, Graph
    from s3grl_amd import _native
    _native.lib()
    t["import_engine_dlopen_ms"] = (time.perf_counter() - c0) * 1e3
    w = workloads.make(a.workload)
    li, _ = w.split.all_links()
    c0 = time.perf_counter()
    torch.cuda.init()
    torch.zeros(1, device="cuda:0")
    torch.cuda.synchronize()
    t["torch_cuda_first_touch_ms"] = (time.perf_counter() - c0) * 1e3

    def timed(name, fn):
        torch.cuda.synchronize()
        c = time.perf_counter()
        r = fn()
        torch.cuda.synchronize()
        t[name] = (time.perf_counter() - c) * 1e3
        return r

    eng = timed("context_create_ms", lambda: Engine("cuda:0", preload=0))
    if "--preload" in sys.argv:
        timed("preload_all_ms", lambda: eng.preload(7))
        t["preload_units_ms"] = {k: round(v, 2) for k, v in eng.preload_ms.items()}
    ip = torch.as_tensor(np.asarray(w.A.indptr, dtype=np.int64)).to(eng.device)
    ix = torch.as_tensor(np.asarray(w.A.indices, dtype=np.int32)).to(eng.device)
    xd = torch.as_tensor(w.X).to(device=eng.device, dtype=torch.float32).contiguous()
    links = eng.links(li)
    g1 = timed("graph_first_ms", lambda: Graph(eng, ip, ix, w.A.shape[0]))
    g2 = timed("graph_second_ms", lambda: Graph(eng, ip, ix, w.A.shape[0]))
    g2.close()
    g3 = timed("graph_third_ms", lambda: Graph(eng, ip, ix, w.A.shape[0]))
    g3.close()
    f1 = timed("features_first_ms", lambda: Features(eng, xd, "auto"))
    f2 = timed("features_second_ms", lambda: Features(eng, xd, "auto"))
    f2.close()
    K = w.sign_k

    def step():
        if w.mode == "sop":
            return eng.precompute(g1, f1, links, mode="sop", sign_k=K).rows
        p = eng.plan(g1, links, mode=w.mode, num_hops=w.num_hops, sign_k=K)
        r = p.run(f1)
        p.close()
        return r

    for i, nm in enumerate(("step_first_ms", "step_second_ms", "step_third_ms")):
        timed(nm, step)
    t["workload"] = a.workload
    t["num_nodes"] = int(w.A.shape[0])
    t["links"] = int(li.shape[1])
    if a.json:
        print(json.dumps(t))
    else:
        for k, v in t.items():
            print(f"{k:32s} {v if not isinstance(v, float) else round(v, 3)}")
    eng.close()


if __name__ == "__main__":
    main()
