#!/usr/bin/env python3
"""Can the link kernels (VALU-bound) and the gather (memory-bound) overlap?  Two contexts on two
streams: gather of plan A while plan B is being created (count + link kernels), against the two
run back to back."""
import sys, time
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import numpy as np, torch
from s3grl_amd import workloads
from s3grl_amd.engine import Engine

w = workloads.make("pubmed_pos_k3")
li, y = w.split.all_links()
K, F = w.sign_k, w.X.shape[1]
sA, sB = torch.cuda.Stream(), torch.cuda.Stream()
with torch.cuda.stream(sA):
    eA = Engine("cuda:0")
with torch.cuda.stream(sB):
    eB = Engine("cuda:0")
gA, xA, lA = eA.graph(w.A), eA.features(w.X), eA.links(li)
gB, xB, lB = eB.graph(w.A), eB.features(w.X), eB.links(li)
out = torch.empty((2 * li.shape[1], K + 1, F + 1), device="cuda:0")
out2 = torch.empty_like(out)
def sync(): torch.cuda.synchronize()
for rep in range(3):
    pA = eA.plan(gA, lA, mode="pos", num_hops=3, sign_k=K); sync()
    t0 = time.perf_counter(); pA.run(xA, out); sync(); t_g = time.perf_counter() - t0
    t0 = time.perf_counter(); pB = eB.plan(gB, lB, mode="pos", num_hops=3, sign_k=K); sync(); t_p = time.perf_counter() - t0
    pB.close()
    t0 = time.perf_counter()
    pA.run(xA, out)                                    # async on stream A
    pB = eB.plan(gB, lB, mode="pos", num_hops=3, sign_k=K)   # stream B, host syncs inside
    sync(); t_both = time.perf_counter() - t0
    pA.close(); pB.close()
    print(f"gather alone {t_g*1e3:.2f} ms, plan alone {t_p*1e3:.2f} ms, sum {1e3*(t_g+t_p):.2f}; concurrent {t_both*1e3:.2f} ms")
