"""End-to-end rates of the drop-in path on the headline workload (PCIe note)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import __graft_entry__ as ge
ge.build()
from s3grl_amd import workloads, tuned_SIGN
from s3grl_amd.engine import Engine

w = workloads.make("pubmed_pos_k3")
li, y = w.split.all_links()
L = li.shape[1]
eng = Engine("cuda:0")
g = eng.graph(w.A); x = eng.features(w.X); links = eng.links(li)
res = eng.precompute(g, x, links, mode="pos", num_hops=3, sign_k=3)
torch.cuda.synchronize()
t0 = time.perf_counter(); res = eng.precompute(g, x, links, mode="pos", num_hops=3, sign_k=3); torch.cuda.synchronize(); t1 = time.perf_counter()
print("device only: %.1f ms  %.2f M pairs/s" % ((t1 - t0) * 1e3, L / (t1 - t0) / 1e6))
host = torch.empty(res.rows.shape, dtype=torch.float32, pin_memory=True)
host.copy_(res.rows); torch.cuda.synchronize()
t0 = time.perf_counter(); res = eng.precompute(g, x, links, mode="pos", num_hops=3, sign_k=3); host.copy_(res.rows, non_blocking=True); torch.cuda.synchronize(); t1 = time.perf_counter()
print("+ D2H into a pinned buffer (%.2f GB): %.1f ms  %.2f M pairs/s  (%.1f GB/s)" % (host.numel() * 4 / 1e9, (t1 - t0) * 1e3, L / (t1 - t0) / 1e6, host.numel() * 4 / 1e9 / (t1 - t0)))
t0 = time.perf_counter(); pg = res.rows.cpu(); t1 = time.perf_counter()
print("pageable .cpu() alone: %.1f ms" % ((t1 - t0) * 1e3))
A = w.A; X = torch.from_numpy(w.X); link_index = torch.from_numpy(li)
kw = {"sign_k": 3, "k_node_set_strategy": "intersection"}
import io, contextlib
with contextlib.redirect_stdout(io.StringIO()):
    tuned_SIGN.OptimizedSignOperations.get_PoS_prepped_ds(link_index[:, :1000], 3, A, 1.0, None, False, None, X, 1, kw, None)
    t0 = time.perf_counter()
    lst = tuned_SIGN.OptimizedSignOperations.get_PoS_prepped_ds(link_index, 3, A, 1.0, None, False, None, X, 1, kw, None)
    t1 = time.perf_counter()
print("drop-in get_PoS_prepped_ds -> list of %d LinkData on the host: %.2f s  %.2f M pairs/s" % (len(lst), t1 - t0, L / (t1 - t0) / 1e6))
