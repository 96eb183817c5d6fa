#!/usr/bin/env python3
"""Benchmark of the S3GRL operator precompute on MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--workload pubmed_pos_k3]

One "step" = one pass of the hot path (plan: extraction + operator rows; run: feature gather)
over the workload's whole link list (train/valid/test x pos/neg, 164 000 links for PubMed) with
the graph and X already resident in HBM.  Prints ONE JSON line (rank 0).

N > 1: STRONG scaling, as BASELINE.json's north_star describes it — the SAME link list is assigned to
the N GPUs once at set-up (graph + X replicated): pair-aware shards balanced by the engine's per-link cost
model (`parallel.ShardPlan`: both directions of a pair on one rank; `--contiguous-shards` = cost-balanced
contiguous ranges), every rank runs the engine on its links and the result is reassembled on every rank by
RCCL all-gathers over xGMI that are timed inside the step (`s3grl_amd/parallel.py`: the pieces of a shard
are gathered while the next piece is computed; what a rank can form itself — operator 0, reversed
duplicates, the cheapest links — does not travel).  value = links of the whole list / max-over-ranks step
time.  A rank that dies or a rendezvous that times out ends the run with a non-zero exit code and a
one-line reason on stderr.
Launched either by `python -m torch.distributed.run --nproc-per-node N bench.py --gpus N ...`
(RANK / LOCAL_RANK / WORLD_SIZE in the environment) or directly as `python bench.py --gpus N`,
which starts the N ranks itself as child processes BEFORE anything touches the GPU.
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import time
from pathlib import Path

import numpy as np

REPO = Path(__file__).resolve().parent
sys.path.insert(0, str(REPO))

# Ceilings, /opt/skills/guides/MI355X_MICROARCH.md (§ Memory hierarchy):
HBM_PEAK_GBS = 8000.0      # HBM3E spec peak (6.3 TB/s is what a float4 copy achieves)
L2_PEAK_GBS = 34500.0      # aggregate L2 bandwidth, 8 XCDs x 4 MiB
MALL_ROWS_GBS = 8600.0     # uniformly random rows of a 38 MB table served by the Infinity Cache
MALL_ROWS_BIG_GBS = 7900.0  # the same from a 151 MB table (7.4-7.9 measured)
MALL_BYTES = 256 << 20


def algorithmic_bytes(stats, F, K):
    """B_link = 8n + 4 vol(S) + 4 n F + 4 R (K+1)(1+F), summed exactly from the plan.
    Returns (whole path over ALL links, gather share over all links, gather share over the links
    the gather launch actually processes)."""
    n, vol, R = stats["total_nodes"], stats["total_volume"], stats["total_rows"]
    out_bytes = 4 * R * (K + 1) * (1 + F)
    gather_all = 4 * n * F + out_bytes
    gather_launch = 4 * stats.get("extracted_nodes", n) * F + out_bytes
    return 8 * n + 4 * vol + gather_all, gather_all, gather_launch


def hierarchical_roofline(levels, kernel_ms):
    """levels: [(name, bytes, peak_GBs, what)].  A kernel cannot finish before its slowest level
    has moved its bytes: t >= max_i bytes_i / peak_i.  frac = that bound / measured time (<= 1);
    `bound` names the level that sets it."""
    rows = []
    for name, nbytes, peak, what in levels:
        if nbytes is None:
            continue
        t_ms = nbytes / (peak * 1e9) * 1e3
        rows.append({"level": name, "bytes": int(nbytes), "peak_GBps": peak, "min_ms": t_ms,
                     "GBps": nbytes / (kernel_ms * 1e-3) / 1e9 if kernel_ms > 0 else 0.0,
                     "frac": t_ms / kernel_ms if kernel_ms > 0 else 0.0, "what": what})
    best = max(rows, key=lambda r: r["frac"])
    return best, rows


def committed_pmc(workload, L):
    """A PMC summary committed under profiles/ for this workload, newest round first."""
    cands = sorted((REPO / "profiles").glob("r*_pmc*.json"), reverse=True)
    for p in cands:
        try:
            rec = json.loads(p.read_text())
        except Exception:
            continue
        if rec.get("workload") == workload and rec.get("links") == L and rec.get("hbm_bytes_per_launch"):
            return rec, p.name
    return None, None


_PMC_PASSES = []     # [(merged, disp)] of this run's counter passes: one set serves every kernel family


def _run_pmc_passes(args):
    import shutil
    import tempfile

    if _PMC_PASSES:
        return _PMC_PASSES[0]
    sys.path.insert(0, str(REPO / "tools"))
    import pmc_collect

    rocprof = shutil.which("rocprofv3") or "/opt/rocm/bin/rocprofv3"
    if not Path(rocprof).exists():
        return None
    # never under a profiler already (a trace of this command must not start counter passes of its own)
    if any(k.startswith(("ROCPROF", "ROCP_")) for k in os.environ) or \
            "rocprof" in os.environ.get("LD_PRELOAD", "").lower():
        sys.stderr.write("[bench] running under a profiler: counter passes skipped\n")
        return None
    merged, disp = {}, {}
    base = Path(tempfile.mkdtemp(prefix="s3grl_pmc_", dir="/tmp"))
    env = dict(os.environ, TMPDIR="/tmp")
    for grp in (["FETCH_SIZE"], ["WRITE_SIZE"], ["TCC_HIT_sum", "TCC_MISS_sum"]):
        d = base / grp[0]
        cmd = ["timeout", "-k", "10", "150", rocprof, "--pmc", *grp, "--output-format", "csv", "-d", str(d),
               "--", sys.executable, str(REPO / "bench.py"), "--steps", "1", "--warmup", "0",
               "--no-cpu-baseline", "--no-api", "--no-pmc", "--workload", args.workload]
        r = subprocess.run(cmd, cwd="/tmp", env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT)
        if r.returncode != 0:
            sys.stderr.write(f"[bench] counter pass {grp} failed rc={r.returncode}\n")
            return None
        vals, dd = pmc_collect.read_pass(d)
        for fam, cs in vals.items():
            merged.setdefault(fam, {}).update(cs)
        disp.update(dd)
    shutil.rmtree(base, ignore_errors=True)
    _PMC_PASSES.append((merged, disp))
    return _PMC_PASSES[0]


def collect_pmc(args, kernel_family, per_step=False):
    """Counter passes of this very command as child processes (one rocprofv3 pass per counter
    group, no trace: MI355X_MICROARCH.md, HBM section), the program itself after `--`.
    per_step: the family runs as several launches per step (the link kernels: one per LDS class) —
    report the sum over the launches of the one step the child runs instead of a per-launch mean."""
    passes = _run_pmc_passes(args)
    if passes is None:
        return None
    merged, disp = passes
    if isinstance(kernel_family, (list, tuple)):   # several kernels share the phase: their counters add up
        fams = [f for f in kernel_family if f in merged]
        g = {}
        for f in fams:
            for k, v in merged[f].items():
                g[k] = g.get(k, 0.0) + v
        launches = max(sum(disp.get(f, 0) for f in fams), 1)
        kernel_family = " + ".join(fams)
    else:
        g = merged.get(kernel_family)
        launches = max(disp.get(kernel_family, 1), 1)
    if not g or "FETCH_SIZE" not in g or "WRITE_SIZE" not in g:
        return None
    n = 1 if per_step else launches
    rec = {"kernel": kernel_family, "dispatches": launches, "per": "step (sum over its launches)" if per_step else "launch",
           "FETCH_SIZE_KB": g["FETCH_SIZE"] / n, "WRITE_SIZE_KB": g["WRITE_SIZE"] / n,
           "hbm_bytes_per_launch": (2.0 * g["FETCH_SIZE"] + g["WRITE_SIZE"]) * 1024.0 / n,
           "counters": merged}
    if "TCC_HIT_sum" in g:
        rec["l2_hit_rate"] = g["TCC_HIT_sum"] / max(g["TCC_HIT_sum"] + g.get("TCC_MISS_sum", 0.0), 1.0)
    return rec


def link_kernel_roofline(args, w, link_index, stats, phase_ms, K, F, collect):
    """Roofline block of the link-kernel phase (extraction + operator coefficients): bound = HBM in
    the sense of SURVEY §8(d) (the CSR terms 8n + 4 vol(S) of B_link), achieved = those bytes over the
    phase's duration between HIP events.  One-hop plans on big graphs (link_full_kernel) never touch
    vol(S): the bytes they physically request are counted exactly from the plan."""
    ms = phase_ms["propagate"]
    onehop = stats.get("oriented_entries", 0) > 0
    hub_links = stats.get("hub_links", 0)
    # (the families whose counters add up to the phase; collect_pmc keeps those that ran)
    kname = (("link_full_kernel + link_hub_kernel + link_tiny_kernel" if hub_links
              else "link_full_kernel + link_tiny_kernel") if onehop else "link_kernel")
    n_ext, vol, sup, pairs = stats["extracted_nodes"], stats["total_volume"], stats["total_support"], stats["num_row_pairs"]
    folded = stats.get("folded_links", 0)
    L = link_index.shape[1]
    alg = 8 * n_ext + 4 * vol
    deg = np.diff(w.A.indptr)
    ends = int(deg[link_index[0]].sum() + deg[link_index[1]].sum())
    # support counts a folded link twice (algorithmic); the kernels write one list per extracted pair
    sup_ext = sup * (L - folded) / max(L, 1)
    if onehop:
        # links served from a cached hub neighbourhood (link_hub_kernel) report what they requested themselves
        ends_full = max(ends * (L - folded) / max(L, 1) - stats.get("hub_endpoint_entries", 0), 0)
        reads = {"endpoint_rows": 4 * ends_full,                             # the two CSR rows, staged in LDS
                 "oriented_rows": 4 * stats["oriented_entries"],             # probes of the masked adjacency
                 "row_bounds": 16 * (n_ext - stats.get("hub_nodes", 0)),     # indptr + fwd_indptr pairs per node
                 "hub_links": stats.get("hub_read_bytes", 0)}                # endpoint rows, walked rows, staged cache
    else:
        reads = {"csr_rows": 4 * vol * (1 + max(K - 1, 1)), "row_bounds": 8 * n_ext * K,
                 "node_lists": 4 * n_ext}
    writes = {"node_ids": 4 * n_ext, "coefficients": 8 * K * sup_ext, "jobs_and_limits": (64 + 12 * K) * pairs,
              "levels_and_row_nodes": (128 + 16) * (L - folded)}
    physical = sum(reads.values()) + sum(writes.values())
    traffic, traffic_source, l2_hit, pmc = None, None, None, None
    if collect:
        rec = collect_pmc(args, kname.split(" + ") if " + " in kname else kname, per_step=True)
        if rec:
            traffic, l2_hit = rec["hbm_bytes_per_launch"], rec.get("l2_hit_rate")
            traffic_source = "collected by this run: child rocprofv3 --pmc passes of the same command, summed over the step's launches"
            pmc = {k: rec[k] for k in rec if k != "counters"}
    sec = ms * 1e-3
    return {
        "kernel": kname, "kernel_ms": ms,
        "kernel_ms_what": "the link-kernel phase between HIP events on the context's stream: one launch per LDS "
                          "class on forked streams, running concurrently",
        "bound": "hbm", "achieved": alg / sec / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
        "frac": alg / sec / 1e9 / HBM_PEAK_GBS,
        "algorithmic_bytes": alg,
        "algorithmic_what": "SURVEY §8(d) CSR terms of B_link over the extracted links: 8n + 4 vol(S)",
        "physical_bytes": physical, "physical_GBps": physical / sec / 1e9,
        "physical_reads": reads, "physical_writes": writes,
        "traffic": traffic, "traffic_source": traffic_source, "l2_hit_rate": l2_hit, "pmc": pmc,
        "limiter": ("LDS issue of the K pulls (cached hub rows + found edges, a few entries per row) and of the "
                    "neighbour tests of the walked rows (link_hub_kernel); dependent-load latency of the per-link "
                    "chain rows -> rank merge -> hash -> oriented-row probes -> CSR -> K pulls (link_full_kernel; "
                    "links of at most 64 nodes: one lane per node, no barrier, link_tiny_kernel): "
                    "the classes' serial times add up to the phase, HBM (this fraction) is not the bound") if onehop else
                   "VALU issue of the row walk (76-86 % busy at under half of its lanes), DESIGN §2.1",
        "phase_ms": phase_ms,
        "folded_links": folded, "hub_links": hub_links, "mean_subgraph_nodes": stats["total_nodes"] / max(L, 1),
    }


def choose_replicate_fraction(dist, backend, eng, parallel, compute, shards, li_dev, cost, rank, world, K, F,
                              exchange_op0, chunks):
    """How much of the list should every rank compute itself instead of receiving it?  One compute-only
    step of the sharded list and one all-gather of the full payload are timed (set-up, RCCL only); with the
    cost model's share of the cheapest g of the list, a step takes about
        max( own compute without them + their compute in full ,  first piece + (1 - g) * all-gather )
    and g is taken from a grid.  Rank 0 decides for everybody.  What can fail on ONE rank (the probe
    buffer, the compute-only steps) runs before any collective, and the ranks agree on its success first:
    a rank that failed must not leave the others waiting, nor build a ShardPlan of its own."""
    import torch

    if backend != "nccl" and not os.environ.get("S3GRL_BENCH_TUNE_ANY_BACKEND"):   # (rehearsal hook: exercise this code over gloo)
        return 0.0, {"note": "not RCCL: the exchange cannot be timed, nothing replicated"}
    small = eng.device if backend == "nccl" else "cpu"      # where the few-number collectives live
    L = li_dev.shape[1]
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(4)]
    buf, t_local, err = None, 0.0, None
    try:
        sent = [sum(nr) for nr in shards.transport(chunks, eng.device)[3]]
        rmax = 2 * max(sent)
        kx = K + 1 if exchange_op0 else K
        buf = torch.empty((world * rmax, kx, F + 1), dtype=torch.float32, device=eng.device)
        for rep_ in range(2):      # (the second run: buffers and caches warm)
            ev[0].record()
            parallel.sharded_precompute(compute, li_dev, rank=rank, world_size=world, gather=False, rows_per_link=2,
                                        row_shape=(K + 1, F + 1), device=eng.device, reuse_buffers=True, shards=shards)
            ev[1].record()
            torch.cuda.synchronize()
        t_local = ev[0].elapsed_time(ev[1])
    except Exception as e:
        err = repr(e)
    agreed = torch.tensor([0 if err else 1], dtype=torch.int64, device=small)
    dist.all_reduce(agreed, op=dist.ReduceOp.MIN)
    if int(agreed.item()) == 0:
        del buf
        return 0.0, {"error": err or "the probe failed on another rank", "note": "nothing replicated (all ranks)"}
    t_comp = torch.tensor([t_local], device=small)
    t_all = t_comp.clone()
    dist.all_reduce(t_all, op=dist.ReduceOp.SUM)
    dist.all_reduce(t_comp, op=dist.ReduceOp.MAX)
    best = None
    for _ in range(2):
        dist.barrier()
        ev[2].record()
        parallel._all_gather(buf, buf[rank * rmax:(rank + 1) * rmax], None)
        ev[3].record()
        torch.cuda.synchronize()
        t = ev[2].elapsed_time(ev[3])
        best = t if best is None else min(best, t)
    del buf
    t_ag = torch.tensor([best], device=small)
    dist.all_reduce(t_ag, op=dist.ReduceOp.MAX)
    t_comp, t_all, t_ag = float(t_comp), float(t_all), float(t_ag)
    c = np.sort(np.asarray(cost, dtype=np.float64))
    share = np.concatenate([[0.0], np.cumsum(c) / max(c.sum(), 1e-30)])   # cost share of the cheapest k links
    grid = [0.0, 0.05, 0.1, 0.15, 0.2, 0.25, 0.3, 0.35, 0.4, 0.45, 0.5, 0.55, 0.6]
    est = {}
    for gfr in grid:
        cf = float(share[int(L * gfr)])
        comp = t_comp * (1.0 - cf) + cf * t_all          # own share without them + all of them
        wire = t_comp * (1.0 - cf) / max(chunks, 1) + (1.0 - gfr) * t_ag
        est[gfr] = max(comp, wire) + 0.15 * (gfr > 0)     # (one more plan per step)
    pick = min(grid, key=lambda gfr: est[gfr])
    choice = torch.tensor([pick], device=small)
    dist.broadcast(choice, src=0)
    pick = round(float(choice), 4)
    return pick, {"compute_only_ms_max": t_comp, "compute_only_ms_sum": t_all, "allgather_alone_ms": t_ag,
                  "estimated_step_ms": {str(k): round(v, 3) for k, v in est.items()}}


def rccl_summary(path):
    """What RCCL said about the communicator it built (NCCL_DEBUG=INFO, subsystems INIT and GRAPH, written to
    `path` by this rank): channels, the transports of its connections, ring / tree lines — the facts DESIGN §6's
    estimates depend on (a direct all-gather over 7 xGMI links, or a ring through one).  Never raises."""
    import re

    out = {"log": str(path)}
    try:
        text = Path(path).read_text(errors="replace")
    except OSError as e:
        return dict(out, error=repr(e))
    lines = [ln.split("NCCL INFO", 1)[1].strip() for ln in text.splitlines() if "NCCL INFO" in ln]
    out["info_lines"] = len(lines)
    ch = [ln for ln in lines if re.match(r"Channel \d+/\d+", ln)]
    if ch:
        m = re.match(r"Channel \d+/(\d+)", ch[0])
        out["channels"] = int(m.group(1)) if m else len(ch)
        out["first_ring"] = ch[0][:160]
    via = {}
    for ln in lines:
        m = re.search(r"via (\S+)", ln)
        if m and "->" in ln:
            via[m.group(1)] = via.get(m.group(1), 0) + 1
    if via:
        out["connections_via"] = via
    for key, pat in (("trees", r"^Trees "), ("rings_connected", r"Connected all rings"), ("trees_connected", r"Connected all trees"),
                     ("thresholds", r"threadThresholds"), ("comm", r"comm 0x\S+ rank \d+ nranks \d+"),
                     ("version", r"(RCCL|NCCL) version"), ("xgmi", r"(?i)xgmi"), ("algo", r"(?i)\balgo")):
        hit = [ln for ln in lines if re.search(pat, ln)]
        if hit:
            out[key] = hit[0][:200] if key not in ("xgmi", "algo") else [h[:160] for h in hit[:4]]
    return out


def cpu_baseline(w, link_index, y, budget_s, max_links):
    """The oracle (reference-structured CPU restatement, oracle/s3grl_oracle.py) timed on one
    core over a bounded sample of the same link list.  kind = "port"."""
    import torch

    import oracle

    torch.set_num_threads(1)
    rng = np.random.default_rng(6)
    half = max_links // 2
    idx = np.concatenate([rng.choice(np.flatnonzero(y == 1), half, replace=False),
                          rng.choice(np.flatnonzero(y == 0), half, replace=False)])
    idx = idx[rng.permutation(len(idx))]
    kw = {"sign_k": w.sign_k, "k_node_set_strategy": "intersection"}
    fn = {"pos": oracle.get_PoS_prepped_ds, "pos_plus": oracle.get_PoS_Plus_prepped_ds}.get(w.mode)
    done, t0 = 0, time.perf_counter()
    chunk = 50
    if w.mode == "sop_restricted":
        t0 = time.perf_counter()
        P = oracle.global_normalized_powers(w.A, w.sign_k, np.float32)
        sel = idx[:min(max_links, 400)]
        oracle.get_SoP_restricted_ds(P, link_index[:, sel], w.num_hops, w.A, w.X, 1, dtype=np.float32)
        done = len(sel)
    elif w.mode == "sop":
        t0 = time.perf_counter()
        P = oracle.global_normalized_powers(w.A, w.sign_k, np.float32)
        sel = idx[:max_links]
        oracle.get_SoP_prepped_ds(P, link_index[:, sel], w.A, w.X, 1, dtype=np.float32)
        done = len(sel)
    else:
        while done < len(idx) and time.perf_counter() - t0 < budget_s:
            sel = idx[done:done + chunk]
            fn(link_index[:, sel], w.num_hops, w.A, w.X, 1, kw, dtype=np.float32)
            done += len(sel)
    dt = time.perf_counter() - t0
    return {"value": done / dt, "unit": "link pairs/s", "cores": 1, "kind": "port",
            "sample": f"{done} links (half pos, half neg, seed 6) of the same list in {dt:.1f} s, "
                      f"1 thread; host has {os.cpu_count()} cpus"}


def cpu_baseline_native(w, link_index, y, max_links):
    """Second, stronger CPU figure: the plain-C restatement (oracle/s3grl_oracle_c.c,
    row propagation instead of SpGEMM — the engine's own formulation) on all host cores."""
    from oracle import c_oracle

    c_oracle.build()
    rng = np.random.default_rng(6)
    half = min(max_links, len(y)) // 2
    idx = np.concatenate([rng.choice(np.flatnonzero(y == 1), half, replace=False),
                          rng.choice(np.flatnonzero(y == 0), half, replace=False)])
    idx = idx[rng.permutation(len(idx))]
    threads = c_oracle.cpu_threads()
    c_oracle.pos_rows(link_index[:, idx[:64]], w.num_hops, w.A, w.X, w.sign_k, plus=w.mode == "pos_plus",
                      threads=threads)                       # page in, spin up the team
    t0 = time.perf_counter()
    c_oracle.pos_rows(link_index[:, idx], w.num_hops, w.A, w.X, w.sign_k, plus=w.mode == "pos_plus",
                      threads=threads)
    dt = time.perf_counter() - t0
    return {"value": len(idx) / dt, "unit": "link pairs/s", "cores": threads, "kind": "port",
            "sample": f"{len(idx)} links (half pos, half neg, seed 6) of the same list in {dt:.1f} s, "
                      f"C + OpenMP restatement, fp64 accumulation, {threads} threads"}


def end_to_end_api(w, link_index, y):
    """The same list through the reference's own operator API (s3grl_amd.tuned_SIGN, the drop-in
    boundary): pos and neg calls per split like sgrl_link_pred.py:195-203, result = per-link Data
    sequences with CPU tensors, then the caller's `pos_list + neg_list`.  D2H included."""
    import torch

    from s3grl_amd import tuned_SIGN as ts

    xt = torch.from_numpy(w.X)
    kw = {"sign_k": w.sign_k, "use_feature": True, "sign_type": "PoS" if w.mode != "sop" else "SoP",
          "optimize_sign": True, "k_heuristic": 1 if w.mode == "pos_plus" else 0,
          "k_node_set_strategy": "intersection"}
    calls = []
    for s in ("train", "valid", "test"):
        pos, neg = w.split.links[s]
        calls += [(torch.from_numpy(pos), 1), (torch.from_numpy(neg), 0)]

    def one(li, yy):
        if w.mode == "sop":
            return ts.OptimizedSignOperations.get_SoP_prepped_ds([None] * w.sign_k, li, w.A, xt, yy)
        fn = ts.OptimizedSignOperations.get_PoS_Plus_prepped_ds if w.mode == "pos_plus" \
            else ts.OptimizedSignOperations.get_PoS_prepped_ds
        return fn(li, w.num_hops, w.A, 1.0, None, False, None, xt, yy, kw, None)

    import contextlib
    import io

    def timed(device_output, env=None):
        """The six calls + the list concatenation; device_output: S3GRL_OUTPUT_DEVICE=cuda (the
        per-link objects view device memory: no copy to the host — what a GPU-side loader wants)."""
        if device_output:
            os.environ["S3GRL_OUTPUT_DEVICE"] = "cuda"
        for k, v in (env or {}).items():
            os.environ[k] = v
        try:
            best = None
            for _ in range(2):
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                total, lists = 0, []
                with contextlib.redirect_stdout(io.StringIO()):
                    for li, yy in calls:
                        if li.shape[1] == 0:
                            continue
                        lst = one(li, yy)
                        total += len(lst)
                        lists.append(lst)
                    for i in range(0, len(lists) - 1, 2):
                        _ = lists[i] + lists[i + 1]
                torch.cuda.synchronize()
                dt = time.perf_counter() - t0
                best = dt if best is None else min(best, dt)
                del lists, lst
            return best, total
        finally:
            os.environ.pop("S3GRL_OUTPUT_DEVICE", None)
            for k in (env or {}):
                os.environ.pop(k, None)

    best = None
    for _ in range(2):          # first pass uploads A and x and sizes the pinned staging buffer
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        total, lists = 0, []
        with contextlib.redirect_stdout(io.StringIO()):    # the operators print their flow name
            for li, yy in calls:
                if li.shape[1] == 0:
                    continue
                lst = one(li, yy)
                total += len(lst)
                lists.append(lst)
            for i in range(0, len(lists) - 1, 2):
                _ = lists[i] + lists[i + 1]                 # sgrl_link_pred.py:204
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        best = dt if best is None else min(best, dt)
        first = lists[0][0]
        _ = first.x.shape, first[f"x{w.sign_k}"].shape
        del lists, first, lst
    try:
        dev_s, dev_total = timed(True)
        on_device = {"value": dev_total / dev_s, "seconds": dev_s,
                     "what": "the same six calls with S3GRL_OUTPUT_DEVICE=cuda: the per-link objects view device "
                             "memory (no copy to the host)"}
    except Exception as e:
        on_device = {"error": repr(e)}
    try:
        pin_s, pin_total = timed(False, {"S3GRL_HOST_OUTPUT": "pinned"})
        pinned_pool = {"value": pin_total / pin_s, "seconds": pin_s,
                       "what": "S3GRL_HOST_OUTPUT=pinned, best of two passes: the results in pooled page-locked memory, "
                               "reused once the previous pass's lists are dropped (a loop; a real run makes one cold pass)"}
    except Exception as e:
        pinned_pool = {"error": repr(e)}
    ts.clear_cache()
    return {"value": total / best, "unit": "link pairs/s", "seconds": best, "links": total,
            "what": "OptimizedSignOperations.get_*_prepped_ds of s3grl_amd.tuned_SIGN over the 6 "
                    "(split, pos/neg) calls + the caller's list concatenation; per-link objects "
                    "with CPU tensors (D2H included: fresh huge-page pageable memory filled through a page-locked ring)",
            "device_output": on_device, "pinned_pool": pinned_pool}


def cold_run(workload):
    """A fresh child process that runs the drop-in's six calls ONCE (tools/cold_run.py) — the figure a reference run
    reports as its prep time; bench.py's own numbers are warm steps.  Returns the child's JSON (or an error)."""
    def child(*extra):
        cmd = [sys.executable, str(REPO / "tools" / "cold_run.py"), "--workload", workload] + list(extra)
        try:
            r = subprocess.run(cmd, capture_output=True, text=True, timeout=300, cwd=str(REPO))
            lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
            if r.returncode != 0 or not lines:
                return {"error": "exit code %d: %s" % (r.returncode, (r.stderr or "").strip()[-300:])}
            return json.loads(lines[-1])
        except Exception as e:
            return {"error": repr(e)}

    out = child()
    # the same with the import-time warm-up switched off: HIP initialisation, the context and the code objects
    # then load inside the first call (what the warm-up thread otherwise does while the caller reads its dataset)
    keys = ("prep_wall_s", "link_pairs_per_s", "first_call_ms", "error")
    cold = child("--no-warmup")
    out["without_warmup"] = {k: cold.get(k) for k in keys if k in cold}
    # ... and with the warm-up finished before the clock starts: a caller that spends a few tenths of a second
    # between importing the module and its first operator call (the reference loads its dataset there)
    warm = child("--wait-warmup")
    out["warmup_finished_first"] = {k: warm.get(k) for k in keys if k in warm}
    return out


def visible_gpus(default):
    """GPUs of this node WITHOUT any HIP / torch call (the parent of the ranks must not initialise
    the GPU: its children exec): the visibility variables when set, else the KFD topology nodes
    that have SIMDs (CPU nodes have none).  `default` when neither can be read."""
    for var in ("HIP_VISIBLE_DEVICES", "ROCR_VISIBLE_DEVICES", "CUDA_VISIBLE_DEVICES"):
        v = os.environ.get(var)
        if v is not None:
            return len([t for t in v.split(",") if t.strip() != ""])
    n, seen = 0, False
    for p in Path("/sys/class/kfd/kfd/topology/nodes").glob("*/properties"):
        try:
            props = dict(line.split()[:2] for line in p.read_text().splitlines() if len(line.split()) >= 2)
        except OSError:
            continue
        seen = True
        n += 1 if int(props.get("simd_count", "0")) > 0 else 0
    return n if seen else default


def spawn_ranks(args):
    """`python bench.py --gpus N` with no launcher: start the N ranks as fresh child processes.
    Nothing in this (parent) process has touched the GPU — not even through torch."""
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    ndev = visible_gpus(args.gpus)
    procs = []
    for r in range(args.gpus):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(args.gpus),
                   LOCAL_WORLD_SIZE=str(args.gpus), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        if ndev < args.gpus:                 # rehearsal on a box with fewer GPUs: share them, gloo
            env.setdefault("S3GRL_BENCH_BACKEND", "gloo")
        procs.append(subprocess.Popen([sys.executable, str(REPO / "bench.py")] + sys.argv[1:], env=env))
    # A dead rank must not leave the others (and this parent) waiting in a collective: the first child to
    # fail, or the deadline, ends them all — non-zero exit code, one line saying why.
    deadline = time.monotonic() + float(os.environ.get("S3GRL_BENCH_TIMEOUT", "1500"))
    rc, why = 0, None
    try:
        while True:
            codes = [p.poll() for p in procs]
            bad = [(r, c) for r, c in enumerate(codes) if c not in (None, 0)]
            if bad:
                rc = bad[0][1] if bad[0][1] > 0 else 1
                why = "rank %d exited with code %d" % bad[0]
                break
            if all(c == 0 for c in codes):
                break
            if time.monotonic() > deadline:
                rc, why = 124, "no result after S3GRL_BENCH_TIMEOUT = %.0f s (ranks still running: %s)" % (
                    float(os.environ.get("S3GRL_BENCH_TIMEOUT", "1500")), [r for r, c in enumerate(codes) if c is None])
                break
            time.sleep(0.2)
    finally:
        for p in procs:
            if p.poll() is None:
                p.terminate()
        for p in procs:
            try:
                p.wait(timeout=10)
            except subprocess.TimeoutExpired:
                p.kill()
    if why:
        sys.stderr.write("bench.py --gpus %d FAILED: %s; the other ranks were stopped\n" % (args.gpus, why))
    return rc


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--workload", default="pubmed_pos_k3")
    ap.add_argument("--cpu-seconds", type=float, default=15.0)
    ap.add_argument("--cpu-links", type=int, default=2000)
    ap.add_argument("--cpu-native-links", type=int, default=40000)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-api", action="store_true", help="skip the end_to_end_api measurement")
    ap.add_argument("--no-cold-run", action="store_true", help="skip the cold_run measurement (a fresh child process)")
    ap.add_argument("--collect-pmc", action="store_true",
                    help="(default at N = 1) collect FETCH_SIZE / WRITE_SIZE / L2 hit rate of this command in child "
                         "rocprofv3 passes, about 40 s; falls back to the newest summary under profiles/, labelled")
    ap.add_argument("--no-pmc", action="store_true", help="skip the counter passes (quick runs)")
    ap.add_argument("--max-links", type=int, default=0, help="truncate the link list (debug)")
    ap.add_argument("--chunks", type=int, default=0,
                    help="N > 1: pieces per rank (all-gather of piece c overlaps the compute of c+1); 0 = pieces of about "
                         "8 000 links, between 2 and 8 (the exchange, not the compute, bounds the step: the sooner the "
                         "first piece is on the wire the better, at ~0.15 ms of launches and one sync per piece)")
    ap.add_argument("--contiguous-shards", action="store_true",
                    help="N > 1: contiguous ranges of the list instead of pair-aware shards (comparison)")
    ap.add_argument("--exchange-operator0", action="store_true",
                    help="N > 1: all-gather whole rows (operator 0 = X[node] included) instead of letting every rank "
                         "fill operator 0 from its own copy of X (comparison)")
    ap.add_argument("--replicate-fraction", type=float, default=-1.0,
                    help="N > 1: share of the list (its cheapest links) that every rank computes itself instead of "
                         "receiving it; -1 = chosen at set-up from one measured compute-only step and one measured "
                         "all-gather (RCCL only), 0 = off")
    ap.add_argument("--exchange-mirrors", action="store_true",
                    help="N > 1: all-gather the rows of reversed duplicates too instead of rebuilding them on every "
                         "rank from their primaries' rows (comparison)")
    ap.add_argument("--no-allgather", action="store_true",
                    help="N > 1: every rank keeps its shard (data-parallel consumer); no collective")
    ap.add_argument("--verify", action="store_true",
                    help="N > 1: check the reassembled tensor bit for bit against an unsharded run")
    args = ap.parse_args()
    args.collect_pmc = args.collect_pmc or not args.no_pmc   # single-GPU runs only (checked where it is used)

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(spawn_ranks(args))

    import torch

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    dist = None
    rccl_log = None
    # one rank per GPU; a rehearsal with more ranks than GPUs (S3GRL_BENCH_BACKEND=gloo on a
    # one-GPU box) shares the devices round-robin
    dev_index = local_rank % max(torch.cuda.device_count(), 1)
    backend = os.environ.get("S3GRL_BENCH_BACKEND", "nccl")
    if world > 1:
        import torch.distributed as dist

        import datetime

        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        torch.cuda.set_device(dev_index)
        if backend == "nccl" and "NCCL_DEBUG" not in os.environ:
            # RCCL's own account of the communicator (channels, transports) goes into the JSON line
            rccl_log = "/tmp/s3grl_rccl_%d_rank%d.log" % (os.getppid(), rank)
            os.environ.update(NCCL_DEBUG="INFO", NCCL_DEBUG_SUBSYS="INIT,GRAPH", NCCL_DEBUG_FILE=rccl_log)
        wait = datetime.timedelta(seconds=float(os.environ.get("S3GRL_BENCH_RENDEZVOUS_S", "300")))
        try:
            if backend == "nccl":
                dist.init_process_group("nccl", device_id=torch.device("cuda", dev_index), timeout=wait)
            else:
                dist.init_process_group(backend, timeout=wait)
        except Exception as e:
            sys.stderr.write("bench.py rank %d/%d FAILED: init_process_group(%s) did not complete within %.0f s: %r\n"
                             % (rank, world, backend, wait.total_seconds(), e))
            sys.exit(3)
    torch.cuda.set_device(dev_index)

    import __graft_entry__ as ge

    if rank == 0:
        ge.build()
    if dist is not None:
        dist.barrier()
    from s3grl_amd import parallel, workloads
    from s3grl_amd.engine import Engine

    w = workloads.make(args.workload)
    link_index, y = w.split.all_links()
    if args.max_links:
        link_index, y = link_index[:, :args.max_links], y[:args.max_links]
    L = link_index.shape[1]
    F, K = w.X.shape[1], w.sign_k

    torch.cuda.synchronize()
    t_ctx = time.perf_counter()
    eng = Engine(f"cuda:{dev_index}")
    torch.cuda.synchronize()
    t_ctx = time.perf_counter() - t_ctx
    # one-off operand preparation, outside the step like the uploads — reported, not hidden:
    # graph = CSR validation + degree order (+ oriented rows on big graphs); features = aligned copy,
    # density count and the packed rows
    ip_d = torch.as_tensor(np.asarray(w.A.indptr, dtype=np.int64)).to(eng.device)
    ix_d = torch.as_tensor(np.asarray(w.A.indices, dtype=np.int32)).to(eng.device)
    x_d = torch.as_tensor(w.X).to(device=eng.device, dtype=torch.float32).contiguous()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    from s3grl_amd.engine import Features, Graph
    g = Graph(eng, ip_d, ix_d, w.A.shape[0])
    torch.cuda.synchronize()
    t1 = time.perf_counter()
    x = Features(eng, x_d, "auto")
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    prepare = {"graph_prepare_ms": (t1 - t0) * 1e3, "features_prepare_ms": (t2 - t1) * 1e3,
               "context_ms": t_ctx * 1e3, "code_objects_ms": eng.preload_ms,
               "note": "graph / features: once per (graph, X), inputs already on the device; context_ms: once per "
                       "process — the engine context and the library's GPU code (code_objects_ms per unit: HIP would "
                       "otherwise load them at the first launch of each kernel family, i.e. inside the first graph / "
                       "plan / run).  None of it is part of a step; cold_run is the figure that includes all of it"}
    links = eng.links(link_index)
    fixed_rows = w.mode in ("pos", "sop", "sop_restricted")

    out = None
    traffic_req = None

    def step_single():
        nonlocal out
        if w.mode == "sop":
            res = eng.precompute(g, x, links, mode="sop", sign_k=K, out=out)
            out = res.rows
            return res.stats
        plan = eng.plan(g, links, mode=w.mode, num_hops=w.num_hops, sign_k=K)
        if out is None:
            out = torch.empty((plan.total_rows, K + 1, F + 1), dtype=torch.float32,
                              device=eng.device)
        plan.run(x, out)
        st = dict(plan.stats)
        plan.close()
        return st

    # ---- N > 1: shard, compute, all-gather ---------------------------------------------------
    shard_info = {}
    if world > 1:
        # shard weights, once at set-up (like the uploads): exact subgraph sizes from the engine's
        # sizing pass; SoP extracts no subgraph, its cost is one gather per link + the scalar ball
        # a reversed duplicate is priced at what it costs once it is folded into its primary, and the
        # assignment keeps the two directions of a pair on one rank (parallel.shard_assignment)
        cost = parallel.measured_cost(eng, g, link_index, w.num_hops, mode=w.mode) if w.mode != "sop" \
            else parallel.sop_cost(eng, g, w.A, link_index)
        if args.chunks <= 0:
            args.chunks = min(8, max(2, int(round(L / world / 8000.0))))
        li_dev = torch.as_tensor(link_index).to(eng.device)
        shards = parallel.ShardPlan(li_dev, world, cost, pair_aware=not args.contiguous_shards, device=eng.device)
        gather = not args.no_allgather
        timers = {}
        fold_stats = {}
        if fixed_rows:
            compute = parallel.engine_compute(eng, g, x, mode=w.mode, num_hops=w.num_hops, sign_k=K, stats=fold_stats)

            x_dev = x.tensor

            def fill_operator0(fl):
                """operator 0 of every link, [z | X[node]] with z = 1 for both centre rows (what the engine
                writes there: plain copies of the replicated X) — formed on every rank instead of exchanged"""
                fl[:, :, 0, 0] = 1.0
                fl[:, 0, 0, 1:] = x_dev[li_dev[0]]
                fl[:, 1, 0, 1:] = x_dev[li_dev[1]]

            # --- links every rank computes itself (the exchange, not the compute, bounds the step) ------------
            replicate_info = None
            frac = args.replicate_fraction
            if gather and not args.contiguous_shards and frac != 0.0:
                if frac < 0.0:      # (its one-rank failures are agreed on inside; a failing collective ends the run)
                    frac, replicate_info = choose_replicate_fraction(
                        dist, backend, eng, parallel, compute, shards, li_dev, cost, rank, world, K, F,
                        args.exchange_operator0, args.chunks)
                if frac > 0.0:
                    rep = parallel.replicate_cheapest(link_index, cost, frac)
                    shards = parallel.ShardPlan(li_dev, world, cost, pair_aware=True, device=eng.device, replicate=rep)
                    replicate_info = dict(replicate_info or {}, fraction=frac, links=int(rep.sum()),
                                          cost_share=float(np.asarray(cost)[rep].sum() / max(np.asarray(cost).sum(), 1e-30)))
            shard_info["replicated"] = replicate_info

            def step_sharded():
                return parallel.sharded_precompute(
                    compute, li_dev, rank=rank, world_size=world, gather=gather,
                    rows_per_link=2, chunks=args.chunks if gather else 1, row_shape=(K + 1, F + 1),
                    device=eng.device, timers=timers, reuse_buffers=True, shards=shards,
                    local_operator0=None if args.exchange_operator0 else fill_operator0,
                    mirror_rows=not args.exchange_mirrors and not args.contiguous_shards)
        else:
            def compute_ragged(shard):
                res = eng.precompute(g, x, eng.links(shard), mode=w.mode, num_hops=w.num_hops, sign_k=K)
                fold_stats["links"] = fold_stats.get("links", 0) + res.stats["num_links"]
                fold_stats["folded_links"] = fold_stats.get("folded_links", 0) + res.stats["folded_links"]
                return res.rows, res.row_ptr

            def step_sharded():
                return parallel.sharded_precompute(compute_ragged, li_dev, rank=rank, world_size=world,
                                                   gather=gather, shards=shards)
        b = shards.bounds
        shard_info.update({"bounds": b, "links_per_rank": [b[r + 1] - b[r] for r in range(world)]})
        if fixed_rows and gather and not args.exchange_mirrors and not args.contiguous_shards:
            sent = [sum(nr) for nr in shards.transport(args.chunks, eng.device)[3]]
            shard_info["links_sent_per_rank"] = sent
            shard_info["links_not_sent"] = L - sum(sent)

    def step():
        if world > 1:
            return step_sharded()
        return step_single()

    def barrier():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    stats = None
    for _ in range(max(args.warmup, 0)):
        stats = step()
    barrier()
    eng.set_profiling(True)
    if world > 1:
        fold_stats.clear()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        stats = step()
    barrier()
    dt_local = time.perf_counter() - t0
    tm = eng.timings()
    eng.set_profiling(False)
    if world == 1 and w.mode != "sop" and not x.is_sparse and rank == 0:
        # measurement, outside the timed region: the bytes the gather launch requests (exact)
        plan = eng.plan(g, links, mode=w.mode, num_hops=w.num_hops, sign_k=K)
        traffic_req = plan.gather_traffic(x)
        plan.close()
    dt = dt_local
    per_rank = None
    if dist is not None:
        t = torch.tensor([dt], dtype=torch.float64, device=eng.device if backend == "nccl" else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
        mine = {"rank": rank, "step_ms": dt_local / args.steps * 1e3,
                "structure_ms": tm["structure_ms"] / args.steps, "propagate_ms": tm["propagate_ms"] / args.steps,
                "gather_ms": tm["gather_ms"] / args.steps,
                "sop_ms": (tm["sop_setup_ms"] + tm["sop_run_ms"]) / args.steps,
                "plans_per_step": tm["plans"] / args.steps,
                "links": fold_stats.get("links", 0) // max(args.steps, 1),
                "folded_links": fold_stats.get("folded_links", 0) // max(args.steps, 1)}
        per_rank = [None] * world
        dist.all_gather_object(per_rank, mine)

    verify = None
    if world > 1 and args.verify:
        rows_sh, ptr_sh, _ = step()
        if fixed_rows:
            ref = torch.empty((2 * L, K + 1, F + 1), dtype=torch.float32, device=eng.device)
            parallel.engine_compute(eng, g, x, mode=w.mode, num_hops=w.num_hops, sign_k=K)(li_dev, ref)
        else:
            ref = eng.precompute(g, x, links, mode=w.mode, num_hops=w.num_hops, sign_k=K).rows
        if args.no_allgather:
            mine_ix = shards.order_dev[shard_info["bounds"][rank]:shard_info["bounds"][rank + 1]]
            ok = fixed_rows and torch.equal(rows_sh.view((-1, 2) + tuple(ref.shape[1:])),
                                            ref.view((L, 2) + tuple(ref.shape[1:]))[mine_ix])
        else:
            ok = rows_sh.shape == ref.shape and torch.equal(rows_sh, ref)
        flag = torch.tensor([1 if ok else 0], dtype=torch.int64,
                            device=eng.device if backend == "nccl" else "cpu")
        dist.all_reduce(flag, op=dist.ReduceOp.MIN)
        verify = bool(flag.item())
        del ref

    # one all-gather of the whole padded tensor on its own, outside the timed region: what the
    # collective costs when nothing overlaps it
    allgather_alone = None
    if world > 1 and fixed_rows and not args.no_allgather and backend == "nccl":
        rmax = 2 * max(shard_info.get("links_sent_per_rank") or shard_info["links_per_rank"])   # rows that travel
        kx = K + 1 if args.exchange_operator0 else K          # operators that travel
        buf = torch.empty((world * rmax, kx, F + 1), dtype=torch.float32, device=eng.device)
        torch.cuda.synchronize()
        dist.barrier()
        ts = []
        for _ in range(3):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            dist.all_gather_into_tensor(buf, buf[rank * rmax:(rank + 1) * rmax])
            e1.record()
            torch.cuda.synchronize()
            ts.append(e0.elapsed_time(e1))
        nbytes = buf.numel() * 4
        allgather_alone = {"ms": min(ts), "bytes": nbytes,
                           "algbw_GBps": nbytes / (min(ts) * 1e-3) / 1e9,
                           "busbw_GBps": nbytes * (world - 1) / world / (min(ts) * 1e-3) / 1e9}
        del buf

    if rank == 0:
        ms_per_step = dt / args.steps * 1e3
        value = L * args.steps / dt
        line = {
            "metric": "link-pair precomputes/sec", "value": value, "unit": "link pairs/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": ms_per_step,
            "higher_is_better": True, "scaling": "strong", "vs_baseline": None,
            "dtype": "f64" if w.mode == "sop" else "f32", "data": "synthetic",
            "config": {"workload": args.workload, "mode": w.mode, "sign_k": K, "num_hops": w.num_hops,
                       "links_per_step": L, "num_nodes": w.split.num_nodes,
                       "features": F, "graph": "real topology, synthetic features (BASELINE.md §3)"},
            "prepare": prepare,
        }
        if world > 1:
            comp = [r["structure_ms"] + r["propagate_ms"] + r["gather_ms"] + r["sop_ms"] for r in per_rank]
            line["multi_gpu"] = {
                "sharding": ("contiguous link ranges" if args.contiguous_shards else
                             "pair-aware shards (both directions of a pair on one rank, so the fold of reversed "
                             "duplicates survives; pairs in order of first appearance)") +
                            " balanced by the engine's per-link cost model on exact subgraph sizes (the sizing "
                            "pass over the whole list, once at set-up: parallel.measured_cost / "
                            "s3grl_plan_link_cost, a folded duplicate priced at its output rows); graph + X replicated",
                "folded_links_total": sum(r["folded_links"] for r in per_rank),
                "collective": None if args.no_allgather else
                              ("%d padded all_gather_into_tensor per step (pieces of a shard are gathered on "
                               "RCCL's stream while the next piece is computed) + scatter into list order" % args.chunks +
                               ("" if args.exchange_operator0 else
                                "; operators 1..K travel, operator 0 (= [1 | X[node]]) is filled by every rank itself") +
                               ("" if args.exchange_mirrors or args.contiguous_shards else
                                "; a link that follows its reverse in a rank's piece does not travel: its rows are the "
                                "primary's in swapped order, rebuilt on every rank (%d of %d links)"
                                % (shard_info.get("links_not_sent", 0), L))
                               if fixed_rows else "sizes + one padded all_gather_into_tensor + compaction"),
                "backend": backend, "rccl": rccl_summary(rccl_log) if rccl_log else None,
                "links_per_rank": shard_info["links_per_rank"],
                "links_sent_per_rank": shard_info.get("links_sent_per_rank"),
                "replicated_links": shard_info.get("replicated"),
                "per_rank": per_rank,
                "compute_ms_per_rank": comp,
                "imbalance_max_over_mean": max(comp) / (sum(comp) / len(comp)) if sum(comp) > 0 else None,
                "exposed_comm_and_host_ms": ms_per_step - max(comp),
                "allgather_alone": allgather_alone,
                "verified_bit_equal_to_unsharded": verify,
                "note": "value counts the links of the WHOLE list once (strong scaling); with "
                        "--no-allgather every rank keeps its shard (what a data-parallel trainer "
                        "consumes) and no collective runs",
            }
        if world == 1 and w.mode != "sop" and stats:
            path_bytes, gather_all, gather_bytes = algorithmic_bytes(stats, F, K)
            launches = max(tm["gather_launches"], 1.0)
            gather_ms = tm["gather_ms"] / launches
            kname = "gather_packed_kernel" if x.is_packed else ("gather_narrow_kernel" if F <= 128 else "gather_kernel")
            traffic, traffic_source, l2_hit = None, None, None
            if args.collect_pmc and world == 1:
                # (the child runs ONE step: the sum over the family's dispatches is the launch of a step)
                rec = collect_pmc(args, kname.split(" + ") if " + " in kname else kname, per_step=True)
                if rec:
                    traffic, l2_hit = rec["hbm_bytes_per_launch"], rec.get("l2_hit_rate")
                    traffic_source = "collected by this run: child rocprofv3 --pmc passes of the same command"
                    line["pmc"] = {k: rec[k] for k in rec if k != "counters"}
                    line["pmc_counters"] = rec["counters"]
            if traffic is None:
                rec, name = committed_pmc(args.workload, L)
                if rec:
                    traffic, l2_hit = rec["hbm_bytes_per_launch"], rec.get("l2_hit_rate")
                    traffic_source = f"profiles/{name} (an earlier run of this workload, NOT this run)"
            req = traffic_req or {}
            operand_bytes = (x.nnz * 16 + 32 * w.X.shape[0] * ((F + 511) // 512)) if x.is_packed \
                else 4 * w.X.shape[0] * ((F + 3) // 4 * 4)
            requested = sum(req.get(k, 0) for k in ("ids", "headers", "features", "coefficients", "output",
                                                    "x_rows", "job_meta"))
            stream = sum(req.get(k, 0) for k in ("ids", "coefficients", "output"))
            mall_peak = MALL_ROWS_GBS if operand_bytes <= (64 << 20) else \
                (MALL_ROWS_BIG_GBS if operand_bytes <= MALL_BYTES else HBM_PEAK_GBS)
            levels = [
                ("l2", requested or None, L2_PEAK_GBS,
                 "every byte the kernel's loads and stores request (exact, from the plan: node ids, "
                 "row headers, feature chunks, coefficients, operator-0 rows, output) against the "
                 "aggregate L2 bandwidth"),
                ("fabric", traffic, mall_peak,
                 "bytes that leave the L2s (rocprofv3 FETCH_SIZE x2 + WRITE_SIZE: Infinity-Cache hits "
                 "included; FETCH_SIZE tallies 128-byte line requests at 64 B — calibrated, "
                 "profiles/*pmc_calibration.json) against the Infinity-Cache random-row rate for an operand of %.0f MB"
                 % (operand_bytes / 1e6)),
                ("hbm", stream or None, HBM_PEAK_GBS,
                 "bytes that must cross HBM: the node-id and coefficient lists the link kernels wrote "
                 "(GBs, read once) and the output (written once); the feature operand itself stays in "
                 "the Infinity Cache" if operand_bytes <= MALL_BYTES else
                 "bytes that must cross HBM: id/coefficient lists, output"),
            ]
            best, rows = hierarchical_roofline(levels, gather_ms)
            phase_ms = {"structure": tm["structure_ms"] / max(tm["plans"], 1.0),
                        "propagate": tm["propagate_ms"] / max(tm["plans"], 1.0), "gather": gather_ms}
            dominant = max(phase_ms, key=phase_ms.get)
            line["dominant_phase"] = dominant
            if dominant == "propagate":
                # the link kernels dominate (config 5: one-hop plans on a big graph): THEIR block is the
                # line's roofline; the gather's block moves to roofline_gather
                line["roofline"] = link_kernel_roofline(args, w, link_index, stats, phase_ms, K, F,
                                                        collect=args.collect_pmc and world == 1)
            line["roofline" if dominant != "propagate" else "roofline_gather"] = {
                "kernel": kname, "kernel_ms": gather_ms,
                "bound": best["level"], "achieved": best["GBps"], "peak": best["peak_GBps"], "unit": "GB/s",
                "frac": best["frac"], "levels": rows,
                "traffic": traffic, "traffic_source": traffic_source, "l2_hit_rate": l2_hit,
                "requested_bytes": req, "physical_bytes": requested,
                "feature_operand_bytes": operand_bytes,
                "algorithmic": {
                    "bytes_per_launch": gather_bytes, "GBps": gather_bytes / (gather_ms * 1e-3) / 1e9,
                    "frac_of_hbm_peak": gather_bytes / (gather_ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
                    "bytes_all_links": gather_all, "path_bytes_per_step": path_bytes,
                    "path_GBps": path_bytes / (ms_per_step * 1e-3) / 1e9,
                    "note": "SURVEY §8(d) contract figure on DENSE fp32 rows of X, every gathered link: "
                            "4nF + 4R(K+1)(1+F).  It is not traffic: the packed operand stores only the "
                            "non-zero 16-byte chunks and X sits in the Infinity Cache, so this rate can "
                            "exceed the HBM peak; `frac` above is computed from physical bytes"},
                "phase_ms": phase_ms,
                "link_kernels": {
                    "ms": tm["propagate_ms"] / max(tm["plans"], 1.0),
                    "algorithmic_bytes": 8 * stats["extracted_nodes"] + 4 * stats["total_volume"],
                    "note": "8n + 4 vol(S) once per extracted link (SURVEY §8d CSR terms); bound by VALU "
                            "issue / dependent-load latency, not bytes (DESIGN §4)"},
                "mean_subgraph_nodes": stats["total_nodes"] / max(L, 1),
                "folded_links": stats.get("folded_links", 0),
                "feature_operand": ("packed rows: %d non-zero 16-byte chunks of %d" % (x.nnz, w.X.shape[0] * ((F + 3) // 4)))
                                   if x.is_packed else "dense rows",
            }
        if world == 1 and w.mode == "sop":
            runs = max(tm["sop_runs"], 1.0)
            rows_ms = tm["sop_rows_ms"] / runs
            N = w.split.num_nodes
            ldy = (F + 1) // 2 * 2
            req_read = L * (2 * (K + 1) * 4 * F + 3 * 8 * K + 16)    # (+ the lo planes of the rows that cancel: a few %)
            out_bytes = L * 2 * (K + 1) * 4 * (F + 1)
            table = (K + 1) * N * ldy * 4
            traffic, traffic_source, l2_hit = None, None, None
            if args.collect_pmc and world == 1:
                rec = collect_pmc(args, "sop_rows_kernel")
                if rec:
                    traffic, l2_hit = rec["hbm_bytes_per_launch"], rec.get("l2_hit_rate")
                    traffic_source = "collected by this run: child rocprofv3 --pmc passes of the same command"
                    line["pmc"] = {k: rec[k] for k in rec if k != "counters"}
                    line["pmc_counters"] = rec["counters"]
            if traffic is None:
                rec, name = committed_pmc(args.workload, L)
                if rec:
                    traffic, l2_hit = rec["hbm_bytes_per_launch"], rec.get("l2_hit_rate")
                    traffic_source = f"profiles/{name} (an earlier run of this workload, NOT this run)"
            levels = [
                ("l2", req_read + out_bytes, L2_PEAK_GBS,
                 "bytes requested: f32 rows Y_i[src], Y_i[dst], i = 0..K, + scalars, + the f32 output"),
                ("hbm", traffic if traffic else out_bytes + min(table, req_read), HBM_PEAK_GBS,
                 ("measured bytes beyond L2 (FETCH_SIZE x2 + WRITE_SIZE)" if traffic else
                  "lower bound of the bytes that cross HBM: the output once + every row of the f32 "
                  "Y table once") + "; the table (%.0f MB) exceeds the 256 MB Infinity Cache" % (table / 1e6)),
            ]
            best, rws = hierarchical_roofline(levels, rows_ms)
            setup_ms = tm["sop_setup_ms"] / max(tm["sop_setups"], 1.0)
            spmm_ms = tm["sop_spmm_ms"] / max(tm["sop_setups"], 1.0)
            nnz = int(w.A.nnz)
            spmm_alg = K * (4 * nnz + 16 * N * F)
            alg_link = 4 * F * (2 * K + 2) + 8 * (K + 1) * (F + 1)
            line["roofline"] = {
                "kernel": "sop_rows_kernel", "kernel_ms": rows_ms,
                "bound": best["level"], "achieved": best["GBps"], "peak": best["peak_GBps"], "unit": "GB/s",
                "frac": best["frac"], "levels": rws,
                "traffic": traffic, "traffic_source": traffic_source, "l2_hit_rate": l2_hit,
                "algorithmic": {
                    "bytes_per_link": alg_link, "bytes_per_launch": alg_link * L,
                    "GBps": alg_link * L / (rows_ms * 1e-3) / 1e9 if rows_ms > 0 else None,
                    "note": "SURVEY §8(d) SoP figure in fp32 terms: 4F(2K+2) + 8(K+1)(1+F) per link "
                            "(the scalar-ball CSR term is reported with the scalar phase); the kernel "
                            "reads the Y rows as f32 and forms the rows that cancel again from an f32 hi + lo "
                            "pair (DESIGN §2.3)"},
                "phase_ms": {"setup_total": setup_ms, "setup_spmm": spmm_ms,
                             "run_total": tm["sop_run_ms"] / runs, "rows_kernel": rows_ms,
                             "ball_scalars": tm["sop_run_ms"] / runs - rows_ms},
                "setup": {"kernel": "spmm_norm_kernel x K", "ms": spmm_ms,
                          "algorithmic_bytes": spmm_alg,
                          "GBps": spmm_alg / (spmm_ms * 1e-3) / 1e9 if spmm_ms > 0 else None,
                          "frac_of_hbm_peak": spmm_alg / (spmm_ms * 1e-3) / 1e9 / HBM_PEAK_GBS if spmm_ms > 0 else None,
                          "note": "K (4 nnz + 16 N F): f64 Y read and written once per operator "
                                  "(SURVEY §8d setup term, f64); included in value like the reference's "
                                  "timed region includes its global powers"},
            }
        if world == 1 and not args.no_api and w.mode != "sop_restricted":   # (not a reference flow: no operator API)
            try:
                line["end_to_end_api"] = end_to_end_api(w, link_index, y)
            except Exception as e:   # the bench line must not die on the optional leg
                line["end_to_end_api"] = {"error": repr(e)}
        if world == 1 and not args.no_cold_run and not args.no_api and not ge.under_profiler() and w.mode != "sop_restricted":
            line["cold_run"] = cold_run(args.workload)
        if world == 1 and not args.no_cpu_baseline:
            line["cpu_baseline"] = cpu_baseline(w, link_index, y, args.cpu_seconds, args.cpu_links)
            line["speedup_vs_cpu_baseline"] = value / line["cpu_baseline"]["value"]
            if w.mode not in ("sop", "sop_restricted"):
                line["cpu_baseline_native"] = cpu_baseline_native(w, link_index, y, args.cpu_native_links)
        print(json.dumps(line), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()
    eng.close()


if __name__ == "__main__":
    main()
