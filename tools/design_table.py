#!/usr/bin/env python3
"""Markdown tables of DESIGN.md Part II straight from the committed bench lines: python tools/design_table.py r04"""
import json
import sys
from pathlib import Path

REPO = Path(__file__).resolve().parent.parent


def load(tag, wl):
    p = REPO / "profiles" / f"{tag}_bench_{wl}.json"
    if not p.exists():
        return None
    return json.loads(p.read_text().strip().splitlines()[-1])


def stats(tag, wl, kernel):
    p = REPO / "profiles" / f"{tag}_kernel_stats_{wl}.csv"
    if not p.exists():
        return None
    import csv

    for r in csv.DictReader(open(p)):
        if kernel in r["Name"]:
            return float(r["AverageNs"]) / 1e6, int(r["Calls"])
    return None


def main(tag):
    wls = ["pubmed_pos_k3", "pubmed_pos_k3_dense", "pubmed_sop_k3", "pubmed_pos_k5", "collab_pos_k3", "cora_posplus_k3",
           "cora_posplus_k3_real", "usair_pos_k2", "pubmed_sop_k3_2hop"]
    print("| workload | link pairs/s | ms/step | structure | link kernels | dominant kernel | ms (bench / rocprof avg) | bound | frac | L2 hit |")
    print("|---|---|---|---|---|---|---|---|---|---|")
    for wl in wls:
        d = load(tag, wl)
        if d is None:
            continue
        r = d.get("roofline") or {}
        ph = r.get("phase_ms") or {}
        k = (r.get("kernel") or "").split(" + ")[0]
        st = stats(tag, wl, k) if k else None
        lv = next((x for x in r.get("levels", []) if x["level"] == r.get("bound")), None)
        bound = f"{r.get('bound')}: {lv['bytes'] / 1e9:.1f} GB at {lv['GBps'] / 1e3:.2f} of {lv['peak_GBps'] / 1e3:.1f} TB/s" if lv else str(r.get("bound"))
        if wl.startswith("pubmed_sop"):
            structure, links = f"setup {ph.get('setup_total', 0):.2f}, scalars {ph.get('ball_scalars', 0):.2f}", "—"
        else:
            structure, links = f"{ph.get('structure', 0):.2f}", f"{ph.get('propagate', 0):.2f}"
        ms = f"{r.get('kernel_ms', 0):.2f}" + (f" / {st[0]:.2f} ({st[1]} calls)" if st else "")
        hit = r.get("l2_hit_rate")
        print(f"| {wl} | {d['value'] / 1e6:.2f} M | {d['ms_per_step']:.2f} | {structure} | {links} | `{r.get('kernel')}` | {ms} | "
              f"{bound} | {r.get('frac', 0):.2f} | {hit if hit is None else round(hit, 2)} |")
    print()
    print("| workload | cold run, six calls once (s) | … warm-up finished first | … no warm-up | warm steps through the API (pairs/s) | device output | graph / features / context prepare (ms) | CPU oracle 1 thread (pairs/s) | C restatement (pairs/s, threads) |")
    print("|---|---|---|---|---|---|---|---|---|")
    for wl in wls:
        d = load(tag, wl)
        if d is None:
            continue
        c = d.get("cold_run") or {}
        e = d.get("end_to_end_api") or {}
        p = d.get("prepare") or {}
        cb, cn = d.get("cpu_baseline") or {}, d.get("cpu_baseline_native") or {}
        f = lambda x, fmt="{:.3f}": "—" if x is None else fmt.format(x)
        print(f"| {wl} | {f(c.get('prep_wall_s'))} | {f((c.get('warmup_finished_first') or {}).get('prep_wall_s'))} | "
              f"{f((c.get('without_warmup') or {}).get('prep_wall_s'))} | {f(e.get('value'), '{:.3g}')} | "
              f"{f((e.get('device_output') or {}).get('value'), '{:.3g}')} | {f(p.get('graph_prepare_ms'), '{:.2f}')} / "
              f"{f(p.get('features_prepare_ms'), '{:.2f}')} / {f(p.get('context_ms'), '{:.1f}')} | {f(cb.get('value'), '{:.0f}')} | "
              f"{f(cn.get('value'), '{:.0f}')} ({cn.get('cores', '—')}) |")


if __name__ == "__main__":
    main(sys.argv[1] if len(sys.argv) > 1 else "r04")
