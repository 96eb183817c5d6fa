#!/usr/bin/env python3
"""Summarise rocprofv3 counter passes of `bench.py` into profiles/<tag>_pmc.json (+ pmc_latest.json).

    python tools/pmc_collect.py --tag r01_v6 --build "packed rows" DIR [DIR ...]

Each DIR is the `-d` output directory of ONE `rocprofv3 --pmc ... -- python3 bench.py --steps 1
--warmup 0 --no-cpu-baseline` pass (counters are collected in separate passes, never together with
a trace: /opt/skills/guides/MI355X_MICROARCH.md, HBM section).  Counter values are summed per
kernel family over the dispatches of the run; with --steps 1 --warmup 0 that is one step.

gfx950 correction applied here and recorded in the output: FETCH_SIZE reports 1/2 of the bytes of
16-byte-per-lane coalesced reads, so read bytes = 2 * FETCH_SIZE * 1024; WRITE_SIZE is exact."""
import argparse
import csv
import json
import sys
from collections import defaultdict
from pathlib import Path

REPO = Path(__file__).resolve().parent.parent
FAMILIES = ["gather_packed_kernel", "gather_sparse_kernel", "gather_narrow_kernel", "gather_kernel", "link_full_kernel", "link_hub_kernel", "link_tiny_kernel", "link_csr_kernel", "link_kernel",
            "csr_count_kernel", "count_balls_kernel", "count1_kernel", "count_kernel", "combine_kernel", "sop_rows_kernel", "sop_scalar_kernel",
            "spmm_norm_kernel"]


def family(name):
    for f in FAMILIES:
        if "::" + f + "<" in name or "::" + f + "(" in name:
            return f
    return None


def read_pass(d):
    """{family: {counter: sum}} and {family: dispatches} from every *counter_collection.csv under d."""
    vals = defaultdict(lambda: defaultdict(float))
    disp = defaultdict(set)
    files = list(Path(d).rglob("*counter_collection.csv"))
    if not files:
        sys.exit(f"{d}: no *counter_collection.csv")
    for f in files:
        with open(f, newline="") as fh:
            rd = csv.DictReader(fh)
            cols = {c.lower(): c for c in rd.fieldnames}
            kn, cn, cv = cols["kernel_name"], cols["counter_name"], cols["counter_value"]
            did = cols.get("dispatch_id")
            for row in rd:
                fam = family(row[kn])
                if fam is None:
                    continue
                vals[fam][row[cn]] += float(row[cv])
                if did:
                    disp[fam].add(row[did])
    return vals, {k: len(v) for k, v in disp.items()}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("dirs", nargs="+")
    ap.add_argument("--tag", required=True)
    ap.add_argument("--build", default="")
    ap.add_argument("--workload", default="pubmed_pos_k3")
    ap.add_argument("--links", type=int, default=164000)
    ap.add_argument("--bench-json", help="bench.py output of the same build (for the algorithmic bytes)")
    args = ap.parse_args()

    merged = defaultdict(dict)
    dispatches = {}
    for d in args.dirs:
        vals, disp = read_pass(d)
        for fam, cs in vals.items():
            merged[fam].update(cs)
        dispatches.update(disp)
    gather = next((f for f in FAMILIES[:3] + ["sop_rows_kernel"] if f in merged), None)
    if gather is None:
        sys.exit("no gather / sop_rows kernel in the counter files")
    g = merged[gather]
    out = {"workload": args.workload, "links": args.links, "build": args.build, "kernel": gather,
           "dispatches_in_run": dispatches,
           "correction": "gfx950 FETCH_SIZE reports 1/2 of the bytes of 16-B-per-lane coalesced reads "
                         "(MI355X_MICROARCH.md, HBM): read bytes = 2*FETCH_SIZE*1024; WRITE_SIZE exact",
           "counters": {fam: dict(cs) for fam, cs in merged.items()},
           "commands": ["rocprofv3 --pmc <one counter group> --output-format csv -d <DIR> -- "
                        "python3 bench.py --steps 1 --warmup 0 --no-cpu-baseline   (one pass per group)"]}
    n = max(dispatches.get(gather, 1), 1)
    if "FETCH_SIZE" in g and "WRITE_SIZE" in g:
        out["FETCH_SIZE_KB"] = g["FETCH_SIZE"] / n
        out["WRITE_SIZE_KB"] = g["WRITE_SIZE"] / n
        out["hbm_bytes_per_launch"] = (2.0 * g["FETCH_SIZE"] + g["WRITE_SIZE"]) * 1024.0 / n
    if "TCC_HIT_sum" in g and "TCC_MISS_sum" in g:
        out["l2_hit_rate"] = g["TCC_HIT_sum"] / max(g["TCC_HIT_sum"] + g["TCC_MISS_sum"], 1.0)
    if args.bench_json:
        b = json.loads(Path(args.bench_json).read_text().strip().splitlines()[-1])
        out["algorithmic_bytes_per_launch"] = b["roofline"]["algorithmic"]["bytes_per_launch"]
        out["bench_kernel_ms"] = b["roofline"]["kernel_ms"]
    out["note"] = ("FETCH_SIZE counts fabric-side requests, Infinity-Cache hits included: with X resident in "
                   "the 256 MB Infinity Cache this is cache-served fabric traffic, not DRAM traffic.")
    (REPO / "profiles").mkdir(exist_ok=True)
    for name in (f"{args.tag}_pmc.json", "pmc_latest.json"):
        (REPO / "profiles" / name).write_text(json.dumps(out, indent=1) + "\n")
    print(json.dumps({k: out[k] for k in out if k not in ("counters", "commands", "correction", "note")}, indent=1))


if __name__ == "__main__":
    main()
