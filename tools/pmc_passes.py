#!/usr/bin/env python3
"""Counter passes of one bench step (GPU box): one rocprofv3 --pmc pass per group (never with a
trace), per-kernel-family sums printed as JSON.
    python3 tools/pmc_passes.py --workload pubmed_pos_k3 --out gpurun_out/pmc_gather.json \
        "GRBM_GUI_ACTIVE TA_BUSY_avr" "SQ_WAVE_CYCLES SQ_WAIT_INST_ANY" ..."""
import argparse, json, os, shutil, subprocess, sys, tempfile
from pathlib import Path
REPO = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(REPO / "tools"))
import pmc_collect

ap = argparse.ArgumentParser()
ap.add_argument("groups", nargs="+")
ap.add_argument("--workload", default="pubmed_pos_k3")
ap.add_argument("--out", default=str(REPO / "gpurun_out" / "pmc_passes.json"))
a = ap.parse_args()
# build before any profiler starts (a compiler child of a profiled process would exec with the GPU
# initialised by the tool's preload)
sys.path.insert(0, str(REPO))
import __graft_entry__ as _ge
_ge.build()
merged, disp = {}, {}
base = Path(tempfile.mkdtemp(prefix="s3grl_pmc_", dir="/tmp"))
for i, grp in enumerate(a.groups):
    d = base / f"g{i}"
    print(f"[pmc_passes] pass {i}: {grp}", flush=True)
    cmd = ["timeout", "-k", "10", "150", "rocprofv3", "--pmc", *grp.split(), "--output-format", "csv", "-d", str(d),
           "--", sys.executable, str(REPO / "bench.py"), "--steps", "1", "--warmup", "0", "--no-cpu-baseline", "--no-api", "--no-pmc",
           "--workload", a.workload]
    r = subprocess.run(cmd, cwd="/tmp", env=dict(os.environ, TMPDIR="/tmp"), stdout=subprocess.PIPE, stderr=subprocess.STDOUT)
    if r.returncode:
        print(f"pass {grp!r} failed rc={r.returncode}: {r.stdout.decode()[-400:]}", file=sys.stderr)
        continue
    vals, dd = pmc_collect.read_pass(d)
    for fam, cs in vals.items():
        merged.setdefault(fam, {}).update(cs)
    disp.update(dd)
    Path(a.out).write_text(json.dumps({"workload": a.workload, "dispatches": disp, "counters": merged}, indent=1))
shutil.rmtree(base, ignore_errors=True)
Path(a.out).write_text(json.dumps({"workload": a.workload, "dispatches": disp, "counters": merged}, indent=1))
print(json.dumps({k: v for k, v in merged.items() if "gather" in k or "sop_rows" in k}, indent=1))
