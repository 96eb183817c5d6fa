#!/bin/bash
# SQ / cache counters per kernel of one bench workload, one rocprofv3 --pmc pass per counter set.
# Usage (GPU box): bash tools/sq_probe.sh [out_dir] [workload] [kernel regex]
# Never combined with a trace (see the README).  Classes run one after the other (S3GRL_SERIAL_CLASSES).
set -e
out=${1:-gpurun_out/sq_probe}
wl=${2:-collab_pos_k3}
rx=${3:-link_hub_kernel<[^>]*>|link_full_kernel<[^>]*>|link_kernel<[^>]*>|gather_[a-z_]*kernel<[^>]*>}
mkdir -p "$out"
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
rocprofv3 --list-avail > "$out/avail.txt" 2>&1 || true
i=0
if [ -n "$SQ_PROBE_SETS" ]; then   # own counter sets, separated by ';'  (e.g. "TCP_UTCL1_REQUEST_sum TCP_UTCL1_TRANSLATION_MISS_sum")
  IFS=';' read -ra sets <<< "$SQ_PROBE_SETS"
else
  sets=("SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU" \
           "SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" \
           "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY" \
           "SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_ANY" \
           "SQ_INST_CYCLES_SALU SQ_THREAD_CYCLES_VALU SQ_INSTS_SMEM SQ_INSTS_FLAT")
fi
for set in "${sets[@]}"; do
  # (a sixth pass with FETCH_SIZE WRITE_SIZE aborted inside rocprofv3 on collab_pos_k3 and hung the call:
  # the byte counters come from bench.py's own passes, tools/pmc_passes.py)
  echo "pass $((i+1)): $set"
  i=$((i+1))
  S3GRL_SERIAL_CLASSES=1 timeout -k 10 150 rocprofv3 --pmc $set -d "$out/p$i" -o p --output-format csv -- \
    python3 bench.py --workload $wl --steps 1 --warmup 0 --no-cpu-baseline --no-api --no-pmc --no-cold-run \
    > "$out/p$i.json" 2> "$out/p$i.err" || echo "pass $i failed"
done
python3 - "$out" "$rx" <<'PY'
import csv, sys, collections, re
from pathlib import Path
rx = re.compile(sys.argv[2])
vals = collections.defaultdict(lambda: collections.defaultdict(float))
calls = collections.defaultdict(set)
for f in Path(sys.argv[1]).rglob("*counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        m = rx.search(r["Kernel_Name"])
        if m:
            vals[m.group(0)][r["Counter_Name"]] += float(r["Counter_Value"])
            calls[m.group(0)].add((str(f), r.get("Dispatch_Id", "")))
for k, d in sorted(vals.items()):
    print(k)
    for c, v in sorted(d.items()):
        print(f"   {c:36s} {v:.4g}")
    wc, busy = d.get("SQ_WAVE_CYCLES"), d.get("SQ_BUSY_CYCLES")
    if wc and busy:   # (SQ_WAVE_CYCLES / SQ_ACTIVE_INST_* count quad-cycles; SQ_BUSY_CYCLES cycles, summed over 32 SEs)
        simd = busy / 32 * 1024
        print(f"   -> VALU busy {4 * d.get('SQ_ACTIVE_INST_VALU', 0) / simd:.2f} of the SIMD cycles, "
              f"waves per SIMD {4 * wc / simd:.1f}, waiting {d.get('SQ_WAIT_ANY', 0) / wc:.2f} of their cycles, "
              f"lanes per VALU instruction {d.get('SQ_THREAD_CYCLES_VALU', 0) / max(d.get('SQ_ACTIVE_INST_VALU', 1), 1):.1f}")
PY
