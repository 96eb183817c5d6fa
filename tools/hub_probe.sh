#!/bin/bash
# Config 5 (collab_pos_k3): phase cycles of the one-hop kernels and their classes one after the other.
# Usage (GPU box): bash tools/hub_probe.sh [out_dir]
set -e
out=${1:-gpurun_out/hub_probe}
mkdir -p "$out"
S3GRL_DEBUG=1 S3GRL_DEBUG_STAMPS=1 python bench.py --workload collab_pos_k3 --steps 2 --warmup 1 --no-cpu-baseline \
  --no-api --no-pmc --no-cold-run > "$out/stamps.json" 2> "$out/stamps.err"
grep "phase cycles\|big class\|classes:" "$out/stamps.err" | tail -6
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
S3GRL_SERIAL_CLASSES=1 rocprofv3 --kernel-trace --stats -d "$out/serial" -o serial --output-format csv -- \
  python bench.py --workload collab_pos_k3 --steps 4 --warmup 1 --no-cpu-baseline --no-api --no-pmc --no-cold-run \
  > "$out/serial.json" 2> "$out/serial.err"
python - "$out" <<'PY'
import csv, glob, sys
for f in glob.glob(sys.argv[1] + "/serial/**/*kernel_stats.csv", recursive=True):
    rows = sorted(csv.DictReader(open(f)), key=lambda r: -float(r["TotalDurationNs"]))
    for r in rows[:22]:
        print(f'{r["Name"][:90]:90s} calls {r["Calls"]:>5s} avg {float(r["AverageNs"])/1e6:8.3f} ms')
PY
