# Per-kernel times of the link phase with the classes launched one after the other (GPU box):
#   gpurun -- 'bash tools/class_times.sh TAG [workload]'
set -o pipefail
cd $GRAFT_REPO_ROOT; TAG=${1:-cls}; WL=${2:-collab_pos_k3}
O=gpurun_out/$TAG; mkdir -p $O
python3 -c 'import __graft_entry__ as g; g.build()' > $O/build.log 2>&1 || { tail -20 $O/build.log; exit 1; }
S3GRL_DEBUG=1 timeout -k 10 200 python bench.py --workload $WL --no-cpu-baseline --no-api --no-pmc --steps 10 > $O/bench.json 2> $O/bench.err
grep -m2 "hub cache\|classes" $O/bench.err
python3 -c "
import json
d = json.loads(open('$O/bench.json').read().strip().splitlines()[-1])
print('step %.2f ms  %.2f M/s' % (d['ms_per_step'], d['value'] / 1e6), d['roofline']['phase_ms'], d.get('prepare'))"
(cd /tmp && export TMPDIR=/tmp && S3GRL_SERIAL_CLASSES=1 timeout -k 10 240 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$O/trace -- python3 $GRAFT_REPO_ROOT/bench.py --workload $WL --steps 5 --warmup 1 --no-cpu-baseline --no-api --no-pmc > $GRAFT_REPO_ROOT/$O/trace.log 2>&1)
f=$(find $O/trace -name "*kernel_stats.csv" | head -1)
[ -n "$f" ] && python3 - <<PY
import csv, re
rows = list(csv.DictReader(open("$f")))
for r in rows:
    n = r["Name"]
    if any(k in n for k in ("link_", "count", "gather", "hub_")):
        m = re.search(r"(link_\w+|count\w*|gather\w+|hub_\w+)(<[^>]*>)?", n)
        print("   %-50s calls %5s avg %8.3f ms total %8.1f ms" % (m.group(0)[:50] if m else n[:50], r["Calls"], float(r["AverageNs"]) / 1e6, float(r["TotalDurationNs"]) / 1e6))
PY
rm -rf $O/trace
# phase stamps of the hub classes (S3GRL_STAMP_CLASSES="21 22 23 24": one class per run)
for c in ${S3GRL_STAMP_CLASSES:-}; do
  S3GRL_ONLY_CLASS=$c S3GRL_SERIAL_CLASSES=1 S3GRL_DEBUG_STAMPS=1 timeout -k 10 200 python bench.py --workload $WL --no-cpu-baseline --no-api --no-pmc --steps 1 --warmup 0 > /dev/null 2> $O/stamps_$c.err
  echo "class $c: $(grep 'link_hub_kernel phase' $O/stamps_$c.err | tail -1)"
done
