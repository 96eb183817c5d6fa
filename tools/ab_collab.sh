# A/B of two builds on the collab-scale workload (GPU box): per-class times and phase stamps.
#   gpurun -- 'bash tools/ab_collab.sh TAG libA.so libB.so'   (paths relative to the repo root)
set -o pipefail
cd $GRAFT_REPO_ROOT; TAG=$1; shift
O=gpurun_out/$TAG; mkdir -p $O
for lib in "$@"; do
  name=$(basename $lib .so)
  echo "== $name"
  S3GRL_LIB=$GRAFT_REPO_ROOT/$lib timeout -k 10 200 python bench.py --workload ${WL:-collab_pos_k3} --no-cpu-baseline --no-api --no-pmc --steps 10 > $O/ab_$name.json 2> $O/ab_$name.err
  python3 -c "
import json
d = json.loads(open('$O/ab_$name.json').read().strip().splitlines()[-1])
print('  step %.2f ms' % d['ms_per_step'], d['roofline']['phase_ms'])"
  S3GRL_LIB=$GRAFT_REPO_ROOT/$lib S3GRL_SERIAL_CLASSES=1 S3GRL_DEBUG_STAMPS=1 timeout -k 10 200 python bench.py --workload ${WL:-collab_pos_k3} --no-cpu-baseline --no-api --no-pmc --steps 2 --warmup 1 > $O/ab_${name}_stamps.json 2> $O/ab_${name}_stamps.err
  grep "link_full_kernel, big class" $O/ab_${name}_stamps.err | tail -1
  (cd /tmp && export TMPDIR=/tmp && S3GRL_LIB=$GRAFT_REPO_ROOT/$lib S3GRL_SERIAL_CLASSES=1 timeout -k 10 240 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$O/trace_$name -- python3 $GRAFT_REPO_ROOT/bench.py --workload ${WL:-collab_pos_k3} --steps 5 --warmup 1 --no-cpu-baseline --no-api --no-pmc > $GRAFT_REPO_ROOT/$O/trace_$name.log 2>&1)
  f=$(find $O/trace_$name -name "*kernel_stats.csv" | head -1)
  [ -n "$f" ] && python3 - <<PY
import csv
rows = list(csv.DictReader(open("$f")))
for r in rows:
    n = r["Name"]
    if "link_full_kernel" in n or "count1" in n or "gather" in n:
        print("   %-60s calls %s avg %.3f ms total %.1f ms" % (n[n.find("link_full") if "link_full" in n else 0:][:60], r["Calls"], float(r["AverageNs"]) / 1e6, float(r["TotalDurationNs"]) / 1e6))
PY
  rm -rf $O/trace_$name
done
