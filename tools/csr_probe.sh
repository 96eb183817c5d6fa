# Where the induced-CSR link kernels spend their time (GPU box):
#   gpurun -- 'bash tools/csr_probe.sh TAG [workload]'
# step + phases, per-kernel times with the classes one after the other (rocprofv3 --kernel-trace), and the
# phase stamps of link_csr_kernel (cycles summed over workgroups; shares, not totals).
set -o pipefail
cd $GRAFT_REPO_ROOT; TAG=${1:-csr}; WL=${2:-pubmed_pos_k5}
O=gpurun_out/$TAG; mkdir -p $O
python3 -c 'import __graft_entry__ as g; g.build()' > $O/build.log 2>&1 || { tail -20 $O/build.log; exit 1; }
S3GRL_DEBUG=1 timeout -k 10 200 python bench.py --workload $WL --no-cpu-baseline --no-api --no-pmc --steps 10 > $O/bench.json 2> $O/bench.err
grep -m1 "classes" $O/bench.err
python3 -c "
import json
d = json.loads(open('$O/bench.json').read().strip().splitlines()[-1])
print('step %.2f ms  %.2f M/s' % (d['ms_per_step'], d['value'] / 1e6), d['roofline']['phase_ms'])"
(cd /tmp && export TMPDIR=/tmp && S3GRL_SERIAL_CLASSES=1 timeout -k 10 240 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$O/trace -- python3 $GRAFT_REPO_ROOT/bench.py --workload $WL --steps 5 --warmup 1 --no-cpu-baseline --no-api --no-pmc > $GRAFT_REPO_ROOT/$O/trace.log 2>&1)
f=$(find $O/trace -name "*kernel_stats.csv" | head -1)
[ -n "$f" ] && python3 - <<PY
import csv, re
rows = list(csv.DictReader(open("$f")))
for r in rows:
    n = r["Name"]
    if any(k in n for k in ("link_", "count", "gather", "csr_", "classify", "scan")):
        m = re.search(r"(link_\w+|count\w*|gather\w+|csr_\w+|classify\w+|scan\w+)(<[^>]*>)?", n)
        print("   %-50s calls %5s avg %8.3f ms total %8.1f ms" % (m.group(0)[:50] if m else n[:50], r["Calls"], float(r["AverageNs"]) / 1e6, float(r["TotalDurationNs"]) / 1e6))
PY
rm -rf $O/trace
S3GRL_SERIAL_CLASSES=1 S3GRL_DEBUG_STAMPS=1 timeout -k 10 200 python bench.py --workload $WL --no-cpu-baseline --no-api --no-pmc --steps 1 --warmup 0 > /dev/null 2> $O/stamps.err
grep 'link_csr_kernel phase' $O/stamps.err | tail -1
