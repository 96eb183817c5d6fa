# HIP API timeline of a cold process (GPU box): which calls of the first Graph() / first step are slow?
#   gpurun -- 'bash tools/cold_trace.sh TAG [workload]'
set -o pipefail
cd $GRAFT_REPO_ROOT; TAG=${1:-cold}; WL=${2:-usair_pos_k2}
O=$GRAFT_REPO_ROOT/gpurun_out/$TAG; mkdir -p $O
python3 -c 'import __graft_entry__ as g; g.build()' > $O/build.log 2>&1 || { tail -20 $O/build.log; exit 1; }
(cd /tmp && export TMPDIR=/tmp && timeout -k 10 240 rocprofv3 --hip-trace --kernel-trace --output-format csv -d $O/trace -- python3 $GRAFT_REPO_ROOT/tools/cold_probe.py --workload $WL > $O/probe.log 2>&1)
cat $O/probe.log | grep -v amdgpu.ids
f=$(find $O/trace -name "*hip_api_trace.csv" | head -1)
[ -n "$f" ] && python3 - <<PY
import csv
rows = list(csv.DictReader(open("$f")))
print(len(rows), "HIP API calls; columns:", list(rows[0].keys()))
key_s = [k for k in rows[0] if "Start" in k][0]; key_e = [k for k in rows[0] if "End" in k][0]
name = [k for k in rows[0] if k in ("Function", "Name")][0]
for r in rows: r["dur"] = (int(r[key_e]) - int(r[key_s])) / 1e6
rows.sort(key=lambda r: int(r[key_s]))
t0 = int(rows[0][key_s])
slow = [r for r in rows if r["dur"] > 0.3]
print("calls slower than 0.3 ms, in time order (ms since first call, duration, name):")
for r in slow[:120]:
    print("  %10.2f  %8.2f  %s" % ((int(r[key_s]) - t0) / 1e6, r["dur"], r[name]))
import collections
agg = collections.defaultdict(lambda: [0, 0.0])
for r in rows:
    agg[r[name]][0] += 1; agg[r[name]][1] += r["dur"]
print("by function:")
for k, (c, t) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:15]:
    print("  %-40s calls %6d total %9.2f ms" % (k, c, t))
PY
rm -rf $O/trace
