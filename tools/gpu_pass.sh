# One GPU pass (through gpurun, from the repo root): the GPU test suite, then short bench lines.
#   gpurun --timeout 1100 -- 'bash tools/gpu_pass.sh r03a [pytest args]'
set -o pipefail
cd $GRAFT_REPO_ROOT; TAG=${1:-pass}; shift
O=gpurun_out/$TAG; mkdir -p $O
python3 -c 'import __graft_entry__ as g; g.build()' > $O/build.log 2>&1 || { echo "build failed"; tail -30 $O/build.log; exit 1; }
timeout -k 10 800 python -m pytest tests -m gpu -q -x "$@" > $O/pytest.log 2>&1; rc=$?; echo "pytest rc=$rc"; tail -15 $O/pytest.log
[ $rc -ne 0 ] && exit $rc
for wl in ${WORKLOADS:-pubmed_pos_k3 pubmed_pos_k5 collab_pos_k3 cora_posplus_k3 usair_pos_k2}; do
  timeout -k 10 300 python bench.py --workload $wl --no-cpu-baseline --no-api --no-pmc --steps 10 > $O/bench_$wl.json 2> $O/bench_$wl.err; echo "$wl rc=$?"
  python3 - <<PY
import json
try:
    d = json.loads(open("$O/bench_$wl.json").read().strip().splitlines()[-1])
    r = d.get("roofline", {})
    print("  ", d["config"]["workload"], "%.2f M pairs/s" % (d["value"] / 1e6), "%.2f ms" % d["ms_per_step"], r.get("phase_ms"), d.get("prepare"))
except Exception as e:
    print("  no line:", e)
PY
done
