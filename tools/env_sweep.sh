# Sweep of an environment hook over short bench runs (GPU box):
#   gpurun -- 'bash tools/env_sweep.sh r03c pubmed_pos_k3 "S3GRL_SPLIT_T=0" "S3GRL_SPLIT_T=4096 S3GRL_SPLIT_SEG_SHIFT=11" ...'
set -o pipefail
cd $GRAFT_REPO_ROOT; TAG=$1; WL=$2; shift 2
O=gpurun_out/$TAG; mkdir -p $O
python3 -c 'import __graft_entry__ as g; g.build()' > $O/build.log 2>&1 || { echo "build failed"; tail -30 $O/build.log; exit 1; }
i=0
for setting in "$@"; do
  i=$((i+1))
  env $setting timeout -k 10 200 python bench.py --workload $WL --no-cpu-baseline --no-api --no-pmc --steps ${STEPS:-20} > $O/sweep_${WL}_$i.json 2> $O/sweep_${WL}_$i.err
  python3 - <<PY
import json
try:
    d = json.loads(open("$O/sweep_${WL}_$i.json").read().strip().splitlines()[-1])
    r = d.get("roofline_gather") or d.get("roofline", {})
    print("%-60s %.3f ms  %s" % ("$setting", d["ms_per_step"], {k: round(v, 3) for k, v in r.get("phase_ms", {}).items()}))
except Exception as e:
    print("$setting: no line:", e)
PY
done
