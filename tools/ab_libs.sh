# A/B of alternate builds of the library on short bench runs (GPU box):
#   gpurun -- 'bash tools/ab_libs.sh TAG WORKLOAD libA.so libB.so ...'   (paths relative to the repo root)
set -o pipefail
cd $GRAFT_REPO_ROOT; TAG=$1; WL=$2; shift 2
O=gpurun_out/$TAG; mkdir -p $O
for rep in 1 2; do
for lib in "$@"; do
  name=$(basename $lib .so)
  S3GRL_LIB=$GRAFT_REPO_ROOT/$lib timeout -k 10 200 python bench.py --workload $WL --no-cpu-baseline --no-api --no-pmc --steps ${STEPS:-20} > $O/ab_${WL}_$name.json 2> $O/ab_${WL}_$name.err
  python3 -c "
import json
d = json.loads(open('$O/ab_${WL}_$name.json').read().strip().splitlines()[-1])
r = d.get('roofline_gather') or d['roofline']
print('%-28s %-16s step %.3f ms' % ('$name', '$WL', d['ms_per_step']), {k: round(v, 3) for k, v in r['phase_ms'].items()})"
done
done
