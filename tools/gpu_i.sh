#!/bin/bash
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r02i
mkdir -p $O
cd $R
timeout -k 10 500 python -m pytest tests/test_gpu_parity.py -x -q -k "packed or diffusion or headline or feature or sign_k or cora or tiny" > $O/pytest_sel.log 2>&1; rc=$?; echo "sel pytest rc=$rc"; tail -5 $O/pytest_sel.log
[ $rc -ne 0 ] && exit $rc
for v in masked unmasked; do
  if [ $v = unmasked ]; then export S3GRL_GATHER_UNMASKED=1; else unset S3GRL_GATHER_UNMASKED; fi
  for wl in pubmed_pos_k3 pubmed_pos_k5 cora_posplus_k3; do
    timeout -k 10 300 python bench.py --workload $wl --no-cpu-baseline --no-api --steps 10 > $O/bench_${wl}_$v.json 2> $O/bench_${wl}_$v.err; echo "$wl $v rc=$?"
  done
done
