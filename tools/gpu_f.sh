#!/bin/bash
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r02f
mkdir -p $O
cd $R
rm -f $R/gpurun_out/parity_errors.jsonl
timeout -k 10 600 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; rc=$?; echo "pytest rc=$rc"; tail -4 $O/pytest.log
[ $rc -ne 0 ] && exit $rc
for wl in pubmed_pos_k3 pubmed_pos_k5 pubmed_sop_k3 collab_pos_k3 cora_posplus_k3 usair_pos_k2 pubmed_pos_k3_dense; do
  timeout -k 10 300 python bench.py --workload $wl --no-cpu-baseline --steps 8 > $O/bench_$wl.json 2> $O/bench_$wl.err; echo "$wl rc=$?"
done
