#!/bin/bash
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r02g
mkdir -p $O
cd $R
timeout -k 10 400 python -m pytest tests -m gpu -x -q -k "sop or hybrid or flow" > $O/pytest_sop.log 2>&1; rc=$?; echo "sop pytest rc=$rc"; tail -4 $O/pytest_sop.log
[ $rc -ne 0 ] && exit $rc
timeout -k 10 300 python bench.py --workload pubmed_sop_k3 --no-cpu-baseline --collect-pmc > $O/bench_sop.json 2> $O/bench_sop.err; echo "sop rc=$?"; tail -c 300 $O/bench_sop.err
