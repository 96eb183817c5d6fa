#!/bin/bash
# first GPU pass of round 2: GPU tests, then the bench lines with in-run counters
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r02a
mkdir -p $O
cd $R
timeout -k 10 700 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc=$?" | tee -a $O/pytest.log
tail -3 $O/pytest.log
timeout -k 10 400 python bench.py --collect-pmc > $O/bench_k3.json 2> $O/bench_k3.err; echo "k3 rc=$?"
tail -c 600 $O/bench_k3.err
