#!/bin/bash
# second GPU pass of round 2: new GPU tests, all bench lines, rehearsal of the N > 1 path
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r02b
mkdir -p $O
cd $R
rm -f $R/gpurun_out/parity_errors.jsonl
timeout -k 10 500 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; rc=$?; echo "pytest rc=$rc"; tail -15 $O/pytest.log
[ $rc -ne 0 ] && exit $rc
timeout -k 10 300 python bench.py --no-cpu-baseline > $O/bench_k3.json 2> $O/bench_k3.err; echo "k3 rc=$?"; tail -c 300 $O/bench_k3.err
timeout -k 10 300 python bench.py --gpus 2 --steps 3 --warmup 1 --verify > $O/bench_k3_g2.json 2> $O/bench_k3_g2.err; echo "g2 rc=$?"; tail -c 600 $O/bench_k3_g2.err
timeout -k 10 300 python bench.py --gpus 3 --steps 3 --warmup 1 --verify --workload usair_pos_k2 --chunks 2 > $O/bench_usair_g3.json 2> $O/bench_usair_g3.err; echo "g3 rc=$?"; tail -c 600 $O/bench_usair_g3.err
timeout -k 10 400 python bench.py --workload pubmed_pos_k3_dense --collect-pmc --no-cpu-baseline --no-api > $O/bench_dense.json 2> $O/bench_dense.err; echo "dense rc=$?"; tail -c 300 $O/bench_dense.err
timeout -k 10 400 python bench.py --workload pubmed_sop_k3 --collect-pmc --no-cpu-baseline > $O/bench_sop.json 2> $O/bench_sop.err; echo "sop rc=$?"; tail -c 300 $O/bench_sop.err
for wl in pubmed_pos_k5 collab_pos_k3 cora_posplus_k3 usair_pos_k2; do
  timeout -k 10 300 python bench.py --workload $wl --no-cpu-baseline --steps 5 > $O/bench_$wl.json 2> $O/bench_$wl.err; echo "$wl rc=$?"; tail -c 200 $O/bench_$wl.err
done
