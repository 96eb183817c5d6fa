set -o pipefail
cd $GRAFT_REPO_ROOT; O=gpurun_out/r02j; mkdir -p $O
timeout -k 10 700 python -m pytest tests -m gpu -q > $O/pytest.log 2>&1; rc=$?; echo "pytest rc=$rc"; tail -4 $O/pytest.log
[ $rc -ne 0 ] && exit $rc
for wl in pubmed_pos_k3 pubmed_pos_k5 cora_posplus_k3 usair_pos_k2; do
  timeout -k 10 300 python bench.py --workload $wl --no-cpu-baseline --no-api --no-pmc --steps 10 > $O/bench_$wl.json 2> $O/bench_$wl.err; echo "$wl rc=$?"
done
