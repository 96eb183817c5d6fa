set -o pipefail
cd $GRAFT_REPO_ROOT; O=gpurun_out/r02s; mkdir -p $O
timeout -k 10 700 python -m pytest tests -m gpu -q > $O/pytest.log 2>&1; rc=$?; echo "pytest rc=$rc"; tail -5 $O/pytest.log
for wl in collab_pos_k3 pubmed_pos_k3; do
  timeout -k 10 400 python bench.py --workload $wl --no-cpu-baseline --no-api --steps 5 > $O/bench_${wl}_new.json 2> $O/bench_${wl}_new.err; echo "$wl rc=$?"
done
