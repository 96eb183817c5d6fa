set -o pipefail
cd $GRAFT_REPO_ROOT; O=gpurun_out/r02j; mkdir -p $O
timeout -k 10 300 python -m pytest tests/test_gpu_parity.py -m gpu -q -k "feature_widths" > $O/pytest2.log 2>&1; rc=$?; echo "pytest rc=$rc"; tail -4 $O/pytest2.log
for wl in pubmed_pos_k3 pubmed_pos_k5 cora_posplus_k3 usair_pos_k2 collab_pos_k3; do
  timeout -k 10 300 python bench.py --workload $wl --no-cpu-baseline --no-api --steps 10 > $O/bench_$wl.json 2> $O/bench_$wl.err; echo "$wl rc=$?"
done
