set -o pipefail
cd $GRAFT_REPO_ROOT; O=gpurun_out/r02q; mkdir -p $O
run() { # name workload
  timeout -k 10 300 python bench.py --workload $2 --no-cpu-baseline --no-api --steps 10 > $O/bench_$2_$1.json 2> $O/bench_$2_$1.err; echo "$2 $1 rc=$?"
}
for wl in pubmed_pos_k3 pubmed_pos_k5 cora_posplus_k3; do
run base $wl
S3GRL_EXPERIMENT_RELABEL=desc run desc $wl
S3GRL_EXPERIMENT_RELABEL=asc run asc $wl
done
