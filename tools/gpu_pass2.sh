set -o pipefail
cd $GRAFT_REPO_ROOT; O=gpurun_out/r02k; mkdir -p $O; rm -f $O/*
for wl in pubmed_pos_k3 pubmed_pos_k5; do
for v in auto 0 6144 12288 24576 49152; do
  if [ $v = auto ]; then unset S3GRL_DM_MIN_NEED; else export S3GRL_DM_MIN_NEED=$v; fi
  S3GRL_DEBUG=1 timeout -k 10 300 python bench.py --workload $wl --no-cpu-baseline --no-api --steps 10 > $O/bench_${wl}_$v.json 2> $O/bench_${wl}_$v.err; echo "$wl $v rc=$?"
done; done
