set -o pipefail
cd $GRAFT_REPO_ROOT; O=gpurun_out/r02bm; mkdir -p $O
run() { timeout -k 10 300 python bench.py --workload pubmed_pos_k5 --no-cpu-baseline --no-api --no-pmc --steps 5 > $O/bench_$1.json 2> $O/bench_$1.err; echo "$1 rc=$?"; }
run base
S3GRL_FORCE_HASH=1 S3GRL_DEBUG=1 run hash
