set -o pipefail
cd $GRAFT_REPO_ROOT; O=gpurun_out/r02z5; mkdir -p $O
run() { timeout -k 10 300 python bench.py --workload $2 --no-cpu-baseline --no-api --no-pmc --steps 10 > $O/bench_$2_$1.json 2> $O/bench_$2_$1.err; echo "$2 $1 rc=$?"; }
for wl in pubmed_pos_k3 pubmed_pos_k5; do
run base $wl
S3GRL_T_CLASS3=256 run c3_256 $wl
S3GRL_T_CLASS4=512 run c4_512 $wl
S3GRL_T_CLASS1=128 run c1_128 $wl
S3GRL_T_CLASS3=256 S3GRL_T_CLASS4=512 run c3_256_c4_512 $wl
done
