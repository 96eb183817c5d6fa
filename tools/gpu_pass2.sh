set -o pipefail
cd $GRAFT_REPO_ROOT; O=gpurun_out/r02m; mkdir -p $O
run() { # name workload
  timeout -k 10 300 python bench.py --workload $2 --no-cpu-baseline --no-api --steps 10 > $O/bench_$2_$1.json 2> $O/bench_$2_$1.err; echo "$2 $1 rc=$?"
}
for wl in pubmed_pos_k3 pubmed_pos_k5 cora_posplus_k3; do
run auto $wl
S3GRL_BOUNDS=3072,6144,12288,24576,65536,163840 S3GRL_T_CLASS0=64 S3GRL_T_CLASS1=128 run b3k_64_128 $wl
S3GRL_BOUNDS=3072,6144,12288,24576,65536,163840 S3GRL_T_CLASS0=128 S3GRL_T_CLASS1=128 run b3k_128_128 $wl
S3GRL_BOUNDS=4096,8192,16384,32768,65536,163840 S3GRL_T_CLASS0=128 S3GRL_T_CLASS1=128 run b4k_128_128 $wl
S3GRL_BOUNDS=4096,8192,16384,32768,65536,163840 S3GRL_T_CLASS0=128 S3GRL_T_CLASS1=256 run b4k_128_256 $wl
S3GRL_BOUNDS=8192,12288,24576,49152,98304,163840 S3GRL_T_CLASS0=128 run b8k_128 $wl
done
