set -o pipefail
cd $GRAFT_REPO_ROOT; O=gpurun_out/r02big; mkdir -p $O
timeout -k 10 700 python -m pytest tests/test_gpu_onehop.py tests/test_gpu_parity.py -m gpu -q -x -k "onehop or collab or power_law or many_common or sign_k" > $O/pytest.log 2>&1; rc=$?; echo "pytest rc=$rc"; tail -3 $O/pytest.log
run() { timeout -k 10 300 python bench.py --workload collab_pos_k3 --no-cpu-baseline --no-api --no-pmc --steps 5 > $O/bench_$1.json 2> $O/bench_$1.err; echo "$1 rc=$?"; }
run split
S3GRL_BIG_ONE_LAUNCH=1 run one
