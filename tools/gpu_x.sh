set -o pipefail
cd $GRAFT_REPO_ROOT; O=gpurun_out/r02sop; mkdir -p $O
timeout -k 10 600 python -m pytest tests -m gpu -q -x -k "sop or SoP" > $O/pytest.log 2>&1; rc=$?; echo "pytest rc=$rc"; tail -3 $O/pytest.log
run() { timeout -k 10 300 python bench.py --workload pubmed_sop_k3 --no-cpu-baseline --no-api $2 --steps 10 > $O/bench_$1.json 2> $O/bench_$1.err; echo "$1 rc=$?"; }
run sorted ""
S3GRL_SOP_UNSORTED=1 run unsorted "--no-pmc"
