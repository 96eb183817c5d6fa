#!/bin/bash
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r02c
mkdir -p $O
cd $R
timeout -k 10 400 python -m pytest tests/test_gpu_onehop.py -x -q > $O/pytest_onehop.log 2>&1; rc=$?; echo "onehop pytest rc=$rc"; tail -12 $O/pytest_onehop.log
[ $rc -ne 0 ] && exit $rc
timeout -k 10 300 python bench.py --workload collab_pos_k3 --no-cpu-baseline --no-api --steps 5 > $O/bench_collab.json 2> $O/bench_collab.err; echo "collab rc=$?"; tail -c 300 $O/bench_collab.err
S3GRL_DEBUG=1 timeout -k 10 200 python bench.py --workload collab_pos_k3 --no-cpu-baseline --no-api --steps 1 --warmup 0 > /dev/null 2> $O/collab_classes.err; grep s3grl $O/collab_classes.err | head -3
timeout -k 10 300 python tools/d2h_probe.py > $O/d2h_probe.log 2>&1; echo "probe rc=$?"; head -12 $O/d2h_probe.log
timeout -k 10 600 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; rc=$?; echo "pytest rc=$rc"; tail -5 $O/pytest.log
