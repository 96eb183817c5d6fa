set -o pipefail
cd $GRAFT_REPO_ROOT; O=gpurun_out/r02z6; mkdir -p $O
S3GRL_DEBUG=1 timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "big_graph" > $O/pytest.log 2>&1; rc=$?; echo "pytest rc=$rc"; tail -15 $O/pytest.log
