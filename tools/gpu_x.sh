set -o pipefail
cd $GRAFT_REPO_ROOT; O=gpurun_out/r02cn; mkdir -p $O
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "many_common" > $O/pytest.log 2>&1; rc=$?; echo "pytest rc=$rc"; tail -15 $O/pytest.log
