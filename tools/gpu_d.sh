#!/bin/bash
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r02d
mkdir -p $O
cd $R
timeout -k 10 400 python -m pytest tests/test_gpu_onehop.py tests/test_gpu_parity.py -x -q -k "onehop or collab or power_law or sop" > $O/pytest_sel.log 2>&1; rc=$?; echo "sel pytest rc=$rc"; tail -5 $O/pytest_sel.log
[ $rc -ne 0 ] && exit $rc
S3GRL_DEBUG=1 timeout -k 10 300 python bench.py --workload collab_pos_k3 --no-cpu-baseline --no-api --steps 5 > $O/bench_collab.json 2> $O/bench_collab.err; echo "collab rc=$?"; grep "s3grl\]" $O/bench_collab.err | head -2
timeout -k 10 300 python bench.py --workload pubmed_sop_k3 --no-cpu-baseline > $O/bench_sop.json 2> $O/bench_sop.err; echo "sop rc=$?"
timeout -k 10 300 python bench.py --no-cpu-baseline > $O/bench_k3.json 2> $O/bench_k3.err; echo "k3 rc=$?"
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace_collab -- python3 $R/bench.py --workload collab_pos_k3 --no-cpu-baseline --no-api --steps 3 --warmup 1 > $O/trace_collab.log 2>&1; echo "trace rc=$?"
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace_sop -- python3 $R/bench.py --workload pubmed_sop_k3 --no-cpu-baseline --no-api --steps 5 --warmup 1 > $O/trace_sop.log 2>&1; echo "trace sop rc=$?"
timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/calib -- python3 $R/tools/pmc_calibrate.py --run $O/calib_known.json > $O/calib.log 2>&1; echo "calib rc=$?"
python3 $R/tools/pmc_calibrate.py --parse $O/calib $O/calib_known.json --out $O/pmc_calibration.json
find $O -name "*kernel_trace.csv" -delete; find $O -name "*counter_collection.csv" -delete
for d in trace_collab trace_sop; do f=$(find $O/$d -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f $O/${d}_kernel_stats.csv && head -12 $f; done
