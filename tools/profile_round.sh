#!/bin/bash
# Round profile (run ON THE GPU BOX through gpurun, from the repo root):
#   gpurun --timeout 1200 -- 'bash tools/profile_round.sh r02'
# For every workload below: a rocprofv3 --kernel-trace --stats summary of `bench.py`, and the bench
# line itself with the counters collected in the same run (bench.py --collect-pmc: one child
# rocprofv3 --pmc pass per counter group, never together with a trace; the program itself after `--`).
# Plus the FETCH_SIZE calibration (tools/pmc_calibrate.py).  Everything lands in
# gpurun_out/<tag>/; copy what is to be judged into profiles/ afterwards:
#   for f in gpurun_out/r02/*.{json,csv}; do cp $f profiles/r02_$(basename $f); done
set -o pipefail
TAG=${1:-round}
PART=${2:-all}      # a: traces + calibration + the counter runs of the PubMed PoS / SoP lines; b: the other bench lines
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/$TAG
[ "$PART" = b ] || rm -rf "$O"; mkdir -p "$O"
# build BEFORE any profiler: under rocprofv3 a compiler child would inherit the tool's preload, i.e.
# be a GPU-initialised process that execs (bench.py / build() refuse to compile there)
(cd $R && python3 -c 'import __graft_entry__ as g; g.build()') > $O/build.log 2>&1 || { echo "build failed"; tail -20 $O/build.log; exit 1; }
cd /tmp && export TMPDIR=/tmp
rc=0
if [ "$PART" != b ]; then
for wl in pubmed_pos_k3 pubmed_pos_k3_dense pubmed_sop_k3 pubmed_pos_k5 collab_pos_k3; do
  timeout -k 10 240 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace_$wl -- python3 $R/bench.py --workload $wl --steps 10 --warmup 2 --no-cpu-baseline --no-api --no-pmc > $O/trace_$wl.log 2>&1 || rc=1
  f=$(find $O/trace_$wl -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f $O/kernel_stats_$wl.csv
  rm -rf $O/trace_$wl
done
timeout -k 10 240 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/calib -- python3 $R/tools/pmc_calibrate.py --run $O/calib_known.json > $O/calib.log 2>&1 || rc=1
python3 $R/tools/pmc_calibrate.py --parse $O/calib $O/calib_known.json --out $O/pmc_calibration.json || rc=1
rm -rf $O/calib
cd $R
for wl in pubmed_pos_k3 pubmed_sop_k3; do
  timeout -k 10 420 python3 bench.py --workload $wl --collect-pmc > $O/bench_$wl.json 2> $O/bench_$wl.err || rc=1
done
fi
if [ "$PART" != a ]; then
cd $R
for wl in pubmed_pos_k3_dense pubmed_pos_k5 collab_pos_k3 cora_posplus_k3 cora_posplus_k3_real usair_pos_k2 pubmed_sop_k3_2hop; do
  timeout -k 10 300 python3 bench.py --workload $wl --collect-pmc > $O/bench_$wl.json 2> $O/bench_$wl.err || rc=1
done
fi
echo "profile_round rc=$rc"; ls $O
exit $rc
