#!/bin/bash
# Round profile of the default bench (run ON THE GPU BOX through gpurun, from the repo root):
#   gpurun -- 'bash tools/profile_round.sh v8'
# leaves gpurun_out/<tag>/{trace,fetch,write,tcc,sq}/ + bench.json; afterwards, in the build
# container:  python tools/pmc_collect.py --tag r01_<tag> --build "..." --bench-json gpurun_out/<tag>/bench.json \
#                 gpurun_out/<tag>/{fetch,write,tcc,sq}
# Counter passes are separate runs without any trace (MI355X_MICROARCH.md, HBM section); the
# program follows `--` directly (no env/bash hop under rocprofv3).
set -o pipefail
TAG=${1:-round}
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/$TAG
rm -rf "$O"; mkdir -p "$O"
cd /tmp && export TMPDIR=/tmp
ONE="python3 $R/bench.py --steps 1 --warmup 0 --no-cpu-baseline"
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -- python3 $R/bench.py --steps 10 --warmup 2 --no-cpu-baseline > $O/trace.log 2>&1 &&
timeout -k 10 200 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/fetch -- $ONE > $O/fetch.log 2>&1 &&
timeout -k 10 200 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/write -- $ONE > $O/write.log 2>&1 &&
timeout -k 10 200 rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum --output-format csv -d $O/tcc -- $ONE > $O/tcc.log 2>&1 &&
timeout -k 10 200 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SMEM SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_WAVES --output-format csv -d $O/sq -- $ONE > $O/sq.log 2>&1 &&
cd $R && timeout -k 10 300 python3 bench.py > $O/bench.json 2> $O/bench.err
rc=$?
find $O -name "*kernel_trace.csv" -delete
echo "profile_round rc=$rc"; tail -c 400 $O/bench.json
exit $rc
