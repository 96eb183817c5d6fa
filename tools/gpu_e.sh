#!/bin/bash
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r02e
mkdir -p $O
cd $R
timeout -k 10 400 python -m pytest tests/test_gpu_onehop.py tests/test_gpu_parity.py -x -q -k "onehop or collab or power_law" > $O/pytest_sel.log 2>&1; rc=$?; echo "sel pytest rc=$rc"; tail -5 $O/pytest_sel.log
[ $rc -ne 0 ] && exit $rc
timeout -k 10 300 python bench.py --workload collab_pos_k3 --no-cpu-baseline --no-api --steps 5 > $O/bench_collab.json 2> $O/bench_collab.err; echo "collab rc=$?"
cd /tmp && export TMPDIR=/tmp
S3GRL_SERIAL_CLASSES=1 timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace_collab -- python3 $R/bench.py --workload collab_pos_k3 --no-cpu-baseline --no-api --steps 3 --warmup 1 > $O/trace_collab.log 2>&1; echo "trace rc=$?"
find $O -name "*kernel_trace.csv" > $O/trace_files.txt
python3 - <<PY
import csv,glob,collections
f=open("$O/trace_files.txt").read().split()
agg=collections.defaultdict(list)
for p in f:
    for r in csv.DictReader(open(p)):
        n=r["Kernel_Name"]
        if "link_full" in n or "link_kernel" in n or "count1" in n:
            key=n.split("(")[0][-40:]+" grid="+r["Grid_Size_X"]+" lds="+r.get("LDS_Block_Size","?")
            agg[key].append((int(r["End_Timestamp"])-int(r["Start_Timestamp"]))/1e6)
for k,v in sorted(agg.items(), key=lambda kv:-sum(kv[1])):
    print("%-90s calls %d avg %.2f ms"%(k,len(v),sum(v)/len(v)))
PY
find $O -name "*kernel_trace.csv" -delete
