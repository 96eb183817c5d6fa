#!/bin/bash
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r02h
mkdir -p $O
cd $R
timeout -k 10 300 python bench.py --gpus 4 --steps 3 --warmup 1 --verify > $O/bench_k3_g4.json 2> $O/bench_k3_g4.err; echo "g4 rc=$?"; tail -c 300 $O/bench_k3_g4.err
timeout -k 10 300 python bench.py --gpus 2 --steps 3 --warmup 1 --verify --workload cora_posplus_k3 > $O/bench_cora_g2.json 2> $O/bench_cora_g2.err; echo "cora g2 rc=$?"; tail -c 300 $O/bench_cora_g2.err
timeout -k 10 300 python bench.py --gpus 2 --steps 3 --warmup 1 --verify --no-allgather > $O/bench_k3_g2_noag.json 2> $O/bench_k3_g2_noag.err; echo "g2 noag rc=$?"; tail -c 300 $O/bench_k3_g2_noag.err
timeout -k 10 300 python bench.py --gpus 3 --steps 3 --warmup 1 --verify --workload pubmed_sop_k3 --chunks 2 > $O/bench_sop_g3.json 2> $O/bench_sop_g3.err; echo "sop g3 rc=$?"; tail -c 300 $O/bench_sop_g3.err
