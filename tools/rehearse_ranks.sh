# N ranks sharing ONE GPU over gloo, launched the way the driver launches a multi-GPU run
# (python -m torch.distributed.run ... bench.py --gpus N), with --verify: the sharded result must be
# bit-equal to the unsharded one.  GPU box:  gpurun -- 'bash tools/rehearse_ranks.sh r03s 2 pubmed_pos_k3'
# At most 5 ranks on a gpurun box: the launcher counts towards its limit of 6 processes per GPU.
# BENCH_ARGS="--replicate-fraction 0.3" adds arguments to the bench line.
set -o pipefail
cd $GRAFT_REPO_ROOT; TAG=$1; N=${2:-2}; shift 2
O=gpurun_out/$TAG; mkdir -p $O
python3 -c 'import __graft_entry__ as g; g.build()' > $O/build.log 2>&1 || { echo "build failed"; exit 1; }
for wl in "$@"; do
  port=$((29500 + RANDOM % 500))
  S3GRL_BENCH_BACKEND=gloo timeout -k 10 400 python -m torch.distributed.run --nnodes=1 --nproc-per-node $N \
      --master-addr 127.0.0.1 --master-port $port bench.py --gpus $N --steps 3 --warmup 1 --verify --workload $wl $BENCH_ARGS \
      > $O/ranks${N}_$wl.json 2> $O/ranks${N}_$wl.err
  rc=$?
  python3 - <<PY
import json
try:
    d = json.loads([l for l in open("$O/ranks${N}_$wl.json").read().splitlines() if l.startswith("{")][-1])
    m = d["multi_gpu"]
    print("$wl N=$N rc=$rc", "verified:", m["verified_bit_equal_to_unsharded"], "links/rank", m["links_per_rank"],
          "folded", m.get("folded_links_total"), "imbalance %.3f" % (m["imbalance_max_over_mean"] or 0))
except Exception as e:
    print("$wl N=$N rc=$rc: no line:", e)
PY
done
