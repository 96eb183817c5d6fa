import os, sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent)); sys.path.insert(0, str(Path(__file__).resolve().parent.parent / "tests"))
import numpy as np, torch
from conftest import csr_from_undirected, load_extract
from s3grl_amd.engine import Engine
eng = Engine("cuda:0")
name = sys.argv[1] if len(sys.argv) > 1 else "probe5"
g = load_extract(name); n = int(g["num_nodes"]); A = csr_from_undirected(n, g["edges"])
links = eng.links(g["links"].T)
X = np.random.default_rng(17).standard_normal((n, 19)).astype(np.float32)
f = eng.features(X)
for K in (1, 2, 3):
    for mode in ("pos", "pos_plus"):
        outs = {}
        for tag, env in (("bitmap", {}), ("hash", {"S3GRL_FORCE_HASH": "1", "S3GRL_NO_ONEHOP": "1"}),
                         ("onehop", {"S3GRL_FORCE_HASH": "1", "S3GRL_FORCE_ONEHOP": "1"}),
                         ("onehop_hbm", {"S3GRL_FORCE_HASH": "1", "S3GRL_FORCE_ONEHOP": "1", "S3GRL_FORCE_BM_HBM": "1"})):
            for k in ("S3GRL_FORCE_HASH", "S3GRL_NO_ONEHOP", "S3GRL_FORCE_ONEHOP", "S3GRL_FORCE_BM_HBM"):
                os.environ.pop(k, None)
            os.environ.update(env)
            os.environ["S3GRL_DEBUG"] = "1"
            G = eng.graph(A)
            p = eng.plan(G, links, mode=mode, num_hops=1, sign_k=K, full_stats=True)
            r = p.run(f)
            st = {k: p.stats[k] for k in ("total_nodes", "total_volume", "total_sub_edges", "total_support", "total_rows")}
            outs[tag] = (st, r.cpu().numpy(), p.row_nodes().cpu().numpy())
            p.close(); G.close()
        base = outs["bitmap"]
        for tag, (st, r, rn) in outs.items():
            err = float(np.abs(r - base[1]).max()) if r.shape == base[1].shape else -1
            print(K, mode, tag, st, "maxabs diff vs bitmap %.2e" % err, "rows eq", np.array_equal(rn, base[2]))
