#!/usr/bin/env python3
"""Test AUC of the paper's Cora entry on the engine + harness for a few epoch budgets (GPU box): sizes the
budget of tests/test_gpu_harness.py::test_cora_real_features_paper_config_auc."""
import sys
import time
from pathlib import Path

REPO = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(REPO))
sys.path.insert(0, str(REPO / "tests"))

if __name__ == "__main__":
    import test_gpu_harness as t

    for epochs in [int(a) for a in sys.argv[1:]] or [4, 8, 12]:
        for seed in (1, 2, 3):
            t0 = time.time()
            auc = t._cora_paper_run(seed, epochs)
            print(f"epochs {epochs:3d} seed {seed}: test AUC {auc:.4f}  ({time.time() - t0:.1f} s)", flush=True)
