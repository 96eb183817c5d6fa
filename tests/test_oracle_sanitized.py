"""the plain-C restatement (oracle/s3grl_oracle_c.c) built with
-fsanitize=address,undefined (`make -C oracle asan`) and run over the golden cases in a child
process (the sanitizer runtime must be the first library of the process: LD_PRELOAD).  Any
out-of-bounds access, use-after-free or undefined arithmetic in the checker aborts the child.
Host only — GPU sanitizers are not available on the pool."""
import os
import subprocess
import sys
from pathlib import Path

import pytest

REPO = Path(__file__).resolve().parent.parent

CHILD = r"""
import numpy as np
import sys
sys.path.insert(0, %(repo)r)
sys.path.insert(0, %(repo)r + "/tests")
from conftest import DIFFUSION_NAMES, EXTRACT_NAMES, csr_from_undirected, load_diffusion, load_extract
from oracle import c_oracle

def ragged(blob, key, i):
    off = blob[key + "_off"]
    return blob[key][off[i]:off[i + 1]]

checked = 0
for name in EXTRACT_NAMES:
    g = load_extract(name)
    A = csr_from_undirected(int(g["num_nodes"]), g["edges"])
    for h in g["hops"]:
        node_ptr, nodes, dists = c_oracle.extract(g["links"].T, int(h), A)
        for li in range(len(g["links"])):
            mine = nodes[node_ptr[li]:node_ptr[li + 1]]
            md = dists[node_ptr[li]:node_ptr[li + 1]]
            order = np.lexsort((mine, md))
            assert np.array_equal(mine[order], ragged(g, f"h{h}_nodes", li)), (name, h, li)
            assert np.array_equal(md[order], ragged(g, f"h{h}_dists", li)), (name, h, li)
            checked += 1
for name in DIFFUSION_NAMES:
    g = load_diffusion(name)
    n, K, h = int(g["num_nodes"]), int(g["K"]), int(g["num_hops"])
    A = csr_from_undirected(n, g["edges"])
    X32 = g["X"].astype(np.float32)
    for tag, plus in (("pos", False), ("plus", True)):
        for threads in (1, 3):
            rows, row_ptr, row_nodes, _ = c_oracle.pos_rows(g["links"].T, h, A, X32, K, plus=plus, threads=threads)
            assert np.array_equal(row_ptr, g[f"{tag}_row_ptr"])
            assert np.array_equal(row_nodes, g[f"{tag}_rows_global"])
            np.testing.assert_allclose(rows, g[f"{tag}_rows"], rtol=1e-6, atol=1e-7)
            checked += 1
print("SANITIZED_OK", checked)
"""


def _runtime(name):
    out = subprocess.run(["gcc", f"-print-file-name={name}"], capture_output=True, text=True).stdout.strip()
    return out if out and os.path.isabs(out) and os.path.exists(out) else None


def test_c_restatement_under_asan_ubsan():
    asan = _runtime("libasan.so")
    if asan is None:
        pytest.skip("gcc has no libasan here")
    r = subprocess.run(["make", "-s", "-C", str(REPO / "oracle"), "asan"], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-2000:]
    lib = REPO / "oracle" / "_build" / "libs3grl_oracle_c_asan.so"
    env = dict(os.environ, LD_PRELOAD=asan, S3GRL_ORACLE_C_LIB=str(lib), OMP_NUM_THREADS="3",
               # CPython itself is not leak-clean; everything else stays fatal
               ASAN_OPTIONS="detect_leaks=0:abort_on_error=1:halt_on_error=1",
               UBSAN_OPTIONS="halt_on_error=1:print_stacktrace=1")
    r = subprocess.run([sys.executable, "-c", CHILD % {"repo": str(REPO)}], env=env, capture_output=True,
                       text=True, timeout=600)
    assert r.returncode == 0 and "SANITIZED_OK" in r.stdout, (r.stdout[-1500:], r.stderr[-3000:])
    assert "ERROR: AddressSanitizer" not in r.stderr and "runtime error" not in r.stderr, r.stderr[-3000:]
