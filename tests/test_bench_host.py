"""Host-side behaviour of bench.py that needs no GPU: the launcher of `--gpus N` must not outlive a dead
rank, and the RCCL summary parser must take whatever the log holds."""
import subprocess
import sys
import time
from pathlib import Path

REPO = Path(__file__).resolve().parent.parent


def test_a_dead_rank_ends_the_run_with_a_reason():
    """No GPU here: every rank dies at `torch.cuda.set_device`.  The parent reports the first failure on one
    line, stops the others and exits non-zero — it does not wait for a rendezvous that cannot happen."""
    t0 = time.time()
    r = subprocess.run([sys.executable, str(REPO / "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0",
                        "--no-cpu-baseline", "--no-api", "--no-pmc"], capture_output=True, text=True, timeout=300,
                       env={**__import__("os").environ, "S3GRL_BENCH_BACKEND": "gloo", "S3GRL_BENCH_RENDEZVOUS_S": "20",
                            "HIP_VISIBLE_DEVICES": "", "CUDA_VISIBLE_DEVICES": ""})
    assert r.returncode != 0
    lines = [ln for ln in r.stderr.splitlines() if "bench.py --gpus 2 FAILED:" in ln]
    assert len(lines) == 1 and "rank" in lines[0], r.stderr[-2000:]
    assert r.stdout.strip() == ""                      # no JSON line from a failed run
    assert time.time() - t0 < 240


def test_rccl_summary_never_raises(tmp_path):
    sys.path.insert(0, str(REPO))
    import bench

    log = tmp_path / "rccl.log"
    log.write_text("h:1:1 [0] NCCL INFO NCCL version 2.22.3 / RCCL\n"
                   "h:1:1 [0] NCCL INFO Channel 00/16 :    0   1\n"
                   "h:1:1 [0] NCCL INFO Channel 00 : 0[0] -> 1[1] via P2P/IPC\n"
                   "h:1:1 [0] NCCL INFO Connected all rings\n"
                   "garbage line without the marker\n")
    s = bench.rccl_summary(log)
    assert s["channels"] == 16 and s["connections_via"] == {"P2P/IPC": 1} and "rings_connected" in s
    assert "error" in bench.rccl_summary(tmp_path / "missing.log")
    (tmp_path / "empty.log").write_text("")
    assert bench.rccl_summary(tmp_path / "empty.log")["info_lines"] == 0
