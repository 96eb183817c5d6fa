#!/usr/bin/env python3
"""Calibrate rocprofv3's FETCH_SIZE on gfx950 for the access shapes of the engine's kernels.

On the GPU box:
    cd /tmp && export TMPDIR=/tmp
    rocprofv3 --pmc FETCH_SIZE --output-format csv -d $R/gpurun_out/calib -- python3 $R/tools/pmc_calibrate.py --run $R/gpurun_out/calib_known.json
    python3 $R/tools/pmc_calibrate.py --parse $R/gpurun_out/calib $R/gpurun_out/calib_known.json --out $R/gpurun_out/pmc_calibration.json
--run reads a 1 GiB buffer (far beyond L2; every shape once) through s3grl_calibration_read and
records the bytes each launch requested; --parse divides them by the counter: factor = bytes
requested / (FETCH_SIZE x 1024).  MI355X_MICROARCH.md gives 2.0 for 16-byte-per-lane coalesced
streams; the other shapes are what this tool is for.  Commit the result as
profiles/rNN_pmc_calibration.json: bench.py applies the factor of the kernel's shape.
"""
import argparse
import csv
import ctypes as C
import json
import sys
from pathlib import Path

REPO = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(REPO))

SHAPES = [  # (name, pattern, row_bytes)
    ("stream_16B_per_lane", 0, 0), ("stream_8B_per_lane", 1, 0), ("stream_4B_per_lane", 2, 0),
    ("rows_656B", 3, 656), ("rows_2000B", 3, 2000), ("rows_12208B", 3, 12208), ("rows_256B", 3, 256),
    ("rows_64B", 3, 64),
]


def run(out):
    import torch

    from s3grl_amd import _native as N
    from s3grl_amd.engine import Engine

    eng = Engine("cuda:0")
    nbytes = 1 << 30
    buf = torch.zeros(nbytes // 4, dtype=torch.float32, device=eng.device)
    known = []
    for name, pattern, row_bytes in SHAPES:
        rows = (nbytes // max(row_bytes, 1)) if pattern == 3 else 0
        req = C.c_int64()
        N.check(N.lib().s3grl_calibration_read(eng._ctx, C.c_void_p(buf.data_ptr()), nbytes, pattern, rows,
                                               row_bytes, C.byref(req)), "s3grl_calibration_read")
        known.append({"shape": name, "pattern": pattern, "row_bytes": row_bytes, "requested_bytes": int(req.value)})
    Path(out).write_text(json.dumps(known, indent=1))
    eng.close()


def parse(d, known_path, out):
    known = json.loads(Path(known_path).read_text())
    fetch = []   # per dispatch of a calib kernel, in launch order
    for f in sorted(Path(d).rglob("*counter_collection.csv")):
        with open(f, newline="") as fh:
            rd = csv.DictReader(fh)
            cols = {c.lower(): c for c in rd.fieldnames}
            rows = [r for r in rd if "calib_" in r[cols["kernel_name"]] and r[cols["counter_name"]] == "FETCH_SIZE"]
            rows.sort(key=lambda r: int(r[cols["dispatch_id"]]))
            per = {}
            for r in rows:
                per[int(r[cols["dispatch_id"]])] = per.get(int(r[cols["dispatch_id"]]), 0.0) + float(r[cols["counter_value"]])
            fetch += [per[k] for k in sorted(per)]
    if len(fetch) != len(known):
        sys.exit(f"{len(fetch)} calibration dispatches in {d}, expected {len(known)}")
    res = {"note": "factor = bytes the loads requested / (FETCH_SIZE x 1024); 1 GiB buffer read once "
                   "(beyond L2), rows shapes at pseudo-random 16-byte-aligned places", "shapes": {}}
    for k, fs in zip(known, fetch):
        res["shapes"][k["shape"]] = {"requested_bytes": k["requested_bytes"], "FETCH_SIZE_KB": fs,
                                     "factor": k["requested_bytes"] / (fs * 1024.0) if fs else None}
    Path(out).write_text(json.dumps(res, indent=1) + "\n")
    print(json.dumps({k: round(v["factor"], 3) for k, v in res["shapes"].items()}))


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--run", metavar="KNOWN_JSON")
    ap.add_argument("--parse", nargs=2, metavar=("DIR", "KNOWN_JSON"))
    ap.add_argument("--out", default=str(REPO / "gpurun_out" / "pmc_calibration.json"))
    a = ap.parse_args()
    if a.run:
        run(a.run)
    elif a.parse:
        parse(a.parse[0], a.parse[1], a.out)
