#!/usr/bin/env python3
"""HBM bytes per launch of each orbx kernel from a FETCH_SIZE pass and a WRITE_SIZE pass.

Units and corrections: both counters are in KiB.  On gfx950 FETCH_SIZE tallies the 128-byte
memory-side read requests at 64 bytes, so it reports half of the bytes read; measured here with
tools/ubench/fetch_calib.hip on a 1 GiB buffer: factor 2.000 for coalesced 16-byte and 4-byte
loads (2.06 / 2.25 raw-to-true when the same loads are misaligned by 3 / 1 bytes, because they
touch one more line per wave).  WRITE_SIZE is exact for 16-byte and 4-byte stores.
    traffic = 2 * FETCH_SIZE * 1024 + WRITE_SIZE * 1024      bytes per launch

usage: traffic_summary.py <dir with rd/ and wr/> <out.json>
"""
import csv
import json
import os
import sys

from kname import is_orbx, kernel_name  # noqa: F401
from collections import defaultdict

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from orb_slam2_amd import source_hash  # noqa: E402


def per_launch(path):
    acc, ids = defaultdict(float), defaultdict(set)
    with open(path) as f:
        for row in csv.DictReader(f):
            name = kernel_name(row["Kernel_Name"])
            if not name.startswith("orbx::"):
                continue
            acc[name] += float(row["Counter_Value"])
            ids[name].add(row["Dispatch_Id"])
    return {k: (acc[k] / len(ids[k]), len(ids[k])) for k in acc}


def main():
    d, out = sys.argv[1], sys.argv[2]
    rd = per_launch(d + "/rd/pmc_counter_collection.csv")
    wr = per_launch(d + "/wr/pmc_counter_collection.csv")
    # one k_octree launch per extraction step: its launch count is the number of steps sampled
    steps = rd["orbx::k_octree"][1]
    res = {"_note": "bytes per extraction step of the profiled command (512 frames for the default configuration) summed over a kernel's launches of that step, averaged over "
                    "the %d steps of `bench.py --steps 2 --warmup 1`; fetch = 2 * FETCH_SIZE KiB (gfx950 correction, "
                    "calibrated), write = WRITE_SIZE KiB" % steps}
    for k in sorted(rd):
        fetch = 2.0 * rd[k][0] * rd[k][1] * 1024.0 / steps
        write = wr.get(k, (0.0, 0))[0] * wr.get(k, (0.0, 0))[1] * 1024.0 / steps
        res[k.replace("orbx::", "")] = {"launches_sampled": rd[k][1], "fetch_bytes": fetch, "write_bytes": write,
                                        "traffic_bytes": fetch + write}
    res["_source_sha256"] = source_hash()
    with open(out, "w") as f:
        json.dump(res, f, indent=1)
    for k, v in res.items():
        if not k.startswith("_"):
            print("%-20s launches %3d  fetch %8.1f MB  write %8.1f MB" % (k, v["launches_sampled"], v["fetch_bytes"] / 1e6,
                                                                        v["write_bytes"] / 1e6))


if __name__ == "__main__":
    main()
