#!/usr/bin/env python3
"""Wave-instruction counts per extraction step of each orbx kernel, from the SQ counter passes of
tools/pmc_multi.sh (SQ_INSTS_VALU / SALU / LDS / VMEM_RD / VMEM_WR, and the LDS array's cycles: SQ_LDS_IDX_ACTIVE with
the share SQ_LDS_BANK_CONFLICT of them spent on bank conflicts).  A kernel's counts are summed
over its launches and divided by the number of steps sampled (= launches of k_octree, one per step).

usage: valu_summary.py <out.json> <pmc_counter_collection.csv> [more csv ...]
"""
import csv
import json
import os
import sys

from kname import is_orbx, kernel_name  # noqa: F401
from collections import defaultdict

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from orb_slam2_amd import source_hash  # noqa: E402

WANT = ("SQ_INSTS_VALU", "SQ_INSTS_SALU", "SQ_INSTS_LDS", "SQ_INSTS_VMEM_RD", "SQ_INSTS_VMEM_WR", "SQ_WAVES",
        "SQ_LDS_BANK_CONFLICT", "SQ_LDS_IDX_ACTIVE")


def main():
    out, paths = sys.argv[1], sys.argv[2:]
    res = defaultdict(dict)
    for path in paths:
        acc, ids = defaultdict(float), defaultdict(set)
        with open(path) as f:
            for row in csv.DictReader(f):
                name = kernel_name(row["Kernel_Name"])
                if not name.startswith("orbx::") or row["Counter_Name"] not in WANT:
                    continue
                name = name[len("orbx::"):]
                acc[(name, row["Counter_Name"])] += float(row["Counter_Value"])
                ids[name].add(row["Dispatch_Id"])
        steps = len(ids.get("k_octree", ())) or 1
        for (name, ctr), v in acc.items():
            res[name][ctr] = v / steps
    res["_note"] = ("wave instructions (and LDS array cycles, in units of 4 clocks) per extraction step of the profiled "
                    "`bench.py --steps 2 --warmup 1` command (512 frames for the default configuration), summed over a "
                    "kernel's launches of the step; one rocprofv3 --pmc pass per counter group")
    res["_source_sha256"] = source_hash()
    with open(out, "w") as f:
        json.dump(res, f, indent=1, sort_keys=True)
    for k in sorted(res):
        if not k.startswith("_"):
            print(k, {c: "%.3g" % v for c, v in sorted(res[k].items())})


if __name__ == "__main__":
    main()
