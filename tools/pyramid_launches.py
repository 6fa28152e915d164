#!/usr/bin/env python3
"""Durations of, and gaps between, the pyramid launches of one batch, from a rocprofv3 --kernel-trace CSV
of `ORBX_NO_OVERLAP=1 bench.py` (stages one after the other: every kernel alone on the GPU).

usage: pyramid_launches.py <kernel_trace.csv>
"""
import csv
import sys

from kname import kernel_name


def main():
    rows = []
    for r in csv.DictReader(open(sys.argv[1])):
        name = kernel_name(r["Kernel_Name"])
        if name.startswith("orbx::"):
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), name[len("orbx::"):].split("<")[0]))
    rows.sort()
    # the last complete run of consecutive k_resize_rows launches
    runs, cur = [], []
    for a, b, n in rows:
        if n == "k_resize_rows":
            cur.append((a, b))
        elif cur:
            runs.append(cur)
            cur = []
    run = runs[-2]
    t0 = run[0][0]
    prev = None
    for i, (a, b) in enumerate(run):
        print("level %d  start %8.1f us  duration %7.1f us  gap before %5.1f us" % (i + 1, (a - t0) / 1e3, (b - a) / 1e3,
                                                                               0.0 if prev is None else (a - prev) / 1e3))
        prev = b
    print("span %.1f us, sum of durations %.1f us" % ((run[-1][1] - t0) / 1e3, sum(b - a for a, b in run) / 1e3))


if __name__ == "__main__":
    main()
