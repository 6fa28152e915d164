#!/usr/bin/env python3
"""When each kernel of a steady-state extraction step runs, from a rocprofv3 --kernel-trace CSV of bench.py.

usage: step_timeline.py <kernel_trace.csv> [steps to skip from the end = 20]
Takes one FAST launch well inside the timed region as the origin and prints, for every orbx kernel launch that
overlaps the window from that launch's start to the next FAST launch's start, its start and end in ms relative
to the origin (the pyramid's launches of one batch as one line).
"""
import csv
import sys

from kname import kernel_name


def main():
    rows = []
    for r in csv.DictReader(open(sys.argv[1])):
        name = kernel_name(r["Kernel_Name"])
        if name.startswith("orbx::"):
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), name[len("orbx::"):].split("<")[0], r["Queue_Id"]))
    rows.sort()
    fast = [r for r in rows if r[2] == "k_fast_tiles"]
    skip = int(sys.argv[2]) if len(sys.argv) > 2 else 20
    i = len(fast) - skip
    t0, t1 = fast[i][0], fast[i + 1][0]
    print("window: one FAST launch start to the next = %.3f ms" % ((t1 - t0) / 1e6))
    groups = []
    for a, b, n, q in rows:
        if b < t0 or a > t1:
            continue
        if groups and groups[-1][2] == n and n == "k_resize_rows" and a - groups[-1][1] < 50000:
            groups[-1][1] = b
            groups[-1][4] += 1
        else:
            groups.append([a, b, n, q, 1])
    for a, b, n, q, k in groups:
        print("%-20s queue %s  %7.3f .. %7.3f ms  (%.3f ms, %d launch%s)" % (n, q, (a - t0) / 1e6, (b - t0) / 1e6, (b - a) / 1e6, k,
                                                                            "" if k == 1 else "es"))


if __name__ == "__main__":
    main()
