#!/usr/bin/env python3
"""Sum rocprofv3 counter rows per (kernel, counter); print per-dispatch averages."""
import csv
import sys

from kname import is_orbx, kernel_name  # noqa: F401
from collections import defaultdict

acc = defaultdict(float)
calls = defaultdict(set)
with open(sys.argv[1]) as f:
    for row in csv.DictReader(f):
        name = kernel_name(row["Kernel_Name"])
        if not name.startswith("orbx::"):
            continue
        acc[(name, row["Counter_Name"])] += float(row["Counter_Value"])
        calls[name].add(row["Dispatch_Id"])
for (name, ctr), v in sorted(acc.items()):
    n = len(calls[name])
    print("%-28s %-26s total %.4g  per-dispatch %.4g  (%d dispatches)" % (name, ctr, v, v / n, n))
