#!/usr/bin/env python3
"""Which hardware queue each orbx kernel ran on, from a rocprofv3 --kernel-trace CSV.

usage: queue_map.py <kernel_trace.csv>
Prints, per (queue id, stream id if the trace has one, kernel), the number of dispatches, and per queue the
share of the traced interval during which at least one of its kernels ran.
"""
import csv
import sys
from collections import defaultdict

from kname import kernel_name


def main():
    rows = list(csv.DictReader(open(sys.argv[1])))
    has_stream = "Stream_Id" in rows[0]
    per = defaultdict(int)
    busy = defaultdict(list)
    for r in rows:
        name = kernel_name(r["Kernel_Name"])
        if not name.startswith("orbx::"):
            continue
        q = r["Queue_Id"]
        s = r["Stream_Id"] if has_stream else "-"
        per[(q, s, name[len("orbx::"):])] += 1
        busy[q].append((int(r["Start_Timestamp"]), int(r["End_Timestamp"])))
    for (q, s, name), n in sorted(per.items()):
        print("queue %-4s stream %-4s %-24s %6d dispatches" % (q, s, name, n))
    t0 = min(a for v in busy.values() for a, _ in v)
    t1 = max(b for v in busy.values() for _, b in v)
    for q, iv in sorted(busy.items()):
        iv.sort()
        tot, cur_a, cur_b = 0, None, None
        for a, b in iv:
            if cur_b is None or a > cur_b:
                if cur_b is not None:
                    tot += cur_b - cur_a
                cur_a, cur_b = a, b
            else:
                cur_b = max(cur_b, b)
        tot += cur_b - cur_a
        print("queue %-4s busy %.1f %% of the traced %.2f ms" % (q, 100.0 * tot / (t1 - t0), (t1 - t0) / 1e6))


if __name__ == "__main__":
    main()
