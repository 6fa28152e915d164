#!/bin/bash
# Kernel trace of tools/dbg/ctx_order_check.py (contexts created one after the other in one process) and the hardware
# queue each kernel ran on, per context (tools/queue_map.py over slices of the trace would mix them: this prints the
# queue of each kernel name in launch order, compressed).
set -e
OUT=$PWD/gpurun_out
D=/tmp/orbx_co
rm -rf $D && mkdir -p $D
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $D -- python3 $GRAFT_REPO_ROOT/tools/dbg/ctx_order_check.py c3 c3 c3 > $OUT/ctx_order.txt 2> $OUT/ctx_order.err
CSV=$(find $D -name '*kernel_trace.csv' | head -1)
python3 - "$CSV" >> $OUT/ctx_order.txt <<'PY'
import csv, sys
sys.path.insert(0, "/root/repo/tools")
from kname import kernel_name
rows = []
for r in csv.DictReader(open(sys.argv[1])):
    n = kernel_name(r["Kernel_Name"])
    if n.startswith("orbx::"):
        rows.append((int(r["Start_Timestamp"]), n[6:].split("<")[0], r["Queue_Id"], r.get("Stream_Id", "-")))
rows.sort()
last = None
for t, n, q, s in rows:
    key = (n, q, s)
    if key != last:
        pass
    last = key
# mapping changes over time: print every change of (kernel -> queue, stream)
seen = {}
for t, n, q, s in rows:
    if seen.get(n) != (q, s):
        print("%12.3f ms  %-20s queue %s stream %s" % ((t - rows[0][0]) / 1e6, n, q, s))
        seen[n] = (q, s)
PY
cat $OUT/ctx_order.txt
