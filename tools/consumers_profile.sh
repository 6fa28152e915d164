#!/bin/bash
# Kernel durations of the stages around the extractor (`rocprofv3 --kernel-trace --stats`, one run per rate tool),
# merged into gpurun_out/<tag>_consumers_kernel_stats.csv.  On the GPU box through gpurun:
#   gpurun -- 'bash tools/consumers_profile.sh r02'
set -e
TAG=${1:-rXX}
G=$PWD/gpurun_out
export TMPDIR=/tmp
for t in bow_match_rate init_match_rate local_map_rate projection_rate frustum_rate; do
    timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $G/prof_c_$t -o trace -- python3 tools/$t.py > $G/prof_c_$t.log 2>&1
done
python3 - "$G" "$TAG" <<'PY'
import csv, glob, sys
g, tag = sys.argv[1], sys.argv[2]
rows = []
for t in ("bow_match_rate", "init_match_rate", "local_map_rate", "projection_rate", "frustum_rate"):
    f = glob.glob("%s/prof_c_%s/**/*kernel_stats*.csv" % (g, t), recursive=True)[0]
    for r in csv.DictReader(open(f)):
        name = r["Name"]
        if "orbx::" not in name or not any(k in name for k in ("k_search_", "k_bow_", "k_undistort_grid", "k_fuse", "k_stereo", "k_project_")):
            continue  # the extractor's own kernels run once, overlapped, as set-up of the workload
        rows.append((t, name.split("(")[0].replace("void ", ""), int(r["Calls"]), float(r["AverageNs"]) / 1e3,
                     float(r["MinNs"]) / 1e3, float(r["MaxNs"]) / 1e3))
    print(open("%s/prof_c_%s.log" % (g, t)).read().strip().splitlines()[-1])
with open("%s/%s_consumers_kernel_stats.csv" % (g, tag), "w") as o:
    o.write("# rocprofv3 --kernel-trace --stats over tools/<workload>.py (512 KITTI-shape frames or pairs per launch)\n")
    o.write("workload,kernel,calls,avg_us,min_us,max_us\n")
    for r in rows:
        o.write("%s,%s,%d,%.1f,%.1f,%.1f\n" % r)
PY
rm -rf $G/prof_c_*
cat $G/${TAG}_consumers_kernel_stats.csv
