"""The two-instance stereo pattern with HIP's 4 hardware queues per process and with 8."""
import sys, os, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bench
for q in ("4", "8"):
    os.environ["GPU_MAX_HW_QUEUES"] = q
    r = bench.stereo_pair_latency(300)
    print("hardware queues", q, {k: v for k, v in r.items() if k != "workload"}, flush=True)
