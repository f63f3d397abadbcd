"""From the rocprofv3 --kernel-trace of a bench.py run: the average duration of the attention kernel over the run's TIMED
region = its last K dispatches (bench.py launches 1 + K unprimed + ~100 ms of priming + W warm-up + K timed steps), next to the
all-dispatch average that `--stats` prints (which mixes in the unprimed launches).
    python3 profiles/timed_region.py <stats dir, e.g. gpurun_out/r02_cfg2_stats> <K> > profiles/r02_kernel_timed_cfg2.json"""
import csv
import glob
import json
import os
import sys

csv.field_size_limit(1 << 30)
d, k = sys.argv[1], int(sys.argv[2])
path = glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True)[0]
rows = []
with open(path, newline="") as f:
    for r in csv.DictReader(f):
        if "fwd_mfma_" in r["Kernel_Name"] or "fwd_f32_mfma" in r["Kernel_Name"]:   # (fwd_mfma_kernel, fwd_mfma_dual_kernel)
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]) - int(r["Start_Timestamp"])))
rows.sort()
dur = [x for _, x in rows]
last = dur[-k:]
print(json.dumps({"source": os.path.relpath(path), "dispatches": len(dur), "avg_all_us": sum(dur) / len(dur) / 1e3,
                  "timed_region_dispatches": k, "timed_avg_us": sum(last) / k / 1e3, "timed_min_us": min(last) / 1e3,
                  "timed_max_us": max(last) / 1e3}, indent=1))
